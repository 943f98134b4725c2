// fcn_x3.hip -- the FCN engine's 16-bit kernels and executor a third time, as the split-precision engine RCV_F16X3: IEEE-half
// hi / lo operand pairs, three MFMAs per product, fp32 accumulation (fp32-grade maps at a third of the fp16 matrix rate).
// fcn.hip is written against H16 / RCV_MFMA16 / kX3; this unit only selects the mode.  The C ABI lives in fcn.hip.
#define RCV_FCN_X3 1
#include "fcn.hip"
