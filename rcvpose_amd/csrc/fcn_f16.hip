// fcn_f16.hip -- the FCN engine's 16-bit kernels and executor once more, with IEEE half (RCV_F16) as the storage type.
// fcn.hip is written against H16 / RCV_MFMA16; this unit only selects the other type.  The C ABI lives in fcn.hip.
#define RCV_FCN_FP16 1
#include "fcn.hip"
