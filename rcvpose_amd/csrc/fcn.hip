// fcn.hip -- the FCN half of the RCVPose hot path on MI355X (gfx950 / CDNA4).
//
// A small native executor for the layer program of DenseFCNResNet152 (reference models/fcnresnet.py:48-191)
// plus the hand-written kernels it launches.  Activations are NHWC (channels innermost) so that the
// K dimension of every convolution -- (tap, channel) -- is contiguous for both operands:
//
//   k_conv<T,BM,BN>  implicit-GEMM convolution on the matrix cores, im2col-free:
//                    out[m, n] = act( sum_{tap,c} in[pixel(m) + tap, c] * w[n, tap, c] + bias[n] (+ res[m, n]) )
//                    M = B*Hout*Wout pixels, N = Cout, K = KH*KW*(C0+C1), walked 64-channel slice by slice with the taps
//                    innermost (L2 reuse of the shifted reads).  The A operand is gathered on the
//                    fly from one or two channel-concatenated sources (torch.cat is never materialised),
//                    zero-filled at the borders, staged through XOR-swizzled LDS tiles (16-byte chunk ^= (row>>1)&7:
//                    the 16 lanes of a ds_read_b128 group land on 16 distinct slots of the 256-byte bank row) with register
//                    double-buffering (global loads of tile t+1 are in flight while tile t is multiplied).
//                    T = H16 : v_mfma_f32_16x16x32_bf16, fp32 accumulate      (performance mode)
//                    T = float  : v_mfma_f32_16x16x4_f32, exact fp32 fma chain   (parity mode, 1e-4 gate)
//                    BatchNorm is folded into w / bias on the host; ReLU and the residual add are fused
//                    into the epilogue.  blockIdx.z walks the `groups` networks (one per keypoint) that
//                    share the input image.
//   k_conv_dma       the same GEMM for bf16 layers with Cout % 128 == 0: 256 x 128 x 64 tiles, operands by buffer-load
//                    LDS-DMA, four-phase K steps with the two waves of a SIMD staggered (see the kernel)
//   k_conv_halo      bf16: 3x3 / stride-1 layers whose image tiles into <= 256-pixel rectangles: the A operand is a halo'd pixel
//                    patch kept in LDS per channel slice (one load instead of nine tap tiles); bit-identical to k_conv_dma
//   k_conv_block     bf16: 3x3 + 1x1 expand (+ residual) of a bottleneck block, and where it follows the next block's 1x1
//                    reduce, fused in one kernel; staggered wave groups in the 1x1 phases of the 256-wide form; bit-identical
//   k_tail           bf16: last 2x upsampling + conv7 + head (conv8) fused; bit-identical to the separate kernels
//   k_stem_mfma      bf16: conv1 7x7/2 + BN + ReLU as an MFMA GEMM over an im2col tile shared by the networks
//   k_stem           the same layer computed directly from the NCHW fp32 image (fp32 mode)
//   k_maxpool, k_upsample2 (bilinear, align_corners=False), k_head (conv8 1x1 -> seg / radial planes; fp32 mode)
//   k_preprocess     FCResBackbone's uint8 -> normalised fp32 image (reference AccumulatorSpace.py:140-151)
#include <string.h>

#include <vector>

#include "common.hpp"

namespace rcv {
namespace {

// The 16-bit storage type of the engine.  This file is compiled twice: as is (bfloat16: RCV_BF16) and through fcn_f16.hip
// with RCV_FCN_FP16 defined (IEEE half: RCV_F16 -- 11 instead of 8 significand bits, the same MFMA rate; values beyond
// 65504 saturate to infinity, which BN-folded ResNet activations stay far away from).  Everything below is written against
// H16 / h16x8 / RCV_MFMA16; the kernels live in an anonymous namespace, so the two translation units do not collide.
#if defined(RCV_FCN_X3)
// Third compilation (fcn_x3.hip): the split-precision engine RCV_F16X3.  Every activation and weight is a PAIR of IEEE halves
// hi = half(v), lo = half(v - hi) (22 significand bits together), a product is three MFMAs hi*hi + hi*lo + lo*hi accumulated in
// fp32 (the lo*lo term is below 2^-22), i.e. fp32-grade results at a third of the fp16 matrix rate instead of a sixteenth
// (v_mfma_f32_16x16x4_f32).  Storage: NHWC with the channels in groups of 32, [32 hi | 32 lo] = one 128-byte K row of the
// GEMM tiles, so a tensor looks like a 16-bit tensor with twice the channels to every loader; weights likewise, scaled per
// output channel by a power of two (undone exactly in the epilogue) so that their lo halves stay normal numbers.
typedef _Float16 H16;
#define RCV_MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_f16
constexpr bool kX3 = true;
#elif defined(RCV_FCN_FP16)
typedef _Float16 H16;
#define RCV_MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_f16
constexpr bool kX3 = false;
#else
typedef __bf16 H16;
#define RCV_MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
constexpr bool kX3 = false;
#endif
typedef __attribute__((ext_vector_type(8))) H16 bf16x8;  // (name kept: eight 16-bit elements = one MFMA operand)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <typename T> struct Traits;
template <> struct Traits<float> {
    static constexpr int KE = 32;  // elements in one 128-byte K row of an LDS tile
    static constexpr int EPV = 4;  // elements per 16-byte vector
};
template <> struct Traits<H16> {
    static constexpr int KE = 64;
    static constexpr int EPV = 8;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(H16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ H16 from_f32<H16>(float v) { return (H16)v; }

// ---- split-precision storage (kX3): element (pixel, channel c) of a C-channel tensor lives at halves
// pixel * 2C + x3_off(c) (hi) and + 32 (lo)
template <typename T> constexpr bool is_x3() { return kX3 && sizeof(T) == 2; }
__device__ __forceinline__ int x3_off(int c) { return ((c >> 5) << 6) + (c & 31); }
// offset (in elements of T) of channel n of row m in an [M][C] NHWC tensor
template <typename T> __device__ __forceinline__ long long nhwc_off(long long m, int C, int n)
{
    if constexpr (is_x3<T>()) return m * (2LL * C) + x3_off(n);
    else return m * C + n;
}
// v -> (hi, lo); magnitudes beyond the half range saturate (65504 + 65504) instead of turning into inf - inf.  r04: the
// saturation is the hardware's -- every kernel of the split engine sets MODE.FP16_OVFL (RCV_X3_MODE), under which a float -> half
// conversion that overflows returns +-65504 instead of an infinity (tools/probe/ovfl_probe.hip: v_cvt_f16_f32 and
// v_cvt_pk_f16_f32 alike; a true infinity stays one) -- so the two v_med3_f32 per element that used to clamp by hand are gone
// (16 of the 36 VALU instructions of a split of eight).  Same values for every finite input.
#if defined(RCV_FCN_X3)
#define RCV_X3_MODE() __builtin_amdgcn_s_setreg(((1 - 1) << 11) | (23 << 6) | 1, 1)  /* hwreg(HW_REG_MODE, 23, 1) = 1: FP16_OVFL */
#else
#define RCV_X3_MODE()
#endif
__device__ __forceinline__ void x3_split(float v, H16 &hi, H16 &lo)
{
    hi = (H16)v;
    lo = (H16)(v - (float)hi);
}
// eight consecutive channels as two 16-byte vectors.  Written on pairs: gfx950 converts two floats to packed halves in one
// instruction (v_cvt_pk_f16_f32) and subtracts / adds two floats in one (v_pk_add_f32) -- 36 VALU instructions per eight
// elements instead of 56 + 8 to pack; element for element the same IEEE operations as x3_split.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// (-DRCV_X3_SCALAR_HELPERS: the element-by-element forms, for an A-B check of the outputs' hash -- tools/fcn_hash.py)
typedef _Float16 x3_h2 __attribute__((ext_vector_type(2)));
#if defined(RCV_FCN_X3) && !defined(RCV_X3_SCALAR_HELPERS)
// Two elements at a time.  r04: the conversions half -> float never stand alone -- v_fma_mix_f32 reads a half (either half of a
// register, op_sel) as an operand of a float FMA, so `(float)hi + (float)lo` is fma(hi, 1, lo) and `v - (float)hi` is fma(hi, -1, v),
// one instruction per element and one rounding like the convert-then-add forms: identical results over 10^9 operand pairs incl.
// subnormal halves (tools/probe/mix_probe.hip).  A split of eight: 16 VALU instructions instead of 20 (36 at the start of r04), a join: 8 (12).
// Where it pays is where instruction SLOTS are scarce (the chain kernels' epilogue pieces between MFMAs: -5 % on k_chain2_x3<256>);
// a loop that is bound by vector THROUGHPUT loses (k_tail_x3's patch build: 6.1 -> 7.4 ms -- the mixed FMA issues slower than a
// conversion), so MIX is a per-call-site choice and the stand-alone split / join of eight keep the conversions.
template <bool MIX = false>
__device__ __forceinline__ f32x2 x3_join2(unsigned wh, unsigned wl)  // (float)hi + (float)lo of the two elements of a register pair
{
    if constexpr (MIX) {
        f32x2 r;
        asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r.x) : "v"(wh), "v"(wl));
        asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r.y) : "v"(wh), "v"(wl));
        return r;
    } else
        return __builtin_convertvector(__builtin_bit_cast(x3_h2, wh), f32x2) + __builtin_convertvector(__builtin_bit_cast(x3_h2, wl), f32x2);
}
template <bool MIX = false>
__device__ __forceinline__ void x3_split2(float v0, float v1, unsigned &wh, unsigned &wl)  // two values -> packed hi halves, packed lo halves
{
    const f32x2 c = {v0, v1};
    const x3_h2 h = __builtin_convertvector(c, x3_h2);  // (saturating: RCV_X3_MODE)
    wh = __builtin_bit_cast(unsigned, h);
    f32x2 r;
    if constexpr (MIX) {
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r.x) : "v"(wh), "v"(v0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r.y) : "v"(wh), "v"(v1));
    } else
        r = c - __builtin_convertvector(h, f32x2);
    wl = __builtin_bit_cast(unsigned, __builtin_convertvector(r, x3_h2));
}
__device__ __forceinline__ void x3_split8(const float *v, uint4 &hi, uint4 &lo)
{
    unsigned *wh = (unsigned *)&hi, *wl = (unsigned *)&lo;
#pragma unroll
    for (int e = 0; e < 4; e++) x3_split2(v[2 * e], v[2 * e + 1], wh[e], wl[e]);
}
__device__ __forceinline__ void x3_join8(const uint4 &hi, const uint4 &lo, float *v)
{
    const unsigned *wh = (const unsigned *)&hi, *wl = (const unsigned *)&lo;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const f32x2 s = x3_join2(wh[e], wl[e]);
        v[2 * e] = s[0];
        v[2 * e + 1] = s[1];
    }
}
#else
// (the other translation units never run these; the chain kernels' templates are compiled there all the same)
template <bool MIX = false>
__device__ __forceinline__ f32x2 x3_join2(unsigned wh, unsigned wl)
{
    return __builtin_convertvector(__builtin_bit_cast(x3_h2, wh), f32x2) + __builtin_convertvector(__builtin_bit_cast(x3_h2, wl), f32x2);
}
template <bool MIX = false>
__device__ __forceinline__ void x3_split2(float v0, float v1, unsigned &wh, unsigned &wl)
{
    const f32x2 c = {v0, v1};
    const x3_h2 h = __builtin_convertvector(c, x3_h2);
    wh = __builtin_bit_cast(unsigned, h);
    wl = __builtin_bit_cast(unsigned, __builtin_convertvector(c - __builtin_convertvector(h, f32x2), x3_h2));
}
__device__ __forceinline__ void x3_split8(const float *v, uint4 &hi, uint4 &lo)
{
    H16 *ph = (H16 *)&hi, *pl = (H16 *)&lo;
#pragma unroll
    for (int e = 0; e < 8; e++) x3_split(v[e], ph[e], pl[e]);
}
__device__ __forceinline__ void x3_join8(const uint4 &hi, const uint4 &lo, float *v)
{
    const H16 *ph = (const H16 *)&hi, *pl = (const H16 *)&lo;
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = (float)ph[e] + (float)pl[e];
}
#endif

// ---- t3: the 3-byte form of a split-precision tensor (r06, engine option 14).  A bottleneck stage's trunk -- the 4*CMID-channel block
// output -- is written once and read once, by the next block's chain kernel as its RESIDUAL: a VALU path (every other consumer takes its
// operand by LDS-DMA straight into the MFMA layout and needs the [32 hi | 32 lo] rows).  Between two chain kernels the tensor therefore
// travels as sign / 5-bit exponent / 18-bit mantissa: per 32-channel group 64 bytes of top halves (an IEEE half, truncated) + 32 bytes
// of the next eight mantissa bits = 96 instead of 128 bytes per group and pixel, rounded to nearest even from the fp32 value (19
// significant bits; the [hi | lo] pair carries 22).  The chains of layers 1-2 are HBM-bound: -20 % of their bytes.  Emulated on the
// CPU (tools/probe/storage_bits_probe.py --stages): maps 3.8-7.8e-6 from the fp32 forward against 3.4-7.3e-6 without.
__device__ __forceinline__ float t3_decode(unsigned top16, unsigned low8)
{
    const unsigned bits = ((top16 & 0x8000u) << 16) | ((top16 & 0x7fffu) << 13) | (low8 << 5);
    return __uint_as_float(bits) * 0x1p112f;  // (the half's exponent field sits in the float's: rebias; subnormal halves come out exact)
}
__device__ __forceinline__ void t3_encode(float v, unsigned &top16, unsigned &low8)
{
    const unsigned b = __float_as_uint(v * 0x1p-112f);
    unsigned m = b & 0x7fffffffu;
    m = m + 0xFu + ((m >> 5) & 1u);      // nearest even at bit 5
    m = min(m, 0x0F7FFFE0u);             // the format's largest finite value (65535.875): saturate like the [hi | lo] split does
    top16 = ((b >> 16) & 0x8000u) | (m >> 13);
    low8 = (m >> 5) & 0xffu;
}

// ---- range guard of the half-precision stores (RCV_F16X3 and RCV_F16).  A magnitude beyond 65504 is clamped by x3_split (the
// stored pair is finite and wrong) or becomes an infinity in RCV_F16: every epilogue that rounds to halves keeps the largest
// magnitude it stored in one register (v_max3_f32 with |.| modifiers: four instructions per eight values) and ORs a bit into the
// engine's sticky flag word at its end when that exceeded the half range -- rcv_fcn_forward / rcv_fcn_range_status report it as
// RCV_E_RANGE.  bf16 has fp32's range: no guard there.
#if defined(RCV_FCN_X3) || defined(RCV_FCN_FP16)
constexpr bool kHalfRange = true;
#else
constexpr bool kHalfRange = false;
#endif
constexpr unsigned RNG_BIT = kX3 ? 1u : 2u;  // bit 0: an f16x3 pair was clamped, bit 1: an RCV_F16 store overflowed to infinity
template <int N> __device__ __forceinline__ float rng_see(float m, const float *v)
{
    if constexpr (kHalfRange) {
#pragma unroll
        for (int e = 0; e + 1 < N; e += 2) m = fmaxf(fmaxf(m, fabsf(v[e])), fabsf(v[e + 1]));
        if constexpr (N & 1) m = fmaxf(m, fabsf(v[N - 1]));
    }
    return m;
}
__device__ __forceinline__ void rng_flush(unsigned *flag, float m)
{
    if constexpr (kHalfRange) {
        if (m > 65504.f && flag) atomicOr(flag, RNG_BIT);
    }
}

// ---- q6: the fp6 (e2m3) side band of the split-precision engine (r05).  A 128-byte row of a q6 tensor keeps the 32 hi halves of
// its channel group and, in place of the 32 lo halves, four 16-byte chunks = the blocks of v_mfma_scale_f32_16x16x128_f8f6f4:
// 16 e2m3 codes (12 bytes, element j in bits [6j, 6j + 6)) of hi[0:16], hi[16:32], lo[0:16], lo[16:32] and the block's e8m0
// scale in byte 12 (activations) / byte 14 (weights).  Block scale = 2^(floor(log2 max|block|) - 2): the largest element lands
// in e2m3's top binade [4, 8) (7.75 and above saturate at 7.5); conversion = v_cvt_scalef32_pk32_fp6_f16 (round to nearest
// even, saturating; tools/probe/fp6_mfma_probe.hip).
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
__device__ __forceinline__ f32x4 q6_mfma(i32x8 a, i32x8 b, f32x4 c, int scale_a, int scale_b)
{
#if defined(RCV_FCN_X3)
    // cbsz = blgp = 2: both operands e2m3; the A scale is byte 0 of its register, the B scale byte 2
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2, 2, 0, scale_a, 2, scale_b);
#else
    return c;
#endif
}
struct ConvParams {
    const void *src0, *src1, *res, *w;
    void *dst;
    const float *bias;
    long long src0_gs, src1_gs, res_gs, dst_gs, w_gs;  // per-group strides in elements
    int bias_gs;
    int C0, C1, Cout;
    int Hin, Win, Hout, Wout;
    int KS, stride, pad, relu;
    int M, K;  // GEMM sizes per group
    unsigned *range;  // the engine's sticky range-flag word (half-precision engines), or null
    int q6;           // split engine: src0 and w are in the q6 row layout, the cross terms run in fp6 (engine option 11)
    int splits;       // k_conv_lat (engine option 12): the K steps are dealt over `splits` workgroups per output tile (blockIdx.z = group * splits + split),
    float *partial;   // ... which leave their raw fp32 sums in partial[group][split][M][Cout] for k_splitk_finish; 0 / null: off
};

// Tile order.  Blocks are numbered so that the N tiles of one M tile are neighbours (they share the gathered
// A rows, the big operand) and -- blocks being dealt round-robin over the 8 XCDs -- so that neighbours in that
// order land on the SAME XCD and meet in its L2 (bijective XCD remap, cdna_hip_programming.md T1).  Placement only
// affects speed.
__device__ __forceinline__ void tile_of_block(int n_tiles, int &mt, int &nt)
{
    const unsigned total = gridDim.x * gridDim.y;
    const unsigned id = blockIdx.x + gridDim.x * blockIdx.y;
    const unsigned q = total / 8, r = total % 8, xcd = id % 8;
    const unsigned s = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + id / 8;
    nt = (int)(s % (unsigned)n_tiles);
    mt = (int)(s / (unsigned)n_tiles);
}

// ---------------------------------------------------------------------------------------------
// implicit-GEMM convolution
// ---------------------------------------------------------------------------------------------
template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void k_conv(const ConvParams p)
{
    RCV_X3_MODE();
    constexpr int KE = Traits<T>::KE, EPV = Traits<T>::EPV;
    constexpr int AR = BM / 32, BR = BN / 32;  // 16-byte rows each thread stages per tile
    constexpr int MT = BM / 32, NT = BN / 32;  // 16x16 MFMA tiles per wave (waves are 2 x 2)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *As = smem;                   // [2][BM][128 B]
    unsigned char *Bs = smem + 2 * BM * 128;    // [2][BN][128 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int mt_, nt_;
    tile_of_block(p.Cout / BN, mt_, nt_);
    const int g = blockIdx.z, m0 = mt_ * BM, n0 = nt_ * BN;
    const int chunk = tid & 7, rowb = tid >> 3;
    const int Ctot = p.C0 + p.C1;
    const T *src0 = (const T *)p.src0 + (long long)g * p.src0_gs;
    const T *src1 = p.src1 ? (const T *)p.src1 + (long long)g * p.src1_gs : nullptr;
    const T *wg = (const T *)p.w + (long long)g * p.w_gs;

    int a_hi0[AR], a_wi0[AR], a_pix[AR];
    const int HWo = p.Hout * p.Wout;
#pragma unroll
    for (int q = 0; q < AR; q++) {
        const int m = m0 + rowb + 32 * q;
        if (m < p.M) {
            const int img = m / HWo, rem = m - img * HWo;
            const int ho = rem / p.Wout, wo = rem - ho * p.Wout;
            a_hi0[q] = ho * p.stride - p.pad;
            a_wi0[q] = wo * p.stride - p.pad;
            a_pix[q] = img * p.Hin * p.Win;
        } else {
            a_hi0[q] = -100000;  // every tap falls outside -> zeros
            a_wi0[q] = -100000;
            a_pix[q] = 0;
        }
    }

    uint4 ra0[AR], rb0[BR], ra1[AR], rb1[BR];  // two register sets: global loads run two K tiles ahead
    int tap_r = 0, tap_s = 0, cg = 0;  // position of the next K chunk: tap (r, s) and concat channel

// global -> registers for the next K chunk (A gathered with zero fill at the borders, B = weights)
#define RCV_FETCH(RA, RB, LIVE)                                                                                        \
    {                                                                                                        \
        const T *src_;                                                                                       \
        int Cs_, c_;                                                                                         \
        if (cg < p.C0) { src_ = src0; Cs_ = p.C0; c_ = cg; } else { src_ = src1; Cs_ = p.C1; c_ = cg - p.C0; } \
        const int kcol_ = (tap_r * p.KS + tap_s) * Ctot + cg + chunk * EPV;                                  \
        _Pragma("unroll") for (int q = 0; q < AR; q++) {                                                     \
            const int hi_ = a_hi0[q] + tap_r, wi_ = a_wi0[q] + tap_s;                                        \
            uint4 v_ = make_uint4(0u, 0u, 0u, 0u);                                                           \
            if ((LIVE) && (unsigned)hi_ < (unsigned)p.Hin && (unsigned)wi_ < (unsigned)p.Win)                \
                v_ = *(const uint4 *)(src_ + ((long long)(a_pix[q] + hi_ * p.Win + wi_) * Cs_ + c_ + chunk * EPV)); \
            RA[q] = v_;                                                                                      \
        }                                                                                                    \
        _Pragma("unroll") for (int q = 0; q < BR; q++) {                                                     \
            uint4 w_ = make_uint4(0u, 0u, 0u, 0u);                                                           \
            if (LIVE) w_ = *(const uint4 *)(wg + (long long)(n0 + rowb + 32 * q) * p.K + kcol_);             \
            RB[q] = w_;                                                                                      \
        }                                                                                                    \
        if (++tap_s == p.KS) { tap_s = 0; if (++tap_r == p.KS) { tap_r = 0; cg += KE; } }                    \
    }
// registers -> XOR-swizzled LDS tile
#define RCV_STASH(buf, RA, RB)                                                                                     \
    {                                                                                                        \
        _Pragma("unroll") for (int q = 0; q < AR; q++) {                                                     \
            const int row_ = rowb + 32 * q;                                                                  \
            *(uint4 *)(As + (buf) * BM * 128 + row_ * 128 + ((chunk ^ ((row_ >> 1) & 7)) << 4)) = RA[q];            \
        }                                                                                                    \
        _Pragma("unroll") for (int q = 0; q < BR; q++) {                                                     \
            const int row_ = rowb + 32 * q;                                                                  \
            *(uint4 *)(Bs + (buf) * BN * 128 + row_ * 128 + ((chunk ^ ((row_ >> 1) & 7)) << 4)) = RB[q];            \
        }                                                                                                    \
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int r16 = lane & 15, kq = lane >> 4;
// one K tile of MFMAs out of LDS buffer `buf`
#define RCV_COMPUTE(buf)                                                                                     \
    {                                                                                                        \
        const unsigned char *Ab = As + (buf) * BM * 128 + (wm * (BM / 2) + r16) * 128;                       \
        const unsigned char *Bb = Bs + (buf) * BN * 128 + (wn * (BN / 2) + r16) * 128;                       \
        if constexpr (is_x3<T>()) {  /* [32 hi | 32 lo] K row: hi*hi, hi*lo, lo*hi */                    \
            bf16x8 ah[MT], al[MT], bh[NT], bl[NT];                                                           \
            const int chh = (kq ^ ((r16 >> 1) & 7)) << 4, chl = ((4 + kq) ^ ((r16 >> 1) & 7)) << 4;         \
            _Pragma("unroll") for (int i = 0; i < MT; i++) { ah[i] = *(const bf16x8 *)(Ab + i * 16 * 128 + chh); al[i] = *(const bf16x8 *)(Ab + i * 16 * 128 + chl); } \
            _Pragma("unroll") for (int j = 0; j < NT; j++) { bh[j] = *(const bf16x8 *)(Bb + j * 16 * 128 + chh); bl[j] = *(const bf16x8 *)(Bb + j * 16 * 128 + chl); } \
            _Pragma("unroll") for (int i = 0; i < MT; i++)                                                   \
                _Pragma("unroll") for (int j = 0; j < NT; j++) acc[i][j] = RCV_MFMA16(ah[i], bh[j], acc[i][j], 0, 0, 0); \
            if (p.q6) {  /* q6 rows: the second halves are fp6 blocks -- one scaled MFMA for hi*lo + lo*hi (see k_conv_halo) */ \
                _Pragma("unroll") for (int j = 0; j < NT; j++) {                                             \
                    const i32x4 bq_ = __builtin_bit_cast(i32x4, bl[j]);                                      \
                    const i32x8 b8_ = {bq_[0], bq_[1], bq_[2], bq_[3], 0, 0, 0, 0};                          \
                    _Pragma("unroll") for (int i = 0; i < MT; i++) {                                         \
                        const i32x4 aq_ = __builtin_bit_cast(i32x4, al[i]);                                  \
                        acc[i][j] = q6_mfma(__builtin_shufflevector(aq_, aq_, 0, 1, 2, 3, -1, -1, -1, -1), b8_, acc[i][j], aq_[3], bq_[3]); \
                    }                                                                                        \
                }                                                                                            \
            } else {                                                                                         \
                _Pragma("unroll") for (int i = 0; i < MT; i++)                                               \
                    _Pragma("unroll") for (int j = 0; j < NT; j++) acc[i][j] = RCV_MFMA16(ah[i], bl[j], acc[i][j], 0, 0, 0); \
                _Pragma("unroll") for (int i = 0; i < MT; i++)                                               \
                    _Pragma("unroll") for (int j = 0; j < NT; j++) acc[i][j] = RCV_MFMA16(al[i], bh[j], acc[i][j], 0, 0, 0); \
            }                                                                                                \
        } else if constexpr (sizeof(T) == 2) {                                                               \
            _Pragma("unroll") for (int kk = 0; kk < 2; kk++) {                                               \
                bf16x8 af[MT], bfr[NT];                                                                      \
                const int ch = ((kk * 4 + kq) ^ ((r16 >> 1) & 7)) << 4;                                             \
                _Pragma("unroll") for (int i = 0; i < MT; i++) af[i] = *(const bf16x8 *)(Ab + i * 16 * 128 + ch);  \
                _Pragma("unroll") for (int j = 0; j < NT; j++) bfr[j] = *(const bf16x8 *)(Bb + j * 16 * 128 + ch); \
                _Pragma("unroll") for (int i = 0; i < MT; i++)                                               \
                    _Pragma("unroll") for (int j = 0; j < NT; j++)                                           \
                        acc[i][j] = RCV_MFMA16(af[i], bfr[j], acc[i][j], 0, 0, 0); \
            }                                                                                                \
        } else {                                                                                             \
            _Pragma("unroll") for (int ks = 0; ks < 8; ks++) {                                               \
                float af[MT], bfr[NT];                                                                       \
                const int ch = ((ks ^ ((r16 >> 1) & 7)) << 4) + kq * 4;                                             \
                _Pragma("unroll") for (int i = 0; i < MT; i++) af[i] = *(const float *)(Ab + i * 16 * 128 + ch);  \
                _Pragma("unroll") for (int j = 0; j < NT; j++) bfr[j] = *(const float *)(Bb + j * 16 * 128 + ch); \
                _Pragma("unroll") for (int i = 0; i < MT; i++)                                               \
                    _Pragma("unroll") for (int j = 0; j < NT; j++)                                           \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0); \
            }                                                                                                \
        }                                                                                                    \
    }

    const int nk = p.K / KE;
    // software pipeline, two tiles deep: while tile t is multiplied out of LDS, tile t+1 sits in one register
    // set (stored to the other LDS buffer after the MFMAs) and the loads of tile t+2 are in flight in the other
    RCV_FETCH(ra0, rb0, true);
    RCV_FETCH(ra1, rb1, nk > 1);
    RCV_STASH(0, ra0, rb0);
    __syncthreads();
    int kt = 0;
    for (; kt + 2 <= nk - 1; kt += 2) {
        RCV_FETCH(ra0, rb0, true);      // tile kt+2
        RCV_COMPUTE(0);                 // tile kt
        RCV_STASH(1, ra1, rb1);         // tile kt+1
        __syncthreads();
        RCV_FETCH(ra1, rb1, kt + 3 < nk);  // tile kt+3 (predicated off past the end)
        RCV_COMPUTE(1);                 // tile kt+1
        RCV_STASH(0, ra0, rb0);         // tile kt+2
        __syncthreads();
    }
    // tail: kt is even; tiles kt (in LDS buffer 0) and possibly kt+1 (in register set 1) remain
    RCV_COMPUTE(0);
    if (kt + 1 < nk) {
        RCV_STASH(1, ra1, rb1);
        __syncthreads();
        RCV_COMPUTE(1);
    }
#undef RCV_FETCH
#undef RCV_STASH
#undef RCV_COMPUTE

    // epilogue: the accumulators (C/D layout: col = lane & 15, row = (lane >> 4) * 4 + e) go through LDS so that
    // bias + residual + ReLU and the NHWC store run on whole 16-byte vectors of contiguous channels.
    // Two rounds (one per wave row) keep the fp32 staging tile within the K-loop's LDS footprint.
    T *dst = (T *)p.dst + (long long)g * p.dst_gs;
    const T *res = p.res ? (const T *)p.res + (long long)g * p.res_gs : nullptr;
    const float *bias = p.bias + (long long)g * p.bias_gs;
    constexpr int SROW = BN + 4;                 // padded fp32 row: the 4 row groups of a store hit disjoint banks
    constexpr int CPR = BN / EPV;                // 16-byte output chunks per tile row
    constexpr int CHUNKS = (BM / 2) * CPR;
    constexpr int PER = (CHUNKS + 255) / 256;    // chunks per thread per round
    float *stage = (float *)smem;
    // the residual tile is fetched for both rounds up front: its latency hides behind the LDS staging
    constexpr int RV = is_x3<T>() ? 2 : 1;       // 16-byte vectors per chunk (split storage: hi and lo)
    uint4 rres[2][PER][RV];
    if (res) {
#pragma unroll
        for (int round = 0; round < 2; round++)
#pragma unroll
            for (int q = 0; q < PER; q++) {
                const int c = q * 256 + tid, row = c / CPR, cc = c - row * CPR;
                const int m = m0 + round * (BM / 2) + row;
                const bool live = c < CHUNKS && m < p.M;
                const T *rp = res + nhwc_off<T>(m, p.Cout, n0 + cc * EPV);
#pragma unroll
                for (int h = 0; h < RV; h++) rres[round][q][h] = live ? *(const uint4 *)(rp + 32 * h) : make_uint4(0u, 0u, 0u, 0u);
            }
    }
    float rmax = 0.f;  // largest magnitude stored (range guard of the half-precision engines)
#pragma unroll
    for (int round = 0; round < 2; round++) {
        __syncthreads();  // K loop (round 0) / previous round's readers are done with LDS
        if (wm == round) {
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const int nl = wn * (BN / 2) + j * 16 + r16;
                const float bv = bias[n0 + nl];
                const float sc = is_x3<T>() ? bias[p.Cout + n0 + nl] : 1.f;  // (split mode: the weights' power-of-two scale, undone here)
#pragma unroll
                for (int i = 0; i < MT; i++)
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        stage[(i * 16 + kq * 4 + e) * SROW + nl] = is_x3<T>() ? __fmaf_rn(acc[i][j][e], sc, bv) : acc[i][j][e] + bv;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int c = q * 256 + tid, row = c / CPR, cc = c - row * CPR;
            const int m = m0 + round * (BM / 2) + row;
            if (c >= CHUNKS || m >= p.M) continue;
            const long long o = nhwc_off<T>(m, p.Cout, n0 + cc * EPV);
            float v[EPV];
            const float4 *sp4 = (const float4 *)(stage + row * SROW + cc * EPV);
#pragma unroll
            for (int qq = 0; qq < EPV / 4; qq++) {
                const float4 t = sp4[qq];
                v[qq * 4 + 0] = t.x; v[qq * 4 + 1] = t.y; v[qq * 4 + 2] = t.z; v[qq * 4 + 3] = t.w;
            }
            if (res) {
                if constexpr (is_x3<T>()) {
                    float rv[8];
                    x3_join8(rres[round][q][0], rres[round][q][RV - 1], rv);
#pragma unroll
                    for (int qq = 0; qq < EPV; qq++) v[qq] += rv[qq];
                } else {
                    const T *rp = (const T *)&rres[round][q][0];
#pragma unroll
                    for (int qq = 0; qq < EPV; qq++) v[qq] += to_f32(rp[qq]);
                }
            }
            if (p.relu) {
#pragma unroll
                for (int qq = 0; qq < EPV; qq++) v[qq] = fmaxf(v[qq], 0.f);
            }
            if constexpr (sizeof(T) == 2) rmax = rng_see<EPV>(rmax, v);
            if constexpr (is_x3<T>()) {
                uint4 oh, ol;
                x3_split8(v, oh, ol);
                *(uint4 *)(dst + o) = oh;
                *(uint4 *)(dst + o + 32) = ol;
            } else {
                uint4 outv;
                T *op = (T *)&outv;
#pragma unroll
                for (int qq = 0; qq < EPV; qq++) op[qq] = from_f32<T>(v[qq]);
                *(uint4 *)(dst + o) = outv;
            }
        }
    }
    if constexpr (sizeof(T) == 2) rng_flush(p.range, rmax);
}

// ---------------------------------------------------------------------------------------------
// k_conv_dma: the same implicit GEMM with the operand tiles brought in by LDS-DMA (buffer_load_dwordx4 ... lds,
// 16 bytes per lane, 1 KB per wave-instruction) instead of through registers.  bf16 only.
//   tile 256 x 128 x 64, 512 threads = 4 x 2 waves of 64 x 64, three LDS stages of 48 KB: the pieces of K step
//   kt+2 are issued during the two load phases of step kt and stay in flight across barriers (counted
//   s_waitcnt vmcnt, raw s_barrier); no staging VGPRs, no ds_write.
//   The LDS image is lane-linear per DMA ([8 rows][8 x 16 B]); the bank swizzle is applied to the per-lane
//   SOURCE chunk (chunk ^= (row>>1)&7) and undone by the same XOR on the fragment reads.
//   Padding taps / rows beyond M use an out-of-range buffer offset: the hardware writes zeros.
// ---------------------------------------------------------------------------------------------
constexpr int DMA_BM = 256, DMA_BN = 128, DMA_STAGES = 3;
constexpr int DMA_STAGE_BYTES = (DMA_BM + DMA_BN) * 128;

__global__ __launch_bounds__(512) void k_conv_dma(const ConvParams p)
{
    RCV_X3_MODE();
    using T = H16;
    constexpr int KE = 64, EPV = 8, BM = DMA_BM, BN = DMA_BN;
    constexpr int MT = 4, NT = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int mt_, nt_;
    tile_of_block(p.Cout / BN, mt_, nt_);
    const int g = blockIdx.z, m0 = mt_ * BM, n0 = nt_ * BN;
    const int Ctot = p.C0 + p.C1;
    const T *src0 = (const T *)p.src0 + (long long)g * p.src0_gs;
    const T *src1 = p.src1 ? (const T *)p.src1 + (long long)g * p.src1_gs : nullptr;
    const T *wg = (const T *)p.w + (long long)g * p.w_gs;

    // the residual tile (if any) is requested before anything else: for the short-K expand convolutions the
    // kernel is a stream of (operand tile, residual tile, output tile) and this overlaps the first two
    const T *res = p.res ? (const T *)p.res + (long long)g * p.res_gs : nullptr;
    constexpr int E_CPR = DMA_BN / 8, E_PER = (64 * E_CPR) / 512;
    constexpr int RV = kX3 ? 2 : 1;  // 16-byte vectors per 8-channel chunk (split storage: hi and lo)
    uint4 rres[4][E_PER][RV];
// (split mode: the K loop holds 64 more fragment registers and the residual tile is twice as wide -- it is requested after the
// loop instead, behind the LDS staging of the accumulators)
#define RCV_DMA_RESIDUAL()                                                                                     \
    if (res) {                                                                                                 \
        _Pragma("unroll") for (int round = 0; round < 4; round++)                                              \
            _Pragma("unroll") for (int q = 0; q < E_PER; q++) {                                                \
                const int c = q * 512 + tid, row = c / E_CPR, cc = c - row * E_CPR;                            \
                const int m = m0 + round * 64 + row;                                                           \
                const T *rp = res + nhwc_off<T>(m, p.Cout, n0 + cc * 8);                                       \
                _Pragma("unroll") for (int h = 0; h < RV; h++)                                                 \
                    rres[round][q][h] = m < p.M ? *(const uint4 *)(rp + 32 * h) : make_uint4(0u, 0u, 0u, 0u);  \
            }                                                                                                  \
    }
    if constexpr (!kX3) RCV_DMA_RESIDUAL();

    // loader role of this lane: 4 A rows (wave*32 + c*8 + lane/8) and 2 B rows (wave*16 + c*8 + lane/8), 16-byte
    // slot lane%8 of the LDS row, which receives the source chunk slot ^ ((row>>1)&7).
    // The pieces are buffer loads (buffer_load_dwordx4 ... lds): the per-lane byte offset of a row (its own input
    // pixel, its channel chunk) is fixed for the whole tile, the tap / channel-slice displacement of a K step goes
    // into the (wave-uniform) base address of the buffer descriptor, i.e. is scalar arithmetic, and padding taps or
    // rows beyond M get an out-of-range offset, for which the hardware returns zeros.  Per piece that leaves three
    // VALU instructions (test the tap bit, select), cheap enough to sit between MFMAs.
    const int lrow = lane >> 3, lpos = lane & 7;
    constexpr int OOB = (int)0x80000000;
    int a_off0[4], a_off1[4];
    unsigned a_ok[4];
    const int HWo = p.Hout * p.Wout;
    // (image, row, column) of the tile's first output pixel by integer division (once, wave-uniform); the 256 rows
    // of the tile are then reached with two small float divisions each ((t + 0.5) * (1/d) is exact far beyond the
    // range 0 .. 256 + Wout that occurs here) instead of eight ~40-instruction integer divisions per lane
    const int img0 = m0 / HWo, rem0 = m0 - img0 * HWo;
    const int ho0 = rem0 / p.Wout, wo0 = rem0 - ho0 * p.Wout;
    const float rW = 1.0f / (float)p.Wout, rH = 1.0f / (float)p.Hout;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int row = wave * 32 + c * 8 + lrow;
        const int chunk = (lpos ^ ((row >> 1) & 7)) * EPV;
        a_off0[c] = OOB; a_off1[c] = OOB; a_ok[c] = 0u;
        if (m0 + row < p.M) {
            const int t = wo0 + row;
            const int q1 = (int)(((float)t + 0.5f) * rW);
            const int wo = t - q1 * p.Wout, h = ho0 + q1;
            const int q2 = (int)(((float)h + 0.5f) * rH);
            const int ho = h - q2 * p.Hout, img = img0 + q2;
            const int hc = ho * p.stride, wc = wo * p.stride;  // the pixel under tap (pad, pad): always inside the image
            const int lin = (img * p.Hin + hc) * p.Win + wc;
            a_off0[c] = (lin * p.C0 + chunk) * 2;
            a_off1[c] = (lin * p.C1 + chunk) * 2;
            // taps inside the image: rows r with 0 <= hc - pad + r < Hin, columns q likewise -> bit r * KS + q
            const int rlo = max(p.pad - hc, 0), rhi = min(p.Hin - 1 - hc + p.pad, p.KS - 1);
            const int qlo = max(p.pad - wc, 0), qhi = min(p.Win - 1 - wc + p.pad, p.KS - 1);
            const unsigned colmask = qhi >= qlo ? ((2u << qhi) - 1u) & ~((1u << qlo) - 1u) : 0u;
            for (int r = rlo; r <= rhi; r++) a_ok[c] |= colmask << (r * p.KS);
        }
    }
    int b_off[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int row = wave * 16 + c * 8 + lrow;
        b_off[c] = (row * p.K + (lpos ^ ((row >> 1) & 7)) * EPV) * 2;
    }
    const char *b_base = (const char *)(wg + (long long)n0 * p.K);
    const int n_img = __builtin_amdgcn_readfirstlane(p.M / HWo);  // (uniform, but the division runs on the VALU)
    const int a_bytes0 = (int)((long long)p.C0 * p.Hin * p.Win * n_img * 2), a_bytes1 = (int)((long long)p.C1 * p.Hin * p.Win * n_img * 2);
    const int b_bytes = BN * p.K * 2;
    int tap_r = 0, tap_s = 0, cg = 0;

// the pieces [LO, HI) of the next K step (0..3: A rows, 4..5: B rows) into `stage`; ADV: advance the K position
#define RCV_DMA_ISSUE(stage, LO, HI, ADV)                                                                      \
    {                                                                                                          \
        unsigned char *As_ = smem + (stage) * DMA_STAGE_BYTES;                                                 \
        unsigned char *Bs_ = As_ + BM * 128;                                                                   \
        const bool first_ = cg < p.C0;                                                                         \
        const int Cs_ = first_ ? p.C0 : p.C1, c_ = first_ ? cg : cg - p.C0;                                    \
        const int tap_ = tap_r * p.KS + tap_s;                                                                 \
        if ((LO) < 4) {                                                                                        \
            const long long disp_ = ((long long)((tap_r - p.pad) * p.Win + (tap_s - p.pad)) * Cs_ + c_) * 2;  \
            const char *abase_ = (const char *)(first_ ? src0 : src1) + disp_;                                 \
            const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc((void *)abase_, 0, first_ ? a_bytes0 : a_bytes1, 0x00020000); \
            _Pragma("unroll") for (int c = (LO); c < ((HI) < 4 ? (HI) : 4); c++) {                             \
                const int off_ = ((a_ok[c] >> tap_) & 1u) ? (first_ ? a_off0[c] : a_off1[c]) : OOB;            \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_, (__attribute__((address_space(3))) void *)(As_ + (wave * 32 + c * 8) * 128), 16, off_, 0, 0, 0); \
            }                                                                                                  \
        }                                                                                                      \
        if ((HI) > 4) {                                                                                        \
            const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc((void *)(b_base + (long long)(tap_ * Ctot + cg) * 2), 0, b_bytes, 0x00020000); \
            _Pragma("unroll") for (int c = ((LO) > 4 ? (LO) - 4 : 0); c < (HI) - 4; c++)                       \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_, (__attribute__((address_space(3))) void *)(Bs_ + (wave * 16 + c * 8) * 128), 16, b_off[c], 0, 0, 0); \
        }                                                                                                      \
        if (ADV) {                                                                                             \
            tap_s += 1;                                                                                        \
            const bool wrap_ = tap_s == p.KS;                                                                  \
            tap_s = wrap_ ? 0 : tap_s;                                                                         \
            tap_r += wrap_ ? 1 : 0;                                                                            \
            const bool wrap2_ = tap_r == p.KS;                                                                 \
            tap_r = wrap2_ ? 0 : tap_r;                                                                        \
            cg += wrap2_ ? KE : 0;                                                                             \
        }                                                                                                      \
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, kq = lane >> 4;
    const int ch0 = ((0 * 4 + kq) ^ ((r16 >> 1) & 7)) << 4, ch1 = ((1 * 4 + kq) ^ ((r16 >> 1) & 7)) << 4;

// A K step is four phases separated by workgroup barriers: load (fragments of the first 32 K columns + three DMA
// pieces of step kt+2), 16 MFMAs, load (second 32 columns + the other three pieces), 16 MFMAs.  The two waves of
// a SIMD (wave w and w+4) run the same sequence one barrier apart, so that on every SIMD one wave is in its
// MFMA phase while the other issues LDS reads and DMA pieces (and their waits).
//   data of step kt+1 is waited for (counted vmcnt) at the end of the second load phase of step kt by every wave,
//   which is at least one barrier before any wave reads it; the stage that step kt+2 goes to was last read in the
//   second load phase of step kt-1, at least one barrier before the first piece is issued.
#define RCV_DMA_STEP(ISSUE)                                                                                    \
    {                                                                                                          \
        const unsigned char *Ab = smem + stage * DMA_STAGE_BYTES + (wm * 64 + r16) * 128;                      \
        const unsigned char *Bb = smem + stage * DMA_STAGE_BYTES + BM * 128 + (wn * 64 + r16) * 128;           \
        const int st2 = stage + 2 >= DMA_STAGES ? stage + 2 - DMA_STAGES : stage + 2;                          \
        bf16x8 af[MT], bfr[NT];                                                                                \
        _Pragma("unroll") for (int i = 0; i < MT; i++) af[i] = *(const bf16x8 *)(Ab + i * 16 * 128 + ch0);     \
        _Pragma("unroll") for (int j = 0; j < NT; j++) bfr[j] = *(const bf16x8 *)(Bb + j * 16 * 128 + ch0);    \
        if (ISSUE) RCV_DMA_ISSUE(st2, 0, 3, false);                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
        __builtin_amdgcn_s_barrier();                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; i++)                                                         \
            _Pragma("unroll") for (int j = 0; j < NT; j++)                                                     \
                acc[i][j] = RCV_MFMA16(af[i], bfr[j], acc[i][j], 0, 0, 0);        \
        __builtin_amdgcn_s_setprio(0);                                                                         \
        __builtin_amdgcn_s_barrier();                                                                          \
        _Pragma("unroll") for (int i = 0; i < MT; i++) af[i] = *(const bf16x8 *)(Ab + i * 16 * 128 + ch1);     \
        _Pragma("unroll") for (int j = 0; j < NT; j++) bfr[j] = *(const bf16x8 *)(Bb + j * 16 * 128 + ch1);    \
        if (ISSUE) {                                                                                           \
            RCV_DMA_ISSUE(st2, 3, 6, true);                                                                    \
            asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");                                        \
        } else {                                                                                               \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                        \
        }                                                                                                      \
        __builtin_amdgcn_s_barrier();                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; i++)                                                         \
            _Pragma("unroll") for (int j = 0; j < NT; j++)                                                     \
                acc[i][j] = RCV_MFMA16(af[i], bfr[j], acc[i][j], 0, 0, 0);        \
        __builtin_amdgcn_s_setprio(0);                                                                         \
        __builtin_amdgcn_s_barrier();                                                                          \
        stage = stage + 1 == DMA_STAGES ? 0 : stage + 1;                                                       \
    }

// Split precision: the 128-byte K row is [32 hi | 32 lo], so ch0 addresses the hi halves and ch1 the lo halves of both
// operands, and the step multiplies hi*hi, hi*lo, lo*hi (48 MFMAs instead of 32 on the same LDS bytes -- the step is bound by
// the matrix pipe, not by LDS reads).  The MFMAs are dealt 24 / 24 over the two phases: first load phase hi of A, hi of B and
// the lo of B's first two column tiles; per output element the order is always hh, hl, lh (as in k_conv).
#define RCV_DMA_STEP_X3(ISSUE)                                                                                 \
    {                                                                                                          \
        const unsigned char *Ab = smem + stage * DMA_STAGE_BYTES + (wm * 64 + r16) * 128;                      \
        const unsigned char *Bb = smem + stage * DMA_STAGE_BYTES + BM * 128 + (wn * 64 + r16) * 128;           \
        const int st2 = stage + 2 >= DMA_STAGES ? stage + 2 - DMA_STAGES : stage + 2;                          \
        bf16x8 ah[MT], al[MT], bh[NT], bl[NT];                                                                 \
        _Pragma("unroll") for (int i = 0; i < MT; i++) ah[i] = *(const bf16x8 *)(Ab + i * 16 * 128 + ch0);     \
        _Pragma("unroll") for (int j = 0; j < NT; j++) bh[j] = *(const bf16x8 *)(Bb + j * 16 * 128 + ch0);     \
        _Pragma("unroll") for (int j = 0; j < NT / 2; j++) bl[j] = *(const bf16x8 *)(Bb + j * 16 * 128 + ch1); \
        if (ISSUE) RCV_DMA_ISSUE(st2, 0, 3, false);                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
        __builtin_amdgcn_s_barrier();                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; i++)                                                         \
            _Pragma("unroll") for (int j = 0; j < NT; j++) acc[i][j] = RCV_MFMA16(ah[i], bh[j], acc[i][j], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < MT; i++)                                                         \
            _Pragma("unroll") for (int j = 0; j < NT / 2; j++) acc[i][j] = RCV_MFMA16(ah[i], bl[j], acc[i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                         \
        __builtin_amdgcn_s_barrier();                                                                          \
        _Pragma("unroll") for (int i = 0; i < MT; i++) al[i] = *(const bf16x8 *)(Ab + i * 16 * 128 + ch1);     \
        _Pragma("unroll") for (int j = NT / 2; j < NT; j++) bl[j] = *(const bf16x8 *)(Bb + j * 16 * 128 + ch1); \
        if (ISSUE) {                                                                                           \
            RCV_DMA_ISSUE(st2, 3, 6, true);                                                                    \
            asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");                                        \
        } else {                                                                                               \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                        \
        }                                                                                                      \
        __builtin_amdgcn_s_barrier();                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; i++)                                                         \
            _Pragma("unroll") for (int j = NT / 2; j < NT; j++) acc[i][j] = RCV_MFMA16(ah[i], bl[j], acc[i][j], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < MT; i++)                                                         \
            _Pragma("unroll") for (int j = 0; j < NT; j++) acc[i][j] = RCV_MFMA16(al[i], bh[j], acc[i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                         \
        __builtin_amdgcn_s_barrier();                                                                          \
        stage = stage + 1 == DMA_STAGES ? 0 : stage + 1;                                                       \
    }
#define RCV_DMA_STEP_ANY(ISSUE)                                                                                \
    { if constexpr (kX3) RCV_DMA_STEP_X3(ISSUE) else RCV_DMA_STEP(ISSUE) }

    const int nk = p.K / KE;
    const int grp = wave >> 2;  // waves w and w + 4 share a SIMD
    RCV_DMA_ISSUE(0, 0, 6, true);
    if (nk > 1) {
        RCV_DMA_ISSUE(1, 0, 6, true);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();  // the second wave of every SIMD runs one phase behind
    int stage = 0;
    for (int kt = 0; kt + 2 < nk; kt++) RCV_DMA_STEP_ANY(true);
    if (nk > 1) RCV_DMA_STEP_ANY(false);
    RCV_DMA_STEP_ANY(false);
    if (grp == 0) __builtin_amdgcn_s_barrier();  // match the barrier count of the trailing group
#undef RCV_DMA_STEP_ANY
#undef RCV_DMA_STEP_X3
#undef RCV_DMA_STEP
#undef RCV_DMA_ISSUE

    // epilogue: the whole 256 x 128 fp32 tile fits the (now idle) 144 KB of LDS, so all eight waves stage their
    // accumulators at once and the tile leaves as 16-byte vectors of contiguous channels after a single barrier
    T *dst = (T *)p.dst + (long long)g * p.dst_gs;
    const float *bias = p.bias + (long long)g * p.bias_gs;
    constexpr int SROW = BN + 4, CPR = BN / EPV;
    float *stg = (float *)smem;
    if constexpr (kX3) RCV_DMA_RESIDUAL();
#undef RCV_DMA_RESIDUAL
    __syncthreads();  // every wave is done reading the last operand tile
#pragma unroll
    for (int j = 0; j < NT; j++) {
        const int nl = wn * 64 + j * 16 + r16;
        const float bv = bias[n0 + nl];
        const float sc = kX3 ? bias[p.Cout + n0 + nl] : 1.f;  // (split mode: the weights' power-of-two scale, undone here)
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
            for (int e = 0; e < 4; e++)
                stg[(wm * 64 + i * 16 + kq * 4 + e) * SROW + nl] = kX3 ? __fmaf_rn(acc[i][j][e], sc, bv) : acc[i][j][e] + bv;
    }
    __syncthreads();
    float rmax = 0.f;
#pragma unroll
    for (int round = 0; round < 4; round++) {
#pragma unroll
        for (int q = 0; q < E_PER; q++) {
            const int c = q * 512 + tid, row = c / CPR, cc = c - row * CPR;
            const int m = m0 + round * 64 + row;
            if (m >= p.M) continue;
            const long long o = nhwc_off<T>(m, p.Cout, n0 + cc * EPV);
            float v[EPV];
            const float4 *sp4 = (const float4 *)(stg + (round * 64 + row) * SROW + cc * EPV);
#pragma unroll
            for (int qq = 0; qq < EPV / 4; qq++) {
                const float4 t = sp4[qq];
                v[qq * 4 + 0] = t.x; v[qq * 4 + 1] = t.y; v[qq * 4 + 2] = t.z; v[qq * 4 + 3] = t.w;
            }
            if (res) {
                if constexpr (kX3) {
                    float rv[8];
                    x3_join8(rres[round][q][0], rres[round][q][RV - 1], rv);
#pragma unroll
                    for (int qq = 0; qq < EPV; qq++) v[qq] += rv[qq];
                } else {
                    const T *rp = (const T *)&rres[round][q][0];
#pragma unroll
                    for (int qq = 0; qq < EPV; qq++) v[qq] += to_f32(rp[qq]);
                }
            }
            if (p.relu) {
#pragma unroll
                for (int qq = 0; qq < EPV; qq++) v[qq] = fmaxf(v[qq], 0.f);
            }
            rmax = rng_see<EPV>(rmax, v);
            if constexpr (kX3) {
                uint4 oh, ol;
                x3_split8(v, oh, ol);
                *(uint4 *)(dst + o) = oh;
                *(uint4 *)(dst + o + 32) = ol;
            } else {
                uint4 outv;
                T *op = (T *)&outv;
#pragma unroll
                for (int qq = 0; qq < EPV; qq++) op[qq] = from_f32<T>(v[qq]);
                *(uint4 *)(dst + o) = outv;
            }
        }
    }
    rng_flush(p.range, rmax);
}

// ---------------------------------------------------------------------------------------------
// k_conv_lat (bf16): the latency variant for launches with few output tiles (one or a few frames: BASELINE.json configs[1],
// batch 1).  There a workgroup walks its whole K range alone -- 36 ... 432 dependent steps -- and the time of a layer is
// (steps) x (time of one step), which with k_conv's two-deep register staging is one L2 / HBM round trip per two steps.
// Here: small tiles (64 x 64, so that even M = 300 rows give 80+ workgroups; four of the eight waves multiply, 2 x 2 of
// 32 x 32), the operands by LDS-DMA into a ring of LAT_STAGES 16 KB stages, LAT_STAGES - 1 steps in flight, all eight
// waves issuing (one A and one B piece each per step), ONE barrier per step, counted vmcnt.  Same K order and
// the same MFMA sequence per output element as k_conv / k_conv_dma: bit-identical results (tests/test_gpu_fcn.py), which
// is what keeps a frame's maps independent of the batch it travels in.  No split-K by default: a different summation order would
// break exactly that (engine option 12 is the opt-in).  r06: the step's instruction stream cut to a third (see RCV_LAT_ISSUE_NOW).
// ---------------------------------------------------------------------------------------------
#ifndef RCV_LAT_AHEAD
#define RCV_LAT_AHEAD 1  // 1: fragment reads one K step ahead of the MFMAs (bit-identical).  r04, heavy issue path: 4.41 vs 4.30 ms per batch-1 f16x3 forward (slower); r06, lean issue path: see DESIGN.md
#endif
constexpr int LAT_BM = 64, LAT_BN = 64, LAT_STAGES = 8;
constexpr int LAT_STAGE_BYTES = (LAT_BM + LAT_BN) * 128;

// (plain function: the buffer builtins are device-only and a template kernel's body is also instantiated for the host)
__device__ __forceinline__ void lat_dma16(const void *base, int num_bytes, unsigned char *lds_wave_base, int voffset)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, num_bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds_wave_base, 16, voffset, 0, 0, 0);
}

__global__ __launch_bounds__(512) void k_conv_lat(const ConvParams p)
{
    RCV_X3_MODE();
    using T = H16;
    constexpr int KE = 64, EPV = 8, BM = LAT_BM, BN = LAT_BN, NS = LAT_STAGES;
    constexpr int MT = 2, NT = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // eight waves bring the operands in (one A and one B piece of 8 rows x 128 B each per K step: a DMA piece costs its
    // wave ~100 cycles of issue, so the pieces of a step are spread over the waves); waves 0..3 multiply (2 x 2 of
    // 32 x 32).  (Sixteen waves with twelve pure loaders measured slower: 3.70 vs 2.42 ms per forward at batch 1.)
    const bool mma = wave < 4;
    const int wm = (wave >> 1) & 1, wn = wave & 1;
    int mt_, nt_;
    tile_of_block(p.Cout / BN, mt_, nt_);
    // split-K (engine option 12, one or a few frames in flight): blockIdx.z = group * splits + split; this workgroup walks K steps
    // [kt0, kt1) of the same (slice-major, taps inside) order and leaves raw sums for k_splitk_finish
    const int nsplit = p.splits > 1 ? p.splits : 1;
    const int g = blockIdx.z / nsplit, ksp = blockIdx.z - g * nsplit, m0 = mt_ * BM, n0 = nt_ * BN;
    const int Ctot = p.C0 + p.C1;
    const T *src0 = (const T *)p.src0 + (long long)g * p.src0_gs;
    const T *src1 = p.src1 ? (const T *)p.src1 + (long long)g * p.src1_gs : nullptr;
    const T *wg = (const T *)p.w + (long long)g * p.w_gs;
    const T *res = (p.res && nsplit == 1) ? (const T *)p.res + (long long)g * p.res_gs : nullptr;
    // residual tile: 64 rows x 8 chunks of 16 B = 512 chunks, one per thread, requested up front
    uint4 rres = make_uint4(0u, 0u, 0u, 0u), rres_lo = make_uint4(0u, 0u, 0u, 0u);
    if (res) {
        const int row = tid >> 3, cc = tid & 7;
        const int m = m0 + row;
        if (m < p.M) {
            const T *rp = res + nhwc_off<T>(m, p.Cout, n0 + cc * 8);
            rres = *(const uint4 *)rp;
            if constexpr (kX3) rres_lo = *(const uint4 *)(rp + 32);
        }
    }
    // loader role of this lane: A row wave*8 + lane/8, B row likewise; 16-byte slot lane%8 of the LDS row receives the
    // source chunk slot ^ ((row>>1)&7) (k_conv_dma's scheme)
    const int lrow = lane >> 3, lpos = lane & 7;
    constexpr int OOB = (int)0x80000000;
    int a_off0 = OOB, a_off1 = OOB, b_off;
    unsigned a_ok = 0u;
    const int HWo = p.Hout * p.Wout;
    {
        const int row = wave * 8 + lrow;
        const int chunk = (lpos ^ ((row >> 1) & 7)) * EPV;
        const int m = m0 + row;
        if (m < p.M) {
            const int img = m / HWo, rem = m - img * HWo;
            const int ho = rem / p.Wout, wo = rem - ho * p.Wout;
            const int hc = ho * p.stride, wc = wo * p.stride;
            const int lin = (img * p.Hin + hc) * p.Win + wc;
            a_off0 = (lin * p.C0 + chunk) * 2;
            a_off1 = (lin * p.C1 + chunk) * 2;
            const int rlo = max(p.pad - hc, 0), rhi = min(p.Hin - 1 - hc + p.pad, p.KS - 1);
            const int qlo = max(p.pad - wc, 0), qhi = min(p.Win - 1 - wc + p.pad, p.KS - 1);
            const unsigned colmask = qhi >= qlo ? ((2u << qhi) - 1u) & ~((1u << qlo) - 1u) : 0u;
            for (int r = rlo; r <= rhi; r++) a_ok |= colmask << (r * p.KS);
        }
        b_off = (row * p.K + chunk) * 2;
    }
    const char *b_base = (const char *)(wg + (long long)n0 * p.K);
    const int n_img = __builtin_amdgcn_readfirstlane(p.M / HWo);
    const int a_bytes0 = (int)((long long)p.C0 * p.Hin * p.Win * n_img * 2), a_bytes1 = (int)((long long)p.C1 * p.Hin * p.Win * n_img * 2);
    const int b_bytes = BN * p.K * 2;
    const int nk_all = p.K / KE;
    const int kt0 = (int)((long long)nk_all * ksp / nsplit), kt1 = (int)((long long)nk_all * (ksp + 1) / nsplit);
    const int nk = kt1 - kt0;  // K steps of this workgroup
    const int taps = p.KS * p.KS, tap0 = kt0 % taps;
    int tap_r = tap0 / p.KS, tap_s = tap0 - (tap0 / p.KS) * p.KS, cg = (kt0 / taps) * KE, issued = 0;

// The two pieces of the next K step into stage `st_`.  r06: a step of this kernel is a chain of barrier -> issue -> fragment reads ->
// MFMAs walked 36 ... 864 times by a launch with few tiles, and at 8 (bf16) / 24 (split) MFMAs per step it is the INSTRUCTION STREAM
// that sets its length: ~100 scalar instructions per step went into 64-bit operand bases, per-piece descriptors, selects between the
// two sources and a compare-and-branch ladder for the counted wait.  Now: one descriptor per operand tensor and workgroup; the weights'
// step offset is the DMA's scalar offset; the patch displacement of a tap is a 32-bit scalar added to the lane's pixel offset; the
// source (descriptor, channel count, pixel offsets) changes only where the channel slice crosses into the second tensor; and the
// steady-state loop waits with a constant count (the tail of the K range drains through the ladder).
    const __amdgpu_buffer_rsrc_t rA0 = __builtin_amdgcn_make_buffer_rsrc((void *)src0, 0, a_bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA1 = __builtin_amdgcn_make_buffer_rsrc((void *)(src1 ? src1 : src0), 0, src1 ? a_bytes1 : a_bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void *)b_base, 0, b_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rA = rA0;   // source of the slice being issued
    int Cs = p.C0, c_in = cg, a_off = a_off0;
    auto lat_source = [&]() {          // (after cg moved)
        const bool first_ = cg < p.C0;
        rA = first_ ? rA0 : rA1;
        Cs = first_ ? p.C0 : p.C1;
        c_in = first_ ? cg : cg - p.C0;
        a_off = first_ ? a_off0 : a_off1;
    };
    lat_source();
#define RCV_LAT_ISSUE_NOW(st_)                                                                                 \
    {                                                                                                          \
        unsigned char *As_ = smem + (st_) * LAT_STAGE_BYTES;                                                   \
        unsigned char *Bs_ = As_ + BM * 128;                                                                   \
        const int tap_ = tap_r * p.KS + tap_s;                                                                 \
        const int disp_ = (((tap_r - p.pad) * p.Win + (tap_s - p.pad)) * Cs + c_in) * 2;                       \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void *)(As_ + wave * 8 * 128), 16, \
                                                 ((a_ok >> tap_) & 1u) ? a_off + disp_ : OOB, 0, 0, 0);        \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void *)(Bs_ + wave * 8 * 128), 16, \
                                                 b_off, (tap_ * Ctot + cg) * 2, 0, 0);                         \
        issued++;                                                                                              \
        tap_s += 1;                                                                                            \
        if (tap_s == p.KS) {                                                                                   \
            tap_s = 0;                                                                                         \
            tap_r += 1;                                                                                        \
            if (tap_r == p.KS) {                                                                               \
                tap_r = 0;                                                                                     \
                cg += KE;                                                                                      \
                lat_source();                                                                                  \
            }                                                                                                  \
        }                                                                                                      \
    }
#define RCV_LAT_ISSUE(st_)                                                                                     \
    if (issued < nk) RCV_LAT_ISSUE_NOW(st_)

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, kq = lane >> 4;
    const int ch0 = ((0 * 4 + kq) ^ ((r16 >> 1) & 7)) << 4, ch1 = ((1 * 4 + kq) ^ ((r16 >> 1) & 7)) << 4;
    // this wave's fragments of one K step (A rows wm*32 .. +31, B rows wn*32 .. +31; two 32-wide K halves each)
    bf16x8 af[2][2][MT], bfr[2][2][NT];  // [register set][K half][tile]
    auto read_frags = [&](int set, int st) {
        const unsigned char *Ab = smem + st * LAT_STAGE_BYTES + (wm * 32 + r16) * 128;
        const unsigned char *Bb = smem + st * LAT_STAGE_BYTES + BM * 128 + (wn * 32 + r16) * 128;
#pragma unroll
        for (int i = 0; i < MT; i++) { af[set][0][i] = *(const bf16x8 *)(Ab + i * 16 * 128 + ch0); af[set][1][i] = *(const bf16x8 *)(Ab + i * 16 * 128 + ch1); }
#pragma unroll
        for (int j = 0; j < NT; j++) { bfr[set][0][j] = *(const bf16x8 *)(Bb + j * 16 * 128 + ch0); bfr[set][1][j] = *(const bf16x8 *)(Bb + j * 16 * 128 + ch1); }
    };
    auto multiply = [&](int set) {
        if constexpr (kX3) {  // [32 hi | 32 lo] K rows: hi*hi, hi*lo, lo*hi
#pragma unroll
            for (int t = 0; t < 3; t++)
#pragma unroll
                for (int i = 0; i < MT; i++)
#pragma unroll
                    for (int j = 0; j < NT; j++) acc[i][j] = RCV_MFMA16(af[set][t == 2 ? 1 : 0][i], bfr[set][t == 1 ? 1 : 0][j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; kk++)
#pragma unroll
                for (int i = 0; i < MT; i++)
#pragma unroll
                    for (int j = 0; j < NT; j++) acc[i][j] = RCV_MFMA16(af[set][kk][i], bfr[set][kk][j], acc[i][j], 0, 0, 0);
        }
    };
    // counted wait: `younger` K steps (two pieces each, wave-uniform) were issued after the one needed now
    auto wait_steps = [&](int younger) {
        if (younger >= 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (younger == 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (younger == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    static_assert(NS == 8, "the counted waits are written for 8 stages");
#if RCV_LAT_AHEAD
    // The fragment reads run ONE STEP AHEAD of the MFMAs.  A step is barrier -> issue -> eight fragment reads -> their LDS latency ->
    // the MFMAs, a dependent chain walked 36 ... 864 times by a launch with few tiles.  Here the barrier of step kt certifies step
    // kt + 1 (all eight stages are issued up front and the counted wait looks one step further), the multiplying waves request the
    // fragments of step kt + 1 into a second register set and multiply step kt's, which they have held since the previous step: the
    // LDS round trip leaves the chain.  The stage of step kt is free as soon as that barrier is passed (its fragments are in
    // registers) and takes step kt + 8.  Same products in the same order.  (r04, with the heavy issue path: 4.41 against 4.30 ms per
    // batch-1 f16x3 forward -- slower; r06, lean issue path: see RCV_LAT_AHEAD.)
#pragma unroll
    for (int s_ = 0; s_ < NS; s_++) RCV_LAT_ISSUE(s_);
    wait_steps(min(NS - 1, nk - 1));  // step 0 has landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    if (mma) read_frags(0, 0);
    int stage = 0, kt = 0;
#define RCV_LAT_STEP(par_, WAIT_, ISSUE_)                                                                      \
    {                                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* this wave's reads of `stage` are complete: it may be refilled */ \
        WAIT_;                                             /* step kt + 1 has landed (this wave's pieces) */     \
        __builtin_amdgcn_s_barrier();                      /* ... and everybody's; everybody holds step kt in registers */ \
        const int nxt_ = stage + 1 == NS ? 0 : stage + 1;                                                      \
        if (mma && kt + 1 < nk) read_frags((par_) ^ 1, nxt_);                                                  \
        ISSUE_;                                                                                                \
        if (mma) multiply(par_);                                                                               \
        stage = nxt_;                                                                                          \
        kt++;                                                                                                  \
    }
    // steady state, two steps per trip (the register set of a step is a compile-time index): step kt + NS exists and takes the
    // stage of step kt; NS - 2 steps younger than step kt + 1 are in flight
    while (kt + NS + 1 < nk) {
        RCV_LAT_STEP(0, asm volatile("s_waitcnt vmcnt(12)" ::: "memory"), RCV_LAT_ISSUE_NOW(stage));
        RCV_LAT_STEP(1, asm volatile("s_waitcnt vmcnt(12)" ::: "memory"), RCV_LAT_ISSUE_NOW(stage));
    }
    // tail: the last steps to issue (if any), then the drain
    while (kt < nk) {
        RCV_LAT_STEP(0, if (kt + 1 < nk) wait_steps(min(NS - 2, min(issued, nk) - 2 - kt)), RCV_LAT_ISSUE(stage));
        if (kt < nk) RCV_LAT_STEP(1, if (kt + 1 < nk) wait_steps(min(NS - 2, min(issued, nk) - 2 - kt)), RCV_LAT_ISSUE(stage));
    }
#undef RCV_LAT_STEP
#else
#pragma unroll
    for (int s_ = 0; s_ < NS - 1; s_++) RCV_LAT_ISSUE(s_);
    int stage = 0;
    int kt = 0;
    // steady state: step kt + NS - 1 exists and is issued into the stage step kt - 1 was read from; NS - 2 younger steps in flight
    for (; kt + NS - 1 < nk; kt++) {
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // step kt has landed for every wave; every wave is done reading step kt - 1
        const int free_ = stage == 0 ? NS - 1 : stage - 1;
        RCV_LAT_ISSUE_NOW(free_);
        if (mma) {
            read_frags(0, stage);
            multiply(0);
        }
        stage = stage + 1 == NS ? 0 : stage + 1;
    }
    // tail: nothing left to issue; steps issued after step kt: nk - 1 - kt (<= NS - 2), two pieces each (wave-uniform)
    for (; kt < nk; kt++) {
        wait_steps(min(NS - 2, nk - 1 - kt));
        __builtin_amdgcn_s_barrier();
        if (mma) {
            read_frags(0, stage);
            multiply(0);
        }
        stage = stage + 1 == NS ? 0 : stage + 1;
    }
#endif
#undef RCV_LAT_ISSUE
#undef RCV_LAT_ISSUE_NOW
    // epilogue: fp32 tile through LDS, 16-byte vectors of contiguous channels (bias + residual + ReLU fused)
    T *dst = (T *)p.dst + (long long)g * p.dst_gs;
    const float *bias = p.bias + (long long)g * p.bias_gs;
    constexpr int SROW = BN + 4;
    float *stg = (float *)smem;
    __syncthreads();
    if (nsplit > 1) {  // raw sums -> partial[g][split][m][n]; bias, scale, residual, ReLU and the store happen in k_splitk_finish
        if (mma) {
#pragma unroll
            for (int j = 0; j < NT; j++)
#pragma unroll
                for (int i = 0; i < MT; i++)
#pragma unroll
                    for (int e = 0; e < 4; e++) stg[(wm * 32 + i * 16 + kq * 4 + e) * SROW + wn * 32 + j * 16 + r16] = acc[i][j][e];
        }
        __syncthreads();
        const int row = tid >> 3, cc = tid & 7, m = m0 + row;
        if (m < p.M) {
            float *pp = p.partial + (((long long)g * nsplit + ksp) * p.M + m) * p.Cout + n0 + cc * EPV;
            const float4 *sp4 = (const float4 *)(stg + row * SROW + cc * EPV);
            *(float4 *)pp = sp4[0];
            *(float4 *)(pp + 4) = sp4[1];
        }
        return;
    }
    if (mma) {
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int nl = wn * 32 + j * 16 + r16;
            const float bv = bias[n0 + nl];
            const float sc = kX3 ? bias[p.Cout + n0 + nl] : 1.f;
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int e = 0; e < 4; e++)
                    stg[(wm * 32 + i * 16 + kq * 4 + e) * SROW + nl] = kX3 ? __fmaf_rn(acc[i][j][e], sc, bv) : acc[i][j][e] + bv;
        }
    }
    __syncthreads();
    float rmax = 0.f;
    {
        const int row = tid >> 3, cc = tid & 7;
        const int m = m0 + row;
        if (m < p.M) {
            float v[EPV];
            const float4 *sp4 = (const float4 *)(stg + row * SROW + cc * EPV);
#pragma unroll
            for (int qq = 0; qq < 2; qq++) {
                const float4 t = sp4[qq];
                v[qq * 4 + 0] = t.x; v[qq * 4 + 1] = t.y; v[qq * 4 + 2] = t.z; v[qq * 4 + 3] = t.w;
            }
            if (res) {
                if constexpr (kX3) {
                    float rv[8];
                    x3_join8(rres, rres_lo, rv);
#pragma unroll
                    for (int qq = 0; qq < EPV; qq++) v[qq] += rv[qq];
                } else {
                    const T *rp = (const T *)&rres;
#pragma unroll
                    for (int qq = 0; qq < EPV; qq++) v[qq] += to_f32(rp[qq]);
                }
            }
            if (p.relu) {
#pragma unroll
                for (int qq = 0; qq < EPV; qq++) v[qq] = fmaxf(v[qq], 0.f);
            }
            T *op_ = dst + nhwc_off<T>(m, p.Cout, n0 + cc * EPV);
            rmax = rng_see<EPV>(rmax, v);
            if constexpr (kX3) {
                uint4 oh, ol;
                x3_split8(v, oh, ol);
                *(uint4 *)op_ = oh;
                *(uint4 *)(op_ + 32) = ol;
            } else {
                uint4 outv;
                T *op = (T *)&outv;
#pragma unroll
                for (int qq = 0; qq < EPV; qq++) op[qq] = from_f32<T>(v[qq]);
                *(uint4 *)op_ = outv;
            }
        }
    }
    rng_flush(p.range, rmax);
}

// k_splitk_finish (engine option 12): the `splits` partial sums of k_conv_lat are added in split order (a fixed order: results are
// run-to-run identical, though not the bits of the unsplit kernel) and go through k_conv_lat's epilogue -- scale / bias, residual,
// ReLU, range guard, store.  One thread per (row, eight channels).
__global__ __launch_bounds__(256) void k_splitk_finish(const ConvParams p)
{
    RCV_X3_MODE();
    using T = H16;
    constexpr int EPV = 8;
    const int g = blockIdx.y, cpr = p.Cout / EPV;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)p.M * cpr) return;
    const int m = (int)(idx / cpr), n = (int)(idx - (long long)m * cpr) * EPV;
    float v[EPV];
#pragma unroll
    for (int q = 0; q < EPV; q++) v[q] = 0.f;
    for (int s_ = 0; s_ < p.splits; s_++) {
        const float *pp = p.partial + (((long long)g * p.splits + s_) * p.M + m) * p.Cout + n;
        const float4 a = *(const float4 *)pp, b = *(const float4 *)(pp + 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    const float *bias = p.bias + (long long)g * p.bias_gs;
#pragma unroll
    for (int q = 0; q < EPV; q++) v[q] = kX3 ? __fmaf_rn(v[q], bias[p.Cout + n + q], bias[n + q]) : v[q] + bias[n + q];
    if (p.res) {
        const T *rp = (const T *)p.res + (long long)g * p.res_gs + nhwc_off<T>(m, p.Cout, n);
        const uint4 rh = *(const uint4 *)rp;
        if constexpr (kX3) {
            const uint4 rl = *(const uint4 *)(rp + 32);
            float rv[8];
            x3_join8(rh, rl, rv);
#pragma unroll
            for (int q = 0; q < EPV; q++) v[q] += rv[q];
        } else {
            const T *r8 = (const T *)&rh;
#pragma unroll
            for (int q = 0; q < EPV; q++) v[q] += to_f32(r8[q]);
        }
    }
    if (p.relu) {
#pragma unroll
        for (int q = 0; q < EPV; q++) v[q] = fmaxf(v[q], 0.f);
    }
    float rmax = rng_see<EPV>(0.f, v);
    T *op_ = (T *)p.dst + (long long)g * p.dst_gs + nhwc_off<T>(m, p.Cout, n);
    if constexpr (kX3) {
        uint4 oh, ol;
        x3_split8(v, oh, ol);
        *(uint4 *)op_ = oh;
        *(uint4 *)(op_ + 32) = ol;
    } else {
        uint4 outv;
        T *o8 = (T *)&outv;
#pragma unroll
        for (int q = 0; q < EPV; q++) o8[q] = from_f32<T>(v[q]);
        *(uint4 *)op_ = outv;
    }
    rng_flush(p.range, rmax);
}

// ---------------------------------------------------------------------------------------------
// k_conv_halo (bf16): 3x3 / stride 1 / pad 1 convolutions with the A operand read from a halo'd pixel patch.
// k_conv_dma brings the [256 px][64 ch] A tile of a channel slice in nine times, once per tap (32 KB per K step through
// the CU's 64 B/clk vector-memory path, next to 16 KB of weights).  Here a workgroup owns a th x tw pixel rectangle of one
// image (th * tw = 64 * MT pixels at most: 256, 320 with BN = 128 or 384 with BN = 64 -- the taller the tile, the more MFMAs
// behind each barrier) and keeps, per 64-channel slice, the (th+2) x (tw+2) patch around it in LDS (<= 56 / 64 rows-of-8,
// double buffered, filled by LDS-DMA during the previous slice's nine steps, seven / eight pieces per wave); the A fragments of
// tap (dh, dw) are ds_read_b128 at the patch row of the pixel shifted by (dh-1) * (tw+2) + (dw-1).  Pixels outside the
// image are out-of-range buffer offsets (zeros), per patch row and independent of the tap.  Weights: k_conv_dma's ring of
// three 16 KB stages, two pieces per wave and step.  Same K order (channel-slice-major, taps innermost) and MFMA sequence
// per output element as k_conv_dma: bit-identical results.  Four-phase step, wave groups one barrier apart, as there.
// Every step issues the same number of pieces (beyond the end: out-of-range offsets), so one counted vmcnt serves all.
// ---------------------------------------------------------------------------------------------
#ifdef RCV_HALO_STAMPS
// diagnostic build only (tools/halo_stamps.py): cycle stamps of ONE K step (slice 2, tap 4) of the split-precision halo kernel,
// waves 0 and 4 (the two wave groups of SIMD 0) of one workgroup in the middle of the launch
__device__ unsigned g_halo_stamps[2][16];
__device__ unsigned g_chain_stamps[2][64];  // k_chain_x3<256>, chunk 3: four stamps per step (top, after the counted wait, after the barrier, after the issue) + the end
#define RCV_HSTAMP(idx)                                                                                        \
    if (stamp_on) {                                                                                            \
        unsigned long long t_;                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                         \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        hst[idx] = (unsigned)t_;                                                                               \
    }
#else
#define RCV_HSTAMP(idx)
#endif
// 16 halves (two registers quads: this lane's chunk and its neighbour's) -> 12 bytes of codes + the scale byte
__device__ __forceinline__ i32x4 q6_block(const uint4 &lo8, const uint4 &hi8, int scale_byte_pos)
{
#if defined(RCV_FCN_X3)
    const unsigned w[8] = {lo8.x, lo8.y, lo8.z, lo8.w, hi8.x, hi8.y, hi8.z, hi8.w};
    unsigned m = 0u;  // max |.| of the 16 halves as a bit pattern (|half| orders like an unsigned integer)
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const unsigned a = w[e] & 0x7fff7fffu;
        m = max(m, max(a & 0xffffu, a >> 16));
    }
    const float amax = (float)__builtin_bit_cast(_Float16, (unsigned short)m);
    const int ex = max((int)(__float_as_uint(amax) >> 23), 3) - 2;  // e8m0 byte of the block scale (an all-zero block: 2^-126)
    f16x32 v;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const x3_h2 h = __builtin_bit_cast(x3_h2, w[e]);
        v[2 * e] = h[0];
        v[2 * e + 1] = h[1];
    }
#pragma unroll
    for (int e = 16; e < 32; e++) v[e] = (_Float16)0.f;
    const i32x6 c = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, __uint_as_float((unsigned)ex << 23));
    return i32x4{c[0], c[1], c[2], ex << (8 * scale_byte_pos)};
#else
    return i32x4{0, 0, 0, 0};
#endif
}
// In place: [32 hi | 32 lo] rows -> q6 rows.  A wave takes 64 rows (8 KB) per pass: lane = (row in a group of 8, chunk), eight
// coalesced 1 KB loads, all of them before the first store (the lo chunks are overwritten); the even lane of a pair converts
// the pair's 16 channels and stores one 16-byte chunk.
__global__ __launch_bounds__(256) void k_q6_pack(H16 *t, long long n_rows)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long long)gridDim.x * 4;
    for (long long r0 = wave * 64; r0 < n_rows; r0 += n_waves * 64) {
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const long long row = r0 + k * 8 + (lane >> 3);
            v[k] = row < n_rows ? *(const uint4 *)((const char *)t + row * 128 + (lane & 7) * 16) : uint4{0u, 0u, 0u, 0u};
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint4 nb;  // the neighbouring chunk (lane ^ 1)
            nb.x = (unsigned)__builtin_amdgcn_mov_dpp((int)v[k].x, 0xb1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
            nb.y = (unsigned)__builtin_amdgcn_mov_dpp((int)v[k].y, 0xb1, 0xf, 0xf, true);
            nb.z = (unsigned)__builtin_amdgcn_mov_dpp((int)v[k].z, 0xb1, 0xf, 0xf, true);
            nb.w = (unsigned)__builtin_amdgcn_mov_dpp((int)v[k].w, 0xb1, 0xf, 0xf, true);
            const long long row = r0 + k * 8 + (lane >> 3);
            if ((lane & 1) == 0 && row < n_rows) {
                const i32x4 q = q6_block(v[k], nb, 0);
                *(i32x4 *)((char *)t + row * 128 + 64 + ((lane & 7) >> 1) * 16) = q;
            }
        }
    }
}
struct HaloParams {
    const void *src0, *src1, *w;
    void *dst;
    const float *bias;
    long long src0_gs, src1_gs, dst_gs, w_gs;
    int bias_gs;
    int C0, C1, Cout, H, W, n_img, relu, K;
    int th, tw, tiles_y, tiles_x;
    int c1_shift;     // log2(C1 / C0) (signed) when a second source has another channel count: its patch offsets are the first source's, rescaled
    int groups;       // composed tail (persistent workgroups): the work items are (group, rectangle), dealt by the kernel itself
    int out_b, out_img0;  // composed tail: images per group of the output planes, and the first image of this launch (batches beyond the 32-bit DMA range run in slices)
    unsigned *range;  // sticky range-flag word (ConvParams::range)
    // parity-composed tail (template parameter TAILC, engine option 13): the head's weights / [bias | scale] record and the output planes
    const void *w8;
    const float *b8;
    long long w8_gs;
    int b8_gs, rad_f16;
    float *seg, *radial;
};
// log2(c1 / c0) if the ratio is a power of two (0 without a second source), else INT_MIN: k_conv_halo rescales its patch offsets by shifts
inline int halo_c1_shift(int c0, int c1)
{
    if (c1 <= 0 || c1 == c0) return 0;
    for (int s = 1; s < 16; s++) {
        if (c0 << s == c1) return s;
        if (c1 << s == c0) return -s;
    }
    return (int)0x80000000;
}
// patch pieces (8 rows x 128 B) per wave and slice, one per tap 0 .. PPW - 1.  BN = 128: 7 -> 448 patch rows, a 320-pixel rectangle
// with its halo (10 x 42); BN = 64 (half the weight stages): 8 -> 512 rows, a 384-pixel rectangle (14 x 34)
constexpr int halo_ppw(int bn) { return bn == 64 ? 8 : 7; }
constexpr int halo_patch_rows(int bn) { return halo_ppw(bn) * 64; }
constexpr int halo_patch(int bn) { return halo_ppw(bn) * 8 * 1024; }  // bytes per patch buffer
constexpr int halo_lds(int bn) { return 2 * halo_patch(bn) + 3 * bn * 128; }  // 160 KB for BN = 128, 152 KB for BN = 64
static_assert(halo_lds(128) <= 160 * 1024 && halo_lds(64) <= 160 * 1024, "k_conv_halo's LDS plan");

// (plain function: the buffer builtins are device-only and a template kernel's body is also instantiated for the host)
__device__ __forceinline__ void halo_dma16(const void *base, int num_bytes, unsigned char *lds_wave_base, int voffset)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, num_bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds_wave_base, 16, voffset, 0, 0, 0);
}

// ... through a descriptor built once per workgroup, the wave-uniform part of the address in the instruction's scalar offset
__device__ __forceinline__ void halo_dma16r(__amdgpu_buffer_rsrc_t r, unsigned char *lds_wave_base, int voffset, int soffset)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds_wave_base, 16, voffset, soffset, 0, 0);
}

template <int BN, int MT, bool Q6 = false, bool TAILC = false>
__global__ __launch_bounds__(512) void k_conv_halo(const HaloParams p)
{
    RCV_X3_MODE();
    using T = H16;
    constexpr int EPV = 8, NT = BN / 32, RW = 16 * MT;   // 8 waves = 4 x 2 of RW pixels x BN / 2 channels (tile: 4 * RW rows)
    constexpr int BPW = BN / 64;                             // weight pieces per wave and K step
    constexpr int HALO_SKIP = -128;  // byte offset of a patch row outside the image: beyond any tensor, and still negative after the rescaling shifts
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int HALO_PPW = halo_ppw(BN), HALO_PATCH = halo_patch(BN);
    unsigned char *sP = smem;                    // [2][HALO_PATCH]
    unsigned char *sB = smem + 2 * HALO_PATCH;   // [3][BN][128 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // The composed tail (TAILC) runs with PERSISTENT workgroups, one per CU, each walking its (group, rectangle) items itself: a
    // workgroup of that launch lives 18 K steps, so dispatch, index arithmetic and the first operands' latency were a fifth of its
    // life -- and its epilogue needs no LDS, so the next item's patch and first weights are requested BEFORE the epilogue of this one
    // runs.  The items of an XCD are a contiguous range (its workgroups take them round-robin: neighbours meet in its L2).  Every
    // other form: one item per workgroup, numbered by tile_of_block (a restructuring of all forms cost the long-K launches more in
    // lost hoisting than it gave: DESIGN.md 8).
#ifndef RCV_HALO_PERSIST_ALL
#define RCV_HALO_PERSIST_ALL 0  // 1: every form of the split engine with persistent workgroups (A-B)
#endif
    constexpr bool PERSIST = kX3 && (TAILC || BN == 64 || RCV_HALO_PERSIST_ALL);  // (BN = 64: the 18- and 27-step launches of layer 1 and conv_up1, -2 ... -4 %)
    // the epilogue from registers (no LDS: the next item's first operands go out BEFORE it runs) wants the MFMA operands swapped: a lane
    // then holds four consecutive CHANNELS of one pixel instead of four pixels of one channel; the same products in the same K order
#ifndef RCV_HALO_REGEPI_ALL
#define RCV_HALO_REGEPI_ALL 0  // 1 (with RCV_HALO_PERSIST_ALL): the plain forms too store from registers with swapped operands (A-B: K loop +5-8 %)
#endif
    constexpr bool REGEPI = PERSIST && !Q6 && (TAILC || RCV_HALO_REGEPI_ALL), HALO_SWAP = REGEPI;
    const int tpi = p.tiles_y * p.tiles_x;
    const int n_nt = p.Cout / BN, n_mt = p.n_img * tpi;
    int item = 0, item_step = 0, item_end = 0;
    int mt_, nt_, g;
    if constexpr (PERSIST) {  // items = (group, rectangle, channel block), the channel blocks of a rectangle neighbours
        const unsigned total = (unsigned)(n_nt * n_mt) * (unsigned)p.groups;
        const unsigned xcd = blockIdx.x % 8u, slot = blockIdx.x / 8u, q8 = total / 8u, r8 = total % 8u;
        const unsigned lo = xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8;
        item = (int)(lo + slot);
        item_end = (int)(lo + q8 + (xcd < r8 ? 1u : 0u));
        item_step = (int)((gridDim.x - xcd + 7u) / 8u);  // workgroups of this XCD
        if (item >= item_end) return;
        nt_ = item % n_nt;
        mt_ = (item / n_nt) % n_mt;
        g = item / (n_nt * n_mt);
    } else {
        tile_of_block(n_nt, mt_, nt_);
        g = blockIdx.z;
    }
    int n0 = nt_ * BN;
    int img = mt_ / tpi, trem = mt_ - img * tpi;
    int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    int y0 = ty * p.th, x0 = tx * p.tw;
    const int pitch = p.tw + 2, prows = (p.th + 2) * pitch, npx = p.th * p.tw;
    const int Ctot = p.C0 + p.C1;
    const T *src0 = (const T *)p.src0 + (long long)g * p.src0_gs;
    const T *src1 = p.src1 ? (const T *)p.src1 + (long long)g * p.src1_gs : nullptr;
    const T *wg = (const T *)p.w + (long long)g * p.w_gs;
    // PERSIST: the current item moves on to the next one (decode, operand bases, descriptors, patch offsets) and its first operands go out
    auto next_item_setup = [&](int it_) {
        item = it_;
        nt_ = item % n_nt;
        mt_ = (item / n_nt) % n_mt;
        g = item / (n_nt * n_mt);
        n0 = nt_ * BN;
        img = mt_ / tpi;
        trem = mt_ - img * tpi;
        ty = trem / p.tiles_x;
        tx = trem - ty * p.tiles_x;
        y0 = ty * p.th;
        x0 = tx * p.tw;
        src0 = (const T *)p.src0 + (long long)g * p.src0_gs;
        src1 = p.src1 ? (const T *)p.src1 + (long long)g * p.src1_gs : nullptr;
    };
    const int lrow = lane >> 3, lpos = lane & 7;
    const int r16 = lane & 15, kq = lane >> 4;
    const float rP = 1.0f / (float)pitch, rT = 1.0f / (float)p.tw;

    // loader: patch piece k of this wave = patch rows (k * 8 + wave) * 8 + lrow; pixel index in the image or -1
    int lin[HALO_PPW];
    auto fill_lin = [&]() {
#pragma unroll
    for (int k = 0; k < HALO_PPW; k++) {
        int lrow_ = lrow, lpos_ = lpos;
        if constexpr (PERSIST) {  // (the lane's part comes from the exec mask, not from registers that would have to live across the K loop -- and
                                  // nothing of this arithmetic may be hoisted out of the item loop)
            const int lane_ = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            lrow_ = lane_ >> 3;
            lpos_ = lane_ & 7;
            asm volatile("" : "+v"(lrow_), "+v"(lpos_));
        }
        const int pr = (k * 8 + wave) * 8 + lrow_;
        const int py = (int)(((float)pr + 0.5f) * rP), px = pr - py * pitch;
        const int y = y0 - 1 + py, x = x0 - 1 + px;
        // r06: lin[k] is the piece's BYTE OFFSET in the first source, pixel * (C0 * 2) + its swizzled 16-byte chunk (HALO_SKIP = -128 for
        // rows outside the image: out of range for the buffer load, zeros in LDS).  It is the same for every channel slice -- the slice
        // moves the descriptor's base -- so a patch piece costs the load phase NO vector instruction (r02-r05: unpack, multiply, swizzle,
        // a compare and an exec-mask round trip, ~10 instructions per piece in the phase the partner wave's MFMAs wait for); slices of a
        // second source with another channel count rescale it by the power of two C1 / C0 (arithmetic shifts: the sentinel stays negative).
        // The patch is swizzled with chunk ^= key(patch position), key = (px + tw * py) & 7.  The taps read the patch rows of 16 CONSECUTIVE PIXELS of the rectangle from any
        // start; along the rectangle's pixel order this key counts on across the end of a rectangle row, where the patch row
        // jumps by the two halo columns -- with the r02/r03 key (patch row & 7, continuous only inside a row) every 16-pixel tile
        // that wraps (three of four in the 15 x 20 rectangles of layer 3) read with 2-way bank conflicts: rocprofv3 counted
        // SQ_LDS_BANK_CONFLICT at 0.9 of SQ_ACTIVE_INST_LDS in k_conv_halo<128,5> and zero in the other kernels (r04).
        const int key = (px + p.tw * py) & 7;
        if constexpr (TAILC) {  // the composed tail reads the low-resolution tensor with REPLICATE padding (the upsampling's clamp)
            const int yc = min(max(y, 0), p.H - 1), xc = min(max(x, 0), p.W - 1);
            lin[k] = pr < prows ? ((img * p.H + yc) * p.W + xc) * (p.C0 * 2) + ((lpos_ ^ key) << 4) : HALO_SKIP;
        } else
            lin[k] = (pr < prows && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) ? ((img * p.H + y) * p.W + x) * (p.C0 * 2) + ((lpos_ ^ key) << 4) : HALO_SKIP;
    }
    };
    fill_lin();
    int b_off[BPW];
#pragma unroll
    for (int c = 0; c < BPW; c++) {
        const int row = wave * (8 * BPW) + c * 8 + lrow;
        b_off[c] = (row * p.K + (lpos ^ ((row >> 1) & 7)) * EPV) * 2;
    }
    const char *b_base = (const char *)(wg + (long long)n0 * p.K);
    const int a_bytes0 = (int)((long long)p.C0 * p.H * p.W * p.n_img * 2), a_bytes1 = (int)((long long)p.C1 * p.H * p.W * p.n_img * 2);
    const int b_bytes = BN * p.K * 2;
    const int nsl = Ctot >> 6;
    const int c1_shift = p.c1_shift;

    // fragment rows: tile row r = wm * RW + i * 16 + r16 -> patch row of its pixel (centre tap); rows beyond the rectangle
    // read pixel 0's (their results are never stored)
    int pr0[MT];
#pragma unroll
    for (int i = 0; i < MT; i++) {
        int r = wm * RW + i * 16 + r16;
        r = r < npx ? r : 0;
        const int py = (int)(((float)r + 0.5f) * rT), px = r - py * p.tw;
        pr0[i] = (py + 1) * pitch + px + 1;
    }

// patch piece k of channel slice sl_ into patch buffer (sl_ & 1); slices beyond the last one: zeros
// r06: the load phases are what the partner wave's MFMA phase waits for at the barrier, and a scalar instruction costs them an issue
// slot like any other.  The three operand tensors get ONE descriptor each per workgroup; what changes per step is a scalar byte
// offset (the instruction's soffset): (tap * Ctot + slice * 64) * 2 for the weights, the slice's channel offset for a patch -- ~30
// scalar instructions per step less than building a descriptor (64-bit base, selects between the two sources) per piece.  Pieces past
// the end (the weights of steps nsteps, nsteps + 1; the patch of slice nsl) re-read offset 0: nobody consumes them.
    __amdgpu_buffer_rsrc_t rA0 = __builtin_amdgcn_make_buffer_rsrc((void *)src0, 0, a_bytes0, 0x00020000);
    __amdgpu_buffer_rsrc_t rA1 = __builtin_amdgcn_make_buffer_rsrc((void *)(src1 ? src1 : src0), 0, src1 ? a_bytes1 : a_bytes0, 0x00020000);
    __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void *)b_base, 0, b_bytes, 0x00020000);
    // the patch source of the NEXT slice (set at the top of every slice): descriptor, channel offset, rescaling of the offsets
    __amdgpu_buffer_rsrc_t rAn = rA0;
    int soffA = 0;
    bool xfn = false;
#define RCV_HALO_NEXT_SLICE(sl_)                                                                               \
    {                                                                                                          \
        const int sn_ = (sl_) + 1 < nsl ? (sl_) + 1 : 0, cgn_ = sn_ << 6;                                      \
        const bool firstn_ = cgn_ < p.C0;                                                                      \
        rAn = firstn_ ? rA0 : rA1;                                                                             \
        soffA = (firstn_ ? cgn_ : cgn_ - p.C0) * 2;                                                            \
        xfn = !firstn_ && c1_shift != 0;                                                                       \
    }
// patch piece k of the next slice into patch buffer ((sl_) & 1); sl_ = 0: the prologue's (first source, channel offset 0)
#define RCV_HALO_PATCH(sl_, k)                                                                                 \
    {                                                                                                          \
        int off_ = lin[k];                                                                                     \
        if (xfn) { /* (wave-uniform branch) */                                                                 \
            asm volatile("" : "+v"(off_)); /* (here, every time: hoisted, the second source's offsets cost HALO_PPW registers) */ \
            const int hi_ = off_ & ~127;                                                                       \
            off_ = (c1_shift > 0 ? hi_ << c1_shift : hi_ >> -c1_shift) | (off_ & 127);                         \
        }                                                                                                      \
        halo_dma16r(rAn, sP + ((sl_) & 1) * HALO_PATCH + ((k) * 8 + wave) * 1024, off_, soffA);                \
    }
// weight piece c of K step (slice sl_, tap t_) into ring stage st_
#define RCV_HALO_B(sl_, t_, st_, c)                                                                            \
    {                                                                                                          \
        halo_dma16r(rB, sB + (st_) * (BN * 128) + (wave * (8 * BPW) + (c) * 8) * 128, b_off[c],                \
                    (sl_) < nsl ? ((t_) * Ctot + ((sl_) << 6)) * 2 : 0);                                       \
    }

    f32x4 acc[MT][NT];
    const int ch0 = ((0 * 4 + kq) ^ ((r16 >> 1) & 7)) << 4, ch1 = ((1 * 4 + kq) ^ ((r16 >> 1) & 7)) << 4;

// One K step = (slice sl, tap TAP), four phases as in k_conv_dma.  Pieces issued: first load phase the first weight piece
// of step + 2, second load phase its second piece and (taps 0..6) one patch piece of slice sl + 1; the counted wait of the
// second load phase leaves exactly those in flight, so the weights of step + 1 (and by tap 7 the whole next patch) have
// landed at least one barrier before anybody reads them.  The patch buffer refilled during slice sl was last read in the
// second load phase of slice sl - 1's last step, two barriers before the first piece goes out.
#define RCV_HALO_STEP(TAP)                                                                                     \
    {                                                                                                          \
        constexpr int T2_ = ((TAP) + 2) % 9, S2_ = ((TAP) + 2) / 9;                                            \
        const int shift_ = ((TAP) / 3 - 1) * pitch + ((TAP) % 3 - 1);                                          \
        const unsigned char *Pb = sP + (sl & 1) * HALO_PATCH;                                                  \
        const unsigned char *Bb = sB + stage * (BN * 128) + (wn * (BN / 2) + r16) * 128;                       \
        const int st2 = stage + 2 >= 3 ? stage + 2 - 3 : stage + 2;                                            \
        int ao_[MT];                                                                                           \
        const int kx_ = (kq ^ ((r16 + ((TAP) / 3) * p.tw + (TAP) % 3) & 7)) << 4;                              \
        _Pragma("unroll") for (int i = 0; i < MT; i++) ao_[i] = (pr0[i] + shift_) * 128 + kx_;                 \
        bf16x8 af[MT], bfr[NT];                                                                                \
        _Pragma("unroll") for (int i = 0; i < MT; i++) af[i] = *(const bf16x8 *)(Pb + ao_[i]);                 \
        _Pragma("unroll") for (int j = 0; j < NT; j++) bfr[j] = *(const bf16x8 *)(Bb + j * 16 * 128 + ch0);    \
        RCV_HALO_B(sl + S2_, T2_, st2, 0);                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
        __builtin_amdgcn_s_barrier();                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; i++)                                                         \
            _Pragma("unroll") for (int j = 0; j < NT; j++)                                                     \
                acc[i][j] = RCV_MFMA16(af[i], bfr[j], acc[i][j], 0, 0, 0);        \
        __builtin_amdgcn_s_setprio(0);                                                                         \
        __builtin_amdgcn_s_barrier();                                                                          \
        _Pragma("unroll") for (int i = 0; i < MT; i++) af[i] = *(const bf16x8 *)(Pb + (ao_[i] ^ 64));          \
        _Pragma("unroll") for (int j = 0; j < NT; j++) bfr[j] = *(const bf16x8 *)(Bb + j * 16 * 128 + ch1);    \
        if constexpr (BPW == 2) RCV_HALO_B(sl + S2_, T2_, st2, 1);                                             \
        if constexpr ((TAP) < HALO_PPW) {                                                                             \
            RCV_HALO_PATCH(sl + 1, TAP);                                                                       \
            if constexpr (BPW == 2) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");                \
            else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");                                   \
        } else {                                                                                               \
            if constexpr (BPW == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");                \
            else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");                                   \
        }                                                                                                      \
        __builtin_amdgcn_s_barrier();                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; i++)                                                         \
            _Pragma("unroll") for (int j = 0; j < NT; j++)                                                     \
                acc[i][j] = RCV_MFMA16(af[i], bfr[j], acc[i][j], 0, 0, 0);        \
        __builtin_amdgcn_s_setprio(0);                                                                         \
        __builtin_amdgcn_s_barrier();                                                                          \
        stage = stage + 1 == 3 ? 0 : stage + 1;                                                                \
    }

// BN = 64: half the MFMAs per step, so one load phase (fragments of all 64 K columns) and one MFMA phase per step -- two
// barriers instead of four.  Same distances: data is waited for in the load phase one step (two barriers) before it is
// read, a buffer is refilled two barriers after its last read.
#define RCV_HALO_STEP2(TAP)                                                                                    \
    {                                                                                                          \
        constexpr int T2_ = ((TAP) + 2) % 9, S2_ = ((TAP) + 2) / 9;                                            \
        const int shift_ = ((TAP) / 3 - 1) * pitch + ((TAP) % 3 - 1);                                          \
        const unsigned char *Pb = sP + (sl & 1) * HALO_PATCH;                                                  \
        const unsigned char *Bb = sB + stage * (BN * 128) + (wn * (BN / 2) + r16) * 128;                       \
        const int st2 = stage + 2 >= 3 ? stage + 2 - 3 : stage + 2;                                            \
        bf16x8 af[2][MT], bfr[2][NT];                                                                          \
        _Pragma("unroll") for (int i = 0; i < MT; i++) {                                                       \
            const int ao_ = (pr0[i] + shift_) * 128 + ((kq ^ ((r16 + ((TAP) / 3) * p.tw + (TAP) % 3) & 7)) << 4); \
            af[0][i] = *(const bf16x8 *)(Pb + ao_);                                                            \
            af[1][i] = *(const bf16x8 *)(Pb + (ao_ ^ 64));                                                     \
        }                                                                                                      \
        _Pragma("unroll") for (int j = 0; j < NT; j++) {                                                       \
            bfr[0][j] = *(const bf16x8 *)(Bb + j * 16 * 128 + ch0);                                            \
            bfr[1][j] = *(const bf16x8 *)(Bb + j * 16 * 128 + ch1);                                            \
        }                                                                                                      \
        RCV_HALO_B(sl + S2_, T2_, st2, 0);                                                                     \
        if constexpr ((TAP) < HALO_PPW) {                                                                             \
            RCV_HALO_PATCH(sl + 1, TAP);                                                                       \
            asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");                                        \
        } else {                                                                                               \
            asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");                                        \
        }                                                                                                      \
        __builtin_amdgcn_s_barrier();                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        _Pragma("unroll") for (int kk = 0; kk < 2; kk++)                                                       \
            _Pragma("unroll") for (int i = 0; i < MT; i++)                                                     \
                _Pragma("unroll") for (int j = 0; j < NT; j++)                                                 \
                    acc[i][j] = RCV_MFMA16(af[kk][i], bfr[kk][j], acc[i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                         \
        __builtin_amdgcn_s_barrier();                                                                          \
        stage = stage + 1 == 3 ? 0 : stage + 1;                                                                \
    }

// Split precision (see k_conv_dma): the patch row / weight row is [32 hi | 32 lo]; hi*hi, hi*lo, lo*hi dealt 24 / 24 over the
// two MFMA phases, per output element always in the order hh, hl, lh.
#ifdef RCV_HALO_STAMPS
#define RCV_HSTAMP_X(TAP, idx) if constexpr ((TAP) == 4) { RCV_HSTAMP(idx) }
#else
#define RCV_HSTAMP_X(TAP, idx)
#endif
// (TAILC: operands swapped -- a lane then holds four consecutive CHANNELS of one pixel instead of four pixels of one channel,
// which is what the register-only head epilogue wants; the same products in the same K order)
#define RCV_HALO_MM(a_, b_, c_) (HALO_SWAP ? RCV_MFMA16(b_, a_, c_, 0, 0, 0) : RCV_MFMA16(a_, b_, c_, 0, 0, 0))
// all (tile row i, tile column j in [J0, J1)) products of one operand pair, tile row outermost.  (r06, measured with the operands
// swapped: the column-outermost order, which keeps the MFMA's A operand across consecutive instructions, ran the K loop 26 % SLOWER;
// the swap itself costs the row-outermost loop 5-8 % -- which is why only the composed tail, whose epilogue gains more, swaps)
#define RCV_HALO_MMLOOP(A_, B_, J0, J1)                                                                        \
    _Pragma("unroll") for (int i = 0; i < MT; i++)                                                             \
        _Pragma("unroll") for (int j = (J0); j < (J1); j++) acc[i][j] = RCV_HALO_MM(A_[i], B_[j], acc[i][j]);
#define RCV_HALO_STEP_X3(TAP)                                                                                  \
    {                                                                                                          \
        constexpr int T2_ = ((TAP) + 2) % 9, S2_ = ((TAP) + 2) / 9;                                            \
        RCV_HSTAMP_X(TAP, 0);                                                                                  \
        const int shift_ = ((TAP) / 3 - 1) * pitch + ((TAP) % 3 - 1);                                          \
        const unsigned char *Pb = sP + (sl & 1) * HALO_PATCH;                                                  \
        const unsigned char *Bb = sB + stage * (BN * 128) + (wn * (BN / 2) + r16) * 128;                       \
        const int st2 = stage + 2 >= 3 ? stage + 2 - 3 : stage + 2;                                            \
        int ao_[MT];                                                                                           \
        const int kx_ = (kq ^ ((r16 + ((TAP) / 3) * p.tw + (TAP) % 3) & 7)) << 4;                              \
        _Pragma("unroll") for (int i = 0; i < MT; i++) ao_[i] = (pr0[i] + shift_) * 128 + kx_;                 \
        bf16x8 ah[MT], al[MT], bh[NT], bl[NT];                                                                 \
        _Pragma("unroll") for (int i = 0; i < MT; i++) ah[i] = *(const bf16x8 *)(Pb + ao_[i]);                 \
        _Pragma("unroll") for (int j = 0; j < NT; j++) bh[j] = *(const bf16x8 *)(Bb + j * 16 * 128 + ch0);     \
        _Pragma("unroll") for (int j = 0; j < NT / 2; j++) bl[j] = *(const bf16x8 *)(Bb + j * 16 * 128 + ch1); \
        RCV_HALO_B(sl + S2_, T2_, st2, 0);                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
        RCV_HSTAMP_X(TAP, 1);                                                                                  \
        __builtin_amdgcn_s_barrier();                                                                          \
        RCV_HSTAMP_X(TAP, 2);                                                                                  \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        RCV_HALO_MMLOOP(ah, bh, 0, NT);                                                                        \
        RCV_HALO_MMLOOP(ah, bl, 0, NT / 2);                                                                    \
        __builtin_amdgcn_s_setprio(0);                                                                         \
        RCV_HSTAMP_X(TAP, 3);                                                                                  \
        __builtin_amdgcn_s_barrier();                                                                          \
        RCV_HSTAMP_X(TAP, 4);                                                                                  \
        _Pragma("unroll") for (int i = 0; i < MT; i++) al[i] = *(const bf16x8 *)(Pb + (ao_[i] ^ 64));          \
        _Pragma("unroll") for (int j = NT / 2; j < NT; j++) bl[j] = *(const bf16x8 *)(Bb + j * 16 * 128 + ch1); \
        if constexpr (BPW == 2) RCV_HALO_B(sl + S2_, T2_, st2, 1);                                             \
        if constexpr ((TAP) < HALO_PPW) {                                                                             \
            RCV_HALO_PATCH(sl + 1, TAP);                                                                       \
            if constexpr (BPW == 2) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");                \
            else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");                                   \
        } else {                                                                                               \
            if constexpr (BPW == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");                \
            else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");                                   \
        }                                                                                                      \
        RCV_HSTAMP_X(TAP, 5);                                                                                  \
        __builtin_amdgcn_s_barrier();                                                                          \
        RCV_HSTAMP_X(TAP, 6);                                                                                  \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        RCV_HALO_MMLOOP(ah, bl, NT / 2, NT);                                                                   \
        RCV_HALO_MMLOOP(al, bh, 0, NT);                                                                        \
        __builtin_amdgcn_s_setprio(0);                                                                         \
        RCV_HSTAMP_X(TAP, 7);                                                                                  \
        __builtin_amdgcn_s_barrier();                                                                          \
        RCV_HSTAMP_X(TAP, 8);                                                                                  \
        stage = stage + 1 == 3 ? 0 : stage + 1;                                                                \
    }
// Split precision with the cross terms in fp6 (Q6, r05; engine option 11).  The source tensor and the weights are in the q6 row
// layout (k_q6_pack, include/rcvpose_hip.h): [32 hi halves | 4 x (16 e2m3 codes, block scale)] per 32 channels, so the second
// half of every patch / weight row is read with the very same ds_read_b128 as a lo fragment was -- chunk 4 + kq is block kq of
// v_mfma_scale_f32_16x16x128_f8f6f4 (A: a_hi6[0:16], a_hi6[16:32], a_lo6[0:16], a_lo6[16:32]; B: w_lo6, w_lo6, w_hi6, w_hi6), whose
// elements 16 .. 31 are zero on the B side (registers 3 .. 5: the scale byte sits where the A side has zeros, and the other way
// round; registers 4 .. 5 of the A operand may hold anything: e2m3 has no NaN).  One scaled instruction (17.7 cycles, measured:
// tools/probe/fp6_mfma_probe.hip) replaces the two half-precision MFMAs hi*lo + lo*hi (2 x 17.3): 40 instead of 60 per step,
// 20 (hi*hi) in the first MFMA phase, 20 (cross) in the second.  NOT bit-identical to the three-MFMA form: tolerance-tested.
#define RCV_HALO_STEP_X3Q(TAP)                                                                                 \
    {                                                                                                          \
        constexpr int T2_ = ((TAP) + 2) % 9, S2_ = ((TAP) + 2) / 9;                                            \
        const int shift_ = ((TAP) / 3 - 1) * pitch + ((TAP) % 3 - 1);                                          \
        const unsigned char *Pb = sP + (sl & 1) * HALO_PATCH;                                                  \
        const unsigned char *Bb = sB + stage * (BN * 128) + (wn * (BN / 2) + r16) * 128;                       \
        const int st2 = stage + 2 >= 3 ? stage + 2 - 3 : stage + 2;                                            \
        int ao_[MT];                                                                                           \
        const int kx_ = (kq ^ ((r16 + ((TAP) / 3) * p.tw + (TAP) % 3) & 7)) << 4;                              \
        _Pragma("unroll") for (int i = 0; i < MT; i++) ao_[i] = (pr0[i] + shift_) * 128 + kx_;                 \
        bf16x8 ah[MT], bh[NT];                                                                                 \
        i32x4 aq[MT], bq[NT];                                                                                  \
        _Pragma("unroll") for (int i = 0; i < MT; i++) ah[i] = *(const bf16x8 *)(Pb + ao_[i]);                 \
        _Pragma("unroll") for (int j = 0; j < NT; j++) bh[j] = *(const bf16x8 *)(Bb + j * 16 * 128 + ch0);     \
        RCV_HALO_B(sl + S2_, T2_, st2, 0);                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
        __builtin_amdgcn_s_barrier();                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; i++)                                                         \
            _Pragma("unroll") for (int j = 0; j < NT; j++) acc[i][j] = RCV_MFMA16(ah[i], bh[j], acc[i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                         \
        __builtin_amdgcn_s_barrier();                                                                          \
        _Pragma("unroll") for (int i = 0; i < MT; i++) aq[i] = *(const i32x4 *)(Pb + (ao_[i] ^ 64));           \
        _Pragma("unroll") for (int j = 0; j < NT; j++) bq[j] = *(const i32x4 *)(Bb + j * 16 * 128 + ch1);      \
        if constexpr (BPW == 2) RCV_HALO_B(sl + S2_, T2_, st2, 1);                                             \
        if constexpr ((TAP) < HALO_PPW) {                                                                      \
            RCV_HALO_PATCH(sl + 1, TAP);                                                                       \
            if constexpr (BPW == 2) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");                \
            else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");                                   \
        } else {                                                                                               \
            if constexpr (BPW == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");                \
            else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");                                   \
        }                                                                                                      \
        __builtin_amdgcn_s_barrier();                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        _Pragma("unroll") for (int j = 0; j < NT; j++) {                                                       \
            const i32x8 b8_ = {bq[j][0], bq[j][1], bq[j][2], bq[j][3], 0, 0, 0, 0};                             \
            _Pragma("unroll") for (int i = 0; i < MT; i++)                                                     \
                acc[i][j] = q6_mfma(__builtin_shufflevector(aq[i], aq[i], 0, 1, 2, 3, -1, -1, -1, -1), b8_, acc[i][j], aq[i][3], bq[j][3]); \
        }                                                                                                      \
        __builtin_amdgcn_s_setprio(0);                                                                         \
        __builtin_amdgcn_s_barrier();                                                                          \
        stage = stage + 1 == 3 ? 0 : stage + 1;                                                                \
    }
// The q6 step with ONE load phase and ONE MFMA phase (two barriers per step, as RCV_HALO_STEP2_X3; -DRCV_Q6_TWO_PHASE=1).  Why it
// exists: with 40 instead of 60 MFMAs per step the four-phase form is bound by its load phases (~425 cycles each: nine fragment
// reads, the DMA issue, the counted waits) against 320 cycles of MFMAs -- rocprofv3 counted the matrix cores 0.59 busy in it
// (profiles/r05_q6_pmc.txt; the three-MFMA kernel: 0.85) -- which is why q6 buys -18.5 % and not -33 %.  Why it is off: see
// RCV_Q6_TWO_PHASE.  Same products in the same order per output element (hh, then the scaled cross instruction): identical results.
#define RCV_HALO_STEP2_X3Q(TAP)                                                                                \
    {                                                                                                          \
        constexpr int T2_ = ((TAP) + 2) % 9, S2_ = ((TAP) + 2) / 9;                                            \
        const int shift_ = ((TAP) / 3 - 1) * pitch + ((TAP) % 3 - 1);                                          \
        const unsigned char *Pb = sP + (sl & 1) * HALO_PATCH;                                                  \
        const unsigned char *Bb = sB + stage * (BN * 128) + (wn * (BN / 2) + r16) * 128;                       \
        const int st2 = stage + 2 >= 3 ? stage + 2 - 3 : stage + 2;                                            \
        bf16x8 ah[MT], bh[NT];                                                                                 \
        i32x4 aq[MT], bq[NT];                                                                                  \
        _Pragma("unroll") for (int i = 0; i < MT; i++) {                                                       \
            const int ao_ = (pr0[i] + shift_) * 128 + ((kq ^ ((r16 + ((TAP) / 3) * p.tw + (TAP) % 3) & 7)) << 4); \
            ah[i] = *(const bf16x8 *)(Pb + ao_);                                                               \
            aq[i] = *(const i32x4 *)(Pb + (ao_ ^ 64));                                                         \
        }                                                                                                      \
        _Pragma("unroll") for (int j = 0; j < NT; j++) {                                                       \
            bh[j] = *(const bf16x8 *)(Bb + j * 16 * 128 + ch0);                                                \
            bq[j] = *(const i32x4 *)(Bb + j * 16 * 128 + ch1);                                                 \
        }                                                                                                      \
        RCV_HALO_B(sl + S2_, T2_, st2, 0);                                                                     \
        if constexpr (BPW == 2) RCV_HALO_B(sl + S2_, T2_, st2, 1);                                             \
        if constexpr ((TAP) < HALO_PPW) {                                                                      \
            RCV_HALO_PATCH(sl + 1, TAP);                                                                       \
            if constexpr (BPW == 2) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");                \
            else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");                                   \
        } else {                                                                                               \
            if constexpr (BPW == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");                \
            else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");                                   \
        }                                                                                                      \
        __builtin_amdgcn_s_barrier();                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; i++)                                                         \
            _Pragma("unroll") for (int j = 0; j < NT; j++) acc[i][j] = RCV_MFMA16(ah[i], bh[j], acc[i][j], 0, 0, 0); \
        _Pragma("unroll") for (int j = 0; j < NT; j++) {                                                       \
            const i32x8 b8_ = {bq[j][0], bq[j][1], bq[j][2], bq[j][3], 0, 0, 0, 0};                             \
            _Pragma("unroll") for (int i = 0; i < MT; i++)                                                     \
                acc[i][j] = q6_mfma(__builtin_shufflevector(aq[i], aq[i], 0, 1, 2, 3, -1, -1, -1, -1), b8_, acc[i][j], aq[i][3], bq[j][3]); \
        }                                                                                                      \
        __builtin_amdgcn_s_setprio(0);                                                                         \
        __builtin_amdgcn_s_barrier();                                                                          \
        stage = stage + 1 == 3 ? 0 : stage + 1;                                                                \
    }
// BN = 64, split precision: one load phase, 24 MFMAs
#define RCV_HALO_STEP2_X3(TAP)                                                                                 \
    {                                                                                                          \
        constexpr int T2_ = ((TAP) + 2) % 9, S2_ = ((TAP) + 2) / 9;                                            \
        const int shift_ = ((TAP) / 3 - 1) * pitch + ((TAP) % 3 - 1);                                          \
        const unsigned char *Pb = sP + (sl & 1) * HALO_PATCH;                                                  \
        const unsigned char *Bb = sB + stage * (BN * 128) + (wn * (BN / 2) + r16) * 128;                       \
        const int st2 = stage + 2 >= 3 ? stage + 2 - 3 : stage + 2;                                            \
        bf16x8 af[2][MT], bfr[2][NT];                                                                          \
        _Pragma("unroll") for (int i = 0; i < MT; i++) {                                                       \
            const int ao_ = (pr0[i] + shift_) * 128 + ((kq ^ ((r16 + ((TAP) / 3) * p.tw + (TAP) % 3) & 7)) << 4); \
            af[0][i] = *(const bf16x8 *)(Pb + ao_);                                                            \
            af[1][i] = *(const bf16x8 *)(Pb + (ao_ ^ 64));                                                     \
        }                                                                                                      \
        _Pragma("unroll") for (int j = 0; j < NT; j++) {                                                       \
            bfr[0][j] = *(const bf16x8 *)(Bb + j * 16 * 128 + ch0);                                            \
            bfr[1][j] = *(const bf16x8 *)(Bb + j * 16 * 128 + ch1);                                            \
        }                                                                                                      \
        RCV_HALO_B(sl + S2_, T2_, st2, 0);                                                                     \
        if constexpr (BPW == 2) RCV_HALO_B(sl + S2_, T2_, st2, 1);                                             \
        if constexpr ((TAP) < HALO_PPW) {                                                                             \
            RCV_HALO_PATCH(sl + 1, TAP);                                                                       \
            if constexpr (BPW == 2) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");                \
            else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");                                   \
        } else {                                                                                               \
            if constexpr (BPW == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");                \
            else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");                                   \
        }                                                                                                      \
        __builtin_amdgcn_s_barrier();                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; i++)                                                         \
            _Pragma("unroll") for (int j = 0; j < NT; j++) acc[i][j] = RCV_HALO_MM(af[0][i], bfr[0][j], acc[i][j]); \
        _Pragma("unroll") for (int i = 0; i < MT; i++)                                                         \
            _Pragma("unroll") for (int j = 0; j < NT; j++) acc[i][j] = RCV_HALO_MM(af[0][i], bfr[1][j], acc[i][j]); \
        _Pragma("unroll") for (int i = 0; i < MT; i++)                                                         \
            _Pragma("unroll") for (int j = 0; j < NT; j++) acc[i][j] = RCV_HALO_MM(af[1][i], bfr[0][j], acc[i][j]); \
        __builtin_amdgcn_s_setprio(0);                                                                         \
        __builtin_amdgcn_s_barrier();                                                                          \
        stage = stage + 1 == 3 ? 0 : stage + 1;                                                                \
    }

    const int grp = wave >> 2;
#ifdef RCV_HALO_STAMPS
    unsigned hst[16];
    for (int k = 0; k < 16; k++) hst[k] = 0u;
    const bool stamp_blk = blockIdx.x + gridDim.x * blockIdx.y == (gridDim.x * gridDim.y) / 2 && blockIdx.z == 0 && (wave & 3) == 0;
    bool stamp_on = false;
#endif
// the first operands of the current item: its whole first patch and the weights of steps 0 and 1
#define RCV_HALO_PROLOGUE()                                                                                    \
    {                                                                                                          \
        _Pragma("unroll") for (int k = 0; k < HALO_PPW; k++) RCV_HALO_PATCH(0, k);                             \
        RCV_HALO_B(0, 0, 0, 0);                                                                                \
        if constexpr (BPW == 2) RCV_HALO_B(0, 0, 0, 1);                                                        \
        RCV_HALO_B(0, 1, 1, 0);                                                                                \
        if constexpr (BPW == 2) RCV_HALO_B(0, 1, 1, 1);                                                        \
    }
// PERSIST: on to item it_ and out with its first operands (LDS must be free: every wave past its last barrier / staging read)
#define RCV_HALO_NEXT_ITEM(it_)                                                                                \
    {                                                                                                          \
        next_item_setup(it_);                                                                                  \
        b_base = (const char *)((const T *)p.w + (long long)g * p.w_gs + (long long)n0 * p.K);                 \
        rA0 = __builtin_amdgcn_make_buffer_rsrc((void *)src0, 0, a_bytes0, 0x00020000);                        \
        rA1 = __builtin_amdgcn_make_buffer_rsrc((void *)(src1 ? src1 : src0), 0, src1 ? a_bytes1 : a_bytes0, 0x00020000); \
        rB = __builtin_amdgcn_make_buffer_rsrc((void *)b_base, 0, b_bytes, 0x00020000);                        \
        rAn = rA0;                                                                                             \
        soffA = 0;                                                                                             \
        xfn = false;                                                                                           \
        fill_lin();                                                                                            \
        RCV_HALO_PROLOGUE();                                                                                   \
    }
    RCV_HALO_PROLOGUE();
    bool first_item = true;
next_item_top:
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!PERSIST || first_item) {
        if constexpr (BPW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    } else  // (younger than the prologue's pieces: the previous item's stores)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    int stage = 0;
#ifndef RCV_X3_HALO_TWO_PHASE
#define RCV_X3_HALO_TWO_PHASE 0  // (r03 A-B on one box: 83.9 vs 83.8 ms per 36-frame forward -- the barrier count is not what the kernel waits for)
#endif
    if constexpr (Q6) {
        static_assert(!Q6 || BPW == 2, "the fp6 cross-term form exists for the 128-channel tile (of the split engine: q6_mfma)");
#ifndef RCV_Q6_TWO_PHASE
#define RCV_Q6_TWO_PHASE 0  // 1: one load phase + one MFMA phase per step (below).  Measured r05: 17.9 ms per 35 layer-3 launches against 12.3 in the four-phase form -- its 72 fragment registers beside the 6-wide operand tuples spill 22 registers, the reloads sit in the K loop and every scratch load joins the vmcnt queue of the DMA pieces
#endif
        for (int sl = 0; sl < nsl; sl++) {
            RCV_HALO_NEXT_SLICE(sl);
#if RCV_Q6_TWO_PHASE
            RCV_HALO_STEP2_X3Q(0); RCV_HALO_STEP2_X3Q(1); RCV_HALO_STEP2_X3Q(2);
            RCV_HALO_STEP2_X3Q(3); RCV_HALO_STEP2_X3Q(4); RCV_HALO_STEP2_X3Q(5);
            RCV_HALO_STEP2_X3Q(6); RCV_HALO_STEP2_X3Q(7); RCV_HALO_STEP2_X3Q(8);
#else
            RCV_HALO_STEP_X3Q(0); RCV_HALO_STEP_X3Q(1); RCV_HALO_STEP_X3Q(2);
            RCV_HALO_STEP_X3Q(3); RCV_HALO_STEP_X3Q(4); RCV_HALO_STEP_X3Q(5);
            RCV_HALO_STEP_X3Q(6); RCV_HALO_STEP_X3Q(7); RCV_HALO_STEP_X3Q(8);
#endif
        }
    } else if constexpr (BPW == 2 && kX3 && (!RCV_X3_HALO_TWO_PHASE || MT == 5)) {  // (the 320-row tile would spill in the two-phase form)
        for (int sl = 0; sl < nsl; sl++) {
            RCV_HALO_NEXT_SLICE(sl);
#ifdef RCV_HALO_STAMPS
            stamp_on = stamp_blk && sl == 2;
#endif
            RCV_HALO_STEP_X3(0); RCV_HALO_STEP_X3(1); RCV_HALO_STEP_X3(2);
            RCV_HALO_STEP_X3(3); RCV_HALO_STEP_X3(4); RCV_HALO_STEP_X3(5);
            RCV_HALO_STEP_X3(6); RCV_HALO_STEP_X3(7); RCV_HALO_STEP_X3(8);
        }
    } else if constexpr (kX3) {
        for (int sl = 0; sl < nsl; sl++) {
            RCV_HALO_NEXT_SLICE(sl);
            RCV_HALO_STEP2_X3(0); RCV_HALO_STEP2_X3(1); RCV_HALO_STEP2_X3(2);
            RCV_HALO_STEP2_X3(3); RCV_HALO_STEP2_X3(4); RCV_HALO_STEP2_X3(5);
            RCV_HALO_STEP2_X3(6); RCV_HALO_STEP2_X3(7); RCV_HALO_STEP2_X3(8);
        }
    } else if constexpr (BPW == 2) {
        for (int sl = 0; sl < nsl; sl++) {
            RCV_HALO_NEXT_SLICE(sl);
            RCV_HALO_STEP(0); RCV_HALO_STEP(1); RCV_HALO_STEP(2);
            RCV_HALO_STEP(3); RCV_HALO_STEP(4); RCV_HALO_STEP(5);
            RCV_HALO_STEP(6); RCV_HALO_STEP(7); RCV_HALO_STEP(8);
        }
    } else {
        for (int sl = 0; sl < nsl; sl++) {
            RCV_HALO_NEXT_SLICE(sl);
            RCV_HALO_STEP2(0); RCV_HALO_STEP2(1); RCV_HALO_STEP2(2);
            RCV_HALO_STEP2(3); RCV_HALO_STEP2(4); RCV_HALO_STEP2(5);
            RCV_HALO_STEP2(6); RCV_HALO_STEP2(7); RCV_HALO_STEP2(8);
        }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the zero pieces issued past the end
#if defined(RCV_HALO_STAMPS) && !defined(RCV_CHAIN_STAMPS)
    if (stamp_blk && BN == 128 && MT == 4 && lane == 0)
        for (int k = 0; k < 16; k++) g_halo_stamps[wave >> 2][k] = hst[k];
#endif
#undef RCV_HALO_STEP
#undef RCV_HALO_STEP2
#undef RCV_HALO_STEP_X3
#undef RCV_HALO_STEP_X3Q
#undef RCV_HALO_STEP2_X3Q
#undef RCV_HALO_STEP2_X3

    if constexpr (TAILC && kX3) {
        // ---- parity-composed tail: ReLU + the head straight from the accumulators, no LDS, no barrier.  With the operands swapped
        // (RCV_HALO_MM) lane (r16, kq) holds, of pixel r16 of tile i, the channels n = wn*64 + j*16 + kq*4 + e -- conv7's channel
        // c = (j & 1)*16 + kq*4 + e at the high-resolution pixel of parity 2*ppy + ppx = wn*2 + (j >> 1).  The head's two dot products
        // over the 32 channels are 8 FMAs per lane and (pixel, parity, plane) and a sum over the four kq: v_permlane16_swap on the
        // (seg, radial) pair + add, then v_permlane32_swap on the (ppx = 0, ppx = 1) pair + add -- a reduce-scatter that leaves lane
        // (r16, kq) with plane kq & 1 of column parity kq >> 1: the 32 lanes of a plane store 16 pixels x 2 columns = 128 contiguous bytes.
        static_assert(!TAILC || (BN == 128 && NT == 4), "four parities x 32 channels");
        const int e_g = g, e_img = img, e_y0 = y0, e_x0 = x0;  // this item's identity: the variables move on to the next item below
        int r16_ = r16, kq_ = kq;  // (as in fill_lin: nothing of the epilogue's index arithmetic may be hoisted across the K loop)
        asm volatile("" : "+v"(r16_), "+v"(kq_));
        // what the epilogue reads from memory is requested BEFORE the next item's DMA pieces (vector-memory results return in
        // order: behind them, the epilogue's first instruction would wait for the whole patch)
        const float *bias = p.bias + (long long)e_g * p.bias_gs;
        const H16 *w8 = (const H16 *)p.w8 + (long long)e_g * p.w8_gs;
        float4 bvr[NT], scr[NT];
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int nl = n0 + wn * (BN / 2) + j * 16 + kq_ * 4;
            bvr[j] = *(const float4 *)(bias + nl);
            scr[j] = *(const float4 *)(bias + p.Cout + nl);
        }
        float h0[2][4], h1[2][4];
#pragma unroll
        for (int jj = 0; jj < 2; jj++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int c = jj * 16 + kq_ * 4 + e;
                h0[jj][e] = (float)w8[c] + (float)w8[64 + c];
                h1[jj][e] = (float)w8[32 + c] + (float)w8[96 + c];
            }
        float hb = p.b8[(long long)e_g * p.b8_gs + (kq_ & 1)], hs = p.b8[(long long)e_g * p.b8_gs + 2 + (kq_ & 1)];
        asm volatile("" : "+v"(hb), "+v"(hs), "+v"(h0[0][0]), "+v"(h1[1][3]), "+v"(bvr[0].x), "+v"(scr[NT - 1].w));  // (issued here, not sunk below the DMA)
        const int next_item = item + item_step;
        const bool has_next = PERSIST && next_item < item_end;
        if (has_next) {  // the next item's first operands: LDS is free (every wave has passed its last barrier; this wave's own pieces have landed)
            RCV_HALO_NEXT_ITEM(next_item);
        }
        float part[MT][2][2];
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
            for (int q = 0; q < 4; q++) part[i][q >> 1][q & 1] = 0.f;
        float rmax = 0.f;
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const float bva[4] = {bvr[j].x, bvr[j].y, bvr[j].z, bvr[j].w}, sca[4] = {scr[j].x, scr[j].y, scr[j].z, scr[j].w};
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float v = fmaxf(__fmaf_rn(acc[i][j][e], sca[e], bva[e]), 0.f);
                    rmax = fmaxf(rmax, v);
                    part[i][j >> 1][0] = fmaf(v, h0[j & 1][e], part[i][j >> 1][0]);
                    part[i][j >> 1][1] = fmaf(v, h1[j & 1][e], part[i][j >> 1][1]);
                }
        }
        const int Ho = 2 * p.H, Wo = 2 * p.W;
#pragma unroll
        for (int i = 0; i < MT; i++) {
            float s2[2];
#pragma unroll
            for (int pp = 0; pp < 2; pp++) {  // even rows of lanes: seg summed over a kq pair; odd rows: radial
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(part[i][pp][0]), __float_as_uint(part[i][pp][1]), false, false);
                s2[pp] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            }
            // lanes 0..31: column parity 0 summed over all four kq; lanes 32..63: column parity 1
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(s2[0]), __float_as_uint(s2[1]), false, false);
            const float outv = __fmaf_rn(__uint_as_float(sw[0]) + __uint_as_float(sw[1]), hs, hb);
            const int r = wm * RW + i * 16 + r16_;
            const int py = (int)(((float)r + 0.5f) * rT), px = r - py * p.tw;
            const int y = e_y0 + py, x = e_x0 + px;
            if (r < npx && y < p.H && x < p.W) {
                const long long o = ((long long)(e_g * p.out_b + p.out_img0 + e_img) * Ho + (2 * y + wn)) * Wo + (2 * x + (kq_ >> 1));
                if ((kq_ & 1) == 0) p.seg[o] = outv;
                else if (p.rad_f16) ((_Float16 *)p.radial)[o] = (_Float16)outv;
                else p.radial[o] = outv;
            }
        }
        rng_flush(p.range, rmax);
        if (has_next) {
            first_item = false;
            goto next_item_top;
        }
        return;
    }
    if constexpr (REGEPI && !TAILC) {
        // ---- scale, bias, ReLU, range guard, split, store -- from registers: lane (r16, kq) holds channels nl .. nl + 3 of pixel r16 of
        // tile i (8 bytes of hi halves, 8 bytes of lo halves per tile; a pixel's 128-byte row fills over two tiles).  Element for
        // element the operations of the LDS-staged epilogue below: bit-identical tensors.
        const int e_g = g, e_n0 = n0, e_img = img, e_y0 = y0, e_x0 = x0;
        const int lane_ = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        int r16_ = lane_ & 15, kq_ = lane_ >> 4;  // (from the exec mask and opaque: nothing of the index arithmetic may be hoisted across the K loop)
        asm volatile("" : "+v"(r16_), "+v"(kq_));
        const float *bias = p.bias + (long long)e_g * p.bias_gs;
        float4 bvr[NT], scr[NT];  // requested BEFORE the next item's DMA pieces: vector-memory results return in order
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int nl = e_n0 + wn * (BN / 2) + j * 16 + kq_ * 4;
            bvr[j] = *(const float4 *)(bias + nl);
            scr[j] = *(const float4 *)(bias + p.Cout + nl);
        }
        asm volatile("" : "+v"(bvr[0].x), "+v"(scr[NT - 1].w));
        const int next_item = item + item_step;
        const bool has_next = next_item < item_end;
        if (has_next) RCV_HALO_NEXT_ITEM(next_item);
        T *dst = (T *)p.dst + (long long)e_g * p.dst_gs;
        float rmax = 0.f;
#pragma unroll
        for (int i = 0; i < MT; i++) {
            const int r = wm * RW + i * 16 + r16_;
            const int py = (int)(((float)r + 0.5f) * rT), px = r - py * p.tw;
            const int y = e_y0 + py, x = e_x0 + px;
            const bool ok = r < npx && y < p.H && x < p.W;
            const long long m = (long long)(e_img * p.H + y) * p.W + x;
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const int nl = e_n0 + wn * (BN / 2) + j * 16 + kq_ * 4;
                const float bva[4] = {bvr[j].x, bvr[j].y, bvr[j].z, bvr[j].w}, sca[4] = {scr[j].x, scr[j].y, scr[j].z, scr[j].w};
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    v[e] = __fmaf_rn(acc[i][j][e], sca[e], bva[e]);
                    if (p.relu) v[e] = fmaxf(v[e], 0.f);
                }
                rmax = rng_see<4>(rmax, v);
                uint2 oh, ol;
                x3_split2(v[0], v[1], oh.x, ol.x);
                x3_split2(v[2], v[3], oh.y, ol.y);
                if (ok) {
                    T *op = dst + nhwc_off<T>(m, p.Cout, nl);
                    *(uint2 *)op = oh;
                    *(uint2 *)(op + 32) = ol;
                }
            }
        }
        rng_flush(p.range, rmax);
        if (has_next) {
            first_item = false;
            goto next_item_top;
        }
        return;
    }
    // (persistent form: the epilogue's per-thread index arithmetic is the same for every item -- hidden from the optimiser, or it is
    // hoisted out of the item loop and its ~25 registers live, and spill, across the K loop: that, not the loop's own code, is what
    // made the first persistent attempt of this round 4-8 % slower)
    int tid_e = tid, r16_e = r16, kq_e = kq;
    if constexpr (PERSIST) {
        const int lane_ = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        tid_e = wave * 64 + lane_;
        r16_e = lane_ & 15;
        kq_e = lane_ >> 4;
        asm volatile("" : "+v"(tid_e), "+v"(r16_e), "+v"(kq_e));
    }
    // epilogue: k_conv_dma's (fp32 tile staged in LDS, 16-byte vectors of contiguous channels), in two halves of 2 * RW rows
    // (the 320-row tile would not fit at once); tile row -> image pixel
    T *dst = (T *)p.dst + (long long)g * p.dst_gs;
    const float *bias = p.bias + (long long)g * p.bias_gs;
    constexpr int SROW = BN + 4, CPR = BN / EPV, RPR = 512 / CPR;  // rows per pass of the 512 threads
    float *stg = (float *)smem;
    float rmax = 0.f;
    __syncthreads();
#pragma unroll
    for (int half = 0; half < 2; half++) {
        if ((wm >> 1) == half) {
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const int nl = wn * (BN / 2) + j * 16 + r16_e;
                const float bv = bias[n0 + nl];
                const float sc = kX3 ? bias[p.Cout + n0 + nl] : 1.f;  // (split mode: the weights' power-of-two scale, undone here)
#pragma unroll
                for (int i = 0; i < MT; i++)
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        stg[((wm & 1) * RW + i * 16 + kq_e * 4 + e) * SROW + nl] = kX3 ? __fmaf_rn(acc[i][j][e], sc, bv) : acc[i][j][e] + bv;
            }
        }
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < (2 * RW + RPR - 1) / RPR; pass++) {
            const int lr = pass * RPR + tid_e / CPR, cc = tid_e % CPR;  // row within the half, 16-byte chunk
            const int r = half * 2 * RW + lr;
            const int py = (int)(((float)r + 0.5f) * rT), px = r - py * p.tw;
            const int y = y0 + py, x = x0 + px;
            if (lr >= 2 * RW || r >= npx || y >= p.H || x >= p.W) continue;
            const long long o = nhwc_off<T>((long long)(img * p.H + y) * p.W + x, p.Cout, n0 + cc * EPV);
            float v[EPV];
            const float4 *sp4 = (const float4 *)(stg + lr * SROW + cc * EPV);
#pragma unroll
            for (int qq = 0; qq < EPV / 4; qq++) {
                const float4 t = sp4[qq];
                v[qq * 4 + 0] = t.x; v[qq * 4 + 1] = t.y; v[qq * 4 + 2] = t.z; v[qq * 4 + 3] = t.w;
            }
            if (p.relu) {
#pragma unroll
                for (int qq = 0; qq < EPV; qq++) v[qq] = fmaxf(v[qq], 0.f);
            }
            rmax = rng_see<EPV>(rmax, v);
            if constexpr (kX3) {
                uint4 oh, ol;
                x3_split8(v, oh, ol);
                *(uint4 *)(dst + o) = oh;
                *(uint4 *)(dst + o + 32) = ol;
            } else {
                uint4 outv;
                T *op = (T *)&outv;
#pragma unroll
                for (int qq = 0; qq < EPV; qq++) op[qq] = from_f32<T>(v[qq]);
                *(uint4 *)(dst + o) = outv;
            }
        }
        if (half == 0) __syncthreads();
    }
    rng_flush(p.range, rmax);
    if constexpr (PERSIST) {
        if (item + item_step < item_end) {
            __syncthreads();  // every thread is done reading the staging tile: the next item's patch may land on it
            RCV_HALO_NEXT_ITEM(item + item_step);
            first_item = false;
            goto next_item_top;
        }
    }
#undef RCV_HALO_B
#undef RCV_HALO_PATCH
#undef RCV_HALO_PROLOGUE
#undef RCV_HALO_NEXT_SLICE
#undef RCV_HALO_NEXT_ITEM
#undef RCV_HALO_MM
#undef RCV_HALO_MMLOOP
}

// ---------------------------------------------------------------------------------------------
// k_stem_mfma (bf16): the same 7x7/2 stem as an MFMA GEMM.  A workgroup owns 8 x 32 output pixels of one image and
// serves every group (keypoint network) from one im2col image:
//   1. the fp32 input patch (3 x 21 x 69) is staged in LDS,
//   2. every lane expands the 147 taps of its four pixels to bf16 MFMA A fragments in registers (k = (r*7+s)*3+c, the
//      weight order; k 147..159 zero) -- built once, used by all groups,
//   3. per group the [64][147] weights are laid into the B tile [64][160] (zero padded), 4 waves x (64 px x 64 ch)
//      run 80 MFMAs (16x16x32, operands swapped so that a lane holds consecutive channels), and the result leaves
//      through v_permlane16_swap as 16-byte NHWC vectors after bias + ReLU.
// The input is rounded to bf16 (the fp32 path keeps k_stem<float>); accumulation is fp32.
// ---------------------------------------------------------------------------------------------
constexpr int STEM_TH = 8, STEM_TW = 32, STEM_KP = 160, STEM_PITCH = 336;  // bytes per LDS row: 160 bf16 + 16 (bank spread)
constexpr int STEM_IH = STEM_TH * 2 + 5, STEM_IW = STEM_TW * 2 + 5;
constexpr int STEM_MAXG = 3;   // networks whose weights stay in LDS side by side (more: the kernel is launched per slice of groups)
constexpr int STEM_RUN = 5;    // consecutive tiles of a tile row per workgroup
constexpr int STEM_NH = kX3 ? 2 : 1;  // 16-bit planes per value (split precision: hi and lo)
constexpr int STEM_LDS = STEM_NH * (STEM_MAXG * 64 * STEM_PITCH + 3 * STEM_IH * (STEM_IW + 1) * 2) + STEM_KP * 4;  // weights, input patch (16-bit), tap offsets: 74 KB (split: 147 KB)

// (r02) The im2col tile no longer exists in LDS: every lane builds the MFMA A fragments of its own pixels straight from the
// input patch (the same bf16 values in the same k order, so the result is unchanged) and keeps them in registers for all
// groups.  The weights of all groups are laid into LDS ONCE per workgroup, which then walks STEM_RUN tiles of its tile row
// (before: 21 KB of weights scattered into LDS per group and tile); the patch is stored as 16-bit values -- the rounding the
// fragments need anyway -- so that two workgroups still share a CU and one's loads / stores overlap the other's MFMAs.
// Split precision (kX3): the image and the weights are half hi / lo pairs -- two patch planes, two weight tiles per group
// ([64*147 hi][64*147 lo] in global memory, scaled per output channel: bias[64 + co] undoes it), three MFMAs per (pixel group,
// channel tile, K chunk): hi*hi, hi*lo, lo*hi.  147 KB of LDS: one workgroup per CU.
__global__ __launch_bounds__(256, (kX3 ? 1 : 2)) void k_stem_mfma(const float *x, const H16 *w, const float *bias, H16 *dst, int H, int W,
                                                      int Ho, int Wo, int groups, long long w_gs, int bias_gs, long long dst_gs, unsigned *range)
{
    RCV_X3_MODE();
    using T = H16;
    float rmax = 0.f;  // range guard
    constexpr int NH = STEM_NH;
    constexpr int WPLANE = STEM_MAXG * 64 * STEM_PITCH, IPLANE = 3 * STEM_IH * (STEM_IW + 1) * 2;  // bytes of one weight / patch plane
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *sBall = smem;                                // [NH][groups][64][STEM_PITCH]
    T(*s_in)[STEM_IH][STEM_IW + 1] = (T(*)[STEM_IH][STEM_IW + 1])(smem + NH * WPLANE);  // [NH][3][IH][IW + 1]
    int *s_koff = (int *)(smem + NH * (WPLANE + IPLANE));  // [STEM_KP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (Wo + STEM_TW - 1) / STEM_TW, runs_x = (tiles_x + STEM_RUN - 1) / STEM_RUN;
    const int ty = blockIdx.x / runs_x, run = blockIdx.x - ty * runs_x;
    const int img = blockIdx.y;
    const int ho0 = ty * STEM_TH, hi0 = ho0 * 2 - 3;
    // ---- once per workgroup: every group's [64][147] weights into its B tile [64][160] (zero padded)
    for (int gh = 0; gh < groups * NH; gh++) {
        const int g = gh / NH, hp = gh - g * NH;  // group, plane (0: hi, 1: lo)
        const T *wg = w + (long long)g * w_gs + hp * (64 * 147);
        unsigned char *sB = sBall + hp * WPLANE + g * 64 * STEM_PITCH;
        // 64 x 147 contiguous weights as 16-byte vectors (1176 of them): all of a thread's loads are issued before the
        // first LDS write; the eight elements of a vector are scattered to their (out channel, k) cells
        constexpr int NV = 64 * 147 / 8, PER = (NV + 255) / 256;
        uint4 wv[PER];
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int i = u * 256 + tid;
            wv[u] = *(const uint4 *)(wg + (i < NV ? i : 0) * 8);
        }
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int i = u * 256 + tid;
            if (i < NV) {
                int co = (i * 8) / 147, k = i * 8 - co * 147;
                const T *pv = (const T *)&wv[u];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    *(T *)(sB + co * STEM_PITCH + k * 2) = pv[e];
                    if (++k == 147) { k = 0; co++; }
                }
            }
        }
        for (int i = tid; i < 64 * (STEM_KP - 147); i += 256) {
            const int co = i / (STEM_KP - 147), k = 147 + i - co * (STEM_KP - 147);
            *(T *)(sB + co * STEM_PITCH + k * 2) = (T)0.f;
        }
    }
    if (tid < STEM_KP) {  // patch offset of tap k = (r*7 + s)*3 + c; the padding taps (k >= 147) are zeroed below
        const int k = tid < 147 ? tid : 0;
        const int c = k % 3, rs = k / 3, r = rs / 7, q = rs - r * 7;
        s_koff[tid] = (c * STEM_IH + r) * (STEM_IW + 1) + q;
    }
    const int r16 = lane & 15, kq = lane >> 4;
    const int e_ch = (kq & 1) * 16 + (kq >> 1) * 8;  // + 32 for the second channel-tile pair
    // the input patch of a tile: all of a thread's loads are in flight together (a rolled loop at the top of every tile waited
    // for HBM 17 times per tile).  Issuing the NEXT tile's loads before the MFMA phase costs 17 more live registers: spills.
    constexpr int NE = 3 * STEM_IH * STEM_IW, PERT = (NE + 255) / 256;
#define RCV_STEM_PATCH(tx_)                                                                                    \
    {                                                                                                          \
        const int wi0_ = (tx_) * STEM_TW * 2 - 3;                                                              \
        _Pragma("unroll") for (int u = 0; u < PERT; u++) {                                                     \
            const int i = u * 256 + tl;                                                                       \
            const int c = i / (STEM_IH * STEM_IW), rem = i - c * (STEM_IH * STEM_IW);                          \
            const int r = rem / STEM_IW, q = rem - r * STEM_IW;                                                \
            const int hi = hi0 + r, wi = wi0_ + q;                                                             \
            const bool in = i < NE && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;                \
            pv[u] = x[in ? ((long long)(img * 3 + c) * H + hi) * W + wi : 0];                                  \
            pv[u] = in ? pv[u] : 0.f;                                                                          \
        }                                                                                                      \
    }
    const int tx_end = min((run + 1) * STEM_RUN, tiles_x);
    for (int tx = run * STEM_RUN; tx < tx_end; tx++) {
        const int wo0 = tx * STEM_TW;
        {
            float pv[PERT];
            int tl = tid;
            asm volatile("" : "+v"(tl));  // keeps the 17 address computations inside the loop (hoisted, they spill)
            RCV_STEM_PATCH(tx);
            __syncthreads();  // the previous tile's fragments are built: the patch may be overwritten (first tile: weights / offsets done)
#pragma unroll
            for (int u = 0; u < PERT; u++) {
                const int i = u * 256 + tl;
                if (i < NE) {
                    const int c = i / (STEM_IH * STEM_IW), rem = i - c * (STEM_IH * STEM_IW);
                    const int r = rem / STEM_IW, q = rem - r * STEM_IW;
                    if constexpr (kX3) x3_split(pv[u], s_in[c][r][q], s_in[3 + c][r][q]);
                    else s_in[c][r][q] = (T)pv[u];
                }
            }
        }
        __syncthreads();
        // A fragments of this lane: pixel group i (pixel wave*64 + i*16 + r16), K chunk ks: k = ks*32 + kq*8 + e
        bf16x8 afr[NH][4][STEM_KP / 32];
        {
            int koff[STEM_KP / 32][8];
#pragma unroll
            for (int ks = 0; ks < STEM_KP / 32; ks++)
#pragma unroll
                for (int e = 0; e < 8; e++) koff[ks][e] = s_koff[ks * 32 + kq * 8 + e];
#pragma unroll
            for (int hp = 0; hp < NH; hp++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int pix = wave * 64 + i * 16 + r16, py = pix >> 5, px = pix & 31;
                    const T *base = &s_in[3 * hp][py * 2][px * 2];
#pragma unroll
                    for (int ks = 0; ks < STEM_KP / 32; ks++)
#pragma unroll
                        for (int e = 0; e < 8; e++) afr[hp][i][ks][e] = ks * 32 + kq * 8 + e < 147 ? base[koff[ks][e]] : (T)0.f;
                }
        }
        for (int g = 0; g < groups; g++) {
            const unsigned char *sB = sBall + g * 64 * STEM_PITCH;
            f32x4 acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < STEM_KP / 32; ks++) {
                bf16x8 bfr[NH][4];
#pragma unroll
                for (int hp = 0; hp < NH; hp++)
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        bfr[hp][j] = *(const bf16x8 *)(sB + hp * WPLANE + (j * 16 + r16) * STEM_PITCH + (ks * 32 + kq * 8) * 2);
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[i][j] = RCV_MFMA16(bfr[0][j], afr[0][i][ks], acc[i][j], 0, 0, 0);
                if constexpr (kX3) {  // image hi x weight lo, image lo x weight hi
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 4; j++) acc[i][j] = RCV_MFMA16(bfr[NH - 1][j], afr[0][i][ks], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 4; j++) acc[i][j] = RCV_MFMA16(bfr[0][j], afr[NH - 1][i][ks], acc[i][j], 0, 0, 0);
                }
            }
            // acc[i][j][e] = channel j*16 + kq*4 + e of pixel wave*64 + i*16 + r16 (operands swapped); see k_conv_dma's
            // notes on the permlane16 swap: afterwards the lane owns 8 consecutive channels
            const float *bg = bias + (long long)g * bias_gs;
            T *out = dst + (long long)g * dst_gs;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int pix = wave * 64 + i * 16 + r16;
                const int ho = ho0 + (pix >> 5), wo = wo0 + (pix & 31);
#pragma unroll
                for (int pr = 0; pr < 2; pr++) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[i][2 * pr][e]), __float_as_uint(acc[i][2 * pr + 1][e]), false, false);
                        v[e] = __uint_as_float(sw[0]);
                        v[4 + e] = __uint_as_float(sw[1]);
                    }
                    const int ch = e_ch + pr * 32;
                    const float4 b0 = *(const float4 *)(bg + ch), b1 = *(const float4 *)(bg + ch + 4);
                    if constexpr (kX3) {
                        const float4 s0 = *(const float4 *)(bg + 64 + ch), s1 = *(const float4 *)(bg + 64 + ch + 4);
                        v[0] = __fmaf_rn(v[0], s0.x, b0.x); v[1] = __fmaf_rn(v[1], s0.y, b0.y); v[2] = __fmaf_rn(v[2], s0.z, b0.z); v[3] = __fmaf_rn(v[3], s0.w, b0.w);
                        v[4] = __fmaf_rn(v[4], s1.x, b1.x); v[5] = __fmaf_rn(v[5], s1.y, b1.y); v[6] = __fmaf_rn(v[6], s1.z, b1.z); v[7] = __fmaf_rn(v[7], s1.w, b1.w);
#pragma unroll
                        for (int qq = 0; qq < 8; qq++) v[qq] = fmaxf(v[qq], 0.f);
                        rmax = rng_see<8>(rmax, v);
                        uint4 oh, ol;
                        x3_split8(v, oh, ol);
                        T *op_ = out + ((long long)(img * Ho + ho) * Wo + wo) * 128 + x3_off(ch);
                        if (ho < Ho && wo < Wo) { *(uint4 *)op_ = oh; *(uint4 *)(op_ + 32) = ol; }
                    } else {
                    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
                    v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                    rmax = rng_see<8>(rmax, v);
                    uint4 outv;
                    T *op = (T *)&outv;
#pragma unroll
                    for (int qq = 0; qq < 8; qq++) op[qq] = (T)fmaxf(v[qq], 0.f);
                    if (ho < Ho && wo < Wo) *(uint4 *)(out + ((long long)(img * Ho + ho) * Wo + wo) * 64 + ch) = outv;
                    }
                }
            }
        }
    }
#undef RCV_STEM_PATCH
    rng_flush(range, rmax);
}

// bilinear blend shared by k_upsample2 and the fused tail kernel: hy * (hx * p00 + lx * p01) + ly * (hx * p10 + lx * p11)
// with the fusion of multiplies and adds written out, so that every caller rounds identically whatever the compiler
// would contract in its context
__device__ __forceinline__ float bilerp(float hy, float hx, float ly, float lx, float p00, float p01, float p10, float p11)
{
    const float t0 = __fmaf_rn(lx, p01, __fmul_rn(hx, p00));
    const float t1 = __fmaf_rn(lx, p11, __fmul_rn(hx, p10));
    return __fmaf_rn(hy, t0, __fmul_rn(ly, t1));
}
// the same on eight channels, two per instruction (v_pk_mul_f32 / v_pk_fma_f32), in two stages so that the horizontal blends
// of a source cell are formed once per destination column: element for element bilerp's operations
__device__ __forceinline__ void bilerp8_rows(float hx, float lx, const float *p00, const float *p01, const float *p10, const float *p11,
                                             f32x2 *t0, f32x2 *t1)
{
#ifdef RCV_X3_SCALAR_HELPERS
#pragma unroll
    for (int e = 0; e < 8; e++) {
        t0[e >> 1][e & 1] = __fmaf_rn(lx, p01[e], __fmul_rn(hx, p00[e]));
        t1[e >> 1][e & 1] = __fmaf_rn(lx, p11[e], __fmul_rn(hx, p10[e]));
    }
#else
    const f32x2 hx2 = {hx, hx}, lx2 = {lx, lx};
#pragma unroll
    for (int e = 0; e < 4; e++) {
        t0[e] = __builtin_elementwise_fma(lx2, (f32x2){p01[2 * e], p01[2 * e + 1]}, hx2 * (f32x2){p00[2 * e], p00[2 * e + 1]});
        t1[e] = __builtin_elementwise_fma(lx2, (f32x2){p11[2 * e], p11[2 * e + 1]}, hx2 * (f32x2){p10[2 * e], p10[2 * e + 1]});
    }
#endif
}
__device__ __forceinline__ void bilerp8_col(float hy, float ly, const f32x2 *t0, const f32x2 *t1, float *out)
{
#ifdef RCV_X3_SCALAR_HELPERS
#pragma unroll
    for (int e = 0; e < 8; e++) out[e] = __fmaf_rn(hy, t0[e >> 1][e & 1], __fmul_rn(ly, t1[e >> 1][e & 1]));
#else
    const f32x2 hy2 = {hy, hy}, ly2 = {ly, ly};
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const f32x2 r = __builtin_elementwise_fma(hy2, t0[e], ly2 * t1[e]);
        out[2 * e] = r[0];
        out[2 * e + 1] = r[1];
    }
#endif
}
// source row/column and weights of destination index d for the 2x upsampling: ATen's area_pixel_compute_source_index
// (align_corners = False): s = max(0.5 * (d + 0.5) - 0.5, 0)
__device__ __forceinline__ void up2_src(int d, int n_in, int &i0, int &i1, float &l, float &h)
{
    const float sf = fmaxf(0.5f * (d + 0.5f) - 0.5f, 0.f);
    i0 = (int)sf;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l = sf - i0;
    h = 1.f - l;
}

// ---------------------------------------------------------------------------------------------
// k_conv_block (bf16): the second and third convolution of a bottleneck block in one kernel --
//     mid = relu(conv3x3(src) + b1)            (Cmid = 256 channels, BN folded)
//     dst = relu(conv1x1(mid) + b2 + res)      (Cout = 1024)
// -- so that `mid` never leaves the CU (reference models/fcnresnet.py:27-43).  A workgroup owns 128 output pixels:
//   phase 1  [128 x 256] = A(128 x K1) * W1^T: k_conv_dma's loader and four-phase K step with the tile roles swapped
//            (two A pieces + four B pieces per wave and step, 8 waves = 2 x 4 of 64 x 64, three 48 KB stages);
//   hand-over: bias + ReLU, rounded to bf16 exactly like the stored activation, written into LDS as the A operand of
//            phase 2 (four K slabs of [128 rows][128 B], the same chunk swizzle as the DMA tiles);
//   phase 2  eight column chunks of [128 x 128] = mid(128 x 256) * W2^T: the weights stream through a ring of three 16 KB
//            LDS slots (two DMA pieces per wave and K step, issued two steps ahead), the residual tile of a chunk is
//            DMA'd into LDS during the chunk's K steps, the result leaves through v_permlane16_swap as 16-byte vectors.
// Same MFMA sequence per output element as k_conv_dma on the two layers: bit-identical results.
// ---------------------------------------------------------------------------------------------
#ifdef RCV_BLK_STAMPS
// diagnostic build only (tools/blk_stamps.py): cycle stamps of one chunk of the staggered schedule, two waves of one workgroup
__device__ unsigned g_blk_stamps[2][80];
#define RCV_STAMP(idx)                                                                                         \
    if (nc == 3) {                                                                                             \
        unsigned long long t_;                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        stamps[idx] = (unsigned)t_;                                                                            \
    }
#else
#define RCV_STAMP(idx)
#endif
struct BlockParams {
    const void *src, *w1, *w2, *res;
    void *dst;
    const float *b1, *b2;
    long long src_gs, w1_gs, w2_gs, res_gs, dst_gs;
    int b1_gs, b2_gs;
    int C0, Hin, Win, Hout, Wout, stride, M;
    // chained variant: the next block's 1x1 reduce (4*CMID -> CMID, ReLU) on the tile just produced
    const void *w3;
    void *dst3;
    const float *b3;
    long long w3_gs, dst3_gs;
    int b3_gs;
    unsigned *range;  // sticky range-flag word (ConvParams::range)
    int q6_dst3;      // k_chain2_x3: dst3 feeds a q6 op only (engine option 11) -- write q6 rows instead of [hi | lo] rows
    int t3_res, t3_dst;  // k_chain2_x3 (engine option 14): the residual arrives / the block output leaves in the 3-byte form (t3_decode / t3_encode)
};
constexpr int BLK_BM = 128;
constexpr int BLK_RING = 3, BLK_SLOT = 128 * 128;  // expand-weight ring: 16 KB slots (128 out channels x 64 K)
// LDS bytes of k_conv_block<CMID, .>: the larger of phase 1 (three operand stages) and phase 2 (weight ring, residual /
// output chunk, and either the `mid` tile or the two reduce-weight slices), plus the fp32 biases of the 1x1 layers
// CMID = 64 is planned for TWO workgroups per CU (80 KB each, the whole LDS): its expand layer has two stream steps, so two
// ring slots, one residual buffer, and the 1x1 biases come from global memory instead of LDS.
constexpr int blk_lds(int cmid)
{
    const int p1 = 3 * (BLK_BM + cmid) * 128;
    const int ring = cmid == 64 ? 2 : BLK_RING, rbuf = cmid == 128 ? 2 : 1;
    const int p2 = ring * BLK_SLOT + rbuf * BLK_BM * 256 + (2 * cmid * 128 > BLK_BM * cmid * 2 ? 2 * cmid * 128 : BLK_BM * cmid * 2);
    return (p1 > p2 ? p1 : p2) + (cmid == 64 ? 0 : 5 * cmid * 4);
}
static_assert(blk_lds(64) * 2 <= 160 * 1024, "two k_conv_block<64> workgroups share a CU");

// s_waitcnt vmcnt(n) for an n that is a compile-time constant after unrolling
__device__ __forceinline__ void blk_vmwait(int n)
{
    switch (n) {
#define RCV_VMW(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
        RCV_VMW(0) RCV_VMW(1) RCV_VMW(2) RCV_VMW(3) RCV_VMW(4) RCV_VMW(5) RCV_VMW(6) RCV_VMW(7) RCV_VMW(8) RCV_VMW(9) RCV_VMW(10)
        RCV_VMW(11) RCV_VMW(12) RCV_VMW(13) RCV_VMW(14) RCV_VMW(15) RCV_VMW(16) RCV_VMW(17) RCV_VMW(18) RCV_VMW(19) RCV_VMW(20)
        RCV_VMW(21) RCV_VMW(22) RCV_VMW(23) RCV_VMW(24) RCV_VMW(25) RCV_VMW(26) RCV_VMW(27) RCV_VMW(28) RCV_VMW(29) RCV_VMW(30)
#undef RCV_VMW
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

__device__ __forceinline__ void blk_dma16(const void *base, int num_bytes, unsigned char *lds_wave_base, int voffset)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, num_bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds_wave_base, 16, voffset, 0, 0, 0);
}

// (... through a descriptor built once per workgroup, the wave-uniform part of the address in the instruction's scalar offset.  A plain
// function, like blk_dma16: a template kernel's body is instantiated for the host too, where the buffer builtins do not exist)
__device__ __forceinline__ void blk_dma16r(__amdgpu_buffer_rsrc_t r, unsigned char *lds_wave_base, int voffset, int soffset)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds_wave_base, 16, voffset, soffset, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t blk_rsrc(const void *base, int num_bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, num_bytes, 0x00020000);
}

template <int CMID, bool CHAIN, bool STAG = false>
__global__ __launch_bounds__(512, (CMID == 64 ? 4 : 2)) void k_conv_block(const BlockParams p)
{
    RCV_X3_MODE();
    static_assert(!STAG || (CHAIN && CMID == 256), "the staggered schedule is written for the chained 256-wide block");
    using T = H16;
    constexpr int KE = 64, EPV = 8, BM = BLK_BM, KS = 3, PAD = 1;
    constexpr int COUT = 4 * CMID;
    constexpr int OOB = (int)0x80000000;
    // phase 1: 8 waves = WM1 x WN1 over the [128 x CMID] tile
    constexpr int WN1 = CMID >= 128 ? 4 : 2, WM1 = 8 / WN1;
    constexpr int MT1 = BM / WM1 / 16, NT1 = CMID / WN1 / 16;
    constexpr int STAGE = (BM + CMID) * 128;
    constexpr int BPW1 = CMID / 64;           // B pieces (8 rows x 128 B) per wave and K step; A: 2
    constexpr int NP1 = 2 + BPW1, H1 = (NP1 + 1) / 2;
    constexpr int A2_BYTES = BM * CMID * 2;
    // phase 2: COUT / 128 chunks of 128 output channels, KS2 K steps of 64 each
    constexpr int NCH = COUT / 128, KS2 = CMID / 64, S2 = NCH * KS2;
    constexpr int W3P = CMID / 64;            // reduce-weight pieces per wave and K slice (two slices per chunk)
    constexpr int NT3 = CMID / 32;            // 16-channel tiles of a wave's CMID / 2 reduce outputs
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN1, wn = wave % WN1;
    // consecutive pixel tiles (which share their halo rows) on the same XCD: bijective remap of the round-robin dispatch
    int tile_;
    {
        const unsigned total = gridDim.x, id = blockIdx.x;
        const unsigned q = total / 8, r = total % 8, xcd = id % 8;
        tile_ = (int)((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + id / 8);
    }
    const int g = blockIdx.z, m0 = tile_ * BM;
    const int lrow = lane >> 3, lpos = lane & 7;
    const int r16 = lane & 15, kq = lane >> 4;
    const T *src = (const T *)p.src + (long long)g * p.src_gs;
    const T *w1 = (const T *)p.w1 + (long long)g * p.w1_gs;
    const T *w2 = (const T *)p.w2 + (long long)g * p.w2_gs;
    const T *res = (const T *)p.res + (long long)g * p.res_gs;
    T *dst = (T *)p.dst + (long long)g * p.dst_gs;
    const int K1 = KS * KS * p.C0;

    // biases: the 3x3 layer's in registers (epilogue layout), the 1x1 layers' in LDS; all fetched right away so that the
    // compiler's wait for these (ordinary) loads happens here and not in the middle of the DMA streams
    float bias1[NT1][4];
#pragma unroll
    for (int j = 0; j < NT1; j++)
#pragma unroll
        for (int e = 0; e < 4; e++) bias1[j][e] = p.b1[(long long)g * p.b1_gs + wn * (CMID / WN1) + j * 16 + kq * 4 + e];
    constexpr bool GBIAS = CMID == 64;  // no LDS left for the 1x1 biases: read from global memory where they are used
    float *sB2 = (float *)(smem + blk_lds(CMID) - 5 * CMID * 4);  // [COUT] expand bias, [CMID] reduce bias
    if (!GBIAS) {
        for (int q = tid; q < COUT; q += 512) sB2[q] = p.b2[(long long)g * p.b2_gs + q];
        if (CHAIN && tid < CMID) sB2[COUT + tid] = p.b3[(long long)g * p.b3_gs + tid];
    }
    const float *b2g = p.b2 + (long long)g * p.b2_gs;
#pragma unroll
    for (int j = 0; j < NT1; j++)
#pragma unroll
        for (int e = 0; e < 4; e++) asm volatile("" : "+v"(bias1[j][e]));

    // ---- phase 1 loader: 2 A rows (wave*16 + c*8 + lrow) and BPW1 B rows (wave*8*BPW1 + c*8 + lrow) per lane
    int a_off[2];
    unsigned a_ok[2];
    const int HWo = p.Hout * p.Wout;
    const int img0 = m0 / HWo, rem0 = m0 - img0 * HWo;
    const int ho0 = rem0 / p.Wout, wo0 = rem0 - ho0 * p.Wout;
    const float rW = 1.0f / (float)p.Wout, rH = 1.0f / (float)p.Hout;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int row = wave * 16 + c * 8 + lrow;
        const int chunk = (lpos ^ ((row >> 1) & 7)) * EPV;
        a_off[c] = OOB; a_ok[c] = 0u;
        if (m0 + row < p.M) {
            const int t = wo0 + row;
            const int q1 = (int)(((float)t + 0.5f) * rW);
            const int wo = t - q1 * p.Wout, h = ho0 + q1;
            const int q2 = (int)(((float)h + 0.5f) * rH);
            const int ho = h - q2 * p.Hout, img = img0 + q2;
            const int hc = ho * p.stride, wc = wo * p.stride;
            a_off[c] = (((img * p.Hin + hc) * p.Win + wc) * p.C0 + chunk) * 2;
            const int rlo = max(PAD - hc, 0), rhi = min(p.Hin - 1 - hc + PAD, KS - 1);
            const int qlo = max(PAD - wc, 0), qhi = min(p.Win - 1 - wc + PAD, KS - 1);
            const unsigned colmask = qhi >= qlo ? ((2u << qhi) - 1u) & ~((1u << qlo) - 1u) : 0u;
            for (int r = rlo; r <= rhi; r++) a_ok[c] |= colmask << (r * KS);
        }
    }
    int b_off[BPW1];
#pragma unroll
    for (int c = 0; c < BPW1; c++) {
        const int row = wave * (8 * BPW1) + c * 8 + lrow;
        b_off[c] = (row * K1 + (lpos ^ ((row >> 1) & 7)) * EPV) * 2;
    }
    const int n_img = __builtin_amdgcn_readfirstlane(p.M / HWo);
    const int a_bytes = (int)((long long)p.C0 * p.Hin * p.Win * n_img * 2);
    const int b_bytes = CMID * K1 * 2;
    int tap_r = 0, tap_s = 0, cg = 0;

// pieces [LO, HI) of the next K step (0..1: A rows, 2..: B rows) into `stage`
#define RCV_BLK_ISSUE(stage, LO, HI, ADV)                                                                      \
    {                                                                                                          \
        unsigned char *As_ = smem + (stage) * STAGE;                                                           \
        unsigned char *Bs_ = As_ + BM * 128;                                                                   \
        const int tap_ = tap_r * KS + tap_s;                                                                   \
        if ((LO) < 2) {                                                                                        \
            const long long disp_ = ((long long)((tap_r - PAD) * p.Win + (tap_s - PAD)) * p.C0 + cg) * 2;     \
            const char *abase_ = (const char *)src + disp_;                                                    \
            _Pragma("unroll") for (int c = (LO); c < ((HI) < 2 ? (HI) : 2); c++) {                             \
                const int off_ = ((a_ok[c] >> tap_) & 1u) ? a_off[c] : OOB;                                    \
                blk_dma16(abase_, a_bytes, As_ + (wave * 16 + c * 8) * 128, off_);                             \
            }                                                                                                  \
        }                                                                                                      \
        if ((HI) > 2) {                                                                                        \
            const char *bbase_ = (const char *)w1 + (long long)(tap_ * p.C0 + cg) * 2;                         \
            _Pragma("unroll") for (int c = ((LO) > 2 ? (LO) - 2 : 0); c < (HI) - 2; c++)                       \
                blk_dma16(bbase_, b_bytes, Bs_ + (wave * (8 * BPW1) + c * 8) * 128, b_off[c]);                 \
        }                                                                                                      \
        if (ADV) {                                                                                             \
            tap_s += 1;                                                                                        \
            const bool wrap_ = tap_s == KS;                                                                    \
            tap_s = wrap_ ? 0 : tap_s;                                                                         \
            tap_r += wrap_ ? 1 : 0;                                                                            \
            const bool wrap2_ = tap_r == KS;                                                                   \
            tap_r = wrap2_ ? 0 : tap_r;                                                                        \
            cg += wrap2_ ? KE : 0;                                                                             \
        }                                                                                                      \
    }
#define RCV_BLK_VMWAIT(N)                                                                                      \
    {                                                                                                          \
        if constexpr ((N) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                               \
        else if constexpr ((N) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                          \
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                                                  \
    }

    f32x4 acc[MT1][NT1];
#pragma unroll
    for (int i = 0; i < MT1; i++)
#pragma unroll
        for (int j = 0; j < NT1; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ch0 = ((0 * 4 + kq) ^ ((r16 >> 1) & 7)) << 4, ch1 = ((1 * 4 + kq) ^ ((r16 >> 1) & 7)) << 4;

// four phases per K step, the two waves of a SIMD one barrier apart (see k_conv_dma); operands swapped so that a lane
// accumulates four consecutive channels of one pixel
#define RCV_BLK_STEP(ISSUE)                                                                                    \
    {                                                                                                          \
        const unsigned char *Ab = smem + stage * STAGE + (wm * (BM / WM1) + r16) * 128;                        \
        const unsigned char *Bb = smem + stage * STAGE + BM * 128 + (wn * (CMID / WN1) + r16) * 128;           \
        const int st2 = stage + 2 >= 3 ? stage + 2 - 3 : stage + 2;                                            \
        bf16x8 af[MT1], bfr[NT1];                                                                              \
        _Pragma("unroll") for (int i = 0; i < MT1; i++) af[i] = *(const bf16x8 *)(Ab + i * 16 * 128 + ch0);    \
        _Pragma("unroll") for (int j = 0; j < NT1; j++) bfr[j] = *(const bf16x8 *)(Bb + j * 16 * 128 + ch0);   \
        if (ISSUE) RCV_BLK_ISSUE(st2, 0, H1, false);                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
        __builtin_amdgcn_s_barrier();                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT1; i++)                                                        \
            _Pragma("unroll") for (int j = 0; j < NT1; j++)                                                    \
                acc[i][j] = RCV_MFMA16(bfr[j], af[i], acc[i][j], 0, 0, 0);        \
        __builtin_amdgcn_s_setprio(0);                                                                         \
        __builtin_amdgcn_s_barrier();                                                                          \
        _Pragma("unroll") for (int i = 0; i < MT1; i++) af[i] = *(const bf16x8 *)(Ab + i * 16 * 128 + ch1);    \
        _Pragma("unroll") for (int j = 0; j < NT1; j++) bfr[j] = *(const bf16x8 *)(Bb + j * 16 * 128 + ch1);   \
        if (ISSUE) {                                                                                           \
            RCV_BLK_ISSUE(st2, H1, NP1, true);                                                                 \
            RCV_BLK_VMWAIT(NP1);                                                                               \
        } else {                                                                                               \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                   \
        }                                                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
        __builtin_amdgcn_s_barrier();                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT1; i++)                                                        \
            _Pragma("unroll") for (int j = 0; j < NT1; j++)                                                    \
                acc[i][j] = RCV_MFMA16(bfr[j], af[i], acc[i][j], 0, 0, 0);        \
        __builtin_amdgcn_s_setprio(0);                                                                         \
        __builtin_amdgcn_s_barrier();                                                                          \
        stage = stage + 1 == 3 ? 0 : stage + 1;                                                                \
    }

    const int nk = K1 / KE;  // >= 9
    const int grp = wave >> 2;
    RCV_BLK_ISSUE(0, 0, NP1, true);
    RCV_BLK_ISSUE(1, 0, NP1, true);
    RCV_BLK_VMWAIT(NP1);
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    int stage = 0;
    for (int kt = 0; kt + 2 < nk; kt++) RCV_BLK_STEP(true);
    RCV_BLK_STEP(false);
    RCV_BLK_STEP(false);
    if (grp == 0) __builtin_amdgcn_s_barrier();
#undef RCV_BLK_STEP
#undef RCV_BLK_ISSUE
#undef RCV_BLK_VMWAIT
    __syncthreads();  // every wave is done with the operand stages; no DMA is in flight

    // ---- hand-over: mid = bf16(relu(acc + b1)) into the phase-2 A operand (K slabs of [128 rows][128 B], the DMA tiles'
    // chunk swizzle).  acc[i][j][e] = channel wn*(CMID/WN1) + j*16 + kq*4 + e of pixel wm*(128/WM1) + i*16 + r16
    // LDS map of phase 2.  Plain: A2, then the W ring, then the residual tile.  Chained: A2 only lives in LDS until every wave
    // has copied its rows into registers as MFMA fragments; then W ring [0, 48K), residual / output chunk [48K, 80K), two
    // slots of CMID x 128 B for the reduce layer's weights behind it.
    unsigned char *sA2 = smem;
    unsigned char *sW = CHAIN ? smem : smem + A2_BYTES;
    // EXACT (one or two K steps per chunk, CMID <= 128): the residual tile is double buffered -- R(nc + 1) is requested when chunk
    // nc opens -- and every wait counts exactly the operations issued after the one it needs (the chunk's four output stores
    // included: they are buffer stores, issued by every wave whatever its rows).  Before (r02), R(nc) was requested when its own
    // chunk opened and waited for one or two short K steps later, and the small counts of the step waits also drained the
    // previous chunk's stores: both latencies sat in the open in every chunk of the HBM-bound layer-1 / layer-2 blocks.
    constexpr bool EXACT = !STAG && CMID == 128;
    static_assert(!EXACT || KS2 <= 2, "the exact counts below are written for one or two K steps per chunk");
    constexpr int RING = CMID == 64 ? 2 : BLK_RING;  // (two stream steps in all for CMID = 64)
    static_assert(RING == BLK_RING || S2 <= 2, "a ring of two slots serves two stream steps");
    unsigned char *sR = sW + RING * BLK_SLOT;
    unsigned char *sW3 = sR + (EXACT ? 2 : 1) * BM * 256;  // CHAIN only
#pragma unroll
    for (int i = 0; i < MT1; i++) {
        const int row = wm * (BM / WM1) + i * 16 + r16;
#pragma unroll
        for (int j = 0; j < NT1; j++) {
            uint2 pk;
            T *pv = (T *)&pk;
#pragma unroll
            for (int e = 0; e < 4; e++) pv[e] = (T)fmaxf(acc[i][j][e] + bias1[j][e], 0.f);
            const int chn = wn * (CMID / WN1) + j * 16 + kq * 4;  // first of the four channels
            const int slab = chn >> 6, chunk = (chn & 63) >> 3;
            *(uint2 *)(sA2 + slab * (BM * 128) + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4) + (kq & 1) * 8) = pk;
        }
    }
    // phase-2 wave tile: 32 pixels x 64 channels (8 waves = 4 x 2), so that the two 16-byte stores a lane makes per pixel
    // group complete a 128-byte line (64 channels) of the output row
    const int wm2 = wave >> 1, wn2 = wave & 1;
    bf16x8 a2f[2][CMID / 32];  // CHAIN: this wave's 32 rows of `mid` as fragments, [pixel group][K chunk of 32]
    if (CHAIN) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int k8 = 0; k8 < CMID / 32; k8++)
                a2f[i][k8] = *(const bf16x8 *)(sA2 + (k8 >> 1) * (BM * 128) + (wm2 * 32 + i * 16 + r16) * 128 + ((k8 & 1) ? ch1 : ch0));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();  // `mid` is complete (plain) / its LDS is free for the weight ring (chained)

    // ---- phase 2 loaders.  Weights of stream step s = (chunk nc = s / KS2, K slice ks = s % KS2): 128 rows (out channels) x
    // 128 B -> 2 pieces per wave; residual of chunk nc: 128 pixels x 256 B -> 4 pieces per wave (4 rows each); chained reduce
    // layer: two K slices of 64 per chunk, CMID rows (out channels) x 128 B each -> W3P pieces per wave and slice
    int w_off[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int row = wave * 16 + c * 8 + lrow;
        w_off[c] = (row * CMID + (lpos ^ ((row >> 1) & 7)) * EPV) * 2;
    }
    int r_off[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int row = wave * 16 + c * 4 + (lane >> 4), slot = lane & 15;  // 4 rows x 16 chunks per piece
        r_off[c] = m0 + row < p.M ? (int)(((long long)(m0 + row) * COUT + (slot ^ (row & 15)) * EPV) * 2) : OOB;
    }
    int w3_off[W3P];
#pragma unroll
    for (int c = 0; c < W3P; c++) {
        const int row = wave * (8 * W3P) + c * 8 + lrow;
        w3_off[c] = (row * COUT + (lpos ^ ((row >> 1) & 7)) * EPV) * 2;
    }
    const T *w3 = CHAIN ? (const T *)p.w3 + (long long)g * p.w3_gs : nullptr;
    const int w_bytes = 128 * CMID * 2, w3_bytes = CMID * COUT * 2;
    const long long res_bytes_ll = (long long)p.M * COUT * 2;
    const int res_bytes = res_bytes_ll > 0x7fffffffLL ? 0x7fffffff : (int)res_bytes_ll;
#define RCV_BLK_W(s_)                                                                                          \
    {                                                                                                          \
        const int nc_ = (s_) / KS2, ks_ = (s_) % KS2;                                                          \
        const char *wb_ = (const char *)w2 + ((long long)nc_ * 128 * CMID + ks_ * KE) * 2;                     \
        unsigned char *slot_ = sW + ((s_) % RING) * BLK_SLOT;                                                  \
        _Pragma("unroll") for (int c = 0; c < 2; c++) blk_dma16(wb_, w_bytes, slot_ + (wave * 16 + c * 8) * 128, w_off[c]); \
    }
#define RCV_BLK_R(nc_)                                                                                         \
    {                                                                                                          \
        const char *rb_ = (const char *)res + (long long)(nc_) * 128 * 2;                                      \
        unsigned char *rbuf_ = sR + (EXACT ? ((nc_) & 1) * (BM * 256) : 0);                                    \
        _Pragma("unroll") for (int c = 0; c < 4; c++) blk_dma16(rb_, res_bytes, rbuf_ + (wave * 16 + c * 4) * 256, r_off[c]); \
    }
#define RCV_BLK_W3(nc_)                                                                                        \
    {                                                                                                          \
        _Pragma("unroll") for (int h = 0; h < 2; h++) {                                                        \
            const char *wb_ = (const char *)w3 + ((long long)(nc_) * 128 + h * KE) * 2;                        \
            _Pragma("unroll") for (int c = 0; c < W3P; c++)                                                    \
                blk_dma16(wb_, w3_bytes, sW3 + h * (CMID * 128) + (wave * (8 * W3P) + c * 8) * 128, w3_off[c]); \
        }                                                                                                      \
    }
#define RCV_BLK_W3H(nc_, h_)                                                                                   \
    {                                                                                                          \
        const char *wb_ = (const char *)w3 + ((long long)(nc_) * 128 + (h_) * KE) * 2;                         \
        _Pragma("unroll") for (int c = 0; c < W3P; c++)                                                        \
            blk_dma16(wb_, w3_bytes, sW3 + (h_) * (CMID * 128) + (wave * (8 * W3P) + c * 8) * 128, w3_off[c]); \
    }
#define RCV_BLK_W3Q(nc_, h_, q_)                                                                               \
    {                                                                                                          \
        const char *wb_ = (const char *)w3 + ((long long)(nc_) * 128 + (h_) * KE) * 2;                         \
        _Pragma("unroll") for (int c = (q_) * (W3P / 2); c < ((q_) + 1) * (W3P / 2); c++)                      \
            blk_dma16(wb_, w3_bytes, sW3 + (h_) * (CMID * 128) + (wave * (8 * W3P) + c * 8) * 128, w3_off[c]); \
    }
    // order of issue per wave: at the first step of a chunk R(nc) [4 pieces] and, chained, both reduce-weight slices
    // [2 * W3P]; then W(s + 2) [2] at every step s < S2 - 2.  At the top of step s the pieces issued after W(s) are those
    // of step s - 1: W(s + 1) and, when step s - 1 opened a chunk, that chunk's R (and W3).
    constexpr int OPEN_PIECES = 4 + (CHAIN ? 2 * W3P : 0);
    constexpr bool STAGGER = STAG;
    f32x4 acc3[2][NT3];  // CHAIN: [pixel group][16-channel tile of this wave's CMID / 2 reduce outputs]
    if (CHAIN) {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < NT3; j++) acc3[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (STAGGER) {
        // Staggered schedule (CMID = 256, chained).  Every chunk is a chain of 18 segments, each closed by one s_barrier:
        //     Le0 Me0 Le1 Me1 Le2 Me2 Le3 Me3 | (-) P | P (-) | Lr0 Mr0 Lr1 Mr1 Lr2 Mr2 Lr3 Mr3      (-): an empty segment
        // L = DMA issue + fragment reads (whole K step into registers), M = 16 MFMAs, P = the chunk's epilogue.  Waves 4..7
        // run one barrier behind waves 0..3 (one extra barrier at the start, waves 0..3 one at the end), so that the MFMAs
        // of one wave of a SIMD cover the LDS / DMA work of the other; around P the groups are level (the epilogue is
        // latency-bound, two of them side by side cost little more than one).  With the lag a buffer may be read two
        // barriers after the owner's wait for it and overwritten two barriers after its last read (sR holds only the
        // rounded output tile, whose rows are private to a pair of waves of one group).  The residual does not pass through
        // LDS here: its HBM latency would hold up every DMA piece issued after it (in-order return), so each lane loads its
        // own 4 x 16 B for the next chunk into registers right after the epilogue, where the DMA queue is quiet.
        // Issue order per wave and chunk (s0 = first stream step), [pieces]; no segment carries more than four:
        //     Le0: W(s0+2) [2], W3(nc, h=1) first half [2]     Le1: W(s0+3) [2], W3(nc, 1) second half [2]
        //     Le2: W(s0+4) [2]     Le3: W(s0+5) [2]             Lr0: residual of chunk nc+1 [4 loads]
        //     Lr2: W3(nc+1, h=0) first half [2]                 Lr3: W3(nc+1, 0) second half [2]
        // Waits: Le1 vmcnt(6) -> W(s0+2);  Le2 vmcnt(4) -> W(s0+3);  Lr0 (before the loads) vmcnt(4) -> W3(nc, 1);
        //        Lr3 vmcnt(8) -> W(s0+4), W(s0+5).  Stores are never counted as younger pieces (a wave may skip them).
        const int grp2 = wave >> 2;
        const int e_base = wn2 * 64 + (kq & 1) * 16 + (kq >> 1) * 8;  // first of this lane's 8 channels (pr adds 32)
        const T *rrow[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int m = m0 + wm2 * 32 + i * 16 + r16;
            rrow[i] = res + (long long)(m < p.M ? m : p.M - 1) * COUT + e_base;
        }
        u32x4 rg[2][2];  // residual of the coming chunk: [pixel group][pr]
        RCV_BLK_W(0);
        RCV_BLK_W(1);
        RCV_BLK_W3H(0, 0);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int pr = 0; pr < 2; pr++) rg[i][pr] = *(const u32x4 *)(rrow[i] + pr * 32);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (grp2 == 1) __builtin_amdgcn_s_barrier();
        const unsigned sB2a = (unsigned)(size_t)sB2 + e_base * 4, sRa = (unsigned)(size_t)sR;
#ifdef RCV_BLK_STAMPS
        unsigned stamps[80];
#pragma unroll
        for (int k = 0; k < 80; k++) stamps[k] = 0u;
#endif
#pragma unroll
        for (int nc = 0; nc < NCH; nc++) {
            constexpr int KS2c = 4;
            const int s0 = nc * KS2c;
            const bool more = nc + 1 < NCH;
            f32x4 a2[2][4];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) a2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS2c; ks++) {
                // ---- Le(ks)
                RCV_STAMP(4 * ks);
                if (ks == 0) { RCV_BLK_W(s0 + 2); RCV_STAMP(44); RCV_BLK_W3Q(nc, 1, 0); RCV_STAMP(45); }
                if (ks == 1) { RCV_BLK_W(s0 + 3); RCV_STAMP(46); RCV_BLK_W3Q(nc, 1, 1); RCV_STAMP(47); }
                if (ks == 2 && more) RCV_BLK_W(s0 + 4);
                if (ks == 3 && more) RCV_BLK_W(s0 + 5);
                const unsigned char *Bb = sW + ((s0 + ks) % BLK_RING) * BLK_SLOT + (wn2 * 64 + r16) * 128;
                bf16x8 bfr[2][4];
#pragma unroll
                for (int kk = 0; kk < 2; kk++)
#pragma unroll
                    for (int j = 0; j < 4; j++) bfr[kk][j] = *(const bf16x8 *)(Bb + j * 16 * 128 + (kk ? ch1 : ch0));
                RCV_STAMP(40 + ks);
                if (ks == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                if (ks == 2) {
                    if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                }
                if (ks == 3 && !more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // W3(nc, 1), before the last stores
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                RCV_STAMP(4 * ks + 1);
                __builtin_amdgcn_s_barrier();
                // ---- Me(ks)
                RCV_STAMP(4 * ks + 2);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kk = 0; kk < 2; kk++)
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            a2[i][j] = RCV_MFMA16(bfr[kk][j], a2f[i][ks * 2 + kk], a2[i][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                RCV_STAMP(4 * ks + 3);
                __builtin_amdgcn_s_barrier();
            }
            if (grp2 == 0) __builtin_amdgcn_s_barrier();
            RCV_STAMP(16);
            // ---- P: bias + residual + ReLU; 16-byte stores; the rounded tile into sR as the reduce layer's A operand.
            // LDS traffic through inline asm: compiler-visible LDS accesses here make hipcc drain vmcnt for the DMAs in flight.
            {
                f32x4 bq[2][2];  // bias: [pr][first / second four channels]
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:128\n\tds_read_b128 %3, %4 offset:144"
                             : "=&v"(bq[0][0]), "=&v"(bq[0][1]), "=&v"(bq[1][0]), "=&v"(bq[1][1]) : "v"(sB2a + nc * 128 * 4) : "memory");
                float v[2][2][8];
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int pr = 0; pr < 2; pr++)
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(a2[i][2 * pr][e]), __float_as_uint(a2[i][2 * pr + 1][e]), false, false);
                            v[i][pr][e] = __uint_as_float(sw[0]);
                            v[i][pr][4 + e] = __uint_as_float(sw[1]);
                        }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[1][0]), "+v"(bq[1][1])::"memory");
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int row = wm2 * 32 + i * 16 + r16, m = m0 + row;
#pragma unroll
                    for (int pr = 0; pr < 2; pr++) {
                        const int e_ch = e_base + pr * 32;
                        const float bb[8] = {bq[pr][0][0], bq[pr][0][1], bq[pr][0][2], bq[pr][0][3], bq[pr][1][0], bq[pr][1][1], bq[pr][1][2], bq[pr][1][3]};
                        const u32x4 rq = rg[i][pr];
                        const T *rp = (const T *)&rq;
                        u32x4 outv;
                        T *op = (T *)&outv;
#pragma unroll
                        for (int e = 0; e < 8; e++) op[e] = (T)fmaxf((v[i][pr][e] + bb[e]) + (float)rp[e], 0.f);
                        if (m < p.M) *(u32x4 *)(dst + (long long)m * COUT + nc * 128 + e_ch) = outv;
                        const unsigned yaddr = sRa + row * 256 + (((e_ch >> 3) ^ (row & 15)) << 4);
                        asm volatile("ds_write_b128 %0, %1" ::"v"(yaddr), "v"(outv) : "memory");
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            RCV_STAMP(17);
            __builtin_amdgcn_s_barrier();
            if (grp2 == 1) __builtin_amdgcn_s_barrier();
            // ---- reduce layer on the chunk: acc3 += Y[32 px x 128] * W3[128 out x 128]^T, four K steps of 32
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int h = t >> 1, kk = t & 1;
                // ---- Lr(t)
                RCV_STAMP(18 + 4 * t);
                if (t == 0 && more) {
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // W3(nc, 1); the youngest four are W(s0+4), W(s0+5) or stores
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int pr = 0; pr < 2; pr++) rg[i][pr] = *(const u32x4 *)(rrow[i] + (nc + 1) * 128 + pr * 32);
                }
                if (t == 2 && more) RCV_BLK_W3Q(nc + 1, 0, 0);
                if (t == 3 && more) RCV_BLK_W3Q(nc + 1, 0, 1);
                bf16x8 yf[2], bf3[NT3];
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int row = wm2 * 32 + i * 16 + r16;
                    yf[i] = *(const bf16x8 *)(sR + row * 256 + (((h * 8 + kk * 4 + kq) ^ (row & 15)) << 4));
                }
                const unsigned char *B3 = sW3 + h * (CMID * 128) + (wn2 * (CMID / 2) + r16) * 128 + (kk ? ch1 : ch0);
#pragma unroll
                for (int j = 0; j < NT3; j++) bf3[j] = *(const bf16x8 *)(B3 + j * 16 * 128);
                if (t == 3 && more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                RCV_STAMP(18 + 4 * t + 1);
                __builtin_amdgcn_s_barrier();
                // ---- Mr(t)
                RCV_STAMP(18 + 4 * t + 2);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < NT3; j++) acc3[i][j] = RCV_MFMA16(bf3[j], yf[i], acc3[i][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                RCV_STAMP(18 + 4 * t + 3);
                __builtin_amdgcn_s_barrier();
            }
            RCV_STAMP(34);
        }
        if (grp2 == 0) __builtin_amdgcn_s_barrier();
#ifdef RCV_BLK_STAMPS
        if (lane == 0 && (wave & 3) == 0 && blockIdx.x == 100 && g == 0)
            for (int k = 0; k < 80; k++) g_blk_stamps[wave >> 2][k] = stamps[k];
#endif
    } else {
    // operations a wave issues, in order (EXACT): W(0) W(1) R(0) | per step, after its barrier: [chunk opens: W3(nc), R(nc + 1)]
    // W(s + 2) | after a chunk's last step: its ST output stores.  Every count below is "issued after X, before this wait".
    constexpr int P3 = CHAIN ? 2 * W3P : 0, ST = 4;
    auto open_n = [&](int c) { return EXACT ? P3 + (c + 1 < NCH ? 4 : 0) : 4 + P3; };
    auto step_n = [&](int st) { return (st % KS2 == 0 ? open_n(st / KS2) : 0) + (st + 2 < S2 ? 2 : 0); };
    auto closes = [&](int st) { return st % KS2 == KS2 - 1; };
    const __amdgpu_buffer_rsrc_t dst_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, res_bytes, 0x00020000);
    RCV_BLK_W(0);
    if (S2 > 1) RCV_BLK_W(1);
    if (EXACT) RCV_BLK_R(0);
#pragma unroll
    for (int nc = 0; nc < NCH; nc++) {
        f32x4 a2[2][4];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) a2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // GBIAS: this lane's 2 x 8 expand biases of the chunk, requested before the chunk's pieces (older than everything the
        // epilogue waits for)
        float4 gb[2][2];
        if (GBIAS) {
#pragma unroll
            for (int pr = 0; pr < 2; pr++) {
                const int e_ch = wn2 * 64 + (2 * pr + (kq & 1)) * 16 + (kq >> 1) * 8;
                gb[pr][0] = *(const float4 *)(b2g + nc * 128 + e_ch);
                gb[pr][1] = *(const float4 *)(b2g + nc * 128 + e_ch + 4);
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS2; ks++) {
            const int s_ = nc * KS2 + ks;
            if (EXACT) {  // W(s_) has landed
                if (s_ == 0) blk_vmwait((S2 > 1 ? 2 : 0) + 4);
                else if (s_ == 1) blk_vmwait(4 + step_n(0) + (closes(0) ? ST : 0));
                else blk_vmwait((closes(s_ - 2) ? ST : 0) + step_n(s_ - 1) + (closes(s_ - 1) ? ST : 0));
            } else if (s_ + 1 >= S2) {  // younger pieces: W(s_ + 1) and the chunk-opening pieces of step s_ - 1; the last step simply drains
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (s_ >= 1 && (s_ - 1) % KS2 == 0) {
                blk_vmwait(OPEN_PIECES + 2);
            } else {
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // step s_ has landed for everybody; slot (s_ + 2) % 3 and (ks == 0) the sR buffer / sW3 are free
            if (ks == 0) {
                if (EXACT) {  // the reduce weights first: the chunk's epilogue waits for them and leaves R(nc + 1) in flight
                    if (CHAIN) RCV_BLK_W3(nc);
                    if (nc + 1 < NCH) RCV_BLK_R(nc + 1);
                } else {
                    RCV_BLK_R(nc);
                    if (CHAIN) RCV_BLK_W3(nc);
                }
            }
            if (s_ + 2 < S2) RCV_BLK_W(s_ + 2);
            const unsigned char *Ab = sA2 + ks * (BM * 128) + (wm2 * 32 + r16) * 128;
            const unsigned char *Bb = sW + (s_ % RING) * BLK_SLOT + (wn2 * 64 + r16) * 128;
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                bf16x8 af[2], bfr[4];
                const int ch = kk ? ch1 : ch0;
#pragma unroll
                for (int i = 0; i < 2; i++) af[i] = CHAIN ? a2f[i][ks * 2 + kk] : *(const bf16x8 *)(Ab + i * 16 * 128 + ch);
#pragma unroll
                for (int j = 0; j < 4; j++) bfr[j] = *(const bf16x8 *)(Bb + j * 16 * 128 + ch);
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) a2[i][j] = RCV_MFMA16(bfr[j], af[i], a2[i][j], 0, 0, 0);
            }
        }
        // epilogue of the chunk.  With four K steps per chunk the step waits (vmcnt 2) already cover R(nc) and the reduce
        // weights; with one or two steps they are waited for here: everything issued before the chunk opened (R(nc) among it)
        // and W3(nc), the first piece of the opening, have landed when only the later pieces of this chunk are outstanding.
        if (KS2 <= 2) {
            int later = EXACT && nc + 1 < NCH ? 4 : 0;
#pragma unroll
            for (int k = 0; k < KS2; k++) later += nc * KS2 + k + 2 < S2 ? 2 : 0;
            blk_vmwait(later);
            __builtin_amdgcn_s_barrier();
        }
        unsigned char *sRc = sR + (EXACT ? (nc & 1) * (BM * 256) : 0);
        // Chained: the rounded outputs go back into the residual's LDS slot, where they are the A operand (K = this chunk's
        // 128 channels) of the reduce layer.
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int row = wm2 * 32 + i * 16 + r16, m = m0 + row;
#pragma unroll
            for (int pr = 0; pr < 2; pr++) {
                const int e_ch = wn2 * 64 + (2 * pr + (kq & 1)) * 16 + (kq >> 1) * 8;  // first of this lane's 8 channels in the chunk
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(a2[i][2 * pr][e]), __float_as_uint(a2[i][2 * pr + 1][e]), false, false);
                    v[e] = __uint_as_float(sw[0]);
                    v[4 + e] = __uint_as_float(sw[1]);
                }
                // (LDS traffic through inline asm: compiler-visible LDS accesses here make hipcc drain vmcnt(0), i.e. wait for
                // the next chunk's weight pieces that are already in flight)
                const unsigned baddr = (unsigned)(size_t)(sB2 + nc * 128 + e_ch);
                const unsigned raddr = (unsigned)(size_t)(sRc + row * 256 + (((e_ch >> 3) ^ (row & 15)) << 4));
                f32x4 bA, bB;
                u32x4 rq;
                if (GBIAS) {
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(rq) : "v"(raddr) : "memory");
                    bA = f32x4{gb[pr][0].x, gb[pr][0].y, gb[pr][0].z, gb[pr][0].w};
                    bB = f32x4{gb[pr][1].x, gb[pr][1].y, gb[pr][1].z, gb[pr][1].w};
                } else {
                    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %4\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(bA), "=&v"(bB), "=&v"(rq) : "v"(baddr), "v"(raddr) : "memory");
                }
                const float bb[8] = {bA[0], bA[1], bA[2], bA[3], bB[0], bB[1], bB[2], bB[3]};
                const T *rp = (const T *)&rq;
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = fmaxf((v[e] + bb[e]) + (float)rp[e], 0.f);
                u32x4 outv;
                T *op = (T *)&outv;
#pragma unroll
                for (int e = 0; e < 8; e++) op[e] = (T)v[e];
                // a buffer store: issued by every wave (rows beyond M get an out-of-range offset and are dropped), so that the
                // counted waits above can rely on four stores per wave and chunk
                __builtin_amdgcn_raw_buffer_store_b128(outv, dst_rsrc, m < p.M ? (int)(((long long)m * COUT + nc * 128 + e_ch) * 2) : OOB, 0, 0);
                if (CHAIN) asm volatile("ds_write_b128 %0, %1" ::"v"(raddr), "v"(outv) : "memory");
            }
        }
        if (CHAIN) {
            // reduce layer, K slab of this chunk: acc3 += Y[32 px x 128] * W3[CMID/2 out x 128]^T.  Both weight slices were
            // issued at the chunk's first step and are older than everything the later steps waited for.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // every wave's part of Y is in LDS
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int kk = 0; kk < 2; kk++) {
                    bf16x8 yf[2];
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        const int row = wm2 * 32 + i * 16 + r16;
                        yf[i] = *(const bf16x8 *)(sRc + row * 256 + (((h * 8 + kk * 4 + kq) ^ (row & 15)) << 4));
                    }
                    const unsigned char *Bb = sW3 + h * (CMID * 128) + (wn2 * (CMID / 2) + r16) * 128 + (kk ? ch1 : ch0);
                    constexpr int JG = NT3 < 4 ? NT3 : 4;
#pragma unroll
                    for (int jh = 0; jh < NT3 / JG; jh++) {
                        bf16x8 bfr[JG];
#pragma unroll
                        for (int j = 0; j < JG; j++) bfr[j] = *(const bf16x8 *)(Bb + (jh * JG + j) * 16 * 128);
#pragma unroll
                        for (int i = 0; i < 2; i++)
#pragma unroll
                            for (int j = 0; j < JG; j++)
                                acc3[i][jh * JG + j] = RCV_MFMA16(bfr[j], yf[i], acc3[i][jh * JG + j], 0, 0, 0);
                    }
                }
        }
    }
    }  // !STAGGER
    if (CHAIN) {
        // the next block's first activation: relu(acc3 + b3) rounded to bf16, 16-byte vectors of 8 channels
        T *dst3 = (T *)p.dst3 + (long long)g * p.dst3_gs;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int row = wm2 * 32 + i * 16 + r16, m = m0 + row;
#pragma unroll
            for (int pr = 0; pr < NT3 / 2; pr++) {
                const int e_ch = wn2 * (CMID / 2) + (2 * pr + (kq & 1)) * 16 + (kq >> 1) * 8;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc3[i][2 * pr][e]), __float_as_uint(acc3[i][2 * pr + 1][e]), false, false);
                    v[e] = __uint_as_float(sw[0]);
                    v[4 + e] = __uint_as_float(sw[1]);
                }
                const float *b3p = GBIAS ? p.b3 + (long long)g * p.b3_gs + e_ch : sB2 + COUT + e_ch;
                const float4 bA = *(const float4 *)b3p, bB = *(const float4 *)(b3p + 4);
                const float bb[8] = {bA.x, bA.y, bA.z, bA.w, bB.x, bB.y, bB.z, bB.w};
                uint4 outv;
                T *op = (T *)&outv;
#pragma unroll
                for (int e = 0; e < 8; e++) op[e] = (T)fmaxf(v[e] + bb[e], 0.f);
                if (m < p.M) *(uint4 *)(dst3 + (long long)m * CMID + e_ch) = outv;
            }
        }
    }
#undef RCV_BLK_W3
#undef RCV_BLK_W3H
#undef RCV_BLK_W3Q
#undef RCV_BLK_W
#undef RCV_BLK_R
}

// ---------------------------------------------------------------------------------------------
// stem: conv1 7x7 stride 2 pad 3 (3 -> 64, no bias) + BN (folded) + ReLU; NCHW fp32 image -> NHWC T
// one workgroup = 16 x 16 output pixels x 64 channels of one (group, image)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_stem(const float *x, const T *w, const float *bias, T *dst, int B, int H, int W,
                                              int Ho, int Wo, long long w_gs, int bias_gs, long long dst_gs, unsigned *range)
{
    RCV_X3_MODE();
    float rmax = 0.f;  // range guard (half-precision engines)
    constexpr int TP = 16, IN = TP * 2 + 5;  // 37 x 37 input patch per channel
    __shared__ float s_in[3][IN][IN + 1];
    __shared__ __attribute__((aligned(16))) float s_w[147][64];  // [tap*3 + c][cout]
    const int tid = threadIdx.x;
    const int tiles_x = (Wo + TP - 1) / TP;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int img = blockIdx.y, g = blockIdx.z;
    const int ho0 = ty * TP, wo0 = tx * TP;
    const int hi0 = ho0 * 2 - 3, wi0 = wo0 * 2 - 3;
    for (int i = tid; i < 3 * IN * IN; i += 256) {
        const int c = i / (IN * IN), r = (i / IN) % IN, s = i % IN;
        const int hi = hi0 + r, wi = wi0 + s;
        float v = 0.f;
        if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) v = x[((long long)(img * 3 + c) * H + hi) * W + wi];
        s_in[c][r][s] = v;
    }
    const T *wg = w + (long long)g * w_gs;  // [64][7][7][3]
    for (int i = tid; i < 147 * 64; i += 256) {
        const int co = i / 147, t = i - co * 147;
        if constexpr (is_x3<T>()) s_w[t][co] = to_f32(wg[i]) + to_f32(wg[147 * 64 + i]);  // [all hi][all lo], scaled (see the epilogue)
        else s_w[t][co] = to_f32(wg[i]);
    }
    __syncthreads();
    const int py = tid >> 4, px = tid & 15;
    const int ho = ho0 + py, wo = wo0 + px;
    const float *bg = bias + (long long)g * bias_gs;
    T *out = dst + (long long)g * dst_gs + ((long long)(img * Ho + ho) * Wo + wo) * (is_x3<T>() ? 128 : 64);
    for (int cb = 0; cb < 64; cb += 16) {
        float acc[16];
#pragma unroll
        for (int q = 0; q < 16; q++) acc[q] = 0.f;
        for (int r = 0; r < 7; r++)
            for (int s = 0; s < 7; s++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float v = s_in[c][py * 2 + r][px * 2 + s];
                    const float4 *wv = (const float4 *)&s_w[(r * 7 + s) * 3 + c][cb];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float4 ww = wv[q];
                        acc[q * 4 + 0] = fmaf(v, ww.x, acc[q * 4 + 0]);
                        acc[q * 4 + 1] = fmaf(v, ww.y, acc[q * 4 + 1]);
                        acc[q * 4 + 2] = fmaf(v, ww.z, acc[q * 4 + 2]);
                        acc[q * 4 + 3] = fmaf(v, ww.w, acc[q * 4 + 3]);
                    }
                }
        if (ho < Ho && wo < Wo) {
            if constexpr (is_x3<T>()) {
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const float v = fmaxf(__fmaf_rn(acc[q], bg[64 + cb + q], bg[cb + q]), 0.f);
                    rmax = fmaxf(rmax, v);
                    x3_split(v, out[x3_off(cb + q)], out[x3_off(cb + q) + 32]);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const float v = fmaxf(acc[q] + bg[cb + q], 0.f);
                    if constexpr (sizeof(T) == 2) rmax = fmaxf(rmax, v);
                    out[cb + q] = from_f32<T>(v);
                }
            }
        }
    }
    if constexpr (sizeof(T) == 2) rng_flush(range, rmax);
}

// maxpool 3x3 stride 2 pad 1, NHWC; one thread = EPV channels of one output pixel
template <typename T>
__global__ __launch_bounds__(256) void k_maxpool(const T *src, T *dst, int C, int Hin, int Win, int Hout, int Wout,
                                                 long long total_vec, long long src_gs, long long dst_gs)
{
    RCV_X3_MODE();
    constexpr int EPV = Traits<T>::EPV;
    const int g = blockIdx.y;
    const int cv = C / EPV;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total_vec; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % cv);
        const long long pix = i / cv;
        const int wo = (int)(pix % Wout), ho = (int)((pix / Wout) % Hout), img = (int)(pix / ((long long)Wout * Hout));
        float best[EPV];
#pragma unroll
        for (int e = 0; e < EPV; e++) best[e] = -3.0e38f;
        for (int r = 0; r < 3; r++) {
            const int hi = ho * 2 - 1 + r;
            if ((unsigned)hi >= (unsigned)Hin) continue;
            for (int s = 0; s < 3; s++) {
                const int wi = wo * 2 - 1 + s;
                if ((unsigned)wi >= (unsigned)Win) continue;
                const T *sp = src + (long long)g * src_gs + nhwc_off<T>(((long long)img * Hin + hi) * Win + wi, C, c * EPV);
                const uint4 raw = *(const uint4 *)sp;
                if constexpr (is_x3<T>()) {
                    float f[8];
                    x3_join8(raw, *(const uint4 *)(sp + 32), f);
#pragma unroll
                    for (int e = 0; e < EPV; e++) best[e] = fmaxf(best[e], f[e]);
                } else {
                    const T *pv = (const T *)&raw;
#pragma unroll
                    for (int e = 0; e < EPV; e++) best[e] = fmaxf(best[e], to_f32(pv[e]));
                }
            }
        }
        T *dp = dst + (long long)g * dst_gs + nhwc_off<T>(pix, C, c * EPV);
        if constexpr (is_x3<T>()) {
            uint4 oh, ol;
            x3_split8(best, oh, ol);
            *(uint4 *)dp = oh;
            *(uint4 *)(dp + 32) = ol;
        } else {
            uint4 o;
            T *po = (T *)&o;
#pragma unroll
            for (int e = 0; e < EPV; e++) po[e] = from_f32<T>(best[e]);
            *(uint4 *)dp = o;
        }
    }
}

// bilinear x2, align_corners=False (nn.Upsample, reference models/fcnresnet.py:93), NHWC
template <typename T>
__global__ __launch_bounds__(256) void k_upsample2(const T *src, T *dst, int C, int Hin, int Win, long long total_vec,
                                                   long long src_gs, long long dst_gs)
{
    RCV_X3_MODE();
    // One thread = one source cell (a, b) (a in [-1, Hin-1], b in [-1, Win-1]) and one 16-byte channel chunk: the up to
    // four destination pixels rows {2a+1, 2a+2} x columns {2b+1, 2b+2} all interpolate between source rows (a, a+1) and
    // columns (b, b+1) (clamped at the borders, where the weight of the clamped neighbour is zero or both neighbours are
    // the same pixel), so the four source vectors are loaded once for four outputs.  Weights per destination index:
    // up2_src (ATen's area_pixel_compute_source_index, align_corners = False).
    constexpr int EPV = Traits<T>::EPV;
    const int g = blockIdx.y;
    const int cv = C / EPV, Hout = Hin * 2, Wout = Win * 2;
    const int cells_w = Win + 1, cells_h = Hin + 1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total_vec; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % cv);
        const long long cell = i / cv;
        const int b = (int)(cell % cells_w) - 1, a = (int)((cell / cells_w) % cells_h) - 1;
        const int img = (int)(cell / ((long long)cells_w * cells_h));
        const int r0 = max(a, 0), r1 = min(a + 1, Hin - 1), c0 = max(b, 0), c1 = min(b + 1, Win - 1);
        constexpr int PS = is_x3<T>() ? 2 : 1;  // elements of T per channel and pixel
        const T *base = src + (long long)g * src_gs + (long long)img * Hin * Win * C * PS + nhwc_off<T>(0, C, c * EPV);
        const T *a00 = base + ((long long)r0 * Win + c0) * C * PS, *a01 = base + ((long long)r0 * Win + c1) * C * PS;
        const T *a10 = base + ((long long)r1 * Win + c0) * C * PS, *a11 = base + ((long long)r1 * Win + c1) * C * PS;
        const uint4 q00 = *(const uint4 *)a00, q01 = *(const uint4 *)a01, q10 = *(const uint4 *)a10, q11 = *(const uint4 *)a11;
        float f00[EPV], f01[EPV], f10[EPV], f11[EPV];
        if constexpr (is_x3<T>()) {
            x3_join8(q00, *(const uint4 *)(a00 + 32), f00);
            x3_join8(q01, *(const uint4 *)(a01 + 32), f01);
            x3_join8(q10, *(const uint4 *)(a10 + 32), f10);
            x3_join8(q11, *(const uint4 *)(a11 + 32), f11);
        } else {
            const T *p00 = (const T *)&q00, *p01 = (const T *)&q01, *p10 = (const T *)&q10, *p11 = (const T *)&q11;
#pragma unroll
            for (int e = 0; e < EPV; e++) { f00[e] = to_f32(p00[e]); f01[e] = to_f32(p01[e]); f10[e] = to_f32(p10[e]); f11[e] = to_f32(p11[e]); }
        }
        T *out = dst + (long long)g * dst_gs + (long long)img * Hout * Wout * C * PS + nhwc_off<T>(0, C, c * EPV);
#pragma unroll
        for (int dy = 1; dy <= 2; dy++) {
            const int y = 2 * a + dy;
            if ((unsigned)y >= (unsigned)Hout) continue;
            int y0, y1;
            float ly, hy;
            up2_src(y, Hin, y0, y1, ly, hy);
#pragma unroll
            for (int dx = 1; dx <= 2; dx++) {
                const int x = 2 * b + dx;
                if ((unsigned)x >= (unsigned)Wout) continue;
                int x0, x1;
                float lx, hx;
                up2_src(x, Win, x0, x1, lx, hx);
                T *op = out + ((long long)y * Wout + x) * C * PS;
                if constexpr (is_x3<T>()) {
                    float bl[8];
                    f32x2 t0[4], t1[4];
                    bilerp8_rows(hx, lx, f00, f01, f10, f11, t0, t1);
                    bilerp8_col(hy, ly, t0, t1, bl);
                    uint4 oh, ol;
                    x3_split8(bl, oh, ol);
                    *(uint4 *)op = oh;
                    *(uint4 *)(op + 32) = ol;
                } else {
                    uint4 o;
                    T *po = (T *)&o;
#pragma unroll
                    for (int e = 0; e < EPV; e++) po[e] = from_f32<T>(bilerp(hy, hx, ly, lx, f00[e], f01[e], f10[e], f11[e]));
                    *(uint4 *)op = o;
                }
            }
        }
    }
}

// conv8: 1x1, Cin -> 2, bias; writes the two planes seg / radial as fp32 [G,B,1,H,W]
// (reference models/fcnresnet.py:186-191: out[:, :1], out[:, 1:])
template <typename T>
__global__ __launch_bounds__(256) void k_head(const T *src, const T *w, const float *bias, float *seg, float *radial, int C,
                                              long long npix, long long src_gs, long long w_gs, int bias_gs, int rad_f16)
{
    RCV_X3_MODE();
    constexpr int EPV = Traits<T>::EPV;
    const int g = blockIdx.y;
    const T *wg = w + (long long)g * w_gs;
    const float b0 = bias[(long long)g * bias_gs], b1 = bias[(long long)g * bias_gs + 1];
    // split mode: weights as [2C hi][2C lo] scaled by the powers of two behind the biases
    const float s0 = is_x3<T>() ? bias[(long long)g * bias_gs + 2] : 1.f, s1 = is_x3<T>() ? bias[(long long)g * bias_gs + 3] : 1.f;
    for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (long long)gridDim.x * 256) {
        float a0 = 0.f, a1 = 0.f;
        const T *ps = src + (long long)g * src_gs + nhwc_off<T>(pix, C, 0);
        for (int c = 0; c < C; c += EPV) {
            if constexpr (is_x3<T>()) {
                const T *pp = ps + x3_off(c);
                float f[8];
                x3_join8(*(const uint4 *)pp, *(const uint4 *)(pp + 32), f);
#pragma unroll
                for (int e = 0; e < EPV; e++) {
                    a0 = fmaf(f[e], to_f32(wg[c + e]) + to_f32(wg[2 * C + c + e]), a0);
                    a1 = fmaf(f[e], to_f32(wg[C + c + e]) + to_f32(wg[3 * C + c + e]), a1);
                }
            } else {
                const uint4 raw = *(const uint4 *)(ps + c);
                const T *pv = (const T *)&raw;
#pragma unroll
                for (int e = 0; e < EPV; e++) {
                    const float v = to_f32(pv[e]);
                    a0 = fmaf(v, to_f32(wg[c + e]), a0);
                    a1 = fmaf(v, to_f32(wg[C + c + e]), a1);
                }
            }
        }
        const float o0 = is_x3<T>() ? __fmaf_rn(a0, s0, b0) : a0 + b0, o1 = is_x3<T>() ? __fmaf_rn(a1, s1, b1) : a1 + b1;
        seg[(long long)g * npix + pix] = o0;
        if (rad_f16) ((_Float16 *)radial)[(long long)g * npix + pix] = (_Float16)o1;  // engine option 8
        else radial[(long long)g * npix + pix] = o1;
    }
}

__global__ __launch_bounds__(256) void k_preprocess(const uint8_t *rgb, float *x, int HW, long long total)
{
    RCV_X3_MODE();
    // img/255, -mean, /std in float64, then .float()  (reference AccumulatorSpace.py:143-151)
    const double mean[3] = {0.485, 0.456, 0.406}, sd[3] = {0.229, 0.224, 0.225};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % 3);
        const long long pix = i / 3;
        const long long img = pix / HW, p = pix - img * HW;
        double v = (double)rgb[i] / 255.0;
        v = v - mean[c];
        v = v / sd[c];
        x[(img * 3 + c) * HW + p] = (float)v;
    }
}

// ---------------------------------------------------------------------------------------------
// k_tail (bf16): the last three layers in one kernel -- 2x bilinear upsampling of the 64-channel map, conv7
// (3x3, 64 -> 32, folded BN, ReLU) and the 1x1 head (32 -> 2) -- so that neither the 480x640x64 upsampled map nor
// the 480x640x32 conv7 output ever reaches HBM.  A workgroup walks one strip of 8 x 32-pixel tiles of one
// (group, image); its waves keep the conv7 weights as MFMA fragments in registers for the whole strip.  Per tile:
//   source patch (6 x 18 px) -> LDS;  upsampled patch U (10 x 34 px, zero outside the image: conv padding) built
//   with k_upsample2's arithmetic and rounding;  18 K steps of MFMAs with the A fragments read from U at shifted
//   pixel addresses (one per tap);  bias + ReLU + bf16 rounding into LDS;  one thread per pixel runs the head's
//   sequential fp32 FMA chain.  Same operations in the same order as k_upsample2 -> k_conv -> k_head: identical
//   outputs (tests/test_gpu_fcn.py).
// ---------------------------------------------------------------------------------------------
constexpr int TAIL_TH = 8, TAIL_TW = 32, TAIL_UW = TAIL_TW + 2, TAIL_UH = TAIL_TH + 2;
constexpr int TAIL_SH = TAIL_TH / 2 + 2, TAIL_SW = TAIL_TW / 2 + 2;
constexpr int TAIL_HP = 80;  // bytes per pixel of the conv7 staging rows (64 + bank spread)
constexpr int TAIL_UP = 160;  // bytes per pixel of the upsampled patch: 128 + 32 -- conflict-free for ds_read_b128's lane groups (16 consecutive pixels, two 16-byte columns per group; 144 is 2-way)
constexpr int TAIL_SPIECES = (TAIL_SH * TAIL_SW + 7) / 8;  // DMA pieces (8 pixels x 128 B) of the source patch
constexpr int TAIL_LDS_MAIN = TAIL_UH * TAIL_UW * TAIL_UP + TAIL_SPIECES * 1024;  // U (later the conv7 rows) + source patch
constexpr int TAIL_LDS = TAIL_LDS_MAIN + 64 * 4;  // + the head weights as fp32 (read back as broadcasts)
static_assert(256 * TAIL_HP <= TAIL_UH * TAIL_UW * TAIL_UP, "the conv7 rows reuse the upsampled patch's LDS");

// (plain function: the buffer builtins are device-only and a template kernel's body is also instantiated for the host)
__device__ __forceinline__ void tail_dma16(const void *base, int num_bytes, unsigned char *lds_wave_base, int voffset)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, num_bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds_wave_base, 16, voffset, 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void k_tail(const H16 *src, const H16 *w7, const float *b7, const H16 *w8, const float *b8,
                                                 float *seg, float *radial, int B, int Hin, int Win, long long src_gs, long long w7_gs,
                                                 int b7_gs, long long w8_gs, int b8_gs, int rad_f16, unsigned *range)
{
    RCV_X3_MODE();
    using T = H16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *sU = smem;                                        // [340][TAIL_UP] upsampled patch ...
    unsigned char *sH = sU;                                          // ... later [256][TAIL_HP] conv7 output (bf16)
    unsigned char *sS = smem + TAIL_UH * TAIL_UW * TAIL_UP;          // [6][18][128 B] source patch of the NEXT tile by LDS-DMA
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int Ho = Hin * 2, Wo = Win * 2;
    const int ty = blockIdx.x, img = blockIdx.y, g = blockIdx.z;
    const int ho0 = ty * TAIL_TH;
    const T *sg = src + (long long)g * src_gs + (long long)img * Hin * Win * 64;
    // wave = pixel quarter: 64 pixels x all 32 output channels (two 16-channel tiles).  Its conv7 weights stay in registers
    // as MFMA fragments for the whole strip: lane (co = j*16 + r16, k = tap*64 + kk*32 + kq*8).  (r02: was 128 pixels x 16
    // channels per wave -- every A fragment of the patch was then read from LDS by two waves; the kernel is LDS-read-bound.)
    const int pq = wave;
    bf16x8 bw[2][9][2];
    {
        const T *wg = w7 + (long long)g * w7_gs;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int t = 0; t < 9; t++)
#pragma unroll
                for (int kk = 0; kk < 2; kk++) bw[j][t][kk] = *(const bf16x8 *)(wg + (j * 16 + r16) * 576 + t * 64 + kk * 32 + kq * 8);
    }
    float bias7[2][4];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int e = 0; e < 4; e++) bias7[j][e] = b7[(long long)g * b7_gs + j * 16 + kq * 4 + e];
    float *hw = (float *)(smem + TAIL_LDS_MAIN);  // head weights [2][32] (in LDS: 64 fewer registers per lane)
    if (tid < 64) hw[tid] = (float)w8[(long long)g * w8_gs + tid];
    const float hb0 = b8[(long long)g * b8_gs], hb1 = b8[(long long)g * b8_gs + 1];
    // pixel of this lane in each of the wave's four 16-pixel groups, as index into U (tap (0,0))
    int ubase[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int m = pq * 64 + i * 16 + r16;
        ubase[i] = (m >> 5) * TAIL_UW + (m & 31);
    }
    const int sy0 = ho0 / 2 - 1;
    // source patch of a tile (6 x 18 pixels x 128 B, rows / columns clamped into the image: only clamped positions are ever
    // referenced) as TAIL_SPIECES DMA pieces of 8 pixels, piece q by wave q % 4.  It is requested while the previous tile is
    // in its MFMA phase, so that the tile loop never waits for HBM (r02; before, every tile began with a synchronous load).
    const int src_bytes = Hin * Win * 128;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    int s_row[(TAIL_SPIECES + 3) / 4], s_col[(TAIL_SPIECES + 3) / 4];  // byte offset of the patch row; patch column, or -1
#pragma unroll
    for (int k = 0; k < (TAIL_SPIECES + 3) / 4; k++) {
        const int pp = (k * 4 + wv) * 8 + (lane >> 3);
        const int a = pp / TAIL_SW, b = pp - a * TAIL_SW;
        s_row[k] = min(max(sy0 + a, 0), Hin - 1) * Win * 128 + (lane & 7) * 16;
        s_col[k] = (k * 4 + wv < TAIL_SPIECES && pp < TAIL_SH * TAIL_SW) ? b : -1;
    }
#define RCV_TAIL_SOURCE(sx0_)                                                                                  \
    _Pragma("unroll") for (int k = 0; k < (TAIL_SPIECES + 3) / 4; k++)                                         \
        if (k * 4 + wv < TAIL_SPIECES)                                                                       \
            tail_dma16(sg, src_bytes, sS + (k * 4 + wv) * 1024,                                              \
                       s_col[k] >= 0 ? s_row[k] + min(max((sx0_) + s_col[k], 0), Win - 1) * 128 : (int)0x80000000);
    RCV_TAIL_SOURCE(-1);
    for (int tx = 0; tx * TAIL_TW < Wo; tx++) {
        const int wo0 = tx * TAIL_TW;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the source patch (and its stores of the previous tile)
        __syncthreads();  // the whole patch is in sS; every thread is done with the previous tile's conv7 rows in sU
        // ---- upsampled patch: one task = one source cell (rows ai, ai+1; columns bi, bi+1 of the source patch) and one
        // channel chunk -> the 2 x 2 patch pixels (2ai.., 2bi..) that interpolate inside that cell (k_upsample2's scheme;
        // the patch rows / columns pair up with the cells because ho0 and wo0 are even)
#pragma unroll 1
        for (int v = tid; v < (TAIL_UH / 2) * (TAIL_UW / 2) * 8; v += 256) {
            const int c = v & 7, cell = v >> 3, ai = cell / (TAIL_UW / 2), bi = cell - ai * (TAIL_UW / 2);
            const unsigned char *r0 = sS + (ai * TAIL_SW + bi) * 128 + c * 16, *r1 = r0 + TAIL_SW * 128;
            const uint4 q00 = *(const uint4 *)r0, q01 = *(const uint4 *)(r0 + 128), q10 = *(const uint4 *)r1, q11 = *(const uint4 *)(r1 + 128);
            const T *p00 = (const T *)&q00, *p01 = (const T *)&q01, *p10 = (const T *)&q10, *p11 = (const T *)&q11;
            float f00[8], f01[8], f10[8], f11[8];
#pragma unroll
            for (int e = 0; e < 8; e++) { f00[e] = (float)p00[e]; f01[e] = (float)p01[e]; f10[e] = (float)p10[e]; f11[e] = (float)p11[e]; }
#pragma unroll
            for (int dy = 0; dy < 2; dy++) {
                const int uy = 2 * ai + dy, y = ho0 - 1 + uy;
                int y0, y1;
                float ly, hy;
                up2_src(y, Hin, y0, y1, ly, hy);
#pragma unroll
                for (int dx = 0; dx < 2; dx++) {
                    const int ux = 2 * bi + dx, x = wo0 - 1 + ux;
                    int x0, x1;
                    float lx, hx;
                    up2_src(x, Win, x0, x1, lx, hx);
                    uint4 o = make_uint4(0u, 0u, 0u, 0u);
                    if ((unsigned)y < (unsigned)Ho && (unsigned)x < (unsigned)Wo) {
                        T *po = (T *)&o;
#pragma unroll
                        for (int e = 0; e < 8; e++) po[e] = (T)bilerp(hy, hx, ly, lx, f00[e], f01[e], f10[e], f11[e]);
                    }
                    const int u = uy * TAIL_UW + ux;
                    *(uint4 *)(sU + u * TAIL_UP + c * 16) = o;
                }
            }
        }
        __syncthreads();  // U is complete, sS is free
        if ((tx + 1) * TAIL_TW < Wo) RCV_TAIL_SOURCE((wo0 + TAIL_TW) / 2 - 1);
        // ---- conv7: K = 9 taps x 64 channels, in k_conv's order (tap-major, 32 channels per MFMA)
        f32x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int toff = (t / 3) * TAIL_UW + (t % 3);
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                bf16x8 af[4];
#pragma unroll
                for (int i = 0; i < 4; i++)  // constant offset from the lane's pixel: folds into the ds_read immediate
                    af[i] = *(const bf16x8 *)(sU + ubase[i] * TAIL_UP + kq * 16 + (toff * TAIL_UP + kk * 64));
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[i][j] = RCV_MFMA16(bw[j][t][kk], af[i], acc[i][j], 0, 0, 0);
                asm volatile("" ::: "memory");  // keep the fragment reads of later K steps from being hoisted (registers)
            }
        }
        __syncthreads();  // every wave is done reading U: its LDS takes the conv7 rows
        // ---- bias + ReLU, rounded to bf16 as the stored conv7 activation would be; operands were swapped, so
        // acc[i][j][e] is channel j*16 + kq*4 + e of pixel pq*64 + i*16 + r16
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                uint2 pk;
                T *pv = (T *)&pk;
#pragma unroll
                for (int e = 0; e < 4; e++) pv[e] = (T)fmaxf(acc[i][j][e] + bias7[j][e], 0.f);
                *(uint2 *)(sH + (pq * 64 + i * 16 + r16) * TAIL_HP + (j * 16 + kq * 4) * 2) = pk;
            }
        __syncthreads();
        // ---- head: thread = pixel, the sequential FMA chain of k_head
        {
            const int ho = ho0 + (tid >> 5), wo = wo0 + (tid & 31);
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int c = 0; c < 32; c += 8) {
                const uint4 raw = *(const uint4 *)(sH + tid * TAIL_HP + c * 2);
                const T *pv = (const T *)&raw;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float v = (float)pv[e];
                    a0 = fmaf(v, hw[c + e], a0);
                    a1 = fmaf(v, hw[32 + c + e], a1);
                }
            }
            if (ho < Ho && wo < Wo) {
                const long long o = ((long long)(g * B + img) * Ho + ho) * Wo + wo;
                seg[o] = a0 + hb0;
                if (rad_f16) ((_Float16 *)radial)[o] = (_Float16)(a1 + hb1);  // engine option 8
                else radial[o] = a1 + hb1;
            }
        }
    }
#undef RCV_TAIL_SOURCE
}

// ---------------------------------------------------------------------------------------------
// k_chain_x3<CMID> (split precision): the 1x1 expand of a bottleneck block (CMID -> 4*CMID, + residual, ReLU) and the NEXT
// block's 1x1 reduce (4*CMID -> CMID, ReLU) in one kernel (reference models/fcnresnet.py:30-43 and :24-26 of the following
// block).  Unfused, the 4*CMID-channel block output is written by the expand kernel and read back twice -- as the reduce's
// operand and, one block later, as the residual; in hi / lo storage that tensor is 354 MB per launch at 24 frames x 3 networks,
// and both 1x1 layers sit at the HBM roofline (3.3 TB/s), not at the matrix pipe.  Here the reduce consumes every 128-channel
// chunk of the block output straight from registers, so the tensor is written once and read once.
//   Decomposition: a workgroup owns 128 pixels, a wave 16 of them -- all channels of its pixels, so no activation ever crosses
//   waves.  The wave keeps its rows of the expand's input as MFMA fragments in registers (CMID / 32 slices x (hi, lo)) and the
//   reduce's accumulators (16 px x CMID channels).  Only the weights travel through LDS: a ring of three 32 KB slots filled
//   by LDS-DMA two steps ahead, one K slice per step -- [128 out][32 ch (hi | lo)] of the expand for chunk nc, then four slices
//   [CMID out][32 ch] of the reduce over the chunk just produced.
//   Per chunk: KE expand steps (8 channel tiles x 3 MFMAs each) -> per 32-channel slice: the accumulators (operands swapped:
//   lane = pixel, 4 consecutive channels) go through a wave-private LDS transposition as fp32 and come back as 8 consecutive
//   channels per lane; + bias (x scale) + residual, ReLU, split into hi / lo -- the two 16-byte vectors are the block output
//   (stored as full 128-byte lines per pixel) AND the reduce's operand fragments -> 3 MFMAs x CMID / 16 tiles.
//   Every global access is a 16-byte vector; the counted vmcnt of a step covers exactly what was issued since its weights.
// Results: bit-identical to the two separate kernels -- same K order and MFMA sequence per output element (the operands are
// swapped, which transposes the product and nothing else), the same fma / add / max in the epilogues
// (tests/test_gpu_fcn_x3.py::test_x3_kernel_variants_identical), so a frame's maps do not depend on which path its batch takes.
// Measured: LDS-bound (every wave reads every weight fragment: 2 ds_read_b128 per 3 MFMAs, matrix cores 35 % busy), forward
// -5 % against the two HBM-bound kernels at 36 frames x 3 networks.
// ---------------------------------------------------------------------------------------------
// LDS accesses the compiler must not see: after an LDS-DMA it guards every visible LDS access that might alias the DMA's
// destination with s_waitcnt vmcnt(0), which drains the weight prefetch.  The caller waits on lgkmcnt itself.
__device__ __forceinline__ void lds_st128(unsigned addr, f32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
// (reads come with their wait in ONE asm statement: the compiler may schedule a consumer of the output anywhere after the
// statement that defines it, also in front of a separate s_waitcnt)
__device__ __forceinline__ void lds_ld128x2(unsigned a0, unsigned a1, f32x4 &v0, f32x4 &v1)
{
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(a0), "v"(a1) : "memory");
}
__device__ __forceinline__ void lds_ld128x4(unsigned a0, unsigned a1, unsigned a2, unsigned a3, f32x4 &v0, f32x4 &v1, f32x4 &v2, f32x4 &v3)
{
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
}

constexpr int CH_BM = 128;
constexpr int CH_SLOT = 32 * 1024, CH_RING = 3;
constexpr int CH_XP = 144;  // bytes per pixel row of a wave's transposition scratch: 32 fp32 + 16 (bank spread)
constexpr int CH_BIAS = (2 * 1024 + 2 * 256) * 4;  // [bias | scale] of both layers as fp32 (largest block: 256 -> 1024 -> 256)
constexpr int CH_LDS = CH_RING * CH_SLOT + 8 * 16 * CH_XP + CH_BIAS;

template <int CMID>
__global__ __launch_bounds__(512) void k_chain_x3(const BlockParams p)
{
    RCV_X3_MODE();
    using T = H16;
    constexpr int COUT = 4 * CMID, NC = COUT / 128, KE = CMID / 32, KR = 4;  // 128-channel chunks; K slices per chunk: expand, reduce
    constexpr int NTR = CMID / 16;                                            // 16-channel tiles of the reduce's output
    constexpr int PE = 2, PR = CMID / 64;                                     // DMA pieces (8 rows x 128 B) per wave and slice
    constexpr int SPC = KE + KR;                                              // steps per chunk
    static_assert(CMID * 128 <= CH_SLOT, "a reduce-weight slice fills at most one ring slot");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4, lrow = lane >> 3, lpos = lane & 7;
    const int g = blockIdx.z;
    const int m0 = blockIdx.x * CH_BM;
    const T *src = (const T *)p.src + (long long)g * p.src_gs;
    const T *res = (const T *)p.res + (long long)g * p.res_gs;
    T *dst = (T *)p.dst + (long long)g * p.dst_gs;
    T *dst3 = (T *)p.dst3 + (long long)g * p.dst3_gs;
    const char *w2 = (const char *)((const T *)p.w2 + (long long)g * p.w2_gs);
    const char *w3 = (const char *)((const T *)p.w3 + (long long)g * p.w3_gs);
    const float *b2 = p.b2 + (long long)g * p.b2_gs, *b3 = p.b3 + (long long)g * p.b3_gs;  // [bias | scale]
    const int w2_bytes = COUT * CMID * 4, w3_bytes = CMID * COUT * 4;
    // (wave-private transposition scratch rows follow the ring: smem + CH_RING * CH_SLOT + wave * 16 * CH_XP)
    // [bias | scale] of both layers in LDS: read per step as ds_read_b128 (as global loads they would sit in the vector-memory
    // queue behind the weight pieces issued just before, and waiting for them would drain the prefetch)
    float *sB2 = (float *)(smem + CH_RING * CH_SLOT + 8 * 16 * CH_XP), *sB3 = sB2 + 2 * COUT;
    for (int i = tid; i < 2 * COUT; i += 512) sB2[i] = b2[i];
    for (int i = tid; i < 2 * CMID; i += 512) sB3[i] = b3[i];
    // LDS byte addresses for the inline-asm accesses (the dynamic LDS block starts at address 0 of the workgroup's allocation)
    const unsigned aX = (unsigned)(CH_RING * CH_SLOT + wave * 16 * CH_XP + r16 * CH_XP);
    const unsigned aB2 = (unsigned)(CH_RING * CH_SLOT + 8 * 16 * CH_XP), aB3 = aB2 + 2 * COUT * 4;

    // this lane's pixel (rows beyond M: the last row's data, never stored)
    const int m = m0 + wave * 16 + r16;
    const long long mrow = m < p.M ? m : p.M - 1;

    // ---- weights of step (chunk nc, slice q) into ring slot `slot`
    // expand slice: rows = out channels nc*128 .. +127 of w2 [COUT][KE][64 halves]; reduce slice t: all CMID rows of w3 [CMID][COUT/32][64], K slice nc*4 + t
    // lane -> (row of the piece, 16-byte position); the source chunk is position ^ ((row >> 1) & 7) (the tiles' bank swizzle)
    auto issue = [&](int nc, int q, int slot) {
        unsigned char *sl = smem + slot * CH_SLOT;
        if (q < KE) {
#pragma unroll
            for (int c = 0; c < PE; c++) {
                const int row = (wave * PE + c) * 8 + lrow;
                const int off = ((nc * 128 + row) * (2 * CMID) + q * 64 + ((lpos ^ ((row >> 1) & 7)) << 3)) * 2;
                blk_dma16(w2, w2_bytes, sl + (wave * PE + c) * 1024, off);
            }
        } else {
#pragma unroll
            for (int c = 0; c < PR; c++) {
                const int row = (wave * PR + c) * 8 + lrow;
                const int off = (row * (2 * COUT) + (nc * 4 + (q - KE)) * 64 + ((lpos ^ ((row >> 1) & 7)) << 3)) * 2;
                blk_dma16(w3, w3_bytes, sl + (wave * PR + c) * 1024, off);
            }
        }
    };
    // VM operations the step at position q of a chunk issues at its top: the DMA pieces of the step two ahead, and -- at the first
    // step of a chunk -- the chunk's residual vectors (4 slices x (hi, lo)), a whole expand phase before they are needed
    auto pieces_for = [&](int q2) { return q2 < KE ? PE : PR; };
    auto loads_at = [&](int q_) { return q_ == 0 ? 2 * KR : 0; };

    // ---- prologue: the expand's operand rows of this lane's pixel, the weights of steps 0 and 1
    bf16x8 aE[KE][2];
    {
        const T *sp = src + mrow * (2 * CMID) + kq * 8;
#pragma unroll
        for (int s_ = 0; s_ < KE; s_++) {
            aE[s_][0] = *(const bf16x8 *)(sp + s_ * 64);
            aE[s_][1] = *(const bf16x8 *)(sp + s_ * 64 + 32);
        }
    }
    issue(0, 0, 0);
    issue(0, 1, 1);
    f32x4 accR[NTR];
#pragma unroll
    for (int j = 0; j < NTR; j++) accR[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accE[8];
    uint4 rres[KR][2];  // the chunk's residual vectors (hi, lo) per reduce step
    const int swz = (r16 >> 1) & 7;
    const int fh = ((kq ^ swz) << 4), fl = (((4 + kq) ^ swz) << 4);  // fragment chunks (hi, lo) inside a 128-byte weight row
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (lgkmcnt: the bias rows written above)
    __builtin_amdgcn_s_barrier();

#if defined(RCV_HALO_STAMPS) && defined(RCV_CHAIN_STAMPS)
    unsigned hst[64];
    for (int k = 0; k < 64; k++) hst[k] = 0u;
    const bool stamp_blk = blockIdx.x == gridDim.x / 2 && blockIdx.z == 0 && (wave & 3) == 0 && CMID == 256;
    bool stamp_on = false;
#define RCV_CSTAMP(idx) RCV_HSTAMP(idx)
#else
#define RCV_CSTAMP(idx)
#endif
    int slot = 0;
    for (int nc = 0; nc < NC; nc++) {
        const bool last = nc == NC - 1;
#if defined(RCV_HALO_STAMPS) && defined(RCV_CHAIN_STAMPS)
        stamp_on = stamp_blk && nc == 3;
#endif
#pragma unroll
        for (int q = 0; q < SPC; q++) {
            // ---- wait for this step's weights (issued two steps ago): everything this wave issued after them may stay in flight
            RCV_CSTAMP(q * 4 + 0)
            if (nc > 0 || q >= 1) {
                if (nc > 0 || q >= 2) {  // (steps 0 and 1: the prologue waited)
                    const int qm1 = (q + SPC - 1) % SPC, qm2 = (q + SPC - 2) % SPC;
                    const int s_m2 = qm2 >= KE ? 2 : 0, s_m1 = qm1 >= KE ? 2 : 0;       // stores of the two previous steps
                    const bool next_exists = !(last && q == SPC - 1);
                    const int i_m1 = next_exists ? pieces_for((q + 1) % SPC) : 0;       // pieces step g-1 issued for step g+1
                    blk_vmwait(loads_at(qm2) + s_m2 + i_m1 + loads_at(qm1) + s_m1);
                }
                RCV_CSTAMP(q * 4 + 1)
                __builtin_amdgcn_s_barrier();  // every wave's pieces have landed; everybody is done with the slot refilled below
                RCV_CSTAMP(q * 4 + 2)
            }
            // ---- issue for step g + 2
            {
                const int q2 = (q + 2) % SPC, nc2 = nc + (q + 2 >= SPC ? 1 : 0);
                if (nc2 < NC) issue(nc2, q2, slot + 2 >= CH_RING ? slot + 2 - CH_RING : slot + 2);
                if (q == 0) {  // this chunk's residual: 8 channels kq*8.. of the slices nc*4 + t (hi, lo)
#pragma unroll
                    for (int t2 = 0; t2 < KR; t2++) {
                        const T *rp = res + mrow * (2 * COUT) + (nc * 4 + t2) * 64 + kq * 8;
                        rres[t2][0] = *(const uint4 *)rp;
                        rres[t2][1] = *(const uint4 *)(rp + 32);
                    }
                }
            }
            RCV_CSTAMP(q * 4 + 3)
            const unsigned char *sl = smem + slot * CH_SLOT;
            if (q < KE) {
                // ---- expand step: chunk nc, K slice q -- 8 channel tiles, operands swapped (lane = pixel, 4 channels per tile)
                if (q == 0) {
#pragma unroll
                    for (int j = 0; j < 8; j++) accE[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                // (the weight fragments of tile j + 1 are requested before the three MFMAs of tile j: left to itself hipcc reuses one
                // register set and the wave sits out an LDS round trip per pair of tiles -- matrix cores 0.35 busy)
                bf16x8 wh[2], wl[2];
                wh[0] = *(const bf16x8 *)(sl + r16 * 128 + fh);
                wl[0] = *(const bf16x8 *)(sl + r16 * 128 + fl);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if (j + 1 < 8) {
                        wh[(j + 1) & 1] = *(const bf16x8 *)(sl + ((j + 1) * 16 + r16) * 128 + fh);
                        wl[(j + 1) & 1] = *(const bf16x8 *)(sl + ((j + 1) * 16 + r16) * 128 + fl);
                    }
                    accE[j] = RCV_MFMA16(wh[j & 1], aE[q][0], accE[j], 0, 0, 0);
                    accE[j] = RCV_MFMA16(wl[j & 1], aE[q][0], accE[j], 0, 0, 0);
                    accE[j] = RCV_MFMA16(wh[j & 1], aE[q][1], accE[j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);  // keep the order: reads of j + 1, then the MFMAs of j
                }
            } else {
                // ---- reduce step t: the chunk's channels t*32 .. +31 -> transposition, epilogue of the expand, store, 3 x NTR MFMAs
                const int t = q - KE;
                // acc[e] = channel (tile)*16 + kq*4 + e of pixel r16, scaled and biased, as fp32 into the scratch row of the pixel
                f32x4 bv[2], sv[2];
                {
                    const int ch = nc * 128 + 2 * t * 16 + kq * 4;
                    lds_ld128x4(aB2 + ch * 4, aB2 + (ch + 16) * 4, aB2 + (COUT + ch) * 4, aB2 + (COUT + ch + 16) * 4, bv[0], bv[1], sv[0], sv[1]);
                }
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const f32x4 a = accE[2 * t + h];
                    lds_st128(aX + h * 64 + kq * 16, f32x4{__fmaf_rn(a[0], sv[h][0], bv[h][0]), __fmaf_rn(a[1], sv[h][1], bv[h][1]),
                                                           __fmaf_rn(a[2], sv[h][2], bv[h][2]), __fmaf_rn(a[3], sv[h][3], bv[h][3])});
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (wave-private rows: no barrier)
                float v[8];
                {
                    f32x4 v0, v1;
                    lds_ld128x2(aX + kq * 32, aX + kq * 32 + 16, v0, v1);
                    v[0] = v0[0]; v[1] = v0[1]; v[2] = v0[2]; v[3] = v0[3]; v[4] = v1[0]; v[5] = v1[1]; v[6] = v1[2]; v[7] = v1[3];
                }
                float rv[8];
                x3_join8(rres[t][0], rres[t][1], rv);
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = fmaxf(v[e] + rv[e], 0.f);
                rng_flush(p.range, rng_see<8>(0.f, v));  // (checked on the spot: the kernel has no register to carry a maximum)
                uint4 oh, ol;
                x3_split8(v, oh, ol);
                {   // (rows beyond M recompute row M - 1 and store the same values there: every wave issues the same number of
                    // vector-memory operations, which is what the counted waits rely on)
                    T *op = dst + mrow * (2 * COUT) + (nc * 4 + t) * 64 + kq * 8;
                    *(uint4 *)op = oh;
                    *(uint4 *)(op + 32) = ol;
                }
                const bf16x8 ah = *(const bf16x8 *)&oh, al = *(const bf16x8 *)&ol;
                bf16x8 wh[2], wl[2];
                wh[0] = *(const bf16x8 *)(sl + r16 * 128 + fh);
                wl[0] = *(const bf16x8 *)(sl + r16 * 128 + fl);
#pragma unroll
                for (int j = 0; j < NTR; j++) {
                    if (j + 1 < NTR) {
                        wh[(j + 1) & 1] = *(const bf16x8 *)(sl + ((j + 1) * 16 + r16) * 128 + fh);
                        wl[(j + 1) & 1] = *(const bf16x8 *)(sl + ((j + 1) * 16 + r16) * 128 + fl);
                    }
                    accR[j] = RCV_MFMA16(wh[j & 1], ah, accR[j], 0, 0, 0);
                    accR[j] = RCV_MFMA16(wl[j & 1], ah, accR[j], 0, 0, 0);
                    accR[j] = RCV_MFMA16(wh[j & 1], al, accR[j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (q == SPC - 1) { RCV_CSTAMP(SPC * 4) }
            slot = slot + 1 == CH_RING ? 0 : slot + 1;
        }
    }
#if defined(RCV_HALO_STAMPS) && defined(RCV_CHAIN_STAMPS)
    if (stamp_blk && lane == 0)
        for (int k = 0; k < 64; k++) g_chain_stamps[wave >> 2][k] = hst[k];
#endif
    // ---- the reduce's epilogue: bias (x scale), ReLU, hi / lo, through the same transposition, 32 channels at a time
#pragma unroll
    for (int t = 0; t < NTR / 2; t++) {
        f32x4 bv[2], sv[2];
        {
            const int ch = 2 * t * 16 + kq * 4;
            lds_ld128x4(aB3 + ch * 4, aB3 + (ch + 16) * 4, aB3 + (CMID + ch) * 4, aB3 + (CMID + ch + 16) * 4, bv[0], bv[1], sv[0], sv[1]);
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const f32x4 a = accR[2 * t + h];
            lds_st128(aX + h * 64 + kq * 16, f32x4{fmaxf(__fmaf_rn(a[0], sv[h][0], bv[h][0]), 0.f), fmaxf(__fmaf_rn(a[1], sv[h][1], bv[h][1]), 0.f),
                                                   fmaxf(__fmaf_rn(a[2], sv[h][2], bv[h][2]), 0.f), fmaxf(__fmaf_rn(a[3], sv[h][3], bv[h][3]), 0.f)});
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float v[8];
        {
            f32x4 v0, v1;
            lds_ld128x2(aX + kq * 32, aX + kq * 32 + 16, v0, v1);  // (the wait inside also covers the rewrite of the row by the next slice)
            v[0] = v0[0]; v[1] = v0[1]; v[2] = v0[2]; v[3] = v0[3]; v[4] = v1[0]; v[5] = v1[1]; v[6] = v1[2]; v[7] = v1[3];
        }
        rng_flush(p.range, rng_see<8>(0.f, v));
        uint4 oh, ol;
        x3_split8(v, oh, ol);
        T *op = dst3 + mrow * (2 * CMID) + t * 64 + kq * 8;
        *(uint4 *)op = oh;
        *(uint4 *)(op + 32) = ol;
    }
}

// ---------------------------------------------------------------------------------------------
// k_chain2_x3<CMID> (split precision, r04): the same two layers as k_chain_x3 -- a bottleneck's 1x1 expand (+ residual, ReLU)
// and the next block's 1x1 reduce -- with the same decomposition (a workgroup owns 128 pixels, a wave 16 of them with all their
// channels; only weights travel through LDS, by LDS-DMA two steps ahead) and the same arithmetic per output element (K order,
// hh / hl / lh per slice, fma / add / max of the epilogue): bit-identical outputs.  What changed is where the time went:
//   * No transposition.  With the operands swapped the accumulator of a 16-channel tile holds, per lane (pixel r16, quarter
//     kq), the four output channels of MFMA rows kq*4 .. +3 -- while the next MFMA (and a 16-byte store) wants eight CONSECUTIVE
//     channels kq*8 .. +7 per lane.  k_chain_x3 sent every 32-channel slice through a wave-private fp32 LDS round trip (three
//     dependent LDS latencies per reduce step, in front of its MFMAs).  Which weight row feeds which MFMA row is free, though:
//     lane r16 of tile jj of a 32-channel slice reads the weights of output channel (r16 >> 2) * 8 + jj * 4 + (r16 & 3), so the
//     two tiles' accumulators of a lane ARE channels kq*8 .. +3 and kq*8+4 .. +7 of its pixel.  Nothing moves; the per-element
//     sums are the same MFMAs on the same operands.  The weight image's bank swizzle follows the new row pattern
//     (c2_key: 16 lanes of a ds_read_b128 group -> 16 distinct 16-byte columns), and every fragment read of a step is one
//     base register + an immediate offset.
//   * Half chunks.  A chunk is 64 output channels of the expand (two K slices of the reduce) instead of 128: 16 accumulator and
//     16 residual registers less, which pay for a second weight-fragment set in flight, and the residual arrives as four loads
//     per wave eight times per block instead of eight loads four times (the stalls of k_chain_x3 sat where those bursts met
//     the queued stores).  An expand step multiplies two K slices (2 x [64 out][32 ch] = 16 KB of weights) so that a step
//     still carries 24 MFMAs per wave behind its barrier.
//   * The epilogue of slice t + 1 (scale, bias, residual, ReLU, split, store -- ~90 VALU instructions per lane) is in program
//     order BEFORE the MFMAs of reduce step t and free of LDS round trips, so the scheduler interleaves it with those MFMAs
//     instead of both waves of a SIMD doing VALU work while the matrix pipe idles; slice 0's sits in the last expand step.
// ---------------------------------------------------------------------------------------------
#ifndef RCV_C2_SE
#define RCV_C2_SE 0  // K slices of the expand per step (2: 24 MFMAs per wave behind a barrier; 4: 48, two barriers less per half chunk).  0 = per
                     // form (r06, same box, ms per forward): the 256-wide form 4 (with the spread issue below 22.75 against 23.55 at 2; 4 alone 23.0),
                     // the 128-wide form 2 (7.23 against 7.51 at 4)
#endif
constexpr int c2_se(int cmid)  // K slices of the expand per step
{
    const int want = RCV_C2_SE ? RCV_C2_SE : (cmid == 256 ? 4 : 2);
    return want < cmid / 32 ? want : cmid / 32;
}
constexpr int c2_slot(int cmid) { return cmid * 128 > c2_se(cmid) * 8192 ? cmid * 128 : c2_se(cmid) * 8192; }  // ring slot: the larger of the two step images
#ifndef RCV_C2_DIST
#define RCV_C2_DIST 2  // steps between the issue of a step's weights and their use (ring slots = RCV_C2_DIST + 1)
#endif
#ifndef RCV_C2_DIAG
#define RCV_C2_DIAG 0  // timing diagnostics with WRONG outputs: 1 = residual loads / block-output stores always hit the same line (no HBM traffic), 2 = weights always of the first chunk (L2-hot), 3 = no weight DMA after the first half chunk, 4 = 1 + 3, 5 = no epilogue pieces (r06; invalid: the expand's MFMAs are dead code then), 6 = no residual loads, 7 = no block-output stores, 8 = neither (the epilogue's arithmetic alone), 9 = residual / block output addressed slice-major ([slice][pixel][32 ch (hi | lo)]: 2 KB contiguous per wave and slice instead of 16 x 128 B at the row stride; same bytes)
#endif
#ifndef RCV_C2_DEPHASE
#define RCV_C2_DEPHASE 0  // cycles per phase unit of the start-up delay (blockIdx.x % 6 units): spreads the workgroups' store / load bursts
#endif
constexpr int C2_DIST = RCV_C2_DIST, C2_RING = C2_DIST + 1;
constexpr int c2_lds(int cmid) { return C2_RING * c2_slot(cmid) + (2 * 4 * cmid + 2 * cmid) * 4; }  // ring + [bias | scale] of both layers
static_assert(c2_lds(256) <= 160 * 1024, "k_chain2_x3's LDS plan");
// bank swizzle of the weight images: 16-byte column c of row r lives at column c ^ c2_key(r).  A fragment read takes, for its 16
// lanes, the rows (r16 >> 2) * 8 + (r16 & 3) (+ multiples of 4 and 32): bits 1, 3 and 4 of the row separate them into eight
// keys x two row parities = sixteen distinct 16-byte columns of the 256-byte bank row.
__device__ __forceinline__ int c2_key(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }
// LDS reads by inline asm (not tracked by the compiler's lgkmcnt model; its own counted waits only get stricter): the bias /
// scale vectors of an epilogue are requested at the top of a step and waited for where they are used
__device__ __forceinline__ void lds_req128(unsigned a, f32x4 &v) { asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); }
__device__ __forceinline__ void lds_arrive4(f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

// NW (r06): waves per workgroup.  8 = 128 pixels per workgroup.  4 (64 pixels): the 128-wide form's 54 KB of LDS and 162 registers let THREE
// four-wave workgroups share a CU (12 waves) where one eight-wave workgroup ran alone -- each brings its own weight ring (twice the L2 ->
// LDS weight traffic per pixel), a wave issues twice the DMA pieces per step
#ifndef RCV_C2_SPREAD
#define RCV_C2_SPREAD -1  // 1 = the weight pieces of a step are issued between its tiles instead of all at its top (needs RCV_C2_UNCOND), 0 = at its top,
                          // -1 = per form (r06): on for the 256-wide form with its four-slice expand steps (-0.3 ms on top of them; with two-slice steps
                          // +0.4) and the 128-wide form (7.00-7.07 against 7.11-7.13), off for the 64-wide form (5.73-5.76 against 5.70-5.71)
#endif
constexpr bool c2_spread(int cmid) { return RCV_C2_SPREAD < 0 ? cmid >= 128 : RCV_C2_SPREAD != 0; }
#ifndef RCV_C2_NT
#define RCV_C2_NT 0  // cache policy experiment: 1 = the block-output stores nontemporal, 2 = the residual loads, 3 = both
#endif
#ifndef RCV_C2_UNCOND
#define RCV_C2_UNCOND 1  // 0 = the weight pieces past the last half chunk are branched around (r04-r05: see the issue in the step loop)
#endif
static_assert(RCV_C2_UNCOND || RCV_C2_SPREAD == 0, "the spread issue counts on every step having its pieces");
#ifndef RCV_C2_FD
#define RCV_C2_FD 2  // tiles of weight fragments in flight ahead of the MFMAs.  Measured r06 (same box, ms per 36 launches of the 256-wide form): 2: 22.96-23.05, 3: 23.02-23.04, 4: 23.03-23.08 -- the fragments' LDS latency is not what the kernel waits for
#endif
template <int CMID, bool T3 = false, int NW = 8>
__global__ __launch_bounds__(64 * NW, (NW == 4 ? 3 : (CMID == 64 && !T3) ? 4 : 2)) void k_chain2_x3(const BlockParams p)  // (CMID = 64: two workgroups per CU = 128 registers -- the HBM-bound form lives on its occupancy)
{
    RCV_X3_MODE();
    using T = H16;
    constexpr int COUT = 4 * CMID, HC = COUT / 64, KE = CMID / 32;  // half chunks of 64 output channels; K slices of the expand
    constexpr int FD = RCV_C2_FD, NF = FD + 1;  // weight-fragment sets requested ahead of the tile being multiplied
    static_assert(!T3 || CMID <= 128, "the 3-byte trunk exists where the kernel is HBM-bound (layers 1-2); the 256-wide form has no VALU slot to spare");
    constexpr int SE = c2_se(CMID);                                 // K slices of the expand per step
    constexpr int ES = KE / SE, SPC = ES + 2;                       // expand steps + two reduce steps per half chunk
    constexpr int NTR = CMID / 16;                                  // 16-channel tiles of the reduce's output
    constexpr int PE = SE * 8 / NW, PR = CMID / 64 * 8 / NW;        // DMA pieces (8 rows x 128 B) per wave: expand step, reduce step
    constexpr int SLOT = c2_slot(CMID);
    constexpr bool SPREAD = c2_spread(CMID);
    constexpr bool C2_NOLOAD = RCV_C2_DIAG == 6 || RCV_C2_DIAG == 8, C2_NOSTORE = RCV_C2_DIAG == 7 || RCV_C2_DIAG == 8;  // (timing only)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4, lrow = lane >> 3, lpos = lane & 7;
    const int g = blockIdx.z;
    const int m0 = blockIdx.x * (16 * NW);
    if constexpr (RCV_C2_DEPHASE > 0) {
        // Workgroups that start together run their half chunks in step: every CU stores, then every CU loads, and the memory
        // system sees bursts it takes microseconds to drain -- which the in-order vmcnt then charges to the weight stream.  A start-up
        // delay of blockIdx.x % 6 units spreads the phases over the half chunk's period (placement only: results unchanged).
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const unsigned long long wait = (unsigned long long)(blockIdx.x % 6u) * RCV_C2_DEPHASE;
        while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
    const T *src = (const T *)p.src + (long long)g * p.src_gs;
    const T *res = (const T *)p.res + (long long)g * p.res_gs;
    T *dst = (T *)p.dst + (long long)g * p.dst_gs;
    T *dst3 = (T *)p.dst3 + (long long)g * p.dst3_gs;
    const char *w2 = (const char *)((const T *)p.w2 + (long long)g * p.w2_gs);
    const char *w3 = (const char *)((const T *)p.w3 + (long long)g * p.w3_gs);
    const float *b2 = p.b2 + (long long)g * p.b2_gs, *b3 = p.b3 + (long long)g * p.b3_gs;  // [bias | scale]
    const int w2_bytes = COUT * CMID * 4, w3_bytes = CMID * COUT * 4;
    float *sB2 = (float *)(smem + C2_RING * SLOT), *sB3 = sB2 + 2 * COUT;
    for (int i = tid; i < 2 * COUT; i += 64 * NW) sB2[i] = b2[i];
    for (int i = tid; i < 2 * CMID; i += 64 * NW) sB3[i] = b3[i];
    const unsigned aB2 = (unsigned)(C2_RING * SLOT) + kq * 32, aB3 = (unsigned)(C2_RING * SLOT + 2 * COUT * 4) + kq * 32;  // this lane's 8 channels

    const int m = m0 + wave * 16 + r16;
    const long long mrow = m < p.M ? m : p.M - 1;  // (rows beyond M recompute and re-store row M - 1: equal VM operation counts)

    // ---- weights of step (half chunk h, position q) into ring slot `slot`: lane -> (row of the piece, 16-byte column).  r06: one
    // descriptor per weight tensor, the lane's part of a piece's offset (row, swizzled column) precomputed, the step's part (half chunk,
    // K slices) in the DMA's scalar offset: no vector instruction per piece (bit-identical; 24.07 vs 24.10 ms per 36 launches of the
    // 256-wide form on one box: this kernel does not wait for its instruction count)
    const __amdgpu_buffer_rsrc_t rW2 = blk_rsrc(w2, w2_bytes), rW3 = blk_rsrc(w3, w3_bytes);
    int voffE[PE], voffR[PR];
#pragma unroll
    for (int c = 0; c < PE; c++) {
        const int pid = wave * PE + c, ks = pid >> 3, row = (pid & 7) * 8 + lrow;
        voffE[c] = (row * (2 * CMID) + ks * 64 + ((lpos ^ c2_key(row)) << 3)) * 2;
    }
#pragma unroll
    for (int c = 0; c < PR; c++) {
        const int row = (wave * PR + c) * 8 + lrow;
        voffR[c] = (row * (2 * COUT) + ((lpos ^ c2_key(row)) << 3)) * 2;
    }
    auto issue_piece = [&](int h, int q, int slot, int c) {  // piece c of the step's pieces_for(q)
        unsigned char *sl = smem + slot * SLOT;
        const int hh = RCV_C2_DIAG == 2 ? 0 : h;
        if (q < ES)  // [SE K slices][64 out channels h*64 ..][32 ch (hi | lo)] of w2 [COUT][KE][64 halves]
            blk_dma16r(rW2, sl + (wave * PE + c) * 1024, voffE[c], (hh * 64 * (2 * CMID) + SE * q * 64) * 2);
        else         // [CMID out channels][32 ch] of w3 [CMID][COUT / 32][64 halves], K slice h*2 + t
            blk_dma16r(rW3, sl + (wave * PR + c) * 1024, voffR[c], (hh * 2 + (q - ES)) * 128);
    };
    auto issue = [&](int h, int q, int slot) {
        if (q < ES) {
#pragma unroll
            for (int c = 0; c < PE; c++) issue_piece(h, q, slot, c);
        } else {
#pragma unroll
            for (int c = 0; c < PR; c++) issue_piece(h, q, slot, c);
        }
    };
    // vector-memory operations a step issues besides the weights of the step two ahead: the half chunk's residual vectors
    // (2 slices x (hi, lo)) at its first step; the stores of slice 0 in the last expand step and of slice 1 in reduce step 0
    auto pieces_for = [&](int q2) { return q2 < ES ? PE : PR; };
    auto loads_at = [&](int q_) { return q_ == 0 && !C2_NOLOAD ? 4 : 0; };
    auto stores_at = [&](int q_) { return (q_ == ES - 1 || q_ == ES) && !C2_NOSTORE ? 2 : 0; };
    auto extra_at = [&](int q_) { return loads_at(q_) + stores_at(q_); };

    bf16x8 aE[KE][2];
    {
        const T *sp = src + mrow * (2 * CMID) + kq * 8;
#pragma unroll
        for (int s_ = 0; s_ < KE; s_++) {
            aE[s_][0] = *(const bf16x8 *)(sp + s_ * 64);
            aE[s_][1] = *(const bf16x8 *)(sp + s_ * 64 + 32);
        }
    }
#pragma unroll
    for (int k = 0; k < C2_DIST; k++) issue(k / SPC, k % SPC, k);
    f32x4 accR[NTR];
#pragma unroll
    for (int j = 0; j < NTR; j++) accR[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accE[4];
    uint4 rres[2][2];
    bf16x8 ah, al;        // operand fragments of the reduce step in flight (slice t of the half chunk, hi / lo)
    float rmax = 0.f;     // range guard: largest magnitude stored
    // fragment addressing: row (r16 >> 2) * 8 + (r16 & 3) of a 32-row group, columns kq (hi) and 4 + kq (lo) under the row's key
    const int fkey = ((r16 >> 1) & 1) | ((r16 >> 2) << 1);
    const unsigned frow = (unsigned)(((r16 >> 2) * 8 + (r16 & 3)) * 128);
    const unsigned fhi = frow + ((kq ^ fkey) << 4), flo = frow + (((4 + kq) ^ fkey) << 4);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // (r06) The compiler's wait-count pass does not read the wait above: to it the operand fragments loaded before the loop stay
    // "in flight" at the loop's entry, and it guards their first uses in EVERY iteration with its own s_waitcnt vmcnt(23), (22), ...
    // (4) ladder -- in-order counts that are tighter than the counted waits below and drain the weight stream a step early.  Using
    // the registers here makes it place its wait where it costs nothing.
#pragma unroll
    for (int s_ = 0; s_ < KE; s_++) asm volatile("" : "+v"(aE[s_][0]), "+v"(aE[s_][1]));
    __builtin_amdgcn_s_barrier();

    // The epilogue of slice t of half chunk h (accumulators of tiles 2t, 2t + 1 -> eight consecutive channels of this lane's pixel:
    // scale, bias, + residual, ReLU, split, store; the two 16-byte vectors are also the reduce's operand fragments), cut into
    // EP_N pieces of a few VALU instructions.  The steps hand them out one or three per tile, between the MFMAs, each slot
    // closed by a scheduling fence: both waves of a SIMD then alternate matrix and vector work instead of doing ~90 VALU
    // instructions side by side while the matrix pipe idles.  Element for element the operations of k_chain_x3's epilogue.
    constexpr int EP_N = 12;
    f32x2 evp[4], ervp[4];  // the slice's eight values / residuals as pairs
    f32x4 ebi0, ebi1, esc0, esc1;
    uint4 eoh, eol;
    auto ep_request = [&](int h, int t) {  // bias / scale of the slice's eight channels: LDS reads left in flight
        const unsigned a = aB2 + (unsigned)(h * 64 + t * 32) * 4;
        lds_req128(a, ebi0); lds_req128(a + 16, ebi1); lds_req128(a + COUT * 4, esc0); lds_req128(a + COUT * 4 + 16, esc1);
    };
    auto ep_piece = [&](int h, int t, int k) {
        if constexpr (RCV_C2_DIAG == 5) return;  // (NOT a valid "no epilogue" timing: the expand's accumulators are dead then and the compiler drops its MFMAs -- the build
                                                 // times the reduce alone, tools/halo_stamps.py --chain2 shows expand steps of ~100 cycles.  Use 8.)
        unsigned *wh_ = (unsigned *)&eoh, *wl_ = (unsigned *)&eol;
        const unsigned *rh = (const unsigned *)&rres[t][0], *rl = (const unsigned *)&rres[t][1];
        switch (k) {
        case 0:  // (LDS reads complete in order: the fragment reads issued after the request have been consumed by now)
            if constexpr (FD == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ebi0), "+v"(ebi1), "+v"(esc0), "+v"(esc1));  // (at most FD tiles' fragments are newer: two reads each)
            else if constexpr (FD == 3) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(ebi0), "+v"(ebi1), "+v"(esc0), "+v"(esc1));
            else asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(ebi0), "+v"(ebi1), "+v"(esc0), "+v"(esc1));
            // (r06: pairs -- v_pk_fma_f32 / v_pk_add_f32 do two of the same IEEE operations per instruction: 19 % fewer vector instructions,
            // no time gained.  tools/probe/mfma_valu_coissue_probe.hip says why: beside a SIMD's MFMAs an ordinary vector instruction
            // costs ~1.6 cycles (mostly hidden), a packed-fp32 one ~7.7: it issues at half rate and is not hidden.)
            evp[0] = __builtin_elementwise_fma(__builtin_shufflevector(accE[2 * t], accE[2 * t], 0, 1), __builtin_shufflevector(esc0, esc0, 0, 1), __builtin_shufflevector(ebi0, ebi0, 0, 1));
            evp[1] = __builtin_elementwise_fma(__builtin_shufflevector(accE[2 * t], accE[2 * t], 2, 3), __builtin_shufflevector(esc0, esc0, 2, 3), __builtin_shufflevector(ebi0, ebi0, 2, 3));
            break;
        case 1:
            evp[2] = __builtin_elementwise_fma(__builtin_shufflevector(accE[2 * t + 1], accE[2 * t + 1], 0, 1), __builtin_shufflevector(esc1, esc1, 0, 1), __builtin_shufflevector(ebi1, ebi1, 0, 1));
            evp[3] = __builtin_elementwise_fma(__builtin_shufflevector(accE[2 * t + 1], accE[2 * t + 1], 2, 3), __builtin_shufflevector(esc1, esc1, 2, 3), __builtin_shufflevector(ebi1, ebi1, 2, 3));
            break;
        case 2: case 3: {  // residual: hi + lo of channels 4(k-2) .. +3
            const int b = (k - 2) * 2;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                if (T3 && p.t3_res) {  // words b + e of the top halves; bytes 2 (b + e), 2 (b + e) + 1 of the low bytes
                    const unsigned lo2 = rl[(b + e) >> 1] >> (16 * ((b + e) & 1));
                    ervp[b + e] = f32x2{t3_decode(rh[b + e] & 0xffffu, lo2 & 0xffu), t3_decode(rh[b + e] >> 16, (lo2 >> 8) & 0xffu)};
                } else
                    ervp[b + e] = x3_join2<true>(rh[b + e], rl[b + e]);
            }
            break;
        }
        case 4: case 5: {
            const int b = (k - 4) * 2;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const f32x2 s2 = evp[b + e] + ervp[b + e];
                evp[b + e] = f32x2{fmaxf(s2.x, 0.f), fmaxf(s2.y, 0.f)};
            }
            break;
        }
        case 6: rmax = rng_see<8>(rmax, (const float *)evp); break;
        case 7: case 8: case 9: case 10: {  // split of elements 2(k-7), 2(k-7) + 1 (x3_split8's operations)
            const int e = k - 7;
            x3_split2<true>(evp[e].x, evp[e].y, wh_[e], wl_[e]);
            break;
        }
        default: {  // 11: the block output (full 128-byte lines per pixel over the wave's two stores)
            if (T3 && p.t3_dst) {  // the slice's eight values (still in ev) in the 3-byte form, encoded where they are stored: nothing of it
                                   // stays live across the MFMA phases (the [hi | lo] words above remain the reduce's operand)
                unsigned e3h[4], e3l[2];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    unsigned t0, l0, t1, l1;
                    t3_encode(evp[e].x, t0, l0);
                    t3_encode(evp[e].y, t1, l1);
                    e3h[e] = t0 | (t1 << 16);
                    const unsigned two = l0 | (l1 << 8);
                    if ((e & 1) == 0) e3l[e >> 1] = two;
                    else e3l[e >> 1] |= two << 16;
                }
                char *op = (char *)dst + mrow * (3 * COUT) + (h * 2 + t) * 96;
                *(uint4 *)(op + kq * 16) = make_uint4(e3h[0], e3h[1], e3h[2], e3h[3]);
                *(uint2 *)(op + 64 + kq * 8) = make_uint2(e3l[0], e3l[1]);
            } else {
                T *op = RCV_C2_DIAG == 9 ? dst + ((long long)(h * 2 + t) * p.M + mrow) * 64 + kq * 8
                                         : dst + mrow * (2 * COUT) + ((RCV_C2_DIAG == 1 || RCV_C2_DIAG == 4 ? 0 : h) * 2 + t) * 64 + kq * 8;
                if constexpr (C2_NOSTORE) {
                    asm volatile("" :: "v"(eoh.x), "v"(eoh.y), "v"(eoh.z), "v"(eoh.w), "v"(eol.x), "v"(eol.y), "v"(eol.z), "v"(eol.w));
                } else if constexpr ((RCV_C2_NT & 1) != 0) {
                    __builtin_nontemporal_store(*(const i32x4 *)&eoh, (i32x4 *)op);
                    __builtin_nontemporal_store(*(const i32x4 *)&eol, (i32x4 *)(op + 32));
                } else {
                    *(uint4 *)op = eoh;
                    *(uint4 *)(op + 32) = eol;
                }
            }
            break;
        }
        }
    };

    bf16x8 fwh[NF], fwl[NF];  // weight fragments (hi, lo) of FD + 1 tiles: FD in flight ahead of the one being multiplied
#if defined(RCV_HALO_STAMPS) && defined(RCV_CHAIN2_STAMPS)
    // diagnostic build (tools/halo_stamps.py --chain2): four stamps per step over half chunks 6 and 7 of one workgroup in the
    // middle of the launch -- step top, after the counted vmcnt wait, after the barrier, after the issue (DMA pieces, residual)
    unsigned hst[64];
    for (int k = 0; k < 64; k++) hst[k] = 0u;
    const bool stamp_blk = blockIdx.x == gridDim.x / 2 && blockIdx.z == 0 && (wave & 3) == 0 && CMID == 256;
    bool stamp_on = false;
#define RCV_C2STAMP(k) RCV_HSTAMP(((h - 6) * SPC + q) * 4 + (k))
#else
#define RCV_C2STAMP(k)
#endif
    int slot = 0;
    for (int h = 0; h < HC; h++) {
#if defined(RCV_HALO_STAMPS) && defined(RCV_CHAIN2_STAMPS)
        stamp_on = stamp_blk && (h == 6 || h == 7);
#endif
#pragma unroll
        for (int q = 0; q < SPC; q++) {
            RCV_C2STAMP(0)
            // ---- this step's weights (issued two steps ago) have landed in every wave's share; the slot refilled below is free
            if (h > 0 || q >= 1) {
                if (h > 0 || q >= C2_DIST) {  // (the first C2_DIST steps: the prologue waited)
                    // younger than this step's weights (issued C2_DIST steps ago, first thing of that step): that step's loads and
                    // stores, then per step since: the pieces for the step C2_DIST ahead of it (if there is one) and its loads / stores
                    // (RCV_C2_SPREAD: the pieces ride between the step's tiles, after its residual loads and before its stores)
                    int n = SPREAD ? stores_at((q + 2 * SPC - C2_DIST) % SPC) : extra_at((q + 2 * SPC - C2_DIST) % SPC);
#pragma unroll
                    for (int k = 1; k < C2_DIST; k++) {
                        const bool exists = RCV_C2_UNCOND || h * SPC + q + k < HC * SPC;
                        n += (exists ? pieces_for((q + k) % SPC) : 0) + extra_at((q + 2 * SPC - C2_DIST + k) % SPC);
                    }
                    blk_vmwait(n);
                }
                RCV_C2STAMP(1)
                __builtin_amdgcn_s_barrier();
            }
            RCV_C2STAMP(2)
            __builtin_amdgcn_sched_barrier(0);
            {
                const int q2 = (q + C2_DIST) % SPC, h2 = h + (q + C2_DIST) / SPC;
                // (r06: issued unconditionally -- past the last half chunk the pieces re-read its weights into the free slot, 2 steps of
                // 96.  A branch around the issue makes the compiler's wait for the residual registers assume the path WITHOUT the
                // pieces: s_waitcnt vmcnt(2) where 12 younger operations are in flight, i.e. the weight stream drained twice per half chunk.)
                const int h2c = RCV_C2_UNCOND ? (h2 < HC ? h2 : HC - 1) : h2;
                if (!SPREAD && (RCV_C2_UNCOND || h2 < HC) && !((RCV_C2_DIAG == 3 || RCV_C2_DIAG == 4) && h2 > 0)) issue(h2c, q2, slot + C2_DIST >= C2_RING ? slot + C2_DIST - C2_RING : slot + C2_DIST);
                if (q == 0) {  // this half chunk's residual: channels kq*8 .. +7 of the slices h*2 + t (hi, lo)
#pragma unroll
                    for (int t2 = 0; t2 < 2; t2++) {
                        if constexpr (C2_NOLOAD) {
                            asm volatile("" : "=v"(rres[t2][0].x), "=v"(rres[t2][0].y), "=v"(rres[t2][0].z), "=v"(rres[t2][0].w), "=v"(rres[t2][1].x), "=v"(rres[t2][1].y), "=v"(rres[t2][1].z), "=v"(rres[t2][1].w));
                        } else if (T3 && p.t3_res) {  // 96 bytes per group: [32 top halves | 32 low bytes] (two loads, like the [hi | lo] row)
                            const char *rp = (const char *)res + mrow * (3 * COUT) + (h * 2 + t2) * 96;
                            rres[t2][0] = *(const uint4 *)(rp + kq * 16);
                            const uint2 lo8 = *(const uint2 *)(rp + 64 + kq * 8);
                            rres[t2][1] = make_uint4(lo8.x, lo8.y, 0u, 0u);
                        } else {
                            const T *rp = RCV_C2_DIAG == 9 ? res + ((long long)(h * 2 + t2) * p.M + mrow) * 64 + kq * 8  // (diag 9: slice-major addresses, same bytes)
                                                           : res + mrow * (2 * COUT) + ((RCV_C2_DIAG == 1 || RCV_C2_DIAG == 4 ? 0 : h) * 2 + t2) * 64 + kq * 8;  // (diag 1: one hot line)
                            if constexpr ((RCV_C2_NT & 2) != 0) {
                                const i32x4 n0 = __builtin_nontemporal_load((const i32x4 *)rp), n1 = __builtin_nontemporal_load((const i32x4 *)(rp + 32));
                                rres[t2][0] = *(const uint4 *)&n0;
                                rres[t2][1] = *(const uint4 *)&n1;
                            } else {
                                rres[t2][0] = *(const uint4 *)rp;
                                rres[t2][1] = *(const uint4 *)(rp + 32);
                            }
                        }
                    }
                }
            }
            RCV_C2STAMP(3)
            // RCV_C2_SPREAD: the weights of the step C2_DIST ahead, one piece before every (tiles / pieces)-th tile of this step: the
            // texture path takes a 1 KB piece per ~16 cycles and CU, and eight waves issuing 2-4 pieces each right behind the barrier
            // queue there for 200-500 cycles with the matrix pipe idle (stamps: "issue")
            auto spread_issue = [&](int tile, int ntiles) {
                if constexpr (SPREAD) {
                    const int q2 = (q + C2_DIST) % SPC, h2 = h + (q + C2_DIST) / SPC;
                    const int np = pieces_for(q2), every = ntiles / np;
                    if (tile % every == 0 && tile / every < np) {
                        const int h2c = h2 < HC ? h2 : HC - 1;
                        issue_piece(h2c, q2, slot + C2_DIST >= C2_RING ? slot + C2_DIST - C2_RING : slot + C2_DIST, tile / every);
                    }
                }
            };
            const unsigned sbase = (unsigned)(slot * SLOT);
            const unsigned char *slh = smem + sbase + fhi, *sll = smem + sbase + flo;
            auto rd = [&](int buf, int o) {
                fwh[buf] = *(const bf16x8 *)(slh + o);
                fwl[buf] = *(const bf16x8 *)(sll + o);
            };
            if (q < ES) {
                // ---- expand step: K slices 2q, 2q + 1 x four 16-channel tiles, tile pair (= slice of the half chunk) outermost, so
                // that slice 0's accumulators are final half a step before the step ends and its epilogue rides on the second half
                if (q == 0) {
#pragma unroll
                    for (int j = 0; j < 4; j++) accE[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                constexpr int NTE = 4 * SE;  // tiles of the step: [tile pair][K slice][tile of the pair]
                auto off = [&](int i) { return ((i >> 1) % SE) * 8192 + (i / (2 * SE)) * 4096 + (i & 1) * 512; };
                if (q == ES - 1) ep_request(h, 0);
                __builtin_amdgcn_sched_barrier(0);  // (the requests stay older than every fragment read of the step)
#pragma unroll
                for (int f = 0; f < FD; f++)
                    if (f < NTE) rd(f, off(f));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NTE; i++) {
                    spread_issue(i, NTE);
                    if (i + FD < NTE) rd((i + FD) % NF, off(i + FD));
                    const int ks = (i >> 1) % SE;
                    f32x4 &a = accE[(i / (2 * SE)) * 2 + (i & 1)];
                    a = RCV_MFMA16(fwh[i % NF], aE[SE * q + ks][0], a, 0, 0, 0);
                    a = RCV_MFMA16(fwl[i % NF], aE[SE * q + ks][0], a, 0, 0, 0);
                    a = RCV_MFMA16(fwh[i % NF], aE[SE * q + ks][1], a, 0, 0, 0);
                    if (q == ES - 1 && i >= NTE / 2) {  // slice 0 is final: its epilogue rides on the second tile pair's MFMAs
                        constexpr int PPT = (EP_N + NTE / 2 - 1) / (NTE / 2);
#pragma unroll
                        for (int k = 0; k < PPT; k++)
                            if ((i - NTE / 2) * PPT + k < EP_N) ep_piece(h, 0, (i - NTE / 2) * PPT + k);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (q == ES - 1) {
                    ah = *(const bf16x8 *)&eoh;
                    al = *(const bf16x8 *)&eol;
                }
            } else {
                // ---- reduce step t: 3 x NTR MFMAs on slice t's fragments; beside them, in step 0, the epilogue of slice 1
                const int t = q - ES;
                auto off = [](int j) { return (j >> 1) * 4096 + (j & 1) * 512; };
                if (t == 0) ep_request(h, 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < FD; f++)
                    if (f < NTR) rd(f, off(f));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NTR; j++) {
                    spread_issue(j, NTR);
                    if (j + FD < NTR) rd((j + FD) % NF, off(j + FD));
                    accR[j] = RCV_MFMA16(fwh[j % NF], ah, accR[j], 0, 0, 0);
                    accR[j] = RCV_MFMA16(fwl[j % NF], ah, accR[j], 0, 0, 0);
                    accR[j] = RCV_MFMA16(fwh[j % NF], al, accR[j], 0, 0, 0);
                    if (t == 0) {  // EP_N pieces over the tiles from the second on (the bias reads are older than tile 2's fragments)
                        constexpr int PPT = (EP_N + NTR - 2) / (NTR - 1);
#pragma unroll
                        for (int k = 0; k < PPT; k++)
                            if (j >= 1 && (j - 1) * PPT + k < EP_N) ep_piece(h, 1, (j - 1) * PPT + k);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (t == 0) {
                    ah = *(const bf16x8 *)&eoh;
                    al = *(const bf16x8 *)&eol;
                }
            }
            slot = slot + 1 == C2_RING ? 0 : slot + 1;
        }
#if defined(RCV_HALO_STAMPS) && defined(RCV_CHAIN2_STAMPS)
        if (stamp_on && h == 7) {
            const int q = SPC;  // (the end of the second stamped half chunk)
            RCV_C2STAMP(0)
        }
#endif
    }
#if defined(RCV_HALO_STAMPS) && defined(RCV_CHAIN2_STAMPS)
    if (stamp_blk && lane == 0)
        for (int k = 0; k < 64; k++) g_chain_stamps[wave >> 2][k] = hst[k];
#endif
#undef RCV_C2STAMP
    // ---- the reduce's epilogue: scale, bias, ReLU, hi / lo; tiles 2t, 2t + 1 are channels t*32 + kq*8 .. +7 of this lane's pixel
#pragma unroll
    for (int t = 0; t < NTR / 2; t++) {
        f32x4 bi0, bi1, sc0, sc1;
        const unsigned a = aB3 + (unsigned)(t * 32) * 4;
        lds_req128(a, bi0); lds_req128(a + 16, bi1); lds_req128(a + CMID * 4, sc0); lds_req128(a + CMID * 4 + 16, sc1);
        lds_arrive4(bi0, bi1, sc0, sc1);
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            v[e] = fmaxf(__fmaf_rn(accR[2 * t][e], sc0[e], bi0[e]), 0.f);
            v[4 + e] = fmaxf(__fmaf_rn(accR[2 * t + 1][e], sc1[e], bi1[e]), 0.f);
        }
        rmax = rng_see<8>(rmax, v);
        uint4 oh, ol;
        x3_split8(v, oh, ol);
        T *op = dst3 + mrow * (2 * CMID) + t * 64 + kq * 8;
        *(uint4 *)op = oh;
        if (p.q6_dst3) {
            // q6 row (k_q6_pack's result, written here): the lanes kq, kq ^ 1 of a pixel hold channels 0 .. 7 and 8 .. 15 of a 16-channel
            // block.  v_permlane16_swap(oh, ol) leaves, in the even rows of lanes, (hi of the pair's first lane, hi of its second) and in
            // the odd rows (lo of the first, lo of the second): every lane holds one whole block -- even kq the hi block, odd kq the lo
            // block of channels (kq >> 1) * 16 .. + 15 -- converts it and stores chunk 4 + (kq >> 1) + 2 * (kq & 1).
            uint4 b0, b1;
#if defined(RCV_FCN_X3)
#define RCV_SWAP16(c)                                                                          \
    {                                                                                          \
        const auto r_ = __builtin_amdgcn_permlane16_swap(oh.c, ol.c, false, false);            \
        b0.c = r_[0];                                                                          \
        b1.c = r_[1];                                                                          \
    }
            RCV_SWAP16(x) RCV_SWAP16(y) RCV_SWAP16(z) RCV_SWAP16(w)
#undef RCV_SWAP16
#else
            b0 = oh; b1 = ol;
#endif
            *(i32x4 *)(op + 32 + ((kq >> 1) + 2 * (kq & 1) - kq) * 8) = q6_block(b0, b1, 0);
        } else
            *(uint4 *)(op + 32) = ol;
    }
    if constexpr (RCV_C2_DIAG == 0) rng_flush(p.range, rmax);  // (the diagnostic builds compute garbage on purpose)
}

// ---------------------------------------------------------------------------------------------
// k_chain3_x3<CMID> (split precision, r04 experiment): k_chain2_x3's two layers with FOUR waves per workgroup, one per SIMD, each
// owning 32 pixels (two 16-pixel sub-tiles) and the whole register file (512 per lane).  Why: with every memory stream removed
// k_chain2_x3<256> still takes 530 of its 686 us (RCV_C2_DIAG) -- each of its eight waves re-reads every weight fragment for 16
// pixels (two ds_read_b128 per three MFMAs: two thirds of the LDS array's time), and a ~76-instruction VALU epilogue per slice finds
// one free issue slot per MFMA between two waves that issue MFMAs in lockstep.  Here a fragment pair feeds SIX MFMAs (both
// sub-tiles), so the LDS traffic halves, and the lone wave of a SIMD has twelve vector slots per tile for the epilogue pieces and the
// DMA pieces, which are dealt out between the tiles instead of bunched at the step's top.  Same decomposition otherwise (weights
// through a three-slot LDS ring two steps ahead, half chunks of 64 output channels, weight rows chosen so that a lane's accumulators
// are eight consecutive channels), same MFMA sequence per output element: bit-identical to k_chain2_x3 / k_chain_x3 / the plain kernels.
// ---------------------------------------------------------------------------------------------
template <int CMID>
__global__ __launch_bounds__(256) void k_chain3_x3(const BlockParams p)
{
    RCV_X3_MODE();
    using T = H16;
    constexpr int COUT = 4 * CMID, HC = COUT / 64, KE = CMID / 32;
    constexpr int SE = 2, ES = KE / SE, SPC = ES + 2;   // two K slices per expand step; expand steps + two reduce steps per half chunk
    constexpr int NTR = CMID / 16;
    constexpr int PE = 16 / 4, PR = (CMID / 8) / 4;     // DMA pieces (8 rows x 128 B) per wave: expand step (16 KB), reduce step
    constexpr int SLOT = CMID * 128 > 16384 ? CMID * 128 : 16384;
    constexpr int NS = 2;                               // 16-pixel sub-tiles per wave
    static_assert(CMID == 256, "register plan written for the 256-wide block");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4, lrow = lane >> 3, lpos = lane & 7;
    const int g = blockIdx.z;
    const int m0 = blockIdx.x * CH_BM;
    const T *src = (const T *)p.src + (long long)g * p.src_gs;
    const T *res = (const T *)p.res + (long long)g * p.res_gs;
    T *dst = (T *)p.dst + (long long)g * p.dst_gs;
    T *dst3 = (T *)p.dst3 + (long long)g * p.dst3_gs;
    const char *w2 = (const char *)((const T *)p.w2 + (long long)g * p.w2_gs);
    const char *w3 = (const char *)((const T *)p.w3 + (long long)g * p.w3_gs);
    const float *b2 = p.b2 + (long long)g * p.b2_gs, *b3 = p.b3 + (long long)g * p.b3_gs;
    const int w2_bytes = COUT * CMID * 4, w3_bytes = CMID * COUT * 4;
    float *sB2 = (float *)(smem + 3 * SLOT), *sB3 = sB2 + 2 * COUT;
    for (int i = tid; i < 2 * COUT; i += 256) sB2[i] = b2[i];
    for (int i = tid; i < 2 * CMID; i += 256) sB3[i] = b3[i];
    const unsigned aB2 = (unsigned)(3 * SLOT) + kq * 32, aB3 = (unsigned)(3 * SLOT + 2 * COUT * 4) + kq * 32;

    long long mrow[NS];
#pragma unroll
    for (int s_ = 0; s_ < NS; s_++) {
        const int m = m0 + wave * 32 + s_ * 16 + r16;
        mrow[s_] = m < p.M ? m : p.M - 1;  // (rows beyond M recompute and re-store row M - 1: equal VM operation counts)
    }

    // ---- piece c of this wave's share of the weights of step (half chunk h, position q) into ring slot `slot`
    auto issue_piece = [&](int h, int q, int slot, int c) {
        unsigned char *sl = smem + slot * SLOT;
        if (q < ES) {
            const int pid = wave * PE + c, ks = pid >> 3, row = (pid & 7) * 8 + lrow;
            const int off = ((h * 64 + row) * (2 * CMID) + (SE * q + ks) * 64 + ((lpos ^ c2_key(row)) << 3)) * 2;
            blk_dma16(w2, w2_bytes, sl + pid * 1024, off);
        } else {
            const int pid = wave * PR + c, row = pid * 8 + lrow;
            const int off = (row * (2 * COUT) + (h * 2 + (q - ES)) * 64 + ((lpos ^ c2_key(row)) << 3)) * 2;
            blk_dma16(w3, w3_bytes, sl + pid * 1024, off);
        }
    };
    auto pieces_for = [&](int q2) { return q2 < ES ? PE : PR; };
    // besides the pieces: the half chunk's residual vectors (2 slices x (hi, lo) x 2 sub-tiles) at its first step; the stores of
    // slice 0 (2 x 2 sub-tiles) in the last expand step and of slice 1 in reduce step 0
    auto extra_at = [&](int q_) { return (q_ == 0 ? 4 * NS : 0) + (q_ == ES - 1 || q_ == ES ? 2 * NS : 0); };

    bf16x8 aE[NS][KE][2];
#pragma unroll
    for (int s_ = 0; s_ < NS; s_++) {
        const T *sp = src + mrow[s_] * (2 * CMID) + kq * 8;
#pragma unroll
        for (int k_ = 0; k_ < KE; k_++) {
            aE[s_][k_][0] = *(const bf16x8 *)(sp + k_ * 64);
            aE[s_][k_][1] = *(const bf16x8 *)(sp + k_ * 64 + 32);
        }
    }
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
        for (int c = 0; c < PE; c++) issue_piece(0, k, k, c);  // (steps 0 and 1 are expand steps: ES >= 2)
    static_assert(ES >= 2, "the prologue issues two expand steps");
    f32x4 accR[NS][NTR];
#pragma unroll
    for (int s_ = 0; s_ < NS; s_++)
#pragma unroll
        for (int j = 0; j < NTR; j++) accR[s_][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accE[NS][4];
    uint4 rres[NS][2][2];
    bf16x8 ah[NS], al[NS];
    float rmax = 0.f;
    const int fkey = ((r16 >> 1) & 1) | ((r16 >> 2) << 1);
    const unsigned frow = (unsigned)(((r16 >> 2) * 8 + (r16 & 3)) * 128);
    const unsigned fhi = frow + ((kq ^ fkey) << 4), flo = frow + (((4 + kq) ^ fkey) << 4);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // epilogue of slice t of half chunk h for sub-tile s_, in EP_N pieces (k_chain2_x3's, element for element)
    constexpr int EP_N = 12;
    float ev[8], erv[8];
    f32x4 ebi0, ebi1, esc0, esc1;
    uint4 eoh, eol;
    auto ep_request = [&](int h, int t) {
        const unsigned a = aB2 + (unsigned)(h * 64 + t * 32) * 4;
        lds_req128(a, ebi0); lds_req128(a + 16, ebi1); lds_req128(a + COUT * 4, esc0); lds_req128(a + COUT * 4 + 16, esc1);
    };
    auto ep_piece = [&](int h, int t, int s_, int k) {
        unsigned *wh_ = (unsigned *)&eoh, *wl_ = (unsigned *)&eol;
        const unsigned *rh = (const unsigned *)&rres[s_][t][0], *rl = (const unsigned *)&rres[s_][t][1];
        switch (k) {
        case 0:
            if (s_ == 0) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ebi0), "+v"(ebi1), "+v"(esc0), "+v"(esc1));
#pragma unroll
            for (int e = 0; e < 4; e++) ev[e] = __fmaf_rn(accE[s_][2 * t][e], esc0[e], ebi0[e]);
            break;
        case 1:
#pragma unroll
            for (int e = 0; e < 4; e++) ev[4 + e] = __fmaf_rn(accE[s_][2 * t + 1][e], esc1[e], ebi1[e]);
            break;
        case 2: case 3: {
            const int b = (k - 2) * 2;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const f32x2 s2 = x3_join2<true>(rh[b + e], rl[b + e]);
                erv[2 * (b + e)] = s2[0];
                erv[2 * (b + e) + 1] = s2[1];
            }
            break;
        }
        case 4: case 5: {
            const int b = (k - 4) * 4;
#pragma unroll
            for (int e = 0; e < 4; e++) ev[b + e] = fmaxf(ev[b + e] + erv[b + e], 0.f);
            break;
        }
        case 6: rmax = rng_see<8>(rmax, ev); break;
        case 7: case 8: case 9: case 10: {
            const int e = k - 7;
            x3_split2<true>(ev[2 * e], ev[2 * e + 1], wh_[e], wl_[e]);
            break;
        }
        default: {
            T *op = dst + mrow[s_] * (2 * COUT) + (h * 2 + t) * 64 + kq * 8;
            *(uint4 *)op = eoh;
            *(uint4 *)(op + 32) = eol;
            break;
        }
        }
    };

    bf16x8 fwh[3], fwl[3];
    int slot = 0;
    for (int h = 0; h < HC; h++) {
#pragma unroll
        for (int q = 0; q < SPC; q++) {
            if (h > 0 || q >= 1) {
                if (h > 0 || q >= 2) {
                    // this step's weights were the LAST vector-memory operations of the step before last (every tile body ends with
                    // its DMA pieces, the last piece goes out at the last tile): younger are exactly the operations of the previous
                    // step -- its loads / stores and the pieces it dealt out for the next step (if there is one)
                    const bool exists = h * SPC + q + 1 < HC * SPC;
                    blk_vmwait((exists ? pieces_for((q + 1) % SPC) : 0) + extra_at((q + 2 * SPC - 1) % SPC));
                }
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
            // the step two ahead, whose pieces are dealt out between this step's tiles
            const int q2 = (q + 2) % SPC, h2 = h + (q + 2) / SPC;
            const int slot2 = slot + 2 >= 3 ? slot + 2 - 3 : slot + 2;
            const int np2 = h2 < HC ? pieces_for(q2) : 0;
            if (q == 0) {
#pragma unroll
                for (int s_ = 0; s_ < NS; s_++)
#pragma unroll
                    for (int t2 = 0; t2 < 2; t2++) {
                        const T *rp = res + mrow[s_] * (2 * COUT) + (h * 2 + t2) * 64 + kq * 8;
                        rres[s_][t2][0] = *(const uint4 *)rp;
                        rres[s_][t2][1] = *(const uint4 *)(rp + 32);
                    }
            }
            const unsigned sbase = (unsigned)(slot * SLOT);
            const unsigned char *slh = smem + sbase + fhi, *sll = smem + sbase + flo;
            auto rd = [&](int buf, int o) {
                fwh[buf] = *(const bf16x8 *)(slh + o);
                fwl[buf] = *(const bf16x8 *)(sll + o);
            };
            if (q < ES) {
                if (q == 0) {
#pragma unroll
                    for (int s_ = 0; s_ < NS; s_++)
#pragma unroll
                        for (int j = 0; j < 4; j++) accE[s_][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                constexpr int NTE = 4 * SE;
                auto off = [&](int i) { return ((i >> 1) % SE) * 8192 + (i / (2 * SE)) * 4096 + (i & 1) * 512; };
                if (q == ES - 1) ep_request(h, 0);
                __builtin_amdgcn_sched_barrier(0);
                rd(0, off(0));
                rd(1, off(1));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NTE; i++) {
                    if (i + 2 < NTE) rd((i + 2) % 3, off(i + 2));
                    const int ks = (i >> 1) % SE, ti = (i / (2 * SE)) * 2 + (i & 1);
#pragma unroll
                    for (int s_ = 0; s_ < NS; s_++) {
                        f32x4 &a = accE[s_][ti];
                        a = RCV_MFMA16(fwh[i % 3], aE[s_][SE * q + ks][0], a, 0, 0, 0);
                        a = RCV_MFMA16(fwl[i % 3], aE[s_][SE * q + ks][0], a, 0, 0, 0);
                        a = RCV_MFMA16(fwh[i % 3], aE[s_][SE * q + ks][1], a, 0, 0, 0);
                    }
                    if (q == ES - 1 && i >= NTE / 2) {  // slice 0 is final: both sub-tiles' epilogues over the second tile pair
                        constexpr int PPT = (2 * EP_N + NTE / 2 - 1) / (NTE / 2);
#pragma unroll
                        for (int k = 0; k < PPT; k++) {
                            const int pc = (i - NTE / 2) * PPT + k;
                            if (pc < 2 * EP_N) {
                                ep_piece(h, 0, pc / EP_N, pc % EP_N);
                                if (pc % EP_N == EP_N - 1) { ah[pc / EP_N] = *(const bf16x8 *)&eoh; al[pc / EP_N] = *(const bf16x8 *)&eol; }
                            }
                        }
                    }
                    // the step-after-next's pieces, dealt evenly over the tiles; the tile body ends with them (see the counted wait)
#pragma unroll
                    for (int c = 0; c < PR; c++)
                        if (c >= i * np2 / NTE && c < (i + 1) * np2 / NTE) issue_piece(h2, q2, slot2, c);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                const int t = q - ES;
                auto off = [](int j) { return (j >> 1) * 4096 + (j & 1) * 512; };
                if (t == 0) ep_request(h, 1);
                __builtin_amdgcn_sched_barrier(0);
                rd(0, off(0));
                rd(1, off(1));
                __builtin_amdgcn_sched_barrier(0);
                bf16x8 nh[NS], nl[NS];
#pragma unroll
                for (int j = 0; j < NTR; j++) {
                    if (j + 2 < NTR) rd((j + 2) % 3, off(j + 2));
#pragma unroll
                    for (int s_ = 0; s_ < NS; s_++) {
                        accR[s_][j] = RCV_MFMA16(fwh[j % 3], ah[s_], accR[s_][j], 0, 0, 0);
                        accR[s_][j] = RCV_MFMA16(fwl[j % 3], ah[s_], accR[s_][j], 0, 0, 0);
                        accR[s_][j] = RCV_MFMA16(fwh[j % 3], al[s_], accR[s_][j], 0, 0, 0);
                    }
                    if (t == 0) {  // slice 1's epilogues (both sub-tiles) over the tiles from the second on
                        constexpr int PPT = (2 * EP_N + NTR - 2) / (NTR - 1);
#pragma unroll
                        for (int k = 0; k < PPT; k++) {
                            const int pc = (j - 1) * PPT + k;
                            if (j >= 1 && pc < 2 * EP_N) {
                                ep_piece(h, 1, pc / EP_N, pc % EP_N);
                                if (pc % EP_N == EP_N - 1) { nh[pc / EP_N] = *(const bf16x8 *)&eoh; nl[pc / EP_N] = *(const bf16x8 *)&eol; }
                            }
                        }
                    }
#pragma unroll
                    for (int c = 0; c < PR; c++)
                        if (c >= j * np2 / NTR && c < (j + 1) * np2 / NTR) issue_piece(h2, q2, slot2, c);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (t == 0) {
#pragma unroll
                    for (int s_ = 0; s_ < NS; s_++) { ah[s_] = nh[s_]; al[s_] = nl[s_]; }
                }
            }
            slot = slot + 1 == 3 ? 0 : slot + 1;
        }
    }
    // ---- the reduce's epilogue
#pragma unroll
    for (int t = 0; t < NTR / 2; t++) {
        f32x4 bi0, bi1, sc0, sc1;
        const unsigned a = aB3 + (unsigned)(t * 32) * 4;
        lds_req128(a, bi0); lds_req128(a + 16, bi1); lds_req128(a + CMID * 4, sc0); lds_req128(a + CMID * 4 + 16, sc1);
        lds_arrive4(bi0, bi1, sc0, sc1);
#pragma unroll
        for (int s_ = 0; s_ < NS; s_++) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                v[e] = fmaxf(__fmaf_rn(accR[s_][2 * t][e], sc0[e], bi0[e]), 0.f);
                v[4 + e] = fmaxf(__fmaf_rn(accR[s_][2 * t + 1][e], sc1[e], bi1[e]), 0.f);
            }
            rmax = rng_see<8>(rmax, v);
            uint4 oh, ol;
            x3_split8(v, oh, ol);
            T *op = dst3 + mrow[s_] * (2 * CMID) + t * 64 + kq * 8;
            *(uint4 *)op = oh;
            *(uint4 *)(op + 32) = ol;
        }
    }
    rng_flush(p.range, rmax);
}

// ---------------------------------------------------------------------------------------------
// k_tail_x3 (split precision): the same three layers -- 2x bilinear upsampling of the 64-channel map, conv7 (3x3, 64 -> 32,
// ReLU) and the 1x1 head -- for the RCV_F16X3 engine.  Unfused they move 17 GB per 72 (frame, network) items through HBM
// (the 480x640x64 upsampled map alone is 79 MB per item in hi / lo storage); fused only the 240x320x64 source is read.
// Four waves on a 4 x 32 tile: wave = (pixel half of the tile, 16-channel half of conv7's outputs); its 18 x (hi, lo) weight
// fragments stay in registers (144 VGPRs) for the whole strip, the A fragments (hi and lo of a 32-channel slice) come from the
// upsampled patch U in LDS: 8 reads per 12 MFMAs and wave.  The kernel's phases are serial inside a workgroup -- building U is
// ~400 VALU instructions per (cell, 8 channels), as long as the tile's MFMAs -- so the tile is sized for TWO workgroups per CU
// (77 KB of LDS each) and the head runs on all four waves (thread = (pixel, plane)).  Measured at 153 items: 6.2 ms for the
// 8 x 32 tile / 512 threads / one workgroup per CU, 5.8 ms now, for 2.1 ms of MFMAs: the stamps (tools/halo_stamps.py --tail)
// show one wave's build phase and the other's MFMA phase on the same SIMD slowing each other down about as much as running
// one after the other -- a SIMD issues one vector instruction at a time, and the kernel's VALU work (join, blend, split:
// inherent to hi / lo storage) is as long as its matrix work.  Starting the CU's second workgroup half a tile late: no change.  U's pixel pitch is 288 B (256 + 32): conflict-free for ds_read_b128's lane
// groups like k_tail's 160.  K order = k_conv's in split mode: channel slice outermost, taps inside, hh / hl / lh per step; the
// upsampling, conv7's rounding to hi + lo and the head's FMA chain are the unfused kernels' -- identical outputs.
// ---------------------------------------------------------------------------------------------
#ifndef RCV_TX_PIPE
#define RCV_TX_PIPE 0  // 1: conv7's fragments of K step s + 1 requested before the MFMAs of step s (measured: 5.8 -> 6.3-6.4 ms, off)
#endif
constexpr int TX_UP = 288;                                   // bytes per pixel of the upsampled patch
constexpr int TX_TH = 4, TX_TW = 32, TX_UH = TX_TH + 2, TX_UW = TX_TW + 2, TX_SH = TX_TH / 2 + 2, TX_SW = TX_TW / 2 + 2;
constexpr int TX_NPX = TX_TH * TX_TW;                        // output pixels per tile
constexpr int TX_SPIX = TX_SH * TX_SW;                       // source pixels per tile (4 x 18)
constexpr int TX_SPIECES = (TX_SPIX + 3) / 4;                // DMA pieces: 4 pixels x 256 B
constexpr int TX_HP = 144;                                   // bytes per pixel of the conv7 rows: [32 hi | 32 lo] + 16 (bank spread)
constexpr int TX_LDS_MAIN = TX_UH * TX_UW * TX_UP + TX_SPIECES * 1024;
constexpr int TX_LDS = TX_LDS_MAIN + 64 * 4;
static_assert(TX_NPX == 128, "the head maps thread -> (pixel, plane) with 128 pixels per tile");
static_assert(TX_NPX * TX_HP <= TX_UH * TX_UW * TX_UP, "the conv7 rows reuse the upsampled patch's LDS");
static_assert(2 * TX_LDS <= 160 * 1024, "two workgroups per CU");
constexpr int TX_RING_ROWS = 10;                             // interior tile rows per workgroup of the ring pass (two tiles each)

__global__ __launch_bounds__(256, 2) void k_tail_x3(const H16 *src, const H16 *w7, const float *b7, const H16 *w8, const float *b8,
                                                 float *seg, float *radial, int B, int Hin, int Win, long long src_gs, long long w7_gs,
                                                 int b7_gs, long long w8_gs, int b8_gs, int rad_f16, unsigned *range, int ring_only)
{
    RCV_X3_MODE();
    using T = H16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *sU = smem;                                        // [6 x 34][TX_UP] upsampled patch ...
    unsigned char *sH = sU;                                          // ... later [128][TX_HP] conv7 output
    unsigned char *sS = smem + TX_UH * TX_UW * TX_UP;                // [72][256 B] source patch of the NEXT tile by LDS-DMA
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int Ho = Hin * 2, Wo = Win * 2;
    const int img = blockIdx.y, g = blockIdx.z;
    const T *sg = src + (long long)g * src_gs + (long long)img * Hin * Win * 128;
    const int pq = wave & 1, nt = wave >> 1;
    float rmax = 0.f;  // range guard: the largest conv7 activation stored as hi + lo
    // conv7 weights of this wave's 16 output channels: [co][tap][slice][hi 32 | lo 32]
    bf16x8 bw[2][9][2];  // [slice][tap][hi / lo]
    {
        const T *wg = w7 + (long long)g * w7_gs + (long long)(nt * 16 + r16) * 1152;
#pragma unroll
        for (int sl = 0; sl < 2; sl++)
#pragma unroll
            for (int t = 0; t < 9; t++)
#pragma unroll
                for (int h = 0; h < 2; h++) bw[sl][t][h] = *(const bf16x8 *)(wg + (t * 2 + sl) * 64 + h * 32 + kq * 8);
    }
    float bias7[4], sc7[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        bias7[e] = b7[(long long)g * b7_gs + nt * 16 + kq * 4 + e];
        sc7[e] = b7[(long long)g * b7_gs + 32 + nt * 16 + kq * 4 + e];
    }
    float *hw = (float *)(smem + TX_LDS_MAIN);  // head weights [2][32] as fp32 (hi + lo, as k_head forms them)
    if (tid < 64) hw[tid] = (float)w8[(long long)g * w8_gs + tid] + (float)w8[(long long)g * w8_gs + 64 + tid];
    const float hb0 = b8[(long long)g * b8_gs], hb1 = b8[(long long)g * b8_gs + 1];
    const float hs0 = b8[(long long)g * b8_gs + 2], hs1 = b8[(long long)g * b8_gs + 3];
    int ubase[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int m = pq * 64 + i * 16 + r16;
        ubase[i] = (m >> 5) * TX_UW + (m & 31);
    }
    const int src_bytes = Hin * Win * 256;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    constexpr int SPW = (TX_SPIECES + 3) / 4;  // pieces per wave
    int s_a[SPW], s_col[SPW];  // patch row and patch column (or -1) of this lane's pixel of piece k
#pragma unroll
    for (int k = 0; k < SPW; k++) {
        const int pp = (k * 4 + wv) * 4 + (lane >> 4);
        const int a = pp / TX_SW, b = pp - a * TX_SW;
        s_a[k] = a;
        s_col[k] = (k * 4 + wv < TX_SPIECES && pp < TX_SPIX) ? b : -1;
    }
// source patch of the tile whose first output row / column is ho0_ / wo0_: rows ho0_ / 2 - 1 .., columns wo0_ / 2 - 1 .., clamped
#define RCV_TX_SOURCE(ho0_, wo0_)                                                                              \
    _Pragma("unroll") for (int k = 0; k < SPW; k++)                                                            \
        if (k * 4 + wv < TX_SPIECES)                                                                           \
            tail_dma16(sg, src_bytes, sS + (k * 4 + wv) * 1024,                                                \
                       s_col[k] >= 0 ? min(max((ho0_) / 2 - 1 + s_a[k], 0), Hin - 1) * Win * 256 + (lane & 15) * 16 + \
                                           min(max((wo0_) / 2 - 1 + s_col[k], 0), Win - 1) * 256                \
                                     : (int)0x80000000);
#if defined(RCV_HALO_STAMPS) && defined(RCV_TAIL_STAMPS)
    unsigned hst[16];
    for (int k = 0; k < 16; k++) hst[k] = 0u;
    const bool stamp_blk = blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && blockIdx.z == 0 && (wave == 0 || wave == 3);
    bool stamp_on = false;
#define RCV_TSTAMP(idx) RCV_HSTAMP(idx)
#else
#define RCV_TSTAMP(idx)
#endif
    // The workgroup's tiles.  Normally tile row blockIdx.x, left to right.  ring_only (engine option 13): only the tiles that touch
    // the image border -- the parity-composed tail (k_conv_halo<.., TAILC>) cannot express the outermost ring of output pixels,
    // where the upsampling clamps and conv7 pads with zeros; this pass, launched after it, recomputes the border tiles with the
    // exact three-layer arithmetic (11.5 % of the tiles at 480 x 640).  Workgroup 0 takes the first tile row, 1 the last, and
    // workgroup 2 + j the first and the last tile of TX_RING_ROWS interior rows (a workgroup's start-up -- conv7's weights into
    // registers -- costs about as much as two tiles: one workgroup per interior row made the pass 1.3 instead of 0.7 ms).
    const int ntx = (Wo + TX_TW - 1) / TX_TW, nty = (Ho + TX_TH - 1) / TX_TH;
    const int wgx = blockIdx.x;
    const bool sparse = ring_only && wgx >= 2;
    const int row0 = !ring_only ? wgx : wgx == 0 ? 0 : wgx == 1 ? nty - 1 : 1 + (wgx - 2) * TX_RING_ROWS;
    const int ntile = !sparse ? ntx : 2 * max(0, min(TX_RING_ROWS, nty - 1 - row0));
    auto tile_ho0 = [&](int q_) { return (sparse ? row0 + (q_ >> 1) : row0) * TX_TH; };
    auto tile_wo0 = [&](int q_) { return (sparse ? ((q_ & 1) ? ntx - 1 : 0) : q_) * TX_TW; };
    if (ntile > 0) RCV_TX_SOURCE(tile_ho0(0), tile_wo0(0));
    for (int tq = 0; tq < ntile; tq++) {
        const int ho0 = tile_ho0(tq), wo0 = tile_wo0(tq);
#if defined(RCV_HALO_STAMPS) && defined(RCV_TAIL_STAMPS)
        stamp_on = stamp_blk && tq == 10;
#endif
        RCV_TSTAMP(0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RCV_TSTAMP(1)
        __syncthreads();  // the whole source patch is in sS; every thread is done with the previous tile's conv7 rows in sU
        RCV_TSTAMP(2)
        // ---- upsampled patch: one task = one source cell and one 8-channel chunk -> its 2 x 2 patch pixels (k_upsample2's scheme)
#pragma unroll 1
        for (int v = tid; v < (TX_UH / 2) * (TX_UW / 2) * 8; v += 256) {
            const int c = v & 7, cell = v >> 3, ai = cell / (TX_UW / 2), bi = cell - ai * (TX_UW / 2);
            const int coff = x3_off(c * 8) * 2;  // byte offset of the chunk's hi halves inside a pixel; lo: + 64
            const unsigned char *r0 = sS + (ai * TX_SW + bi) * 256 + coff, *r1 = r0 + TX_SW * 256;
            float f00[8], f01[8], f10[8], f11[8];
            x3_join8(*(const uint4 *)r0, *(const uint4 *)(r0 + 64), f00);
            x3_join8(*(const uint4 *)(r0 + 256), *(const uint4 *)(r0 + 256 + 64), f01);
            x3_join8(*(const uint4 *)r1, *(const uint4 *)(r1 + 64), f10);
            x3_join8(*(const uint4 *)(r1 + 256), *(const uint4 *)(r1 + 256 + 64), f11);
#pragma unroll
            for (int dx = 0; dx < 2; dx++) {
                const int ux = 2 * bi + dx, x = wo0 - 1 + ux;
                int x0, x1;
                float lx, hx;
                up2_src(x, Win, x0, x1, lx, hx);
                f32x2 t0[4], t1[4];
                bilerp8_rows(hx, lx, f00, f01, f10, f11, t0, t1);
#pragma unroll
                for (int dy = 0; dy < 2; dy++) {
                    const int uy = 2 * ai + dy, y = ho0 - 1 + uy;
                    int y0, y1;
                    float ly, hy;
                    up2_src(y, Hin, y0, y1, ly, hy);
                    unsigned char *up = sU + (uy * TX_UW + ux) * TX_UP + coff;
                    if ((unsigned)y < (unsigned)Ho && (unsigned)x < (unsigned)Wo) {
                        float bl[8];
                        bilerp8_col(hy, ly, t0, t1, bl);
                        uint4 oh, ol;
                        x3_split8(bl, oh, ol);
                        *(uint4 *)up = oh;
                        *(uint4 *)(up + 64) = ol;
                    } else {  // conv7's zero padding (rare: a branch of its own, skipped by whole waves inside the image)
                        *(uint4 *)up = make_uint4(0u, 0u, 0u, 0u);
                        *(uint4 *)(up + 64) = make_uint4(0u, 0u, 0u, 0u);
                    }
                }
            }
        }
        RCV_TSTAMP(3)
        __syncthreads();  // U is complete, sS is free
        RCV_TSTAMP(4)
        if (tq + 1 < ntile) RCV_TX_SOURCE(tile_ho0(tq + 1), tile_wo0(tq + 1));
        // ---- conv7: K = 2 channel slices x 9 taps x [32 hi | 32 lo]
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (RCV_TX_PIPE: the fragments of K step s + 1 requested before the MFMAs of step s, two register sets.  The stamps show
        // 500 cycles per step for 192 of MFMAs either way: the step is not waiting for its own reads -- the SIMD's other wave, of
        // the CU's second workgroup, is issuing VALU instructions of its build phase -- and the pipelined form lost 8 %)
        bf16x8 ah[2][4], al[2][4];
#define RCV_TX_FRAGS(st_)                                                                                      \
    {                                                                                                          \
        constexpr int sl_ = (st_) / 9, t_ = (st_) % 9, toff_ = (t_ / 3) * TX_UW + (t_ % 3);                   \
        _Pragma("unroll") for (int i = 0; i < 4; i++) {                                                        \
            ah[(st_) & 1][i] = *(const bf16x8 *)(sU + ubase[i] * TX_UP + kq * 16 + (toff_ * TX_UP + sl_ * 128));      \
            al[(st_) & 1][i] = *(const bf16x8 *)(sU + ubase[i] * TX_UP + kq * 16 + (toff_ * TX_UP + sl_ * 128 + 64)); \
        }                                                                                                      \
    }
#define RCV_TX_STEP(st_)                                                                                       \
    {                                                                                                          \
        if constexpr (RCV_TX_PIPE && (st_) + 1 < 18) RCV_TX_FRAGS((st_) + 1 < 18 ? (st_) + 1 : 0)              \
        if constexpr (!RCV_TX_PIPE) RCV_TX_FRAGS(st_)                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        constexpr int sl_ = (st_) / 9, t_ = (st_) % 9, b_ = (st_) & 1;                                         \
        _Pragma("unroll") for (int i = 0; i < 4; i++) acc[i] = RCV_MFMA16(bw[sl_][t_][0], ah[b_][i], acc[i], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < 4; i++) acc[i] = RCV_MFMA16(bw[sl_][t_][1], ah[b_][i], acc[i], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < 4; i++) acc[i] = RCV_MFMA16(bw[sl_][t_][0], al[b_][i], acc[i], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }
        if constexpr (RCV_TX_PIPE) RCV_TX_FRAGS(0)
        RCV_TX_STEP(0) RCV_TX_STEP(1) RCV_TX_STEP(2) RCV_TX_STEP(3) RCV_TX_STEP(4) RCV_TX_STEP(5)
        RCV_TX_STEP(6) RCV_TX_STEP(7) RCV_TX_STEP(8) RCV_TX_STEP(9) RCV_TX_STEP(10) RCV_TX_STEP(11)
        RCV_TX_STEP(12) RCV_TX_STEP(13) RCV_TX_STEP(14) RCV_TX_STEP(15) RCV_TX_STEP(16) RCV_TX_STEP(17)
#undef RCV_TX_STEP
#undef RCV_TX_FRAGS
        RCV_TSTAMP(5)
        __syncthreads();  // every wave is done reading U: its LDS takes the conv7 rows
        RCV_TSTAMP(6)
        // ---- bias + ReLU, stored as hi + lo like the conv7 activation; operands were swapped, so acc[i][e] is channel
        // nt*16 + kq*4 + e of pixel pq*64 + i*16 + r16
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint2 ph, pl;
            T *vh = (T *)&ph, *vl = (T *)&pl;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float v7 = fmaxf(__fmaf_rn(acc[i][e], sc7[e], bias7[e]), 0.f);
                rmax = fmaxf(rmax, v7);
                x3_split(v7, vh[e], vl[e]);
            }
            unsigned char *hp = sH + (pq * 64 + i * 16 + r16) * TX_HP + (nt * 16 + kq * 4) * 2;
            *(uint2 *)hp = ph;
            *(uint2 *)(hp + 64) = pl;
        }
        RCV_TSTAMP(7)
        __syncthreads();
        RCV_TSTAMP(8)
        // ---- head: thread = pixel, the sequential FMA chain of k_head
        {   // thread = (pixel, output plane): waves 0-1 the seg plane, waves 2-3 the radial plane
            const int px = tid & (TX_NPX - 1), oc = tid >> 7;
            const int ho = ho0 + (px >> 5), wo = wo0 + (px & 31);
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < 32; c += 8) {
                float f[8];
                x3_join8(*(const uint4 *)(sH + px * TX_HP + c * 2), *(const uint4 *)(sH + px * TX_HP + 64 + c * 2), f);
#pragma unroll
                for (int e = 0; e < 8; e++) a = fmaf(f[e], hw[oc * 32 + c + e], a);
            }
            if (ho < Ho && wo < Wo) {
                const long long o = ((long long)(g * B + img) * Ho + ho) * Wo + wo;
                if (oc == 0) seg[o] = __fmaf_rn(a, hs0, hb0);
                else {
                    const float o1 = __fmaf_rn(a, hs1, hb1);
                    if (rad_f16) ((_Float16 *)radial)[o] = (_Float16)o1;  // engine option 8
                    else radial[o] = o1;
                }
            }
        }
        RCV_TSTAMP(9)
#if defined(RCV_HALO_STAMPS) && defined(RCV_TAIL_STAMPS)
        if (stamp_on && lane == 0)
            for (int k = 0; k < 16; k++) g_chain_stamps[wave == 0 ? 0 : 1][k] = hst[k];
#endif
    }
#undef RCV_TX_SOURCE
    rng_flush(range, rmax);
}

// rcv_fcn_range_status: flag[1] = flag[0], atomically cleared on request
__global__ void k_range_fetch(unsigned *flag, int clear) { flag[1] = clear ? atomicExch(flag, 0u) : atomicOr(flag, 0u); }

template <typename T>
__global__ void k_slot_to_f32(const T *src, float *dst, long long n)
{
    RCV_X3_MODE();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        if constexpr (is_x3<T>()) {  // every tensor has a multiple of 32 channels: logical element i -> its group of 32
            const long long o = ((i >> 5) << 6) + (i & 31);
            dst[i] = to_f32(src[o]) + to_f32(src[o + 32]);
        } else dst[i] = to_f32(src[i]);
    }
}

}  // namespace
}  // namespace rcv

using namespace rcv;

// ---------------------------------------------------------------------------------------------
// executor
// ---------------------------------------------------------------------------------------------
struct rcv_fcn {
    std::vector<rcv_op> ops;
    std::vector<char> q6_ok;  // per op: its source tensor has no other reader before it is overwritten (k_q6_pack may rewrite it in place)
    std::vector<int64_t> slot_elems;
    int groups = 1;
    int dtype = RCV_BF16;
    const void *weights = nullptr;
    const float *bias = nullptr;
    bool use_mfma_stem = true;
    bool use_tail = true;  // fused upsample + conv7 + head kernel (bf16)
    bool use_block = true;  // fused 3x3 + expand 1x1 kernel for the 256 -> 256 -> 1024 bottleneck halves (bf16)
    bool use_chain = true;  // ... with the next block's 1x1 reduce chained in where it follows directly
    bool use_stagger = true;  // ... and, for the 256-wide chained block, the staggered schedule of the 1x1 phases
    int use_halo = 1;  // halo-patch kernel for the 3x3 / stride-1 layers it covers (k_conv_halo); 2: also at poor tile fits
    bool profiling = false;
    std::vector<hipEvent_t> events;  // 2 per op when profiling
    bool events_valid = false;
    bool attrs_set = false;
    bool use_dma = true;  // LDS-DMA conv kernel for the shapes it covers (bf16, Cout % 128 == 0, enough tiles)
    bool radial_f16 = false;  // the radial plane leaves the head as float16 (BASELINE.json stress configuration)
    bool use_lat = true;      // latency kernel (k_conv_lat) for launches with few tiles (small batches)
    int use_chain2 = 1;       // split engine: 1 = k_chain2_x3 (r04), 0 = k_chain_x3, 2 = k_chain3_x3 for the 256-wide block
    int use_c2w4 = 0;         // split engine (15, measured experiment, off): k_chain2_x3<128> as four-wave workgroups of 64 pixels, three to a CU instead of
                              // one eight-wave workgroup: bit-identical, 7.94 against 7.26 ms per 8 launches (twice the DMA issue per wave)
    int use_t3 = 0;           // split engine (14, measured experiment, off): the trunk of a bottleneck stage travels between the chain kernels of layer 2
                              // in the 3-byte form (sign / 5-bit exponent / 18-bit mantissa; t3_encode): -20 % of those kernels' bytes, maps 5.8e-6 away
                              // -- and 9.5 against 7.2 ms per 8 launches: the ~140 VALU instructions per slice of encode + decode cost the chain
                              // kernel more than the bytes give back (2: also layer 1, whose 128-register budget the form does not fit)
    int use_tailc = 1;        // split engine (13): (upsampling, conv7, head) as one parity-composed 3x3 on the low-resolution tensor with the head in
                              // its register-only epilogue (+ the border tiles by k_tail_x3): r06 4.2 + 0.75 ms per forward of 53 x 3 against
                              // k_tail_x3's 6.2.  1: where an image has >= 64 rectangles (a rule of the image size alone: a frame's maps do not
                              // depend on the batch it travels in); 2: wherever the halo tiling applies (tests); 0: k_tail_x3
    int use_splitk = 0;       // latency option (12): split-K in k_conv_lat + k_splitk_finish for launches with few tiles (one or a few frames)
    float *splitk_ws = nullptr;  // ... its partial sums (allocated when the option is switched on, never on the hot path)
    size_t splitk_bytes = 0;
    int use_q6 = 1;           // split engine: fp6 cross terms (k_conv_halo<.., true>) for the ops whose weights come with a q6 copy (rcv_op::wq_off)
    // range guard of the half-precision engines: one sticky flag word on the device (kernels OR bits into it), mirrored into
    // pinned host memory by a 4-byte copy at the end of every forward
    unsigned *range_dev = nullptr;
    volatile unsigned *range_host = nullptr;
};

namespace {


size_t esize(int dtype) { return dtype == RCV_F32 || dtype == RCV_F16X3 ? 4 : 2; }  // (split storage: two halves per element)

void slot_offsets(const rcv_fcn *h, int B, std::vector<size_t> &off, size_t &total)
{
    off.resize(h->slot_elems.size());
    size_t o = 0;
    for (size_t i = 0; i < h->slot_elems.size(); i++) {
        off[i] = o;
        o += align_up((size_t)h->slot_elems[i] * (size_t)h->groups * (size_t)B * esize(h->dtype), 256);
    }
    total = o;
}

template <typename T, int BM, int BN>
int launch_conv_variant(const ConvParams &p, int groups, hipStream_t st, bool set_attr)
{
    const size_t lds = (size_t)(BM + BN) * 128 * 2;
    if (set_attr)
        RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv<T, BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid(p.Cout / BN, (p.M + BM - 1) / BM, groups);
    hipLaunchKernelGGL((k_conv<T, BM, BN>), grid, dim3(256), lds, st, p);
    RCV_LAUNCH_CHECK();
    return RCV_OK;
}

int launch_conv_dma(const ConvParams &p, int groups, hipStream_t st, bool set_attr)
{
    const size_t lds = (size_t)DMA_STAGES * DMA_STAGE_BYTES;
    if (set_attr)
        RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_dma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid(p.Cout / DMA_BN, (p.M + DMA_BM - 1) / DMA_BM, groups);
    hipLaunchKernelGGL(k_conv_dma, grid, dim3(512), lds, st, p);
    RCV_LAUNCH_CHECK();
    return RCV_OK;
}

// k_conv_halo where it applies: 3x3 / stride 1 / pad 1, 16-bit operands, a pixel rectangle of <= 256 pixels -- or of 320 for
// Cout % 128 == 0 -- whose halo'd patch fits the DMA pieces and that covers the image with little waste.  Among the rectangles
// that do, the one with the least modelled time: rounds of 256 workgroups x rows per tile / the kernel's matrix-core rate at
// that tile height (r03 counters: 0.73 busy with 256 rows, 0.83 with 320 -- 60 MFMAs per phase pair instead of 48 against the
// same barriers; BN = 64: 24 MFMAs per step with 256 rows, 36 with 384).  `units` = workgroups per tile position (images x
// networks x output-channel tiles).
bool halo_tile(int H, int W, int mode, int bn, long long units, int &th, int &tw)
{
    const int cand[13][2] = {{8, 32}, {6, 40}, {12, 20}, {16, 16}, {15, 20}, {8, 40}, {20, 16}, {10, 32}, {16, 20}, {12, 32}, {16, 24}, {8, 48}, {24, 16}};
    const int tall_px = bn == 128 ? 320 : 384;  // pixels of the tall tile (five / six 16-row tiles per wave row)
    const double thr = mode == 2 ? 0.3 : 0.9;  // mode 2 (tests): also where rectangles hang over the image edge a lot
    double best = 0.0;
    bool found = false;
    for (auto &c : cand) {
        const int px = c[0] * c[1];
        if ((c[0] + 2) * (c[1] + 2) > halo_patch_rows(bn) || px > tall_px) continue;
        const bool tall = px > 256;
        const double rows = tall ? (double)tall_px : 256.0;
        const long long tiles = (long long)((H + c[0] - 1) / c[0]) * ((W + c[1] - 1) / c[1]);
        const double util = (double)H * W / ((double)tiles * rows);
        if (util < thr) continue;
        const double rounds = (double)((tiles * units + 255) / 256);
        // matrix-core busy fractions of the four forms (r03 counters; the 384-row form of BN = 64 estimated from its step)
        const double rate = bn == 128 ? (tall ? 0.83 : 0.73) : (tall ? 0.62 : 0.52);
        const double cost = rounds * rows / rate;
        if (!found || cost < best) { best = cost; th = c[0]; tw = c[1]; found = true; }
    }
    return found;
}

// the composed tail's workgroups are persistent -- one per CU (160 KB of LDS each), or one per work item when there are fewer
dim3 halo_persistent_grid(long long items)
{
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount : 256;
        n_cu -= n_cu % 8;  // (the item ranges are dealt per XCD)
        if (n_cu < 8) n_cu = 8;
    }
    return dim3((unsigned)(items < n_cu ? items : n_cu), 1, 1);
}

int launch_conv_halo(const ConvParams &p, int B, int groups, int mode, hipStream_t st)
{
    HaloParams q;
    memset(&q, 0, sizeof(q));
    const int bn = p.Cout % 128 == 0 ? 128 : 64;
    if (p.q6 && (bn != 128 || p.src1 || !kX3)) return RCV_E_UNSUPPORTED;
    if (!halo_tile(p.Hout, p.Wout, mode, bn, (long long)B * groups * (p.Cout / bn), q.th, q.tw)) return RCV_E_UNSUPPORTED;
    q.src0 = p.src0; q.src1 = p.src1; q.w = p.w; q.dst = p.dst; q.bias = p.bias;
    q.src0_gs = p.src0_gs; q.src1_gs = p.src1_gs; q.dst_gs = p.dst_gs; q.w_gs = p.w_gs; q.bias_gs = p.bias_gs;
    q.C0 = p.C0; q.C1 = p.C1; q.Cout = p.Cout; q.H = p.Hout; q.W = p.Wout; q.n_img = B; q.relu = p.relu; q.K = p.K;
    q.tiles_y = (q.H + q.th - 1) / q.th; q.tiles_x = (q.W + q.tw - 1) / q.tw;
    q.range = p.range;
    q.c1_shift = halo_c1_shift(p.C0, p.C1);
    if (q.c1_shift == (int)0x80000000) return RCV_E_UNSUPPORTED;
    q.groups = groups;
    dim3 grid(p.Cout / bn, (unsigned)(B * q.tiles_y * q.tiles_x), groups);
    if (kX3 && (bn == 64 || RCV_HALO_PERSIST_ALL)) grid = halo_persistent_grid((long long)(p.Cout / bn) * B * q.tiles_y * q.tiles_x * groups);
    if (p.q6) {  // fp6 cross terms (option 11): source and weights in the q6 row layout
        if (q.th * q.tw > 256) hipLaunchKernelGGL((k_conv_halo<128, 5, true>), grid, dim3(512), halo_lds(128), st, q);
        else hipLaunchKernelGGL((k_conv_halo<128, 4, true>), grid, dim3(512), halo_lds(128), st, q);
        RCV_LAUNCH_CHECK();
        return RCV_OK;
    }
    if (q.th * q.tw > 256) {
        if (bn == 128) hipLaunchKernelGGL((k_conv_halo<128, 5>), grid, dim3(512), halo_lds(128), st, q);
        else hipLaunchKernelGGL((k_conv_halo<64, 6>), grid, dim3(512), halo_lds(64), st, q);
    } else if (bn == 128) hipLaunchKernelGGL((k_conv_halo<128, 4>), grid, dim3(512), halo_lds(128), st, q);
    else hipLaunchKernelGGL((k_conv_halo<64, 4>), grid, dim3(512), halo_lds(64), st, q);
    RCV_LAUNCH_CHECK();
    return RCV_OK;
}

int launch_conv_lat(const ConvParams &p, int groups, hipStream_t st, float *splitk_ws = nullptr, size_t splitk_bytes = 0)
{
    // Split-K (engine option 12, opt-in): with one or a few frames in flight a launch has far fewer tiles than the chip has CUs and
    // its time is its chain of dependent K steps (~0.37 us each).  The steps are dealt over S workgroups per tile -- S from the layer
    // alone (tiles, steps), never from timing -- that leave raw fp32 sums; k_splitk_finish adds them in split order and applies the
    // epilogue.  Worth a second launch (~4.5 us) only where it removes more than ~16 steps from the chain.
    const long long tiles = (long long)(p.Cout / LAT_BN) * ((p.M + LAT_BM - 1) / LAT_BM) * groups;
    const int nk = p.K / 64;
    int S = 1;
    if (splitk_ws && tiles > 0) {
        S = (int)std::min<long long>(256 / tiles, nk / 8);   // fill the chip once; at least 8 steps per split
        if (S < 2 || nk - nk / S < 16) S = 1;
        while (S > 1 && (size_t)groups * S * p.M * p.Cout * sizeof(float) > splitk_bytes) S--;
    }
    ConvParams q = p;
    q.splits = S > 1 ? S : 0;
    q.partial = S > 1 ? splitk_ws : nullptr;
    dim3 grid(p.Cout / LAT_BN, (p.M + LAT_BM - 1) / LAT_BM, groups * (S > 1 ? S : 1));
    hipLaunchKernelGGL(k_conv_lat, grid, dim3(512), LAT_STAGES * LAT_STAGE_BYTES, st, q);
    RCV_LAUNCH_CHECK();
    if (S > 1) {
        const long long n = (long long)p.M * (p.Cout / 8);
        hipLaunchKernelGGL(k_splitk_finish, dim3((unsigned)((n + 255) / 256), groups), dim3(256), 0, st, q);
        RCV_LAUNCH_CHECK();
    }
    return RCV_OK;
}

template <typename T>
int launch_conv(const ConvParams &p, int groups, hipStream_t st, bool set_attr, bool allow_dma, bool allow_lat = false, float *splitk_ws = nullptr,
                size_t splitk_bytes = 0)
{
    constexpr int KE = Traits<T>::KE;
    if (p.C0 % KE || p.C1 % KE || p.Cout % 32 || p.K % KE) return RCV_E_UNSUPPORTED;
    // few output tiles (a frame or two in flight) and a long K: the layer's time is its chain of K steps -> latency kernel
    // (measured at batch 1: it pays from K = 1024 on; short-K layers keep k_conv, whose tiles are cheaper to set up)
    if (sizeof(T) == 2 && allow_lat && allow_dma && p.Cout % LAT_BN == 0 && p.K >= 1024 * (kX3 ? 2 : 1) &&
        (long long)((p.M + 127) / 128) * ((p.Cout + 127) / 128) * groups <= 128)
        return launch_conv_lat(p, groups, st, splitk_ws, splitk_bytes);
    if (sizeof(T) == 2 && allow_dma && p.Cout % DMA_BN == 0 &&
        (long long)((p.M + DMA_BM - 1) / DMA_BM) * (p.Cout / DMA_BN) * groups >= 192)
        return launch_conv_dma(p, groups, st, set_attr);
    const int BN = p.Cout % 128 == 0 ? 128 : (p.Cout % 64 == 0 ? 64 : 32);
    // enough workgroups to cover the 256 CUs at least twice, otherwise halve the M tile
    const long long blocks128 = (long long)((p.M + 127) / 128) * (p.Cout / BN) * groups;
    const bool small = blocks128 < 512;
    if (set_attr) {  // first call: register every variant's LDS size once
        int rc;
        ConvParams q = p;
        (void)q;
        rc = RCV_OK;
        (void)rc;
    }
    if (BN == 128) return small ? launch_conv_variant<T, 64, 128>(p, groups, st, set_attr) : launch_conv_variant<T, 128, 128>(p, groups, st, set_attr);
    if (BN == 64) return small ? launch_conv_variant<T, 64, 64>(p, groups, st, set_attr) : launch_conv_variant<T, 128, 64>(p, groups, st, set_attr);
    return small ? launch_conv_variant<T, 64, 32>(p, groups, st, set_attr) : launch_conv_variant<T, 128, 32>(p, groups, st, set_attr);
}

// fp6 cross terms (engine option 11) apply to op i: it comes with a q6 copy of its weights and its source feeds nothing else
bool q6_active(const rcv_fcn *h, size_t i)
{
    const rcv_op &op = h->ops[i];
    return kX3 && h->dtype == RCV_F16X3 && h->use_q6 && op.kind == RCV_OP_CONV && op.wq_off > 0 && op.src1 < 0 && op.res < 0 &&
           op.cout % 128 == 0 && op.c0 % 32 == 0 && h->q6_ok[i];
}

template <typename T>
int run_op(rcv_fcn *h, const rcv_op &op, const float *x, int B, float *seg, float *radial, char *ws,
           const std::vector<size_t> &off, hipStream_t st, bool src_is_q6 = false)
{
    const int G = h->groups;
    constexpr int ES = is_x3<T>() ? 2 : 1;  // elements of T per logical element (split storage: hi and lo halves)
    auto slot = [&](int s) -> T * { return s >= 0 ? (T *)(ws + off[s]) : nullptr; };
    auto gstride = [&](int s) -> long long { return s >= 0 ? (long long)h->slot_elems[s] * B * ES : 0; };
    const T *w = (const T *)h->weights + op.w_off;
    const float *bias = h->bias + op.b_off;
    switch (op.kind) {
    case RCV_OP_STEM: {
        if (op.cout != 64 || op.ksize != 7 || op.stride != 2 || op.pad != 3) return RCV_E_UNSUPPORTED;
        if constexpr (sizeof(T) == 2) {
            if (h->use_mfma_stem) {  // MFMA stem (16-bit and split engines); option 2 = 0 selects the direct fp32-input kernel
                if (!h->attrs_set)
                    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_stem_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, STEM_LDS));
                const int tiles_x = (op.wout + STEM_TW - 1) / STEM_TW, runs_x = (tiles_x + STEM_RUN - 1) / STEM_RUN;
                dim3 grid(((op.hout + STEM_TH - 1) / STEM_TH) * runs_x, B);
                for (int g0 = 0; g0 < G; g0 += STEM_MAXG) {  // the weights of up to STEM_MAXG networks share the LDS
                    const int gn = std::min(STEM_MAXG, G - g0);
                    hipLaunchKernelGGL(k_stem_mfma, grid, dim3(256), STEM_LDS, st, x, (const H16 *)w + (long long)g0 * 64 * 147 * ES, bias + g0 * 64 * ES,
                                       (H16 *)slot(op.dst) + (long long)g0 * gstride(op.dst), op.hin, op.win, op.hout, op.wout, gn,
                                       (long long)64 * 147 * ES, 64 * ES, gstride(op.dst), h->range_dev);
                    RCV_LAUNCH_CHECK();
                }
                return RCV_OK;
            }
        }
        dim3 grid(((op.hout + 15) / 16) * ((op.wout + 15) / 16), B, G);
        hipLaunchKernelGGL((k_stem<T>), grid, dim3(256), 0, st, x, w, bias, slot(op.dst), B, op.hin, op.win, op.hout,
                           op.wout, (long long)64 * 147 * ES, 64 * ES, gstride(op.dst), h->range_dev);
        RCV_LAUNCH_CHECK();
        return RCV_OK;
    }
    case RCV_OP_MAXPOOL: {
        const long long tv = (long long)B * op.hout * op.wout * (op.c0 / Traits<T>::EPV);
        dim3 grid((unsigned)std::min<long long>((tv + 255) / 256, 65535), G);
        hipLaunchKernelGGL((k_maxpool<T>), grid, dim3(256), 0, st, slot(op.src0), slot(op.dst), op.c0, op.hin, op.win,
                           op.hout, op.wout, tv, gstride(op.src0), gstride(op.dst));
        RCV_LAUNCH_CHECK();
        return RCV_OK;
    }
    case RCV_OP_UPSAMPLE2: {
        const long long tv = (long long)B * (op.hin + 1) * (op.win + 1) * (op.c0 / Traits<T>::EPV);  // source cells x chunks
        dim3 grid((unsigned)std::min<long long>((tv + 255) / 256, 65535), G);
        hipLaunchKernelGGL((k_upsample2<T>), grid, dim3(256), 0, st, slot(op.src0), slot(op.dst), op.c0, op.hin, op.win,
                           tv, gstride(op.src0), gstride(op.dst));
        RCV_LAUNCH_CHECK();
        return RCV_OK;
    }
    case RCV_OP_HEAD: {
        if (op.cout != 2 || op.c0 % Traits<T>::EPV) return RCV_E_UNSUPPORTED;
        const long long npix = (long long)B * op.hin * op.win;
        dim3 grid((unsigned)std::min<long long>((npix + 255) / 256, 65535), G);
        hipLaunchKernelGGL((k_head<T>), grid, dim3(256), 0, st, slot(op.src0), w, bias, seg, radial, op.c0, npix,
                           gstride(op.src0), (long long)2 * op.c0 * ES, 2 * ES, h->radial_f16 ? 1 : 0);
        RCV_LAUNCH_CHECK();
        return RCV_OK;
    }
    case RCV_OP_CONV: {
        ConvParams p;
        memset(&p, 0, sizeof(p));
        p.src0 = slot(op.src0); p.src1 = slot(op.src1); p.res = slot(op.res); p.dst = slot(op.dst);
        p.w = w; p.bias = bias;
        p.src0_gs = gstride(op.src0); p.src1_gs = gstride(op.src1); p.res_gs = gstride(op.res); p.dst_gs = gstride(op.dst);
        // split storage: the loaders see 16-bit tensors with twice the channels ([32 hi | 32 lo] per group of 32); Cout stays
        // the real channel count (tile indexing, bias) and the epilogues place hi / lo themselves
        p.C0 = op.c0 * ES; p.C1 = op.src1 >= 0 ? op.c1 * ES : 0; p.Cout = op.cout;
        p.Hin = op.hin; p.Win = op.win; p.Hout = op.hout; p.Wout = op.wout;
        p.KS = op.ksize; p.stride = op.stride; p.pad = op.pad; p.relu = op.relu;
        p.M = B * op.hout * op.wout;
        p.K = op.ksize * op.ksize * (p.C0 + p.C1);
        p.w_gs = (long long)op.cout * p.K;
        p.bias_gs = op.cout * ES;  // (split: [bias | scale] per group)
        p.range = h->range_dev;
        if constexpr (is_x3<T>()) {
            // fp6 cross terms (option 11) for an op that comes with a q6 copy of its weights: its source tensor is rewritten in place
            // into the q6 row layout (nothing else reads it: q6_ok), whichever kernel then runs the op -- a frame's maps do not
            // depend on the batch it travels in
            if (q6_active(h, &op - h->ops.data())) {
                if (!src_is_q6) {  // (the fused producer k_chain2_x3 writes q6 rows itself)
                    const long long n_rows = (long long)G * B * op.hin * op.win * (op.c0 / 32);
                    hipLaunchKernelGGL(k_q6_pack, dim3((unsigned)std::min<long long>((n_rows + 255) / 256, 256 * 8)), dim3(256), 0, st, (H16 *)p.src0, n_rows);
                    RCV_LAUNCH_CHECK();
                }
                p.w = (const T *)h->weights + op.wq_off;
                p.q6 = 1;
            }
        }
        // the LDS-DMA kernels address a tensor through a buffer descriptor with 32-bit byte offsets: only below 2 GiB per group
        auto fits32 = [&](int s) { return s < 0 || (long long)h->slot_elems[s] * B * (long long)sizeof(T) * ES < 0x7fffffffLL; };
        const bool dma_ok = h->use_dma && fits32(op.src0) && fits32(op.src1) && fits32(op.res) && fits32(op.dst);
        if constexpr (sizeof(T) == 2) {
            int th, tw;
            const int hbn = p.Cout % 128 == 0 ? 128 : 64;
            if (h->use_halo && dma_ok && op.ksize == 3 && op.stride == 1 && op.pad == 1 && op.res < 0 && p.Cout % 64 == 0 &&
                p.C0 % 64 == 0 && p.C1 % 64 == 0 && op.hin == op.hout && op.win == op.wout && halo_c1_shift(p.C0, p.C1) != (int)0x80000000 &&
                halo_tile(op.hout, op.wout, h->use_halo, hbn, (long long)B * G * (p.Cout / hbn), th, tw) &&
                (long long)B * ((op.hout + th - 1) / th) * ((op.wout + tw - 1) / tw) * (p.Cout / hbn) * G >= 192)
                return launch_conv_halo(p, B, G, h->use_halo, st);
        }
        // (q6 ops outside the halo kernel's reach: the plain kernel, same arithmetic -- the DMA / latency kernels have no q6 form)
        return launch_conv<T>(p, G, st, !h->attrs_set, dma_ok && !p.q6, h->use_lat && !p.q6, h->use_splitk ? h->splitk_ws : nullptr, h->splitk_bytes);
    }
    default: return RCV_E_UNSUPPORTED;
    }
}

// The kernels with more than 64 KB of dynamic LDS need their limit raised once per process.  Done for every variant
// up front (rcv_fcn_create): which variant a layer uses depends on the batch size of the call, not only on the engine.
template <typename T, int BM, int BN>
int register_conv_variant()
{
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv<T, BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (BM + BN) * 128 * 2));
    return RCV_OK;
}
template <typename T>
int register_conv_variants()
{
    int rc;
    if ((rc = register_conv_variant<T, 128, 128>()) != RCV_OK) return rc;
    if ((rc = register_conv_variant<T, 64, 128>()) != RCV_OK) return rc;
    if ((rc = register_conv_variant<T, 128, 64>()) != RCV_OK) return rc;
    if ((rc = register_conv_variant<T, 64, 64>()) != RCV_OK) return rc;
    if ((rc = register_conv_variant<T, 128, 32>()) != RCV_OK) return rc;
    return register_conv_variant<T, 64, 32>();
}
int register_kernel_attributes()
{
    static bool done = false;
    if (done) return RCV_OK;
    int rc;
    if ((rc = register_conv_variants<float>()) != RCV_OK) return rc;
    if ((rc = register_conv_variants<H16>()) != RCV_OK) return rc;
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_dma, hipFuncAttributeMaxDynamicSharedMemorySize, DMA_STAGES * DMA_STAGE_BYTES));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_lat, hipFuncAttributeMaxDynamicSharedMemorySize, LAT_STAGES * LAT_STAGE_BYTES));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_halo<128, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, halo_lds(128)));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_halo<128, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, halo_lds(128)));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_halo<64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, halo_lds(64)));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_halo<64, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, halo_lds(64)));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_halo<128, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, halo_lds(128)));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_halo<128, 5, true>, hipFuncAttributeMaxDynamicSharedMemorySize, halo_lds(128)));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_halo<128, 4, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, halo_lds(128)));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_halo<128, 5, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, halo_lds(128)));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_stem_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, STEM_LDS));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_tail, hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_tail_x3, hipFuncAttributeMaxDynamicSharedMemorySize, TX_LDS));
    if (getenv("RCV_DEBUG_OCCUPANCY")) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_tail_x3, 256, TX_LDS);
        fprintf(stderr, "k_tail_x3: %d workgroups per CU (LDS %d B)\n", nb, TX_LDS);
    }
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_chain_x3<64>, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_chain_x3<128>, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_chain_x3<256>, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_chain2_x3<64>, hipFuncAttributeMaxDynamicSharedMemorySize, c2_lds(64)));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_chain2_x3<128>, hipFuncAttributeMaxDynamicSharedMemorySize, c2_lds(128)));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_chain2_x3<128, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, c2_lds(128)));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_chain2_x3<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, c2_lds(64)));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_chain2_x3<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, c2_lds(128)));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_chain2_x3<256>, hipFuncAttributeMaxDynamicSharedMemorySize, c2_lds(256)));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_chain3_x3<256>, hipFuncAttributeMaxDynamicSharedMemorySize, c2_lds(256)));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_block<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, blk_lds(64)));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_block<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, blk_lds(64)));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_block<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, blk_lds(128)));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_block<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, blk_lds(128)));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_block<256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, blk_lds(256)));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_block<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, blk_lds(256)));
    RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_block<256, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, blk_lds(256)));
    done = true;
    return RCV_OK;
}

template <typename T>
int forward_impl(rcv_fcn *h, const float *x, int B, float *seg, float *radial, char *ws, hipStream_t st)
{
    std::vector<size_t> off;
    size_t total;
    slot_offsets(h, B, off, total);
    if (h->profiling && h->events.size() != 2 * h->ops.size()) {
        for (auto e : h->events) (void)hipEventDestroy(e);
        h->events.resize(2 * h->ops.size());
        for (auto &e : h->events) RCV_HIP_CHECK(hipEventCreate(&e));
    }
    std::vector<char> src_q6(h->ops.size() + 3, 0);  // per op: its source was written as q6 rows by its (fused) producer
    std::vector<char> res_t3(h->ops.size() + 3, 0);  // per op: its residual was written in the 3-byte form by the chain kernel before it
    // split engine: ops i, i + 1 = a 1x1 expand (+ residual, ReLU) and the next block's 1x1 reduce (ReLU) that run as ONE chain kernel
    // (option 4) -- when the launch has enough 128-pixel tiles to fill the chip (bit-identical to the two kernels, so the choice is free)
    auto chain_at = [&](size_t i) -> bool {
        if (!(kX3 && sizeof(T) == 2 && h->use_block && i + 1 < h->ops.size() && h->ops[i].kind == RCV_OP_CONV && h->ops[i + 1].kind == RCV_OP_CONV))
            return false;
        const rcv_op &a = h->ops[i], &b = h->ops[i + 1];
        const int cmid = a.c0;
        const long long M = (long long)B * a.hout * a.wout;
        return a.ksize == 1 && a.stride == 1 && a.pad == 0 && a.src1 < 0 && a.res >= 0 && a.relu && a.cout == 4 * cmid &&
               (cmid == 64 || cmid == 128 || cmid == 256) && b.ksize == 1 && b.stride == 1 && b.pad == 0 && b.src0 == a.dst &&
               b.src1 < 0 && b.res < 0 && b.relu && b.c0 == 4 * cmid && b.cout == cmid && b.hout == a.hout && b.wout == a.wout &&
               b.dst != a.src0 && b.dst != a.res && b.dst != a.dst && a.dst != a.src0 && a.dst != a.res &&
               (M + CH_BM - 1) / CH_BM * h->groups >= 192 && M * 4 * cmid * 4 < 0x7fffffffLL;
    };
    // The block output of chain launch i (ops i, i + 1) may travel in the 3-byte form (option 14) iff the only other reader of that
    // tensor is the RESIDUAL of a later chain launch of the same width (CMID <= 128: the HBM-bound forms): -> that launch's index, or 0
    auto t3_reader = [&](size_t i) -> size_t {
        if (!(h->use_t3 && h->use_chain2 == 1 && (h->ops[i].c0 == 128 || (h->use_t3 == 2 && h->ops[i].c0 == 64)))) return 0;  // (64: A-B only, see use_t3)
        const int s = h->ops[i].dst;
        size_t reader = 0;
        for (size_t j = i + 2; j < h->ops.size(); j++) {
            const rcv_op &o = h->ops[j];
            const bool reads = o.src0 == s || o.src1 == s || o.res == s;
            if (reads) {
                if (reader || o.res != s || o.src0 == s || o.src1 == s || !chain_at(j) || o.c0 != h->ops[i].c0) return 0;
                reader = j;
            }
            if (o.dst == s) break;  // (the slot takes another tensor from here on)
        }
        return reader;
    };
    for (size_t i = 0; i < h->ops.size(); i++) {
        if constexpr (sizeof(T) == 2) {
            // 3x3 (-> 256 channels, ReLU) followed by the 1x1 expand to 1024 channels with residual and ReLU: the second
            // half of a layer-3 bottleneck block as one kernel (option 4)
            if (!kX3 && h->use_block && i + 1 < h->ops.size() && h->ops[i].kind == RCV_OP_CONV && h->ops[i + 1].kind == RCV_OP_CONV &&
                (long long)h->slot_elems[h->ops[i + 1].dst] * B * 2 < 0x7fffffffLL) {  // (the widest tensor it touches; 32-bit DMA offsets)
                const rcv_op &a = h->ops[i], &b = h->ops[i + 1];
                const long long M = (long long)B * a.hout * a.wout;
                const int cmid = a.cout;
                if (a.ksize == 3 && a.pad == 1 && (cmid == 64 || cmid == 128 || cmid == 256) && a.c0 == cmid && a.src1 < 0 && a.res < 0 &&
                    a.relu && b.ksize == 1 && b.stride == 1 && b.pad == 0 && b.c0 == cmid && b.cout == 4 * cmid && b.src0 == a.dst &&
                    b.src1 < 0 && b.res >= 0 && b.relu && b.hout == a.hout && b.wout == a.wout && b.dst != a.src0 && b.dst != b.res &&
                    (M + BLK_BM - 1) / BLK_BM * h->groups >= 192 && M * 4 * cmid * 2 < 0x7fffffffLL) {
                    if (!h->attrs_set) {
                        RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_block<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, blk_lds(64)));
                        RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_block<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, blk_lds(64)));
                        RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_block<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, blk_lds(128)));
                        RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_block<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, blk_lds(128)));
                        RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_block<256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, blk_lds(256)));
                        RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_block<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, blk_lds(256)));
                        RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_block<256, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, blk_lds(256)));
                    }
                    auto gs = [&](int s) -> long long { return (long long)h->slot_elems[s] * B; };
                    BlockParams q;
                    memset(&q, 0, sizeof(q));
                    q.src = ws + off[a.src0]; q.res = ws + off[b.res]; q.dst = ws + off[b.dst];
                    q.w1 = (const H16 *)h->weights + a.w_off; q.w2 = (const H16 *)h->weights + b.w_off;
                    q.b1 = h->bias + a.b_off; q.b2 = h->bias + b.b_off;
                    q.src_gs = gs(a.src0); q.res_gs = gs(b.res); q.dst_gs = gs(b.dst);
                    q.w1_gs = (long long)a.cout * 9 * a.c0; q.w2_gs = (long long)b.cout * b.c0;
                    q.b1_gs = a.cout; q.b2_gs = b.cout;
                    q.C0 = a.c0; q.Hin = a.hin; q.Win = a.win; q.Hout = a.hout; q.Wout = a.wout; q.stride = a.stride; q.M = (int)M;
                    q.range = h->range_dev;
                    // the next block's 1x1 reduce (1024 -> 256, ReLU) reads exactly the tile this kernel produces: chained in
                    bool chain = false;
                    if (h->use_chain && i + 2 < h->ops.size() && h->ops[i + 2].kind == RCV_OP_CONV) {
                        const rcv_op &c = h->ops[i + 2];
                        // (its output must not share a slot with anything the kernel still reads: other tiles are in flight)
                        chain = c.ksize == 1 && c.stride == 1 && c.pad == 0 && c.c0 == 4 * cmid && c.cout == cmid && c.src0 == b.dst &&
                                c.src1 < 0 && c.res < 0 && c.relu && c.hout == b.hout && c.wout == b.wout && c.dst != a.src0 &&
                                c.dst != b.res && c.dst != b.dst;
                        if (chain) {
                            q.w3 = (const H16 *)h->weights + c.w_off; q.b3 = h->bias + c.b_off; q.dst3 = ws + off[c.dst];
                            q.w3_gs = (long long)c.cout * c.c0; q.b3_gs = c.cout; q.dst3_gs = gs(c.dst);
                        }
                    }
                    const int fused = chain ? 3 : 2;
                    if (h->profiling) RCV_HIP_CHECK(hipEventRecord(h->events[2 * i], st));  // time booked on the 3x3 convolution
                    const dim3 grid((unsigned)((M + BLK_BM - 1) / BLK_BM), 1, h->groups);
                    if (cmid == 256) {
                        if (chain && h->use_stagger) hipLaunchKernelGGL((k_conv_block<256, true, true>), grid, dim3(512), blk_lds(256), st, q);
                        else if (chain) hipLaunchKernelGGL((k_conv_block<256, true>), grid, dim3(512), blk_lds(256), st, q);
                        else hipLaunchKernelGGL((k_conv_block<256, false>), grid, dim3(512), blk_lds(256), st, q);
                    } else if (cmid == 128) {
                        if (chain) hipLaunchKernelGGL((k_conv_block<128, true>), grid, dim3(512), blk_lds(128), st, q);
                        else hipLaunchKernelGGL((k_conv_block<128, false>), grid, dim3(512), blk_lds(128), st, q);
                    } else {
                        if (chain) hipLaunchKernelGGL((k_conv_block<64, true>), grid, dim3(512), blk_lds(64), st, q);
                        else hipLaunchKernelGGL((k_conv_block<64, false>), grid, dim3(512), blk_lds(64), st, q);
                    }
                    RCV_LAUNCH_CHECK();
                    if (h->profiling)
                        for (int e = 1; e < 2 * fused; e++) RCV_HIP_CHECK(hipEventRecord(h->events[2 * i + e], st));
                    i += fused - 1;
                    continue;
                }
            }
            // split engine: 1x1 expand (+ residual, ReLU) followed by the next block's 1x1 reduce (ReLU): one kernel (option 4) when
            // the launch has enough 128-pixel tiles to fill the chip (bit-identical to the two kernels, so the choice is free)
            if (chain_at(i)) {
                const rcv_op &a = h->ops[i], &b = h->ops[i + 1];
                const int cmid = a.c0;
                const long long M = (long long)B * a.hout * a.wout;
                {
                    auto gs = [&](int s) -> long long { return (long long)h->slot_elems[s] * B * 2; };
                    BlockParams q;
                    memset(&q, 0, sizeof(q));
                    q.src = ws + off[a.src0]; q.res = ws + off[a.res]; q.dst = ws + off[a.dst]; q.dst3 = ws + off[b.dst];
                    q.w2 = (const H16 *)h->weights + a.w_off; q.w3 = (const H16 *)h->weights + b.w_off;
                    q.b2 = h->bias + a.b_off; q.b3 = h->bias + b.b_off;
                    q.src_gs = gs(a.src0); q.res_gs = gs(a.res); q.dst_gs = gs(a.dst); q.dst3_gs = gs(b.dst);
                    q.w2_gs = (long long)a.cout * a.c0 * 2; q.w3_gs = (long long)b.cout * b.c0 * 2;
                    q.b2_gs = 2 * a.cout; q.b3_gs = 2 * b.cout;
                    q.M = (int)M;
                    q.range = h->range_dev;
                    // the reduce's output feeds the next block's 3x3: where that runs with fp6 cross terms, k_chain2_x3 writes q6 rows
                    if (h->use_chain2 == 1 && i + 2 < h->ops.size() && h->ops[i + 2].src0 == b.dst && q6_active(h, i + 2)) {
                        q.q6_dst3 = 1;
                        src_q6[i + 2] = 1;
                    }
                    // the trunk between two chain kernels of layers 1-2 in the 3-byte form (option 14)
                    q.t3_res = res_t3[i];
                    if (const size_t rd = t3_reader(i)) {
                        q.t3_dst = 1;
                        res_t3[rd] = 1;
                    }
                    if (h->profiling) RCV_HIP_CHECK(hipEventRecord(h->events[2 * i], st));  // time booked on the expand
                    const dim3 grid((unsigned)((M + CH_BM - 1) / CH_BM), 1, h->groups);
                    if (h->use_chain2 == 2 && cmid == 256) {  // (option 10 = 2: four waves x 32 pixels, one wave per SIMD)
                        hipLaunchKernelGGL((k_chain3_x3<256>), grid, dim3(256), c2_lds(256), st, q);
                    } else if (h->use_chain2) {  // (option 10: the r04 form; 0: k_chain_x3, bit-identical output)
                        const bool t3 = q.t3_res || q.t3_dst;  // (a separate instantiation: the [hi | lo] form keeps its registers and its occupancy)
                        if (cmid == 256) hipLaunchKernelGGL((k_chain2_x3<256>), grid, dim3(512), c2_lds(256), st, q);
                        else if (cmid == 128 && t3) hipLaunchKernelGGL((k_chain2_x3<128, true>), grid, dim3(512), c2_lds(128), st, q);
                        else if (cmid == 128 && h->use_c2w4)  // (option 15: three four-wave workgroups per CU instead of one eight-wave one)
                            hipLaunchKernelGGL((k_chain2_x3<128, false, 4>), dim3((unsigned)((M + 63) / 64), 1, h->groups), dim3(256), c2_lds(128), st, q);
                        else if (cmid == 128) hipLaunchKernelGGL((k_chain2_x3<128>), grid, dim3(512), c2_lds(128), st, q);
                        else if (t3) hipLaunchKernelGGL((k_chain2_x3<64, true>), grid, dim3(512), c2_lds(64), st, q);
                        else hipLaunchKernelGGL((k_chain2_x3<64>), grid, dim3(512), c2_lds(64), st, q);
                    } else if (cmid == 256) hipLaunchKernelGGL((k_chain_x3<256>), grid, dim3(512), CH_LDS, st, q);
                    else if (cmid == 128) hipLaunchKernelGGL((k_chain_x3<128>), grid, dim3(512), CH_LDS, st, q);
                    else hipLaunchKernelGGL((k_chain_x3<64>), grid, dim3(512), CH_LDS, st, q);
                    RCV_LAUNCH_CHECK();
                    if (h->profiling)
                        for (int e = 1; e < 4; e++) RCV_HIP_CHECK(hipEventRecord(h->events[2 * i + e], st));
                    i += 1;
                    continue;
                }
            }
            // upsample x2 -> conv7 (3x3, 64 -> 32, ReLU) -> head (1x1, 32 -> 2): one fused kernel (option 3)
            if (h->use_tail && i + 2 < h->ops.size() + 0 && h->ops[i].kind == RCV_OP_UPSAMPLE2 && h->ops[i + 1].kind == RCV_OP_CONV &&
                h->ops[i + 2].kind == RCV_OP_HEAD) {
                const rcv_op &u = h->ops[i], &c = h->ops[i + 1], &d = h->ops[i + 2];
                if (u.c0 == 64 && c.src0 == u.dst && c.src1 < 0 && c.res < 0 && c.c0 == 64 && c.cout == 32 && c.ksize == 3 &&
                    c.stride == 1 && c.pad == 1 && c.relu && d.src0 == c.dst && d.c0 == 32 && d.cout == 2 && u.hout == 2 * u.hin &&
                    u.wout == 2 * u.win && c.hin == u.hout && c.win == u.wout) {
                    if (!h->attrs_set) {
                        RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_tail, hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS));
                        RCV_HIP_CHECK(hipFuncSetAttribute((const void *)k_tail_x3, hipFuncAttributeMaxDynamicSharedMemorySize, TX_LDS));
                    }
                    if (h->profiling) {  // the fused kernel's time is booked on the convolution
                        RCV_HIP_CHECK(hipEventRecord(h->events[2 * i], st));
                        RCV_HIP_CHECK(hipEventRecord(h->events[2 * i + 1], st));
                        RCV_HIP_CHECK(hipEventRecord(h->events[2 * i + 2], st));
                    }
                    dim3 grid((u.hout + TAIL_TH - 1) / TAIL_TH, B, h->groups);
                    int tth = 0, ttw = 0;
                    const bool tailc = kX3 && h->use_tailc && c.wq_off > 0 && u.c0 == 64 && h->use_halo &&
                                       halo_tile(u.hin, u.win, h->use_halo, 128, (long long)B * h->groups, tth, ttw) &&
                                       (h->use_tailc == 2 || (long long)u.hin * u.win >= 64 * 320);  // (>= 64 rectangles of 320 pixels per image)
                    if (tailc) {
                        // (r05, option 13) conv7 on the x2 upsampling == a 3x3 convolution on the LOW-resolution tensor with 4 x 32
                        // parity-stacked output channels (fcn_engine.compose_upsample_conv3): the halo kernel with replicate padding
                        // and the head in its epilogue; the outermost ring of output pixels, where clamp and zero padding differ
                        // from that, is recomputed by k_tail_x3 on the border tiles.  No upsampled tensor, no VALU patch build in the interior.
                        // The DMA addresses a group's tensor with 32-bit byte offsets: a batch beyond that range (> ~109 frames at 480 x 640)
                        // runs in slices of images, so that the form -- and with it a frame's bits -- does not depend on the batch size.
                        const long long img_elems = h->slot_elems[u.src0];  // elements of one image's tensor per group (16-bit units: x2 for the hi / lo pair)
                        int max_imgs = (int)std::max<long long>(1, 0x7fffffffLL / (img_elems * 4) - 1);
                        if (const char *e = getenv("RCV_TAILC_MAX_IMAGES")) max_imgs = std::max(1, std::min(max_imgs, atoi(e)));  // (test hook: small slices)
                        for (int b0 = 0; b0 < B; b0 += max_imgs) {
                            const int nb = std::min(max_imgs, B - b0);
                            HaloParams q;
                            memset(&q, 0, sizeof(q));
                            q.th = tth; q.tw = ttw;
                            q.src0 = ws + off[u.src0] + (size_t)b0 * img_elems * 4; q.src0_gs = img_elems * B * 2;
                            q.w = (const H16 *)h->weights + c.wq_off; q.w_gs = (long long)128 * 9 * 128;
                            q.bias = h->bias + c.b_off + (long long)h->groups * 64; q.bias_gs = 256;
                            q.C0 = 128; q.C1 = 0; q.Cout = 128; q.H = u.hin; q.W = u.win; q.n_img = nb; q.relu = 1; q.K = 9 * 128;
                            q.tiles_y = (q.H + q.th - 1) / q.th; q.tiles_x = (q.W + q.tw - 1) / q.tw;
                            q.range = h->range_dev;
                            q.w8 = (const H16 *)h->weights + d.w_off; q.w8_gs = 128; q.b8 = h->bias + d.b_off; q.b8_gs = 4;
                            q.seg = seg; q.radial = radial; q.rad_f16 = h->radial_f16 ? 1 : 0;
                            q.groups = h->groups; q.out_b = B; q.out_img0 = b0;
                            const dim3 hgrid = halo_persistent_grid((long long)nb * q.tiles_y * q.tiles_x * h->groups);  // one workgroup per CU
                            if (q.th * q.tw > 256) hipLaunchKernelGGL((k_conv_halo<128, 5, false, true>), hgrid, dim3(512), halo_lds(128), st, q);
                            else hipLaunchKernelGGL((k_conv_halo<128, 4, false, true>), hgrid, dim3(512), halo_lds(128), st, q);
                            RCV_LAUNCH_CHECK();
                        }
                        // the border tiles again, with the three-layer arithmetic (k_tail_x3, ring_only)
                        const int nty = (u.hout + TX_TH - 1) / TX_TH, ring_wgs = nty < 2 ? 1 : 2 + (nty - 2 + TX_RING_ROWS - 1) / TX_RING_ROWS;
                        hipLaunchKernelGGL(k_tail_x3, dim3(ring_wgs, B, h->groups), dim3(256), TX_LDS, st, (const H16 *)(ws + off[u.src0]),
                                           (const H16 *)h->weights + c.w_off, h->bias + c.b_off, (const H16 *)h->weights + d.w_off,
                                           h->bias + d.b_off, seg, radial, B, u.hin, u.win, (long long)h->slot_elems[u.src0] * B * 2,
                                           (long long)32 * 576 * 2, 64, (long long)128, 4, h->radial_f16 ? 1 : 0, h->range_dev, 1);
                        RCV_LAUNCH_CHECK();
                    } else if constexpr (kX3)
                        hipLaunchKernelGGL(k_tail_x3, dim3((u.hout + TX_TH - 1) / TX_TH, B, h->groups), dim3(256), TX_LDS, st, (const H16 *)(ws + off[u.src0]),
                                           (const H16 *)h->weights + c.w_off, h->bias + c.b_off, (const H16 *)h->weights + d.w_off,
                                           h->bias + d.b_off, seg, radial, B, u.hin, u.win, (long long)h->slot_elems[u.src0] * B * 2,
                                           (long long)32 * 576 * 2, 64, (long long)128, 4, h->radial_f16 ? 1 : 0, h->range_dev, 0);
                    else
                        hipLaunchKernelGGL(k_tail, grid, dim3(256), TAIL_LDS, st, (const H16 *)(ws + off[u.src0]),
                                           (const H16 *)h->weights + c.w_off, h->bias + c.b_off, (const H16 *)h->weights + d.w_off,
                                           h->bias + d.b_off, seg, radial, B, u.hin, u.win, (long long)h->slot_elems[u.src0] * B,
                                           (long long)32 * 576, 32, (long long)64, 2, h->radial_f16 ? 1 : 0, h->range_dev);
                    RCV_LAUNCH_CHECK();
                    if (h->profiling) {
                        RCV_HIP_CHECK(hipEventRecord(h->events[2 * i + 3], st));
                        RCV_HIP_CHECK(hipEventRecord(h->events[2 * i + 4], st));
                        RCV_HIP_CHECK(hipEventRecord(h->events[2 * i + 5], st));
                    }
                    i += 2;
                    continue;
                }
            }
        }
        if (h->profiling) RCV_HIP_CHECK(hipEventRecord(h->events[2 * i], st));
        int rc = run_op<T>(h, h->ops[i], x, B, seg, radial, ws, off, st, src_q6[i] != 0);
        if (rc != RCV_OK) return rc;
        if (h->profiling) RCV_HIP_CHECK(hipEventRecord(h->events[2 * i + 1], st));
    }
    h->events_valid = h->profiling;
    h->attrs_set = true;
    return RCV_OK;
}

}  // namespace

// Per-translation-unit entry points of the 16-bit engine: this file's own (bfloat16) and, compiled through fcn_f16.hip, the
// half-precision twin's.  The C ABI below exists once, in the bfloat16 unit, and dispatches on rcv_fcn::dtype.
namespace rcv {
#if defined(RCV_FCN_X3)
#define RCV_TU_SUFFIX(name) name##_x3
#elif defined(RCV_FCN_FP16)
#define RCV_TU_SUFFIX(name) name##_f16
#endif
#ifdef RCV_TU_SUFFIX
int RCV_TU_SUFFIX(fcn_forward)(rcv_fcn *h, const float *x, int B, float *seg, float *radial, char *ws, hipStream_t st)
{
    return forward_impl<H16>(h, x, B, seg, radial, ws, st);
}
int RCV_TU_SUFFIX(fcn_register)() { return register_kernel_attributes(); }
int RCV_TU_SUFFIX(fcn_debug_slot)(const void *src, float *out_dev, long long n, hipStream_t st)
{
    const unsigned blocks = (unsigned)std::min<long long>((n + 255) / 256, 65535);
    hipLaunchKernelGGL((k_slot_to_f32<H16>), dim3(blocks), dim3(256), 0, st, (const H16 *)src, out_dev, n);
    RCV_LAUNCH_CHECK();
    return RCV_OK;
}
#if defined(RCV_FCN_X3)
int fcn_q6_pack_x3(void *rows, long long n_rows, hipStream_t st)
{
    hipLaunchKernelGGL(k_q6_pack, dim3((unsigned)std::min<long long>((n_rows + 255) / 256, 256 * 8)), dim3(256), 0, st, (H16 *)rows, n_rows);
    RCV_LAUNCH_CHECK();
    return RCV_OK;
}
#endif
#if defined(RCV_HALO_STAMPS) && defined(RCV_FCN_X3)
}  // namespace rcv
extern "C" int rcv_debug_halo_stamps(unsigned *out32)
{
    RCV_HIP_CHECK(hipDeviceSynchronize());
    RCV_HIP_CHECK(hipMemcpyFromSymbol(out32, HIP_SYMBOL(rcv::g_halo_stamps), sizeof(unsigned) * 32));
    return RCV_OK;
}
extern "C" int rcv_debug_chain_stamps(unsigned *out128)
{
    RCV_HIP_CHECK(hipDeviceSynchronize());
    RCV_HIP_CHECK(hipMemcpyFromSymbol(out128, HIP_SYMBOL(rcv::g_chain_stamps), sizeof(unsigned) * 128));
    return RCV_OK;
}
namespace rcv {
#endif
#else
int fcn_forward_f16(rcv_fcn *h, const float *x, int B, float *seg, float *radial, char *ws, hipStream_t st);
int fcn_register_f16();
int fcn_debug_slot_f16(const void *src, float *out_dev, long long n, hipStream_t st);
int fcn_forward_x3(rcv_fcn *h, const float *x, int B, float *seg, float *radial, char *ws, hipStream_t st);
int fcn_register_x3();
int fcn_debug_slot_x3(const void *src, float *out_dev, long long n, hipStream_t st);
int fcn_q6_pack_x3(void *rows, long long n_rows, hipStream_t st);
#endif
}  // namespace rcv

#ifndef RCV_TU_SUFFIX
extern "C" {

int rcv_preprocess_rgb(const uint8_t *rgb_dev, int B, int H, int W, float *x_dev, void *stream)
{
    if (!rgb_dev || !x_dev || B <= 0 || H <= 0 || W <= 0) return RCV_E_INVALID;
    const long long total = (long long)B * H * W * 3;
    hipLaunchKernelGGL(k_preprocess, dim3((unsigned)std::min<long long>((total + 255) / 256, 65535)), dim3(256), 0,
                       (hipStream_t)stream, rgb_dev, x_dev, H * W, total);
    RCV_LAUNCH_CHECK();
    return RCV_OK;
}

int rcv_fcn_create(const rcv_op *ops, int n_ops, int n_slots, const int64_t *slot_elems, int groups, int dtype,
                   const void *weights_dev, const float *bias_dev, rcv_fcn **out)
{
    if (!ops || n_ops <= 0 || n_slots <= 0 || !slot_elems || groups <= 0 || !weights_dev || !bias_dev || !out)
        return RCV_E_INVALID;
    if (dtype != RCV_F32 && dtype != RCV_BF16 && dtype != RCV_F16 && dtype != RCV_F16X3) return RCV_E_UNSUPPORTED;
    for (int i = 0; i < n_ops; i++) {
        const rcv_op &o = ops[i];
        const int slots[4] = {o.src0, o.src1, o.res, o.dst};
        for (int s : slots)
            if (s >= n_slots || s < -3) return RCV_E_INVALID;
    }
    {
        int rc = register_kernel_attributes();
        if (rc == RCV_OK && dtype == RCV_F16) rc = rcv::fcn_register_f16();
        if (rc == RCV_OK && dtype == RCV_F16X3) rc = rcv::fcn_register_x3();
        if (rc != RCV_OK) return rc;
    }
    if (dtype == RCV_F16X3) {
        // split storage keeps the channels in groups of [32 hi | 32 lo]: every tensor an op touches needs a multiple of 32
        // channels (the network input and the two-plane head output excepted) -- anything else would be mis-indexed silently
        for (int i = 0; i < n_ops; i++) {
            const rcv_op &o = ops[i];
            const bool in_ok = o.kind == RCV_OP_STEM || (o.c0 % 32 == 0 && (o.src1 < 0 || o.c1 % 32 == 0));
            const bool out_ok = o.kind == RCV_OP_HEAD || o.cout % 32 == 0;
            if (!in_ok || !out_ok) return RCV_E_UNSUPPORTED;
        }
    }
    rcv_fcn *h = new rcv_fcn();
    if (dtype == RCV_F16 || dtype == RCV_F16X3) {  // the range guard's flag word and its pinned mirror (not on the hot path)
        hipError_t e = hipMalloc((void **)&h->range_dev, 256);
        if (e == hipSuccess) e = hipMemset(h->range_dev, 0, 256);
        if (e == hipSuccess) e = hipHostMalloc((void **)&h->range_host, 256, hipHostMallocDefault);
        if (e != hipSuccess) {
            rcv::set_hip_error(e, "range-flag allocation", __FILE__, __LINE__);
            if (h->range_dev) (void)hipFree(h->range_dev);
            delete h;
            return RCV_E_HIP;
        }
        *h->range_host = 0u;
    }
    h->ops.assign(ops, ops + n_ops);
    // q6 (option 11): k_q6_pack rewrites an op's source slot in place, which is only sound while nothing else reads that slot
    // between its producer and its next writer
    h->q6_ok.assign(n_ops, 0);
    for (int i = 0; i < n_ops; i++) {
        const rcv_op &o = ops[i];
        if (o.kind != RCV_OP_CONV || o.wq_off <= 0 || o.src0 < 0 || o.src1 >= 0) continue;
        bool ok = true;
        for (int j = i - 1; j >= 0 && ok; j--) {  // back to the producer: no reader in between
            if (ops[j].dst == o.src0) break;
            ok = ops[j].src0 != o.src0 && ops[j].src1 != o.src0 && ops[j].res != o.src0;
        }
        for (int j = i + 1; j < n_ops && ok; j++) {  // forward to the next writer: no reader either
            if (ops[j].src0 == o.src0 || ops[j].src1 == o.src0 || ops[j].res == o.src0) ok = false;
            if (ops[j].dst == o.src0) break;
        }
        h->q6_ok[i] = ok ? 1 : 0;
    }
    h->slot_elems.assign(slot_elems, slot_elems + n_slots);
    h->groups = groups;
    h->dtype = dtype;
    h->weights = weights_dev;
    h->bias = bias_dev;
    *out = h;
    return RCV_OK;
}

void rcv_fcn_destroy(rcv_fcn *h)
{
    if (!h) return;
    for (auto e : h->events) (void)hipEventDestroy(e);
    if (h->range_dev) (void)hipFree(h->range_dev);
    if (h->range_host) (void)hipHostFree((void *)h->range_host);
    if (h->splitk_ws) (void)hipFree(h->splitk_ws);
    delete h;
}

size_t rcv_fcn_workspace_bytes(const rcv_fcn *h, int B)
{
    if (!h || B <= 0) return 0;
    std::vector<size_t> off;
    size_t total;
    slot_offsets(h, B, off, total);
    return total;
}

int rcv_fcn_forward(rcv_fcn *h, const float *x_dev, int B, float *seg_dev, float *radial_dev, void *ws, size_t ws_bytes,
                    void *stream)
{
    if (!h || !x_dev || !seg_dev || !radial_dev || !ws || B <= 0) return RCV_E_INVALID;
    if (ws_bytes < rcv_fcn_workspace_bytes(h, B)) return RCV_E_WORKSPACE;
    if (h->dtype == RCV_F32) return forward_impl<float>(h, x_dev, B, seg_dev, radial_dev, (char *)ws, (hipStream_t)stream);
    if (h->dtype == RCV_BF16) return forward_impl<H16>(h, x_dev, B, seg_dev, radial_dev, (char *)ws, (hipStream_t)stream);
    // half-precision storage: a forward that already COMPLETED with a value beyond the half range makes every later call fail
    // (sticky, like a HIP error) until rcv_fcn_range_status clears it -- its maps were clipped, not exact
    if (h->range_host && *h->range_host) return RCV_E_RANGE;
    int rc = h->dtype == RCV_F16 ? rcv::fcn_forward_f16(h, x_dev, B, seg_dev, radial_dev, (char *)ws, (hipStream_t)stream)
                                 : rcv::fcn_forward_x3(h, x_dev, B, seg_dev, radial_dev, (char *)ws, (hipStream_t)stream);
    if (rc == RCV_OK && h->range_dev)
        RCV_HIP_CHECK(hipMemcpyAsync((void *)h->range_host, h->range_dev, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream));
    return rc;
}

int rcv_fcn_range_status(rcv_fcn *h, unsigned *flags_out, int clear, void *stream)
{
    if (!h) return RCV_E_INVALID;
    unsigned f = 0u;
    if (h->range_dev) {
        // read -- and, if asked, clear -- in ONE atomic exchange on the device: a forward of the next batch may be running on another
        // stream, and an atomicOr of its kernels landing between a separate read and a memset would be lost (r05; the word's second
        // slot receives the value)
        hipLaunchKernelGGL(k_range_fetch, dim3(1), dim3(1), 0, (hipStream_t)stream, h->range_dev, clear);
        RCV_LAUNCH_CHECK();
        RCV_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
        RCV_HIP_CHECK(hipMemcpy(&f, h->range_dev + 1, sizeof(unsigned), hipMemcpyDeviceToHost));
        *h->range_host = clear ? 0u : f;
    }
    if (flags_out) *flags_out = f;
    return f ? RCV_E_RANGE : RCV_OK;
}

int rcv_fcn_debug_slot(rcv_fcn *h, const void *ws, int B, int g, int slot, float *out_dev, void *stream)
{
    if (!h || !ws || !out_dev || slot < 0 || slot >= (int)h->slot_elems.size() || g < 0 || g >= h->groups) return RCV_E_INVALID;
    std::vector<size_t> off;
    size_t total;
    slot_offsets(h, B, off, total);
    const long long n = (long long)h->slot_elems[slot] * B;
    const unsigned blocks = (unsigned)std::min<long long>((n + 255) / 256, 65535);
    if (h->dtype == RCV_F16)
        return rcv::fcn_debug_slot_f16((const H16 *)((const char *)ws + off[slot]) + (long long)g * n, out_dev, n, (hipStream_t)stream);
    if (h->dtype == RCV_F16X3)  // (two halves per element)
        return rcv::fcn_debug_slot_x3((const H16 *)((const char *)ws + off[slot]) + 2 * (long long)g * n, out_dev, n, (hipStream_t)stream);
    if (h->dtype == RCV_F32)
        hipLaunchKernelGGL((k_slot_to_f32<float>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const float *)((const char *)ws + off[slot]) + (long long)g * n, out_dev, n);
    else
        hipLaunchKernelGGL((k_slot_to_f32<H16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const H16 *)((const char *)ws + off[slot]) + (long long)g * n, out_dev, n);
    RCV_LAUNCH_CHECK();
    return RCV_OK;
}

int rcv_fcn_debug_q6_pack(void *rows_dev, long long n_rows, void *stream)
{
    if (!rows_dev || n_rows <= 0) return RCV_E_INVALID;
    return rcv::fcn_q6_pack_x3(rows_dev, n_rows, (hipStream_t)stream);
}

#ifdef RCV_BLK_STAMPS
extern "C" int rcv_debug_blk_stamps(unsigned *out160)
{
    RCV_HIP_CHECK(hipDeviceSynchronize());
    RCV_HIP_CHECK(hipMemcpyFromSymbol(out160, HIP_SYMBOL(g_blk_stamps), sizeof(unsigned) * 160));
    return RCV_OK;
}
#endif
int rcv_fcn_set_option(rcv_fcn *h, int key, int value)
{
    if (!h) return RCV_E_INVALID;
    if (key == 1) { h->use_dma = value != 0; h->attrs_set = false; return RCV_OK; }
    if (key == 2) { h->use_mfma_stem = value != 0; h->attrs_set = false; return RCV_OK; }
    if (key == 3) { h->use_tail = value != 0; h->attrs_set = false; return RCV_OK; }
    if (key == 4) { h->use_block = value != 0; h->attrs_set = false; return RCV_OK; }
    if (key == 5) { h->use_chain = value != 0; h->attrs_set = false; return RCV_OK; }
    if (key == 6) { h->use_stagger = value != 0; h->attrs_set = false; return RCV_OK; }
    if (key == 7) { h->use_halo = value; h->attrs_set = false; return RCV_OK; }
    if (key == 8) { h->radial_f16 = value != 0; return RCV_OK; }
    if (key == 9) { h->use_lat = value != 0; return RCV_OK; }
    if (key == 10) { h->use_chain2 = value; return RCV_OK; }
    if (key == 11) { h->use_q6 = value; return RCV_OK; }
    if (key == 13) { h->use_tailc = value; return RCV_OK; }
    if (key == 14) { h->use_t3 = value; return RCV_OK; }
    if (key == 15) { h->use_c2w4 = value; return RCV_OK; }
    if (key == 12) {
        if (value && !h->splitk_ws) {
            const size_t bytes = (size_t)48 << 20;  // the largest split launch of the FCN programs at up to 3 networks x 1 frame needs 11 MB
            RCV_HIP_CHECK(hipMalloc((void **)&h->splitk_ws, bytes));
            h->splitk_bytes = bytes;
        }
        h->use_splitk = value != 0;
        return RCV_OK;
    }
    return RCV_E_INVALID;
}

int rcv_fcn_set_profiling(rcv_fcn *h, int enable)
{
    if (!h) return RCV_E_INVALID;
    h->profiling = enable != 0;
    h->events_valid = false;
    return RCV_OK;
}

int rcv_fcn_get_op_times(rcv_fcn *h, double *ms_out, int n_ops)
{
    if (!h || !ms_out || n_ops != (int)h->ops.size() || !h->events_valid) return RCV_E_INVALID;
    for (size_t i = 0; i < h->ops.size(); i++) {
        RCV_HIP_CHECK(hipEventSynchronize(h->events[2 * i + 1]));
        float ms = 0.f;
        RCV_HIP_CHECK(hipEventElapsedTime(&ms, h->events[2 * i], h->events[2 * i + 1]));
        ms_out[i] = ms;
    }
    return RCV_OK;
}

int rcv_fcn_get_profile(rcv_fcn *h, double *conv_ms, double *other_ms, double *conv_flop)
{
    if (!h || !conv_ms || !other_ms || !conv_flop) return RCV_E_INVALID;
    if (!h->events_valid) return RCV_E_INVALID;
    double cm = 0, om = 0;
    for (size_t i = 0; i < h->ops.size(); i++) {
        RCV_HIP_CHECK(hipEventSynchronize(h->events[2 * i + 1]));
        float ms = 0.f;
        RCV_HIP_CHECK(hipEventElapsedTime(&ms, h->events[2 * i], h->events[2 * i + 1]));
        if (h->ops[i].kind == RCV_OP_CONV) cm += ms; else om += ms;
    }
    *conv_ms = cm;
    *other_ms = om;
    double fl = 0;  // per image, per group: 2 * M * N * K
    for (auto &o : h->ops)
        if (o.kind == RCV_OP_CONV)
            fl += 2.0 * o.hout * o.wout * o.cout * (double)o.ksize * o.ksize * (o.c0 + (o.src1 >= 0 ? o.c1 : 0));
    *conv_flop = fl;
    return RCV_OK;
}

}  // extern "C"
#endif  // !RCV_TU_SUFFIX (the C ABI exists once, in the bfloat16 unit)
