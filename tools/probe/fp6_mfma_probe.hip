// v_mfma_scale_f32_16x16x128_f8f6f4 with fp6 (e2m3) operands on gfx950: operand layout, scale semantics, zero-padded blocks, rate.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/fp6_mfma_probe.hip -o tools/probe/fp6_mfma_probe && tools/probe/fp6_mfma_probe
// Assumed (checked here against a host emulation with exactly representable data):
//   lane l holds row (A) / column (B) l & 15 and the 32 K elements of block l >> 4 (k = 32 * (l >> 4) + j), element j in bits
//   [6j, 6j + 6) of its six operand registers; e2m3 = sign | 2-bit exponent (bias 1) | 3-bit mantissa; the scale operand's
//   byte op_sel selects is an e8m0 exponent (value 2^(byte - 127)) for the lane's own (row, block).
// Also: (2) products against garbage are harmless when the other side's elements are zero (fp6 has no NaN / infinity),
// (3) cycles per instruction next to v_mfma_f32_16x16x32_f16 (s_memtime around 256 back-to-back instructions, one wave per SIMD),
// (4) what v_cvt_scalef32_pk32_fp6_f16 does with its scale (divides or multiplies), its rounding and saturation.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v32h __attribute__((ext_vector_type(32)));

static float e2m3(unsigned c)
{
    const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7;
    const float v = e == 0 ? m / 8.0f : (1.0f + m / 8.0f) * (float)(1 << (e - 1));
    return s ? -v : v;
}

__global__ void k_mfma(const unsigned *a, const unsigned *b, const unsigned *sa, const unsigned *sb, float *d, int garbage)
{
    const int l = threadIdx.x;
    v8i av, bv;
    for (int q = 0; q < 6; q++) { av[q] = a[l * 6 + q]; bv[q] = b[l * 6 + q]; }
    av[6] = garbage ? 0x7fffffff : 0; av[7] = garbage ? 0x12345678 : 0; bv[6] = 0; bv[7] = 0;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 2, 2, 0, (int)sa[l], 0, (int)sb[l]);
    for (int e = 0; e < 4; e++) d[((l >> 4) * 4 + e) * 16 + (l & 15)] = acc[e];  // row = (l >> 4) * 4 + e, col = l & 15
}

// scale byte selection: the same with the scales in byte 1 of the operand (op_sel = 1)
__global__ void k_mfma_sel(const unsigned *a, const unsigned *b, const unsigned *sa, const unsigned *sb, float *d)
{
    const int l = threadIdx.x;
    v8i av = {0, 0, 0, 0, 0, 0, 0, 0}, bv = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = 0; q < 6; q++) { av[q] = a[l * 6 + q]; bv[q] = b[l * 6 + q]; }
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 2, 2, 1, (int)((sa[l] << 8) | 0x55), 1, (int)((sb[l] << 8) | 0xaa));
    // NOTE argument order of the builtin: (a, b, c, cbsz, blgp, opsel_a, scale_a, opsel_b, scale_b)
    for (int e = 0; e < 4; e++) d[((l >> 4) * 4 + e) * 16 + (l & 15)] = acc[e];
}

template <int MODE>
__global__ void k_rate(unsigned long long *cyc, float *sink, int iters)
{
    v8i av, bv;
    for (int q = 0; q < 8; q++) { av[q] = (threadIdx.x * 2654435761u + q * 40503u) & (MODE == 1 ? 0x1b6db6db : 0x3b3b3b3b); bv[q] = (threadIdx.x * 40503u + q * 2654435761u) & (MODE == 1 ? 0x1b6db6db : 0x3b3b3b3b); }
    v8h ah, bh;
    for (int q = 0; q < 8; q++) { ah[q] = (_Float16)(((threadIdx.x + q) % 13) * 0.25f - 1.5f); bh[q] = (_Float16)(((threadIdx.x * 3 + q) % 11) * 0.125f - 0.5f); }
    v4f acc[8];
    for (int q = 0; q < 8; q++) acc[q] = v4f{0.f, 0.f, 0.f, 0.f};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if constexpr (MODE == 0) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[q], 0, 0, 0);
            else if constexpr (MODE == 1) acc[q] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc[q], 2, 2, 0, 127, 0, 127);
            else if constexpr (MODE == 2) acc[q] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc[q], 0, 0, 0, 127, 0, 127);
            else acc[q] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc[q], 2, 0, 0, 127, 0, 127);  // A fp6, B fp8
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int q = 0; q < 8; q++) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void k_cvt(const _Float16 *x, unsigned *out, float scale)
{
    v32h v;
    for (int j = 0; j < 32; j++) v[j] = x[threadIdx.x * 32 + j];
    v6i r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, scale);
    for (int q = 0; q < 6; q++) out[threadIdx.x * 6 + q] = r[q];
}

static unsigned get6(const unsigned *w, int j) { const int bit = 6 * j; unsigned long long two = w[bit >> 5] | ((unsigned long long)((bit >> 5) + 1 < 6 ? w[(bit >> 5) + 1] : 0u) << 32); return (unsigned)(two >> (bit & 31)) & 63u; }
static void put6(unsigned *w, int j, unsigned c) { const int bit = 6 * j; unsigned long long two = (unsigned long long)(c & 63u) << (bit & 31); w[bit >> 5] |= (unsigned)two; if ((bit >> 5) + 1 < 6) w[(bit >> 5) + 1] |= (unsigned)(two >> 32); }

int main()
{
    unsigned ha[64 * 6], hb[64 * 6], hsa[64], hsb[64];
    float A[16][128], B[128][16];
    int failures = 0;
    for (int trial = 0; trial < 3; trial++) {
        // trial 0: every element used; trial 1: elements 16..31 of every block zero on the B side, garbage on the A side; trial 2 = 0 with op_sel 1
        memset(ha, 0, sizeof(ha)); memset(hb, 0, sizeof(hb));
        srand(7 + trial);
        double SA[16][4], SB[16][4];
        for (int l = 0; l < 64; l++) {
            const int r = l & 15, blk = l >> 4;
            hsa[l] = 127 + (rand() % 7) - 3; hsb[l] = 127 + (rand() % 5) - 2;
            SA[r][blk] = ldexp(1.0, (int)hsa[l] - 127); SB[r][blk] = ldexp(1.0, (int)hsb[l] - 127);
            for (int j = 0; j < 32; j++) {
                unsigned ca = rand() & 63, cb = rand() & 63;
                if (trial == 1 && j >= 16) cb = 0;
                put6(ha + l * 6, j, ca); put6(hb + l * 6, j, cb);
                A[r][blk * 32 + j] = e2m3(ca); B[blk * 32 + j][r] = e2m3(cb);
            }
        }
        unsigned *da, *db, *dsa, *dsb; float *dd; float hd[256];
        hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dsa, sizeof(hsa)); hipMalloc(&dsb, sizeof(hsb)); hipMalloc(&dd, sizeof(hd));
        hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
        hipMemcpy(dsa, hsa, sizeof(hsa), hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof(hsb), hipMemcpyHostToDevice);
        if (trial == 2) hipLaunchKernelGGL(k_mfma_sel, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
        else hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd, trial == 1);
        hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
        int bad = 0; double worst = 0;
        for (int r = 0; r < 16; r++)
            for (int c = 0; c < 16; c++) {
                double ref = 0;
                for (int k = 0; k < 128; k++) ref += (double)A[r][k] * SA[r][k >> 5] * (double)B[k][c] * SB[c][k >> 5];
                const double err = fabs(ref - hd[r * 16 + c]) / (fabs(ref) + 1e-3);
                if (err > worst) worst = err;
                if (err > 1e-5) { if (bad < 4) printf("  trial %d [%d][%d]: gpu %.6f ref %.6f\n", trial, r, c, hd[r * 16 + c], ref); bad++; }
            }
        printf("trial %d (%s): %d / 256 outputs off, worst relative difference %.2e\n", trial,
               trial == 0 ? "all 32 elements per block, scales in byte 0" : trial == 1 ? "B elements 16..31 zero, A registers 6..7 garbage" : "scales in byte 1, op_sel 1",
               bad, worst);
        failures += bad != 0;
        hipFree(da); hipFree(db); hipFree(dsa); hipFree(dsb); hipFree(dd);
    }
    // rate
    {
        unsigned long long *dc; float *ds; const int nb = 256 * 4, iters = 512;
        hipMalloc(&dc, nb * 8); hipMalloc(&ds, nb * 64 * 4);
        unsigned long long hc[256 * 4];
        const char *names[4] = {"v_mfma_f32_16x16x32_f16", "scale 16x16x128 fp6 x fp6", "scale 16x16x128 fp8 x fp8", "scale 16x16x128 fp6 x fp8"};
        for (int mode = 0; mode < 4; mode++) {
            for (int rep = 0; rep < 2; rep++) {
                if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(nb), dim3(64), 0, 0, dc, ds, iters);
                if (mode == 1) hipLaunchKernelGGL(k_rate<1>, dim3(nb), dim3(64), 0, 0, dc, ds, iters);
                if (mode == 2) hipLaunchKernelGGL(k_rate<2>, dim3(nb), dim3(64), 0, 0, dc, ds, iters);
                if (mode == 3) hipLaunchKernelGGL(k_rate<3>, dim3(nb), dim3(64), 0, 0, dc, ds, iters);
            }
            hipMemcpy(hc, dc, nb * 8, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < nb; i++) s += (double)hc[i];
            printf("%-28s %.1f cycles per instruction (one wave per SIMD, 8 independent accumulators)\n", names[mode], s / nb / (iters * 8.0));
        }
        hipFree(dc); hipFree(ds);
    }
    // conversion
    {
        _Float16 hx[32]; unsigned ho[6];
        const float vals[32] = {0.f, 0.0625f, 0.125f, 0.1875f, 0.3125f, 0.4375f, 0.9375f, 1.0f, 1.0625f, 1.1875f, 1.9f, 2.0f, 3.75f, 3.9f, 4.0f, 7.5f,
                                7.75f, 8.0f, 100.f, -0.0625f, -0.19f, -1.06f, -7.75f, -100.f, 0.5625f, 0.6875f, 1.3125f, 1.4375f, 2.125f, 2.375f, 5.25f, 5.75f};
        for (int j = 0; j < 32; j++) hx[j] = (_Float16)vals[j];
        _Float16 *dx; unsigned *dout;
        hipMalloc(&dx, sizeof(hx)); hipMalloc(&dout, sizeof(ho));
        hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
        for (float sc : {1.0f, 2.0f, 0.5f}) {
            hipLaunchKernelGGL(k_cvt, dim3(1), dim3(1), 0, 0, dx, dout, sc);
            hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
            printf("v_cvt_scalef32_pk32_fp6_f16, scale %.1f:", sc);
            for (int j = 0; j < 32; j++) printf(" %g->%g", vals[j], e2m3(get6(ho, j)));
            printf("\n");
        }
        hipFree(dx); hipFree(dout);
    }
    return failures;
}
