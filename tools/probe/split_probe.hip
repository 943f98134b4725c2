// split_probe.hip -- numerics probe for the split-precision (fp16 hi/lo, three MFMAs per product) convolution mode.
// Standalone: hipcc --offload-arch=gfx950 -O2 -o split_probe split_probe.hip && ./split_probe
//   1. do fp16 MFMA operands keep subnormals?
//   2. error of a K-long dot product against float64: fp32 MFMA chain, plain fp16 MFMA, hi/lo split (with and without
//      power-of-two weight scaling).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 h16;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// A [16][K] row-major fp32, B [16][K] row-major fp32 (output = A * B^T, 16 x 16); mode selects the arithmetic
__global__ void k_probe(const float *A, const float *B, const float *bscale, float *out, int K, int mode)
{
    const int lane = threadIdx.x, r16 = lane & 15, kq = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (mode == 0) {  // fp32 MFMA chain
        for (int k = 0; k < K; k += 4)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r16 * K + k + kq], B[r16 * K + k + kq], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 32) {
            h16x8 ah, al, bh, bl;
            for (int e = 0; e < 8; e++) {
                const float a = A[r16 * K + k + kq * 8 + e];
                float b = B[r16 * K + k + kq * 8 + e];
                if (mode == 3) b *= bscale[r16];
                ah[e] = (h16)a;
                al[e] = (h16)(a - (float)ah[e]);
                bh[e] = (h16)b;
                bl[e] = (h16)(b - (float)bh[e]);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
            if (mode >= 2) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
            }
        }
    }
    for (int e = 0; e < 4; e++) {
        float v = acc[e];
        if (mode == 3) v *= 1.0f / bscale[r16];  // column r16 of the output = B row r16
        out[(kq * 4 + e) * 16 + r16] = v;
    }
}

__global__ void k_subnormal(float *out)
{
    const int lane = threadIdx.x;
    h16x8 a, b;
    for (int e = 0; e < 8; e++) { a[e] = (h16)0.f; b[e] = (h16)0.f; }
    if ((lane >> 4) == 0) { a[0] = (h16)9.5367431640625e-07f /* 2^-20: subnormal in fp16 */; b[0] = (h16)1024.f; }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
    // B subnormal, A normal
    for (int e = 0; e < 8; e++) { a[e] = (h16)0.f; b[e] = (h16)0.f; }
    if ((lane >> 4) == 0) { b[0] = (h16)9.5367431640625e-07f; a[0] = (h16)1024.f; }
    f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc2, 0, 0, 0);
    if (lane == 0) out[1] = acc2[0];
    // the conversion itself: does (h16) of a float keep subnormals?
    if (lane == 0) { out[2] = (float)(h16)9.5367431640625e-07f; out[3] = (float)(h16)5.9604644775390625e-08f; }
}

static double gauss()
{
    double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

int main()
{
    float *d_out;
    hipMalloc(&d_out, 4096);
    k_subnormal<<<1, 64>>>(d_out);
    float sub[4];
    hipMemcpy(sub, d_out, sizeof(sub), hipMemcpyDeviceToHost);
    printf("subnormal A (2^-20) x 1024 = %g (expect 0.000976562); subnormal B: %g; cvt keeps 2^-20: %g, 2^-24: %g\n", sub[0], sub[1], sub[2], sub[3]);
    srand(1);
    for (int K : {256, 1024, 2304, 9216, 27648}) {
        for (double wsd : {0.03, 0.001}) {
            std::vector<float> A(16 * K), B(16 * K), sc(16);
            for (auto &v : A) { double g = gauss(); v = g > 0 ? (float)g : 0.f; }
            for (auto &v : B) v = (float)(gauss() * wsd);
            for (int n = 0; n < 16; n++) {
                float mx = 0;
                for (int k = 0; k < K; k++) mx = fmaxf(mx, fabsf(B[n * K + k]));
                int e;
                frexpf(mx, &e);          // mx = m * 2^e, m in [0.5, 1)
                sc[n] = ldexpf(1.f, 14 - e);  // scaled max in [2^13, 2^14)
            }
            float *dA, *dB, *dS;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dS, 64);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dS, sc.data(), 64, hipMemcpyHostToDevice);
            std::vector<double> ex(256);
            double rms = 0;
            for (int m = 0; m < 16; m++)
                for (int n = 0; n < 16; n++) {
                    double s = 0;
                    for (int k = 0; k < K; k++) s += (double)A[m * K + k] * (double)B[n * K + k];
                    ex[m * 16 + n] = s;
                    rms += s * s;
                }
            rms = sqrt(rms / 256);
            printf("K=%5d wsd=%g rms=%.3e :", K, wsd, rms);
            const char *names[4] = {"f32chain", "f16", "split", "split+scale"};
            for (int mode = 0; mode < 4; mode++) {
                k_probe<<<1, 64>>>(dA, dB, dS, d_out, K, mode);
                float o[256];
                hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost);
                double mxe = 0, se = 0;
                for (int i = 0; i < 256; i++) { double e = fabs(o[i] - ex[i]); mxe = fmax(mxe, e); se += e * e; }
                printf("  %s max %.2e rms %.2e", names[mode], mxe / rms, sqrt(se / 256) / rms);
            }
            printf("\n");
            hipFree(dA); hipFree(dB); hipFree(dS);
        }
    }
    return 0;
}
