// mfma_order_probe.hip -- is v_mfma_f32_16x16x32_f16 sensitive to the ORDER of the 32 products it sums?
// The same dot products with the k positions permuted (both operands alike): (a) the 8-element chunks of the lane groups
// swapped 1 <-> 2, (b) a random permutation of the 32 positions.  Counts outputs that differ bitwise.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
typedef _Float16 h16;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void k(const h16 *A, const h16 *B, const int *perm, const float *C, float *out)
{
    const int lane = threadIdx.x, r16 = lane & 15, kq = lane >> 4;
    h16x8 a, b;
    for (int e = 0; e < 8; e++) { const int kk = perm[kq * 8 + e]; a[e] = A[r16 * 32 + kk]; b[e] = B[r16 * 32 + kk]; }
    f32x4 acc;
    for (int e = 0; e < 4; e++) acc[e] = C[(kq * 4 + e) * 16 + r16];
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    for (int e = 0; e < 4; e++) out[(kq * 4 + e) * 16 + r16] = acc[e];
}
int main()
{
    h16 *dA, *dB; int *dP; float *dC, *dO;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dP, 128); hipMalloc(&dC, 1024); hipMalloc(&dO, 1024);
    srand(3);
    long diff_swap = 0, diff_rand = 0, total = 0;
    for (int trial = 0; trial < 400; trial++) {
        std::vector<h16> A(512), B(512); std::vector<float> C(256);
        const int mode = trial % 4;  // magnitudes: uniform, wide dynamic range, with cancellation, big accumulator
        for (int i = 0; i < 512; i++) {
            float x = (rand() / (float)RAND_MAX - 0.5f) * 2.f, y = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
            if (mode == 1) { x *= powf(2.f, (rand() % 20) - 10); y *= powf(2.f, (rand() % 20) - 10); }
            if (mode == 2 && (i & 1)) { x = -(float)A[i - 1]; y = (float)B[i - 1] * (1.f + 1e-3f * (rand() % 8)); }
            A[i] = (h16)x; B[i] = (h16)y;
        }
        for (auto &c : C) c = (rand() / (float)RAND_MAX - 0.5f) * (mode == 3 ? 4096.f : 1.f);
        hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
        float o[3][256];
        for (int v = 0; v < 3; v++) {
            int perm[32];
            for (int i = 0; i < 32; i++) perm[i] = i;
            if (v == 1) for (int e = 0; e < 8; e++) { perm[8 + e] = 16 + e; perm[16 + e] = 8 + e; }
            if (v == 2) for (int i = 31; i > 0; i--) { int j = rand() % (i + 1); int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
            hipMemcpy(dP, perm, 128, hipMemcpyHostToDevice);
            k<<<1, 64>>>(dA, dB, dP, dC, dO);
            hipMemcpy(o[v], dO, 1024, hipMemcpyDeviceToHost);
        }
        for (int i = 0; i < 256; i++) {
            total++;
            diff_swap += memcmp(&o[0][i], &o[1][i], 4) != 0;
            diff_rand += memcmp(&o[0][i], &o[2][i], 4) != 0;
        }
    }
    printf("outputs %ld: differ under chunk swap 1<->2: %ld, under a random permutation of k: %ld\n", total, diff_swap, diff_rand);
    return 0;
}
