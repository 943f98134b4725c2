// Does the ORDER in which a tile grid's MFMAs are issued matter when all operands sit in registers?  (r06: with the operands of
// k_conv_halo's MFMAs swapped its K loop ran 5-8 % slower, with the column-outermost loop order 26 % slower -- same instructions,
// same registers.)  20 accumulator tiles acc[5][4], operands a[5] (tile rows), b[4] (tile columns), v_mfma_f32_16x16x32_f16; cycles per
// MFMA (s_memtime) for the four (role, loop order) combinations, at one and at two waves per SIMD, with and without the three-pass
// hi/lo structure of the split engine.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_operand_order_probe.hip -o /tmp/moo && /tmp/moo
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define MFMA __builtin_amdgcn_mfma_f32_16x16x32_f16

template <int V>
__global__ __launch_bounds__(512) void k(float *sink, unsigned long long *ticks, int iters)
{
    h8 a[5], b[4];
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int q = 0; q < 5; q++) for (int e = 0; e < 8; e++) { s = s * 1664525u + 1013904223u; a[q][e] = (_Float16)(((int)(s >> 20) - 2048) * (1.0f / 1024.0f)); }
    for (int q = 0; q < 4; q++) for (int e = 0; e < 8; e++) { s = s * 1664525u + 1013904223u; b[q][e] = (_Float16)(((int)(s >> 20) - 2048) * (1.0f / 1024.0f)); }
    f4 acc[5][4];
    for (int i = 0; i < 5; i++) for (int j = 0; j < 4; j++) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if constexpr (V == 0) {  // rows outermost, A = row operand (k_conv_halo as it is)
#pragma unroll
            for (int i = 0; i < 5; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = MFMA(a[i], b[j], acc[i][j], 0, 0, 0);
        } else if constexpr (V == 1) {  // rows outermost, operands swapped (A = column operand)
#pragma unroll
            for (int i = 0; i < 5; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = MFMA(b[j], a[i], acc[i][j], 0, 0, 0);
        } else if constexpr (V == 2) {  // columns outermost, A = row operand
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 5; i++) acc[i][j] = MFMA(a[i], b[j], acc[i][j], 0, 0, 0);
        } else {  // columns outermost, swapped
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 5; i++) acc[i][j] = MFMA(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float t = 0.f;
    for (int i = 0; i < 5; i++) for (int j = 0; j < 4; j++) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = t;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 8 + (threadIdx.x >> 6)] = c1 - c0;
}

template <int V> static void run(int waves_per_simd, float *sink, unsigned long long *ticks)
{
    const int iters = 20000, threads = 256 * waves_per_simd, blocks = 256;
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, sink, ticks, 200);  // (warm-up)
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, sink, ticks, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8];
    hipMemcpy(h, ticks + 8 * 100, sizeof(h), hipMemcpyDeviceToHost);
    const double per = (double)h[0] / ((double)iters * 20.0);
    printf("variant %d, %d wave(s) per SIMD: %.2f s_memtime ticks per MFMA and wave (100 MHz: x clock / 100 MHz cycles), %.1f TFLOP/s\n", V, waves_per_simd, per,
           256.0 * 4 * waves_per_simd * iters * 20.0 * 2 * 16 * 16 * 32 / (ms * 1e-3) / 1e12);
}

int main()
{
    float *sink; unsigned long long *ticks;
    hipMalloc(&sink, 256 * 512 * sizeof(float));
    hipMalloc(&ticks, 256 * 8 * sizeof(unsigned long long));
    printf("0: rows outermost, A = row operand   1: rows outermost, swapped   2: columns outermost, A = row operand   3: columns outermost, swapped\n");
    for (int w = 1; w <= 2; w++) { run<0>(w, sink, ticks); run<1>(w, sink, ticks); run<2>(w, sink, ticks); run<3>(w, sink, ticks); }
    return 0;
}
