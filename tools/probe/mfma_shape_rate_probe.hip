// Which rate does the chip HOLD on each fp16 MFMA shape?  The convolution kernels are built on v_mfma_f32_16x16x32_f16; the calibration
// loop (rcv_mfma_calibrate) says the chip sustains ~1.45-1.55 PFLOP/s on it, 58-62 % of the 2.5 PFLOP/s that 2.4 GHz would give: the
// clock drops under matrix load.  v_mfma_f32_32x32x16_f16 does the same flops per cycle with half the A / B register reads per flop.  If the
// drop is a power limit, the shape with fewer operand reads may hold a higher clock.  Bare register loops, every SIMD, 0.3 s each:
// sustained TFLOP/s and the implied clock (flops / (1024 flop per cycle and SIMD x 1024 SIMDs)), at one and two waves per SIMD, with
// independent accumulators (4 for 16x16, 2 for 32x32: the same 16 accumulator registers) and, for the split engine's pattern, chains of
// three dependent MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_shape_rate_probe.hip -o /tmp/msr && /tmp/msr
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int SHAPE, bool CHAIN3>
__global__ __launch_bounds__(512) void k(float *sink, long long iters)
{
    h8 a[3], b[3];
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int q = 0; q < 3; q++)
        for (int e = 0; e < 8; e++) {
            s = s * 1664525u + 1013904223u; a[q][e] = (_Float16)(((int)(s >> 20) - 2048) * (1.0f / 4096.0f));
            s = s * 1664525u + 1013904223u; b[q][e] = (_Float16)(((int)(s >> 20) - 2048) * (1.0f / 4096.0f));
        }
    float t = 0.f;
    if constexpr (SHAPE == 16) {
        f4 acc[4];
        for (int j = 0; j < 4; j++) acc[j] = f4{0.f, 0.f, 0.f, 0.f};
        for (long long it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 2; r++)  // 24 MFMAs of 16384 flop per iteration
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if constexpr (CHAIN3) {
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[j & 1], acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[j & 1], acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[2], acc[j], 0, 0, 0);
                    } else {
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], acc[j], 0, 0, 0);
                        acc[(j + 1) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[1], acc[(j + 1) & 3], 0, 0, 0);
                        acc[(j + 2) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2], b[2], acc[(j + 2) & 3], 0, 0, 0);
                    }
                }
            asm volatile("" ::: "memory");
        }
        for (int j = 0; j < 4; j++) t += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    } else {
        f16v acc[2];
        for (int j = 0; j < 2; j++)
            for (int e = 0; e < 16; e++) acc[j][e] = 0.f;
        for (long long it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 2; r++)  // 12 MFMAs of 32768 flop per iteration
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    if constexpr (CHAIN3) {
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[j], acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[j], acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[2], acc[j], 0, 0, 0);
                    } else {
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc[j], 0, 0, 0);
                        acc[j ^ 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[1], acc[j ^ 1], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2], b[2], acc[j], 0, 0, 0);
                    }
                }
            asm volatile("" ::: "memory");
        }
        for (int j = 0; j < 2; j++)
            for (int e = 0; e < 16; e++) t += acc[j][e];
    }
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int SHAPE, bool CHAIN3> static void run(int waves_per_simd, float *sink, const char *name)
{
    const int threads = 256 * waves_per_simd, blocks = 256;
    const double flop_per_iter_wave = 24.0 * 16384.0;  // (both shapes: 24 x 16384 = 12 x 32768)
    hipLaunchKernelGGL((k<SHAPE, CHAIN3>), dim3(blocks), dim3(threads), 0, 0, sink, 2000LL);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // size the run for ~0.3 s from a short one
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, CHAIN3>), dim3(blocks), dim3(threads), 0, 0, sink, 20000LL);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    const long long iters = (long long)(20000.0 * 300.0 / ms);
    double best = 0.0, last = 0.0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, CHAIN3>), dim3(blocks), dim3(threads), 0, 0, sink, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        last = (double)blocks * 4 * waves_per_simd * iters * flop_per_iter_wave / (ms * 1e-3) / 1e12;
        if (last > best) best = last;
    }
    printf("%-44s %d wave(s) per SIMD: %7.1f TFLOP/s sustained over 0.3 s (third run %7.1f) = %.0f MHz x 1024 flop per cycle and SIMD\n", name, waves_per_simd, best, last,
           last * 1e12 / (1024.0 * 1024.0) / 1e6);
}

int main()
{
    float *sink;
    hipMalloc(&sink, 256 * 512 * sizeof(float));
    for (int w = 1; w <= 2; w++) {
        run<16, false>(w, sink, "v_mfma_f32_16x16x32_f16, independent");
        run<32, false>(w, sink, "v_mfma_f32_32x32x16_f16, independent");
        run<16, true>(w, sink, "v_mfma_f32_16x16x32_f16, chains of three");
        run<32, true>(w, sink, "v_mfma_f32_32x32x16_f16, chains of three");
    }
    return 0;
}
