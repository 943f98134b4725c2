// How many vector-ALU instructions ride for free between the MFMAs of the split engine's tile (three dependent
// v_mfma_f32_16x16x32_f16 on one accumulator, two waves per SIMD)?  (r06: k_chain2_x3<256>'s epilogue arithmetic -- 114 vector
// instructions next to 192 MFMAs per half chunk and wave -- costs 5.7 of the kernel's 23 ms with every load and store removed,
// RCV_C2_DIAG 8 vs 5: ~16 cycles per vector instruction where 4 would be its issue time.)
// Per tile: 3 MFMAs, then NV independent vector instructions of one kind (inline asm, own registers).  Reported: time per tile and
// the MFMA rate, for NV = 0 .. 12, per kind, at two waves per SIMD; and the same with the vector instructions of a whole "slice"
// (4 tiles' worth) issued as one block after the fourth tile.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_valu_coissue_probe.hip -o /tmp/mvc && /tmp/mvc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define MFMA __builtin_amdgcn_mfma_f32_16x16x32_f16

// KIND: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_fma_mix_f32, 3 v_cvt_pk_f16_f32 (was v_cvt_pkrtz), 4 v_max3_f32, 5 v_add_u32 (integer)
template <int KIND> __device__ __forceinline__ void valu(float &x, float &y, f2 &p, unsigned &u)
{
    if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
    else if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p));
    else if constexpr (KIND == 2) asm volatile("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(x) : "v"(u));
    else if constexpr (KIND == 3) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u) : "v"(x), "v"(y));
    else if constexpr (KIND == 4) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
    else asm volatile("v_add_u32 %0, %0, %0" : "+v"(u));
}

template <int KIND, int NV, bool BLOCK>
__global__ __launch_bounds__(512) void k(float *sink, unsigned long long *ticks, int iters)
{
    h8 a[2], b[4];
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int q = 0; q < 2; q++) for (int e = 0; e < 8; e++) { s = s * 1664525u + 1013904223u; a[q][e] = (_Float16)(((int)(s >> 20) - 2048) * (1.0f / 1024.0f)); }
    for (int q = 0; q < 4; q++) for (int e = 0; e < 8; e++) { s = s * 1664525u + 1013904223u; b[q][e] = (_Float16)(((int)(s >> 20) - 2048) * (1.0f / 1024.0f)); }
    f4 acc[4];
    for (int j = 0; j < 4; j++) acc[j] = f4{0.f, 0.f, 0.f, 0.f};
    float x[4] = {1.f, 2.f, 3.f, 4.f}, y = 0.5f;
    f2 p[4] = {{1.f, 2.f}, {3.f, 4.f}, {5.f, 6.f}, {7.f, 8.f}};
    unsigned u[4] = {1u, 2u, 3u, 4u};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            acc[j] = MFMA(a[0], b[j], acc[j], 0, 0, 0);
            acc[j] = MFMA(a[1], b[j], acc[j], 0, 0, 0);
            acc[j] = MFMA(a[0], b[(j + 1) & 3], acc[j], 0, 0, 0);
            if constexpr (!BLOCK) {
#pragma unroll
                for (int v = 0; v < NV; v++) valu<KIND>(x[v & 3], y, p[v & 3], u[v & 3]);
            } else if (j == 3) {
#pragma unroll
                for (int v = 0; v < 4 * NV; v++) valu<KIND>(x[v & 3], y, p[v & 3], u[v & 3]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float t = x[0] + x[1] + x[2] + x[3] + p[0].x + p[1].y + p[2].x + p[3].y + (float)(u[0] + u[1] + u[2] + u[3]);
    for (int j = 0; j < 4; j++) t += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = t;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 8 + (threadIdx.x >> 6)] = c1 - c0;
}

static float *g_sink;
static unsigned long long *g_ticks;
template <int KIND, int NV, bool BLOCK> static double run()
{
    const int iters = 20000, threads = 512, blocks = 256;
    hipLaunchKernelGGL((k<KIND, NV, BLOCK>), dim3(blocks), dim3(threads), 0, 0, g_sink, g_ticks, 200);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NV, BLOCK>), dim3(blocks), dim3(threads), 0, 0, g_sink, g_ticks, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / ((double)iters * 4.0);  // ns per tile (3 MFMAs + NV vector instructions) and wave
}
template <int KIND, bool BLOCK> static void row(const char *name)
{
    const double t[] = {run<KIND, 0, BLOCK>(), run<KIND, 1, BLOCK>(), run<KIND, 2, BLOCK>(), run<KIND, 3, BLOCK>(), run<KIND, 4, BLOCK>(),
                        run<KIND, 6, BLOCK>(), run<KIND, 8, BLOCK>(), run<KIND, 12, BLOCK>()};
    printf("%-18s %s", name, BLOCK ? "block " : "spread");
    for (double v : t) printf(" %7.1f", v);
    printf("   ns per tile (3 MFMAs) and wave, two waves per SIMD\n");
}

int main()
{
    hipMalloc(&g_sink, 256 * 512 * sizeof(float));
    hipMalloc(&g_ticks, 256 * 8 * sizeof(unsigned long long));
    printf("vector instructions per tile:    0       1       2       3       4       6       8      12\n");
    row<0, false>("v_fma_f32"); row<0, true>("v_fma_f32");
    row<1, false>("v_pk_fma_f32"); row<1, true>("v_pk_fma_f32");
    row<2, false>("v_fma_mix_f32"); row<2, true>("v_fma_mix_f32");
    row<3, false>("v_cvt_pk_f16_f32"); row<3, true>("v_cvt_pk_f16_f32");
    row<4, false>("v_max3_f32"); row<4, true>("v_max3_f32");
    row<5, false>("v_add_u32"); row<5, true>("v_add_u32");
    return 0;
}
