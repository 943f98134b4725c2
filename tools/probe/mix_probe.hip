// v_fma_mix_f32 against convert-then-add on gfx950: (float)a + (float)b for all 65536 half patterns a x 4096 patterns b (both
// halves of a dword), and v - (float)h for 2^20 float patterns v; NaN results compare as equal.  Expected: 0 mismatches.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/mix_probe.hip -o tools/probe/mix_probe && tools/probe/mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ float mix_add_lo(unsigned a, unsigned b) { float r; asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float mix_add_hi(unsigned a, unsigned b) { float r; asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float mix_sub_lo(float v, unsigned h) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v)); return r; }
__device__ __forceinline__ float mix_sub_hi(float v, unsigned h) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v)); return r; }
__device__ bool same(float x, float y) { return (x != x && y != y) || __float_as_uint(x) == __float_as_uint(y); }
__global__ void k(unsigned long long *bad)
{
    const unsigned a = blockIdx.x * 256 + threadIdx.x;  // 0 .. 65535
    unsigned long long n = 0;
    for (unsigned j = 0; j < 4096; j++) {
        const unsigned b = (j * 16u + (j >> 8)) & 0xFFFFu;
        const _Float16 ha = __builtin_bit_cast(_Float16, (unsigned short)a), hb = __builtin_bit_cast(_Float16, (unsigned short)b);
        const float ref = (float)ha + (float)hb;
        n += !same(mix_add_lo(a, b), ref);
        n += !same(mix_add_hi(a << 16, b << 16), ref);
        const float v = __uint_as_float((a << 16) ^ (j * 2654435761u));
        const float ref2 = v - (float)hb;
        n += !same(mix_sub_lo(v, b), ref2);
        n += !same(mix_sub_hi(v, b << 16), ref2);
    }
    if (n) atomicAdd(bad, n);
}
int main()
{
    unsigned long long *d, h = 0;
    hipMalloc(&d, 8); hipMemset(d, 0, 8);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d);
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("v_fma_mix_f32 vs convert + add / subtract: %llu mismatches over %llu comparisons\n", h, 4ull * 65536 * 4096);
    return h != 0;
}
