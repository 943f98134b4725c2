// Does MODE.FP16_OVFL (bit 23 of the MODE register) make v_cvt_pk_f16_f32 / v_cvt_f16_f32 saturate on gfx950?
//   hipcc --offload-arch=gfx950 -O2 tools/probe/ovfl_probe.hip -o tools/probe/ovfl_probe && tools/probe/ovfl_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *in, float *out, int n, int mode)
{
    if (mode) __builtin_amdgcn_s_setreg(((1 - 1) << 11) | (23 << 6) | 1, 1);  // hwreg(HW_REG_MODE, 23, 1) = 1
    const int i = threadIdx.x;
    if (i >= n) return;
    const f2 v = {in[i], -in[i]};
    const h2 p = __builtin_convertvector(v, h2);  // v_cvt_pk_f16_f32
    const _Float16 s = (_Float16)in[i];           // v_cvt_f16_f32
    out[3 * i] = (float)p[0];
    out[3 * i + 1] = (float)p[1];
    out[3 * i + 2] = (float)s;
}
int main()
{
    const int n = 8;
    float h_in[n] = {1.0f, 65504.0f, 65519.0f, 65520.0f, 70000.0f, 1.0e6f, 3.0e38f, INFINITY};
    float *d_in, *d_out, h_out[3 * n];
    hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, sizeof(h_out));
    hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; mode++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, d_out, n, mode);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("FP16_OVFL = %d\n", mode);
        for (int i = 0; i < n; i++) printf("  %12g -> pk(+) %10g  pk(-) %10g  scalar %10g\n", h_in[i], h_out[3 * i], h_out[3 * i + 1], h_out[3 * i + 2]);
    }
    return 0;
}
