// core.hip -- status strings, version, HIP error capture for librcvpose_hip.so
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "common.hpp"

namespace rcv {
namespace {
std::mutex g_err_mutex;
std::string g_last_error = "";
}  // namespace

void set_hip_error(hipError_t e, const char *what, const char *file, int line)
{
    std::lock_guard<std::mutex> lock(g_err_mutex);
    char buf[512];
    snprintf(buf, sizeof(buf), "%s (%d) at %s:%d: %s", hipGetErrorString(e), (int)e, file, line, what);
    g_last_error = buf;
}
}  // namespace rcv

namespace rcv {
namespace {
typedef _Float16 cal_h8 __attribute__((ext_vector_type(8)));
typedef float cal_f4 __attribute__((ext_vector_type(4)));
// Measurement aid (bench.py): a bare v_mfma_f32_16x16x32_f16 loop on pseudo-random operands, one wave per SIMD on every CU,
// eight independent accumulators -- what the matrix pipe delivers at the clock THIS device holds under it.  Boxes of the pool
// differ by several per cent there (MI355X_MICROARCH.md, DVFS give-back item 5), which is as much as a round of kernel work moves.
__global__ __launch_bounds__(256) void k_mfma_calibrate(float *sink, unsigned long long *ticks, int iters)
{
    cal_h8 a, b;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        s = s * 1664525u + 1013904223u;
        a[q] = (_Float16)(((int)(s >> 20) - 2048) * (1.0f / 1024.0f));
        s = s * 1664525u + 1013904223u;
        b[q] = (_Float16)(((int)(s >> 20) - 2048) * (1.0f / 1024.0f));
    }
    cal_f4 acc[8];
#pragma unroll
    for (int q = 0; q < 8; q++) acc[q] = cal_f4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 8; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 8; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc[q], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; q++) t += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = t;
    if (threadIdx.x == 0) {
        ticks[2 * blockIdx.x] = c1 - c0;
        ticks[2 * blockIdx.x + 1] = r1 - r0;
    }
}
}  // namespace
}  // namespace rcv

extern "C" {

namespace {
// frees what rcv_mfma_calibrate allocated on every way out (the checks below return early)
struct CalBuffers {
    float *sink = nullptr;
    unsigned long long *ticks = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~CalBuffers()
    {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (sink) (void)hipFree(sink);
        if (ticks) (void)hipFree(ticks);
    }
};
}  // namespace

int rcv_mfma_calibrate(double seconds, double *tflops_out, double *clock_mhz_out, void *stream)
{
    if (!(seconds > 0) || seconds > 10 || !tflops_out) return RCV_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    int dev = 0;
    hipDeviceProp_t prop;
    RCV_HIP_CHECK(hipGetDevice(&dev));
    RCV_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    const int blocks = std::max(1, prop.multiProcessorCount) * 4;  // four workgroups of four waves per CU, taken one after the other per SIMD
    CalBuffers B;
    RCV_HIP_CHECK(hipMalloc((void **)&B.sink, (size_t)blocks * 256 * sizeof(float)));
    RCV_HIP_CHECK(hipMalloc((void **)&B.ticks, (size_t)blocks * 2 * sizeof(unsigned long long)));
    RCV_HIP_CHECK(hipEventCreate(&B.e0));
    RCV_HIP_CHECK(hipEventCreate(&B.e1));
    int iters = 20000;  // 16 MFMAs per iteration
    double tf = 0, mhz = 0;
    for (int pass = 0; pass < 3; pass++) {  // pass 0 sizes the loop (and warms the clock governor), the last one is reported
        RCV_HIP_CHECK(hipEventRecord(B.e0, st));
        hipLaunchKernelGGL(rcv::k_mfma_calibrate, dim3(blocks), dim3(256), 0, st, B.sink, B.ticks, iters);
        RCV_LAUNCH_CHECK();
        RCV_HIP_CHECK(hipEventRecord(B.e1, st));
        RCV_HIP_CHECK(hipEventSynchronize(B.e1));
        float ms = 0.f;
        RCV_HIP_CHECK(hipEventElapsedTime(&ms, B.e0, B.e1));
        const double flop = (double)blocks * 4 * iters * 16.0 * (2.0 * 16 * 16 * 32);
        tf = flop / (ms * 1e-3) / 1e12;
        if (pass == 0) iters = (int)std::min(4.0e6, std::max(1000.0, iters * seconds * 1e3 / std::max(ms, 1e-3f)));
    }
    if (clock_mhz_out) {
        std::vector<unsigned long long> h((size_t)blocks * 2);
        RCV_HIP_CHECK(hipMemcpy(h.data(), B.ticks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<double> f;
        for (int i = 0; i < blocks; i++)
            if (h[2 * i + 1]) f.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);  // s_memrealtime ticks at 100 MHz
        std::sort(f.begin(), f.end());
        mhz = f.empty() ? 0.0 : f[f.size() / 2];
        *clock_mhz_out = mhz;
    }
    *tflops_out = tf;
    return RCV_OK;
}

const char *rcv_strerror(int status)
{
    switch (status) {
    case RCV_OK: return "ok";
    case RCV_E_INVALID: return "invalid argument";
    case RCV_E_HIP: return "HIP runtime error";
    case RCV_E_WORKSPACE: return "workspace too small";
    case RCV_E_EMPTY: return "empty point cloud or radius list";
    case RCV_E_BADGRID: return "vote grid has a non-positive dimension";
    case RCV_E_TOO_LARGE: return "grid, radius or tile count beyond supported range";
    case RCV_E_GRID_BUFFER: return "vote grid buffer smaller than dim^3";
    case RCV_E_UNSUPPORTED: return "unsupported layer shape or dtype";
    case RCV_E_RANGE: return "an activation left the half-precision range of the engine's storage (maps clipped); use RCV_F32";
    default: return "unknown status";
    }
}

int rcv_version(void) { return RCV_ABI_VERSION; }

int rcv_abi_check(int version, size_t sizeof_op, size_t sizeof_acc_result, size_t sizeof_strided, size_t sizeof_icp_result)
{
    return (version == RCV_ABI_VERSION && sizeof_op == sizeof(rcv_op) && sizeof_acc_result == sizeof(rcv_acc_result) &&
            sizeof_strided == sizeof(rcv_strided) && sizeof_icp_result == sizeof(rcv_icp_result))
               ? RCV_OK
               : RCV_E_INVALID;
}

const char *rcv_last_hip_error(void)
{
    static thread_local std::string copy;
    std::lock_guard<std::mutex> lock(rcv::g_err_mutex);
    copy = rcv::g_last_error;
    return copy.c_str();
}

}  // extern "C"
