// core.hip -- status strings, version, HIP error capture for librcvpose_hip.so
#include <mutex>
#include <string>

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

extern "C" {

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

int rcv_version(void) { return 100; }

const char *rcv_last_hip_error(void)
{
    static thread_local std::string copy;
    std::lock_guard<std::mutex> lock(rcv::g_err_mutex);
    copy = rcv::g_last_error;
    return copy.c_str();
}

}  // extern "C"
