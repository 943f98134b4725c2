// Shared helpers for librcvpose_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "rcvpose_hip.h"

namespace rcv {

void set_hip_error(hipError_t e, const char *what, const char *file, int line);

#define RCV_HIP_CHECK(expr)                                        \
    do {                                                           \
        hipError_t _e = (expr);                                    \
        if (_e != hipSuccess) {                                    \
            ::rcv::set_hip_error(_e, #expr, __FILE__, __LINE__);   \
            return RCV_E_HIP;                                      \
        }                                                          \
    } while (0)

#define RCV_LAUNCH_CHECK() RCV_HIP_CHECK(hipGetLastError())

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

constexpr int WAVE = 64;

}  // namespace rcv
