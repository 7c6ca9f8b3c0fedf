// Shared host-side plumbing of libdvslam_hip.so: error string, launch checks.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/dvslam.h"

namespace dvs {

// Per-thread last-error text (the C-ABI keeps no other thread-local state).
char* err_buf();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DVS_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return DVS_OK;
}

#define DVS_REQUIRE(cond, ...) \
    do {                       \
        if (!(cond)) return dvs::fail(DVS_ERR_INVALID, __VA_ARGS__); \
    } while (0)

constexpr int kWave = 64;  // gfx950 wavefront

// 64-lane butterfly sum; every lane ends with the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace dvs
