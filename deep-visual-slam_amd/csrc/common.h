// Shared host-side plumbing of libdvslam_hip.so: error string, launch checks.
#pragma once
#include <hip/hip_runtime.h>

// Timing experiments that switch kernel phases off (and therefore produce WRONG results) are compiled out of the
// product: build with -DDVS_TIMING_EXPERIMENTS=1 to get the DVS_CONV_DEBUG_NOBARRIER / DVS_STEM_DEBUG switches that
// DESIGN.md section 5 quotes.
#ifndef DVS_TIMING_EXPERIMENTS
#define DVS_TIMING_EXPERIMENTS 0
#endif

#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "../../include/dvslam.h"

namespace dvs {

inline int experiment_flags(const char* env) {
#if DVS_TIMING_EXPERIMENTS
    const char* e = getenv(env);
    return e ? atoi(e) : 0;
#else
    (void)env;
    return 0;
#endif
}


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

// ---- per-kernel timing with HIP events on the launch stream (include/dvslam.h: dvs_profile_*)
enum ProfSlot { SLOT_CHAIN_FWD = 0, SLOT_CHAIN_BWD, SLOT_ADAM, SLOT_CONV_FWD, SLOT_CONV_DGRAD, SLOT_CONV_WGRAD,
                SLOT_BN_FWD, SLOT_BN_BWD, SLOT_ATTN, SLOT_ATTN_BWD, SLOT_COUNT };
bool prof_enabled();
void prof_begin(int slot, hipStream_t st, hipEvent_t* start);
void prof_end(int slot, hipStream_t st, hipEvent_t start, double work);
void prof_work(int slot, double work);

struct ProfScope {
    int slot;
    hipStream_t st;
    hipEvent_t start = nullptr;
    bool on;
    double w_ = 0.0;
    ProfScope(int s, hipStream_t stream) : slot(s), st(stream), on(prof_enabled()) {
        if (on) prof_begin(slot, st, &start);
    }
    void work(double w) {
        if (on) {
            prof_work(slot, w);
            w_ += w;
        }
    }
    ~ProfScope() {
        if (on) prof_end(slot, st, start, w_);
    }
};

// dvs_set_deterministic(1): the FORWARD pass repeats bit for bit from run to run -- no split-K forward / data-gradient
// launches (float atomics on the output), BatchNorm partial sums added in one fixed order -- so that two runs take the same
// ReLU / maxpool branches and their gradients differ by smooth rounding noise only.
bool deterministic();
// dvs_set_precision(1): the implicit-GEMM convolutions (forward, data and weight gradient) multiply on the bf16 matrix cores --
// operands rounded to bf16 as they are staged into LDS, fp32 accumulation, fp32 tensors in HBM on both sides; everything else
// (BatchNorm, the loss chain, the optimiser) is unchanged.  The opt-in mode behind the caller's `use_amp` (vo/train.py:44,177-185).
bool precision_bf16();

constexpr int kWave = 64;  // gfx950 wavefront

// 64-lane butterfly sum; every lane ends with the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// 64-lane sum on the DPP path of the vector ALU (no LDS crossbar, no lgkmcnt wait): quad swaps, the two row mirrors, then the
// row broadcasts that gfx9 / CDNA have.  Only LANE 63 ends with the total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}
__device__ __forceinline__ float wave_sum_lane63(float v) {
    v = dpp_add<0xB1, 0xF>(v);      // quad_perm [1, 0, 3, 2]
    v = dpp_add<0x4E, 0xF>(v);      // quad_perm [2, 3, 0, 1]
    v = dpp_add<0x141, 0xF>(v);     // row_half_mirror
    v = dpp_add<0x140, 0xF>(v);     // row_mirror: every lane of a 16-lane row holds the row's sum
    v = dpp_add<0x142, 0xA>(v);     // row_bcast15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);     // row_bcast31 into rows 2 and 3
    return v;
}

}  // namespace dvs
