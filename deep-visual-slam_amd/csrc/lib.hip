// Library-level entry points of libdvslam_hip.so (include/dvslam.h: "library").
#include "common.h"

#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

namespace dvs {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// ------------------------------------------------------------------ HIP-event kernel timing
namespace {
struct Pair {
    hipEvent_t a, b;
    double work;
};
std::mutex g_mu;
std::atomic<bool> g_on{false};
std::vector<Pair> g_pairs[SLOT_COUNT];
double g_ms[SLOT_COUNT];
long g_n[SLOT_COUNT];
double g_work[SLOT_COUNT];   // algorithmic flops (MFMA kernels) or bytes (HBM kernels) of the recorded launches
const char* kSlotNames[SLOT_COUNT] = {"chain_fwd_kernel", "chain_bwd_kernel", "adam_kernel", "conv_fwd_kernel",
                                      "conv_dgrad_kernel", "conv_wgrad_kernel", "bn_fwd_kernel", "bn_bwd_kernel",
                                      "attention_fwd_kernel", "attention_bwd_kernel"};

// DVS_PROFILE_LOG=<path>: one line per recorded launch (slot, algorithmic work, milliseconds) in launch order per slot --
// the per-layer view behind the per-slot totals (tools/per_launch.py)
FILE* g_log = nullptr;

void drain_locked(int slot) {
    for (Pair& p : g_pairs[slot]) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            g_ms[slot] += ms;
            g_n[slot] += 1;
            if (g_log) fprintf(g_log, "%s %.0f %.6f\n", kSlotNames[slot], p.work, ms);
        }
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    g_pairs[slot].clear();
}
}  // namespace

bool prof_enabled() { return g_on.load(std::memory_order_relaxed); }

namespace {
std::atomic<bool> g_deterministic{false};
}
bool deterministic() { return g_deterministic.load(std::memory_order_relaxed); }
std::atomic<int> g_precision{0};
bool precision_bf16() { return g_precision.load(std::memory_order_relaxed) == 1; }
void set_deterministic(bool on) { g_deterministic.store(on); }

void prof_begin(int, hipStream_t st, hipEvent_t* start) {
    if (hipEventCreate(start) != hipSuccess) {
        *start = nullptr;
        return;
    }
    (void)hipEventRecord(*start, st);
}

void prof_work(int slot, double work) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_work[slot] += work;
}

void prof_end(int slot, hipStream_t st, hipEvent_t start, double work) {
    if (!start) return;
    hipEvent_t stop;
    if (hipEventCreate(&stop) != hipSuccess) {
        (void)hipEventDestroy(start);
        return;
    }
    (void)hipEventRecord(stop, st);
    std::lock_guard<std::mutex> lk(g_mu);
    g_pairs[slot].push_back({start, stop, work});
}

}  // namespace dvs


// ------------------------------------------------------------------ on-box peak probes (bench.py: `measured_peaks`)
namespace {
using f32x16_t = __attribute__((ext_vector_type(16))) float;
// one wave per SIMD on every CU, four independent accumulators, operands in registers: the fp32 matrix rate a kernel can reach
__global__ __launch_bounds__(256, 1) void mfma_probe_kernel(float* out, int iters) {
    f32x16_t a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = 1.0f + (float)threadIdx.x * 1e-3f, y = 0.5f - (float)threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; i += 8) {                   // (iters is rounded up to a multiple of 8 by the launcher)
#pragma unroll
        for (int u = 0; u < 8; ++u) {                      // 32 MFMAs per loop test: the scalar loop overhead stays below 1 %
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    if (s == 12345.678f) out[0] = s;                      // never true: keeps the loop alive
}
__global__ __launch_bounds__(256) void copy_probe_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
}  // namespace

extern "C" {

int dvs_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(dvs::g_mu);
    for (int s = 0; s < dvs::SLOT_COUNT; ++s) {
        dvs::drain_locked(s);
        dvs::g_ms[s] = 0.0;
        dvs::g_n[s] = 0;
        dvs::g_work[s] = 0.0;
    }
    if (on && !dvs::g_log) {
        const char* path = getenv("DVS_PROFILE_LOG");
        if (path && *path) dvs::g_log = fopen(path, "a");
    }
    if (dvs::g_log) fflush(dvs::g_log);
    dvs::g_on.store(on != 0);
    return DVS_OK;
}

int dvs_profile_slots(void) { return dvs::SLOT_COUNT; }

int dvs_profile_work(int slot, double* work) {
    DVS_REQUIRE(slot >= 0 && slot < dvs::SLOT_COUNT && work, "dvs_profile_work: bad argument");
    std::lock_guard<std::mutex> lk(dvs::g_mu);
    *work = dvs::g_work[slot];
    return DVS_OK;
}

const char* dvs_profile_slot_name(int slot) {
    return (slot >= 0 && slot < dvs::SLOT_COUNT) ? dvs::kSlotNames[slot] : "";
}

int dvs_profile_read(int slot, double* total_ms, long* launches) {
    DVS_REQUIRE(slot >= 0 && slot < dvs::SLOT_COUNT && total_ms && launches, "dvs_profile_read: bad argument");
    std::lock_guard<std::mutex> lk(dvs::g_mu);
    dvs::drain_locked(slot);
    *total_ms = dvs::g_ms[slot];
    *launches = dvs::g_n[slot];
    return DVS_OK;
}

const char* dvs_last_error(void) { return dvs::err_buf(); }

int dvs_set_deterministic(int on) {
    dvs::set_deterministic(on != 0);
    return DVS_OK;
}

int dvs_get_deterministic(void) { return dvs::deterministic() ? 1 : 0; }

int dvs_set_precision(int mode) {
    if (mode != 0 && mode != 1) return DVS_ERR_INVALID;
    dvs::g_precision.store(mode);
    return DVS_OK;
}

int dvs_get_precision(void) { return dvs::g_precision.load(); }

int dvs_abi_version(void) { return 8; }

const char* dvs_arch(void) { return "gfx950"; }

int dvs_peak_probe_mfma(float* scratch, int iters, double* flops, void* stream) {
    DVS_REQUIRE(scratch && iters > 0 && flops, "dvs_peak_probe_mfma: bad argument");
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    iters = (iters + 7) & ~7;
    hipLaunchKernelGGL(mfma_probe_kernel, dim3((unsigned)cus), dim3(256), 0, static_cast<hipStream_t>(stream), scratch, iters);
    *flops = 2.0 * 32 * 32 * 2 * 4.0 * iters * 4.0 * cus;  // 4 MFMAs per iteration, 4 waves per workgroup
    return dvs::check_launch("dvs_peak_probe_mfma");
}

int dvs_peak_probe_copy(const void* src, void* dst, size_t bytes, void* stream) {
    DVS_REQUIRE(src && dst && bytes >= 16 && (bytes & 15) == 0, "dvs_peak_probe_copy: bad argument");
    hipLaunchKernelGGL(copy_probe_kernel, dim3(256 * 8), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const float4*>(src),
                       static_cast<float4*>(dst), bytes / 16);
    return dvs::check_launch("dvs_peak_probe_copy");
}

}  // extern "C"
