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

int dvs_abi_version(void) { return 6; }

const char* dvs_arch(void) { return "gfx950"; }

}  // extern "C"
