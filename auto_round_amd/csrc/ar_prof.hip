// ar_prof.hip -- optional device-side timing of the hot kernels (ar_profile_enable / ar_profile_read / ar_profile_reset).
// A diagnostic facility for bench.py and tools/: OFF by default; while it is off no entry point allocates or synchronises.
// When on, every profiled launch gets a start/stop hipEvent pair attached to the dispatch itself (hipExtLaunchKernelGGL), so
// the elapsed time between them is the kernel's own duration as rocprofv3 --kernel-trace reports it.
#include <mutex>
#include <vector>

#include "ar_common.hpp"

namespace ar {
namespace {
struct Rec { hipEvent_t e0, e1; int kid; int64_t units; };
std::mutex g_mu;
std::vector<Rec> g_recs;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_free;
volatile int g_on = 0;
}  // namespace

bool prof_on() { return g_on != 0; }

void prof_events(int kernel_id, int64_t units, hipEvent_t* start, hipEvent_t* stop) {
    std::lock_guard<std::mutex> lk(g_mu);
    Rec r; r.kid = kernel_id; r.units = units;
    if (!g_free.empty()) { r.e0 = g_free.back().first; r.e1 = g_free.back().second; g_free.pop_back(); }
    else { (void)hipEventCreate(&r.e0); (void)hipEventCreate(&r.e1); }
    g_recs.push_back(r);
    *start = r.e0; *stop = r.e1;
}
}  // namespace ar

extern "C" int ar_profile_enable(int on) {
    const int prev = ar::g_on;
    ar::g_on = on ? 1 : 0;
    return prev;
}

extern "C" int ar_profile_reset(void) {
    std::lock_guard<std::mutex> lk(ar::g_mu);
    for (auto& r : ar::g_recs) {
        (void)hipEventSynchronize(r.e1);
        ar::g_free.emplace_back(r.e0, r.e1);
    }
    ar::g_recs.clear();
    return AR_OK;
}

extern "C" int ar_profile_read(int kernel_id, int64_t min_units, double* total_ms, double* min_ms, int64_t* launches) {
    std::lock_guard<std::mutex> lk(ar::g_mu);
    double tot = 0.0, mn = 1e30;
    int64_t n = 0;
    for (auto& r : ar::g_recs) {
        if (r.kid != kernel_id || r.units < min_units) continue;
        hipError_t e = hipEventSynchronize(r.e1);
        if (e != hipSuccess) return (int)e;
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, r.e0, r.e1);
        if (e != hipSuccess) return (int)e;
        tot += ms; n += 1;
        if (ms < mn) mn = ms;
    }
    if (total_ms) *total_ms = tot;
    if (min_ms) *min_ms = n ? mn : 0.0;
    if (launches) *launches = n;
    return AR_OK;
}
