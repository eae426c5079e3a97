// common.cpp -- error channel, version, and the in-process kernel timer used by bench.py's roofline line.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <vector>

#include "ibl_common.h"

static thread_local char g_err[512] = "";

int ibl_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* ibl_last_error(void) { return g_err; }
extern "C" int ibl_version(void) { return 100; }

// ---- kernel timer: HIP events recorded on the launch stream around selected kernel families ----------
namespace {
struct ProfSlot {
    hipEvent_t a, b;
    int id;
    double units;
};
std::mutex g_mu;
int g_enabled = 0;
std::vector<ProfSlot> g_pending;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_pool;
double g_ms[IBL_PROF_MAX], g_units[IBL_PROF_MAX];
long long g_launches[IBL_PROF_MAX];
}  // namespace

int ibl_prof_enabled() { return g_enabled; }

void ibl_prof_begin(int id, double units, void* stream, void** token) {
    *token = nullptr;
    if (!g_enabled || id <= 0 || id >= IBL_PROF_MAX) return;
    std::lock_guard<std::mutex> lk(g_mu);
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!g_pool.empty()) {
        ev = g_pool.back();
        g_pool.pop_back();
    } else {
        if (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess) return;
    }
    (void)hipEventRecord(ev.first, (hipStream_t)stream);
    g_pending.push_back({ev.first, ev.second, id, units});
    *token = reinterpret_cast<void*>(g_pending.size());     // 1-based index
}

void ibl_prof_end(void* token, void* stream) {
    if (!token) return;
    std::lock_guard<std::mutex> lk(g_mu);
    const size_t i = reinterpret_cast<size_t>(token) - 1;
    if (i < g_pending.size()) (void)hipEventRecord(g_pending[i].b, (hipStream_t)stream);
}

void ibl_prof_set_units(void* token, double units) {
    if (!token) return;
    std::lock_guard<std::mutex> lk(g_mu);
    const size_t i = reinterpret_cast<size_t>(token) - 1;
    if (i < g_pending.size()) g_pending[i].units = units;
}

static void prof_drain() {
    for (auto& p : g_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            g_ms[p.id] += ms;
            g_units[p.id] += p.units;
            g_launches[p.id] += 1;
        }
        g_pool.emplace_back(p.a, p.b);
    }
    g_pending.clear();
}

extern "C" int ibl_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    prof_drain();
    g_enabled = on ? 1 : 0;
    for (int i = 0; i < IBL_PROF_MAX; ++i) { g_ms[i] = 0; g_units[i] = 0; g_launches[i] = 0; }
    return IBL_OK;
}

extern "C" int ibl_prof_read(int id, double* ms, double* units, int64_t* launches) {
    if (id <= 0 || id >= IBL_PROF_MAX || !ms || !units || !launches) return ibl_set_error(IBL_ERR_ARG, "ibl_prof_read: bad argument");
    std::lock_guard<std::mutex> lk(g_mu);
    prof_drain();
    *ms = g_ms[id]; *units = g_units[id]; *launches = g_launches[id];
    return IBL_OK;
}
