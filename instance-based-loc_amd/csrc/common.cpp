// common.cpp -- error channel, version, and the in-process kernel timer used by bench.py's roofline line.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
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
    bool ended;          // ibl_prof_end recorded `b` in THIS use of the (pooled) event pair; a bracket left on an error path is dropped
};
unsigned g_gen = 1;      // bumped by every drain: a token handed out before it no longer addresses a slot
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
    g_pending.push_back({ev.first, ev.second, id, units, false});
    *token = reinterpret_cast<void*>(((uintptr_t)g_gen << 32) | (uintptr_t)g_pending.size());     // generation | 1-based index
}

// token -> slot of the current generation, or null (stale token: the slots were drained by ibl_prof_read / ibl_prof_enable in between)
static ProfSlot* prof_slot(void* token) {
    const uintptr_t t = reinterpret_cast<uintptr_t>(token);
    const size_t i = (size_t)(t & 0xffffffffu) - 1;
    if ((unsigned)(t >> 32) != g_gen || i >= g_pending.size()) return nullptr;
    return &g_pending[i];
}

void ibl_prof_end(void* token, void* stream) {
    if (!token) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (ProfSlot* p = prof_slot(token)) p->ended = hipEventRecord(p->b, (hipStream_t)stream) == hipSuccess;
}

void ibl_prof_set_units(void* token, double units) {
    if (!token) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (ProfSlot* p = prof_slot(token)) p->units = units;
}

static void prof_drain() {
    for (auto& p : g_pending) {
        float ms = 0.f;
        if (p.ended && hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            g_ms[p.id] += ms;
            g_units[p.id] += p.units;
            g_launches[p.id] += 1;
        }
        g_pool.emplace_back(p.a, p.b);
    }
    g_pending.clear();
    ++g_gen;
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
