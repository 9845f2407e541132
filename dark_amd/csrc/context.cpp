#include "context.hpp"

#include <cstdlib>
#include <cstring>

namespace dk {

const char *kernel_slot_name(int slot) {
    // a slot is named after the kernel it brackets, exactly as rocprofv3 prints it; a slot that brackets several kernels launched
    // back to back carries their common prefix (k_rerank_scan = k_rerank_scan + _a + _c, see the enum)
    static const char *names[K_SLOT_COUNT] = {
        "k_sym_hist",     "k_radix_hist",  "k_radix_scan",    "k_radix_scatter", "k_rerank_reduce", "k_rerank_scan",
        "k_rerank_apply", "k_round_local",  "k_bwt_gather",  "k_dc_summary",    "k_dc_carry",      "k_dc_main",       "k_dc_init",
        "k_ibwt_hist",    "k_ibwt_lf",      "k_ibwt_walk",   "k_ibwt_jump",     "k_ibwt_emit",     "k_lf_finish",  "k_big_classify",
        "k_big_back",     "k_prefix_probe", "k_place_active", "k_plateau_sort", "k_plateau_ranks",
        "k_radix_sort_small", "k_radix_hist_text", "k_radix_scatter_text", "k_isa_partition", "k_isa_assemble", "k_chain", "k_period"};
    return (slot >= 0 && slot < K_SLOT_COUNT) ? names[slot] : nullptr;
}

#ifdef DK_TUNING
int tuning_knob(const char *name, int dflt) {
    const char *e = getenv(name);
    return e && e[0] ? atoi(e) : dflt;
}
#endif

}  // namespace dk

int dk_ctx::fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    err = buf;
    return code;
}

void *dk_ctx::ws_try_alloc_bytes(size_t bytes) {
    const size_t aligned = (bytes + 255) & ~static_cast<size_t>(255);
    if (ws_used + aligned > ws_size) return nullptr;
    void *p = ws + ws_used;
    ws_used += aligned;
    if (ws_used > ws_peak) ws_peak = ws_used;
    ws_poison(p, aligned);
    return p;
}

// tuning build, DK_POISON=<byte>: every allocation from the workspace is filled with that byte first -- a read of memory nobody has written
// gives the same wrong answer every time instead of depending on what the previous call left behind
void dk_ctx::ws_poison(void *p, size_t bytes) {
#ifdef DK_TUNING
    static const int pattern = dk::tuning_knob("DK_POISON", -1);
    if (pattern >= 0) (void)hipMemsetAsync(p, pattern & 0xFF, bytes, stream);
#else
    (void)p; (void)bytes;
#endif
}

void *dk_ctx::ws_alloc_bytes(size_t bytes) {
    const size_t aligned = (bytes + 255) & ~static_cast<size_t>(255);
    if (ws_used + aligned > ws_size) {
        fail(DK_E_NOMEM, "device workspace exhausted: need %zu more bytes, %zu of %zu in use (block larger than dk_capacity?)",
             aligned, ws_used, ws_size);
        return nullptr;
    }
    void *p = ws + ws_used;
    ws_used += aligned;
    if (ws_used > ws_peak) ws_peak = ws_used;
    ws_poison(p, aligned);
    return p;
}

void dk_ctx::prof_begin(int slot, double bytes, hipStream_t on) {
    if (!on) on = stream;
    if (ev_next + 2 > ev_pool.size()) {
        for (int i = 0; i < 64; ++i) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            ev_pool.push_back(e);
        }
    }
    Pending p{slot, ev_pool[ev_next], ev_pool[ev_next + 1], bytes, on};
    ev_next += 2;
    (void)hipEventRecord(p.a, on);
    ev_pending.push_back(p);
}

void dk_ctx::prof_end() {
    if (ev_pending.empty()) return;
    (void)hipEventRecord(ev_pending.back().b, ev_pending.back().on);
}

void dk_ctx::prof_collect() {
    if (ev_pending.empty()) return;
    (void)hipStreamSynchronize(stream);
    if (side_stream) (void)hipStreamSynchronize(side_stream);
    const bool each = DK_KNOB("DK_TRACE_LAUNCHES", 0) != 0;  // (tuning build: every bracketed launch in order, for tools/kernel_breakdown.py)
    for (const Pending &p : ev_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            if (each) fprintf(stderr, "[dk] launch %-20s %9.3f us\n", dk::kernel_slot_name(p.slot), 1e3 * ms);
            stats.kernel_ms[p.slot] += ms;
            stats.kernel_bytes[p.slot] += p.bytes;
            stats.kernel_launches[p.slot] += 1;
        }
    }
    ev_pending.clear();
    ev_next = 0;
}

int dk_ctx::ensure_stage(size_t bytes) {
    if (bytes <= h_stage_size) return DK_OK;
    if (h_stage) (void)hipHostFree(h_stage);
    h_stage = nullptr;
    h_stage_size = 0;
    if (hipHostMalloc(reinterpret_cast<void **>(&h_stage), bytes, hipHostMallocDefault) != hipSuccess)
        return fail(DK_E_NOMEM, "pinned staging allocation of %zu bytes failed", bytes);
    h_stage_size = bytes;
    return DK_OK;
}

int dk_ctx::ensure_slot(size_t index, size_t bytes) {
    if (slots.size() <= index) slots.resize(index + 1);
    StageSlot &sl = slots[index];
    if (bytes <= sl.cap) return DK_OK;
    if (sl.h) (void)hipHostFree(sl.h);
    sl.h = nullptr;
    sl.cap = 0;
    const size_t want = bytes + bytes / 8;  // a little slack so that blocks of similar size do not re-allocate
    if (hipHostMalloc(reinterpret_cast<void **>(&sl.h), want, hipHostMallocDefault) != hipSuccess)
        return fail(DK_E_NOMEM, "pinned staging slot of %zu bytes failed", want);
    sl.cap = want;
    return DK_OK;
}
