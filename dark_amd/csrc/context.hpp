// dk_ctx: one (host thread, GPU) pair.  Owns the HIP stream, the device workspace and the statistics.
// Mirrors the ownership of saca::Constructor (src/saca.rs:344-384: one storage buffer sized for max_n and reused
// across calls) plus block::dc::{Encoder,Decoder} (src/block/dc.rs:21-26,96-102).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/dark_amd.h"

namespace dk {

// kernel slots for dk_stats (index into kernel_ms / kernel_bytes / kernel_launches)
enum KernelSlot : int {
    K_SYM_HIST = 0,
    K_RADIX_HIST,
    K_RADIX_SCAN,      // k_radix_scan
    K_RADIX_SCATTER,
    K_RERANK_REDUCE,
    K_RERANK_SCAN,
    K_RERANK_APPLY,
    K_ROUND_LOCAL,
    K_BWT_GATHER,
    K_DC_SUMMARY,
    K_DC_CARRY,        // k_dc_runscan + k_dc_carry_a/_b/_c + k_fill_u32
    K_DC_MAIN,
    K_DC_INIT,
    K_IBWT_HIST,       // k_ibwt_hist + k_ibwt_scan_a/_b/_c
    K_IBWT_LF,
    K_IBWT_WALK,
    K_IBWT_JUMP,
    K_IBWT_EMIT,
    K_LF_FINISH,           // k_lf_finish (slot 18: k_bucket_store's until round 5)
    K_BIG_CLASSIFY,    // k_big_reduce + k_big_spine + k_big_apply
    K_BIG_BACK,
    K_PREFIX_PROBE,
    K_PLACE_ACTIVE,    // k_place_active + k_rank_active
    K_PLATEAU_SORT,
    K_PLATEAU_RANKS,   // k_plateau_ranks, k_to_inplace, k_plateau_count/_scan/_compact
    K_RADIX_SORT_SMALL,
    K_RADIX_HIST_TEXT,     // k_radix_hist<HS_TEXT>: first pass, digits straight from the text (1 B per key)
    K_RADIX_SCATTER_TEXT,  // k_radix_scatter<false, true>: first pass, keys built from the text (13 B per pair)
    K_ISA_PARTITION,       // k_isa_init + k_isa_split<true> + k_isa_split<false> (inverse permutation through LDS windows)
    K_ISA_ASSEMBLE,        // k_isa_assemble
    K_CHAIN,               // k_chain_extract + _ends + _tiles + _spine + _verdicts + _apply (pair chains; their sort is in the radix slots)
    K_PERIOD,              // k_period_first + _spine + _fill (next break of the block's dominant period, for the period round)
    K_SLOT_COUNT
};
static_assert(K_SLOT_COUNT <= DK_NUM_KERNEL_SLOTS, "grow DK_NUM_KERNEL_SLOTS");
const char *kernel_slot_name(int slot);

struct Timer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

}  // namespace dk

struct dk_ctx {
    int device = -1;
    int numa_node = -1;  // memory node the GPU hangs on (/sys/bus/pci/devices/<bdf>/numa_node; -1: unknown): where the host coder looks for its L3 group first
    size_t max_n = 0;
    hipStream_t stream = nullptr;
    // a second stream for work that needs nothing from what the main stream does meanwhile (the L-first path's deep groups, ordered beside the
    // rounds that follow), forked from and joined to the main stream with the two events
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // device workspace: one allocation, bump-allocated per API call (all stages of a call run in sequence)
    char *ws = nullptr;
    size_t ws_size = 0, ws_used = 0, ws_peak = 0;
    // small pinned host mailbox for counters read back between rounds
    uint32_t *h_mail = nullptr;   // 1024 words, hipHostMalloc
    uint32_t *d_mail = nullptr;   // 1024 words on the device
    // pinned staging for D2H of the DC stream
    char *h_stage = nullptr;
    size_t h_stage_size = 0;
    // extra pinned staging slots for the batch entry point (one block being coded per host thread, one being filled)
    struct StageSlot { char *h = nullptr; size_t cap = 0; };
    std::vector<StageSlot> slots;
    int ensure_slot(size_t index, size_t bytes);
    std::string err;
    struct dk_batch *live_batch = nullptr;  // the streaming batch open on this context (dk_batch_begin .. dk_batch_finish), if any
    // D2H of a large block's distance stream in pieces: a host function behind every piece moves the frontier the host coder waits at
    struct D2hMark { std::atomic<size_t> *frontier; size_t value; };
    std::atomic<size_t> d2h_ready{0};
    std::vector<D2hMark> d2h_marks;
    unsigned last_flags = 0;   // DK_FLAG_* of the block the last block encode coded
    size_t last_consumed = 0;  // bytes of coded stream the last block decode read (records can be concatenated)
    dk_stats stats{};
    bool profiling = false;
    hipEvent_t round_ev[8] = {};  // suffix sort: one per in-place round in flight (live count read back one round late)
    std::vector<hipEvent_t> ev_pool;
    struct Pending { int slot; hipEvent_t a, b; double bytes; hipStream_t on; };
    std::vector<Pending> ev_pending;
    size_t ev_next = 0;

    int fail(int code, const char *fmt, ...) __attribute__((format(printf, 3, 4)));

    void ws_reset() { ws_used = 0; }
    // 256-byte aligned bump allocation; returns nullptr (and records the error) when the workspace is exhausted
    void *ws_alloc_bytes(size_t bytes);
    template <class T> T *ws_alloc(size_t count) { return static_cast<T *>(ws_alloc_bytes(count * sizeof(T))); }
    // the same for OPTIONAL buffers (a faster variant that can be done without): nullptr when it does not fit, no error recorded
    void *ws_try_alloc_bytes(size_t bytes);
    void ws_poison(void *p, size_t bytes);  // tuning build: DK_POISON
    template <class T> T *ws_try_alloc(size_t count) { return static_cast<T *>(ws_try_alloc_bytes(count * sizeof(T))); }
    size_t ws_mark() const { return ws_used; }
    void ws_release(size_t mark) { ws_used = mark; }

    // profiling: bracket a kernel launch with events on the context's stream
    void prof_begin(int slot, double bytes, hipStream_t on = nullptr);  // on: the stream the bracketed launches go to (default: `stream`)
    void prof_end();
    void prof_collect();  // after a stream sync: fold finished event pairs into stats
    int ensure_stage(size_t bytes);
};

namespace dk {

#define DK_HIP(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) return (ctx)->fail(DK_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

#define DK_TRY(expr)            \
    do {                        \
        int rc_ = (expr);       \
        if (rc_ != DK_OK) return rc_; \
    } while (0)

// RAII kernel bracket: DK_LAUNCH(ctx, slot, bytes) { kernel<<<...>>>(...); }
struct LaunchScope {
    dk_ctx *c;
    LaunchScope(dk_ctx *ctx, int slot, double bytes, hipStream_t on = nullptr) : c(ctx) { if (c->profiling) c->prof_begin(slot, bytes, on); }
    ~LaunchScope() { if (c->profiling) c->prof_end(); }
};

// A/B switches of the kernels' variants (DK_XCD, DK_PREFIX, DK_RADIX_WIDE ...): read from the environment only in the TUNING build
// (-DDK_TUNING: dark_amd/libdark_amd_tuning.so, what tests/test_env_variants.py and tools/stage_time.py load); the product library has
// every switch compiled in as its default and never looks at the environment for them.
#ifdef DK_TUNING
int tuning_knob(const char *name, int dflt);
#define DK_KNOB(name, dflt) ([] { static const int v_ = dk::tuning_knob(name, dflt); return v_; }())
#else
#define DK_KNOB(name, dflt) (dflt)
#endif

inline size_t div_up(size_t a, size_t b) { return (a + b - 1) / b; }
inline unsigned ceil_log2_u64(uint64_t v) {  // smallest b with (1 << b) >= v
    unsigned b = 0;
    while (b < 64 && (1ull << b) < v) ++b;
    return b;
}

// ---- device stages (each enqueues on ctx->stream; the ones returning host values synchronise) -------------------
// radix_sort.hip
// first pass of the suffix sort's initial sort straight from the text: key(i) = packed codes of T[i .. i+spk) (<< 8 | code of T[i-1])
struct TextKeys { const uint8_t *t = nullptr; size_t n = 0; const uint8_t *code = nullptr; int bits = 0, spk = 0, with_prev = 0; };
// What the LAST pass of a sort writes beside the sorted keys (the suffix sort's initial sort): the values go to `vals` (the suffix array:
// every suffix at its slot) instead of the ping-pong buffer, and -- when bwt is given -- L[slot] = inv_code[low byte of the key] (the
// symbol in front of the suffix rides in the key's low byte) and *origin = the slot of value 0.  A later stage overwrites the entries of
// suffixes that are not final yet.
// narrow_shift >= 0 (sorts of at most 40 bits = five passes): the sorted keys are only ever compared with their neighbours afterwards, so
// the last pass writes them as 32-bit words, (key >> narrow_shift) cut to 32 bits, into the key buffer it would have written (viewed as
// u32) -- half the bytes to write and to read back.  What a fifth pass sorts by is exactly the bits cut off: two neighbours can differ in
// them only where its digit changes, and those 256 places go to bucket_starts (global index of the first pair of every digit).
struct SortFinalOut {
    uint32_t *vals = nullptr; uint8_t *bwt = nullptr; const uint8_t *inv_code = nullptr; uint32_t *origin = nullptr;
    int narrow_shift = -1; uint32_t *bucket_starts = nullptr;
};
// final_out (may be null; only with the sort of more than 8192 pairs): see SortFinalOut; then `vals` / `vals_alt` are both free on return
int sort_pairs(dk_ctx *ctx, uint64_t *&keys, uint64_t *&keys_alt, uint32_t *&vals, uint32_t *&vals_alt, size_t count,
               int begin_bit, int end_bit, const TextKeys *text = nullptr, const SortFinalOut *final_out = nullptr);
int sort_groups(dk_ctx *ctx, const uint64_t *kin, const uint32_t *vin, uint64_t *kout, uint32_t *vout, const uint32_t *starts, size_t ngroups, size_t npairs,
                uint32_t above, int begin_bit, int end_bit);
int local_sort_tiles(dk_ctx *ctx, uint64_t *d_keys, uint32_t *d_vals, size_t count, int begin_bit, int end_bit);  // experiment hook
// rank[sa[p]] = p for a permutation sa of 0 .. n-1, through LDS windows (radix_sort.hip); scratch_a / scratch_b: n u64 each.
// marked_val (may be null): entries of sa with bit 31 set stand for sa[p] & 0x7FFFFFFF and take their value from marked_val[p] instead of p
int inverse_permutation(dk_ctx *ctx, const uint32_t *sa, size_t n, uint64_t *scratch_a, uint64_t *scratch_b, uint32_t *rank, const uint32_t *marked_val = nullptr);
bool inverse_through_windows(size_t n);  // does the inverse of a permutation of n entries take the LDS-window form?
// suffix_array.hip: d_sa_out may alias nothing in the workspace; d_text is caller or ctx owned
// d_bwt / d_origin / bwt_written (all three or none): a caller that wants L.  The sort carries the symbol in front of every suffix along and
// writes L and the origin word itself; it then sets *bwt_written -- and d_sa_out is SCRATCH WITH UNDEFINED CONTENTS on return (the L-first path
// never completes a suffix array; on the rank path nobody writes SA entries once the ranks exist, and a period round keeps its next-break
// positions there).  Only when *bwt_written comes back false does d_sa_out hold the suffix array (the caller gathers L: bwt_forward_device).
int suffix_array_device(dk_ctx *ctx, const uint8_t *d_text, size_t n, uint32_t *d_sa_out, uint8_t *d_bwt = nullptr,
                        uint32_t *d_origin = nullptr, bool *bwt_written = nullptr);
int bwt_forward_device(dk_ctx *ctx, const uint8_t *d_text, size_t n, uint32_t *d_sa, uint8_t *d_bwt, uint32_t *origin);
// bwt.hip
int bwt_gather_device(dk_ctx *ctx, const uint8_t *d_text, const uint32_t *d_sa, size_t n, uint8_t *d_bwt, uint32_t *origin);
int bwt_inverse_device(dk_ctx *ctx, const uint8_t *d_bwt, size_t n, uint32_t origin, uint8_t *d_out);
// dc.hip: d_run_end may be null
int dc_encode_device(dk_ctx *ctx, const uint8_t *d_bwt, size_t n, uint32_t init_host[256], uint32_t *d_dist, uint8_t *d_sym,
                     uint8_t *d_rank, uint32_t *d_run_end, size_t *m);

}  // namespace dk
