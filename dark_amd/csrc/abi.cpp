// extern "C" boundary (include/dark_amd.h).  Thin: argument checks, workspace carving, stage sequencing, timing.
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "context.hpp"
#include <cstdlib>
#include <atomic>
#include "entropy.hpp"

using namespace dk;

namespace {

int begin_call(dk_ctx *ctx) {
    if (!ctx) return DK_E_ARG;
    // a streaming batch owns the workspace and the staging slots until it is finished: any other call would pull them from under
    // the coding threads
    if (ctx->live_batch) return ctx->fail(DK_E_ARG, "a streaming batch is open on this context: call dk_batch_finish first");
    ctx->err.clear();
    ctx->last_flags = 0;
    ctx->ws_reset();
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(DK_E_HIP, "hipSetDevice(%d) failed", ctx->device);
    return DK_OK;
}
void end_call(dk_ctx *ctx) {
    if (ctx->profiling) ctx->prof_collect();
}
int check_n(dk_ctx *ctx, size_t n) {
    if (n == 0) return ctx->fail(DK_E_ARG, "empty block (the reference panics at src/saca.rs:107)");
    if (n > ctx->max_n) return ctx->fail(DK_E_ARG, "block of %zu bytes exceeds the context capacity %zu", n, ctx->max_n);
    return DK_OK;
}
#ifndef DK_D2H_OVERLAP
#define DK_D2H_OVERLAP 1  // the distance stream of a large single block leaves the GPU in pieces while the host coder already runs (0: A/B builds)
#endif

size_t workspace_bytes(size_t max_n) {
    // Peak of the bump allocator, reached inside the suffix sort of a host-pointer block call (every term is one ws_alloc of the call):
    //   text copy                                   n     (host entry points only)
    //   L                                           n
    //   SA                                         4 n
    //   suffix sort: keys x3 24 n, suffix lists x3 12 n, rank 4 n, slot positions x2 8 n, group ids x2 8 n,
    //                gstart + bigidx 4 n, big-group offsets n / 8, symbols in front x2 2 n            = 62.1 n
    //   on top of that, one of (never two at a time: each is released before the next is taken)
    //     radix sort: per-tile digit tables n / 4 (256 counters per 4096 pairs) + digit plane n (optional, from 2^20 pairs:
    //                 ws_try_alloc -- the sort runs without it when it does not fit)                  = 1.25 n
    //     rerank:     one flag byte per slot n + tile aggregates n / 64 (L-first: + n / 128)          = 1.03 n
    //     pair chains: one verdict byte per record, at most n / 2
    //   and, for the whole of the L-first path (lfirst_path: held TOGETHER with the sort's and the rerank's temporaries above):
    //     the list of deep groups 16 MiB, two giant lists with their arenas 5.6 MiB, the grid-wide measure's results 1 MiB = 23 MiB
    //     (its arena of deep members, the depth tables and the next-break positions live in buffers of the 62.1 n that the path does not use otherwise)
    // = 69.4 n + 23 MiB; every allocation is rounded up to 256 bytes (about forty of them: < 16 KiB).  The DC arrays (10 n) and the inverse
    // BWT's successor table (8 n) are allocated after the sort's temporaries are released and take their place.
    // tests/test_gpu_parity.py::test_workspace_accounting checks peak <= size on contexts sized exactly to their block, and
    // tests/test_gpu_fullsize.py checks the n-proportional term where the constant is negligible (peak - 64 MiB <= 69.4 n at 1e8 bytes).
    const size_t sort_temporaries = 62 * max_n + max_n / 8, io = 6 * max_n, on_top = max_n / 4 + max_n;
    return sort_temporaries + io + on_top + max_n / 4 /* headroom */ + (64u << 20);
}
struct ScopedCall {
    dk_ctx *c;
    explicit ScopedCall(dk_ctx *ctx) : c(ctx) {}
    ~ScopedCall() { end_call(c); }
};

// device-resident forward path up to the compact DC stream in pinned host memory
struct ForwardResult {
    uint32_t init[256];
    uint32_t origin = 0;
    size_t m = 0;
    const uint32_t *dist = nullptr;
    const uint8_t *sym = nullptr;
    const uint8_t *rank = nullptr;
    const uint32_t *run_end = nullptr;
    const std::atomic<size_t> *ready = nullptr;  // set: dist / sym are still arriving (forward_to_stream with overlap)
};

// slot < 0: the context's single staging buffer; otherwise the batch staging slot of that index
// overlap: the caller can work with a stream that is still arriving (ForwardResult::ready) and synchronises the stream afterwards
int forward_to_stream(dk_ctx *ctx, const uint8_t *d_text, size_t n, bool want_ctx_fields, ForwardResult *fr, int slot = -1, bool overlap = false) {
    hipStream_t st = ctx->stream;
    uint8_t *d_bwt = ctx->ws_alloc<uint8_t>(n);
    if (!d_bwt) return DK_E_NOMEM;
    {
        const size_t mark = ctx->ws_mark();
        uint32_t *d_sa = ctx->ws_alloc<uint32_t>(n);
        if (!d_sa) return DK_E_NOMEM;
        DK_TRY(bwt_forward_device(ctx, d_text, n, d_sa, d_bwt, &fr->origin));  // sets stats.ms_sa / ms_bwt
        ctx->ws_release(mark);
    }
    // the DC arrays take the place of the suffix sort's temporaries (only L survives the sort)
    uint32_t *d_dist = ctx->ws_alloc<uint32_t>(n);
    uint8_t *d_sym = ctx->ws_alloc<uint8_t>(n);
    uint8_t *d_rank = want_ctx_fields ? ctx->ws_alloc<uint8_t>(n) : nullptr;
    uint32_t *d_run_end = want_ctx_fields ? ctx->ws_alloc<uint32_t>(n) : nullptr;
    if (!d_dist || !d_sym || (want_ctx_fields && (!d_rank || !d_run_end))) return DK_E_NOMEM;
    Timer t3;
    DK_TRY(dc_encode_device(ctx, d_bwt, n, fr->init, d_dist, d_sym, d_rank, d_run_end, &fr->m));
    ctx->stats.ms_dc = t3.ms();
    Timer t4;
    const size_t m = fr->m;
    const size_t off_sym = 4 * m, off_rank = off_sym + ((m + 15) & ~size_t(15)), off_end = off_rank + ((m + 15) & ~size_t(15));
    char *stage;
    if (slot < 0) {
        DK_TRY(ctx->ensure_stage(off_end + 4 * m + 64));
        stage = ctx->h_stage;
    } else {
        DK_TRY(ctx->ensure_slot(static_cast<size_t>(slot), off_end + 4 * m + 64));
        stage = ctx->slots[static_cast<size_t>(slot)].h;
    }
    fr->ready = nullptr;
    if (DK_D2H_OVERLAP && overlap && !want_ctx_fields && m >= (size_t(1) << 21)) {
        // A large single block: the entropy stage starts on the first piece while the rest is on its way.  Every piece's two copies are
        // followed by a host function that moves the frontier (DcStream::ready); the stream is synchronised by the caller after coding.
        // If anything fails to enqueue, whatever was enqueued is waited for before the error is returned: its host functions point into
        // this context (d2h_marks, d2h_ready), which the next call reuses.
        constexpr size_t PIECES = 16;
        ctx->d2h_ready.store(0, std::memory_order_relaxed);
        ctx->d2h_marks.resize(PIECES);
        hipError_t e = hipSuccess;
        for (size_t c = 0; c < PIECES && e == hipSuccess; ++c) {
            const size_t k0 = m * c / PIECES, k1 = m * (c + 1) / PIECES;
            e = hipMemcpyAsync(stage + 4 * k0, d_dist + k0, 4 * (k1 - k0), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipMemcpyAsync(stage + off_sym + k0, d_sym + k0, k1 - k0, hipMemcpyDeviceToHost, st);
            ctx->d2h_marks[c] = dk_ctx::D2hMark{&ctx->d2h_ready, k1};
            if (e == hipSuccess)
                e = hipLaunchHostFunc(st, [](void *p) {
                    auto *mk = static_cast<dk_ctx::D2hMark *>(p);
                    mk->frontier->store(mk->value, std::memory_order_release);
                }, &ctx->d2h_marks[c]);
        }
        if (e != hipSuccess) {
            (void)hipStreamSynchronize(st);
            ctx->d2h_ready.store(dk::DC_STREAM_POISON, std::memory_order_release);
            return ctx->fail(DK_E_HIP, "D2H of the distance stream: %s", hipGetErrorString(e));
        }
        ctx->stats.ms_d2h = t4.ms();  // time to enqueue; block_encode_common adds what it waits for the stream after coding
        fr->dist = reinterpret_cast<const uint32_t *>(stage);
        fr->sym = reinterpret_cast<const uint8_t *>(stage + off_sym);
        fr->ready = &ctx->d2h_ready;
        return DK_OK;
    }
    DK_HIP(ctx, hipMemcpyAsync(stage, d_dist, 4 * m, hipMemcpyDeviceToHost, st));
    DK_HIP(ctx, hipMemcpyAsync(stage + off_sym, d_sym, m, hipMemcpyDeviceToHost, st));
    if (want_ctx_fields) {
        DK_HIP(ctx, hipMemcpyAsync(stage + off_rank, d_rank, m, hipMemcpyDeviceToHost, st));
        DK_HIP(ctx, hipMemcpyAsync(stage + off_end, d_run_end, 4 * m, hipMemcpyDeviceToHost, st));
    }
    DK_HIP(ctx, hipStreamSynchronize(st));
    ctx->stats.ms_d2h = t4.ms();
    fr->dist = reinterpret_cast<const uint32_t *>(stage);
    fr->sym = reinterpret_cast<const uint8_t *>(stage + off_sym);
    if (want_ctx_fields) {
        fr->rank = reinterpret_cast<const uint8_t *>(stage + off_rank);
        fr->run_end = reinterpret_cast<const uint32_t *>(stage + off_end);
    }
    return DK_OK;
}

unsigned block_flags(const uint32_t init[256], size_t n) {
    unsigned present = 0;
    for (int c = 0; c < 256; ++c) present += init[c] < n ? 1u : 0u;
    return (init[255] < n ? DK_FLAG_HAS_FF : 0u) | (present == 1 ? DK_FLAG_SINGLE_SYMBOL : 0u);
}

int block_encode_common(dk_ctx *ctx, int model_id, const uint8_t *d_text, size_t n, uint8_t *out, size_t out_cap, size_t *out_len) {
    if (model_max_block(model_id) == 0) return ctx->fail(DK_E_MODEL, "unknown model id %d", model_id);
    if (n > model_max_block(model_id))
        return ctx->fail(DK_E_MODEL, "model %d cannot code blocks of %zu bytes without losing bits (limit %llu)", model_id, n,
                         static_cast<unsigned long long>(model_max_block(model_id)));
    ForwardResult fr;
    DK_TRY(forward_to_stream(ctx, d_text, n, model_id == DK_MODEL_RAWDC, &fr, -1, true));
    ctx->last_flags = block_flags(fr.init, n);
    Timer t;
    DcStream s;
    s.n = n; s.init = fr.init; s.dist = fr.dist; s.sym = fr.sym; s.rank = fr.rank; s.run_end = fr.run_end; s.m = fr.m; s.origin = fr.origin;
    s.ready = fr.ready;
    dk::set_preferred_numa(ctx->numa_node);
    int rc = encode_block_stream(model_id, s, out, out_cap, out_len);
    double ms_wait = 0.0;
    if (fr.ready) {  // (a coder that gave up early must not leave copies and host functions of this call behind)
        Timer tw;
        const hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess && !rc) rc = DK_E_HIP;
        ms_wait = tw.ms();
        ctx->stats.ms_d2h += ms_wait;
    }
    ctx->stats.ms_entropy = t.ms() - ms_wait;
    ctx->stats.entropy_threads = static_cast<uint32_t>(dk::last_entropy_threads());
    ctx->stats.entropy_l3_group = dk::last_entropy_group();
    ctx->stats.entropy_l3_numa = static_cast<int16_t>(dk::last_entropy_group_numa());
    ctx->stats.gpu_numa = static_cast<int16_t>(ctx->numa_node);
    if (rc == DK_E_CAPACITY) return ctx->fail(rc, "output buffer of %zu bytes is too small", out_cap);
    if (rc) return ctx->fail(rc, "entropy stage failed (%d)", rc);
    return DK_OK;
}

int block_decode_common(dk_ctx *ctx, int model_id, const uint8_t *in, size_t in_len, size_t n, uint8_t *d_out) {
    if (model_id == DK_MODEL_RAWDC || model_max_block(model_id) == 0) return ctx->fail(DK_E_MODEL, "model %d cannot decode", model_id);
    hipStream_t st = ctx->stream;
    DK_TRY(ctx->ensure_stage(n + 64));
    uint8_t *h_bwt = reinterpret_cast<uint8_t *>(ctx->h_stage);
    uint32_t origin = 0;
    int single = 0;
    Timer t;
    int rc = decode_block_stream(model_id, in, in_len, n, h_bwt, &origin, &single, &ctx->last_consumed);
    ctx->stats.ms_entropy = t.ms();
    if (rc) return ctx->fail(rc, "stream does not decode (corrupt, truncated, wrong model/size, or a block containing byte 0xFF, "
                                 "which the reference format cannot represent: src/block/dc.rs:57-73)");
    Timer t2;
    uint8_t *d_bwt = ctx->ws_alloc<uint8_t>(n);
    if (!d_bwt) return DK_E_NOMEM;
    DK_HIP(ctx, hipMemcpyAsync(d_bwt, h_bwt, n, hipMemcpyHostToDevice, st));
    ctx->stats.ms_h2d = t2.ms();
    if (single) {
        // One-symbol block.  The reference mis-reads origin here (it skips the one sweep distance the encoder wrote) and
        // its bwt::decode then yields a single byte; the text is unambiguous, so return all n bytes.
        DK_HIP(ctx, hipMemcpyAsync(d_out, d_bwt, n, hipMemcpyDeviceToDevice, st));
        DK_HIP(ctx, hipStreamSynchronize(st));
        return DK_OK;
    }
    if (origin >= n) return ctx->fail(DK_E_STREAM, "decoded origin %u is outside the block", origin);
    Timer t3;
    DK_TRY(bwt_inverse_device(ctx, d_bwt, n, origin, d_out));
    ctx->stats.ms_ibwt = t3.ms();
    return DK_OK;
}

}  // namespace

extern "C" {

const char *dk_version(void) { return "dark_amd 0.1 (gfx950, HIP)"; }

int dk_ctx_create(int hip_device, size_t max_n, dk_ctx **out) {
    if (!out || max_n == 0 || max_n > 0x7FFFFFFEull) return DK_E_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return DK_E_NODEVICE;
    if (hip_device < 0 || hip_device >= count) return DK_E_NODEVICE;  // -1 would mean "CPU": there is no such backend
    dk_ctx *c = new (std::nothrow) dk_ctx();
    if (!c) return DK_E_NOMEM;
    c->device = hip_device;
    c->max_n = max_n;
    bool ok = hipSetDevice(hip_device) == hipSuccess && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    {   // the GPU's memory node: the host coder claims its last-level-cache group there first (entropy.cpp: l3_claim_order)
        char bdf[32] = {0};
        if (ok && hipDeviceGetPCIBusId(bdf, sizeof bdf, hip_device) == hipSuccess) {
            for (char *q = bdf; *q; ++q) *q = static_cast<char>(std::tolower(static_cast<unsigned char>(*q)));
            const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/numa_node";
            if (FILE *f = std::fopen(path.c_str(), "r")) {
                int node = -1;
                if (std::fscanf(f, "%d", &node) == 1) c->numa_node = node;
                std::fclose(f);
            }
        }
    }
    c->ws_size = workspace_bytes(max_n);
    ok = ok && hipMalloc(reinterpret_cast<void **>(&c->ws), c->ws_size) == hipSuccess;
    ok = ok && hipMalloc(reinterpret_cast<void **>(&c->d_mail), 1024 * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipHostMalloc(reinterpret_cast<void **>(&c->h_mail), 1024 * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess;
    // (on the context's own stream, and waited for: a memset on the null stream may still be on its way when the first call's kernels -- on a
    //  non-blocking stream, which the null stream does not order -- have written the mailbox: the first suffix sort of a fresh context then read an
    //  empty symbol histogram, one run in four of a test that starts with a tiny block)
    ok = ok && hipMemsetAsync(c->d_mail, 0, 1024 * sizeof(uint32_t), c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;
    for (hipEvent_t &e : c->round_ev) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        dk_ctx_destroy(c);
        return DK_E_NOMEM;
    }
    *out = c;
    return DK_OK;
}

void dk_ctx_destroy(dk_ctx *c) {
    if (!c) return;
    // a batch left open (an error path that never reached dk_batch_finish): its coding threads still read the pinned staging slots
    // freed below -- join them first
    if (c->live_batch) (void)dk_batch_finish(c->live_batch);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->round_ev) if (e) (void)hipEventDestroy(e);
    if (c->side_stream) { (void)hipStreamSynchronize(c->side_stream); (void)hipStreamDestroy(c->side_stream); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ws) (void)hipFree(c->ws);
    if (c->d_mail) (void)hipFree(c->d_mail);
    if (c->h_mail) (void)hipHostFree(c->h_mail);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    for (auto &sl : c->slots) if (sl.h) (void)hipHostFree(sl.h);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

size_t dk_capacity(const dk_ctx *ctx) { return ctx ? ctx->max_n : 0; }
size_t dk_last_consumed(const dk_ctx *ctx) { return ctx ? ctx->last_consumed : 0; }
unsigned dk_last_block_flags(const dk_ctx *ctx) { return ctx ? ctx->last_flags : 0; }
const char *dk_last_error(const dk_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

// ---- device-resident entry points ------------------------------------------------------------------------------------
int dk_dev_suffix_array(dk_ctx *ctx, const uint8_t *d_in, size_t n, uint32_t *d_sa_out) {
    DK_TRY(begin_call(ctx));
    ScopedCall sc(ctx);
    if (!d_in || !d_sa_out) return ctx->fail(DK_E_ARG, "null pointer");
    DK_TRY(check_n(ctx, n));
    Timer t;
    DK_TRY(suffix_array_device(ctx, d_in, n, d_sa_out));
    DK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stats.ms_sa = ctx->stats.ms_total = t.ms();
    return DK_OK;
}

int dk_dev_bwt_forward(dk_ctx *ctx, const uint8_t *d_in, size_t n, uint8_t *d_bwt_out, uint32_t *origin) {
    DK_TRY(begin_call(ctx));
    ScopedCall sc(ctx);
    if (!d_in || !d_bwt_out || !origin) return ctx->fail(DK_E_ARG, "null pointer");
    DK_TRY(check_n(ctx, n));
    Timer t;
    uint32_t *d_sa = ctx->ws_alloc<uint32_t>(n);
    if (!d_sa) return DK_E_NOMEM;
    DK_TRY(bwt_forward_device(ctx, d_in, n, d_sa, d_bwt_out, origin));
    ctx->stats.ms_total = t.ms();
    return DK_OK;
}

int dk_dev_bwt_inverse(dk_ctx *ctx, const uint8_t *d_bwt, size_t n, uint32_t origin, uint8_t *d_out) {
    DK_TRY(begin_call(ctx));
    ScopedCall sc(ctx);
    if (!d_bwt || !d_out) return ctx->fail(DK_E_ARG, "null pointer");
    DK_TRY(check_n(ctx, n));
    if (origin >= n) return ctx->fail(DK_E_ARG, "origin %u outside the block of %zu bytes", origin, n);
    Timer t;
    DK_TRY(bwt_inverse_device(ctx, d_bwt, n, origin, d_out));
    ctx->stats.ms_ibwt = ctx->stats.ms_total = t.ms();
    return DK_OK;
}

int dk_dev_dc_encode(dk_ctx *ctx, const uint8_t *d_bwt, size_t n, uint32_t init[256], uint32_t *d_dist, uint8_t *d_sym, uint8_t *d_rank,
                     size_t *m) {
    DK_TRY(begin_call(ctx));
    ScopedCall sc(ctx);
    if (!d_bwt || !init || !d_dist || !d_sym || !m) return ctx->fail(DK_E_ARG, "null pointer");
    DK_TRY(check_n(ctx, n));
    Timer t;
    DK_TRY(dc_encode_device(ctx, d_bwt, n, init, d_dist, d_sym, d_rank, nullptr, m));
    ctx->stats.ms_dc = ctx->stats.ms_total = t.ms();
    return DK_OK;
}

int dk_dev_block_encode(dk_ctx *ctx, int model_id, const uint8_t *d_in, size_t n, uint8_t *out, size_t out_cap, size_t *out_len) {
    DK_TRY(begin_call(ctx));
    ScopedCall sc(ctx);
    if (!d_in || !out || !out_len) return ctx->fail(DK_E_ARG, "null pointer");
    DK_TRY(check_n(ctx, n));
    Timer t;
    ctx->stats.ms_h2d = 0;
    int rc = block_encode_common(ctx, model_id, d_in, n, out, out_cap, out_len);
    ctx->stats.ms_total = t.ms();
    return rc;
}

int dk_dev_block_decode(dk_ctx *ctx, int model_id, const uint8_t *in, size_t in_len, size_t n, uint8_t *d_out) {
    DK_TRY(begin_call(ctx));
    ScopedCall sc(ctx);
    if (!in || !d_out) return ctx->fail(DK_E_ARG, "null pointer");
    DK_TRY(check_n(ctx, n));
    Timer t;
    int rc = block_decode_common(ctx, model_id, in, in_len, n, d_out);
    ctx->stats.ms_total = t.ms();
    return rc;
}

// Batch of independent blocks on one GPU: the device stages run block after block on the context's stream while a pool of host
// threads codes the distance streams of the blocks already done (the "one block per core, pipelined against the GPU work of
// the next block" deployment of SURVEY.md section 7.8).  Every out[i] is byte-identical to a dk_dev_block_encode of block i.
// Streaming form: dk_batch_begin starts the coding threads, dk_batch_push runs the device stages of one more block on the calling
// thread (it waits while every staging slot is busy) and queues its coding, dk_batch_finish waits for the coders.
}  // extern "C"

struct dk_batch {
    dk_ctx *ctx = nullptr;
    int model_id = 0;
    struct Job { size_t block; int slot; ForwardResult fr; size_t n; uint8_t *out; size_t cap; size_t *out_len; };
    std::mutex mu;
    std::condition_variable cv_job, cv_slot;
    std::deque<Job> queue;
    std::vector<int> free_slots;
    std::vector<std::thread> pool;
    bool done = false;
    size_t pushed = 0;
    int first_rc = DK_OK;
    size_t first_bad = 0;
    void worker() {
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return done || !queue.empty(); });
                if (queue.empty()) return;
                job = queue.front();
                queue.pop_front();
            }
            DcStream s;
            s.n = job.n; s.init = job.fr.init; s.dist = job.fr.dist; s.sym = job.fr.sym; s.rank = job.fr.rank;
            s.run_end = job.fr.run_end; s.m = job.fr.m; s.origin = job.fr.origin;
            const int rc = encode_block_stream(model_id, s, job.out, job.cap, job.out_len, 1);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (rc != DK_OK && first_rc == DK_OK) { first_rc = rc; first_bad = job.block; }
                free_slots.push_back(job.slot);
            }
            cv_slot.notify_one();
        }
    }
};

extern "C" {

int dk_batch_begin(dk_ctx *ctx, int model_id, int host_threads, dk_batch **out) {
    DK_TRY(begin_call(ctx));
    if (!out) return ctx->fail(DK_E_ARG, "null pointer");
    *out = nullptr;
    if (model_max_block(model_id) == 0) return ctx->fail(DK_E_MODEL, "unknown model id %d", model_id);
    dk_batch *b = new (std::nothrow) dk_batch();
    if (!b) return DK_E_NOMEM;
    b->ctx = ctx;
    b->model_id = model_id;
    const size_t workers = static_cast<size_t>(std::max(1, host_threads));
    for (size_t k = 0; k < workers + 1; ++k) b->free_slots.push_back(static_cast<int>(k));
    for (size_t w = 0; w < workers; ++w) b->pool.emplace_back([b] { b->worker(); });
    ctx->live_batch = b;
    *out = b;
    return DK_OK;
}

int dk_batch_push(dk_batch *b, const uint8_t *d_in, size_t n, uint8_t *out, size_t out_cap, size_t *out_len) {
    if (!b) return DK_E_ARG;
    dk_ctx *ctx = b->ctx;
    if (!d_in || !out || !out_len) return ctx->fail(DK_E_ARG, "null pointer in block %zu", b->pushed);
    DK_TRY(check_n(ctx, n));
    if (n > model_max_block(b->model_id)) return ctx->fail(DK_E_MODEL, "model %d cannot code a block of %zu bytes", b->model_id, n);
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(DK_E_HIP, "hipSetDevice(%d) failed", ctx->device);
    int slot;
    {
        std::unique_lock<std::mutex> lk(b->mu);
        b->cv_slot.wait(lk, [&] { return !b->free_slots.empty(); });
        slot = b->free_slots.back();
        b->free_slots.pop_back();
    }
    dk_batch::Job job;
    job.block = b->pushed++;
    job.slot = slot;
    job.n = n; job.out = out; job.cap = out_cap; job.out_len = out_len;
    ctx->ws_reset();
    const int rc = forward_to_stream(ctx, d_in, n, b->model_id == DK_MODEL_RAWDC, &job.fr, slot);
    if (ctx->profiling) ctx->prof_collect();
    if (rc != DK_OK) {
        std::lock_guard<std::mutex> lk(b->mu);
        b->free_slots.push_back(slot);
        return rc;
    }
    {
        std::lock_guard<std::mutex> lk(b->mu);
        b->queue.push_back(job);
    }
    b->cv_job.notify_one();
    return DK_OK;
}

int dk_batch_finish(dk_batch *b) {
    if (!b) return DK_E_ARG;
    {
        std::lock_guard<std::mutex> lk(b->mu);
        b->done = true;
    }
    b->cv_job.notify_all();
    for (auto &th : b->pool) th.join();
    b->ctx->live_batch = nullptr;
    int rc = b->first_rc;
    if (rc != DK_OK) rc = b->ctx->fail(rc, "entropy stage of block %zu failed (%d)", b->first_bad, rc);
    delete b;
    return rc;
}

int dk_dev_batch_encode(dk_ctx *ctx, int model_id, size_t count, const uint8_t *const *d_in, const size_t *n, uint8_t *const *out,
                        const size_t *out_cap, size_t *out_len, int host_threads) {
    if (!ctx) return DK_E_ARG;
    if (!d_in || !n || !out || !out_cap || !out_len || count == 0) { ctx->err.clear(); return ctx->fail(DK_E_ARG, "null pointer or empty batch"); }
    Timer t;
    dk_batch *b = nullptr;
    DK_TRY(dk_batch_begin(ctx, model_id, std::max(1, std::min<int>(host_threads, static_cast<int>(count))), &b));
    int rc = DK_OK;
    for (size_t i = 0; i < count && rc == DK_OK; ++i) rc = dk_batch_push(b, d_in[i], n[i], out[i], out_cap[i], &out_len[i]);
    const int rc2 = dk_batch_finish(b);
    ctx->stats.ms_total = t.ms();
    return rc != DK_OK ? rc : rc2;
}

// Inverse of dk_dev_batch_encode: host threads decode the streams (range decoder + dc::decode, serial per block) into pinned
// slots while the calling thread uploads finished BWTs and runs the inverse BWT on the GPU.
int dk_dev_batch_decode(dk_ctx *ctx, int model_id, size_t count, const uint8_t *const *in, const size_t *in_len, const size_t *n,
                        uint8_t *const *d_out, int host_threads) {
    DK_TRY(begin_call(ctx));
    ScopedCall sc(ctx);
    if (!in || !in_len || !n || !d_out || count == 0) return ctx->fail(DK_E_ARG, "null pointer or empty batch");
    if (model_id == DK_MODEL_RAWDC || model_max_block(model_id) == 0) return ctx->fail(DK_E_MODEL, "model %d cannot decode", model_id);
    size_t max_n = 0;
    for (size_t i = 0; i < count; ++i) {
        if (!in[i] || !d_out[i]) return ctx->fail(DK_E_ARG, "null pointer in block %zu", i);
        DK_TRY(check_n(ctx, n[i]));
        max_n = std::max(max_n, n[i]);
    }
    Timer t;
    const size_t workers = static_cast<size_t>(std::max(1, std::min<int>(host_threads, static_cast<int>(count))));
    const size_t nslots = workers + 1;
    for (size_t k = 0; k < nslots; ++k) DK_TRY(ctx->ensure_slot(k, max_n + 64));
    struct Done { size_t block; int slot; uint32_t origin; int single; int rc; };
    std::mutex mu;
    std::condition_variable cv_slot, cv_done;
    std::vector<int> free_slots;
    for (size_t k = 0; k < nslots; ++k) free_slots.push_back(static_cast<int>(k));
    std::deque<Done> finished;
    size_t next_block = 0;
    auto worker = [&] {
        for (;;) {
            size_t block;
            int slot;
            {
                std::unique_lock<std::mutex> lk(mu);
                if (next_block >= count) return;
                block = next_block++;
                cv_slot.wait(lk, [&] { return !free_slots.empty(); });
                slot = free_slots.back();
                free_slots.pop_back();
            }
            Done d{block, slot, 0, 0, DK_OK};
            d.rc = decode_block_stream(model_id, in[block], in_len[block], n[block], reinterpret_cast<uint8_t *>(ctx->slots[slot].h), &d.origin,
                                       &d.single);
            {
                std::lock_guard<std::mutex> lk(mu);
                finished.push_back(d);
            }
            cv_done.notify_one();
        }
    };
    std::vector<std::thread> pool;
    for (size_t w = 0; w < workers; ++w) pool.emplace_back(worker);
    int rc = DK_OK;
    size_t bad_block = 0;
    hipStream_t st = ctx->stream;
    for (size_t done_count = 0; done_count < count; ++done_count) {
        Done d;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [&] { return !finished.empty(); });
            d = finished.front();
            finished.pop_front();
        }
        if (rc == DK_OK && d.rc != DK_OK) { rc = d.rc; bad_block = d.block; }
        if (rc == DK_OK) {
            ctx->ws_reset();
            const size_t nb = n[d.block];
            uint8_t *d_bwt = ctx->ws_alloc<uint8_t>(nb);
            hipError_t e = d_bwt ? hipMemcpyAsync(d_bwt, ctx->slots[d.slot].h, nb, hipMemcpyHostToDevice, st) : hipErrorOutOfMemory;
            if (e == hipSuccess) e = hipStreamSynchronize(st);  // the slot may be refilled from here on
            if (e != hipSuccess) { rc = ctx->fail(DK_E_HIP, "batch decode upload: %s", hipGetErrorString(e)); }
            {
                std::lock_guard<std::mutex> lk(mu);
                free_slots.push_back(d.slot);
            }
            cv_slot.notify_one();
            if (rc == DK_OK) {
                if (d.single) {
                    if (hipMemcpyAsync(d_out[d.block], d_bwt, nb, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = ctx->fail(DK_E_HIP, "copy failed");
                } else if (d.origin >= nb) {
                    rc = ctx->fail(DK_E_STREAM, "decoded origin %u is outside block %zu", d.origin, d.block);
                } else {
                    rc = bwt_inverse_device(ctx, d_bwt, nb, d.origin, d_out[d.block]);
                }
            }
        } else {
            std::lock_guard<std::mutex> lk(mu);
            free_slots.push_back(d.slot);
            cv_slot.notify_one();
        }
    }
    for (auto &th : pool) th.join();
    (void)hipStreamSynchronize(st);
    ctx->stats.ms_total = t.ms();
    if (rc != DK_OK && ctx->err.empty()) return ctx->fail(rc, "stream of block %zu does not decode (%d)", bad_block, rc);
    return rc;
}

// ---- host-pointer entry points: stage in, run the device path, stage out ---------------------------------------------
int dk_suffix_array(dk_ctx *ctx, const uint8_t *in, size_t n, uint32_t *sa_out) {
    DK_TRY(begin_call(ctx));
    ScopedCall sc(ctx);
    if (!in || !sa_out) return ctx->fail(DK_E_ARG, "null pointer");
    DK_TRY(check_n(ctx, n));
    Timer t;
    uint8_t *d_text = ctx->ws_alloc<uint8_t>(n);
    uint32_t *d_sa = ctx->ws_alloc<uint32_t>(n);
    if (!d_text || !d_sa) return DK_E_NOMEM;
    DK_HIP(ctx, hipMemcpyAsync(d_text, in, n, hipMemcpyHostToDevice, ctx->stream));
    DK_TRY(suffix_array_device(ctx, d_text, n, d_sa));
    DK_HIP(ctx, hipMemcpyAsync(sa_out, d_sa, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    DK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stats.ms_total = t.ms();
    return DK_OK;
}

int dk_bwt_forward(dk_ctx *ctx, const uint8_t *in, size_t n, uint8_t *bwt_out, uint32_t *origin) {
    DK_TRY(begin_call(ctx));
    ScopedCall sc(ctx);
    if (!in || !bwt_out || !origin) return ctx->fail(DK_E_ARG, "null pointer");
    DK_TRY(check_n(ctx, n));
    Timer t;
    uint8_t *d_text = ctx->ws_alloc<uint8_t>(n);
    uint8_t *d_bwt = ctx->ws_alloc<uint8_t>(n);
    uint32_t *d_sa = ctx->ws_alloc<uint32_t>(n);
    if (!d_text || !d_bwt || !d_sa) return DK_E_NOMEM;
    DK_HIP(ctx, hipMemcpyAsync(d_text, in, n, hipMemcpyHostToDevice, ctx->stream));
    DK_TRY(bwt_forward_device(ctx, d_text, n, d_sa, d_bwt, origin));
    DK_HIP(ctx, hipMemcpyAsync(bwt_out, d_bwt, n, hipMemcpyDeviceToHost, ctx->stream));
    DK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stats.ms_total = t.ms();
    return DK_OK;
}

int dk_bwt_inverse(dk_ctx *ctx, const uint8_t *bwt, size_t n, uint32_t origin, uint8_t *out) {
    DK_TRY(begin_call(ctx));
    ScopedCall sc(ctx);
    if (!bwt || !out) return ctx->fail(DK_E_ARG, "null pointer");
    DK_TRY(check_n(ctx, n));
    if (origin >= n) return ctx->fail(DK_E_ARG, "origin %u outside the block of %zu bytes", origin, n);
    Timer t;
    uint8_t *d_bwt = ctx->ws_alloc<uint8_t>(n);
    uint8_t *d_out = ctx->ws_alloc<uint8_t>(n);
    if (!d_bwt || !d_out) return DK_E_NOMEM;
    DK_HIP(ctx, hipMemcpyAsync(d_bwt, bwt, n, hipMemcpyHostToDevice, ctx->stream));
    DK_TRY(bwt_inverse_device(ctx, d_bwt, n, origin, d_out));
    DK_HIP(ctx, hipMemcpyAsync(out, d_out, n, hipMemcpyDeviceToHost, ctx->stream));
    DK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stats.ms_total = t.ms();
    return DK_OK;
}

int dk_dc_encode(dk_ctx *ctx, const uint8_t *bwt, size_t n, uint32_t init[256], uint32_t *dist, uint8_t *sym, uint8_t *rank, size_t *m) {
    DK_TRY(begin_call(ctx));
    ScopedCall sc(ctx);
    if (!bwt || !init || !dist || !sym || !m) return ctx->fail(DK_E_ARG, "null pointer");
    DK_TRY(check_n(ctx, n));
    Timer t;
    uint8_t *d_bwt = ctx->ws_alloc<uint8_t>(n);
    uint32_t *d_dist = ctx->ws_alloc<uint32_t>(n);
    uint8_t *d_sym = ctx->ws_alloc<uint8_t>(n);
    uint8_t *d_rank = ctx->ws_alloc<uint8_t>(n);
    if (!d_bwt || !d_dist || !d_sym || !d_rank) return DK_E_NOMEM;
    DK_HIP(ctx, hipMemcpyAsync(d_bwt, bwt, n, hipMemcpyHostToDevice, ctx->stream));
    DK_TRY(dc_encode_device(ctx, d_bwt, n, init, d_dist, d_sym, d_rank, nullptr, m));
    DK_HIP(ctx, hipMemcpyAsync(dist, d_dist, *m * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    DK_HIP(ctx, hipMemcpyAsync(sym, d_sym, *m, hipMemcpyDeviceToHost, ctx->stream));
    if (rank) DK_HIP(ctx, hipMemcpyAsync(rank, d_rank, *m, hipMemcpyDeviceToHost, ctx->stream));
    DK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stats.ms_total = t.ms();
    return DK_OK;
}

int dk_dc_decode(dk_ctx *ctx, const uint32_t init[256], const uint32_t *dist, size_t m, uint8_t *bwt_out, size_t n, size_t *consumed) {
    if (!init || (!dist && m) || !bwt_out || n == 0) return ctx ? ctx->fail(DK_E_ARG, "null pointer or empty block") : DK_E_ARG;
    int rc = dc_decode_array(init, dist, m, bwt_out, n, consumed);
    if (rc && ctx) return ctx->fail(rc, "distance stream does not rebuild a BWT of %zu bytes", n);
    return rc;
}

int dk_block_encode(dk_ctx *ctx, int model_id, const uint8_t *in, size_t n, uint8_t *out, size_t out_cap, size_t *out_len) {
    DK_TRY(begin_call(ctx));
    ScopedCall sc(ctx);
    if (!in || !out || !out_len) return ctx->fail(DK_E_ARG, "null pointer");
    DK_TRY(check_n(ctx, n));
    Timer t;
    uint8_t *d_text = ctx->ws_alloc<uint8_t>(n);
    if (!d_text) return DK_E_NOMEM;
    DK_HIP(ctx, hipMemcpyAsync(d_text, in, n, hipMemcpyHostToDevice, ctx->stream));
    DK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stats.ms_h2d = t.ms();
    int rc = block_encode_common(ctx, model_id, d_text, n, out, out_cap, out_len);
    ctx->stats.ms_total = t.ms();
    return rc;
}

int dk_block_decode(dk_ctx *ctx, int model_id, const uint8_t *in, size_t in_len, size_t n, uint8_t *out) {
    DK_TRY(begin_call(ctx));
    ScopedCall sc(ctx);
    if (!in || !out) return ctx->fail(DK_E_ARG, "null pointer");
    DK_TRY(check_n(ctx, n));
    Timer t;
    uint8_t *d_out = ctx->ws_alloc<uint8_t>(n);
    if (!d_out) return DK_E_NOMEM;
    DK_TRY(block_decode_common(ctx, model_id, in, in_len, n, d_out));
    Timer t2;
    DK_HIP(ctx, hipMemcpyAsync(out, d_out, n, hipMemcpyDeviceToHost, ctx->stream));
    DK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stats.ms_d2h = t2.ms();
    ctx->stats.ms_total = t.ms();
    return DK_OK;
}

// block::raw::Encoder::encode (src/block/raw.rs:35-59): SA -> BWT on the GPU; origin + L then go through the RawModel -- the dump model
// Out (into `dump`) or the bbb coding model (into `out`; host, bit-serial)
int dk_raw_block_encode(dk_ctx *ctx, int raw_model, const uint8_t *in, size_t n, uint8_t *out, size_t out_cap, size_t *out_len, uint8_t *dump,
                        size_t dump_cap, size_t *dump_len) {
    DK_TRY(begin_call(ctx));
    ScopedCall sc(ctx);
    if (!in || !out || !out_len) return ctx->fail(DK_E_ARG, "null pointer");
    if (raw_model != DK_RAWMODEL_OUT && raw_model != DK_RAWMODEL_BBB) return ctx->fail(DK_E_MODEL, "unknown raw model %d", raw_model);
    if (raw_model == DK_RAWMODEL_OUT && (!dump || !dump_len)) return ctx->fail(DK_E_ARG, "the Out model needs a dump buffer");
    DK_TRY(check_n(ctx, n));
    if (raw_model == DK_RAWMODEL_OUT && (dump_cap < n + 4 || out_cap < 4)) return ctx->fail(DK_E_CAPACITY, "dump needs n + 4 bytes, out 4 bytes");
    Timer t;
    uint8_t *d_text = ctx->ws_alloc<uint8_t>(n);
    uint8_t *d_bwt = ctx->ws_alloc<uint8_t>(n);
    uint32_t *d_sa = ctx->ws_alloc<uint32_t>(n);
    if (!d_text || !d_bwt || !d_sa) return DK_E_NOMEM;
    DK_HIP(ctx, hipMemcpyAsync(d_text, in, n, hipMemcpyHostToDevice, ctx->stream));
    uint32_t origin = 0;
    DK_TRY(bwt_forward_device(ctx, d_text, n, d_sa, d_bwt, &origin));
    if (raw_model == DK_RAWMODEL_OUT) {
        DK_HIP(ctx, hipMemcpyAsync(dump + 4, d_bwt, n, hipMemcpyDeviceToHost, ctx->stream));
        DK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < 4; ++i) dump[i] = static_cast<uint8_t>(origin >> (24 - 8 * i));  // src/block/raw.rs:48-51
        *dump_len = n + 4;
        std::memset(out, 0, 4);  // ari::Encoder::finish of a coder nothing was coded with: low = 0, four bytes
        *out_len = 4;
    } else {
        DK_TRY(ctx->ensure_stage(n + 64));
        uint8_t *h_bwt = reinterpret_cast<uint8_t *>(ctx->h_stage);
        DK_HIP(ctx, hipMemcpyAsync(h_bwt, d_bwt, n, hipMemcpyDeviceToHost, ctx->stream));
        DK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        Timer te;
        const int rc = raw_bbb_encode_stream(h_bwt, n, origin, out, out_cap, out_len);
        ctx->stats.ms_entropy = te.ms();
        if (rc == DK_E_CAPACITY) return ctx->fail(rc, "output buffer of %zu bytes is too small", out_cap);
        if (rc) return ctx->fail(rc, "bbb coding failed (%d)", rc);
        if (dump_len) *dump_len = 0;
    }
    ctx->stats.ms_total = t.ms();
    return DK_OK;
}

int dk_raw_block_decode(dk_ctx *ctx, int raw_model, const uint8_t *in, size_t in_len, size_t n, uint8_t *out) {
    DK_TRY(begin_call(ctx));
    ScopedCall sc(ctx);
    if (!in || !out) return ctx->fail(DK_E_ARG, "null pointer");
    if (raw_model != DK_RAWMODEL_OUT && raw_model != DK_RAWMODEL_BBB) return ctx->fail(DK_E_MODEL, "unknown raw model %d", raw_model);
    DK_TRY(check_n(ctx, n));
    if (raw_model == DK_RAWMODEL_OUT) {
        std::memset(out, 0, n);  // Out::decode returns symbol 0 ("not supported", src/model/raw.rs:71-75): origin 0, a BWT of zeros -> n zeros
        return DK_OK;
    }
    Timer t;
    DK_TRY(ctx->ensure_stage(n + 64));
    uint8_t *h_bwt = reinterpret_cast<uint8_t *>(ctx->h_stage);
    uint32_t origin = 0;
    const int rc = raw_bbb_decode_stream(in, in_len, n, h_bwt, &origin, &ctx->last_consumed);
    ctx->stats.ms_entropy = t.ms();
    if (rc) return ctx->fail(rc, "bbb stream does not decode");
    if (origin >= n) return ctx->fail(DK_E_STREAM, "decoded origin %u is outside the block", origin);
    uint8_t *d_bwt = ctx->ws_alloc<uint8_t>(n);
    uint8_t *d_out = ctx->ws_alloc<uint8_t>(n);
    if (!d_bwt || !d_out) return DK_E_NOMEM;
    DK_HIP(ctx, hipMemcpyAsync(d_bwt, h_bwt, n, hipMemcpyHostToDevice, ctx->stream));
    DK_TRY(bwt_inverse_device(ctx, d_bwt, n, origin, d_out));
    DK_HIP(ctx, hipMemcpyAsync(out, d_out, n, hipMemcpyDeviceToHost, ctx->stream));
    DK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stats.ms_total = t.ms();
    return DK_OK;
}

}  // extern "C" (the helper below is a template)

// ---- several GPUs in one call: one host thread + one context per listed device ---------------------------------------------------------
namespace {
struct DeviceBuffers {  // input / output blocks of one sub-batch in HBM
    std::vector<uint8_t *> ptr;
    ~DeviceBuffers() { for (uint8_t *p : ptr) if (p) (void)hipFree(p); }
};
template <class Work>
int run_per_device(const int *devices, int ndev, size_t count, const size_t *n, char *err, size_t err_cap, Work work) {
    if (!devices || ndev <= 0 || count == 0 || !n) return DK_E_ARG;
    std::vector<int> rcs(static_cast<size_t>(ndev), DK_OK);
    std::vector<std::string> msgs(static_cast<size_t>(ndev));
    std::vector<std::thread> threads;
    for (int r = 0; r < ndev; ++r) {
        threads.emplace_back([&, r] {
            std::vector<size_t> mine;
            size_t max_n = 0;
            for (size_t i = static_cast<size_t>(r); i < count; i += static_cast<size_t>(ndev)) { mine.push_back(i); max_n = std::max(max_n, n[i]); }
            if (mine.empty()) return;
            dk_ctx *ctx = nullptr;
            int rc = dk_ctx_create(devices[r], max_n, &ctx);
            if (rc != DK_OK) { rcs[r] = rc; msgs[r] = "dk_ctx_create failed on device " + std::to_string(devices[r]); return; }
            rc = work(ctx, mine);
            if (rc != DK_OK) { rcs[r] = rc; msgs[r] = dk_last_error(ctx); }
            dk_ctx_destroy(ctx);
        });
    }
    for (auto &t : threads) t.join();
    for (int r = 0; r < ndev; ++r)
        if (rcs[r] != DK_OK) {
            if (err && err_cap) snprintf(err, err_cap, "device %d: %s", devices[r], msgs[r].c_str());
            return rcs[r];
        }
    if (err && err_cap) err[0] = 0;
    return DK_OK;
}
}  // namespace

extern "C" {

int dk_multi_block_encode(const int *devices, int ndev, int model_id, size_t count, const uint8_t *const *in, const size_t *n, uint8_t *const *out,
                          const size_t *out_cap, size_t *out_len, int host_threads_per_gpu, char *err, size_t err_cap) {
    if (!in || !out || !out_cap || !out_len) return DK_E_ARG;
    const size_t group = static_cast<size_t>(std::max(1, host_threads_per_gpu)) + 1;  // blocks resident in HBM per sub-batch
    return run_per_device(devices, ndev, count, n, err, err_cap, [&](dk_ctx *ctx, const std::vector<size_t> &mine) -> int {
        for (size_t lo = 0; lo < mine.size(); lo += group) {
            const size_t cnt = std::min(group, mine.size() - lo);
            DeviceBuffers db;
            std::vector<const uint8_t *> d_in(cnt);
            std::vector<size_t> ns(cnt), caps(cnt), lens(cnt);
            std::vector<uint8_t *> outs(cnt);
            for (size_t k = 0; k < cnt; ++k) {
                const size_t i = mine[lo + k];
                uint8_t *p = nullptr;
                if (!in[i] || !out[i] || n[i] == 0) return ctx->fail(DK_E_ARG, "null pointer or empty block %zu", i);
                if (hipMalloc(reinterpret_cast<void **>(&p), n[i]) != hipSuccess) return ctx->fail(DK_E_NOMEM, "hipMalloc of block %zu failed", i);
                db.ptr.push_back(p);
                if (hipMemcpyAsync(p, in[i], n[i], hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return ctx->fail(DK_E_HIP, "upload of block %zu failed", i);  // (ordered with the kernels that read it)
                d_in[k] = p; ns[k] = n[i]; caps[k] = out_cap[i]; outs[k] = out[i];
            }
            const int rc = dk_dev_batch_encode(ctx, model_id, cnt, d_in.data(), ns.data(), outs.data(), caps.data(), lens.data(), host_threads_per_gpu);
            if (rc != DK_OK) return rc;
            for (size_t k = 0; k < cnt; ++k) out_len[mine[lo + k]] = lens[k];
        }
        return DK_OK;
    });
}

int dk_multi_block_decode(const int *devices, int ndev, int model_id, size_t count, const uint8_t *const *in, const size_t *in_len, const size_t *n,
                          uint8_t *const *out, int host_threads_per_gpu, char *err, size_t err_cap) {
    if (!in || !in_len || !out) return DK_E_ARG;
    const size_t group = static_cast<size_t>(std::max(1, host_threads_per_gpu)) + 1;
    return run_per_device(devices, ndev, count, n, err, err_cap, [&](dk_ctx *ctx, const std::vector<size_t> &mine) -> int {
        for (size_t lo = 0; lo < mine.size(); lo += group) {
            const size_t cnt = std::min(group, mine.size() - lo);
            DeviceBuffers db;
            std::vector<const uint8_t *> ins(cnt);
            std::vector<size_t> lens(cnt), ns(cnt);
            std::vector<uint8_t *> d_out(cnt);
            for (size_t k = 0; k < cnt; ++k) {
                const size_t i = mine[lo + k];
                uint8_t *p = nullptr;
                if (!in[i] || !out[i] || n[i] == 0) return ctx->fail(DK_E_ARG, "null pointer or empty block %zu", i);
                if (hipMalloc(reinterpret_cast<void **>(&p), n[i]) != hipSuccess) return ctx->fail(DK_E_NOMEM, "hipMalloc of block %zu failed", i);
                db.ptr.push_back(p);
                ins[k] = in[i]; lens[k] = in_len[i]; ns[k] = n[i]; d_out[k] = p;
            }
            const int rc = dk_dev_batch_decode(ctx, model_id, cnt, ins.data(), lens.data(), ns.data(), d_out.data(), host_threads_per_gpu);
            if (rc != DK_OK) return rc;
            for (size_t k = 0; k < cnt; ++k)
                if (hipMemcpy(out[mine[lo + k]], d_out[k], ns[k], hipMemcpyDeviceToHost) != hipSuccess) return ctx->fail(DK_E_HIP, "download of block %zu failed", mine[lo + k]);
        }
        return DK_OK;
    });
}

// ---- model / coder level (host only) --------------------------------------------------------------------------------------
int dk_model_encode(int model_id, const uint32_t *dist, const uint8_t *sym, size_t m, uint8_t *out, size_t out_cap, size_t *out_len) {
    if ((!dist || !sym) && m) return DK_E_ARG;
    if (!out || !out_len) return DK_E_ARG;
    return model_encode_stream(model_id, dist, sym, m, out, out_cap, out_len);
}
int dk_model_decode(int model_id, const uint8_t *in, size_t in_len, const uint8_t *sym, size_t m, uint32_t *dist) {
    if (!in || ((!sym || !dist) && m)) return DK_E_ARG;
    return model_decode_stream(model_id, in, in_len, sym, m, dist);
}
int dk_bitcoder_encode(const uint8_t *bits, const uint16_t *flat, size_t nbits, uint8_t *out, size_t out_cap, size_t *out_len) {
    if (((!bits || !flat) && nbits) || !out || !out_len) return DK_E_ARG;
    return bitcoder_encode(bits, flat, nbits, out, out_cap, out_len);
}
int dk_bitcoder_decode(const uint8_t *in, size_t in_len, const uint16_t *flat, size_t nbits, uint8_t *bits) {
    if (!in || ((!flat || !bits) && nbits)) return DK_E_ARG;
    return bitcoder_decode(in, in_len, flat, nbits, bits);
}
int dk_stream_encode(int model_id, size_t n, const uint32_t init[256], const uint32_t *dist, const uint8_t *sym, const uint8_t *rank,
                     const uint32_t *run_end, size_t m, uint32_t origin, uint8_t *out, size_t out_cap, size_t *out_len) {
    DcStream s;
    s.n = n; s.init = init; s.dist = dist; s.sym = sym; s.rank = rank; s.run_end = run_end; s.m = m; s.origin = origin;
    return encode_block_stream(model_id, s, out, out_cap, out_len);
}
int dk_stream_decode(int model_id, const uint8_t *in, size_t in_len, size_t n, uint8_t *bwt_out, uint32_t *origin, int *single_symbol,
                     size_t *consumed) {
    int single = 0;
    int rc = decode_block_stream(model_id, in, in_len, n, bwt_out, origin, &single, consumed);
    if (single_symbol) *single_symbol = single;
    return rc;
}

int dk_raw_stream_encode(int raw_model, const uint8_t *bwt, size_t n, uint32_t origin, uint8_t *out, size_t out_cap, size_t *out_len) {
    if (!bwt || !out || !out_len || n == 0) return DK_E_ARG;
    if (raw_model != DK_RAWMODEL_BBB) return DK_E_MODEL;
    return raw_bbb_encode_stream(bwt, n, origin, out, out_cap, out_len);
}
int dk_raw_stream_decode(int raw_model, const uint8_t *in, size_t in_len, size_t n, uint8_t *bwt_out, uint32_t *origin, size_t *consumed) {
    if (!in || !bwt_out || !origin || n == 0) return DK_E_ARG;
    if (raw_model != DK_RAWMODEL_BBB) return DK_E_MODEL;
    return raw_bbb_decode_stream(in, in_len, n, bwt_out, origin, consumed);
}

// ---- measurement ----------------------------------------------------------------------------------------------------------
int dk_set_profiling(dk_ctx *ctx, int enabled) {
    if (!ctx) return DK_E_ARG;
    ctx->profiling = enabled != 0;
    return DK_OK;
}
int dk_stats_reset(dk_ctx *ctx) {
    if (!ctx) return DK_E_ARG;
    ctx->stats = dk_stats{};
    return DK_OK;
}
int dk_get_stats(const dk_ctx *ctx, dk_stats *out) {
    if (!ctx || !out) return DK_E_ARG;
    *out = ctx->stats;
    out->ws_peak_bytes = ctx->ws_peak;
    out->ws_size_bytes = ctx->ws_size;
    return DK_OK;
}
const char *dk_kernel_name(int slot) { return kernel_slot_name(slot); }

// ---- host coding threads ------------------------------------------------------------------------------------------------------
int dk_set_entropy_threads(int mode) { return set_entropy_thread_mode(mode); }
int dk_host_l3_groups(int min_cores) { return host_l3_groups(min_cores); }
int dk_dbg_l3_claim_order(const int *group_numa, int ngroups, int own, int preferred_numa, int *order_out) {
    if (!group_numa || !order_out || ngroups <= 0 || own < 0 || own >= ngroups) return DK_E_ARG;
    const std::vector<size_t> order = l3_claim_order(std::vector<int>(group_numa, group_numa + ngroups), static_cast<size_t>(own), preferred_numa);
    for (size_t i = 0; i < order.size(); ++i) order_out[i] = static_cast<int>(order[i]);
    return static_cast<int>(order.size());
}
void dk_set_entropy_numa_node(int node) { dk::set_preferred_numa(node); }
void dk_last_entropy_info(int *threads, int *l3_group) {
    if (threads) *threads = last_entropy_threads();
    if (l3_group) *l3_group = last_entropy_group();
}

// ---- debug ------------------------------------------------------------------------------------------------------------------
int dk_dbg_stream_encode_gated(int model_id, size_t n, const uint32_t init[256], const uint32_t *dist, const uint8_t *sym, size_t m, uint32_t origin,
                               uint8_t *out, size_t out_cap, size_t *out_len, const size_t *ready, unsigned stall_ms, int host_threads) {
    if (!ready) return DK_E_ARG;
    static_assert(sizeof(std::atomic<size_t>) == sizeof(size_t) && std::atomic<size_t>::is_always_lock_free, "the caller's word is read as an atomic");
    DcStream s;
    s.n = n; s.init = init; s.dist = dist; s.sym = sym; s.m = m; s.origin = origin;
    s.ready = reinterpret_cast<const std::atomic<size_t> *>(ready);
    s.stall_ms = stall_ms;
    return encode_block_stream(model_id, s, out, out_cap, out_len, host_threads);
}
int dk_dbg_dev_local_sort(dk_ctx *ctx, uint64_t *d_keys, uint32_t *d_vals, size_t count, int begin_bit, int end_bit) {
    DK_TRY(begin_call(ctx));
    ScopedCall sc(ctx);
    if (!d_keys || !d_vals) return ctx->fail(DK_E_ARG, "null pointer");
    return local_sort_tiles(ctx, d_keys, d_vals, count, begin_bit, end_bit);
}
int dk_dbg_dev_sort_pairs(dk_ctx *ctx, uint64_t *d_keys, uint32_t *d_vals, size_t count, int begin_bit, int end_bit) {
    DK_TRY(begin_call(ctx));
    ScopedCall sc(ctx);
    if (!d_keys || !d_vals || count == 0) return ctx->fail(DK_E_ARG, "null pointer or empty input");
    uint64_t *k0 = d_keys, *k1 = ctx->ws_alloc<uint64_t>(count);
    uint32_t *v0 = d_vals, *v1 = ctx->ws_alloc<uint32_t>(count);
    if (!k1 || !v1) return DK_E_NOMEM;
    DK_TRY(sort_pairs(ctx, k0, k1, v0, v1, count, begin_bit, end_bit));
    if (k0 != d_keys) {
        DK_HIP(ctx, hipMemcpyAsync(d_keys, k0, count * 8, hipMemcpyDeviceToDevice, ctx->stream));
        DK_HIP(ctx, hipMemcpyAsync(d_vals, v0, count * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    DK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DK_OK;
}
int dk_dbg_sort_pairs(dk_ctx *ctx, uint64_t *keys, uint32_t *vals, size_t count, int begin_bit, int end_bit) {
    DK_TRY(begin_call(ctx));
    ScopedCall sc(ctx);
    if (!keys || !vals || count == 0) return ctx->fail(DK_E_ARG, "null pointer or empty input");
    uint64_t *k0 = ctx->ws_alloc<uint64_t>(count), *k1 = ctx->ws_alloc<uint64_t>(count);
    uint32_t *v0 = ctx->ws_alloc<uint32_t>(count), *v1 = ctx->ws_alloc<uint32_t>(count);
    if (!k0 || !k1 || !v0 || !v1) return DK_E_NOMEM;
    DK_HIP(ctx, hipMemcpyAsync(k0, keys, count * 8, hipMemcpyHostToDevice, ctx->stream));
    DK_HIP(ctx, hipMemcpyAsync(v0, vals, count * 4, hipMemcpyHostToDevice, ctx->stream));
    DK_TRY(sort_pairs(ctx, k0, k1, v0, v1, count, begin_bit, end_bit));
    DK_HIP(ctx, hipMemcpyAsync(keys, k0, count * 8, hipMemcpyDeviceToHost, ctx->stream));
    DK_HIP(ctx, hipMemcpyAsync(vals, v0, count * 4, hipMemcpyDeviceToHost, ctx->stream));
    DK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DK_OK;
}

}  // extern "C"
