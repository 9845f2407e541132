// Forward BWT emit and inverse BWT.
//
// Forward: replaces compress::bwt::TransformIterator as driven by src/block/dc.rs:45-50 (known answers
// src/saca.rs:411-412): L[j] = T[SA[j]-1], L[j] = T[n-1] where SA[j] == 0, origin = that j.
//
// Inverse: replaces compress::bwt::decode as driven by src/block/dc.rs:154-156 (in-repo analogue
// etc/dark-c/src/archon3.cpp:70-86).  The serial "follow the jump table n times" becomes
//   k_ibwt_hist / scan   per-tile symbol histograms -> stable counting-sort offsets (one 8-bit radix pass over L)
//   k_ibwt_lf            psi[LF(i)] = i, with the origin element placed first in its symbol class (same rule as
//                        the reference's table build), psi[class start of L[origin]] = END
//   k_ibwt_walk          every position that is a multiple of S (plus origin) is a splitter; one lane per splitter
//                        walks psi to the next splitter and records (next splitter, steps)
//   k_ibwt_rank          pointer jumping over the reduced list -> text offset of every splitter
//   k_ibwt_emit          one lane per splitter re-walks its sub-list and writes the text bytes
#include "context.hpp"
#include "device_util.hpp"

namespace dk {
namespace {

// ---- forward ------------------------------------------------------------------------------------------------------
constexpr int BG_PER_THREAD = 16;  // gathers in flight per thread (all addresses first, then all symbols): the kernel is latency-bound
__global__ __launch_bounds__(256) void k_bwt_gather(const uint8_t *__restrict__ t, const uint32_t *__restrict__ sa, size_t n,
                                                     uint8_t *__restrict__ bwt, uint32_t *__restrict__ origin) {
    const size_t j0 = (static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x) * BG_PER_THREAD;
    if (j0 >= n) return;
    if (j0 + BG_PER_THREAD <= n && (reinterpret_cast<uintptr_t>(bwt) & 15) == 0 && (reinterpret_cast<uintptr_t>(sa) & 15) == 0) {
        uint32_t p[BG_PER_THREAD];
#pragma unroll
        for (int q = 0; q < BG_PER_THREAD / 4; ++q) {
            const uint4 v = reinterpret_cast<const uint4 *>(sa + j0)[q];
            p[4 * q] = v.x; p[4 * q + 1] = v.y; p[4 * q + 2] = v.z; p[4 * q + 3] = v.w;
        }
        uint8_t c[BG_PER_THREAD];
#pragma unroll
        for (int k = 0; k < BG_PER_THREAD; ++k) c[k] = t[p[k] ? p[k] - 1 : n - 1];
        uint32_t w[BG_PER_THREAD / 4];
#pragma unroll
        for (int q = 0; q < BG_PER_THREAD / 4; ++q)
            w[q] = c[4 * q] | (static_cast<uint32_t>(c[4 * q + 1]) << 8) | (static_cast<uint32_t>(c[4 * q + 2]) << 16) | (static_cast<uint32_t>(c[4 * q + 3]) << 24);
#pragma unroll
        for (int k = 0; k < BG_PER_THREAD; ++k)
            if (p[k] == 0) *origin = static_cast<uint32_t>(j0 + k);
        *reinterpret_cast<uint4 *>(bwt + j0) = make_uint4(w[0], w[1], w[2], w[3]);
        return;
    }
    for (size_t j = j0; j < n && j < j0 + BG_PER_THREAD; ++j) {
        const uint32_t p = sa[j];
        if (p == 0) { bwt[j] = t[n - 1]; *origin = static_cast<uint32_t>(j); } else { bwt[j] = t[p - 1]; }
    }
}

// ---- inverse ------------------------------------------------------------------------------------------------------
constexpr uint32_t IB_END = 0xFFFFFFFFu;
constexpr int IB_BLOCK = 256;
constexpr int IB_WAVES = IB_BLOCK / 64;
constexpr int IB_SPT = 16;                    // symbols per thread
constexpr int IB_TILE = IB_BLOCK * IB_SPT;    // 4096 symbols per workgroup
constexpr int IB_MAX_CHUNKS = 256;

// 16 private copies of the histogram (copy = lane mod 16): the BWT is made of runs, and LDS atomics of one wave instruction that hit the
// same counter are serialised (one copy: 0.25 ms per 1e8 bytes; the radix sort's digit-plane histogram has the same shape)
__global__ __launch_bounds__(IB_BLOCK) void k_ibwt_hist(const uint8_t *__restrict__ bwt, size_t n, uint32_t *__restrict__ tile_hist) {
    __shared__ uint32_t h[16][256];
    const int tid = threadIdx.x;
    for (int i = tid; i < 16 * 256; i += IB_BLOCK) (&h[0][0])[i] = 0;
    __syncthreads();
    uint32_t *mine = h[tid & 15];
    const size_t base = static_cast<size_t>(blockIdx.x) * IB_TILE;
    static_assert(IB_TILE == IB_BLOCK * 16, "one 16-byte load per thread");
    if (base + IB_TILE <= n && (reinterpret_cast<uintptr_t>(bwt) & 15) == 0) {
        const uint4 v = reinterpret_cast<const uint4 *>(bwt + base)[tid];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int b = 0; b < 4; ++b) atomicAdd(&mine[(w[k] >> (8 * b)) & 0xFFu], 1u);
        }
    } else {
#pragma unroll
        for (int k = 0; k < IB_SPT; ++k) {
            const size_t i = base + static_cast<size_t>(k) * IB_BLOCK + tid;
            if (i < n) atomicAdd(&mine[bwt[i]], 1u);
        }
    }
    __syncthreads();
    uint32_t sum = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) sum += h[c][tid];
    tile_hist[static_cast<size_t>(blockIdx.x) * 256 + tid] = sum;
}
// digit-major exclusive scan of tile_hist, same three phases as the radix sort; class_start[256] also exported
__global__ __launch_bounds__(256) void k_ibwt_scan_a(const uint32_t *__restrict__ tile_hist, size_t ntiles, size_t tpc,
                                                      uint32_t *__restrict__ chunk_sum) {
    const size_t g = blockIdx.x, t0 = g * tpc, t1 = t0 + tpc < ntiles ? t0 + tpc : ntiles;
    uint32_t s = 0;
#pragma unroll 8
    for (size_t t = t0; t < t1; ++t) s += tile_hist[t * 256 + threadIdx.x];
    chunk_sum[g * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_ibwt_scan_b(uint32_t *__restrict__ chunk_sum, size_t nchunks, uint32_t *__restrict__ class_start) {
    __shared__ uint32_t s_tmp[IB_WAVES + 1];
    const int d = threadIdx.x;
    uint32_t run = 0;
#pragma unroll 8
    for (size_t g = 0; g < nchunks; ++g) {
        const uint32_t v = chunk_sum[g * 256 + d];
        chunk_sum[g * 256 + d] = run;
        run += v;
    }
    const uint32_t base = block_excl_sum<IB_WAVES>(run, s_tmp, nullptr);
    class_start[d] = base;
#pragma unroll 8
    for (size_t g = 0; g < nchunks; ++g) chunk_sum[g * 256 + d] += base;
}
__global__ __launch_bounds__(256) void k_ibwt_scan_c(uint32_t *__restrict__ tile_hist, size_t ntiles, size_t tpc,
                                                      const uint32_t *__restrict__ chunk_sum) {
    const size_t g = blockIdx.x, t0 = g * tpc, t1 = t0 + tpc < ntiles ? t0 + tpc : ntiles;
    uint32_t run = chunk_sum[g * 256 + threadIdx.x];
    for (size_t t = t0; t < t1; ++t) {
        const uint32_t v = tile_hist[t * 256 + threadIdx.x];
        tile_hist[t * 256 + threadIdx.x] = run;
        run += v;
    }
}

// psi[LF(i)] = i.  LF is the stable counting-sort destination of L[i], except that the origin element goes first in
// its class (it is the suffix consisting of the last text symbol alone, the smallest of its class).
__global__ __launch_bounds__(IB_BLOCK) void k_ibwt_lf(const uint8_t *__restrict__ bwt, size_t n, uint32_t origin,
                                                       const uint32_t *__restrict__ tile_offs, const uint32_t *__restrict__ class_start,
                                                       uint64_t *__restrict__ psi) {
    __shared__ uint32_t s_cnt[IB_WAVES][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t tile_base = static_cast<size_t>(blockIdx.x) * IB_TILE;
    for (int i = tid; i < IB_WAVES * 256; i += IB_BLOCK) (&s_cnt[0][0])[i] = 0;
    const uint32_t wbase = static_cast<uint32_t>(wave) * (64 * IB_SPT);
    uint32_t sym[IB_SPT], rnk[IB_SPT];
#pragma unroll
    for (int k = 0; k < IB_SPT; ++k) {
        const size_t i = tile_base + wbase + k * 64 + lane;
        sym[k] = i < n ? bwt[i] : 0x100u;  // 0x100 = padding, ranked but never stored
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IB_SPT; ++k) {
        const uint32_t d = sym[k] & 0xFFu;
        const bool pad = sym[k] > 0xFFu;
        const LaneSet same = wave_match<8>(d, __ballot(!pad));
        const uint32_t before = same.before();
        const uint32_t old = pad ? 0u : s_cnt[wave][d];
        __builtin_amdgcn_wave_barrier();
        if (!pad && before == 0) s_cnt[wave][d] = old + same.count();
        __builtin_amdgcn_wave_barrier();
        rnk[k] = old + before;
    }
    __syncthreads();
    {
        const int d = tid;
        uint32_t run = tile_offs[static_cast<size_t>(blockIdx.x) * 256 + d];
#pragma unroll
        for (int w = 0; w < IB_WAVES; ++w) {
            const uint32_t c = s_cnt[w][d];
            s_cnt[w][d] = run;
            run += c;
        }
    }
    __syncthreads();
    const uint32_t c0 = bwt[origin];
    const uint32_t c0_start = class_start[c0];
#pragma unroll
    for (int k = 0; k < IB_SPT; ++k) {
        const size_t i = tile_base + wbase + k * 64 + lane;
        if (i >= n) continue;
        uint32_t dest = s_cnt[wave][sym[k]] + rnk[k];
        if (i == origin) dest = c0_start;
        else if (sym[k] == c0 && i < origin) dest += 1;
        // the origin element has no successor: following it ends the text (the reference stores table[..] = 0 there).
        // The entry also carries the symbol the step emits (L[i]), so that a walk makes ONE dependent random load per step.
        psi[dest] = (static_cast<uint64_t>(sym[k]) << 32) | (i == origin ? IB_END : static_cast<uint32_t>(i));
    }
}

// Splitters: positions that are multiples of S, and origin (the start of the text).  Splitter id of position p:
// p / S for p % S == 0; id nsplit-1 is reserved for origin when origin is not a multiple of S.
__device__ __forceinline__ bool is_splitter(uint32_t p, uint32_t S, uint32_t origin) { return (p % S) == 0 || p == origin; }
__device__ __forceinline__ uint32_t splitter_id(uint32_t p, uint32_t S, uint32_t origin, uint32_t nreg) {
    return (p == origin && (p % S) != 0) ? nreg : p / S;
}

// Text step k visits position cur_k: cur_0 = origin, cur_{k+1} = psi[cur_k]... with the reference's convention the
// text symbol k is L[psi[cur_k]], and the last symbol is L[origin] when psi hits END.
// rec (may be null): the walk also RECORDS the symbols it passes -- the first IB_REC of them, eight at a time, into the splitter's own
// IB_REC-byte stretch of `rec` -- and the position a longer walk stands at after IB_REC steps (resume); k_ibwt_copy then writes the text
// from these records instead of walking the successor table a second time (one random 64-byte line per text byte saved).
constexpr uint32_t IB_REC = 256;
__global__ __launch_bounds__(256) void k_ibwt_walk(const uint64_t *__restrict__ psi, uint32_t n, uint32_t origin, uint32_t S,
                                                    uint32_t nreg, uint32_t nsplit, uint32_t *__restrict__ nxt,
                                                    uint32_t *__restrict__ len, uint32_t *__restrict__ len_keep, uint8_t *__restrict__ rec,
                                                    uint32_t *__restrict__ resume) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsplit) return;
    uint32_t cur;
    if (s == nreg) {
        cur = origin;  // only exists when origin % S != 0
    } else {
        cur = s * S;
        if (cur >= n) { nxt[s] = IB_END; len[s] = 0; if (len_keep) len_keep[s] = 0; return; }
    }
    uint32_t steps = 0, to = IB_END;
    uint64_t acc = 0;
    uint64_t *out = rec ? reinterpret_cast<uint64_t *>(rec + static_cast<size_t>(s) * IB_REC) : nullptr;
    for (;;) {
        const uint64_t e = psi[cur];
        const uint32_t p = static_cast<uint32_t>(e);
        if (rec && steps < IB_REC) {
            acc |= (e >> 32 & 0xFFull) << (8 * (steps & 7u));  // the symbol this step emits
            if ((steps & 7u) == 7u) { out[steps >> 3] = acc; acc = 0; }
        }
        ++steps;  // this step emits one text symbol
        if (p == IB_END) break;
        if (is_splitter(p, S, origin)) { to = splitter_id(p, S, origin, nreg); break; }
        cur = p;
        if (rec && steps == IB_REC) resume[s] = cur;  // where a walk longer than the record goes on
        if (steps > n) break;  // corrupt input: never spin
    }
    if (rec && steps < IB_REC && (steps & 7u)) out[steps >> 3] = acc;  // the last, partial word (the stretch is the splitter's own)
    nxt[s] = to;
    len[s] = steps;
    if (len_keep) len_keep[s] = steps;
}

// pointer jumping: dist_to_end[s] = len[s] + dist_to_end[nxt[s]]
__global__ __launch_bounds__(256) void k_ibwt_jump(const uint32_t *__restrict__ nxt_in, const uint32_t *__restrict__ acc_in,
                                                    uint32_t *__restrict__ nxt_out, uint32_t *__restrict__ acc_out, uint32_t nsplit,
                                                    uint32_t *__restrict__ pending) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsplit) return;
    const uint32_t to = nxt_in[s];
    uint32_t a = acc_in[s];
    if (to == IB_END) {
        nxt_out[s] = IB_END;
        acc_out[s] = a;
        return;
    }
    a += acc_in[to];
    const uint32_t to2 = nxt_in[to];
    nxt_out[s] = to2;
    acc_out[s] = a;
    if (to2 != IB_END) *pending = 1;
}

__global__ __launch_bounds__(256) void k_ibwt_emit(const uint8_t *__restrict__ bwt, const uint64_t *__restrict__ psi, uint32_t n,
                                                    uint32_t origin, uint32_t S, uint32_t nreg, uint32_t nsplit,
                                                    const uint32_t *__restrict__ dist_to_end, uint8_t *__restrict__ out,
                                                    uint32_t *__restrict__ bad) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsplit) return;
    uint32_t cur;
    if (s == nreg) cur = origin;
    else { cur = s * S; if (cur >= n) return; }
    const uint32_t d = dist_to_end[s];
    if (d > n) { *bad = 1; return; }  // not on the text cycle (corrupt input)
    uint32_t k = n - d;  // text offset of the first symbol this splitter emits
    // bytes are gathered into a 64-bit word and leave as one aligned 8-byte store (single bytes only at the ragged ends)
    const bool wide = (reinterpret_cast<uintptr_t>(out) & 7) == 0;
    uint64_t acc = 0;
    uint32_t have = 0;
    auto put = [&](uint8_t b) {
        if (wide && (have > 0 || (k & 7u) == 0)) {
            acc |= static_cast<uint64_t>(b) << (8 * have);
            ++have;
            ++k;
            if (have == 8) {
                *reinterpret_cast<uint64_t *>(out + k - 8) = acc;
                acc = 0;
                have = 0;
            }
        } else {
            out[k++] = b;
        }
    };
    for (;;) {
        const uint64_t e = psi[cur];
        const uint32_t p = static_cast<uint32_t>(e);
        const uint8_t symbol = static_cast<uint8_t>(e >> 32);  // = L[p], or L[origin] on the terminal entry
        if (p == IB_END) { if (k < n) put(symbol); else *bad = 1; break; }
        if (k >= n) { *bad = 1; break; }
        put(symbol);
        if (is_splitter(p, S, origin)) break;
        cur = p;
    }
    for (uint32_t jj = 0; jj < have; ++jj) out[k - have + jj] = static_cast<uint8_t>(acc >> (8 * jj));
}

// the text from the walk's records: splitter s owns text[n - dist_to_end[s] .. + len[s]); its first IB_REC bytes are in its record, the
// rest (walks longer than the record: 2 % of them at S = 64) is walked from resume[s] as k_ibwt_emit does
__global__ __launch_bounds__(256) void k_ibwt_copy(const uint64_t *__restrict__ psi, uint32_t n, uint32_t origin, uint32_t S, uint32_t nreg,
                                                    uint32_t nsplit, const uint32_t *__restrict__ dist_to_end, const uint32_t *__restrict__ len,
                                                    const uint8_t *__restrict__ rec, const uint32_t *__restrict__ resume,
                                                    uint8_t *__restrict__ out, uint32_t *__restrict__ bad) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsplit) return;
    if (s != nreg && static_cast<uint64_t>(s) * S >= n) return;
    const uint32_t d = dist_to_end[s], total = len[s];
    if (d > n || total > d) { *bad = 1; return; }  // not on the text cycle (corrupt input)
    uint32_t k = n - d;
    const bool wide = (reinterpret_cast<uintptr_t>(out) & 7) == 0;
    uint64_t acc = 0;
    uint32_t have = 0;
    auto put = [&](uint8_t b) {
        if (wide && (have > 0 || (k & 7u) == 0)) {
            acc |= static_cast<uint64_t>(b) << (8 * have);
            ++have;
            ++k;
            if (have == 8) {
                *reinterpret_cast<uint64_t *>(out + k - 8) = acc;
                acc = 0;
                have = 0;
            }
        } else {
            out[k++] = b;
        }
    };
    const uint64_t *src = reinterpret_cast<const uint64_t *>(rec + static_cast<size_t>(s) * IB_REC);
    const uint32_t recorded = total < IB_REC ? total : IB_REC;
    for (uint32_t j = 0; j < recorded; j += 8) {
        const uint64_t w = src[j >> 3];
        const uint32_t cnt = recorded - j < 8 ? recorded - j : 8;
        for (uint32_t b = 0; b < cnt; ++b) put(static_cast<uint8_t>(w >> (8 * b)));
    }
    if (total > IB_REC) {
        uint32_t cur = resume[s], left = total - IB_REC;
        while (left--) {
            const uint64_t e = psi[cur];
            put(static_cast<uint8_t>(e >> 32));
            cur = static_cast<uint32_t>(e);
            if (cur == IB_END && left) { *bad = 1; break; }
        }
    }
    for (uint32_t jj = 0; jj < have; ++jj) out[k - have + jj] = static_cast<uint8_t>(acc >> (8 * jj));
}

}  // namespace

int bwt_gather_device(dk_ctx *ctx, const uint8_t *d_text, const uint32_t *d_sa, size_t n, uint8_t *d_bwt, uint32_t *origin) {
    hipStream_t st = ctx->stream;
    uint32_t *d_origin = ctx->d_mail + 8;
    DK_HIP(ctx, hipMemsetAsync(d_origin, 0xFF, sizeof(uint32_t), st));
    {
        LaunchScope ls(ctx, K_BWT_GATHER, 6.0 * n);
        k_bwt_gather<<<dim3(div_up(div_up(n, BG_PER_THREAD), 256)), dim3(256), 0, st>>>(d_text, d_sa, n, d_bwt, d_origin);
    }
    DK_HIP(ctx, hipGetLastError());
    DK_HIP(ctx, hipMemcpyAsync(ctx->h_mail + 8, d_origin, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    DK_HIP(ctx, hipStreamSynchronize(st));
    *origin = ctx->h_mail[8];
    if (*origin >= n) return ctx->fail(DK_E_INTERNAL, "bwt_forward: no suffix 0 in the suffix array");
    return DK_OK;
}

// suffix sort + BWT: the gather is skipped when the sort already wrote L (suffix_array_device, short-prefix path)
int bwt_forward_device(dk_ctx *ctx, const uint8_t *d_text, size_t n, uint32_t *d_sa, uint8_t *d_bwt, uint32_t *origin) {
    hipStream_t st = ctx->stream;
    uint32_t *d_origin = ctx->d_mail + 8;
    DK_HIP(ctx, hipMemsetAsync(d_origin, 0xFF, sizeof(uint32_t), st));
    bool written = false;
    Timer t;
    DK_TRY(suffix_array_device(ctx, d_text, n, d_sa, d_bwt, d_origin, &written));
    DK_HIP(ctx, hipStreamSynchronize(st));
    ctx->stats.ms_sa = t.ms();
    Timer t2;
    if (!written) {
        DK_TRY(bwt_gather_device(ctx, d_text, d_sa, n, d_bwt, origin));
    } else {
        DK_HIP(ctx, hipMemcpyAsync(ctx->h_mail + 8, d_origin, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        DK_HIP(ctx, hipStreamSynchronize(st));
        *origin = ctx->h_mail[8];
        if (*origin >= n) return ctx->fail(DK_E_INTERNAL, "bwt_forward: no suffix 0 in the suffix array");
    }
    ctx->stats.ms_bwt = t2.ms();
    return DK_OK;
}

int bwt_inverse_device(dk_ctx *ctx, const uint8_t *d_bwt, size_t n, uint32_t origin, uint8_t *d_out) {
    if (n == 0 || n > 0xFFFFFFF0ull || origin >= n) return ctx->fail(DK_E_ARG, "bwt_inverse: bad n / origin");
    hipStream_t st = ctx->stream;
    const size_t mark = ctx->ws_mark();
    const size_t ntiles = div_up(n, IB_TILE);
    const size_t tpc = div_up(ntiles, IB_MAX_CHUNKS);
    const size_t nchunks = div_up(ntiles, tpc);
    uint32_t *tile_hist = ctx->ws_alloc<uint32_t>(ntiles * 256);
    uint32_t *chunk_sum = ctx->ws_alloc<uint32_t>(nchunks * 256);
    uint32_t *class_start = ctx->ws_alloc<uint32_t>(256);
    uint64_t *psi = ctx->ws_alloc<uint64_t>(n);
    if (!tile_hist || !chunk_sum || !class_start || !psi) return DK_E_NOMEM;
    {
        LaunchScope ls(ctx, K_IBWT_HIST, 1.0 * n);
        k_ibwt_hist<<<dim3(ntiles), dim3(IB_BLOCK), 0, st>>>(d_bwt, n, tile_hist);
        k_ibwt_scan_a<<<dim3(nchunks), dim3(256), 0, st>>>(tile_hist, ntiles, tpc, chunk_sum);
        k_ibwt_scan_b<<<dim3(1), dim3(256), 0, st>>>(chunk_sum, nchunks, class_start);
        k_ibwt_scan_c<<<dim3(nchunks), dim3(256), 0, st>>>(tile_hist, ntiles, tpc, chunk_sum);
    }
    {
        LaunchScope ls(ctx, K_IBWT_LF, 5.0 * n);
        k_ibwt_lf<<<dim3(ntiles), dim3(IB_BLOCK), 0, st>>>(d_bwt, n, origin, tile_hist, class_start, psi);
    }
    DK_HIP(ctx, hipGetLastError());
    // splitters
    const uint32_t S = n < (1u << 16) ? 8u : 64u;  // measured at 1e8: 16 / 32 / 64 / 128 / 256 -> 11.6 / 8.9 / 7.9 / 8.1 / 8.6 ms
    const uint32_t nreg = static_cast<uint32_t>(div_up(n, S));
    const uint32_t nsplit = nreg + ((origin % S) != 0 ? 1u : 0u);
    uint32_t *nxt = ctx->ws_alloc<uint32_t>(nsplit), *nxt_alt = ctx->ws_alloc<uint32_t>(nsplit);
    uint32_t *acc = ctx->ws_alloc<uint32_t>(nsplit), *acc_alt = ctx->ws_alloc<uint32_t>(nsplit);
    if (!nxt || !nxt_alt || !acc || !acc_alt) return DK_E_NOMEM;
    // records of the walk (IB_REC bytes per splitter = 4 n bytes at S = 64): the text is then copied from them instead of walked again
    const bool record = DK_KNOB("DK_IBWT_RECORD", 1) != 0 && S == 64;
    uint32_t *len_keep = record ? ctx->ws_alloc<uint32_t>(nsplit) : nullptr;
    uint32_t *resume = record ? ctx->ws_alloc<uint32_t>(nsplit) : nullptr;
    uint8_t *rec = record ? ctx->ws_alloc<uint8_t>(static_cast<size_t>(nsplit) * IB_REC) : nullptr;
    if (record && (!len_keep || !resume || !rec)) return DK_E_NOMEM;
    {
        LaunchScope ls(ctx, K_IBWT_WALK, (record ? 5.0 : 4.0) * n);
        k_ibwt_walk<<<dim3(div_up(nsplit, 256)), dim3(256), 0, st>>>(psi, static_cast<uint32_t>(n), origin, S, nreg, nsplit, nxt, acc, len_keep, rec, resume);
    }
    DK_HIP(ctx, hipGetLastError());
    // Pointer jumping halves every chain per step: ceil(log2(nsplit)) + 1 steps finish any single path through the splitters.
    // They are enqueued back to back (no host round trip per step); only the last one reports whether something is still
    // unresolved, which can only mean a cycle (corrupt input).
    uint32_t *d_pending = ctx->d_mail + 10;
    const int steps = static_cast<int>(ceil_log2_u64(nsplit)) + 1;
    for (int it = 0; it < steps; ++it) {
        if (it == steps - 1) DK_HIP(ctx, hipMemsetAsync(d_pending, 0, sizeof(uint32_t), st));
        {
            LaunchScope ls(ctx, K_IBWT_JUMP, 24.0 * nsplit);
            k_ibwt_jump<<<dim3(div_up(nsplit, 256)), dim3(256), 0, st>>>(nxt, acc, nxt_alt, acc_alt, nsplit, d_pending);
        }
        std::swap(nxt, nxt_alt);
        std::swap(acc, acc_alt);
    }
    DK_HIP(ctx, hipGetLastError());
    DK_HIP(ctx, hipMemcpyAsync(ctx->h_mail + 10, d_pending, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    DK_HIP(ctx, hipStreamSynchronize(st));
    if (ctx->h_mail[10] != 0) return ctx->fail(DK_E_STREAM, "bwt_inverse: successor table has a cycle (corrupt input)");
    uint32_t *d_bad = ctx->d_mail + 11;
    DK_HIP(ctx, hipMemsetAsync(d_bad, 0, sizeof(uint32_t), st));
    {
        LaunchScope ls(ctx, K_IBWT_EMIT, (record ? 2.0 : 6.0) * n);
        if (record)
            k_ibwt_copy<<<dim3(div_up(nsplit, 256)), dim3(256), 0, st>>>(psi, static_cast<uint32_t>(n), origin, S, nreg, nsplit, acc, len_keep, rec, resume,
                                                                          d_out, d_bad);
        else
            k_ibwt_emit<<<dim3(div_up(nsplit, 256)), dim3(256), 0, st>>>(d_bwt, psi, static_cast<uint32_t>(n), origin, S, nreg, nsplit, acc,
                                                                          d_out, d_bad);
    }
    DK_HIP(ctx, hipGetLastError());
    DK_HIP(ctx, hipMemcpyAsync(ctx->h_mail + 11, d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    DK_HIP(ctx, hipStreamSynchronize(st));
    ctx->ws_release(mark);
    if (ctx->h_mail[11]) return ctx->fail(DK_E_STREAM, "bwt_inverse: BWT/origin do not describe a single text cycle");
    return DK_OK;
}

}  // namespace dk
