// Stable LSD radix sort of (u64 key, u32 value) pairs, 8 bits per pass -- the workhorse of the suffix sort.
// It takes the place of all the sorting work inside saca() (src/saca.rs:270-340): bucket placement, the induced
// sorts and the naming pass become "sort by packed prefix, then by (group, rank of the suffix h further on)".
//
// Single-pass-per-digit path (default, count < 2^30):
//   k_radix_hist_all one read of the keys -> the 256-bin histograms of ALL passes (LDS atomics) -> digit_base[pass][256]
//   k_radix_scatter<true>  per pass: tiles are handed out by an atomic ticket, each tile ranks its keys stably with wave64
//                    ballots (match-any over the 8 digit bits), publishes its per-digit counts as {flag,value} words and
//                    finds its global offsets by decoupled look-back over the preceding tiles' words (relaxed agent-scope
//                    atomics; the word IS the flag, so no fence is needed), reorders the tile in LDS so that equal digits
//                    are contiguous, then writes runs to HBM.  Every spin is bounded and sets an error word.
// Three-phase path (count >= 2^30, or DK_SORT=classic): k_radix_hist per pass -> k_radix_scan_{a,b,c} -> k_radix_scatter<false>.
// Algorithmic bytes per pass: scatter 12 B/pair read + 12 B/pair written; hist_all 8 B/key once per sort.
#include "context.hpp"
#include <cstdlib>
#include <string>

#include "device_util.hpp"

namespace dk {
namespace {

constexpr int RS_BLOCK = 256;
constexpr int RS_WAVES = RS_BLOCK / 64;
constexpr int RS_KPT = 16;                   // pairs per thread
constexpr int RS_TILE = RS_BLOCK * RS_KPT;   // 4096 pairs per workgroup
constexpr int RS_MAX_CHUNKS = 256;

__device__ __forceinline__ uint32_t digit_of(uint64_t k, int shift) { return static_cast<uint32_t>(k >> shift) & 0xFFu; }

// PAIRS = true: the key of element i is (hi[i] << 32) | lo[i], read from two u32 arrays, and there is no value array
template <bool PAIRS>
__device__ __forceinline__ uint64_t load_key(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ hi,
                                             const uint32_t *__restrict__ lo, size_t i) {
    if (PAIRS) return (static_cast<uint64_t>(hi[i]) << 32) | (lo ? lo[i] : static_cast<uint32_t>(i));  // lo == nullptr: lo[i] = i
    return keys[i];
}

// 16 private copies of the histogram (copy = lane mod 16): text digits are skewed, and LDS atomics of one wave instruction that
// hit the same address are serialised; spreading them over copies cuts that contention up to 16 x.
constexpr int RS_HCOPIES = 16;

template <bool PAIRS>
__global__ __launch_bounds__(RS_BLOCK) void k_radix_hist(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ hi,
                                                          const uint32_t *__restrict__ lo, size_t n, int shift,
                                                          uint32_t *__restrict__ tile_hist) {
    __shared__ uint32_t h[RS_HCOPIES][256];
    const int tid = threadIdx.x;
#pragma unroll
    for (int c = 0; c < RS_HCOPIES; ++c) h[c][tid] = 0;
    __syncthreads();
    uint32_t *mine = h[tid & (RS_HCOPIES - 1)];
    const size_t base = static_cast<size_t>(blockIdx.x) * RS_TILE;
    if (!PAIRS && base + RS_TILE <= n) {  // full tile: two keys per 16-byte load (order inside the tile is irrelevant here)
        const uint4 *p = reinterpret_cast<const uint4 *>(keys + base);
#pragma unroll
        for (int k = 0; k < RS_KPT / 2; ++k) {
            const uint4 v = p[k * RS_BLOCK + tid];
            const uint64_t k0 = (static_cast<uint64_t>(v.y) << 32) | v.x, k1 = (static_cast<uint64_t>(v.w) << 32) | v.z;
            atomicAdd(&mine[digit_of(k0, shift)], 1u);
            atomicAdd(&mine[digit_of(k1, shift)], 1u);
        }
    } else {
#pragma unroll
        for (int k = 0; k < RS_KPT; ++k) {
            const size_t i = base + static_cast<size_t>(k) * RS_BLOCK + tid;
            if (i < n) atomicAdd(&mine[digit_of(load_key<PAIRS>(keys, hi, lo, i), shift)], 1u);
        }
    }
    __syncthreads();
    uint32_t sum = 0;
#pragma unroll
    for (int c = 0; c < RS_HCOPIES; ++c) sum += h[c][tid];
    tile_hist[static_cast<size_t>(blockIdx.x) * 256 + tid] = sum;
}

// phase A: per chunk of tiles, per digit: sum of the tile counts
__global__ __launch_bounds__(256) void k_radix_scan_a(const uint32_t *__restrict__ tile_hist, size_t ntiles,
                                                       size_t tiles_per_chunk, uint32_t *__restrict__ chunk_sum) {
    const size_t g = blockIdx.x;
    const int d = threadIdx.x;
    const size_t t0 = g * tiles_per_chunk;
    const size_t t1 = t0 + tiles_per_chunk < ntiles ? t0 + tiles_per_chunk : ntiles;
    uint32_t s = 0;
#pragma unroll 8
    for (size_t t = t0; t < t1; ++t) s += tile_hist[t * 256 + d];
    chunk_sum[g * 256 + d] = s;
}
// phase B (one workgroup of 1024): digit-major exclusive scan of the chunk sums; 4 threads share a digit's column
__global__ __launch_bounds__(1024) void k_radix_scan_b(uint32_t *__restrict__ chunk_sum, size_t nchunks) {
    __shared__ uint32_t s_tmp[16 + 1];
    __shared__ uint32_t s_part[4][256];
    const int d = threadIdx.x & 255, part = threadIdx.x >> 8;
    const size_t q = (nchunks + 3) / 4;
    const size_t g0 = part * q, g1 = g0 + q < nchunks ? g0 + q : nchunks;
    uint32_t run = 0;
#pragma unroll 8
    for (size_t g = g0; g < g1; ++g) {
        const uint32_t v = chunk_sum[g * 256 + d];
        chunk_sum[g * 256 + d] = run;
        run += v;
    }
    s_part[part][d] = run;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t t = s_part[p][d];
        if (p < part) before += t;
        total += t;
    }
    // exclusive scan of the digit totals over d: only the part-0 threads carry a value
    const uint32_t base = block_excl_sum<16>(part == 0 ? total : 0u, s_tmp, nullptr);
    if (part == 0) s_part[0][d] = base;
    __syncthreads();
    const uint32_t off = s_part[0][d] + before;
#pragma unroll 8
    for (size_t g = g0; g < g1; ++g) chunk_sum[g * 256 + d] += off;
}
// phase C: tile counts -> exclusive global offsets
__global__ __launch_bounds__(256) void k_radix_scan_c(uint32_t *__restrict__ tile_hist, size_t ntiles, size_t tiles_per_chunk,
                                                       const uint32_t *__restrict__ chunk_sum) {
    const size_t g = blockIdx.x;
    const int d = threadIdx.x;
    const size_t t0 = g * tiles_per_chunk;
    const size_t t1 = t0 + tiles_per_chunk < ntiles ? t0 + tiles_per_chunk : ntiles;
    uint32_t run = chunk_sum[g * 256 + d];
    for (size_t t = t0; t < t1; ++t) {
        const uint32_t v = tile_hist[t * 256 + d];
        tile_hist[t * 256 + d] = run;
        run += v;
    }
}

// histograms of every pass in one read of the keys
__global__ __launch_bounds__(RS_BLOCK) void k_radix_hist_all(const uint64_t *__restrict__ keys, size_t n, int begin_bit, int npasses,
                                                              uint32_t *__restrict__ hist) {
    __shared__ uint32_t h[8 * 256];
    const int tid = threadIdx.x;
    for (int i = tid; i < npasses * 256; i += RS_BLOCK) h[i] = 0;
    __syncthreads();
    const size_t stride = static_cast<size_t>(gridDim.x) * RS_BLOCK;
    for (size_t i = static_cast<size_t>(blockIdx.x) * RS_BLOCK + tid; i < n; i += stride) {
        const uint64_t k = keys[i] >> begin_bit;
        for (int p = 0; p < npasses; ++p) atomicAdd(&h[p * 256 + (static_cast<uint32_t>(k >> (8 * p)) & 0xFFu)], 1u);
    }
    __syncthreads();
    for (int i = tid; i < npasses * 256; i += RS_BLOCK)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}
// one workgroup per pass: counts -> exclusive digit starts
__global__ __launch_bounds__(256) void k_radix_base_scan(uint32_t *__restrict__ hist) {
    __shared__ uint32_t s_tmp[RS_WAVES + 1];
    uint32_t *row = hist + static_cast<size_t>(blockIdx.x) * 256;
    const uint32_t v = row[threadIdx.x];
    row[threadIdx.x] = block_excl_sum<RS_WAVES>(v, s_tmp, nullptr);
}

__global__ void k_radix_fold_err(const uint32_t *__restrict__ err, uint32_t *__restrict__ acc) {
    if (*err) *acc = 1;
}

constexpr uint32_t ST_LOCAL = 1u << 30, ST_INCL = 2u << 30, ST_MASK = (1u << 30) - 1;
constexpr uint32_t RS_SPIN_LIMIT = 1u << 22;

// LOOKBACK = false: tile = blockIdx.x, offsets come from the scanned tile_offs table.
// LOOKBACK = true : tile = atomic ticket, `tile_offs` is the status array [ntiles][256] (zeroed), `digit_base` the
//                   exclusive digit starts of this pass, ctrl[0] the ticket counter, ctrl[1] the error word.
template <bool LOOKBACK, bool PAIRS = false>
__global__ __launch_bounds__(RS_BLOCK) void k_radix_scatter(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                             const uint32_t *__restrict__ pair_lo,
                                                             uint64_t *__restrict__ kout, uint32_t *__restrict__ vout, size_t n,
                                                             int shift, uint32_t *__restrict__ tile_offs,
                                                             const uint32_t *__restrict__ digit_base, uint32_t *__restrict__ ctrl,
                                                             uint32_t xcd_tiles) {
    __shared__ uint64_t s_keys[RS_TILE];          // 32 KiB: tile of keys in digit order; reused for the values
    __shared__ uint32_t s_cnt[RS_WAVES][256];     // per-wave digit counters, then exclusive over waves
    __shared__ uint32_t s_start[256];             // tile-local start of each digit
    __shared__ uint32_t s_gbase[256];             // global offset of the digit minus its tile-local start
    __shared__ uint32_t s_tmp[RS_WAVES + 1];
    __shared__ uint32_t s_ticket;
    __shared__ uint32_t s_hist[256];              // LOOKBACK: tile histogram made before the ranking, published early
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t tile = blockIdx.x;
    if (!LOOKBACK && xcd_tiles) {
        // XCD-aware order: blocks b and b+8 share an XCD (round-robin dispatch, speed only); give every XCD one contiguous
        // range of tiles so that neighbouring output runs meet in the same L2.  Grid = 8 * ceil(ntiles / 8).
        const uint32_t per = gridDim.x / 8;
        tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
        if (tile >= xcd_tiles) return;
    }
    if (LOOKBACK) {  // tickets: a tile only ever waits for tiles that already started
        if (tid == 0) s_ticket = atomicAdd(&ctrl[0], 1u);
        __syncthreads();
        tile = s_ticket;
    }
    const size_t tile_base = static_cast<size_t>(tile) * RS_TILE;
    const size_t left = n - tile_base;
    const uint32_t valid = left < static_cast<size_t>(RS_TILE) ? static_cast<uint32_t>(left) : RS_TILE;

    for (int i = tid; i < RS_WAVES * 256; i += RS_BLOCK) (&s_cnt[0][0])[i] = 0;
    if (LOOKBACK) s_hist[tid] = 0;
    __syncthreads();

    // wave w owns pairs [w*1024, (w+1)*1024) of the tile, lane-striped so that loads coalesce
    uint64_t key[RS_KPT];
    uint32_t val[RS_KPT];
    const uint32_t wbase = static_cast<uint32_t>(wave) * (64 * RS_KPT);
#pragma unroll
    for (int k = 0; k < RS_KPT; ++k) {
        const uint32_t li = wbase + k * 64 + lane;
        if (li < valid) {
            key[k] = load_key<PAIRS>(kin, vin, pair_lo, tile_base + li);  // PAIRS: vin holds the high words
            val[k] = PAIRS ? 0u : vin[tile_base + li];
            if (LOOKBACK) atomicAdd(&s_hist[digit_of(key[k], shift)], 1u);
        } else {
            key[k] = ~0ull;  // padding sorts behind every real pair of the tile and is never written
            val[k] = 0;
        }
    }
    __syncthreads();
    uint32_t *mine = nullptr;
    uint32_t my_cnt = 0;
    if (LOOKBACK) {  // publish this tile's counts before the (long) ranking so that successors rarely wait
        my_cnt = s_hist[tid];
        mine = tile_offs + static_cast<size_t>(tile) * 256 + tid;
        __hip_atomic_store(mine, (tile == 0 ? ST_INCL : ST_LOCAL) | my_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // stable rank inside the wave: lanes holding the same digit find each other with 8 ballots
    uint32_t rnk[RS_KPT];
    const uint64_t lt = lanemask_lt(lane);
#pragma unroll
    for (int k = 0; k < RS_KPT; ++k) {
        const uint32_t d = digit_of(key[k], shift);
        uint64_t same = ~0ull;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            same &= bit ? bal : ~bal;
        }
        const uint32_t before = static_cast<uint32_t>(__popcll(same & lt));
        const uint32_t old = s_cnt[wave][d];
        __builtin_amdgcn_wave_barrier();
        if (before == 0) s_cnt[wave][d] = old + static_cast<uint32_t>(__popcll(same));
        __builtin_amdgcn_wave_barrier();
        rnk[k] = old + before;
    }
    __syncthreads();

    {   // one thread per digit: exclusive over waves, then exclusive over digits
        const int d = tid;
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) {
            const uint32_t c = s_cnt[w][d];
            s_cnt[w][d] = run;
            run += c;
        }
        uint32_t goff;
        if (LOOKBACK) {
            // (padding keys of the last tile were never counted in s_hist: they are not real pairs)
            const uint32_t cnt = my_cnt;
            uint32_t excl = 0;
            if (tile != 0) {
                const uint32_t *look = mine - 256;
                uint32_t spins = 0;
                for (;;) {
                    const uint32_t v = __hip_atomic_load(look, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((v >> 30) == 0) {  // predecessor has not published yet
                        if (++spins > RS_SPIN_LIMIT) { ctrl[1] = 1; break; }
                        __builtin_amdgcn_s_sleep(1);
                        continue;
                    }
                    excl += v & ST_MASK;
                    if (v & ST_INCL) break;
                    look -= 256;
                }
                __hip_atomic_store(mine, ST_INCL | (excl + cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            goff = digit_base[d] + excl;
        } else {
            goff = tile_offs[static_cast<size_t>(tile) * 256 + d];
        }
        const uint32_t start = block_excl_sum<RS_WAVES>(run, s_tmp, nullptr);
        s_start[d] = start;
        s_gbase[d] = goff - start;
    }
    __syncthreads();

    uint32_t pos[RS_KPT];
#pragma unroll
    for (int k = 0; k < RS_KPT; ++k) {
        const uint32_t d = digit_of(key[k], shift);
        pos[k] = s_start[d] + s_cnt[wave][d] + rnk[k];
        s_keys[pos[k]] = key[k];
    }
    __syncthreads();

    uint32_t gi[RS_KPT];
#pragma unroll
    for (int k = 0; k < RS_KPT; ++k) {
        const uint32_t p = k * RS_BLOCK + tid;
        const uint64_t kk = s_keys[p];
        gi[k] = s_gbase[digit_of(kk, shift)] + p;
        if (p < valid) kout[gi[k]] = kk;
    }
    if (PAIRS) return;  // keys only
    __syncthreads();
    uint32_t *s_vals = reinterpret_cast<uint32_t *>(s_keys);
#pragma unroll
    for (int k = 0; k < RS_KPT; ++k) s_vals[pos[k]] = val[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RS_KPT; ++k) {
        const uint32_t p = k * RS_BLOCK + tid;
        if (p < valid) vout[gi[k]] = s_vals[p];
    }
}


// ---- chunked single-kernel passes (DK_SORT=chunked) -------------------------------------------------------------------
// The tiles are cut into RS_NCH = 8 contiguous chunks, one per XCD group (blocks b with equal b mod 8; observed placement, speed
// only).  Inside a chunk tiles are handed out by a ticket and find their offsets by decoupled look-back over the chunk's earlier
// tiles only; what precedes the chunk comes from per-chunk digit histograms H[chunk][digit], which the previous pass accumulates
// while it scatters (every element knows its destination, hence its chunk in the next pass) -- so a pass is ONE kernel: no
// per-pass histogram read, no scan kernels, and neighbouring output runs still meet in one XCD's L2.
constexpr int RS_NCH = 8;

struct ChunkGeom {
    uint32_t ntiles, tiles_per_chunk;
    uint64_t div_magic;  // ceil(2^40 / tiles_per_chunk): (x * magic) >> 40 == x / tiles_per_chunk for x < 2^20
};
__device__ __forceinline__ uint32_t chunk_of_tile(uint32_t tile, const ChunkGeom &g) {
    return static_cast<uint32_t>((static_cast<uint64_t>(tile) * g.div_magic) >> 40);
}

// H[chunk][digit] of the input arrangement for the first pass
__global__ __launch_bounds__(RS_BLOCK) void k_chunk_hist(const uint64_t *__restrict__ keys, size_t n, int shift, ChunkGeom g,
                                                          uint32_t *__restrict__ H) {
    __shared__ uint32_t h[RS_HCOPIES][256];
    const int tid = threadIdx.x;
    const uint32_t c = blockIdx.x % RS_NCH, lb = blockIdx.x / RS_NCH, nb = gridDim.x / RS_NCH;
#pragma unroll
    for (int k = 0; k < RS_HCOPIES; ++k) h[k][tid] = 0;
    __syncthreads();
    uint32_t *mine = h[tid & (RS_HCOPIES - 1)];
    const uint32_t t0 = c * g.tiles_per_chunk;
    const uint32_t t1 = t0 + g.tiles_per_chunk < g.ntiles ? t0 + g.tiles_per_chunk : g.ntiles;
    for (uint32_t t = t0 + lb; t < t1; t += nb) {
        const size_t base = static_cast<size_t>(t) * RS_TILE;
#pragma unroll
        for (int k = 0; k < RS_KPT; ++k) {
            const size_t i = base + static_cast<size_t>(k) * RS_BLOCK + tid;
            if (i < n) atomicAdd(&mine[digit_of(keys[i], shift)], 1u);
        }
    }
    __syncthreads();
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < RS_HCOPIES; ++k) sum += h[k][tid];
    if (sum) atomicAdd(&H[c * 256 + tid], sum);
}

// ctrl[0..7] = tickets per chunk, ctrl[8] = error word; status rows follow at ctrl + 64
__global__ __launch_bounds__(RS_BLOCK) void k_radix_chunked(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                             uint64_t *__restrict__ kout, uint32_t *__restrict__ vout, size_t n, int shift,
                                                             int next_shift, ChunkGeom g, const uint32_t *__restrict__ H_cur,
                                                             uint32_t *__restrict__ H_next, uint32_t *__restrict__ ctrl) {
    __shared__ uint64_t s_keys[RS_TILE];
    __shared__ uint32_t s_cnt[RS_WAVES][256];
    __shared__ uint32_t s_start[256];
    __shared__ uint32_t s_gbase[256];
    __shared__ uint32_t s_cbase[256];            // global start of (this chunk, digit)
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_hnext[RS_NCH][256];    // next pass: counts per (destination chunk, next digit), flushed once
    __shared__ uint32_t s_tmp[RS_WAVES + 1];
    __shared__ uint32_t s_ticket;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t c = blockIdx.x % RS_NCH;
    uint32_t *status = ctrl + 64;
    const uint32_t t0 = c * g.tiles_per_chunk;
    if (t0 >= g.ntiles) return;
    const uint32_t tiles_here = (t0 + g.tiles_per_chunk < g.ntiles ? t0 + g.tiles_per_chunk : g.ntiles) - t0;
    {   // where this chunk's share of every digit starts
        uint32_t tot = 0, before = 0;
#pragma unroll
        for (int cc = 0; cc < RS_NCH; ++cc) {
            const uint32_t v = H_cur[cc * 256 + tid];
            if (static_cast<uint32_t>(cc) < c) before += v;
            tot += v;
        }
        s_cbase[tid] = block_excl_sum<RS_WAVES>(tot, s_tmp, nullptr) + before;
#pragma unroll
        for (int cc = 0; cc < RS_NCH; ++cc) s_hnext[cc][tid] = 0;
    }
    const uint64_t lt = lanemask_lt(lane);
    const uint32_t wbase = static_cast<uint32_t>(wave) * (64 * RS_KPT);
    for (;;) {
        __syncthreads();  // previous tile fully written out; LDS reusable
        if (tid == 0) s_ticket = atomicAdd(&ctrl[c], 1u);
        for (int i = tid; i < RS_WAVES * 256; i += RS_BLOCK) (&s_cnt[0][0])[i] = 0;
        s_hist[tid] = 0;
        __syncthreads();
        const uint32_t ticket = s_ticket;
        if (ticket >= tiles_here) break;
        const uint32_t tile = t0 + ticket;
        const size_t tile_base = static_cast<size_t>(tile) * RS_TILE;
        const size_t left = n - tile_base;
        const uint32_t valid = left < static_cast<size_t>(RS_TILE) ? static_cast<uint32_t>(left) : RS_TILE;
        uint64_t key[RS_KPT];
        uint32_t val[RS_KPT];
#pragma unroll
        for (int k = 0; k < RS_KPT; ++k) {
            const uint32_t li = wbase + k * 64 + lane;
            if (li < valid) {
                key[k] = kin[tile_base + li];
                val[k] = vin[tile_base + li];
                atomicAdd(&s_hist[digit_of(key[k], shift)], 1u);
            } else {
                key[k] = ~0ull;
                val[k] = 0;
            }
        }
        __syncthreads();
        const uint32_t my_cnt = s_hist[tid];
        uint32_t *mine = status + static_cast<size_t>(tile) * 256 + tid;
        __hip_atomic_store(mine, (ticket == 0 ? ST_INCL : ST_LOCAL) | my_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // look back over the earlier tiles of THIS chunk only (its first tile publishes INCL at once), BEFORE the long ranking:
        // the sooner the inclusive word is out, the shorter everybody else's walk
        uint32_t excl = 0;
        if (ticket != 0) {
            const uint32_t *look = mine - 256;
            uint32_t spins = 0;
            for (;;) {
                const uint32_t v = __hip_atomic_load(look, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((v >> 30) == 0) {
                    if (++spins > RS_SPIN_LIMIT) { ctrl[8] = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                excl += v & ST_MASK;
                if (v & ST_INCL) break;
                look -= 256;
            }
            __hip_atomic_store(mine, ST_INCL | (excl + my_cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }

        uint32_t rnk[RS_KPT];
#pragma unroll
        for (int k = 0; k < RS_KPT; ++k) {
            const uint32_t d = digit_of(key[k], shift);
            uint64_t same = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (d >> b) & 1u;
                const uint64_t bal = __ballot(bit);
                same &= bit ? bal : ~bal;
            }
            const uint32_t before = static_cast<uint32_t>(__popcll(same & lt));
            const uint32_t old = s_cnt[wave][d];
            __builtin_amdgcn_wave_barrier();
            if (before == 0) s_cnt[wave][d] = old + static_cast<uint32_t>(__popcll(same));
            __builtin_amdgcn_wave_barrier();
            rnk[k] = old + before;
        }
        __syncthreads();
        {
            const int d = tid;
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < RS_WAVES; ++w) {
                const uint32_t cc = s_cnt[w][d];
                s_cnt[w][d] = run;
                run += cc;
            }
            const uint32_t start = block_excl_sum<RS_WAVES>(run, s_tmp, nullptr);
            s_start[d] = start;
            s_gbase[d] = s_cbase[d] + excl - start;
        }
        __syncthreads();
        uint32_t pos[RS_KPT];
#pragma unroll
        for (int k = 0; k < RS_KPT; ++k) {
            const uint32_t d = digit_of(key[k], shift);
            pos[k] = s_start[d] + s_cnt[wave][d] + rnk[k];
            s_keys[pos[k]] = key[k];
        }
        __syncthreads();
        uint32_t gi[RS_KPT];
#pragma unroll
        for (int k = 0; k < RS_KPT; ++k) {
            const uint32_t p = k * RS_BLOCK + tid;
            const uint64_t kk = s_keys[p];
            gi[k] = s_gbase[digit_of(kk, shift)] + p;
            if (p < valid) {
                kout[gi[k]] = kk;
                if (next_shift >= 0) atomicAdd(&s_hnext[chunk_of_tile(gi[k] / RS_TILE, g)][digit_of(kk, next_shift)], 1u);
            }
        }
        __syncthreads();
        uint32_t *s_vals = reinterpret_cast<uint32_t *>(s_keys);
#pragma unroll
        for (int k = 0; k < RS_KPT; ++k) s_vals[pos[k]] = val[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RS_KPT; ++k) {
            const uint32_t p = k * RS_BLOCK + tid;
            if (p < valid) vout[gi[k]] = s_vals[p];
        }
    }
    if (next_shift >= 0) {
#pragma unroll
        for (int cc = 0; cc < RS_NCH; ++cc) {
            const uint32_t v = s_hnext[cc][tid];
            if (v) atomicAdd(&H_next[cc * 256 + tid], v);
        }
    }
}

}  // namespace

// Sorts `count` pairs on key bits [begin_bit, end_bit).  keys/vals are the input buffers, *_alt equally sized scratch;
// on return `keys` and `vals` refer to whichever buffer holds the sorted data (the references are swapped per pass).
static int sort_pairs_classic(dk_ctx *ctx, uint64_t *&keys, uint64_t *&keys_alt, uint32_t *&vals, uint32_t *&vals_alt, size_t count,
                              int begin_bit, int end_bit) {
    const size_t ntiles = div_up(count, RS_TILE);
    const size_t tiles_per_chunk = div_up(ntiles, RS_MAX_CHUNKS);
    const size_t nchunks = div_up(ntiles, tiles_per_chunk);
    const size_t mark = ctx->ws_mark();
    uint32_t *tile_hist = ctx->ws_alloc<uint32_t>(ntiles * 256);
    uint32_t *chunk_sum = ctx->ws_alloc<uint32_t>(nchunks * 256);
    if (!tile_hist || !chunk_sum) return DK_E_NOMEM;
    hipStream_t st = ctx->stream;
    for (int shift = begin_bit; shift < end_bit; shift += 8) {
        {
            LaunchScope ls(ctx, K_RADIX_HIST, 8.0 * count);
            k_radix_hist<false><<<dim3(ntiles), dim3(RS_BLOCK), 0, st>>>(keys, nullptr, nullptr, count, shift, tile_hist);
        }
        {
            LaunchScope ls(ctx, K_RADIX_SCAN, 3.0 * 1024.0 * ntiles);
            k_radix_scan_a<<<dim3(nchunks), dim3(256), 0, st>>>(tile_hist, ntiles, tiles_per_chunk, chunk_sum);
            k_radix_scan_b<<<dim3(1), dim3(1024), 0, st>>>(chunk_sum, nchunks);
            k_radix_scan_c<<<dim3(nchunks), dim3(256), 0, st>>>(tile_hist, ntiles, tiles_per_chunk, chunk_sum);
        }
        {
            LaunchScope ls(ctx, K_RADIX_SCATTER, 24.0 * count);
            static const bool xcd = [] { const char *e = getenv("DK_XCD"); return !(e && e[0] == '0'); }();
            const size_t grid = xcd ? 8 * div_up(ntiles, 8) : ntiles;
            k_radix_scatter<false><<<dim3(grid), dim3(RS_BLOCK), 0, st>>>(keys, vals, nullptr, keys_alt, vals_alt, count, shift, tile_hist,
                                                                         nullptr, nullptr, xcd ? static_cast<uint32_t>(ntiles) : 0u);
        }
        DK_HIP(ctx, hipGetLastError());
        std::swap(keys, keys_alt);
        std::swap(vals, vals_alt);
        ctx->stats.sort_passes += 1;
        ctx->stats.sorted_elements += count;
    }
    ctx->ws_release(mark);
    return DK_OK;
}

static int sort_pairs_onesweep(dk_ctx *ctx, uint64_t *&keys, uint64_t *&keys_alt, uint32_t *&vals, uint32_t *&vals_alt, size_t count,
                               int begin_bit, int end_bit) {
    const size_t ntiles = div_up(count, RS_TILE);
    const int npasses = (end_bit - begin_bit + 7) / 8;
    const size_t mark = ctx->ws_mark();
    // ctrl words (ticket, error) sit in front of the status rows so that one memset clears both
    uint32_t *digit_base = ctx->ws_alloc<uint32_t>(static_cast<size_t>(npasses) * 256);
    uint32_t *ctrl = ctx->ws_alloc<uint32_t>(64 + ntiles * 256);
    if (!digit_base || !ctrl) return DK_E_NOMEM;
    uint32_t *status = ctrl + 64;
    uint32_t *d_err = ctx->d_mail + 12;
    hipStream_t st = ctx->stream;
    DK_HIP(ctx, hipMemsetAsync(digit_base, 0, static_cast<size_t>(npasses) * 256 * sizeof(uint32_t), st));
    {
        LaunchScope ls(ctx, K_RADIX_HIST, 8.0 * count);
        const size_t blocks = std::min<size_t>(div_up(count, RS_BLOCK * 8), 2048);
        k_radix_hist_all<<<dim3(blocks), dim3(RS_BLOCK), 0, st>>>(keys, count, begin_bit, npasses, digit_base);
        k_radix_base_scan<<<dim3(npasses), dim3(256), 0, st>>>(digit_base);
    }
    for (int p = 0; p < npasses; ++p) {
        DK_HIP(ctx, hipMemsetAsync(ctrl, 0, (64 + ntiles * 256) * sizeof(uint32_t), st));
        {
            LaunchScope ls(ctx, K_RADIX_SCATTER, 24.0 * count);
            k_radix_scatter<true><<<dim3(ntiles), dim3(RS_BLOCK), 0, st>>>(keys, vals, nullptr, keys_alt, vals_alt, count, begin_bit + 8 * p,
                                                                          status, digit_base + p * 256, ctrl, 0u);
        }
        DK_HIP(ctx, hipGetLastError());
        // fold the error word of this pass into the mailbox (checked once per suffix sort / debug call)
        k_radix_fold_err<<<dim3(1), dim3(1), 0, st>>>(ctrl + 1, d_err);
        std::swap(keys, keys_alt);
        std::swap(vals, vals_alt);
        ctx->stats.sort_passes += 1;
        ctx->stats.sorted_elements += count;
    }
    ctx->ws_release(mark);
    return DK_OK;
}

static int sort_pairs_chunked(dk_ctx *ctx, uint64_t *&keys, uint64_t *&keys_alt, uint32_t *&vals, uint32_t *&vals_alt, size_t count,
                              int begin_bit, int end_bit) {
    const size_t ntiles = div_up(count, RS_TILE);
    const int npasses = (end_bit - begin_bit + 7) / 8;
    ChunkGeom g;
    g.ntiles = static_cast<uint32_t>(ntiles);
    g.tiles_per_chunk = static_cast<uint32_t>(div_up(ntiles, RS_NCH));
    g.div_magic = ((1ull << 40) + g.tiles_per_chunk - 1) / g.tiles_per_chunk;
    const size_t mark = ctx->ws_mark();
    uint32_t *H = ctx->ws_alloc<uint32_t>(static_cast<size_t>(npasses + 1) * RS_NCH * 256);
    uint32_t *ctrl = ctx->ws_alloc<uint32_t>(64 + ntiles * 256);
    if (!H || !ctrl) return DK_E_NOMEM;
    uint32_t *d_err = ctx->d_mail + 12;
    hipStream_t st = ctx->stream;
    DK_HIP(ctx, hipMemsetAsync(H, 0, static_cast<size_t>(npasses + 1) * RS_NCH * 256 * sizeof(uint32_t), st));
    const size_t grid = RS_NCH * std::min<size_t>(div_up(ntiles, RS_NCH), 160);
    {
        LaunchScope ls(ctx, K_RADIX_HIST, 8.0 * count);
        k_chunk_hist<<<dim3(grid), dim3(RS_BLOCK), 0, st>>>(keys, count, begin_bit, g, H);
    }
    for (int p = 0; p < npasses; ++p) {
        DK_HIP(ctx, hipMemsetAsync(ctrl, 0, (64 + ntiles * 256) * sizeof(uint32_t), st));
        {
            LaunchScope ls(ctx, K_RADIX_SCATTER, 24.0 * count);
            k_radix_chunked<<<dim3(grid), dim3(RS_BLOCK), 0, st>>>(keys, vals, keys_alt, vals_alt, count, begin_bit + 8 * p,
                                                                   p + 1 < npasses ? begin_bit + 8 * (p + 1) : -1, g, H + p * RS_NCH * 256,
                                                                   H + (p + 1) * RS_NCH * 256, ctrl);
        }
        DK_HIP(ctx, hipGetLastError());
        k_radix_fold_err<<<dim3(1), dim3(1), 0, st>>>(ctrl + 8, d_err);
        std::swap(keys, keys_alt);
        std::swap(vals, vals_alt);
        ctx->stats.sort_passes += 1;
        ctx->stats.sorted_elements += count;
    }
    ctx->ws_release(mark);
    return DK_OK;
}

int sort_pairs(dk_ctx *ctx, uint64_t *&keys, uint64_t *&keys_alt, uint32_t *&vals, uint32_t *&vals_alt, size_t count,
               int begin_bit, int end_bit) {
    if (count <= 1 || end_bit <= begin_bit) return DK_OK;
    if (count > 0xFFFFFFFEull) return ctx->fail(DK_E_ARG, "sort_pairs: count too large");
    // default: three-phase passes with the XCD-aware tile order (measured 4.2 TB/s on the scatter); DK_SORT=onesweep selects the
    // single-kernel look-back variant (no per-pass histogram, but ticket order defeats the XCD locality: 2.2 TB/s)
    static const std::string mode = [] { const char *e = getenv("DK_SORT"); return std::string(e ? e : ""); }();
    if (count < (1ull << 30)) {
        if (mode == "onesweep") return sort_pairs_onesweep(ctx, keys, keys_alt, vals, vals_alt, count, begin_bit, end_bit);
        if (mode == "chunked") return sort_pairs_chunked(ctx, keys, keys_alt, vals, vals_alt, count, begin_bit, end_bit);
    }
    return sort_pairs_classic(ctx, keys, keys_alt, vals, vals_alt, count, begin_bit, end_bit);
}

// ---- bucketed scatter: dst[idx[i]] = val[i] for a huge, random idx ---------------------------------------------------
// A random 4-byte store costs a whole 64-byte line at the HBM (read-modify-write).  Partition the (idx, val) pairs by the
// top 8 bits of idx first (one stable radix pass over pairs, 16 B/pair), then store bucket by bucket: a bucket's
// destinations span n/256 words (1.5 MiB at n = 1e8), the XCD-aware tile order keeps one bucket on one XCD, and its lines
// are completed in that XCD's L2 before they are written back.
__global__ __launch_bounds__(RS_BLOCK) void k_bucket_store(const uint64_t *__restrict__ pairs, size_t n, uint32_t ntiles,
                                                            uint32_t *__restrict__ dst) {
    const uint32_t per = gridDim.x / 8;
    const uint32_t tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (tile >= ntiles) return;
    const size_t base = static_cast<size_t>(tile) * RS_TILE;
#pragma unroll
    for (int k = 0; k < RS_KPT; ++k) {
        const size_t i = base + static_cast<size_t>(k) * RS_BLOCK + threadIdx.x;
        if (i < n) {
            const uint64_t p = __builtin_nontemporal_load(pairs + i);  // streamed once: keep it out of the way of the destination lines
            dst[static_cast<uint32_t>(p >> 32)] = static_cast<uint32_t>(p);
        }
    }
}

// dst[idx[i]] = val[i] (val == nullptr: = i), i < count; idx values are distinct and < limit.  `scratch` holds count u64.
int scatter_u32_bucketed(dk_ctx *ctx, const uint32_t *idx, const uint32_t *val, size_t count, size_t limit, uint64_t *scratch,
                         uint32_t *dst) {
    if (count == 0) return DK_OK;
    const size_t ntiles = div_up(count, RS_TILE);
    const size_t tiles_per_chunk = div_up(ntiles, RS_MAX_CHUNKS);
    const size_t nchunks = div_up(ntiles, tiles_per_chunk);
    const size_t mark = ctx->ws_mark();
    uint32_t *tile_hist = ctx->ws_alloc<uint32_t>(ntiles * 256);
    uint32_t *chunk_sum = ctx->ws_alloc<uint32_t>(nchunks * 256);
    if (!tile_hist || !chunk_sum) return DK_E_NOMEM;
    hipStream_t st = ctx->stream;
    const unsigned lb = ceil_log2_u64(limit);
    const int shift = 32 + static_cast<int>(lb > 8 ? lb - 8 : 0);
    const size_t grid = 8 * div_up(ntiles, 8);
    {
        LaunchScope ls(ctx, K_RADIX_HIST, 8.0 * count);
        k_radix_hist<true><<<dim3(ntiles), dim3(RS_BLOCK), 0, st>>>(nullptr, idx, val, count, shift, tile_hist);
    }
    {
        LaunchScope ls(ctx, K_RADIX_SCAN, 3.0 * 1024.0 * ntiles);
        k_radix_scan_a<<<dim3(nchunks), dim3(256), 0, st>>>(tile_hist, ntiles, tiles_per_chunk, chunk_sum);
        k_radix_scan_b<<<dim3(1), dim3(1024), 0, st>>>(chunk_sum, nchunks);
        k_radix_scan_c<<<dim3(nchunks), dim3(256), 0, st>>>(tile_hist, ntiles, tiles_per_chunk, chunk_sum);
    }
    {
        LaunchScope ls(ctx, K_RADIX_SCATTER, 16.0 * count);
        k_radix_scatter<false, true><<<dim3(grid), dim3(RS_BLOCK), 0, st>>>(nullptr, idx, val, scratch, nullptr, count, shift, tile_hist,
                                                                              nullptr, nullptr, static_cast<uint32_t>(ntiles));
    }
    {
        LaunchScope ls(ctx, K_BUCKET_STORE, 12.0 * count);
        // 72 KiB of (unused) dynamic LDS caps residency at 2 workgroups per CU: ~260 K pairs in flight per XCD, less than one
        // bucket, so the lines of the bucket being written stay in that XCD's 4 MiB L2 until they are complete
        static const size_t lds_cap = [] { const char *e = getenv("DK_BUCKET_LDS"); return e ? static_cast<size_t>(atoi(e)) : size_t(72 * 1024); }();
        k_bucket_store<<<dim3(grid), dim3(RS_BLOCK), lds_cap, st>>>(scratch, count, static_cast<uint32_t>(ntiles), dst);
    }
    DK_HIP(ctx, hipGetLastError());
    ctx->ws_release(mark);
    return DK_OK;
}

// error word of the look-back path (non-zero: a spin hit its bound); resets it
int sort_check_error(dk_ctx *ctx) {
    DK_HIP(ctx, hipMemcpyAsync(ctx->h_mail + 12, ctx->d_mail + 12, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    DK_HIP(ctx, hipMemsetAsync(ctx->d_mail + 12, 0, sizeof(uint32_t), ctx->stream));
    DK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->h_mail[12]) return ctx->fail(DK_E_INTERNAL, "radix sort: decoupled look-back timed out");
    return DK_OK;
}

}  // namespace dk
