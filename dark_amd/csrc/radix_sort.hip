// Stable LSD radix sort of (u64 key, u32 value) pairs, 8 bits per pass -- the workhorse of the suffix sort.
// It takes the place of all the sorting work inside saca() (src/saca.rs:270-340): bucket placement, the induced
// sorts and the naming pass become "sort by packed prefix, then by the symbols / ranks further on".
//
// Per pass, three phases:
//   k_radix_hist        per-tile 256-bin digit histograms (LDS atomics over 16 private copies: text digits are skewed); from the second
//                       pass on they are counted from the one-byte DIGIT PLANE the previous pass's scatter wrote beside the pairs
//                       (k_radix_hist_plane: n bytes read instead of 8 n)
//   k_radix_scan_a/b/c  digit-major exclusive scan of the tile histograms -> global offset of every (tile, digit)
//   k_radix_scatter     a tile ranks its pairs stably with wave64 ballots (match-any over the 8 digit bits), reorders them in LDS so
//                       that equal digits are contiguous, then writes runs to HBM.  XCD-aware tile order: every XCD takes one
//                       contiguous range of tiles, so the short output runs of neighbouring tiles meet in the same L2.
// Algorithmic bytes per pass: scatter 12 B/pair read + 12 (+1: digit plane) B/pair written; histogram 8 B/key, 1 B with the plane.
// (Two single-kernel-per-pass variants with decoupled look-back -- global ticket order, and per-XCD chunks with the next pass's
// histograms accumulated while scattering -- were built and measured in round 1: 2.2 TB/s and 1.04 ms per pass of 1e8 pairs against
// 0.81 ms for these three phases.  They lost and were removed; DESIGN.md section 4.1 keeps the numbers.)
#include "context.hpp"
#include <algorithm>
#include <cstdlib>
#include <string>

#include "device_util.hpp"

namespace dk {
namespace {

#ifndef DK_RS_BLOCK
#define DK_RS_BLOCK 256
#endif
constexpr int RS_BLOCK = DK_RS_BLOCK;        // threads per workgroup.  512 (8192-pair tiles, twice as long output runs) was measured
                                             // in round 2: no faster on uniform digits (ACGT 8.9 ms either way), slower on text (5.5
                                             // against 4.3 ms) -- the scatter is not short of coalescing, it sits at 76 % of the copy rate
constexpr int RS_WAVES = RS_BLOCK / 64;
constexpr int RS_KPT = 16;                   // pairs per thread
constexpr int RS_TILE = RS_BLOCK * RS_KPT;   // 4096 pairs per workgroup
constexpr int RS_TEXT_AHEAD = 64 + 16;       // TextKeys: codes staged beyond the tile (spk <= 64)
constexpr int RS_MAX_CHUNKS = 256;

__device__ __forceinline__ uint32_t digit_of(uint64_t k, int shift) { return static_cast<uint32_t>(k >> shift) & 0xFFu; }

// PAIRS = true: the key of element i is (hi[i] << 32) | lo[i], read from two u32 arrays, and there is no value array
template <bool PAIRS>
__device__ __forceinline__ uint64_t load_key(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ hi,
                                             const uint32_t *__restrict__ lo, size_t i) {
    if (PAIRS) return (static_cast<uint64_t>(hi[i]) << 32) | (lo ? lo[i] : static_cast<uint32_t>(i));  // lo == nullptr: lo[i] = i
    return keys[i];
}

// ---- first pass straight from the text -------------------------------------------------------------------------------------------
// The initial sort of the suffix sort needs no key array to start with: the key of position i is the packed codes of T[i .. i+spk)
// (shifted up one byte with the code of T[i-1] below when with_prev), its value is i.  The first pass's histogram and scatter
// kernels build the keys of their tile in LDS from n bytes of text instead of reading 12 n bytes of (key, index) pairs that a
// separate kernel would have had to write first.
// the codes of T[b0 .. b0 + RS_TILE + RS_TEXT_AHEAD) into s_c (zero past the end of the text), the code of the symbol in front of the
// tile into s_c[RS_TILE + RS_TEXT_AHEAD]; ends with a barrier
__device__ __forceinline__ void text_stage_codes(const TextKeys &tk, size_t b0, uint8_t *s_code, uint8_t *s_c) {
    const int tid = threadIdx.x;
    for (int i = tid; i < 256; i += RS_BLOCK) s_code[i] = tk.code[i];
    __syncthreads();
    if (tid == 0) s_c[RS_TILE + RS_TEXT_AHEAD] = s_code[tk.t[b0 ? b0 - 1 : tk.n - 1]];  // code of the symbol in front of the tile
    const bool aligned = (reinterpret_cast<uintptr_t>(tk.t) & 15) == 0;
    for (int o = tid * 16; o < RS_TILE + RS_TEXT_AHEAD; o += RS_BLOCK * 16) {
        const size_t p = b0 + o;
        uint8_t raw[16];
        if (aligned && p + 16 <= tk.n) {
            *reinterpret_cast<uint4 *>(raw) = *reinterpret_cast<const uint4 *>(tk.t + p);
#pragma unroll
            for (int b = 0; b < 16; ++b) raw[b] = s_code[raw[b]];
        } else {
#pragma unroll
            for (int b = 0; b < 16; ++b) raw[b] = (p + b < tk.n) ? s_code[tk.t[p + b]] : 0;  // zero padding past the end of the text
        }
        *reinterpret_cast<uint4 *>(s_c + o) = *reinterpret_cast<const uint4 *>(raw);
    }
    __syncthreads();
}

template <class Sink>  // sink(position inside the tile, key)
__device__ __forceinline__ void text_tile_keys(const TextKeys &tk, size_t b0, uint8_t *s_code, uint8_t *s_c, Sink &&sink) {
    const int tid = threadIdx.x;
    text_stage_codes(tk, b0, s_code, s_c);
    const int base = tid * RS_KPT;  // RS_KPT consecutive positions per thread: a sliding window over the codes
    const int bits = tk.bits, spk = tk.spk;
    const uint64_t mask = (spk * bits >= 64) ? ~0ull : ((1ull << (spk * bits)) - 1ull);
    // 16 staged codes from byte offset `off` on (any alignment), as two little-endian words: three aligned 8-byte LDS reads
    auto bytes16 = [&](int off, uint64_t &lo, uint64_t &hi) {
        const uint64_t *q = reinterpret_cast<const uint64_t *>(s_c + (off & ~7));
        const unsigned sh = static_cast<unsigned>(off & 7) * 8u;
        const uint64_t a = q[0], b = q[1], c = q[2];
        lo = sh ? (a >> sh) | (b << (64u - sh)) : a;
        hi = sh ? (b >> sh) | (c << (64u - sh)) : b;
    };
    uint64_t key = 0;
    for (int j0 = 0; j0 < spk; j0 += 16) {  // first key of the thread: the codes of positions base .. base + spk - 1
        uint64_t lo, hi;
        bytes16(base + j0, lo, hi);
        const int cnt = spk - j0 < 16 ? spk - j0 : 16;
        for (int j = 0; j < cnt; ++j) {
            const uint64_t w = j < 8 ? lo : hi;
            key = (key << bits) | ((w >> (8 * (j & 7))) & 0xFFu);
        }
    }
    uint64_t nlo, nhi, plo = 0, phi = 0;
    bytes16(base + spk, nlo, nhi);  // the codes that enter the window: positions base + spk .. base + spk + 15
    uint32_t before = s_c[RS_TILE + RS_TEXT_AHEAD];
    if (tk.with_prev) {  // codes of positions base - 1 .. base + 14
        if (base) { bytes16(base - 1, plo, phi); } else { bytes16(0, plo, phi); phi = (phi << 8) | (plo >> 56); plo = (plo << 8) | before; }
    }
#pragma unroll
    for (int g = 0; g < RS_KPT; ++g) {
        if (g) {
            const uint64_t w = g - 1 < 8 ? nlo : nhi;
            key = ((key << bits) | ((w >> (8 * ((g - 1) & 7))) & 0xFFu)) & mask;
        }
        uint64_t out = key;
        if (tk.with_prev) {
            const uint64_t w = g < 8 ? plo : phi;
            out = (key << 8) | ((w >> (8 * (g & 7))) & 0xFFu);
        }
        sink(base + g, out);
    }
}

// 16 private copies of the histogram (copy = lane mod 16): text digits are skewed, and LDS atomics of one wave instruction that
// hit the same address are serialised; spreading them over copies cuts that contention up to 16 x.
constexpr int RS_HCOPIES = 16;

template <bool PAIRS, bool TEXT = false>
__global__ __launch_bounds__(RS_BLOCK) void k_radix_hist(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ hi,
                                                          const uint32_t *__restrict__ lo, size_t n, int shift,
                                                          uint32_t *__restrict__ tile_hist, TextKeys tk) {
    __shared__ uint32_t h[RS_HCOPIES][256];
    __shared__ __attribute__((aligned(16))) uint8_t s_c[TEXT ? RS_TILE + RS_TEXT_AHEAD + 16 : 16];
    __shared__ uint8_t s_code[TEXT ? 256 : 4];
    const int tid = threadIdx.x;
    for (int i = tid; i < RS_HCOPIES * 256; i += RS_BLOCK) (&h[0][0])[i] = 0;
    __syncthreads();
    uint32_t *mine = h[tid & (RS_HCOPIES - 1)];
    const size_t base = static_cast<size_t>(blockIdx.x) * RS_TILE;
    if (TEXT) {
        // The first pass sorts by the lowest eight sorted bits of the key = the low eight bits of the packed window of spk codes: only the
        // last ceil(8 / bits) symbols of the window reach them, so the digit slides along the staged codes in three instructions per
        // position -- no key is built (the scatter builds them).  `shift` is the bit the sorted part of the key starts at.
        text_stage_codes(tk, base, s_code, s_c);
        const int p0 = tid * RS_KPT, bits = tk.bits, spk = tk.spk;
        const int last = (8 + bits - 1) / bits < spk ? (8 + bits - 1) / bits : spk;  // symbols that reach the digit
        const uint32_t dmask = spk * bits >= 8 ? 0xFFu : (1u << (spk * bits)) - 1u;  // (a window shorter than the digit)
        uint32_t w = 0;
        for (int j = spk - last; j < spk; ++j) w = (w << bits) | s_c[p0 + j];
#pragma unroll
        for (int g = 0; g < RS_KPT; ++g) {
            if (base + p0 + g < n) atomicAdd(&mine[w & dmask], 1u);
            w = (w << bits) | s_c[p0 + spk + g];
        }
    } else if (!PAIRS && base + RS_TILE <= n) {  // full tile: two keys per 16-byte load (order inside the tile is irrelevant here)
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
    if (tid < 256) {
        uint32_t sum = 0;
#pragma unroll
        for (int c = 0; c < RS_HCOPIES; ++c) sum += h[c][tid];
        tile_hist[static_cast<size_t>(blockIdx.x) * 256 + tid] = sum;
    }
}

// The same histograms from the DIGIT PLANE the previous pass's scatter left behind: one byte per pair (the digit this pass sorts by, in
// the order this pass reads the pairs) instead of the eight bytes of the key -- the histogram pass is a pure streaming read, and seven
// of its eight bytes were never looked at.
__global__ __launch_bounds__(RS_BLOCK) void k_radix_hist_plane(const uint8_t *__restrict__ plane, size_t n, uint32_t *__restrict__ tile_hist) {
    __shared__ uint32_t h[RS_HCOPIES][256];
    const int tid = threadIdx.x;
    for (int i = tid; i < RS_HCOPIES * 256; i += RS_BLOCK) (&h[0][0])[i] = 0;
    __syncthreads();
    uint32_t *mine = h[tid & (RS_HCOPIES - 1)];
    const size_t base = static_cast<size_t>(blockIdx.x) * RS_TILE;
    static_assert(RS_TILE == RS_BLOCK * 16, "one 16-byte load per thread");
    if (base + RS_TILE <= n && (reinterpret_cast<uintptr_t>(plane) & 15) == 0) {
        const uint4 v = reinterpret_cast<const uint4 *>(plane + base)[tid];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int b = 0; b < 4; ++b) atomicAdd(&mine[(w[k] >> (8 * b)) & 0xFFu], 1u);
        }
    } else {
        for (int k = 0; k < 16; ++k) {
            const size_t i = base + static_cast<size_t>(k) * RS_BLOCK + tid;
            if (i < n) atomicAdd(&mine[plane[i]], 1u);
        }
    }
    __syncthreads();
    if (tid < 256) {
        uint32_t sum = 0;
#pragma unroll
        for (int c = 0; c < RS_HCOPIES; ++c) sum += h[c][tid];
        tile_hist[static_cast<size_t>(blockIdx.x) * 256 + tid] = sum;
    }
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
// phase B (one workgroup of 1024): digit-major exclusive scan of the chunk sums; 4 threads share a digit's column.  Eight rows are
// loaded before any is rewritten, so that the loads are in flight together.  (All 64 rows of a thread in registers at once was slower:
// 39 against 16 us -- the guarded 64-entry array went to scratch.)
__global__ __launch_bounds__(1024) void k_radix_scan_b(uint32_t *__restrict__ chunk_sum, size_t nchunks) {
    __shared__ uint32_t s_tmp[16 + 1];
    __shared__ uint32_t s_part[4][256];
    const int d = threadIdx.x & 255, part = threadIdx.x >> 8;
    const size_t q = (nchunks + 3) / 4;
    const size_t g0 = part * q, g1 = g0 + q < nchunks ? g0 + q : nchunks;
    uint32_t run = 0;
    for (size_t g = g0; g < g1; g += 8) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = g + k < g1 ? chunk_sum[(g + k) * 256 + d] : 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (g + k < g1) chunk_sum[(g + k) * 256 + d] = run;
            run += v[k];
        }
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
    for (size_t g = g0; g < g1; g += 8) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = g + k < g1 ? chunk_sum[(g + k) * 256 + d] : 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (g + k < g1) chunk_sum[(g + k) * 256 + d] = v[k] + off;
    }
}
// all three phases in one workgroup, for sorts of few tiles (the big-group lists: a launch costs more than this loop)
__global__ __launch_bounds__(256) void k_radix_scan_small(uint32_t *__restrict__ tile_hist, size_t ntiles) {
    __shared__ uint32_t s_tmp[4 + 1];
    const int d = threadIdx.x;
    uint32_t run = 0;
#pragma unroll 8
    for (size_t t = 0; t < ntiles; ++t) {
        const uint32_t v = tile_hist[t * 256 + d];
        tile_hist[t * 256 + d] = run;
        run += v;
    }
    const uint32_t base = block_excl_sum<4>(run, s_tmp, nullptr);
#pragma unroll 8
    for (size_t t = 0; t < ntiles; ++t) tile_hist[t * 256 + d] += base;
}

// phase C: tile counts -> exclusive global offsets (eight rows loaded before any is rewritten: the loads are in flight together)
__global__ __launch_bounds__(256) void k_radix_scan_c(uint32_t *__restrict__ tile_hist, size_t ntiles, size_t tiles_per_chunk,
                                                       const uint32_t *__restrict__ chunk_sum) {
    const size_t g = blockIdx.x;
    const int d = threadIdx.x;
    const size_t t0 = g * tiles_per_chunk;
    const size_t t1 = t0 + tiles_per_chunk < ntiles ? t0 + tiles_per_chunk : ntiles;
    uint32_t run = chunk_sum[g * 256 + d];
    for (size_t t = t0; t < t1; t += 8) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = t + k < t1 ? tile_hist[(t + k) * 256 + d] : 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (t + k < t1) tile_hist[(t + k) * 256 + d] = run;
            run += v[k];
        }
    }
}

// tile_offs: the scanned per-tile digit offsets.  next_digit (may be null): the digit plane for the next pass (k_radix_hist_plane).  xcd_tiles != 0: XCD-aware tile order over a grid of 8 * ceil(ntiles / 8) blocks.
// Three workgroups per CU, not the four that registers and LDS would allow: with a fourth tile in flight per CU the L2 no longer
// merges the short output runs of neighbouring tiles before it has to evict them (measured, 1e8 text: 3.95 ms of scatter per sort
// against 3.6 ms; two per CU: 3.65 ms but slower overall).  The kernel is bound by that, not by its ballots: a third fewer vector
// instructions in the ranking (wave_match) did not move it.
template <bool PAIRS = false, bool TEXT = false>
__global__ __launch_bounds__(RS_BLOCK) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_radix_scatter(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                             const uint32_t *__restrict__ pair_lo,
                                                             uint64_t *__restrict__ kout, uint32_t *__restrict__ vout, size_t n,
                                                             int shift, const uint32_t *__restrict__ tile_offs, uint32_t xcd_tiles,
                                                             TextKeys tk, uint8_t *__restrict__ next_digit, SortFinalOut fin) {
    __shared__ uint64_t s_keys[RS_TILE];          // tile of keys in digit order; reused for the values
    // per-wave digit counters (then exclusive over waves) | tile-local start of each digit | global offset of the digit minus its
    // tile-local start.  TEXT: the same 6 KiB first hold the staged codes of the tile (the counters are cleared afterwards).
    __shared__ __attribute__((aligned(16))) uint32_t s_tab[RS_WAVES * 256 + 512];
    __shared__ uint32_t s_tmp[RS_WAVES + 1];
    __shared__ uint8_t s_code[TEXT ? 256 : 4];
    static_assert(!TEXT || sizeof(uint32_t) * (RS_WAVES * 256 + 512) >= RS_TILE + RS_TEXT_AHEAD + 16, "code staging does not fit");
    uint32_t (*s_cnt)[256] = reinterpret_cast<uint32_t (*)[256]>(s_tab);
    uint32_t *s_start = s_tab + RS_WAVES * 256;
    uint32_t *s_gbase = s_start + 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t tile = blockIdx.x;
    if (xcd_tiles) {
        // XCD-aware order: blocks b and b+8 share an XCD (round-robin dispatch, speed only); give every XCD one contiguous
        // range of tiles so that neighbouring output runs meet in the same L2.  Grid = 8 * ceil(ntiles / 8).
        const uint32_t per = gridDim.x / 8;
        tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
        if (tile >= xcd_tiles) return;
    }
    const size_t tile_base = static_cast<size_t>(tile) * RS_TILE;
    const size_t left = n - tile_base;
    const uint32_t valid = left < static_cast<size_t>(RS_TILE) ? static_cast<uint32_t>(left) : RS_TILE;
    // the tile's global digit offsets are needed after the ranking only: asked for here, their latency hides behind it
    const uint32_t goff_early = tid < 256 ? tile_offs[static_cast<size_t>(tile) * 256 + tid] : 0u;

    if (TEXT) {  // keys of the tile, in position order
        text_tile_keys(tk, tile_base, s_code, reinterpret_cast<uint8_t *>(s_tab), [&](int o, uint64_t key) { s_keys[o] = key; });
        __syncthreads();
    } else {
        for (int i = tid; i < RS_WAVES * 256; i += RS_BLOCK) s_tab[i] = 0;
        __syncthreads();
    }

    // wave w owns pairs [w*1024, (w+1)*1024) of the tile, lane-striped so that loads coalesce
    uint64_t key[RS_KPT];
    uint32_t val[RS_KPT];
    const uint32_t wbase = static_cast<uint32_t>(wave) * (64 * RS_KPT);
#pragma unroll
    for (int k = 0; k < RS_KPT; ++k) {
        const uint32_t li = wbase + k * 64 + lane;
        if (li < valid) {
            key[k] = TEXT ? s_keys[li] : load_key<PAIRS>(kin, vin, pair_lo, tile_base + li);  // PAIRS: vin holds the high words
            val[k] = TEXT ? static_cast<uint32_t>(tile_base + li) : (PAIRS ? 0u : vin[tile_base + li]);
        } else {
            key[k] = ~0ull;  // padding sorts behind every real pair of the tile and is never written
            val[k] = 0;
        }
    }
    if (TEXT) {  // the codes have been consumed: their place becomes the digit counters
        __syncthreads();
        for (int i = tid; i < RS_WAVES * 256; i += RS_BLOCK) s_tab[i] = 0;
        __syncthreads();
    }

    // stable rank inside the wave: lanes holding the same digit find each other with 8 ballots
    uint32_t rnk[RS_KPT];
#pragma unroll
    for (int k = 0; k < RS_KPT; ++k) {
        const uint32_t d = digit_of(key[k], shift);
        const LaneSet same = wave_match<8>(d, ~0ull);
        const uint32_t before = same.before();
        const uint32_t old = s_cnt[wave][d];
        __builtin_amdgcn_wave_barrier();
        if (before == 0) s_cnt[wave][d] = old + same.count();
        __builtin_amdgcn_wave_barrier();
        rnk[k] = old + before;
    }
    __syncthreads();

    {   // threads 0..255, one per digit: exclusive over waves, then (all threads take part in the scan) exclusive over digits
        const int d = tid & 255;
        const bool owner = tid < 256;
        uint32_t run = 0, goff = 0;
        if (owner) {
#pragma unroll
            for (int w = 0; w < RS_WAVES; ++w) {
                const uint32_t c = s_cnt[w][d];
                s_cnt[w][d] = run;
                run += c;
            }
            goff = goff_early;
        }
        const uint32_t start = block_excl_sum<RS_WAVES>(run, s_tmp, nullptr);
        if (owner) {
            s_start[d] = start;
            s_gbase[d] = goff - start;
        }
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
            if (next_digit) next_digit[gi[k]] = static_cast<uint8_t>(digit_of(kk, shift + 8));  // what the next pass's histogram reads
            if (fin.bwt) fin.bwt[gi[k]] = fin.inv_code[kk & 0xFFu];  // last pass of the suffix sort's initial sort: L rides in the key's low byte
        }
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
        if (p < valid) {
            const uint32_t v = s_vals[p];
            vout[gi[k]] = v;
            if (fin.origin && v == 0) *fin.origin = gi[k];  // suffix 0 (if it is not final yet, the stage that makes it final writes again)
        }
    }
}

// ---- small inputs: the whole sort in ONE workgroup --------------------------------------------------------------------------------
// The big-group lists of the later suffix-sort rounds hold a few thousand pairs: five launches per digit (histogram, three scans,
// scatter) would be nothing but launch latency.  Up to SS_MAX pairs are sorted by one workgroup of 1024 threads, all passes inside
// the kernel: per pass every wave ranks its pairs with the same ballots as k_radix_scatter, the pairs are reordered through LDS, and
// stay in registers (position-striped) between passes.
constexpr int SS_BLOCK = 1024;
constexpr int SS_WAVES = SS_BLOCK / 64;
constexpr int SS_KPT = 8;
constexpr int SS_MAX = SS_BLOCK * SS_KPT;  // 8192 pairs

__global__ __launch_bounds__(SS_BLOCK) void k_radix_sort_small(uint64_t *__restrict__ keys, uint32_t *__restrict__ vals, uint32_t count,
                                                               int begin_bit, int end_bit) {
    __shared__ uint64_t s_keys[SS_MAX];
    __shared__ uint32_t s_vals[SS_MAX];
    __shared__ uint32_t s_cnt[SS_WAVES][256];
    __shared__ uint32_t s_start[256];
    __shared__ uint32_t s_tmp[SS_WAVES + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // wave w owns positions [w * 512, (w + 1) * 512), lane-striped: position order = (wave, k, lane)
    const uint32_t wbase = static_cast<uint32_t>(wave) * (64 * SS_KPT);
    uint64_t key[SS_KPT];
    uint32_t val[SS_KPT];
#pragma unroll
    for (int k = 0; k < SS_KPT; ++k) {
        const uint32_t li = wbase + k * 64 + lane;
        key[k] = li < count ? keys[li] : ~0ull;  // padding: sorts last in every pass (all-ones digits), stays behind the real pairs (stable)
        val[k] = li < count ? vals[li] : 0u;
    }
    for (int shift = begin_bit; shift < end_bit; shift += 8) {
        for (int i = tid; i < SS_WAVES * 256; i += SS_BLOCK) (&s_cnt[0][0])[i] = 0;
        __syncthreads();
        uint32_t rnk[SS_KPT];
#pragma unroll
        for (int k = 0; k < SS_KPT; ++k) {
            const uint32_t d = digit_of(key[k], shift);
            const LaneSet same = wave_match<8>(d, ~0ull);
            const uint32_t before = same.before();
            const uint32_t old = s_cnt[wave][d];
            __builtin_amdgcn_wave_barrier();
            if (before == 0) s_cnt[wave][d] = old + same.count();
            __builtin_amdgcn_wave_barrier();
            rnk[k] = old + before;
        }
        __syncthreads();
        {
            const int d = tid & 255;
            uint32_t run = 0;
            if (tid < 256) {
#pragma unroll
                for (int w = 0; w < SS_WAVES; ++w) {
                    const uint32_t c = s_cnt[w][d];
                    s_cnt[w][d] = run;
                    run += c;
                }
            }
            const uint32_t start = block_excl_sum<SS_WAVES>(run, s_tmp, nullptr);
            if (tid < 256) s_start[d] = start;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SS_KPT; ++k) {
            const uint32_t d = digit_of(key[k], shift);
            const uint32_t p = s_start[d] + s_cnt[wave][d] + rnk[k];
            s_keys[p] = key[k];
            s_vals[p] = val[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SS_KPT; ++k) {
            const uint32_t li = wbase + k * 64 + lane;
            key[k] = s_keys[li];
            val[k] = s_vals[li];
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < SS_KPT; ++k) {
        const uint32_t li = wbase + k * 64 + lane;
        if (li < count) { keys[li] = key[k]; vals[li] = val[k]; }
    }
}

}  // namespace

// Sorts `count` pairs on key bits [begin_bit, end_bit).  keys/vals are the input buffers, *_alt equally sized scratch;
// on return `keys` and `vals` refer to whichever buffer holds the sorted data (the references are swapped per pass).
// Wider digits were built and measured in round 3 (passes of 10 / 9 bits over super-tiles of 16384 pairs with 16-bit LDS counters:
// 56 bits in six passes, 40 bits in four): every pass got 1.6 x slower -- 1024 digits cut a tile's output into runs of four pairs
// (32 bytes of keys, 16 of values), more partial lines than the L2 merges -- 2^28 ACGT 17.5 against 13.4 ms, 1e8 text 15.2 against
// 14.0 ms (profiles/r03_wide_digit_experiment.log).  Removed; eight bits per pass stay.
static int sort_pairs_classic(dk_ctx *ctx, uint64_t *&keys, uint64_t *&keys_alt, uint32_t *&vals, uint32_t *&vals_alt, size_t count,
                              int begin_bit, int end_bit, const TextKeys *text, const SortFinalOut *final_out) {
    const size_t ntiles = div_up(count, RS_TILE);
    const size_t tiles_per_chunk = div_up(ntiles, RS_MAX_CHUNKS);
    const size_t nchunks = div_up(ntiles, tiles_per_chunk);
    const size_t mark = ctx->ws_mark();
    uint32_t *tile_hist = ctx->ws_alloc<uint32_t>(ntiles * 256);
    uint32_t *chunk_sum = ctx->ws_alloc<uint32_t>(nchunks * 256);
    // digit plane: every scatter but the last leaves the next pass's digits behind, one byte per pair.  The histograms get 2x faster,
    // the scatters 10 % slower (the plane's 16-byte runs cost as many L2 write requests as the 64-byte runs of the values).  Net gain,
    // round 3: 2.7 % of the whole suffix sort at 1e8 pairs (12.07 -> 11.74 ms), 1.8 % at 2^28; on from 2^26 pairs, where the keys
    // (8 bytes per pair) no longer fit the 256 MiB Infinity Cache.
    const int plane_mode = DK_KNOB("DK_DIGIT_PLANE", -1);  // tuning build: 1 / 0 = always (from 2^20 pairs) / never
    const size_t plane_from = plane_mode == 1 ? (size_t(1) << 20) : (size_t(1) << 26);
    uint8_t *plane = plane_mode != 0 && end_bit - begin_bit > 8 && count >= plane_from ? ctx->ws_try_alloc<uint8_t>(count) : nullptr;  // optional: the sort runs without it
    if (!tile_hist || !chunk_sum) return DK_E_NOMEM;
    hipStream_t st = ctx->stream;
    for (int shift = begin_bit; shift < end_bit; shift += 8) {
        const bool have_plane = plane && shift > begin_bit;      // written by the previous pass
        uint8_t *emit = plane && shift + 8 < end_bit ? plane : nullptr;  // read by the next one
        const TextKeys *tk = text && shift == begin_bit ? text : nullptr;
        const bool last = shift + 8 >= end_bit;
        const SortFinalOut fin = last && final_out ? *final_out : SortFinalOut{};
        uint32_t *vout = fin.vals ? fin.vals : vals_alt;
        {
            LaunchScope ls(ctx, tk ? K_RADIX_HIST_TEXT : K_RADIX_HIST, (tk ? 1.0 : have_plane ? 1.0 : 8.0) * count);
            if (tk)
                k_radix_hist<false, true><<<dim3(ntiles), dim3(RS_BLOCK), 0, st>>>(nullptr, nullptr, nullptr, count, shift, tile_hist, *tk);
            else if (have_plane)
                k_radix_hist_plane<<<dim3(ntiles), dim3(RS_BLOCK), 0, st>>>(plane, count, tile_hist);
            else
                k_radix_hist<false><<<dim3(ntiles), dim3(RS_BLOCK), 0, st>>>(keys, nullptr, nullptr, count, shift, tile_hist, TextKeys{});
        }
        {
            LaunchScope ls(ctx, K_RADIX_SCAN, 3.0 * 1024.0 * ntiles);
            if (ntiles <= 1024) {
                k_radix_scan_small<<<dim3(1), dim3(256), 0, st>>>(tile_hist, ntiles);
            } else {
                k_radix_scan_a<<<dim3(nchunks), dim3(256), 0, st>>>(tile_hist, ntiles, tiles_per_chunk, chunk_sum);
                k_radix_scan_b<<<dim3(1), dim3(1024), 0, st>>>(chunk_sum, nchunks);
                k_radix_scan_c<<<dim3(nchunks), dim3(256), 0, st>>>(tile_hist, ntiles, tiles_per_chunk, chunk_sum);
            }
        }
        {
            LaunchScope ls(ctx, tk ? K_RADIX_SCATTER_TEXT : K_RADIX_SCATTER, ((tk ? 13.0 : 24.0) + (emit ? 1.0 : 0.0) + (fin.bwt ? 1.0 : 0.0)) * count);
            const bool xcd = DK_KNOB("DK_XCD", 1) != 0;
            const size_t grid = xcd ? 8 * div_up(ntiles, 8) : ntiles;
            if (tk)
                k_radix_scatter<false, true><<<dim3(grid), dim3(RS_BLOCK), 0, st>>>(nullptr, nullptr, nullptr, keys_alt, vout, count, shift, tile_hist,
                                                                                   xcd ? static_cast<uint32_t>(ntiles) : 0u, *tk, emit, fin);
            else
                k_radix_scatter<false><<<dim3(grid), dim3(RS_BLOCK), 0, st>>>(keys, vals, nullptr, keys_alt, vout, count, shift, tile_hist,
                                                                             xcd ? static_cast<uint32_t>(ntiles) : 0u, TextKeys{}, emit, fin);
        }
        DK_HIP(ctx, hipGetLastError());
        std::swap(keys, keys_alt);
        if (!fin.vals) std::swap(vals, vals_alt);  // (the last pass wrote the values to their final home: both ping-pong buffers are free)
        ctx->stats.sort_passes += 1;
        ctx->stats.sorted_elements += count;
    }
    ctx->ws_release(mark);
    return DK_OK;
}

// text != nullptr: the pairs are (key of position i per TextKeys, i) for i < count = text->n; keys / vals need not hold anything on
// entry (they are scratch), the first pass builds the keys from the text.
int sort_pairs(dk_ctx *ctx, uint64_t *&keys, uint64_t *&keys_alt, uint32_t *&vals, uint32_t *&vals_alt, size_t count,
               int begin_bit, int end_bit, const TextKeys *text, const SortFinalOut *final_out) {
    if (final_out && !text) return ctx->fail(DK_E_INTERNAL, "sort_pairs: a final destination only with the text pass");
    if (text && begin_bit != (text->with_prev ? 8 : 0)) return ctx->fail(DK_E_INTERNAL, "sort_pairs: the text pass sorts from the key's first sorted bit");
    if (count > 0xFFFFFFFEull) return ctx->fail(DK_E_ARG, "sort_pairs: count too large");
    if (text && (count != text->n || end_bit <= begin_bit)) return ctx->fail(DK_E_INTERNAL, "sort_pairs: text pass needs at least one digit");
    if (!text && (count <= 1 || end_bit <= begin_bit)) return DK_OK;
    if (!text && count <= static_cast<size_t>(SS_MAX)) {  // in place, one launch
        const int npasses = (end_bit - begin_bit + 7) / 8;
        {
            LaunchScope ls(ctx, K_RADIX_SORT_SMALL, 24.0 * count);
            k_radix_sort_small<<<dim3(1), dim3(SS_BLOCK), 0, ctx->stream>>>(keys, vals, static_cast<uint32_t>(count), begin_bit, end_bit);
        }
        DK_HIP(ctx, hipGetLastError());
        ctx->stats.sort_passes += static_cast<uint32_t>(npasses);
        ctx->stats.sorted_elements += count * static_cast<size_t>(npasses);
        return DK_OK;
    }
    return sort_pairs_classic(ctx, keys, keys_alt, vals, vals_alt, count, begin_bit, end_bit, text, final_out);
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

// ---- inverse permutation through LDS windows: rank[sa[p]] = p for a permutation sa of 0..n-1, n <= 2^27 ------------------------------
// The bucketed store above leans on the L2 to complete the destination lines of a 1.5 MiB window before it writes them back; the
// counters say it does not (round 2, 1e8: WRITE_SIZE 2.27 GB for 0.40 GB of rank stores).  Here the destination is cut into windows of
// W = 2^wbits <= 2^15 words that fit the LDS, and every word leaves for HBM exactly once, in a full coalesced line:
//   k_isa_split<true>   pairs (sa[p], p) into at most 64 SECTIONS of 64 windows each
//   k_isa_split<false>  every section into its windows
//   k_isa_assemble      one workgroup per window: pairs -> LDS[suffix - d W] = p -> rank[d W ...] streamed out
// sa is a permutation, so a section / a window holds EXACTLY the suffixes of its range: its pairs occupy that very range of the pair
// array.  No counting pass and no scan: a tile (4096 pairs through LDS, at most 64 bins, runs of 512 bytes on average) reserves its
// place in every bin with one global atomic per bin on a counter that starts at the bin's first index; the order inside a bin does not
// matter, so a pair's place inside the tile's run is one LDS atomic (no ranking, no stability).  A first attempt with ONE pass into
// all 3052 windows (8-byte scattered stores, 512 workgroups x 3052 open lines) took 1.35 ms for that pass alone at n = 1e8.
// Algorithmic bytes: (4 n + 8 n) + (8 n + 8 n) + (8 n + 4 n) = 40 n.
constexpr int ISA_BLOCK = 1024;
constexpr int ISA_WBITS = 15;                 // largest window = 32768 words = 128 KiB of LDS
constexpr size_t ISA_MAX_N = size_t(1) << (ISA_WBITS + 12);  // at most 64 sections x 64 windows
constexpr int ISP_BLOCK = 256, ISP_IPT = 16, ISP_TILE = ISP_BLOCK * ISP_IPT;  // 4096 pairs per workgroup

// counters[d] = first index of bin d (bins of 2^shift entries)
__global__ __launch_bounds__(256) void k_isa_init(uint32_t *__restrict__ counters, uint32_t bins, int shift) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d < bins) counters[d] = d << shift;
}

// FIRST: item i of the tile is (sa[base + i], base + i), its bin sa >> shift.  Otherwise the items are pairs of ONE section (a tile never
// straddles sections: a section has a multiple of 4096 entries), the bin is the window inside the section: (suffix >> shift) & 63, and
// the counters of that section start at counters[section * 64].
template <bool FIRST>
__global__ __launch_bounds__(ISP_BLOCK) void k_isa_split(const uint32_t *__restrict__ sa, const uint64_t *__restrict__ pairs_in, size_t n, int shift,
                                                          int section_shift, uint32_t *__restrict__ counters, uint64_t *__restrict__ pairs_out) {
    __shared__ uint64_t s_item[ISP_TILE];
    __shared__ uint32_t s_cnt[64], s_start[64], s_gbase[64];
    const int tid = threadIdx.x;
    const size_t base = static_cast<size_t>(blockIdx.x) * ISP_TILE;
    const uint32_t valid = static_cast<uint32_t>(n - base < static_cast<size_t>(ISP_TILE) ? n - base : ISP_TILE);
    uint32_t *cnt_base = FIRST ? counters : counters + ((base >> section_shift) << 6);
    if (tid < 64) s_cnt[tid] = 0;
    __syncthreads();
    uint64_t item[ISP_IPT];
    uint32_t bin[ISP_IPT], r[ISP_IPT];
#pragma unroll
    for (int k = 0; k < ISP_IPT; ++k) {
        const uint32_t i = k * ISP_BLOCK + tid;
        if (i < valid) {
            if (FIRST) {
                const uint32_t v = sa[base + i];
                item[k] = (static_cast<uint64_t>(v) << 32) | static_cast<uint32_t>(base + i);
                bin[k] = v >> shift;
            } else {
                item[k] = pairs_in[base + i];
                bin[k] = (static_cast<uint32_t>(item[k] >> 32) >> shift) & 63u;
            }
            r[k] = atomicAdd(&s_cnt[bin[k] & 63u], 1u);
        }
    }
    __syncthreads();
    if (tid < 64) {  // one wave: exclusive scan of the 64 counts, and the tile's place in every bin (one global atomic per bin in use).
        // (Asking for the place here and using it only after the staging below, to hide the atomic's round trip, was slower: 1.08
        // against 0.95 ms for the two passes at n = 1e8.)
        const uint32_t c = s_cnt[tid];
        const uint32_t incl = wave_incl_sum(c, tid);
        s_start[tid] = incl - c;
        const uint32_t g = c ? atomicAdd(&cnt_base[tid], c) : 0u;
        s_gbase[tid] = g - (incl - c);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ISP_IPT; ++k) {
        const uint32_t i = k * ISP_BLOCK + tid;
        if (i < valid) s_item[s_start[bin[k] & 63u] + r[k]] = item[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ISP_IPT; ++k) {
        const uint32_t q = k * ISP_BLOCK + tid;
        if (q < valid) {
            const uint64_t it = s_item[q];
            const uint32_t b = FIRST ? (static_cast<uint32_t>(it >> 32) >> shift) : ((static_cast<uint32_t>(it >> 32) >> shift) & 63u);
            const size_t at = static_cast<size_t>(s_gbase[b & 63u]) + q;
            if (at < n) pairs_out[at] = it;  // (always true for a permutation: a guard against a corrupt input, never a wild store)
        }
    }
}
__global__ __launch_bounds__(ISA_BLOCK) void k_isa_assemble(const uint64_t *__restrict__ pairs, size_t n, int wbits, uint32_t *__restrict__ rank) {
    extern __shared__ __attribute__((aligned(16))) uint32_t win[];
    const size_t base = static_cast<size_t>(blockIdx.x) << wbits;
    const uint32_t w = 1u << wbits;
    const uint32_t count = static_cast<uint32_t>(n - base < w ? n - base : w);
    const uint32_t even = count & ~1u;
    for (uint32_t i = 2 * threadIdx.x; i < even; i += 2 * ISA_BLOCK) {  // base is a multiple of 1024 pairs: 16-byte aligned loads
        const uint4 v = *reinterpret_cast<const uint4 *>(pairs + base + i);
        win[v.y & (w - 1)] = v.x;  // (suffix << 32 | p): y = suffix, x = p
        win[v.w & (w - 1)] = v.z;
    }
    if (threadIdx.x == 0 && even < count) {
        const uint64_t v = pairs[base + even];
        win[static_cast<uint32_t>(v >> 32) & (w - 1)] = static_cast<uint32_t>(v);
    }
    __syncthreads();
    const uint32_t quads = count & ~3u;
    for (uint32_t i = 4 * threadIdx.x; i < quads; i += 4 * ISA_BLOCK)
        *reinterpret_cast<uint4 *>(rank + base + i) = *reinterpret_cast<const uint4 *>(win + i);
    if (threadIdx.x < count - quads) rank[base + quads + threadIdx.x] = win[quads + threadIdx.x];
}

// rank[sa[p]] = p; sa must be a permutation of 0..n-1, n <= ISA_MAX_N; scratch_a / scratch_b hold n u64 each
static int inverse_permutation_windows(dk_ctx *ctx, const uint32_t *sa, size_t n, uint64_t *scratch_a, uint64_t *scratch_b, uint32_t *rank) {
    const int wbits = n > (size_t(1) << (10 + 12)) ? static_cast<int>(ceil_log2_u64(n)) - 12 : 10;  // at most 4096 windows of at least 1024 words
    const uint32_t nwin = static_cast<uint32_t>(div_up(n, size_t(1) << wbits));
    const bool two_levels = nwin > 64;
    const uint32_t nsec = two_levels ? static_cast<uint32_t>(div_up(nwin, 64)) : 0;
    const size_t ntiles = div_up(n, ISP_TILE);
    const size_t mark = ctx->ws_mark();
    uint32_t *counters = ctx->ws_alloc<uint32_t>(64 + static_cast<size_t>(nsec ? nsec : 1) * 64);
    if (!counters) return DK_E_NOMEM;
    uint32_t *sec_counters = counters, *win_counters = counters + 64;
    hipStream_t st = ctx->stream;
    uint64_t *by_window = scratch_a;
    {
        LaunchScope ls(ctx, K_ISA_PARTITION, (two_levels ? 28.0 : 12.0) * n);
        if (two_levels) {
            k_isa_init<<<dim3(1), dim3(256), 0, st>>>(sec_counters, nsec, wbits + 6);
            k_isa_init<<<dim3(div_up(nsec * 64, 256)), dim3(256), 0, st>>>(win_counters, nsec * 64, wbits);
            k_isa_split<true><<<dim3(ntiles), dim3(ISP_BLOCK), 0, st>>>(sa, nullptr, n, wbits + 6, 0, sec_counters, scratch_a);
            k_isa_split<false><<<dim3(ntiles), dim3(ISP_BLOCK), 0, st>>>(nullptr, scratch_a, n, wbits, wbits + 6, win_counters, scratch_b);
            by_window = scratch_b;
        } else {
            k_isa_init<<<dim3(1), dim3(256), 0, st>>>(sec_counters, nwin, wbits);
            k_isa_split<true><<<dim3(ntiles), dim3(ISP_BLOCK), 0, st>>>(sa, nullptr, n, wbits, 0, sec_counters, scratch_a);
        }
    }
    {
        LaunchScope ls(ctx, K_ISA_ASSEMBLE, 12.0 * n);
        static const bool lds_ok = [] {  // a workgroup may declare more than the default 64 KiB of dynamic LDS only after this
            return hipFuncSetAttribute(reinterpret_cast<const void *>(k_isa_assemble), hipFuncAttributeMaxDynamicSharedMemorySize, 4 << ISA_WBITS) == hipSuccess;
        }();
        if (!lds_ok) return ctx->fail(DK_E_HIP, "k_isa_assemble: cannot reserve %d bytes of LDS", 4 << ISA_WBITS);
        k_isa_assemble<<<dim3(nwin), dim3(ISA_BLOCK), sizeof(uint32_t) << wbits, st>>>(by_window, n, wbits, rank);
    }
    DK_HIP(ctx, hipGetLastError());
    ctx->ws_release(mark);
    return DK_OK;
}

// dst[idx[i]] = val[i] (val == nullptr: = i), i < count; idx values are distinct and < limit.  `scratch` holds count u64.
// val == nullptr with count == limit is the inverse of a permutation: up to 2^27 entries it goes through LDS windows.
int scatter_u32_bucketed(dk_ctx *ctx, const uint32_t *idx, const uint32_t *val, size_t count, size_t limit, uint64_t *scratch,
                         uint64_t *scratch_b, uint32_t *dst) {
    if (count == 0) return DK_OK;
    if (!val && count == limit && count <= ISA_MAX_N && scratch_b) return inverse_permutation_windows(ctx, idx, count, scratch, scratch_b, dst);
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
        k_radix_hist<true><<<dim3(ntiles), dim3(RS_BLOCK), 0, st>>>(nullptr, idx, val, count, shift, tile_hist, TextKeys{});
    }
    {
        LaunchScope ls(ctx, K_RADIX_SCAN, 3.0 * 1024.0 * ntiles);
        k_radix_scan_a<<<dim3(nchunks), dim3(256), 0, st>>>(tile_hist, ntiles, tiles_per_chunk, chunk_sum);
        k_radix_scan_b<<<dim3(1), dim3(1024), 0, st>>>(chunk_sum, nchunks);
        k_radix_scan_c<<<dim3(nchunks), dim3(256), 0, st>>>(tile_hist, ntiles, tiles_per_chunk, chunk_sum);
    }
    {
        LaunchScope ls(ctx, K_RADIX_SCATTER, 16.0 * count);
        k_radix_scatter<true><<<dim3(grid), dim3(RS_BLOCK), 0, st>>>(nullptr, idx, val, scratch, nullptr, count, shift, tile_hist,
                                                                       static_cast<uint32_t>(ntiles), TextKeys{}, nullptr, SortFinalOut{});
    }
    {
        LaunchScope ls(ctx, K_BUCKET_STORE, 12.0 * count);
        // 72 KiB of (unused) dynamic LDS caps residency at 2 workgroups per CU: ~260 K pairs in flight per XCD, less than one
        // bucket, so the lines of the bucket being written stay in that XCD's 4 MiB L2 until they are complete
        const size_t lds_cap = static_cast<size_t>(DK_KNOB("DK_BUCKET_LDS", 72 * 1024));
        k_bucket_store<<<dim3(grid), dim3(RS_BLOCK), lds_cap, st>>>(scratch, count, static_cast<uint32_t>(ntiles), dst);
    }
    DK_HIP(ctx, hipGetLastError());
    ctx->ws_release(mark);
    return DK_OK;
}

}  // namespace dk
