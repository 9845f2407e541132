// Stable LSD radix sort of (u64 key, u32 value) pairs, 8 bits per pass -- the workhorse of the suffix sort.
// It takes the place of all the sorting work inside saca() (src/saca.rs:270-340): bucket placement, the induced
// sorts and the naming pass become "sort by packed prefix, then by the symbols / ranks further on".
//
// Per pass, three launches:
//   k_radix_hist        one workgroup per CHUNK of consecutive tiles keeps a cumulative 256-bin histogram in LDS (16 copies per digit)
//                       and leaves, per tile, the count of every digit in the chunk's earlier tiles; from the second pass on the
//                       digits come from the one-byte DIGIT PLANE the previous pass's scatter wrote beside the pairs (n bytes read
//                       instead of 8 n), in the first pass of the suffix sort straight from the text
//   k_radix_scan        chunk totals -> pairs with a smaller digit + pairs with the same digit in earlier chunks (one wave per digit)
//   k_radix_scatter     a tile ranks its pairs stably with wave64 ballots (match-any over the 8 digit bits), reorders them in LDS so
//                       that equal digits are contiguous, then writes runs to HBM.  XCD-aware tile order: every XCD takes one
//                       contiguous range of tiles, so the short output runs of neighbouring tiles meet in the same L2.
// Algorithmic bytes per pass: scatter 12 B/pair read + 12 (+1: digit plane) B/pair written; histogram 8 B/key, 1 B with the plane,
// + 1 KiB per tile (the table is written once by the histogram and read once by the scatter).
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
#ifndef DK_SCATTER_BLOCK_DEFAULT
#define DK_SCATTER_BLOCK_DEFAULT 256
#endif
constexpr int RS_KPT = 16;                   // pairs per thread
constexpr int RS_TILE = RS_BLOCK * RS_KPT;   // 4096 pairs per workgroup

constexpr int RS_TEXT_AHEAD = 64 + 16;       // TextKeys: codes staged beyond the tile (spk <= 64)

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
// Staging of a tile: the codes of T[b0 .. b0 + RS_TILE + RS_TEXT_AHEAD) into s_c (zero past the end of the text), the code of the symbol in
// front of the tile into s_c[RS_TILE + RS_TEXT_AHEAD].  In two halves, so that a kernel that walks several tiles can ask for the bytes of
// the next one before it works on the current one: text_fetch (global memory -> registers; only when the whole range lies inside an
// aligned text) and text_stage (registers -> code table -> s_c; ends with a barrier; s_code must be loaded and visible).
constexpr int RS_TEXT_TAIL = (RS_TEXT_AHEAD + 15) / 16;  // threads that fetch a second slice (the codes beyond the tile)
static_assert(RS_TEXT_TAIL < 64 && RS_BLOCK > 64, "thread 64 fetches the byte in front of the tile");
struct TextRaw { uint4 a, b; uint32_t before; };
__device__ __forceinline__ bool text_fast(const TextKeys &tk, size_t b0) {
    return (reinterpret_cast<uintptr_t>(tk.t) & 15) == 0 && b0 + RS_TILE + RS_TEXT_TAIL * 16 <= tk.n;
}
__device__ __forceinline__ void text_fetch(const TextKeys &tk, size_t b0, TextRaw &r) {
    const int tid = threadIdx.x;
    if (!text_fast(tk, b0)) return;
    r.a = *reinterpret_cast<const uint4 *>(tk.t + b0 + tid * 16);
    if (tid < RS_TEXT_TAIL) r.b = *reinterpret_cast<const uint4 *>(tk.t + b0 + RS_TILE + tid * 16);
    if (tid == 64) r.before = tk.t[b0 ? b0 - 1 : tk.n - 1];
}
__device__ __forceinline__ uint4 text_codes16(uint4 raw, const uint8_t *s_code) {
    uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        w[k] = s_code[w[k] & 0xFFu] | (static_cast<uint32_t>(s_code[(w[k] >> 8) & 0xFFu]) << 8) | (static_cast<uint32_t>(s_code[(w[k] >> 16) & 0xFFu]) << 16) |
               (static_cast<uint32_t>(s_code[w[k] >> 24]) << 24);
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ void text_stage(const TextKeys &tk, size_t b0, const TextRaw &r, const uint8_t *s_code, uint8_t *s_c) {
    const int tid = threadIdx.x;
    if (text_fast(tk, b0)) {
        *reinterpret_cast<uint4 *>(s_c + tid * 16) = text_codes16(r.a, s_code);
        if (tid < RS_TEXT_TAIL) *reinterpret_cast<uint4 *>(s_c + RS_TILE + tid * 16) = text_codes16(r.b, s_code);
        if (tid == 64) s_c[RS_TILE + RS_TEXT_AHEAD] = s_code[r.before & 0xFFu];
    } else {
        if (tid == 0) s_c[RS_TILE + RS_TEXT_AHEAD] = s_code[tk.t[b0 ? b0 - 1 : tk.n - 1]];  // code of the symbol in front of the tile
        for (int o = tid * 16; o < RS_TILE + RS_TEXT_AHEAD; o += RS_BLOCK * 16) {
            const size_t p = b0 + o;
            uint8_t raw[16];
#pragma unroll
            for (int b = 0; b < 16; ++b) raw[b] = (p + b < tk.n) ? s_code[tk.t[p + b]] : 0;  // zero padding past the end of the text
            *reinterpret_cast<uint4 *>(s_c + o) = *reinterpret_cast<const uint4 *>(raw);
        }
    }
    __syncthreads();
}
// both halves and the code table, for a kernel that sees one tile
__device__ __forceinline__ void text_stage_codes(const TextKeys &tk, size_t b0, uint8_t *s_code, uint8_t *s_c) {
    TextRaw r;
    text_fetch(tk, b0, r);
    for (int i = threadIdx.x; i < 256; i += RS_BLOCK) s_code[i] = tk.code[i];
    __syncthreads();
    text_stage(tk, b0, r, s_code, s_c);
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

// ---- histograms: one workgroup per CHUNK of consecutive tiles ------------------------------------------------------------------------
// The scatter needs, for every (tile, digit), the global index of the tile's first pair with that digit
//      = (pairs with smaller digits) + (pairs with this digit in earlier chunks) + (pairs with this digit in earlier tiles of the chunk).
// A workgroup walks the tiles of its chunk in order and keeps ONE cumulative histogram in LDS (16 copies per digit, never cleared):
// after the tile's 4096 atomics the sum over the copies is the count up to and including this tile -- what the NEXT tile needs as its
// third term (tile_pre), written straight away; the last sum is the chunk's total (chunk_sum, and added to digit_total[digit]).  One
// small kernel (k_radix_scan) turns chunk_sum into the first two terms.  Round 2's form (a histogram workgroup per tile that cleared and
// reduced its 16 copies -- as many LDS accesses as the counting itself -- and three scan kernels that read the 1 KiB-per-tile table once
// and rewrote it once) cost 0.087 + 0.037 ms per pass of 1e8 pairs; this one reads the digits once and writes the table once.
// Layout h[digit][copy], copy = lane mod 16: the bank of a counter is 16 (digit mod 2) + copy, so the 32 lanes of an LDS cycle meet at
// most two to a bank whatever the digits are (h[copy][digit] put digit mod 32 in charge: three to four deep on uniform digits); skewed
// digits still spread over the 16 copies.
constexpr int RS_HCOPIES = 16;
enum HistSource : int { HS_KEYS = 0, HS_PAIRS = 1, HS_PLANE = 2, HS_TEXT = 3 };

// sum of the 16 copies of this thread's digit: four 16-byte reads, rotated so that the 16 lanes of an LDS cycle cover all 64 banks
__device__ __forceinline__ uint32_t hist_copies_sum(const uint32_t *h, int d) {
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint4 v = *reinterpret_cast<const uint4 *>(h + d * RS_HCOPIES + 4 * ((j + (d >> 2)) & 3));
        sum += v.x + v.y + v.z + v.w;
    }
    return sum;
}

template <int SRC>
__global__ __launch_bounds__(RS_BLOCK) void k_radix_hist(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ hi,
                                                          const uint32_t *__restrict__ lo, const uint8_t *__restrict__ plane, size_t n, int shift,
                                                          uint32_t ntiles, uint32_t tiles_per_chunk, uint32_t *__restrict__ tile_pre,
                                                          uint32_t *__restrict__ chunk_sum, uint32_t *__restrict__ digit_total, TextKeys tk) {
    static_assert(RS_BLOCK == 256, "one thread per digit");
    __shared__ __attribute__((aligned(16))) uint32_t h[256 * RS_HCOPIES];
    __shared__ __attribute__((aligned(16))) uint8_t s_c[SRC == HS_TEXT ? RS_TILE + RS_TEXT_AHEAD + 16 : 16];
    __shared__ uint8_t s_code[SRC == HS_TEXT ? 256 : 4];
    const int tid = threadIdx.x;
    for (int i = tid; i < 256 * RS_HCOPIES; i += RS_BLOCK) h[i] = 0;
    if (SRC == HS_TEXT) s_code[tid] = tk.code[tid];
    uint32_t *mine = h + (tid & (RS_HCOPIES - 1));  // counter of digit d: mine[d * RS_HCOPIES]
    const uint32_t t0 = blockIdx.x * tiles_per_chunk;
    const uint32_t t1 = t0 + tiles_per_chunk < ntiles ? t0 + tiles_per_chunk : ntiles;
    const bool plane_aligned = SRC == HS_PLANE && (reinterpret_cast<uintptr_t>(plane) & 15) == 0;
    // PLANE / KEYS: the next tile's digits are asked for before the current tile is counted
    uint4 ahead[SRC == HS_KEYS ? RS_KPT / 2 : 1];
    auto fetch = [&](uint32_t t) {
        const size_t base = static_cast<size_t>(t) * RS_TILE;
        if (SRC == HS_PLANE) {
            if (plane_aligned && base + RS_TILE <= n) ahead[0] = reinterpret_cast<const uint4 *>(plane + base)[tid];
        } else if (SRC == HS_KEYS) {
            if (base + RS_TILE <= n) {
                const uint4 *p = reinterpret_cast<const uint4 *>(keys + base);
#pragma unroll
                for (int k = 0; k < RS_KPT / 2; ++k) ahead[k] = p[k * RS_BLOCK + tid];
            }
        }
    };
    TextRaw text_ahead;  // TEXT: the same for the raw bytes of the text
    if (t0 < t1) {
        if (SRC == HS_TEXT) text_fetch(tk, static_cast<size_t>(t0) * RS_TILE, text_ahead); else fetch(t0);
    }
    __syncthreads();
    uint32_t before = 0;  // pairs with this thread's digit in the chunk's tiles so far
    for (uint32_t t = t0; t < t1; ++t) {
        const size_t base = static_cast<size_t>(t) * RS_TILE;
        if (SRC == HS_TEXT) {
            // The first pass sorts by the lowest eight sorted bits of the key = the low eight bits of the packed window of spk codes: only the
            // last ceil(8 / bits) symbols of the window reach them, so the digit slides along the staged codes in three instructions per
            // position -- no key is built (the scatter builds them).
            const TextRaw raw = text_ahead;
            if (t + 1 < t1) text_fetch(tk, base + RS_TILE, text_ahead);
            text_stage(tk, base, raw, s_code, s_c);
            const int p0 = tid * RS_KPT, bits = tk.bits, spk = tk.spk;
            const int last = (8 + bits - 1) / bits < spk ? (8 + bits - 1) / bits : spk;  // symbols that reach the digit (<= 8)
            const uint32_t dmask = spk * bits >= 8 ? 0xFFu : (1u << (spk * bits)) - 1u;  // (a window shorter than the digit)
            // the 24 codes from position p0 + spk - last on, in three registers (aligned 8-byte LDS reads, a byte at a time would put the
            // 32 lanes of an LDS cycle four deep on eight banks): `last` codes prime the window, one enters per position
            const int off = p0 + spk - last;
            const uint64_t *q = reinterpret_cast<const uint64_t *>(s_c + (off & ~7));
            const unsigned sh = static_cast<unsigned>(off & 7) * 8u;
            const uint64_t a = q[0], b = q[1], c = q[2], e = q[3];
            const uint64_t c0 = sh ? (a >> sh) | (b << (64u - sh)) : a, c1 = sh ? (b >> sh) | (c << (64u - sh)) : b, c2 = sh ? (c >> sh) | (e << (64u - sh)) : c;
            auto code_at = [&](int j) -> uint32_t { return static_cast<uint32_t>((j < 8 ? c0 : j < 16 ? c1 : c2) >> (8 * (j & 7))) & 0xFFu; };
            uint32_t w = 0;
            for (int j = 0; j < last; ++j) w = (w << bits) | code_at(j);
#pragma unroll
            for (int g = 0; g < RS_KPT; ++g) {
                if (base + p0 + g < n) atomicAdd(&mine[(w & dmask) * RS_HCOPIES], 1u);
                w = (w << bits) | code_at(last + g);
            }
        } else if (SRC == HS_PLANE && plane_aligned && base + RS_TILE <= n) {
            const uint4 v = ahead[0];  // (two tiles ahead was measured: no faster -- the kernel runs at the rate of its LDS atomics)
            if (t + 1 < t1) fetch(t + 1);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int b = 0; b < 4; ++b) atomicAdd(&mine[((w[k] >> (8 * b)) & 0xFFu) * RS_HCOPIES], 1u);
            }
        } else if (SRC == HS_KEYS && base + RS_TILE <= n) {  // full tile: two keys per 16-byte load (order inside the tile is irrelevant here)
            uint4 v[RS_KPT / 2];
#pragma unroll
            for (int k = 0; k < RS_KPT / 2; ++k) v[k] = ahead[k];
            if (t + 1 < t1) fetch(t + 1);
#pragma unroll
            for (int k = 0; k < RS_KPT / 2; ++k) {
                const uint64_t k0 = (static_cast<uint64_t>(v[k].y) << 32) | v[k].x, k1 = (static_cast<uint64_t>(v[k].w) << 32) | v[k].z;
                atomicAdd(&mine[digit_of(k0, shift) * RS_HCOPIES], 1u);
                atomicAdd(&mine[digit_of(k1, shift) * RS_HCOPIES], 1u);
            }
        } else {  // pairs, a last partial tile, an unaligned plane
#pragma unroll
            for (int k = 0; k < RS_KPT; ++k) {
                const size_t i = base + static_cast<size_t>(k) * RS_BLOCK + tid;
                if (i < n) {
                    const uint32_t d = SRC == HS_PLANE ? plane[i] : digit_of(load_key<SRC == HS_PAIRS>(keys, hi, lo, i), shift);
                    atomicAdd(&mine[d * RS_HCOPIES], 1u);
                }
            }
        }
        __syncthreads();
        const uint32_t upto = hist_copies_sum(h, tid);
        tile_pre[static_cast<size_t>(t) * 256 + tid] = before;
        before = upto;
        __syncthreads();  // (the next tile's atomics must not reach a counter before its sum has been read)
    }
    if (t0 < t1) {
        chunk_sum[static_cast<size_t>(blockIdx.x) * 256 + tid] = before;
        if (before) atomicAdd(&digit_total[tid], before);
    }
}

// chunk_sum[g][d]: pairs with digit d in chunk g  ->  pairs with a smaller digit + pairs with digit d in earlier chunks.  One wave per
// digit (256 workgroups of one wave: a lane's loads go to rows 1 KiB apart, 64 lines per instruction -- spread over all CUs they cost
// nothing, on 16 CUs they were the kernel's whole time), lane = a stretch of at most RS_MAX_CHUNKS / 64 chunks (registers), one DPP scan
// across the lanes.  digit_total[d] = sum over the chunks (the histogram workgroups added it up).
#ifndef DK_RS_MAX_CHUNKS
#define DK_RS_MAX_CHUNKS 1024
#endif
constexpr int RS_MAX_CHUNKS = DK_RS_MAX_CHUNKS;
__global__ __launch_bounds__(64) void k_radix_scan(uint32_t *__restrict__ chunk_sum, uint32_t nchunks, const uint32_t *__restrict__ digit_total) {
    const int lane = threadIdx.x, d = blockIdx.x;
    uint32_t smaller = 0;
    {
        const uint4 t = reinterpret_cast<const uint4 *>(digit_total)[lane];  // digits 4 lane .. 4 lane + 3
        smaller = (lane * 4 < d ? t.x : 0u) + (lane * 4 + 1 < d ? t.y : 0u) + (lane * 4 + 2 < d ? t.z : 0u) + (lane * 4 + 3 < d ? t.w : 0u);
    }
    const uint32_t per = (nchunks + 63u) / 64u;
    const uint32_t g0 = static_cast<uint32_t>(lane) * per;
    uint32_t v[RS_MAX_CHUNKS / 64], sum = 0;
#pragma unroll
    for (int k = 0; k < RS_MAX_CHUNKS / 64; ++k) {
        const uint32_t g = g0 + k;
        v[k] = (static_cast<uint32_t>(k) < per && g < nchunks) ? chunk_sum[static_cast<size_t>(g) * 256 + d] : 0u;
        sum += v[k];
    }
    smaller = wave_sum(smaller);
    uint32_t run = smaller + wave_incl_sum(sum, lane) - sum;
#pragma unroll
    for (int k = 0; k < RS_MAX_CHUNKS / 64; ++k) {
        const uint32_t g = g0 + k;
        if (static_cast<uint32_t>(k) < per && g < nchunks) chunk_sum[static_cast<size_t>(g) * 256 + d] = run;
        run += v[k];
    }
}

// The offsets of a pass: tile_pre / chunk_base as above, chunk = tile / tiles_per_chunk.
// probe (tuning build, DK_SCATTER_PROBE; 0 in the product): 1 = the ranking loop does its per-wave counter read and leader write a SECOND time on a
// second table -- same addresses, same bank conflicts, result unused: what the counters' LDS traffic costs the kernel, measured instead of argued
// ("per-wave digit counters in registers against the 60 % LDS bank-conflict cycles"); profiles/r05_scatter_counter_probe.log
struct TileOffsets { const uint32_t *tile_pre; const uint32_t *chunk_base; uint32_t tiles_per_chunk; int probe = 0; };

// offs: where the tile's pairs of every digit go (TileOffsets).  next_digit (may be null): the digit plane for the next pass (k_radix_hist_plane).  xcd_tiles != 0: XCD-aware tile order over a grid of 8 * ceil(ntiles / 8) blocks.
// Three workgroups per CU, not the four that registers and LDS would allow: with a fourth tile in flight per CU the L2 no longer
// merges the short output runs of neighbouring tiles before it has to evict them (measured, 1e8 text: 3.95 ms of scatter per sort
// against 3.6 ms; two per CU: 3.65 ms but slower overall).  The kernel is bound by that, not by its ballots: a third fewer vector
// instructions in the ranking (wave_match) did not move it.
// BLOCK threads share the tile of RS_TILE pairs: 256 threads with 16 pairs each (three waves per SIMD), or 512 with 8 (six: DK_SCATTER_BLOCK)
// PACKED (the initial sort of at most 32 key bits where the carried code and the position fit one word: small alphabets -- 2^28 {A,C,G,T}):
// a pair is ONE 64-bit word, (the 32 key bits) << 32 | (code of the symbol in front) << idx_bits | position; the text pass packs it, the
// passes behind it move 8 bytes per pair instead of 12 (keys only, digits at bits 32 and up), the last one unpacks into what SortFinalOut asks for.
struct PackedPairs { int idx_bits = 0; int key_shift = 0; };  // key_shift: where the sorted bits of the 64-bit key begin (8: a code rides in the low byte)
template <bool PAIRS = false, bool TEXT = false, int BLOCK = RS_BLOCK, bool PACKED = false>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(3 * BLOCK / 256, 3 * BLOCK / 256))) void k_radix_scatter(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                             const uint32_t *__restrict__ pair_lo,
                                                             uint64_t *__restrict__ kout, uint32_t *__restrict__ vout, size_t n,
                                                             int shift, TileOffsets offs, uint32_t xcd_tiles,
                                                             TextKeys tk, uint8_t *__restrict__ next_digit, SortFinalOut fin, PackedPairs pk = PackedPairs{}) {
    static_assert(!PACKED || !PAIRS, "packed pairs are single words already");
    constexpr int WAVES = BLOCK / 64, KPT = RS_TILE / BLOCK;
    static_assert(!TEXT || BLOCK == RS_BLOCK, "the text pass builds its keys with RS_BLOCK threads");
    __shared__ uint64_t s_keys[RS_TILE];          // tile of keys in digit order; reused for the values
    // per-wave digit counters (then exclusive over waves) | tile-local start of each digit | global offset of the digit minus its
    // tile-local start.  TEXT: the same 6 KiB first hold the staged codes of the tile (the counters are cleared afterwards).
    __shared__ __attribute__((aligned(16))) uint32_t s_tab[WAVES * 256 + 512];
    __shared__ uint32_t s_tmp[WAVES + 1];
#ifdef DK_TUNING
    __shared__ uint32_t s_probe[WAVES][256];
#endif
    __shared__ uint8_t s_code[TEXT ? 256 : 4];
    // last pass of the suffix sort's initial sort: code -> byte for L.  From LDS: a table read from global memory made every one of a
    // thread's sixteen L stores wait for ALL its memory operations in flight, the key stores before it included (the memory counter
    // is in order): 0.65 ms for that pass of 1e8 pairs against 0.50 for the others.
    __shared__ uint8_t s_inv[256];
    if (fin.bwt && threadIdx.x < 256) s_inv[threadIdx.x] = fin.inv_code[threadIdx.x];  // (visible after the barriers below)
    static_assert(!TEXT || sizeof(uint32_t) * (WAVES * 256 + 512) >= RS_TILE + RS_TEXT_AHEAD + 16, "code staging does not fit");
    uint32_t (*s_cnt)[256] = reinterpret_cast<uint32_t (*)[256]>(s_tab);
    uint32_t *s_start = s_tab + WAVES * 256;
    uint32_t *s_gbase = s_start + 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef DK_TUNING
    if (offs.probe) for (int i = tid; i < WAVES * 256; i += BLOCK) (&s_probe[0][0])[i] = 0;  // (visible behind the barriers in front of the ranking loop)
#endif
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
    const uint32_t goff_early = tid < 256 ? offs.chunk_base[static_cast<size_t>(tile / offs.tiles_per_chunk) * 256 + tid] + offs.tile_pre[static_cast<size_t>(tile) * 256 + tid] : 0u;

    // TEXT: a thread builds the keys of 16 consecutive positions, so the lanes of one LDS store are 128 bytes apart -- all on one pair of
    // banks.  Position o is kept at o with its low four bits rotated by the thread's number: conflict-free both ways.
    auto swz = [](uint32_t o) { return (o & ~15u) | ((o + (o >> 4)) & 15u); };
    if (TEXT) {  // keys of the tile, in position order
        text_tile_keys(tk, tile_base, s_code, reinterpret_cast<uint8_t *>(s_tab), [&](int o, uint64_t key) { s_keys[swz(static_cast<uint32_t>(o))] = key; });
        __syncthreads();
    } else {
        for (int i = tid; i < WAVES * 256; i += BLOCK) s_tab[i] = 0;
        __syncthreads();
    }

    // wave w owns pairs [w*1024, (w+1)*1024) of the tile, lane-striped so that loads coalesce
    uint64_t key[KPT];
    uint32_t val[KPT];
    const uint32_t wbase = static_cast<uint32_t>(wave) * (64 * KPT);
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const uint32_t li = wbase + k * 64 + lane;
        if (li < valid) {
            key[k] = TEXT ? s_keys[swz(li)] : load_key<PAIRS>(kin, vin, pair_lo, tile_base + li);  // PAIRS: vin holds the high words
            val[k] = TEXT ? static_cast<uint32_t>(tile_base + li) : ((PAIRS || PACKED) ? 0u : vin[tile_base + li]);
            if (PACKED && TEXT)
                key[k] = (((key[k] >> pk.key_shift) & 0xFFFFFFFFull) << 32) | (static_cast<uint64_t>(pk.key_shift ? key[k] & 0xFFu : 0u) << pk.idx_bits) | val[k];
        } else {
            key[k] = ~0ull;  // padding sorts behind every real pair of the tile and is never written
            val[k] = 0;
        }
    }
    if (TEXT) {  // the codes have been consumed: their place becomes the digit counters
        __syncthreads();
        for (int i = tid; i < WAVES * 256; i += BLOCK) s_tab[i] = 0;
        __syncthreads();
    }

    // stable rank inside the wave: lanes holding the same digit find each other with 8 ballots
    uint32_t rnk[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const uint32_t d = digit_of(key[k], shift);
        const LaneSet same = wave_match<8>(d, ~0ull);
        const uint32_t before = same.before();
        const uint32_t old = s_cnt[wave][d];
        __builtin_amdgcn_wave_barrier();
        if (before == 0) s_cnt[wave][d] = old + same.count();
        __builtin_amdgcn_wave_barrier();
        rnk[k] = old + before;
#ifdef DK_TUNING
        if (offs.probe) {  // (wave-uniform) the same two LDS accesses once more, on a table of their own
            const uint32_t old2 = s_probe[wave][d];
            __builtin_amdgcn_wave_barrier();
            if (before == 0) s_probe[wave][d] = old2 + same.count();
            __builtin_amdgcn_wave_barrier();
            if (old2 == 0xFFFFFFFFu) rnk[k] += 1;  // (never: keeps the read alive)
        }
#endif
    }
    __syncthreads();

    {   // threads 0..255, one per digit: exclusive over waves, then (all threads take part in the scan) exclusive over digits
        const int d = tid & 255;
        const bool owner = tid < 256;
        uint32_t run = 0, goff = 0;
        if (owner) {
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const uint32_t c = s_cnt[w][d];
                s_cnt[w][d] = run;
                run += c;
            }
            goff = goff_early;
        }
        const uint32_t start = block_excl_sum<WAVES>(run, s_tmp, nullptr);
        if (owner) {
            s_start[d] = start;
            s_gbase[d] = goff - start;
        }
    }
    __syncthreads();

    uint32_t pos[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const uint32_t d = digit_of(key[k], shift);
        pos[k] = s_start[d] + s_cnt[wave][d] + rnk[k];
        s_keys[pos[k]] = key[k];
    }
    __syncthreads();

    uint32_t gi[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const uint32_t p = k * BLOCK + tid;
        const uint64_t kk = s_keys[p];
        gi[k] = s_gbase[digit_of(kk, shift)] + p;
        if (PACKED) {
            if (p < valid) {
                if (fin.vals) {  // the last pass: unpack
                    const uint32_t lo = static_cast<uint32_t>(kk), v = lo & ((1u << pk.idx_bits) - 1u), code = lo >> pk.idx_bits;
                    if (fin.narrow_shift >= 0) reinterpret_cast<uint32_t *>(kout)[gi[k]] = static_cast<uint32_t>(kk >> 32);
                    else kout[gi[k]] = ((kk >> 32) << pk.key_shift) | code;
                    vout[gi[k]] = v;
                    if (fin.bwt) fin.bwt[gi[k]] = s_inv[code];
                    if (fin.origin && v == 0) *fin.origin = gi[k];
                } else {
                    kout[gi[k]] = kk;
                }
                if (next_digit) next_digit[gi[k]] = static_cast<uint8_t>(digit_of(kk, shift + 8));
            }
            continue;
        }
        if (p < valid) {
            if (fin.narrow_shift >= 0) reinterpret_cast<uint32_t *>(kout)[gi[k]] = static_cast<uint32_t>(kk >> fin.narrow_shift);  // SortFinalOut
            else kout[gi[k]] = kk;
            if (next_digit) next_digit[gi[k]] = static_cast<uint8_t>(digit_of(kk, shift + 8));  // what the next pass's histogram reads
            if (fin.bwt) fin.bwt[gi[k]] = s_inv[kk & 0xFFu];  // last pass of the suffix sort's initial sort: L rides in the key's low byte
        }
    }
    if (PAIRS || PACKED) return;  // keys only
    __syncthreads();
    uint32_t *s_vals = reinterpret_cast<uint32_t *>(s_keys);
#pragma unroll
    for (int k = 0; k < KPT; ++k) s_vals[pos[k]] = val[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const uint32_t p = k * BLOCK + tid;
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

// (a grid of more than one workgroup: workgroup b sorts the pairs [b * SS_MAX, (b + 1) * SS_MAX) by themselves -- the local pass of an
// MSD-first sort, measured in round 4: dk_dbg_dev_local_sort)
__global__ __launch_bounds__(SS_BLOCK) void k_radix_sort_small(uint64_t *__restrict__ keys, uint32_t *__restrict__ vals, uint32_t count,
                                                               int begin_bit, int end_bit) {
    {
        const size_t base = static_cast<size_t>(blockIdx.x) * SS_MAX;
        keys += base;
        vals += base;
        count = count - base < static_cast<size_t>(SS_MAX) ? static_cast<uint32_t>(count - base) : static_cast<uint32_t>(SS_MAX);
    }
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

// Many independent small sorts in one launch: workgroup g sorts the pairs [starts[g], starts[g + 1]) by themselves (k_radix_sort_small's loop) when
// their number lies in (lo, hi] -- one launch per size class, a workgroup whose range belongs to another class leaves at once.  hi <= BLOCK * 8.
// The medium groups of the L-first path's big list (lfirst.inc): 257 .. 8192 members each, tens of thousands of them per round.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_sort_groups(const uint64_t *kin, const uint32_t *vin, uint64_t *kout, uint32_t *vout,  // (kout may be kin: no __restrict__)
                                                       const uint32_t *__restrict__ starts, uint32_t lo, uint32_t hi, int begin_bit, int end_bit) {
    constexpr int WAVES = BLOCK / 64, KPT = 8, MAXN = BLOCK * KPT;
    const uint32_t start = starts[blockIdx.x], count = starts[blockIdx.x + 1] - start;
    if (count <= lo || count > hi) return;
    __shared__ uint64_t s_keys[MAXN];
    __shared__ uint32_t s_vals[MAXN];
    __shared__ uint32_t s_cnt[WAVES][256];
    __shared__ uint32_t s_start[256];
    __shared__ uint32_t s_tmp[WAVES + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // pairs per thread this group needs (1 .. KPT): every loop below stops there, so a group costs what its size asks for, not what the class allows
    const int kmax = static_cast<int>((count + BLOCK - 1) / BLOCK);
    const uint32_t wbase = static_cast<uint32_t>(wave) * (64u * static_cast<uint32_t>(kmax));
    uint64_t key[KPT];
    uint32_t val[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        if (k >= kmax) break;
        const uint32_t li = wbase + k * 64 + lane;
        key[k] = li < count ? kin[start + li] : ~0ull;
        val[k] = li < count ? vin[start + li] : 0u;
    }
    for (int shift = begin_bit; shift < end_bit; shift += 8) {
        for (int i = tid; i < WAVES * 256; i += BLOCK) (&s_cnt[0][0])[i] = 0;
        __syncthreads();
        uint32_t rnk[KPT];
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            if (k >= kmax) break;
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
                for (int w = 0; w < WAVES; ++w) {
                    const uint32_t c = s_cnt[w][d];
                    s_cnt[w][d] = run;
                    run += c;
                }
            }
            const uint32_t first = block_excl_sum<WAVES>(tid < 256 ? run : 0u, s_tmp, nullptr);
            if (tid < 256) s_start[d] = first;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            if (k >= kmax) break;
            const uint32_t d = digit_of(key[k], shift);
            const uint32_t p = s_start[d] + s_cnt[wave][d] + rnk[k];
            s_keys[p] = key[k];
            s_vals[p] = val[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            if (k >= kmax) break;
            const uint32_t li = wbase + k * 64 + lane;
            key[k] = s_keys[li];
            val[k] = s_vals[li];
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        if (k >= kmax) break;
        const uint32_t li = wbase + k * 64 + lane;
        if (li < count) { kout[start + li] = key[k]; vout[start + li] = val[k]; }
    }
}

}  // namespace

// ngroups independent sorts on key bits [begin_bit, end_bit): group g = the pairs [starts[g], starts[g + 1]) of kin / vin, more than `above` and at
// most 8192 of them (other sizes are left alone: nothing is written for them); the sorted group goes to the same places of kout / vout (which may be
// kin / vin themselves).
int sort_groups(dk_ctx *ctx, const uint64_t *kin, const uint32_t *vin, uint64_t *kout, uint32_t *vout, const uint32_t *starts, size_t ngroups, size_t npairs,
                uint32_t above, int begin_bit, int end_bit) {
    if (ngroups == 0) return DK_OK;
    hipStream_t st = ctx->stream;
    LaunchScope ls(ctx, K_RADIX_SORT_SMALL, 24.0 * npairs);
    k_sort_groups<256><<<dim3(ngroups), dim3(256), 0, st>>>(kin, vin, kout, vout, starts, above, 2048u, begin_bit, end_bit);
    k_sort_groups<1024><<<dim3(ngroups), dim3(1024), 0, st>>>(kin, vin, kout, vout, starts, std::max(above, 2048u), 8192u, begin_bit, end_bit);
    DK_HIP(ctx, hipGetLastError());
    return DK_OK;
}

// Sorts `count` pairs on key bits [begin_bit, end_bit).  keys/vals are the input buffers, *_alt equally sized scratch;
// on return `keys` and `vals` refer to whichever buffer holds the sorted data (the references are swapped per pass).
// Wider digits were built and measured in round 3 (passes of 10 / 9 bits over super-tiles of 16384 pairs with 16-bit LDS counters:
// 56 bits in six passes, 40 bits in four): every pass got 1.6 x slower -- 1024 digits cut a tile's output into runs of four pairs
// (32 bytes of keys, 16 of values), more partial lines than the L2 merges -- 2^28 ACGT 17.5 against 13.4 ms, 1e8 text 15.2 against
// 14.0 ms (profiles/r03_wide_digit_experiment.log).  Removed; eight bits per pass stay.
// chunking of a sort of ntiles tiles: at most RS_MAX_CHUNKS histogram workgroups, every one walking tiles_per_chunk consecutive tiles
struct ChunkPlan { uint32_t ntiles, tiles_per_chunk, nchunks; };
static ChunkPlan plan_chunks(size_t count) {
    ChunkPlan p;
    p.ntiles = static_cast<uint32_t>(div_up(count, RS_TILE));
    p.tiles_per_chunk = static_cast<uint32_t>(div_up(p.ntiles, RS_MAX_CHUNKS));
    p.nchunks = static_cast<uint32_t>(div_up(p.ntiles, p.tiles_per_chunk));
    return p;
}

static int sort_pairs_classic(dk_ctx *ctx, uint64_t *&keys, uint64_t *&keys_alt, uint32_t *&vals, uint32_t *&vals_alt, size_t count,
                              int begin_bit, int end_bit, const TextKeys *text, const SortFinalOut *final_out) {
    const ChunkPlan cp = plan_chunks(count);
    const size_t ntiles = cp.ntiles;
    const int npasses = (end_bit - begin_bit + 7) / 8;
    const size_t mark = ctx->ws_mark();
    uint32_t *tile_pre = ctx->ws_alloc<uint32_t>(ntiles * 256);
    uint32_t *chunk_sum = ctx->ws_alloc<uint32_t>(static_cast<size_t>(cp.nchunks) * 256);
    uint32_t *digit_total = ctx->ws_alloc<uint32_t>(static_cast<size_t>(npasses) * 256);  // one row per pass, cleared once
    // digit plane: every scatter but the last leaves the next pass's digits behind, one byte per pair.  The histograms get faster,
    // the scatters 10 % slower (the plane's 16-byte runs cost as many L2 write requests as the 64-byte runs of the values).  Net gain,
    // round 3: 2.7 % of the whole suffix sort at 1e8 pairs (12.07 -> 11.74 ms), 1.8 % at 2^28; on from 2^26 pairs, where the keys
    // (8 bytes per pair) no longer fit the 256 MiB Infinity Cache.
    // Round 4: on from 2^20 pairs -- the big lists of the L-first path are 1 to 60 M pairs, sorted round after round: 1e8 bytes of word-like
    // text 19.2 -> 18.3 ms (histograms 2.7 -> 1.6 ms, scatters 8.6 -> 8.8), the 1e8 text block 6.4 -> 6.25.
    const int plane_mode = DK_KNOB("DK_DIGIT_PLANE", -1);  // tuning build: 1 / 0 = always (from 2^20 pairs) / never; 2 = from 2^26 (rounds 2-3)
    const size_t plane_from = plane_mode == 2 ? (size_t(1) << 26) : (size_t(1) << 20);
    uint8_t *plane = plane_mode != 0 && end_bit - begin_bit > 8 && count >= plane_from ? ctx->ws_try_alloc<uint8_t>(count) : nullptr;  // optional: the sort runs without it
    if (!tile_pre || !chunk_sum || !digit_total) return DK_E_NOMEM;
    hipStream_t st = ctx->stream;
    DK_HIP(ctx, hipMemsetAsync(digit_total, 0, static_cast<size_t>(npasses) * 256 * sizeof(uint32_t), st));
    const TileOffsets offs{tile_pre, chunk_sum, cp.tiles_per_chunk, DK_KNOB("DK_SCATTER_PROBE", 0)};
    // packed pairs (PackedPairs above): the initial sort of at most 32 key bits whose carried code and position share a 32-bit word
    const int idx_bits = static_cast<int>(ceil_log2_u64(count));
    const bool packed = text && final_out && final_out->vals && end_bit - begin_bit <= 32 && end_bit - begin_bit > 8 && (begin_bit == 0 || begin_bit == 8) &&
                        (begin_bit ? text->bits : 0) + idx_bits <= 32 && DK_KNOB("DK_PACKED_SORT", 1) != 0;
    const PackedPairs pk{idx_bits, begin_bit};
    if (packed) ctx->stats.sa_route |= DK_ROUTE_PACKED_PAIRS;
    int pass = 0;
    for (int shift = begin_bit; shift < end_bit; shift += 8, ++pass) {
        const int pshift = 32 + (shift - begin_bit);  // where the digit of this pass stands in a packed pair
        const bool have_plane = plane && shift > begin_bit;      // written by the previous pass
        uint8_t *emit = plane && shift + 8 < end_bit ? plane : nullptr;  // read by the next one
        const TextKeys *tk = text && shift == begin_bit ? text : nullptr;
        const bool last = shift + 8 >= end_bit;
        const SortFinalOut fin = last && final_out ? *final_out : SortFinalOut{};
        uint32_t *vout = fin.vals ? fin.vals : vals_alt;
        uint32_t *totals = digit_total + static_cast<size_t>(pass) * 256;
        {
            LaunchScope ls(ctx, tk ? K_RADIX_HIST_TEXT : K_RADIX_HIST, (tk ? 1.0 : have_plane ? 1.0 : 8.0) * count + 1024.0 * ntiles);
            const dim3 grid(cp.nchunks), block(RS_BLOCK);
            if (tk)
                k_radix_hist<HS_TEXT><<<grid, block, 0, st>>>(nullptr, nullptr, nullptr, nullptr, count, shift, cp.ntiles, cp.tiles_per_chunk, tile_pre, chunk_sum, totals, *tk);
            else if (have_plane)
                k_radix_hist<HS_PLANE><<<grid, block, 0, st>>>(nullptr, nullptr, nullptr, plane, count, shift, cp.ntiles, cp.tiles_per_chunk, tile_pre, chunk_sum, totals, TextKeys{});
            else
                k_radix_hist<HS_KEYS><<<grid, block, 0, st>>>(keys, nullptr, nullptr, nullptr, count, packed ? pshift : shift, cp.ntiles, cp.tiles_per_chunk, tile_pre, chunk_sum, totals, TextKeys{});
        }
        {
            LaunchScope ls(ctx, K_RADIX_SCAN, 2.0 * 1024.0 * cp.nchunks);
            k_radix_scan<<<dim3(256), dim3(64), 0, st>>>(chunk_sum, cp.nchunks, totals);
        }
        if (fin.bucket_starts)  // the first chunk's row: pairs with a smaller digit = where every digit's pairs start
            DK_HIP(ctx, hipMemcpyAsync(fin.bucket_starts, chunk_sum, 256 * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        {
            LaunchScope ls(ctx, tk ? K_RADIX_SCATTER_TEXT : K_RADIX_SCATTER,
                           ((packed ? (tk ? 9.0 : last ? 20.0 : 16.0) : tk ? 13.0 : 24.0) + (emit ? 1.0 : 0.0) + (fin.bwt ? 1.0 : 0.0) - (fin.narrow_shift >= 0 ? 4.0 : 0.0)) * count);
            const bool xcd = DK_KNOB("DK_XCD", 1) != 0;
            const size_t grid = xcd ? 8 * div_up(ntiles, 8) : ntiles;
            if (packed && tk)
                k_radix_scatter<false, true, RS_BLOCK, true><<<dim3(grid), dim3(RS_BLOCK), 0, st>>>(nullptr, nullptr, nullptr, keys_alt, vout, count, pshift, offs,
                                                                                                  xcd ? static_cast<uint32_t>(ntiles) : 0u, *tk, emit, fin, pk);
            else if (packed)
                k_radix_scatter<false, false, RS_BLOCK, true><<<dim3(grid), dim3(RS_BLOCK), 0, st>>>(keys, nullptr, nullptr, keys_alt, vout, count, pshift, offs,
                                                                                                   xcd ? static_cast<uint32_t>(ntiles) : 0u, TextKeys{}, emit, fin, pk);
            else if (tk)
                k_radix_scatter<false, true><<<dim3(grid), dim3(RS_BLOCK), 0, st>>>(nullptr, nullptr, nullptr, keys_alt, vout, count, shift, offs,
                                                                                   xcd ? static_cast<uint32_t>(ntiles) : 0u, *tk, emit, fin);
            else if (DK_KNOB("DK_SCATTER_BLOCK", DK_SCATTER_BLOCK_DEFAULT) == 512)
                k_radix_scatter<false, false, 512><<<dim3(grid), dim3(512), 0, st>>>(keys, vals, nullptr, keys_alt, vout, count, shift, offs,
                                                                                    xcd ? static_cast<uint32_t>(ntiles) : 0u, TextKeys{}, emit, fin);
            else
                k_radix_scatter<false><<<dim3(grid), dim3(RS_BLOCK), 0, st>>>(keys, vals, nullptr, keys_alt, vout, count, shift, offs,
                                                                             xcd ? static_cast<uint32_t>(ntiles) : 0u, TextKeys{}, emit, fin);
        }
        DK_HIP(ctx, hipGetLastError());
        std::swap(keys, keys_alt);
        if (!fin.vals && !packed) std::swap(vals, vals_alt);  // (the last pass wrote the values to their final home: both ping-pong buffers are free)
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
    if (final_out && final_out->narrow_shift >= 0 && (end_bit - begin_bit > 40 || !final_out->bucket_starts))
        return ctx->fail(DK_E_INTERNAL, "sort_pairs: narrow keys need a sort of at most five passes and a place for the bucket starts");
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

// experiment hook (tools/local_sort_bench.py): every SS_MAX-pair tile of a device array sorted by itself on bits [begin_bit, end_bit)
int local_sort_tiles(dk_ctx *ctx, uint64_t *d_keys, uint32_t *d_vals, size_t count, int begin_bit, int end_bit) {
    if (count == 0 || count > 0xFFFFFFFFull) return ctx->fail(DK_E_ARG, "local_sort_tiles: count");
    {
        LaunchScope ls(ctx, K_RADIX_SORT_SMALL, 24.0 * count);
        k_radix_sort_small<<<dim3(static_cast<unsigned>(div_up(count, SS_MAX))), dim3(SS_BLOCK), 0, ctx->stream>>>(d_keys, d_vals, static_cast<uint32_t>(count), begin_bit, end_bit);
    }
    DK_HIP(ctx, hipGetLastError());
    DK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DK_OK;
}

// ---- inverse permutation through LDS windows: rank[sa[p]] = p for a permutation sa of 0..n-1, n <= 2^27 ------------------------------
// (Rounds 1-3 partitioned the pairs by the top byte of the suffix and stored bucket by bucket, leaning on the L2 to complete the destination
// lines of a 1.5 MiB window before writing them back; the counters said it did not -- round 2, 1e8: WRITE_SIZE 2.27 GB for 0.40 GB of rank
// stores.  Removed in round 5: every block size goes through the windows.)  The destination is cut into windows of
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
constexpr size_t ISA_TWO_LEVELS_N = size_t(1) << (ISA_WBITS + 12);  // at most 64 sections x 64 windows
constexpr size_t ISA_MAX_N = size_t(1) << 32;                       // three levels (round 4): 64 x 64 x 64 windows of 2^15 words -- every block there is
constexpr int ISP_BLOCK = 256, ISP_IPT = 16, ISP_TILE = ISP_BLOCK * ISP_IPT;  // 4096 pairs per workgroup (8192 with 512 threads: the same 0.95 ms)

// counters[d] = first index of bin d (bins of 2^shift entries)
__global__ __launch_bounds__(256) void k_isa_init(uint32_t *__restrict__ counters, uint32_t bins, int shift) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d < bins) counters[d] = d << shift;
}

// FIRST: item i of the tile is (sa[base + i], base + i), its bin sa >> shift.  Otherwise the items are pairs of ONE section (a tile never
// straddles sections: a section has a multiple of 4096 entries), the bin is the window inside the section: (suffix >> shift) & 63, and
// the counters of that section start at counters[section * 64].
// marked_val (FIRST only, may be null): an entry of sa with bit 31 set is the suffix sa & 0x7FFFFFFF, and its value is marked_val[position]
// instead of its position (the suffix sort's active suffixes take the position of their group's head: no second, random pass over them)
template <bool FIRST>
__global__ __launch_bounds__(ISP_BLOCK) void k_isa_split(const uint32_t *__restrict__ sa, const uint64_t *__restrict__ pairs_in, size_t n, int shift,
                                                          int section_shift, uint32_t *__restrict__ counters, uint64_t *__restrict__ pairs_out,
                                                          const uint32_t *__restrict__ marked_val) {
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
                uint32_t v = sa[base + i], value = static_cast<uint32_t>(base + i);
                if (marked_val && (v & 0x80000000u)) {
                    v &= 0x7FFFFFFFu;
                    value = marked_val[base + i];
                }
                item[k] = (static_cast<uint64_t>(v) << 32) | value;
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
static int inverse_permutation_windows(dk_ctx *ctx, const uint32_t *sa, size_t n, uint64_t *scratch_a, uint64_t *scratch_b, uint32_t *rank,
                                       const uint32_t *marked_val) {
    // up to 2^27 entries: at most 4096 windows of at least 1024 words (two levels: sections, windows); above (round 4, blocks of 135 MB .. 2 GiB
    // on the suffix-array path: it was the bucketed store with its 5.7 x write amplification): windows of 2^15 words and a third level in
    // front -- at most 64 super-sections of 64 sections of 64 windows
    const bool three_levels = n > ISA_TWO_LEVELS_N;
    const int wbits = three_levels ? ISA_WBITS : n > (size_t(1) << (10 + 12)) ? static_cast<int>(ceil_log2_u64(n)) - 12 : 10;
    const uint32_t nwin = static_cast<uint32_t>(div_up(n, size_t(1) << wbits));
    const bool two_levels = nwin > 64;
    const uint32_t nsec = two_levels ? static_cast<uint32_t>(div_up(nwin, 64)) : 0;
    const uint32_t nsup = three_levels ? static_cast<uint32_t>(div_up(nsec, 64)) : 0;
    const size_t ntiles = div_up(n, ISP_TILE);
    const size_t mark = ctx->ws_mark();
    uint32_t *counters = ctx->ws_alloc<uint32_t>(64 + static_cast<size_t>(nsup ? nsup : 1) * 64 + static_cast<size_t>(nsec ? nsec : 1) * 64);
    if (!counters) return DK_E_NOMEM;
    uint32_t *top_counters = counters, *mid_counters = counters + 64, *win_counters = mid_counters + static_cast<size_t>(nsup ? nsup : 1) * 64;
    hipStream_t st = ctx->stream;
    uint64_t *by_window = scratch_a;
    {
        LaunchScope ls(ctx, K_ISA_PARTITION, (three_levels ? 44.0 : two_levels ? 28.0 : 12.0) * n);
        if (three_levels) {
            k_isa_init<<<dim3(1), dim3(256), 0, st>>>(top_counters, nsup, wbits + 12);
            k_isa_init<<<dim3(div_up(nsup * 64, 256)), dim3(256), 0, st>>>(mid_counters, nsup * 64, wbits + 6);
            k_isa_init<<<dim3(div_up(nsec * 64, 256)), dim3(256), 0, st>>>(win_counters, nsec * 64, wbits);
            k_isa_split<true><<<dim3(ntiles), dim3(ISP_BLOCK), 0, st>>>(sa, nullptr, n, wbits + 12, 0, top_counters, scratch_a, marked_val);
            k_isa_split<false><<<dim3(ntiles), dim3(ISP_BLOCK), 0, st>>>(nullptr, scratch_a, n, wbits + 6, wbits + 12, mid_counters, scratch_b, nullptr);
            k_isa_split<false><<<dim3(ntiles), dim3(ISP_BLOCK), 0, st>>>(nullptr, scratch_b, n, wbits, wbits + 6, win_counters, scratch_a, nullptr);
            by_window = scratch_a;
        } else if (two_levels) {
            k_isa_init<<<dim3(1), dim3(256), 0, st>>>(top_counters, nsec, wbits + 6);
            k_isa_init<<<dim3(div_up(nsec * 64, 256)), dim3(256), 0, st>>>(win_counters, nsec * 64, wbits);
            k_isa_split<true><<<dim3(ntiles), dim3(ISP_BLOCK), 0, st>>>(sa, nullptr, n, wbits + 6, 0, top_counters, scratch_a, marked_val);
            k_isa_split<false><<<dim3(ntiles), dim3(ISP_BLOCK), 0, st>>>(nullptr, scratch_a, n, wbits, wbits + 6, win_counters, scratch_b, nullptr);
            by_window = scratch_b;
        } else {
            k_isa_init<<<dim3(1), dim3(256), 0, st>>>(top_counters, nwin, wbits);
            k_isa_split<true><<<dim3(ntiles), dim3(ISP_BLOCK), 0, st>>>(sa, nullptr, n, wbits, 0, top_counters, scratch_a, marked_val);
        }
    }
    {
        LaunchScope ls(ctx, K_ISA_ASSEMBLE, 12.0 * n);
        // a workgroup may declare more than the default 64 KiB of dynamic LDS only after this; the attribute belongs to the function ON THE
        // CURRENT DEVICE (a process may hold contexts on several GPUs), so it is set before every launch -- it costs a table lookup
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_isa_assemble), hipFuncAttributeMaxDynamicSharedMemorySize, 4 << ISA_WBITS) != hipSuccess)
            return ctx->fail(DK_E_HIP, "k_isa_assemble: cannot reserve %d bytes of LDS", 4 << ISA_WBITS);
        k_isa_assemble<<<dim3(nwin), dim3(ISA_BLOCK), sizeof(uint32_t) << wbits, st>>>(by_window, n, wbits, rank);
    }
    DK_HIP(ctx, hipGetLastError());
    ctx->ws_release(mark);
    return DK_OK;
}

// rank[sa[p]] = p (or marked_val[p] for entries of sa with bit 31 set) for a permutation sa of 0 .. n-1; scratch_a / scratch_b hold n u64 each
bool inverse_through_windows(size_t n) { return n <= ISA_MAX_N; }  // (every block: n < 2^31)

int inverse_permutation(dk_ctx *ctx, const uint32_t *sa, size_t n, uint64_t *scratch_a, uint64_t *scratch_b, uint32_t *rank, const uint32_t *marked_val) {
    if (n == 0) return DK_OK;
    if (!inverse_through_windows(n) || !scratch_a || !scratch_b) return ctx->fail(DK_E_INTERNAL, "inverse_permutation: n = %zu", n);
    return inverse_permutation_windows(ctx, sa, n, scratch_a, scratch_b, rank, marked_val);
}

}  // namespace dk
