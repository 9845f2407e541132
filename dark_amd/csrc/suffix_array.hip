// Suffix array on the GPU: radix sort on a packed prefix, text-extension rounds, then prefix doubling.  Replaces
// saca::Constructor::compute (src/saca.rs:368-378) and with it all of saca() (src/saca.rs:270-340); the result is the same array: the
// suffix array is unique, and like the reference there is no sentinel -- a suffix that is a proper prefix of another sorts first.
//
//   1. k_sym_hist        which byte values occur -> host picks an order-preserving code of b = ceil(log2 sigma) bits per symbol
//   2. k_prefix_probe    (large blocks) does a sample of suffixes separate on a short prefix already?  -> symbols the initial sort covers
//   3. sort_pairs        (key, i) by key, key = the first s symbols of suffix i packed big-endian (+ the code of the symbol in FRONT of
//                        the suffix in the low byte when the caller wants L: BwtCarry); the first pass builds the keys from the text
//                        -- and its last pass leaves every suffix at its slot of SA, its symbol in L
//   4. rerank            heads where the key changes (one flag byte per slot); singletons are final where they stand, the rest are
//                        compacted into the ACTIVE list (idx, slot position, group id, symbol in front)
//   5a. text rounds      every group sorted inside its slot range by the next 8 bytes of the text (k_round_local<true>); no ranks needed
//   5b. rank array       built once, for what survived: rank[SA[p]] = p -- the inverse permutation through LDS windows (radix_sort.hip);
//                        active suffixes get their head's position
//   5c. doubling rounds  secondary key rank[i+h]+h (n-1-i past the end: shorter is smaller); general rounds (k_round_local<false>, big
//                        groups through the global sort, rerank) while groups of more than 256 members exist, then in-place rounds
//                        (k_plateau_sort / k_plateau_ranks: no compaction, no scan, live count read back one round late)
// Only members of unresolved groups are ever sorted again (Larsson-Sadakane style filtering).  DESIGN.md section 4.1.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "context.hpp"
#include "device_util.hpp"

namespace dk {
namespace {

constexpr int RR_BLOCK = 256;
constexpr int RR_WAVES = RR_BLOCK / 64;
constexpr int RR_IPT = 8;                    // slots per thread (contiguous)
constexpr int RR_TILE = RR_BLOCK * RR_IPT;   // 2048 slots per workgroup
static_assert(RR_TILE < (1 << 15), "k_rerank_apply scans two tile-local counts in one word");

// Which byte values occur in the text?  (Only presence is needed: the host numbers the present symbols in order.)  16 bytes per load;
// every byte sets a flag in LDS with a plain one-byte store -- no atomics, so a text of four symbols does not serialise 64 lanes on four
// counters (2^28 ACGT: 0.17 -> 0.06 ms; a histogram with LDS atomics was what this kernel used to be).  Racing stores of the same
// value are harmless.  present[s] != 0 <=> s occurs.
__global__ __launch_bounds__(256) void k_sym_hist(const uint8_t *__restrict__ t, size_t n, uint32_t *__restrict__ present) {
    __shared__ uint8_t seen[256];
    seen[threadIdx.x] = 0;
    __syncthreads();
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    const size_t i0 = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(t) & 15) == 0) {
        const size_t n16 = n / 16;
        const uint4 *t16 = reinterpret_cast<const uint4 *>(t);
        for (size_t i = i0; i < n16; i += stride) {
            const uint4 v = t16[i];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                seen[w[k] & 0xFFu] = 1;
                seen[(w[k] >> 8) & 0xFFu] = 1;
                seen[(w[k] >> 16) & 0xFFu] = 1;
                seen[w[k] >> 24] = 1;
            }
        }
        for (size_t i = n16 * 16 + i0; i < n; i += stride) seen[t[i]] = 1;
    } else {
        for (size_t i = i0; i < n; i += stride) seen[t[i]] = 1;
    }
    __syncthreads();
    if (seen[threadIdx.x]) present[threadIdx.x] = 1;  // (plain store: every workgroup writes the same value)
}

// Runs of one byte value.  A thread looks at one aligned 256-byte window: flag[0] = a whole window of one value exists (a run of at least
// 511 bytes holds one), flag[1] += 16-byte pieces of one value (how much of the block lies inside runs: text has next to none, zero-padded
// binaries a lot).  The L-first BWT path refines groups from the text: the suffixes inside a run are one group that does not split that
// way, and where runs are common prefix doubling -- which resolves a run in log(length) rounds -- is the faster path by far (5 MB of 90 %
// zero bytes: 1.8 against 4.5 ms).
__global__ __launch_bounds__(256) void k_run_probe(const uint8_t *__restrict__ t, size_t n, uint32_t *__restrict__ flag) {
    const size_t w = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t p = w * 256;
    uint32_t pieces = 0;
    if (p + 256 <= n && (reinterpret_cast<uintptr_t>(t) & 15) == 0) {
        const uint4 *q = reinterpret_cast<const uint4 *>(t + p);
        const uint32_t rep0 = (q[0].x & 0xFFu) * 0x01010101u;
        uint32_t diff = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint4 v = q[k];
            const uint32_t rep = (v.x & 0xFFu) * 0x01010101u;
            pieces += ((v.x ^ rep) | (v.y ^ rep) | (v.z ^ rep) | (v.w ^ rep)) == 0 ? 1u : 0u;
            diff |= (v.x ^ rep0) | (v.y ^ rep0) | (v.z ^ rep0) | (v.w ^ rep0);
        }
        if (diff == 0) flag[0] = 1u;
    }
    pieces = wave_sum(pieces);
    if ((threadIdx.x & 63) == 0 && pieces) atomicAdd(&flag[1], pieces);
}

// ---- periodic stretches (runs of one byte value are period 1) ---------------------------------------------------------------------
// The suffixes inside a stretch of period p (T[j] == T[j+p] over a long range: zero padding, a^n, (ab)^n, arrays of one 32-bit value) are what
// prefix doubling is slowest on: they stay one group until the depth reaches the stretch's length, log2(length) rounds over all of them.  But
// their order follows from where the stretch ENDS: two suffixes x, y equal in their first h >= p symbols and p-periodic that far are
// ordered by e = LCE(x, x + p) and by how the shorter stretch breaks -- if e(x) < e(y), y still follows the period where x breaks it, so
// x < y exactly when x's breaking symbol T[x + e + p] is smaller than the symbol the period asks for there, T[x + e] (or the text ends).
// One round with the token (direction, e or its complement, the bytes from the break on) as secondary key settles them all (k_round_local).
// k_period_probe counts, for p = 1..8, the aligned 64-byte windows that follow period p; k_period_first / _spine / _fill give every position
// its next break: nb[i] = min { j >= i : j + p >= n or T[j] != T[j+p] }.
__global__ __launch_bounds__(256) void k_period_probe(const uint8_t *__restrict__ t, size_t n, uint32_t *__restrict__ counts) {
    const size_t w = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t p0 = w * 64;
    uint32_t ok = 0;  // bit p-1: the window follows period p
    if (p0 + 72 <= n && (reinterpret_cast<uintptr_t>(t) & 7) == 0) {
        const uint64_t *q = reinterpret_cast<const uint64_t *>(t + p0);
        uint64_t v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = q[k];
        ok = 0xFFu;
#pragma unroll
        for (int p = 1; p <= 8; ++p) {
            uint64_t diff = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) diff |= v[k] ^ (p == 8 ? v[k + 1] : ((v[k] >> (8 * p)) | (v[k + 1] << (64 - 8 * (p & 7)))));
            if (diff) ok &= ~(1u << (p - 1));
        }
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const uint32_t c = static_cast<uint32_t>(__popcll(__ballot((ok >> p) & 1u)));
        if ((threadIdx.x & 63) == 0 && c) atomicAdd(&counts[p], c);
    }
}

constexpr int PB_TILE = 4096;  // positions per workgroup: 256 threads x 16
// sixteen bytes from position j on (zero past the end) as two little-endian words
__device__ __forceinline__ void load16_le(const uint8_t *__restrict__ t, size_t n, size_t j, uint64_t *lo, uint64_t *hi) {
    if (j + 16 <= n && ((reinterpret_cast<uintptr_t>(t) + j) & 15) == 0) {
        const uint4 v = *reinterpret_cast<const uint4 *>(t + j);
        *lo = (static_cast<uint64_t>(v.y) << 32) | v.x;
        *hi = (static_cast<uint64_t>(v.w) << 32) | v.z;
        return;
    }
    uint64_t a = 0, b = 0;
    for (int k = 0; k < 8; ++k) {
        a |= static_cast<uint64_t>(j + k < n ? t[j + k] : 0u) << (8 * k);
        b |= static_cast<uint64_t>(j + 8 + k < n ? t[j + 8 + k] : 0u) << (8 * k);
    }
    *lo = a;
    *hi = b;
}
// sixteen bytes from any position (zero past the end): two unaligned 8-byte loads where the text has them
__device__ __forceinline__ void load16_any(const uint8_t *__restrict__ t, size_t n, size_t j, uint64_t *lo, uint64_t *hi) {
    if (j + 16 <= n) {
        typedef uint64_t __attribute__((aligned(1))) unaligned_u64;
        typedef const unaligned_u64 __attribute__((address_space(1))) *gptr8;
        *lo = *(gptr8)(t + j);
        *hi = *(gptr8)(t + j + 8);
        return;
    }
    load16_le(t, n, j, lo, hi);
}
// bit k of the result: position j + k is a break (T[j+k] != T[j+k+p], or j + k + p >= n); positions at or beyond n are breaks too
__device__ __forceinline__ uint32_t period_breaks16(const uint8_t *__restrict__ t, size_t n, size_t j, int p) {
    uint64_t w0, w1, w2, w3;
    load16_le(t, n, j, &w0, &w1);
    uint64_t s0, s1;
    if (p > 8) {  // long periods (k_period_search): the sixteen bytes p further on, wherever they lie
        load16_any(t, n, j + static_cast<size_t>(p), &s0, &s1);
    } else {
        load16_le(t, n, j + 16, &w2, &w3);
        (void)w3;
        s0 = p == 8 ? w1 : ((w0 >> (8 * p)) | (w1 << (64 - 8 * (p & 7))));
        s1 = p == 8 ? w2 : ((w1 >> (8 * p)) | (w2 << (64 - 8 * (p & 7))));
    }
    const uint64_t x0 = w0 ^ s0, x1 = w1 ^ s1;
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        m |= ((x0 >> (8 * k)) & 0xFFu) ? (1u << k) : 0u;
        m |= ((x1 >> (8 * k)) & 0xFFu) ? (1u << (8 + k)) : 0u;
    }
    // the text's end: every j' with j' + p >= n breaks
    const size_t lim = n >= static_cast<size_t>(p) ? n - p : 0;  // first such position
    if (j + 16 > lim) m |= lim > j ? (0xFFFFu << (lim - j)) & 0xFFFFu : 0xFFFFu;
    return m;
}
// Long periods (9 .. PS_MAX: fixed-size records, tables of one row).  The probe above only looks at p <= 8, and the token needs p <= h -- a long
// period is worth finding only for a block whose doubling rounds would run over everything anyway (giant groups hold most of it), and is used
// once the depth has reached it.  k_period_search: PS_SAMPLES evenly spread positions, a workgroup each; thread i tests p = 9 + i, 9 + i + 256, ...
// for 48 equal bytes and the smallest hit of a sample goes to found[sample].  k_period_count: 64-byte windows that follow a given p (any p).
constexpr int PS_SAMPLES = 32;
constexpr uint32_t PS_MAX = 1u << 18;
__global__ __launch_bounds__(256) void k_period_search(const uint8_t *__restrict__ t, size_t n, uint32_t pmax, uint32_t *__restrict__ found) {
    const size_t x = (2 * static_cast<size_t>(blockIdx.x) + 1) * n / (2 * PS_SAMPLES);
    if (x + pmax + 48 > n) return;  // (found[] was set to ~0: no answer from this sample)
    uint64_t a[6];
    for (int k = 0; k < 6; k += 2) load16_any(t, n, x + 8 * k, &a[k], &a[k + 1]);
    for (uint32_t p = 9 + threadIdx.x; p <= pmax; p += 256) {
        if (p >= found[blockIdx.x]) break;  // (a smaller hit exists already)
        uint64_t b[6];
        for (int k = 0; k < 6; k += 2) load16_any(t, n, x + p + 8 * k, &b[k], &b[k + 1]);
        if (((a[0] ^ b[0]) | (a[1] ^ b[1]) | (a[2] ^ b[2]) | (a[3] ^ b[3]) | (a[4] ^ b[4]) | (a[5] ^ b[5])) == 0) {
            atomicMin(&found[blockIdx.x], p);
            break;
        }
    }
}
__global__ __launch_bounds__(256) void k_period_count(const uint8_t *__restrict__ t, size_t n, uint32_t p, uint32_t *__restrict__ count) {
    const size_t w = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t p0 = w * 64;
    bool ok = false;
    if (p0 + 64 + p <= n) {
        uint64_t diff = 0;
        for (int k = 0; k < 4; ++k) {
            uint64_t a0, a1, b0, b1;
            load16_any(t, n, p0 + 16 * k, &a0, &a1);
            load16_any(t, n, p0 + p + 16 * k, &b0, &b1);
            diff |= (a0 ^ b0) | (a1 ^ b1);
        }
        ok = diff == 0;
    }
    const uint32_t c = static_cast<uint32_t>(__popcll(__ballot(ok)));
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}
constexpr uint32_t PB_NONE = 0xFFFFFFFFu;
// thread tid takes the tile's chunk 255 - tid: an exclusive running max over the threads in tid order then covers the chunks BEHIND its own
__global__ __launch_bounds__(256) void k_period_first(const uint8_t *__restrict__ t, uint32_t n, int p, uint32_t *__restrict__ tile_first) {
    __shared__ uint32_t s_tmp[4 + 1];
    const size_t j = static_cast<size_t>(blockIdx.x) * PB_TILE + static_cast<size_t>(255 - threadIdx.x) * 16;
    const uint32_t m = j < n ? period_breaks16(t, n, j, p) : 0xFFFFu;
    const uint32_t first = m ? static_cast<uint32_t>(j) + static_cast<uint32_t>(__builtin_ctz(m)) : PB_NONE;
    uint32_t total = 0;
    (void)block_excl_max<4>(PB_NONE - first, s_tmp, &total);
    if (threadIdx.x == 0) tile_first[blockIdx.x] = PB_NONE - total;
}
// tile_first[b] -> first break in the tiles behind b (PB_NONE: none)
__global__ __launch_bounds__(1024) void k_period_spine(uint32_t *__restrict__ tile_first, size_t ntiles) {
    __shared__ uint32_t s_tmp[16 + 1];
    const size_t per = (ntiles + 1023) / 1024;
    const size_t c = 1023 - threadIdx.x;  // chunk of tiles, taken in reverse like the positions of a tile
    const size_t b0 = c * per < ntiles ? c * per : ntiles, b1 = b0 + per < ntiles ? b0 + per : ntiles;
    uint32_t first = PB_NONE;
    for (size_t b = b1; b > b0; --b) first = tile_first[b - 1] != PB_NONE ? tile_first[b - 1] : first;
    uint32_t behind = PB_NONE - block_excl_max<16>(PB_NONE - first, s_tmp, nullptr);
    for (size_t b = b1; b > b0; --b) {
        const uint32_t v = tile_first[b - 1];
        tile_first[b - 1] = behind;
        if (v != PB_NONE) behind = v;
    }
}
__global__ __launch_bounds__(256) void k_period_fill(const uint8_t *__restrict__ t, uint32_t n, int p, const uint32_t *__restrict__ tile_behind,
                                                     uint32_t *__restrict__ nb) {
    __shared__ uint32_t s_tmp[4 + 1];
    const size_t j = static_cast<size_t>(blockIdx.x) * PB_TILE + static_cast<size_t>(255 - threadIdx.x) * 16;
    const uint32_t m = j < n ? period_breaks16(t, n, j, p) : 0xFFFFu;
    const uint32_t first = m ? static_cast<uint32_t>(j) + static_cast<uint32_t>(__builtin_ctz(m)) : PB_NONE;
    uint32_t next = PB_NONE - block_excl_max<4>(PB_NONE - first, s_tmp, nullptr);  // first break in the tile's chunks behind mine
    if (next == PB_NONE) next = tile_behind[blockIdx.x];
    if (j >= n) return;
    uint32_t out[16];
#pragma unroll
    for (int k = 15; k >= 0; --k) {
        if ((m >> k) & 1u) next = static_cast<uint32_t>(j) + static_cast<uint32_t>(k);
        out[k] = next;
    }
    if (j + 16 <= n) {
#pragma unroll
        for (int k = 0; k < 4; ++k) reinterpret_cast<uint4 *>(nb + j)[k] = make_uint4(out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]);
    } else {
        for (int k = 0; k < 16 && j + k < n; ++k) nb[j + k] = out[k];
    }
}

__global__ __launch_bounds__(256) void k_sa_descending(uint32_t *__restrict__ sa, size_t n) {
    const size_t j = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j < n) sa[j] = static_cast<uint32_t>(n - 1 - j);
}

// codes of T[p .. p+count) packed big-endian, zero padded past the end of the text (count * bits <= 64)
__device__ __forceinline__ uint64_t text_key(const uint8_t *__restrict__ t, size_t n, const uint8_t *__restrict__ code, size_t p, int bits,
                                             int count) {
    uint64_t key = 0;
    for (int j = 0; j < count; ++j) {
        const size_t q = p + j;
        key = (key << bits) | (q < n ? code[t[q]] : 0u);
    }
    return key;
}

// ---- how many leading symbols does the initial sort need? -------------------------------------------------------------
// m suffixes at distinct, evenly spread (jittered) positions; for each candidate prefix length the prefixes go into one hash table
// and equal ones are counted.  m is about 10 sqrt(n), so a sample without any equal pair says that (with high probability) fewer
// than a few percent of ALL suffixes share their prefix of that length with another suffix (c equal pairs: about c / 50 of them):
// sorting by that prefix alone leaves a small active list, which one text-extension round finishes.  Text-like inputs show thousands
// of equal pairs at every length and keep the full key.
constexpr int PP_MAX_CAND = 4;
struct ProbeCands { int count; int sym[PP_MAX_CAND]; };
constexpr uint64_t PP_EMPTY = ~0ull;

__global__ __launch_bounds__(256) void k_prefix_probe(const uint8_t *__restrict__ t, size_t n, const uint8_t *__restrict__ code, int bits,
                                                      int spk, uint32_t m, uint32_t span, ProbeCands cands, uint64_t *__restrict__ table,
                                                      uint32_t table_mask, uint32_t *__restrict__ dups) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    uint32_t hsh = i * 2654435761u;
    hsh ^= hsh >> 15;
    const size_t pos = static_cast<size_t>(i) * span + (hsh % span);
    if (pos >= n) return;
    const uint64_t key = text_key(t, n, code, pos, bits, spk);
    for (int c = 0; c < cands.count; ++c) {
        const uint64_t tagged = ((key >> (bits * (spk - cands.sym[c]))) << 2) | static_cast<uint64_t>(c);  // at most 58 bits
        uint64_t x = tagged * 0x9E3779B97F4A7C15ull;
        uint32_t slot = static_cast<uint32_t>(x >> 32) & table_mask;
        for (uint32_t tries = 0; tries <= table_mask; ++tries) {
            const uint64_t old = atomicCAS(reinterpret_cast<unsigned long long *>(table + slot), static_cast<unsigned long long>(PP_EMPTY),
                                           static_cast<unsigned long long>(tagged));
            if (old == PP_EMPTY) break;
            if (old == tagged) { atomicAdd(&dups[c], 1u); break; }
            slot = (slot + 1) & table_mask;
        }
    }
}

// ---- rank array for the doubling rounds, built once for whatever the text rounds leave -------------------------------------------
// Every suffix that is final sits in SA; put the active ones at their (provisional) places too, store rank[SA[p]] = p for all p
// (inverse permutation: radix_sort.hip), then give the active ones the position of their group's head.
// head_pos != nullptr: the suffix is stored with bit 31 set and head_pos[its SA position] = the position of its group's head -- the
// inverse permutation then gives an active suffix its head's position straight away (k_isa_split), and k_rank_active is not needed
__global__ __launch_bounds__(256) void k_place_active(const uint32_t *__restrict__ act_idx, const uint32_t *__restrict__ act_pos,
                                                      const uint32_t *__restrict__ act_gid, const uint32_t *__restrict__ gstart, size_t count,
                                                      uint32_t *__restrict__ sa, uint32_t *__restrict__ head_pos) {
    const size_t a = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (a >= count) return;
    const uint32_t p = act_pos[a];
    if (head_pos) {
        sa[p] = act_idx[a] | 0x80000000u;
        head_pos[p] = act_pos[gstart[act_gid[a]]];
    } else {
        sa[p] = act_idx[a];
    }
}
__global__ __launch_bounds__(256) void k_rank_active(const uint32_t *__restrict__ act_idx, const uint32_t *__restrict__ act_pos,
                                                     const uint32_t *__restrict__ act_gid, const uint32_t *__restrict__ gstart, size_t count,
                                                     uint32_t *__restrict__ rank) {
    const size_t a = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (a < count) rank[act_idx[a]] = act_pos[gstart[act_gid[a]]];
}

// ---- rerank: three kernels sharing the per-slot flag logic ------------------------------------------------------
struct RerankAgg { uint32_t surv, heads, last_head, pad; };

// Per-slot flags, computed with lane-contiguous (coalesced) accesses and parked in LDS as one byte per slot:
//   bit 0 head : the key differs from its predecessor (slot 0 is a head)
//   bit 1 surv : the slot's group has more than one member
//   bit 2 oldh : the slot is a head AND already headed its group before this round (gid_in: the group every slot belonged to before
//                the round -- members only move inside their group's slot range, so that is a property of the slot); members of
//                such a group keep their rank, so the rank scatter is skipped
// The keys of one old group are compared among themselves only: across a group boundary a new group starts whatever the keys say,
// so a round's key need not repeat the group id above its secondary key.
constexpr uint32_t F_HEAD = 1, F_SURV = 2, F_OLDH = 4;

// the RR_IPT consecutive slots of this thread: counts and the last head, encoded (slot << 1) | old_head so that one
// max-scan carries both
__device__ __forceinline__ void thread_summary(uint64_t fl, size_t a0, uint32_t &ns, uint32_t &nh, uint32_t &lh) {
    ns = nh = lh = 0;
#pragma unroll
    for (int j = 0; j < RR_IPT; ++j) {
        const uint32_t f = static_cast<uint32_t>(fl >> (8 * j)) & 0xFFu;
        ns += (f & F_SURV) ? 1u : 0u;
        nh += ((f & F_HEAD) && (f & F_SURV)) ? 1u : 0u;
        if (f & F_HEAD) lh = (static_cast<uint32_t>(a0 + j) << 1) | ((f & F_OLDH) ? 1u : 0u);
    }
}

// k_rerank_reduce: the flag byte of every slot (flags_out: whole tiles, the array holds ntiles * RR_TILE bytes, slots past `count` get 0 --
// the apply phase reads these instead of the keys and group ids it would need to work the flags out again) and the aggregates of every
// tile of RR_TILE slots.  A thread owns RR_IPT consecutive slots: their keys, the one before and the one after arrive in registers
// with 16-byte loads (no staging through LDS), the flags never leave the registers.  A workgroup takes tiles_per_wg consecutive tiles
// (524 288 workgroups of one tile each were dispatch-bound at 2^30 slots: 2.1 ms whether the keys had 8 bytes or 4), the loads of the
// next tile in flight while the current one is reduced.
// cmp_shift: low bits of the keys that take no part in the comparison (the byte that carries the symbol in front of the suffix).
// K = uint32_t: the first rerank after a sort that left NARROW keys (SortFinalOut::narrow_shift: 32-bit words, at most five passes).
// The bits a fifth pass sorted by are not in the words: a slot where that pass's digit changes (bucket_starts: 256 places in the whole
// array, those inside the workgroup's stretch are collected once) is a head whatever its word says.  No old groups at that stage.
template <typename K>
struct ReduceSlice { K k[RR_IPT + 2]; uint32_t g[RR_IPT + 2]; };

template <typename K>
__global__ __launch_bounds__(RR_BLOCK) void k_rerank_reduce(const K *__restrict__ keys, size_t count, const uint32_t *__restrict__ gid_in,
                                                             RerankAgg *__restrict__ agg, int cmp_shift, uint8_t *__restrict__ flags_out,
                                                             const uint32_t *__restrict__ bucket_starts, uint32_t ntiles, uint32_t tiles_per_wg) {
    constexpr bool NARROW = sizeof(K) == 4;
    __shared__ uint32_t s_red[2][3][RR_WAVES];
    __shared__ uint32_t s_forced[NARROW ? 256 : 1];
    __shared__ uint32_t s_nforced;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t t0 = blockIdx.x * tiles_per_wg;
    const uint32_t t1 = t0 + tiles_per_wg < ntiles ? t0 + tiles_per_wg : ntiles;
    if (NARROW) {
        static_assert(RR_BLOCK == 256, "one thread per bucket");
        if (tid == 0) s_nforced = 0;
        __syncthreads();
        const uint64_t s = bucket_starts[tid];
        if (s >= static_cast<uint64_t>(t0) * RR_TILE && s <= static_cast<uint64_t>(t1) * RR_TILE) s_forced[atomicAdd(&s_nforced, 1u)] = static_cast<uint32_t>(s);
        __syncthreads();
    }
    auto load_slice = [&](uint32_t tile, ReduceSlice<K> &r) {
        const size_t a0 = static_cast<size_t>(tile) * RR_TILE + static_cast<size_t>(tid) * RR_IPT;
        if (a0 >= 1 && a0 + RR_IPT + 1 <= count && (reinterpret_cast<uintptr_t>(keys) & 15) == 0 && (reinterpret_cast<uintptr_t>(gid_in) & 15) == 0) {
            constexpr int PER = 16 / sizeof(K);  // keys per 16-byte load
            const uint4 *p = reinterpret_cast<const uint4 *>(keys + a0);
#pragma unroll
            for (int q = 0; q < RR_IPT / PER; ++q) {
                const uint4 v = p[q];
                if (NARROW) {
                    r.k[1 + 4 * q] = static_cast<K>(v.x); r.k[2 + 4 * q] = static_cast<K>(v.y); r.k[3 + 4 * q] = static_cast<K>(v.z); r.k[4 + 4 * q] = static_cast<K>(v.w);
                } else {
                    r.k[1 + 2 * q] = static_cast<K>((static_cast<uint64_t>(v.y) << 32) | v.x);
                    r.k[2 + 2 * q] = static_cast<K>((static_cast<uint64_t>(v.w) << 32) | v.z);
                }
            }
            r.k[0] = keys[a0 - 1];
            r.k[RR_IPT + 1] = keys[a0 + RR_IPT];
            if (gid_in) {
                const uint4 g0 = *reinterpret_cast<const uint4 *>(gid_in + a0), g1 = *reinterpret_cast<const uint4 *>(gid_in + a0 + 4);
                r.g[1] = g0.x; r.g[2] = g0.y; r.g[3] = g0.z; r.g[4] = g0.w; r.g[5] = g1.x; r.g[6] = g1.y; r.g[7] = g1.z; r.g[8] = g1.w;
                r.g[0] = gid_in[a0 - 1];
                r.g[RR_IPT + 1] = gid_in[a0 + RR_IPT];
            }
        } else {  // the array's ends
#pragma unroll
            for (int j = 0; j < RR_IPT + 2; ++j) {
                const size_t a = a0 + j;  // slot a - 1
                const bool in = a >= 1 && a - 1 < count;
                r.k[j] = in ? keys[a - 1] : K(0);
                r.g[j] = in && gid_in ? gid_in[a - 1] : 0u;
            }
        }
    };
    static_assert(RR_IPT == 8, "two 16-byte loads of group ids per thread");
    if (t0 >= t1) return;
    ReduceSlice<K> cur, nxt;
    load_slice(t0, cur);
    for (uint32_t tile = t0; tile < t1; ++tile) {
        if (tile + 1 < t1) load_slice(tile + 1, nxt);
        const size_t a0 = static_cast<size_t>(tile) * RR_TILE + static_cast<size_t>(tid) * RR_IPT;
        uint32_t forced = 0;  // bit j: slot a0 + j - 1 starts a bucket (j = 0 .. RR_IPT + 1)
        if (NARROW) {
            const uint32_t nf = s_nforced;
            for (uint32_t i = 0; i < nf; ++i) {
                const uint64_t rel = static_cast<uint64_t>(s_forced[i]) + 1 - a0;  // (mod 2^64: far off when the start lies before the slice)
                if (rel < RR_IPT + 2) forced |= 1u << rel;
            }
        }
        uint64_t fl8 = 0;
#pragma unroll
        for (int j = 0; j < RR_IPT; ++j) {
            const size_t a = a0 + j;
            if (a >= count) break;
            const K prev = NARROW ? cur.k[j] : static_cast<K>(cur.k[j] >> cmp_shift), here = NARROW ? cur.k[j + 1] : static_cast<K>(cur.k[j + 1] >> cmp_shift),
                    next = NARROW ? cur.k[j + 2] : static_cast<K>(cur.k[j + 2] >> cmp_shift);
            // The keys of one old group are compared among themselves only: across a group boundary a new group starts whatever the keys
            // say, so a round's key need not repeat the group id above its secondary key.
            bool old_here = a == 0, old_next = a + 1 >= count;
            if (gid_in) {
                old_here = old_here || cur.g[j] != cur.g[j + 1];
                old_next = old_next || cur.g[j + 2] != cur.g[j + 1];
            }
            const bool head = old_here || here != prev || ((forced >> (j + 1)) & 1u);
            const bool next_head = old_next || next != here || ((forced >> (j + 2)) & 1u);
            const bool oldh = head && gid_in && old_here;
            const uint32_t f = (head ? F_HEAD : 0u) | (!(head && next_head) ? F_SURV : 0u) | (oldh ? F_OLDH : 0u);
            fl8 |= static_cast<uint64_t>(f) << (8 * j);
        }
        *reinterpret_cast<uint64_t *>(flags_out + a0) = fl8;
        uint32_t ns, nh, lh;
        thread_summary(fl8, a0, ns, nh, lh);
        ns = wave_sum(ns);
        nh = wave_sum(nh);
        lh = wave_max(lh);
        uint32_t (*red)[RR_WAVES] = s_red[tile & 1u];  // two sets: a set is written again two tiles later, behind the barrier of the tile between
        if (lane == 0) { red[0][wave] = ns; red[1][wave] = nh; red[2][wave] = lh; }
        __syncthreads();
        if (tid == 0) {
            RerankAgg r{0, 0, 0, 0};
            for (int w = 0; w < RR_WAVES; ++w) {
                r.surv += red[0][w];
                r.heads += red[1][w];
                r.last_head = r.last_head > red[2][w] ? r.last_head : red[2][w];
            }
            agg[tile] = r;
        }
        if (tile + 1 < t1) cur = nxt;
    }
}

// one workgroup: exclusive scan of the per-tile aggregates (sum, sum, running max); totals -> mail[0..1]
__global__ __launch_bounds__(1024) void k_rerank_scan(RerankAgg *__restrict__ agg, size_t ntiles, uint32_t *__restrict__ mail,
                                                      uint32_t *__restrict__ gstart) {
    __shared__ uint32_t s_tmp[16 + 1];
    const int tid = threadIdx.x;
    const size_t per = (ntiles + 1023) / 1024;
    const size_t b0 = static_cast<size_t>(tid) * per;
    const size_t b1 = b0 + per < ntiles ? b0 + per : ntiles;
    uint32_t ns = 0, nh = 0, lh = 0;
#pragma unroll 16
    for (size_t b = b0; b < b1; ++b) {  // (a thread's stretch is contiguous: unrolled, its loads are in flight together)
        const RerankAgg r = agg[b];
        ns += r.surv;
        nh += r.heads;
        lh = lh > r.last_head ? lh : r.last_head;
    }
    uint32_t tot_s, tot_h;
    uint32_t es = block_excl_sum<16>(ns, s_tmp, &tot_s);
    uint32_t eh = block_excl_sum<16>(nh, s_tmp, &tot_h);
    uint32_t el = block_excl_max<16>(lh, s_tmp, nullptr);
#pragma unroll 16
    for (size_t b = b0; b < b1; ++b) {
        const RerankAgg r = agg[b];
        agg[b] = RerankAgg{es, eh, el, 0};
        es += r.surv;
        eh += r.heads;
        el = el > r.last_head ? el : r.last_head;
    }
    if (tid == 0) { mail[0] = tot_s; mail[1] = tot_h; gstart[tot_h] = tot_s; }  // sentinel: one past the last group
}

// The same scan for many tiles, in three phases over chunks of RS_CHUNK tiles (one workgroup reads at the rate of one CU: 524 288
// aggregates of a 2^30-byte block took it 0.4 ms): A reduces every chunk, B (one workgroup) scans the chunk aggregates, C scans inside
// every chunk from its chunk's base.
constexpr int RSC_CHUNK = 1024, RSC_IPT = 4;  // tiles per chunk = 256 threads x 4
__global__ __launch_bounds__(256) void k_rerank_scan_a(const RerankAgg *__restrict__ agg, size_t ntiles, RerankAgg *__restrict__ chunk) {
    __shared__ uint32_t s_red[3][RR_WAVES];
    const size_t b0 = static_cast<size_t>(blockIdx.x) * RSC_CHUNK + static_cast<size_t>(threadIdx.x) * RSC_IPT;
    uint32_t ns = 0, nh = 0, lh = 0;
#pragma unroll
    for (int j = 0; j < RSC_IPT; ++j) {
        if (b0 + j < ntiles) {
            const RerankAgg r = agg[b0 + j];
            ns += r.surv;
            nh += r.heads;
            lh = lh > r.last_head ? lh : r.last_head;
        }
    }
    ns = wave_sum(ns);
    nh = wave_sum(nh);
    lh = wave_max(lh);
    if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = ns; s_red[1][threadIdx.x >> 6] = nh; s_red[2][threadIdx.x >> 6] = lh; }
    __syncthreads();
    if (threadIdx.x == 0) {
        RerankAgg r{0, 0, 0, 0};
        for (int w = 0; w < RR_WAVES; ++w) {
            r.surv += s_red[0][w];
            r.heads += s_red[1][w];
            r.last_head = r.last_head > s_red[2][w] ? r.last_head : s_red[2][w];
        }
        chunk[blockIdx.x] = r;
    }
}
__global__ __launch_bounds__(256) void k_rerank_scan_c(RerankAgg *__restrict__ agg, size_t ntiles, const RerankAgg *__restrict__ chunk) {
    __shared__ uint32_t s_tmp[RR_WAVES + 1];
    const size_t b0 = static_cast<size_t>(blockIdx.x) * RSC_CHUNK + static_cast<size_t>(threadIdx.x) * RSC_IPT;
    RerankAgg r[RSC_IPT];
    uint32_t ns = 0, nh = 0, lh = 0;
#pragma unroll
    for (int j = 0; j < RSC_IPT; ++j) {
        r[j] = b0 + j < ntiles ? agg[b0 + j] : RerankAgg{0, 0, 0, 0};
        ns += r[j].surv;
        nh += r[j].heads;
        lh = lh > r[j].last_head ? lh : r[j].last_head;
    }
    const RerankAgg base = chunk[blockIdx.x];  // exclusive over the chunks (phase B)
    uint32_t es = base.surv + block_excl_sum<RR_WAVES>(ns, s_tmp, nullptr);
    uint32_t eh = base.heads + block_excl_sum<RR_WAVES>(nh, s_tmp, nullptr);
    uint32_t el = block_excl_max<RR_WAVES>(lh, s_tmp, nullptr);
    el = el > base.last_head ? el : base.last_head;
#pragma unroll
    for (int j = 0; j < RSC_IPT; ++j) {
        if (b0 + j < ntiles) agg[b0 + j] = RerankAgg{es, eh, el, 0};
        es += r[j].surv;
        eh += r[j].heads;
        el = el > r[j].last_head ? el : r[j].last_head;
    }
}

// BWT on the way.  When the caller wants L, the initial keys carry, below the sorted bits, the code of the symbol in FRONT of the suffix
// (cmp_shift = 8 low bits that take no part in any comparison).  The first rerank turns it into the symbol itself: a suffix that is
// final writes L[its SA position] (and, for suffix 0, the origin); a suffix that stays active takes the symbol along in a byte list
// beside (idx, pos, gid) -- sym_in / sym_out in the later rounds -- and writes it when it becomes final.  No gather from the text, ever.
struct BwtCarry {
    int cmp_shift;            // first rerank only: low key bits outside the comparison
    uint8_t *bwt;             // L (nullptr: the caller wants the suffix array only)
    const uint8_t *inv_code;  // first rerank: code -> byte
    uint32_t *origin;
    const uint8_t *sym_in;    // later reranks: symbol in front of the suffix in slot a
    uint8_t *sym_out;         // compacted with the active list
};
constexpr BwtCarry NO_CARRY{0, nullptr, nullptr, nullptr, nullptr, nullptr};

// The flags come from k_rerank_reduce's byte array; the tile's lists arrive through LDS with lane-contiguous loads, the compacted outputs
// leave through LDS the same way.  (The first rerank, straight after the initial sort, has its own kernel below.)
__global__ __launch_bounds__(RR_BLOCK) void k_rerank_apply(const uint8_t *__restrict__ flags, const uint32_t *__restrict__ idx,
                                                            const uint32_t *__restrict__ pos_in, size_t count,
                                                            const RerankAgg *__restrict__ agg, uint32_t *__restrict__ rank,
                                                            uint32_t *__restrict__ sa, uint32_t *__restrict__ out_idx,
                                                            uint32_t *__restrict__ out_pos, uint32_t *__restrict__ out_gid,
                                                            uint32_t *__restrict__ gstart, BwtCarry bc) {
    __shared__ __attribute__((aligned(16))) uint32_t s_out[2 * RR_TILE];  // compacted idx | pos
    __shared__ __attribute__((aligned(16))) uint32_t s_idx[RR_TILE];
    __shared__ __attribute__((aligned(16))) uint32_t s_pos[RR_TILE];
    __shared__ __attribute__((aligned(8))) uint8_t s_osym[RR_TILE];        // compacted symbols
    __shared__ uint32_t s_tmp[RR_WAVES + 1];
    const int tid = threadIdx.x;
    const size_t b0 = static_cast<size_t>(blockIdx.x) * RR_TILE;
#pragma unroll
    for (int k = 0; k < RR_IPT; ++k) {
        const int o = k * RR_BLOCK + tid;
        const size_t a = b0 + o;
        if (a < count) {
            s_idx[o] = idx[a];
            s_pos[o] = pos_in[a];
        }
    }
    const size_t a0 = b0 + static_cast<size_t>(tid) * RR_IPT;
    const uint64_t fl = *reinterpret_cast<const uint64_t *>(flags + a0);  // whole tiles are stored: zero past `count`
    uint64_t sym8 = 0;  // bc.bwt: the symbols in front of this thread's eight suffixes
    if (bc.bwt && a0 < count) {
        const uint8_t *src = bc.sym_in;
        if (a0 + RR_IPT <= count && ((reinterpret_cast<uintptr_t>(src) + a0) & 7) == 0) {
            sym8 = *reinterpret_cast<const uint64_t *>(src + a0);
        } else {
            for (int j = 0; j < RR_IPT && a0 + j < count; ++j) sym8 |= static_cast<uint64_t>(src[a0 + j]) << (8 * j);
        }
    }
    __syncthreads();  // s_idx / s_pos are visible
    uint32_t ns, nh, lh;
    thread_summary(fl, a0, ns, nh, lh);
    const RerankAgg base = agg[blockIdx.x];
    // survivors and surviving heads before this thread inside the tile: one scan for both (each is at most RR_TILE = 2^11)
    uint32_t block_surv;
    const uint32_t both = block_excl_sum<RR_WAVES>(ns | (nh << 16), s_tmp, &block_surv);
    block_surv &= 0xFFFFu;
    uint32_t es = both & 0xFFFFu;  // tile-local compaction offset
    uint32_t eh = base.heads + (both >> 16);
    uint32_t el = block_excl_max<RR_WAVES>(lh, s_tmp, nullptr);
    el = el > base.last_head ? el : base.last_head;
    uint32_t *s_oidx = s_out;
    uint32_t *s_opos = s_out + RR_TILE;
    uint32_t my_idx[RR_IPT], my_pos[RR_IPT];
    {
        const uint4 *pi = reinterpret_cast<const uint4 *>(s_idx + tid * RR_IPT);
        const uint4 i0 = pi[0], i1 = pi[1];
        my_idx[0] = i0.x; my_idx[1] = i0.y; my_idx[2] = i0.z; my_idx[3] = i0.w; my_idx[4] = i1.x; my_idx[5] = i1.y; my_idx[6] = i1.z; my_idx[7] = i1.w;
        const uint4 *pp = reinterpret_cast<const uint4 *>(s_pos + tid * RR_IPT);
        const uint4 p0 = pp[0], p1 = pp[1];
        my_pos[0] = p0.x; my_pos[1] = p0.y; my_pos[2] = p0.z; my_pos[3] = p0.w; my_pos[4] = p1.x; my_pos[5] = p1.y; my_pos[6] = p1.z; my_pos[7] = p1.w;
    }
    __syncthreads();              // every thread holds its slice of s_idx in registers:
    uint32_t *s_ogid = s_idx;     // the array now collects the compacted group ids
#pragma unroll
    for (int j = 0; j < RR_IPT; ++j) {
        const size_t a = a0 + j;
        if (a >= count) break;
        const uint32_t f = static_cast<uint32_t>(fl >> (8 * j)) & 0xFFu;
        if (f & F_HEAD) el = (static_cast<uint32_t>(a) << 1) | ((f & F_OLDH) ? 1u : 0u);
        const uint32_t suffix = my_idx[j];
        // SA position of the group's head: the head slot is in this tile (LDS) or in an earlier one (global, rare)
        const size_t hs = el >> 1;
        const uint32_t head_pos = hs >= b0 ? s_pos[hs - b0] : pos_in[hs];
        if (rank && !(el & 1u)) rank[suffix] = head_pos;  // members of a group that kept its head keep their rank: no scatter
        if (!(f & F_SURV)) {  // the group is a singleton: this suffix is in its final place
            if (sa) sa[my_pos[j]] = suffix;  // (null: a BWT caller behind the rank array's construction -- nobody reads SA again, L is what counts)
            if (bc.bwt) {
                bc.bwt[my_pos[j]] = static_cast<uint8_t>(sym8 >> (8 * j));
                if (suffix == 0) *bc.origin = my_pos[j];
            }
        } else {
            if (f & F_HEAD) { gstart[eh] = base.surv + es; ++eh; }  // first slot of the surviving group in the new active list
            s_oidx[es] = suffix;
            s_opos[es] = my_pos[j];
            s_ogid[es] = eh - 1;
            if (bc.bwt) s_osym[es] = static_cast<uint8_t>(sym8 >> (8 * j));
            ++es;
        }
    }
    __syncthreads();
    for (uint32_t o = tid; o < block_surv; o += RR_BLOCK) {
        out_idx[base.surv + o] = s_oidx[o];
        out_pos[base.surv + o] = s_opos[o];
        out_gid[base.surv + o] = s_ogid[o];
        if (bc.bwt) bc.sym_out[base.surv + o] = s_osym[o];
    }
}

// The first rerank, straight after the initial sort: slot a IS SA position a, and the sort's last pass left every suffix at its slot
// (and its symbol in L) -- finals need no store, only the survivors are compacted into the active list; no rank array exists yet.
// A workgroup takes up to RA_FIRST_TILES consecutive tiles (the host aims at about 8192 workgroups) and first looks up, one thread per
// tile, which of them hold a survivor at all: 2^30 random bytes are 524 288 tiles, nearly all of them without one (a workgroup each was
// 1.7 ms of dispatch and two dependent loads).  Every thread loads its own eight consecutive slots with 16-byte loads (no staging
// through LDS), and the loads of the next tile with survivors are in flight while the current one is scanned and compacted.
constexpr int RA_FIRST_TILES = 64;
constexpr int RA_SPARSE = 64;  // survivors per tile up to which the tile's suffixes are fetched one by one
struct RerankSlice { uint64_t fl, sym8; uint4 i0, i1; };
__global__ __launch_bounds__(RR_BLOCK) void k_rerank_apply_first(const uint8_t *__restrict__ flags, const uint32_t *__restrict__ idx, size_t count,
                                                                  const RerankAgg *__restrict__ agg, uint32_t *__restrict__ out_idx,
                                                                  uint32_t *__restrict__ out_pos, uint32_t *__restrict__ out_gid,
                                                                  uint32_t *__restrict__ gstart, const uint32_t *__restrict__ mail, BwtCarry bc,
                                                                  uint32_t ntiles, uint32_t tiles_per_wg, uint32_t sparse_max) {
    __shared__ __attribute__((aligned(16))) uint32_t s_out[3 * RR_TILE];  // compacted idx | pos | gid
    __shared__ __attribute__((aligned(8))) uint8_t s_osym[RR_TILE];        // compacted symbols
    __shared__ uint32_t s_tmp[RR_WAVES + 1];
    __shared__ RerankAgg s_agg[RA_FIRST_TILES + 1];
    const int tid = threadIdx.x;
    const uint32_t tile_first = blockIdx.x * tiles_per_wg;
    const uint32_t tile_end = tile_first + tiles_per_wg < ntiles ? tile_first + tiles_per_wg : ntiles;
    if (static_cast<uint32_t>(tid) <= tiles_per_wg)  // the aggregates in front of each of this workgroup's tiles (and of the one behind them)
        s_agg[tid] = tile_first + tid < ntiles ? agg[tile_first + tid] : RerankAgg{mail[0], 0, 0, 0};
    __syncthreads();
    auto next_with_survivors = [&](uint32_t t) {
        while (t < tile_end && s_agg[t - tile_first].surv == s_agg[t - tile_first + 1].surv) ++t;
        return t;
    };
    // a tile with few survivors (random bytes: two in 2048 slots) fetches suffix and symbol for those alone: one flag byte per slot is all
    // the rest of it costs
    auto sparse = [&](uint32_t t) { return s_agg[t - tile_first + 1].surv - s_agg[t - tile_first].surv <= sparse_max; };
    auto load_slice = [&](uint32_t t, RerankSlice &r) {
        const size_t a0 = static_cast<size_t>(t) * RR_TILE + static_cast<size_t>(tid) * RR_IPT;
        r.fl = *reinterpret_cast<const uint64_t *>(flags + a0);  // whole tiles are stored: zero past `count`
        r.sym8 = 0;
        if (sparse(t)) return;
        if (a0 + RR_IPT <= count && (reinterpret_cast<uintptr_t>(idx) & 15) == 0 && (reinterpret_cast<uintptr_t>(bc.bwt) & 7) == 0) {
            r.i0 = *reinterpret_cast<const uint4 *>(idx + a0);
            r.i1 = *reinterpret_cast<const uint4 *>(idx + a0 + 4);
            if (bc.bwt) r.sym8 = *reinterpret_cast<const uint64_t *>(bc.bwt + a0);  // L[slot] = the symbol in front of the suffix standing there
        } else {
            uint32_t v[RR_IPT];
            for (int j = 0; j < RR_IPT; ++j) {
                v[j] = a0 + j < count ? idx[a0 + j] : 0u;
                if (bc.bwt && a0 + j < count) r.sym8 |= static_cast<uint64_t>(bc.bwt[a0 + j]) << (8 * j);
            }
            r.i0 = make_uint4(v[0], v[1], v[2], v[3]);
            r.i1 = make_uint4(v[4], v[5], v[6], v[7]);
        }
    };
    uint32_t tile = next_with_survivors(tile_first);
    if (tile >= tile_end) return;
    RerankSlice cur, nxt;
    load_slice(tile, cur);
    uint32_t *s_oidx = s_out, *s_opos = s_out + RR_TILE, *s_ogid = s_out + 2 * RR_TILE;
    for (;;) {
        const uint32_t tile_next = next_with_survivors(tile + 1);
        if (tile_next < tile_end) load_slice(tile_next, nxt);
        const size_t a0 = static_cast<size_t>(tile) * RR_TILE + static_cast<size_t>(tid) * RR_IPT;
        uint32_t ns, nh, lh;
        thread_summary(cur.fl, a0, ns, nh, lh);
        const RerankAgg base = s_agg[tile - tile_first];
        // survivors and surviving heads before this thread inside the tile: one scan for both (each is at most RR_TILE = 2^11)
        uint32_t block_surv;
        const uint32_t both = block_excl_sum<RR_WAVES>(ns | (nh << 16), s_tmp, &block_surv);
        block_surv &= 0xFFFFu;
        uint32_t es = both & 0xFFFFu;  // tile-local compaction offset
        uint32_t eh = base.heads + (both >> 16);
        const uint32_t my_idx[RR_IPT] = {cur.i0.x, cur.i0.y, cur.i0.z, cur.i0.w, cur.i1.x, cur.i1.y, cur.i1.z, cur.i1.w};
        const bool few = sparse(tile);
        // The handful of survivors of a sparse tile go straight to the lists, and the tile needs no barrier beyond its scan; otherwise they
        // are compacted in LDS first (direct stores up to a quarter of the tile surviving were measured on 2^28 {A,C,G,T}, 123 survivors per
        // tile: 1.11 against 0.95 ms -- eight rounds of four stores with six lanes in a hundred active).
        const bool direct = few;
#pragma unroll
        for (int j = 0; j < RR_IPT; ++j) {
            const uint32_t f = static_cast<uint32_t>(cur.fl >> (8 * j)) & 0xFFu;  // (zero past `count`)
            if (f & F_SURV) {
                if (f & F_HEAD) { gstart[eh] = base.surv + es; ++eh; }  // first slot of the surviving group in the new active list
                const uint32_t suffix = few ? idx[a0 + j] : my_idx[j];
                const uint8_t symbol = !bc.bwt ? uint8_t(0) : few ? bc.bwt[a0 + j] : static_cast<uint8_t>(cur.sym8 >> (8 * j));
                if (direct) {
                    out_idx[base.surv + es] = suffix;
                    out_pos[base.surv + es] = static_cast<uint32_t>(a0) + j;  // slot a sits at SA position a
                    out_gid[base.surv + es] = eh - 1;
                    if (bc.bwt) bc.sym_out[base.surv + es] = symbol;
                } else {
                    s_oidx[es] = suffix;
                    s_opos[es] = static_cast<uint32_t>(a0) + j;
                    s_ogid[es] = eh - 1;
                    if (bc.bwt) s_osym[es] = symbol;
                }
                ++es;
            }
        }
        if (tile_next >= tile_end && direct) break;
        if (!direct) {
            __syncthreads();
            for (uint32_t o = tid; o < block_surv; o += RR_BLOCK) {
                out_idx[base.surv + o] = s_oidx[o];
                out_pos[base.surv + o] = s_opos[o];
                out_gid[base.surv + o] = s_ogid[o];
                if (bc.bwt) bc.sym_out[base.surv + o] = s_osym[o];
            }
        }
        if (tile_next >= tile_end) break;
        if (!direct) __syncthreads();  // the next tile reuses the staging arrays
        cur = nxt;
        tile = tile_next;
    }
}

// rank == nullptr: no rank array exists yet (first rerank, text rounds)
// narrow_starts != nullptr (first rerank only): `keys` holds 32-bit words (SortFinalOut::narrow_shift), narrow_starts the 256 bucket starts
int rerank(dk_ctx *ctx, const uint64_t *keys, const uint32_t *idx, const uint32_t *pos_in, size_t count, const uint32_t *gid_in, uint32_t *rank,
           uint32_t *sa, uint32_t *out_idx, uint32_t *out_pos, uint32_t *out_gid, uint32_t *gstart, BwtCarry fb = NO_CARRY,
           const uint32_t *narrow_starts = nullptr) {
    const size_t ntiles = div_up(count, RR_TILE);
    const size_t mark = ctx->ws_mark();
    RerankAgg *agg = ctx->ws_alloc<RerankAgg>(ntiles);
    uint8_t *flags = ctx->ws_alloc<uint8_t>(ntiles * RR_TILE);  // whole tiles
    if (!agg || !flags) return DK_E_NOMEM;
    hipStream_t st = ctx->stream;
    {
        LaunchScope ls(ctx, K_RERANK_REDUCE, ((narrow_starts ? 4.0 : 8.0) + (gid_in ? 4.0 : 0.0) + 1.0) * count);
        const uint32_t per_wg = static_cast<uint32_t>(std::min<size_t>(std::max<size_t>(ntiles / 16384, 1), 64));
        const dim3 grid(static_cast<unsigned>(div_up(ntiles, per_wg))), block(RR_BLOCK);
        if (narrow_starts) {
            if (pos_in || gid_in) return ctx->fail(DK_E_INTERNAL, "rerank: narrow keys only in the first rerank");
            k_rerank_reduce<uint32_t><<<grid, block, 0, st>>>(reinterpret_cast<const uint32_t *>(keys), count, nullptr, agg, 0, flags, narrow_starts,
                                                              static_cast<uint32_t>(ntiles), per_wg);
        } else {
            k_rerank_reduce<uint64_t><<<grid, block, 0, st>>>(keys, count, gid_in, agg, pos_in ? 0 : fb.cmp_shift, flags, nullptr, static_cast<uint32_t>(ntiles), per_wg);
        }
    }
    {
        LaunchScope ls(ctx, K_RERANK_SCAN, 32.0 * ntiles);
        if (ntiles <= 4 * RSC_CHUNK) {
            k_rerank_scan<<<dim3(1), dim3(1024), 0, st>>>(agg, ntiles, ctx->d_mail, gstart);
        } else {  // phase B is the one-workgroup scan itself, over the chunk aggregates (totals -> mail, sentinel -> gstart)
            const size_t nchunks = div_up(ntiles, RSC_CHUNK);
            RerankAgg *chunk = ctx->ws_alloc<RerankAgg>(nchunks);
            if (!chunk) return DK_E_NOMEM;
            k_rerank_scan_a<<<dim3(nchunks), dim3(256), 0, st>>>(agg, ntiles, chunk);
            k_rerank_scan<<<dim3(1), dim3(1024), 0, st>>>(chunk, nchunks, ctx->d_mail, gstart);
            k_rerank_scan_c<<<dim3(nchunks), dim3(256), 0, st>>>(agg, ntiles, chunk);
        }
    }
    {
        // flags 1 + suffix 4 (+ position 4) + symbol 1 in; SA 4 + L 1 for finals or 13 compacted out (FIRST: finals already stand in SA / L)
        LaunchScope ls(ctx, K_RERANK_APPLY, (pos_in ? 10.0 + 9.0 : 6.0 + 13.0) * count);
        if (pos_in) {
            k_rerank_apply<<<dim3(ntiles), dim3(RR_BLOCK), 0, st>>>(flags, idx, pos_in, count, agg, rank, sa, out_idx, out_pos, out_gid, gstart, fb);
        } else {
            if (rank) return ctx->fail(DK_E_INTERNAL, "rerank: the first rerank writes no ranks");
            const uint32_t per_wg = static_cast<uint32_t>(std::min<size_t>(std::max<size_t>(ntiles / 8192, 1), RA_FIRST_TILES));
            k_rerank_apply_first<<<dim3(div_up(ntiles, per_wg)), dim3(RR_BLOCK), 0, st>>>(flags, idx, count, agg, out_idx, out_pos, out_gid, gstart, ctx->d_mail, fb,
                                                                                          static_cast<uint32_t>(ntiles), per_wg, static_cast<uint32_t>(DK_KNOB("DK_RA_SPARSE", RA_SPARSE)));
        }
    }
    DK_HIP(ctx, hipGetLastError());
    ctx->ws_release(mark);
    return DK_OK;
}

// ---- big / small classification of the groups of the next round ---------------------------------------------------
// A group of more than LS_MAX members goes through the global radix sort, in the BIG LIST: the slots of all big groups, group after
// group.  bigidx[g] = number of big groups before group g (its dense index j), bigoff[j] = number of slots in big groups before it
// (its offset in the big list); totals -> mail[2] (slots) and mail[4] (groups).  The global sort's key carries j, not the offset,
// above the secondary key: j needs log2(#big groups) bits where the offset needs log2(#slots) -- two giant groups (periodic text)
// cost ONE extra bit instead of 24, i.e. three radix passes less per round.  mail[3] = number of slots in groups of more than
// PL_MAX members (those keep the sort in general rounds; zero = the in-place rounds can take over).
constexpr int LS_MAX = 1024;  // general rounds: groups up to this size are sorted inside a workgroup's LDS
constexpr int PL_BITS = 8;
constexpr int PL_MAX = 1 << PL_BITS;  // in-place (plateau) rounds need every group to have at most this many members (PL_BITS-bit offsets and
                                      // sizes in the slot's meta word).  2048 was measured in round 3: the 1e8 text block then skips its one
                                      // general round (a group of 1336 and 23 076 slots in groups above 256 are all that keeps it), but the
                                      // in-place rounds start one round earlier on 9.9 instead of 9.2 M slots: 11.63 against 11.55-11.66 ms
                                      // (and again with the pair chains in front of them: 11.8 against 10.7 ms -- the chains settle fewer
                                      // pairs at the shallower depth and the 2048-slot halos cost the sort kernel its occupancy)
constexpr int BG_IPT = 16;
constexpr int BG_TILE = 256 * BG_IPT;

__device__ __forceinline__ uint32_t big_size(const uint32_t *__restrict__ gstart, size_t g, size_t groups, uint32_t ls_max) {
    if (g >= groups) return 0;
    const uint32_t sz = gstart[g + 1] - gstart[g];
    return sz > ls_max ? sz : 0u;
}
// `groups` is read from mail[1] on the device: the host does not know it yet when these kernels are enqueued.  The grid is fixed
// (BG_GRID workgroups); every workgroup takes a contiguous stretch of tiles of groups, so that `part` has BG_GRID entries whatever n.
constexpr int BG_GRID = 1024;
__device__ __forceinline__ void big_stretch(size_t groups, size_t *t0, size_t *t1) {
    const size_t ntiles = (groups + BG_TILE - 1) / BG_TILE;
    const size_t per = (ntiles + BG_GRID - 1) / BG_GRID;
    *t0 = static_cast<size_t>(blockIdx.x) * per;
    *t1 = *t0 + per < ntiles ? *t0 + per : ntiles;
}
__global__ __launch_bounds__(256) void k_big_reduce(const uint32_t *__restrict__ gstart, const uint32_t *__restrict__ mail,
                                                     uint2 *__restrict__ part, uint32_t ls_max) {
    __shared__ uint32_t s_w[2][RR_WAVES];
    const size_t groups = mail[1];
    size_t t0, t1;
    big_stretch(groups, &t0, &t1);
    uint32_t sum = 0, cnt = 0, medium = 0;
    for (size_t t = t0; t < t1; ++t) {
        const size_t g0 = t * BG_TILE + static_cast<size_t>(threadIdx.x) * BG_IPT;
        if (g0 >= groups) continue;
        uint32_t prev = gstart[g0];
#pragma unroll
        for (int j = 0; j < BG_IPT && g0 + j < groups; ++j) {
            const uint32_t next = gstart[g0 + j + 1];
            const uint32_t sz = next - prev;
            prev = next;
            sum += sz > ls_max ? sz : 0u;
            cnt += sz > ls_max ? 1u : 0u;
            medium += sz > static_cast<uint32_t>(PL_MAX) ? sz : 0u;
        }
    }
    sum = wave_sum(sum);
    cnt = wave_sum(cnt);
    medium = wave_sum(medium);
    if ((threadIdx.x & 63) == 0) { s_w[0][threadIdx.x >> 6] = sum; s_w[1][threadIdx.x >> 6] = cnt; }
    if ((threadIdx.x & 63) == 0 && medium) atomicAdd(const_cast<uint32_t *>(mail) + 3, medium);
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = make_uint2(s_w[0][0] + s_w[0][1] + s_w[0][2] + s_w[0][3], s_w[1][0] + s_w[1][1] + s_w[1][2] + s_w[1][3]);
}
__global__ __launch_bounds__(BG_GRID) void k_big_spine(uint2 *__restrict__ part, uint32_t *__restrict__ mail) {
    __shared__ uint32_t s_tmp[16 + 1];
    const uint2 v = part[threadIdx.x];
    uint32_t total, total_cnt;
    const uint32_t run = block_excl_sum<BG_GRID / 64>(v.x, s_tmp, &total);
    const uint32_t run_cnt = block_excl_sum<BG_GRID / 64>(v.y, s_tmp, &total_cnt);
    part[threadIdx.x] = make_uint2(run, run_cnt);
    if (threadIdx.x == 0) { mail[2] = total; mail[4] = total_cnt; }
}
__global__ __launch_bounds__(256) void k_big_apply(const uint32_t *__restrict__ gstart, const uint32_t *__restrict__ mail,
                                                    const uint2 *__restrict__ part, uint32_t *__restrict__ bigidx, uint32_t *__restrict__ bigoff,
                                                    uint32_t ls_max) {
    __shared__ uint32_t s_tmp[RR_WAVES + 1];
    const size_t groups = mail[1];
    size_t t0, t1;
    big_stretch(groups, &t0, &t1);
    uint2 base = part[blockIdx.x];
    if (base.x == (blockIdx.x + 1 < gridDim.x ? part[blockIdx.x + 1].x : mail[2])) return;  // no big group in this stretch
    for (size_t t = t0; t < t1; ++t) {
        const size_t g0 = t * BG_TILE + static_cast<size_t>(threadIdx.x) * BG_IPT;
        uint32_t v[BG_IPT], sum = 0, cnt = 0;
#pragma unroll
        for (int j = 0; j < BG_IPT; ++j) { v[j] = big_size(gstart, g0 + j, groups, ls_max); sum += v[j]; cnt += v[j] ? 1u : 0u; }
        uint32_t tile_sum, tile_cnt;
        uint32_t run = base.x + block_excl_sum<RR_WAVES>(sum, s_tmp, &tile_sum);
        uint32_t idx = base.y + block_excl_sum<RR_WAVES>(cnt, s_tmp, &tile_cnt);
#pragma unroll
        for (int j = 0; j < BG_IPT; ++j) {
            if (g0 + j < groups && v[j]) {  // read for big groups only (k_round_local)
                bigidx[g0 + j] = idx;
                bigoff[idx] = run;
                ++idx;
            }
            run += v[j];
        }
        base.x += tile_sum;
        base.y += tile_cnt;
    }
}

// ---- one doubling round for the small groups, entirely inside a workgroup -------------------------------------------
// Tile of LS_TILE slots (+ LS_MAX halo slots on both sides, so that every small group touching the tile is visible):
// gather rank2 for every slot into LDS, then each slot counts the members of its group that sort before it (O(group size)
// LDS reads) and writes itself to its place inside the group's slot range.  Slots of big groups are copied, in slot order,
// to the big list that the global radix sort handles.
constexpr int LS_BLOCK = 256;
constexpr int LS_IPT = 8;
constexpr int LS_TILE = LS_BLOCK * LS_IPT;

__device__ __forceinline__ uint32_t rank2_of(const uint32_t *__restrict__ rank, uint32_t suffix, uint32_t n, uint32_t h) {
    // rank of the suffix h symbols further on, shifted up by h; suffixes that end before that get n-1-i (< h), which
    // orders them among themselves by "shorter first" and in front of every suffix that continues
    const uint64_t p = static_cast<uint64_t>(suffix) + h;
    return p < n ? rank[p] + h : static_cast<uint32_t>(static_cast<uint64_t>(n) + h - 1 - p);
}

// TEXT = true (text-extension round of the short-prefix path): the secondary key is not a rank but the next `tsym` symbols of the
// text after the h already sorted ones, packed like the initial keys (64-bit secondary keys in LDS).
// bits == 0: RAW mode -- the key is the next eight BYTES of the text, big-endian (the byte order is the symbol order, so no code table
// is needed; one or two aligned 8-byte loads per suffix).  Otherwise tsym symbols through the code table (small alphabets: more
// symbols per 64 bits), the table staged in LDS.
// next_break != nullptr (raw mode only): the period round -- a suffix that follows period `period` through the depth reached so far takes the token
// of its stretch's end as secondary key instead of the next eight bytes (k_period_fill; ebits = bits that hold any length up to n)
struct TextSource { const uint8_t *text; const uint8_t *code; int bits; int tsym; const uint32_t *next_break; int period; int ebits; };

// bytes T[p .. p+8) as a big-endian integer, zero padded past the end of the text.  One 8-byte load at the byte address itself (the
// hardware takes unaligned global addresses; two aligned loads and a funnel shift were twice the requests for the same line).
__device__ __forceinline__ uint64_t text_key_raw(const uint8_t *__restrict__ t, size_t n, size_t p) {
    if (p + 8 <= n) {
        typedef uint64_t __attribute__((aligned(1))) unaligned_u64;
        typedef const unaligned_u64 __attribute__((address_space(1))) *global_ptr;  // (a pointer out of a kernel-argument struct would go the flat way)
        return __builtin_bswap64(*(global_ptr)(t + p));
    }
    uint64_t key = 0;
    for (int j = 0; j < 8; ++j) key = (key << 8) | (p + j < n ? t[p + j] : 0u);
    return key;
}
// tsym symbols from p on through the (LDS) code table, packed big-endian, zero padded past the end
__device__ __forceinline__ uint64_t text_key_coded(const uint8_t *__restrict__ t, size_t n, const uint8_t *s_code, size_t p, int bits, int count) {
    uint64_t key = 0;
    int j = 0;
    if (p + static_cast<size_t>(count) + 8 <= n) {
        const uintptr_t addr = reinterpret_cast<uintptr_t>(t) + p;
        typedef const uint64_t __attribute__((address_space(1))) *global_words;  // (an address made of an integer would go the flat way)
        global_words q = (global_words)(addr & ~static_cast<uintptr_t>(7));
        int skip = static_cast<int>(addr & 7u);
        while (j < count) {
            uint64_t w = *q++ >> (8 * skip);
            for (int b = skip; b < 8 && j < count; ++b, ++j) {
                key = (key << bits) | s_code[w & 0xFFu];
                w >>= 8;
            }
            skip = 0;
        }
        return key;
    }
    for (; j < count; ++j) {
        const size_t q = p + j;
        key = (key << bits) | (q < n ? s_code[t[q]] : 0u);
    }
    return key;
}

// Small groups: key_out = the secondary key alone (the rerank compares keys inside an old group only).  Big groups: the global sort
// must keep every group in its range of the big list, so its key is (dense index of the group among the big groups) above the
// secondary key, cut to its leading `kb` bits when both do not fit 64 bits (text rounds: fewer symbols for the big groups); bslot
// remembers which slot every position of the big list came from.
template <bool TEXT>
__global__ __launch_bounds__(LS_BLOCK) void k_round_local(const uint32_t *__restrict__ act_idx, const uint32_t *__restrict__ act_gid,
                                                          const uint32_t *__restrict__ gstart, const uint32_t *__restrict__ bigidx,
                                                          const uint32_t *__restrict__ bigoff,
                                                          const uint32_t *__restrict__ rank, uint32_t n, uint32_t h, int kbits, int kb, int big_carry,
                                                          size_t count, uint64_t *__restrict__ key_out, uint32_t *__restrict__ idx_out,
                                                          uint64_t *__restrict__ bkeys, uint32_t *__restrict__ bidx,
                                                          uint32_t *__restrict__ bslot, TextSource ts,
                                                          const uint8_t *__restrict__ sym_in, uint8_t *__restrict__ sym_out) {
    using R2 = typename std::conditional<TEXT, uint64_t, uint32_t>::type;
    __shared__ R2 s_r2[LS_TILE + 2 * LS_MAX];
    __shared__ uint8_t s_code[TEXT ? 256 : 4];
    const int tid = threadIdx.x;
    if (TEXT && ts.bits) {
        s_code[tid] = ts.code[tid];
        __syncthreads();
    }
    auto second = [&](uint32_t suffix) -> R2 {
        if (TEXT) {
            const size_t p = static_cast<size_t>(suffix) + h;
            if (ts.next_break && p <= n) {
                // all members of a group agree on this test (it reads their first h symbols only); a suffix shorter than h keeps the text key
                // (zero: it ends first) and so sorts in front of its group, where a proper prefix of the others belongs
                const uint32_t b = ts.next_break[suffix], e = b - suffix;  // T[suffix + k] == T[suffix + k + period] for k < e
                if (e + static_cast<uint32_t>(ts.period) >= h) {
                    const size_t brk = static_cast<size_t>(b) + ts.period;  // the first symbol that does not follow the period (<= n)
                    const bool down = brk >= n || ts.text[brk] < ts.text[b];
                    const uint64_t ev = down ? e : ((1u << ts.ebits) - 1u - e);
                    const uint64_t tail = text_key_raw(ts.text, n, brk) >> 32;  // four bytes from the break on
                    return static_cast<R2>((static_cast<uint64_t>(down ? 0u : 1u) << 63) | (ev << (63 - ts.ebits)) | (tail << (31 - ts.ebits)));
                }
            }
            return static_cast<R2>(ts.bits ? text_key_coded(ts.text, n, s_code, p, ts.bits, ts.tsym) : text_key_raw(ts.text, n, p));
        }
        return static_cast<R2>(rank2_of(rank, suffix, n, h));
    };
    const size_t b0 = static_cast<size_t>(blockIdx.x) * LS_TILE;
    uint32_t my_idx[LS_IPT];
    R2 my_r2[LS_IPT];
    // everything a slot needs from global memory is asked for up front (suffix, group and its extent, symbol in front): the loop after
    // the barrier then runs on registers and LDS alone -- a load there would wait behind the stores of the slots before it
    uint32_t my_gs[LS_IPT], my_ge[LS_IPT];
    uint8_t my_sym[LS_IPT];
#pragma unroll
    for (int k = 0; k < LS_IPT; ++k) {
        const size_t a = b0 + static_cast<size_t>(k) * LS_BLOCK + tid;
        my_gs[k] = my_ge[k] = 0;
        my_sym[k] = 0;
        if (a < count) {
            my_idx[k] = act_idx[a];
            const uint32_t g = act_gid[a];
            my_gs[k] = gstart[g];
            my_ge[k] = gstart[g + 1];
            if (sym_in) my_sym[k] = sym_in[a];
        }
    }
#pragma unroll
    for (int k = 0; k < LS_IPT; ++k) {
        const size_t a = b0 + static_cast<size_t>(k) * LS_BLOCK + tid;
        if (a < count) {
            my_r2[k] = second(my_idx[k]);
            s_r2[LS_MAX + k * LS_BLOCK + tid] = my_r2[k];
        }
    }
    // halo: the slots before / after the tile that belong to the group of the tile's first / last slot (nothing else in the halo is ever
    // looked at, and a group above LS_MAX goes through the big list: no halo).  Their number follows from the group's extent, so a halo
    // slot costs its gather only when a group really straddles the tile edge -- and nothing at all otherwise.
    {
        const size_t last = (b0 + LS_TILE <= count ? b0 + LS_TILE : count) - 1;
        const uint32_t g_first = act_gid[b0], g_last = act_gid[last];
        const uint32_t fs = gstart[g_first], fe = gstart[g_first + 1], ls = gstart[g_last], le = gstart[g_last + 1];
        const uint32_t need_left = fe - fs <= static_cast<uint32_t>(LS_MAX) ? static_cast<uint32_t>(b0) - fs : 0u;        // slots fs .. b0 - 1
        const uint32_t need_right = le - ls <= static_cast<uint32_t>(LS_MAX) ? le - static_cast<uint32_t>(last) - 1u : 0u;  // slots last + 1 .. le - 1
        for (uint32_t t = tid; t < need_left; t += LS_BLOCK) {
            const size_t a = b0 - 1 - t;
            s_r2[LS_MAX - 1 - t] = second(act_idx[a]);
        }
        for (uint32_t t = tid; t < need_right; t += LS_BLOCK) {
            const size_t a = last + 1 + t;
            s_r2[LS_MAX + (last + 1 - b0) + t] = second(act_idx[a]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < LS_IPT; ++k) {
        const size_t a = b0 + static_cast<size_t>(k) * LS_BLOCK + tid;
        if (a >= count) continue;
        const uint32_t gs = my_gs[k], ge = my_ge[k];
        if (ge - gs <= static_cast<uint32_t>(LS_MAX)) {
            const R2 mine = my_r2[k];
            // LDS index of slot b is b - (b0 - LS_MAX) = b + LS_MAX - b0 (never negative: gs >= a - LS_MAX + 1)
            const uint32_t base = static_cast<uint32_t>(LS_MAX) - static_cast<uint32_t>(b0);  // mod 2^32 arithmetic
            uint32_t before = 0;
            for (uint32_t b = gs; b < ge; ++b) {
                const R2 v = s_r2[b + base];
                before += (v < mine) || (v == mine && b < static_cast<uint32_t>(a));
            }
            key_out[gs + before] = static_cast<uint64_t>(mine);
            idx_out[gs + before] = my_idx[k];
            if (sym_in) sym_out[gs + before] = my_sym[k];  // the symbol in front of the suffix travels with it (BwtCarry)
        } else {
            const uint32_t bj = bigidx[act_gid[a]];
            const uint32_t bo = bigoff[bj] + (static_cast<uint32_t>(a) - gs);
            const uint64_t bkey = (static_cast<uint64_t>(bj) << kb) | (static_cast<uint64_t>(my_r2[k]) >> (kbits - kb));
            // big_carry: the symbol in front of the suffix rides below the sorted bits, like in the initial sort (BwtCarry)
            bkeys[bo] = big_carry ? (bkey << 8) | my_sym[k] : bkey;
            bidx[bo] = my_idx[k];
            bslot[bo] = static_cast<uint32_t>(a);
        }
    }
}

// sorted big list -> back into the slots of the big groups: a sorted element stays inside its group's range of the big list, and
// position q of that list came from slot bslot[q].  sym_out (BwtCarry): the symbol in front of the suffix comes back in the key's low
// byte (big_carry: it rode below the sorted bits) -- or, when the key had no room for it, from the text again: one random byte per
// member, which on word-like text (half of all suffixes in big groups after the initial sort) was a millisecond per 1e8 bytes.
__global__ __launch_bounds__(256) void k_big_back(const uint64_t *__restrict__ bkeys, const uint32_t *__restrict__ bidx,
                                                   const uint32_t *__restrict__ bslot, size_t nbig, uint64_t *__restrict__ key_out,
                                                   uint32_t *__restrict__ idx_out, const uint8_t *__restrict__ text, uint32_t n,
                                                   uint8_t *__restrict__ sym_out, int big_carry) {
    const size_t bo = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (bo >= nbig) return;
    const uint32_t a = bslot[bo];
    const uint32_t suffix = bidx[bo];
    const uint64_t k = bkeys[bo];
    key_out[a] = big_carry ? k >> 8 : k;
    idx_out[a] = suffix;
    if (sym_out) sym_out[a] = big_carry ? static_cast<uint8_t>(k) : text[suffix ? suffix - 1 : n - 1];
}

// ---- plateau rounds: every group has at most PL_MAX members ---------------------------------------------------------------------
// Once a round ends without a big group there will never be one again (groups only split), and what is left are typically the
// suffixes inside long repeats: the list shrinks slowly for log2(repeat length) rounds.  These rounds run on an IN-PLACE list:
//   idx[a]   suffix in slot a; bit 31 set = the slot is dead (its suffix is final)
//   meta[a]  offset of the slot inside its group (PL_BITS bits) | (group size - 1) << PL_BITS | PL_MOVED
//   pos[a]   SA position of slot a (never changes: members move only inside their group's slot range)
// One kernel per round sorts every group inside LDS by the rank of the suffix h further on and writes each member to its place
// (ping-pong idx / meta / sym): its new group is the run of members with the same secondary rank.  No compaction, no global scan,
// no host round trip: the number of live slots comes back one round late.  Ranks are updated by a second, tiny kernel: updating
// them inside the first would let another workgroup see the new rank of one suffix and the old rank of its group mate -- an order
// that may contradict both the h-order and the 2h-order.
constexpr uint32_t PL_DEAD = 0xFFFFFFFFu, PL_DEAD_BIT = 0x80000000u;
constexpr uint32_t PL_MOVED = 1u << (2 * PL_BITS), PL_OFF_MASK = (1u << PL_BITS) - 1u;  // the slot's group got a new head this round: its members' ranks change

__global__ __launch_bounds__(256) void k_to_inplace(const uint32_t *__restrict__ gid, const uint32_t *__restrict__ gstart, size_t count,
                                                     uint32_t *__restrict__ meta) {
    const size_t a = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (a >= count) return;
    const uint32_t g = gid[a];
    const uint32_t gs = gstart[g], ge = gstart[g + 1];
    meta[a] = (static_cast<uint32_t>(a) - gs) | ((ge - gs - 1u) << PL_BITS);
}

__global__ __launch_bounds__(LS_BLOCK) void k_plateau_sort(const uint32_t *__restrict__ idx_in, const uint32_t *__restrict__ meta_in,
                                                           const uint8_t *__restrict__ sym_in, const uint32_t *__restrict__ pos,
                                                           const uint32_t *__restrict__ rank, uint32_t n, uint32_t h, size_t slots,
                                                           uint32_t *__restrict__ idx_out, uint32_t *__restrict__ meta_out,
                                                           uint8_t *__restrict__ sym_out, uint32_t *__restrict__ sa,
                                                           uint8_t *__restrict__ bwt, uint32_t *__restrict__ origin,
                                                           uint32_t *__restrict__ live, const uint32_t *__restrict__ prev_live) {
    __shared__ uint32_t s_r2[LS_TILE + 2 * PL_MAX];
    __shared__ uint32_t s_cnt[RR_WAVES];
    // launched one round ahead of the host: when the round before left nothing alive there is nothing to do (and nothing to copy:
    // nobody reads the lists again)
    if (prev_live && *prev_live == 0) return;
    const int tid = threadIdx.x;
    const size_t b0 = static_cast<size_t>(blockIdx.x) * LS_TILE;
    uint32_t my_idx[LS_IPT], my_r2[LS_IPT], my_meta[LS_IPT];
    uint8_t my_sym[LS_IPT];
#pragma unroll
    for (int k = 0; k < LS_IPT; ++k) {  // (all the loads of a slot up front: see k_round_local)
        const size_t a = b0 + static_cast<size_t>(k) * LS_BLOCK + tid;
        my_idx[k] = a < slots ? idx_in[a] : PL_DEAD;
        my_meta[k] = (my_idx[k] & PL_DEAD_BIT) ? 0u : meta_in[a];
        my_sym[k] = (sym_in && !(my_idx[k] & PL_DEAD_BIT)) ? sym_in[a] : uint8_t(0);
    }
#pragma unroll
    for (int k = 0; k < LS_IPT; ++k) {
        my_r2[k] = (my_idx[k] & PL_DEAD_BIT) ? 0u : rank2_of(rank, my_idx[k], n, h);
        s_r2[PL_MAX + k * LS_BLOCK + tid] = my_r2[k];
    }
    // halo: only the slots of the groups that straddle the tile's edges (the group of the tile's first live slot reaches back by its
    // offset, the group of its last live slot forward by size - 1 - offset); nothing else in the halo is ever looked at
    {
        const size_t last = (b0 + LS_TILE <= slots ? b0 + LS_TILE : slots) - 1;
        const uint32_t m_first = (idx_in[b0] & PL_DEAD_BIT) ? 0u : meta_in[b0];
        const uint32_t m_last = (idx_in[last] & PL_DEAD_BIT) ? 0u : meta_in[last];
        const uint32_t need_left = m_first & PL_OFF_MASK;
        const uint32_t need_right = (idx_in[last] & PL_DEAD_BIT) ? 0u : ((m_last >> PL_BITS) & PL_OFF_MASK) - (m_last & PL_OFF_MASK);
        for (uint32_t t = tid; t < need_left; t += LS_BLOCK) {  // slots b0 - need_left .. b0 - 1 (inside the list: the group starts there)
            const uint32_t v = idx_in[b0 - 1 - t];
            s_r2[PL_MAX - 1 - t] = (v & PL_DEAD_BIT) ? 0u : rank2_of(rank, v, n, h);
        }
        for (uint32_t t = tid; t < need_right; t += LS_BLOCK) {  // slots last + 1 .. last + need_right
            const uint32_t v = idx_in[last + 1 + t];
            s_r2[PL_MAX + (last + 1 - b0) + t] = (v & PL_DEAD_BIT) ? 0u : rank2_of(rank, v, n, h);
        }
    }
    __syncthreads();
    // three sweeps over the thread's slots: places and new groups from LDS alone; then the SA positions of the slots that became final
    // (one load each, all in flight together); then the stores.  A load between the stores would wait for every store before it.
    uint32_t alive = 0;
    uint32_t dest[LS_IPT], meta_new[LS_IPT], fin_pos[LS_IPT];
#pragma unroll
    for (int k = 0; k < LS_IPT; ++k) {
        const size_t a = b0 + static_cast<size_t>(k) * LS_BLOCK + tid;
        dest[k] = ~0u;  // dead slot (or past the list)
        meta_new[k] = 0;
        if (a >= slots || (my_idx[k] & PL_DEAD_BIT)) continue;
        const uint32_t m = my_meta[k];
        const uint32_t gs = static_cast<uint32_t>(a) - (m & PL_OFF_MASK), ge = gs + ((m >> PL_BITS) & PL_OFF_MASK) + 1u;
        const uint32_t mine = my_r2[k];
        const uint32_t base = static_cast<uint32_t>(PL_MAX) - static_cast<uint32_t>(b0);  // LDS index of slot b = b + base (mod 2^32)
        uint32_t less = 0, eq_before = 0, eq = 0;
        for (uint32_t b = gs; b < ge; ++b) {
            const uint32_t v = s_r2[b + base];
            less += v < mine;
            eq += v == mine;
            eq_before += (v == mine) && b < static_cast<uint32_t>(a);
        }
        dest[k] = gs + less + eq_before;
        const uint32_t moved = less ? PL_MOVED : 0u;  // new head slot gs + less: the rank becomes pos[gs + less]
        meta_new[k] = eq_before | ((eq - 1u) << PL_BITS) | moved;
    }
#pragma unroll
    for (int k = 0; k < LS_IPT; ++k) {
        const bool final_now = dest[k] != ~0u && (meta_new[k] >> PL_BITS & PL_OFF_MASK) == 0;  // alone in its new group
        fin_pos[k] = final_now ? pos[dest[k]] : ~0u;
    }
#pragma unroll
    for (int k = 0; k < LS_IPT; ++k) {
        const size_t a = b0 + static_cast<size_t>(k) * LS_BLOCK + tid;
        if (a >= slots) continue;
        if (my_idx[k] & PL_DEAD_BIT) { idx_out[a] = PL_DEAD; continue; }  // a dead slot is never inside a live group's range
        meta_out[dest[k]] = meta_new[k];
        if (fin_pos[k] != ~0u) {  // final
            const uint32_t p = fin_pos[k];
            if (sa) sa[p] = my_idx[k];
            if (sym_in) bwt[p] = my_sym[k];
            if (origin && my_idx[k] == 0) *origin = p;
            idx_out[dest[k]] = my_idx[k] | PL_DEAD_BIT;  // k_plateau_ranks still needs the suffix; dead for every later round
        } else {
            idx_out[dest[k]] = my_idx[k];
            if (sym_in) sym_out[dest[k]] = my_sym[k];
            ++alive;
        }
    }
    alive = wave_sum(alive);
    if ((tid & 63) == 0) s_cnt[tid >> 6] = alive;
    __syncthreads();
    if (tid == 0) {
        const uint32_t t = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        if (t) atomicAdd(live, t);
    }
}

// the rank updates of the round k_plateau_sort just ran: members of a group with a new head get that head's SA position
__global__ __launch_bounds__(256) void k_plateau_ranks(const uint32_t *__restrict__ idx, const uint32_t *__restrict__ meta,
                                                        const uint32_t *__restrict__ pos, size_t slots, uint32_t *__restrict__ rank,
                                                        const uint32_t *__restrict__ prev_live) {
    if (prev_live && *prev_live == 0) return;  // the sort kernel of this round did nothing
    const size_t a = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (a >= slots) return;
    const uint32_t v = idx[a];
    if (v == PL_DEAD) return;
    const uint32_t m = meta[a];
    if (m & PL_MOVED) rank[v & ~PL_DEAD_BIT] = pos[a - (m & PL_OFF_MASK)];
}

// ---- pair chains: what induced sorting knows and prefix doubling does not ---------------------------------------------------------------
// A long repeat (two copies of a passage) leaves thousands of groups of TWO suffixes (a + k, a + delta + k), k = 0, 1, 2 ...: doubling
// resolves the pair with the shortest common prefix first and needs log2(length) rounds of rank gathers for the rest.  But two suffixes
// that start with the same symbol are ordered like the suffixes one position further on: every pair of the chain is ordered like the pair
// behind the chain's END -- (a + K + 1, a + delta + K + 1), which is no pair any more, so its two ranks differ and decide.  One
// comparison per chain instead of log2(length) gathers per pair.  Copies of copies (two identical halves of a text that repeats itself
// inside, three copies of a passage) leave groups of three and four: EVERY pair inside a group of at most CH_K members is a record,
// each with its own delta, and a group is settled once all of its pairs are.
//   k_chain_extract  every pair inside a live group of 2 .. CH_K members -> a record (delta ‖ lower position; slot of the pair's first
//                    member; how many slots further on the second one stands, in the key's two top bits), appended in any order
//   sort_pairs       records by (delta, lower position): the pairs of a chain become neighbours
//   k_chain_ends     record r continues into r + 1 when that is the pair one position further on with the same delta.  Otherwise the
//                    two ranks behind it decide.  Equal ranks there: the two suffixes sit in a bigger group together -- equal in h
//                    symbols, so the pair h positions further on decides, and so on: a bounded walk in strides of h that ends at ranks
//                    that differ, or beyond the next record of the same delta (then r continues into r + 1 after all: a chain runs
//                    THROUGH the bigger groups on its way), or undecided
//   k_chain_tiles / k_chain_spine / k_chain_verdicts   nearest chain end at or after every record (three-phase scan from the right)
//                    -> the verdict goes to the byte of the pair's first slot, in the plane of its distance
//   k_chain_apply    slots in order: the head of a group whose pairs are all decided puts the members in order, writes them to SA (and
//                    L), gives each its own rank and dies with them
// Groups of more than CH_K members, and chains that end undecided, go on into the in-place rounds.
constexpr uint8_t CH_END = 1, CH_LT = 2, CH_GT = 4;  // verdict byte: chain end | lower position is the smaller suffix | ... the larger
constexpr int CH_TILE = 2048;
constexpr int CH_K = 4;                   // groups of up to four members: six pairs, second member at most three slots behind the first
constexpr int CH_WALK_THROUGH = 1024;     // strides of h through a bigger group towards the next record of the same delta
constexpr int CH_WALK_FREE = 16;          // ... with no such record ahead
constexpr int CH_DIST_SHIFT = 62;         // key: (distance in slots - 1) << 62 | delta << lbits | lower position; lbits <= 31

// (a workgroup takes CH_EXTRACT x 256 slots and reserves its place in the record list with ONE global atomic: one per 256 slots --
// 36 000 atomics on one address for the 9.2 M slots of the 1e8 text block -- was 0.41 ms of a kernel that reads 73 MB)
constexpr int CH_EXTRACT = 16;
__device__ __forceinline__ uint32_t chain_records_of(uint32_t v, uint32_t m, int kmax) {  // how many pairs does this slot open?
    if (v & PL_DEAD_BIT) return 0u;
    const uint32_t off = m & PL_OFF_MASK, last = (m >> PL_BITS) & PL_OFF_MASK;  // last = group size - 1
    return (last >= 1u && last < static_cast<uint32_t>(kmax)) ? last - off : 0u;
}
__global__ __launch_bounds__(256) void k_chain_extract(const uint32_t *__restrict__ idx, const uint32_t *__restrict__ meta, size_t slots, int kmax, int lbits,
                                                        uint64_t *__restrict__ rec_key, uint32_t *__restrict__ rec_slot, uint32_t cap,
                                                        uint32_t *__restrict__ count) {  // count[0] records reserved, count[1] first reservation that did not fit
    __shared__ uint32_t s_n, s_base;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    uint32_t mine = 0;
#pragma unroll
    for (int c = 0; c < CH_EXTRACT; ++c) {
        const size_t a = (static_cast<size_t>(blockIdx.x) * CH_EXTRACT + c) * 256 + threadIdx.x;
        if (a < slots) mine += chain_records_of(idx[a], meta[a], kmax);
    }
    uint32_t at = mine ? atomicAdd(&s_n, mine) : 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
        s_base = s_n ? atomicAdd(count, s_n) : 0u;
        if (s_n && (s_base >= cap || s_n > cap - s_base)) {
            atomicMin(count + 1, s_base);
            s_n = 0;  // nothing of this workgroup is written: the list ends in front of it
        }
    }
    __syncthreads();
    if (!s_n || !mine) return;
    at += s_base;
#pragma unroll
    for (int c = 0; c < CH_EXTRACT; ++c) {
        const size_t a = (static_cast<size_t>(blockIdx.x) * CH_EXTRACT + c) * 256 + threadIdx.x;
        if (a >= slots) continue;
        const uint32_t v = idx[a];
        const uint32_t cnt = chain_records_of(v, meta[a], kmax);
        for (uint32_t d = 1; d <= cnt; ++d) {
            const uint32_t w = idx[a + d] & ~PL_DEAD_BIT;
            const uint32_t lo = v < w ? v : w, hi = v < w ? w : v;
            rec_key[at] = (static_cast<uint64_t>(d - 1u) << CH_DIST_SHIFT) | (static_cast<uint64_t>(hi - lo) << lbits) | lo;
            rec_slot[at] = static_cast<uint32_t>(a);
            ++at;
        }
    }
}

__global__ __launch_bounds__(256) void k_chain_ends(const uint64_t *__restrict__ rec_key, uint32_t m, int lbits, const uint32_t *__restrict__ rank, uint32_t n,
                                                     uint32_t h, uint8_t *__restrict__ rec_verdict) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    const uint64_t mask = (uint64_t(1) << (2 * lbits)) - 1u;
    const uint64_t k = rec_key[r] & mask;
    const uint32_t lo = static_cast<uint32_t>(k & ((uint64_t(1) << lbits) - 1u)), delta = static_cast<uint32_t>(k >> lbits);
    bool cont = false;
    uint32_t gap = 0;  // the next record of this delta stands this many positions further on (0: there is none)
    if (r + 1 < m) {
        const uint64_t k1 = rec_key[r + 1] & mask;
        cont = k1 == k + 1u;
        if (!cont && static_cast<uint32_t>(k1 >> lbits) == delta) gap = static_cast<uint32_t>(k1 - k);
    }
    uint8_t v = 0;
    if (!cont) {
        // the pair one position further on is no record: its ranks decide; while they are equal (one bigger group holds both: h symbols in common) look h further on
        const int max_steps = gap ? CH_WALK_THROUGH : CH_WALK_FREE;
        uint64_t off = 1;
        v = CH_END;
        for (int step = 0; step <= max_steps; ++step) {
            const uint32_t o = static_cast<uint32_t>(off);
            const uint32_t r_lo = rank2_of(rank, lo, n, o), r_hi = rank2_of(rank, lo + delta, n, o);
            if (r_lo != r_hi) {
                v = CH_END | (r_lo < r_hi ? CH_LT : CH_GT);
                break;
            }
            off += h;
            if (gap && off >= gap) {  // equal all the way to the next record of this delta: the chain goes on there
                v = 0;
                break;
            }
            if (off >= n) break;
        }
    }
    rec_verdict[r] = v;
}

// nearest chain end at or after a record = running maximum, from the right, of (index from the right + 1) over the end records
__global__ __launch_bounds__(256) void k_chain_tiles(const uint8_t *__restrict__ rec_verdict, uint32_t m, uint32_t *__restrict__ tile_max) {
    __shared__ uint32_t s_red[4];
    const uint32_t q0 = blockIdx.x * CH_TILE + threadIdx.x * 8;  // index from the right
    uint32_t best = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t q = q0 + j;
        if (q < m && (rec_verdict[m - 1 - q] & CH_END)) best = q + 1;
    }
    best = wave_max(best);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) tile_max[blockIdx.x] = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
}
__global__ __launch_bounds__(1024) void k_chain_spine(uint32_t *__restrict__ tile_max, uint32_t ntiles) {  // exclusive running maximum over the tiles
    __shared__ uint32_t s_tmp[16 + 1];
    const uint32_t per = (ntiles + 1023u) / 1024u;
    const uint32_t b0 = threadIdx.x * per, b1 = b0 + per < ntiles ? b0 + per : ntiles;
    uint32_t mx = 0;
    for (uint32_t b = b0; b < b1; ++b) mx = max(mx, tile_max[b]);
    uint32_t run = block_excl_max<16>(mx, s_tmp, nullptr);
    for (uint32_t b = b0; b < b1; ++b) {
        const uint32_t v = tile_max[b];
        tile_max[b] = run;
        run = max(run, v);
    }
}
__global__ __launch_bounds__(256) void k_chain_verdicts(const uint8_t *__restrict__ rec_verdict, const uint64_t *__restrict__ rec_key,
                                                         const uint32_t *__restrict__ rec_slot, uint32_t m, const uint32_t *__restrict__ tile_max,
                                                         uint8_t *__restrict__ planes, size_t slots) {
    __shared__ uint32_t s_tmp[4 + 1];
    const uint32_t q0 = blockIdx.x * CH_TILE + threadIdx.x * 8;
    uint32_t e[8], mine = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t q = q0 + j;
        e[j] = (q < m && (rec_verdict[m - 1 - q] & CH_END)) ? q + 1 : 0u;
        mine = max(mine, e[j]);
    }
    uint32_t run = max(block_excl_max<4>(mine, s_tmp, nullptr), tile_max[blockIdx.x]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t q = q0 + j;
        if (q >= m) break;
        run = max(run, e[j]);
        if (!run) continue;  // (no chain end at or after this record: the list was cut short in front of one)
        const uint8_t v = rec_verdict[m - run] & (CH_LT | CH_GT);
        if (v) planes[(rec_key[m - 1 - q] >> CH_DIST_SHIFT) * slots + rec_slot[m - 1 - q]] = v;
    }
}

// slots in order: the head of a group whose pairs are all decided settles its members.  planes[d - 1][a] = verdict of the pair (slot a, slot a + d).
__global__ __launch_bounds__(256) void k_chain_apply(uint32_t *__restrict__ idx, const uint32_t *__restrict__ meta, const uint8_t *__restrict__ sym,
                                                      const uint32_t *__restrict__ pos, const uint8_t *__restrict__ planes, size_t slots,
                                                      uint32_t *__restrict__ sa, uint8_t *__restrict__ bwt, uint32_t *__restrict__ origin,
                                                      uint32_t *__restrict__ rank) {
    const size_t a = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (a + 1 < slots && planes[a]) {  // (a pair of neighbouring slots has a verdict: the head of a group, or a member further in)
        const uint32_t mt = meta[a];
        const uint32_t g = ((mt >> PL_BITS) & PL_OFF_MASK) + 1u;
        if ((mt & PL_OFF_MASK) == 0 && g >= 2u && g <= static_cast<uint32_t>(CH_K) && !(idx[a] & PL_DEAD_BIT)) {
            uint32_t x[CH_K], before[CH_K];
#pragma unroll
            for (int i = 0; i < CH_K; ++i) {
                x[i] = static_cast<uint32_t>(i) < g ? idx[a + i] : 0u;
                before[i] = 0;
            }
            bool all = true;
#pragma unroll
            for (int i = 0; i < CH_K; ++i)
#pragma unroll
                for (int j = i + 1; j < CH_K; ++j)
                    if (static_cast<uint32_t>(j) < g) {
                        const uint8_t v = planes[static_cast<size_t>(j - i - 1) * slots + a + i];
                        all = all && v != 0;
                        const bool i_first = (x[i] < x[j]) == (v == CH_LT);  // slot i's suffix is the smaller one
                        before[i_first ? j : i] += 1u;
                    }
            if (all) {
#pragma unroll
                for (int i = 0; i < CH_K; ++i)
                    if (static_cast<uint32_t>(i) < g) {
                        const uint32_t p = pos[a + before[i]];
                        if (sa) sa[p] = x[i];
                        if (bwt) {
                            bwt[p] = sym[a + i];
                            if (x[i] == 0) *origin = p;
                        }
                        if (before[i]) rank[x[i]] = p;  // (the smallest keeps the group's rank: the position of its head slot)
                    }
#pragma unroll
                for (int i = 0; i < CH_K; ++i)
                    if (static_cast<uint32_t>(i) < g) idx[a + i] = PL_DEAD;
            }
        }
    }
}

// compaction of the in-place list (when most of its slots are dead): live slots keep their order, so groups stay contiguous
__global__ __launch_bounds__(RR_BLOCK) void k_plateau_count(const uint32_t *__restrict__ idx, size_t slots, uint32_t *__restrict__ tile_live) {
    __shared__ uint32_t s_cnt[RR_WAVES];
    const size_t b0 = static_cast<size_t>(blockIdx.x) * RR_TILE;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < RR_IPT; ++k) {
        const size_t a = b0 + static_cast<size_t>(k) * RR_BLOCK + threadIdx.x;
        c += (a < slots && !(idx[a] & PL_DEAD_BIT)) ? 1u : 0u;
    }
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_live[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}
__global__ __launch_bounds__(1024) void k_plateau_scan(uint32_t *__restrict__ tile_live, size_t ntiles, uint32_t *__restrict__ total_out) {
    __shared__ uint32_t s_tmp[16 + 1];
    const size_t per = (ntiles + 1023) / 1024;
    const size_t b0 = static_cast<size_t>(threadIdx.x) * per;
    const size_t b1 = b0 + per < ntiles ? b0 + per : ntiles;
    uint32_t sum = 0;
    for (size_t b = b0; b < b1; ++b) sum += tile_live[b];
    uint32_t total;
    uint32_t run = block_excl_sum<16>(sum, s_tmp, &total);
    for (size_t b = b0; b < b1; ++b) {
        const uint32_t v = tile_live[b];
        tile_live[b] = run;
        run += v;
    }
    if (threadIdx.x == 0) *total_out = total;
}
__global__ __launch_bounds__(RR_BLOCK) void k_plateau_compact(const uint32_t *__restrict__ idx, const uint32_t *__restrict__ meta,
                                                              const uint8_t *__restrict__ sym, const uint32_t *__restrict__ pos,
                                                              size_t slots, const uint32_t *__restrict__ tile_base,
                                                              uint32_t *__restrict__ idx_out, uint32_t *__restrict__ meta_out,
                                                              uint8_t *__restrict__ sym_out, uint32_t *__restrict__ pos_out) {
    __shared__ uint32_t s_tmp[RR_WAVES + 1];
    const size_t a0 = static_cast<size_t>(blockIdx.x) * RR_TILE + static_cast<size_t>(threadIdx.x) * RR_IPT;
    uint32_t v[RR_IPT], c = 0;
#pragma unroll
    for (int j = 0; j < RR_IPT; ++j) {
        v[j] = a0 + j < slots ? idx[a0 + j] : PL_DEAD;
        c += (v[j] & PL_DEAD_BIT) ? 0u : 1u;
    }
    uint32_t o = tile_base[blockIdx.x] + block_excl_sum<RR_WAVES>(c, s_tmp, nullptr);
#pragma unroll
    for (int j = 0; j < RR_IPT; ++j) {
        if (v[j] & PL_DEAD_BIT) continue;
        idx_out[o] = v[j];
        meta_out[o] = meta[a0 + j] & (PL_MOVED - 1u);
        pos_out[o] = pos[a0 + j];
        if (sym) sym_out[o] = sym[a0 + j];
        ++o;
    }
}

// enqueue the classification of the groups rerank() just produced and read back (active, groups, big slots, slots in groups > PL_MAX,
// big groups)
int classify_and_read(dk_ctx *ctx, size_t max_groups, uint32_t *gstart, uint32_t *bigidx, uint32_t *bigoff, size_t *active, size_t *groups,
                      size_t *nbig, size_t *nmedium, size_t *nbiggroups, uint32_t ls_max = LS_MAX) {
    hipStream_t st = ctx->stream;
    const size_t mark = ctx->ws_mark();
    (void)max_groups;
    uint2 *part = ctx->ws_alloc<uint2>(BG_GRID);
    if (!part) return DK_E_NOMEM;
    DK_HIP(ctx, hipMemsetAsync(ctx->d_mail + 3, 0, sizeof(uint32_t), st));
    {
        LaunchScope ls(ctx, K_BIG_CLASSIFY, 8.0 * max_groups);
        k_big_reduce<<<dim3(BG_GRID), dim3(256), 0, st>>>(gstart, ctx->d_mail, part, ls_max);
        k_big_spine<<<dim3(1), dim3(BG_GRID), 0, st>>>(part, ctx->d_mail);
        k_big_apply<<<dim3(BG_GRID), dim3(256), 0, st>>>(gstart, ctx->d_mail, part, bigidx, bigoff, ls_max);
    }
    DK_HIP(ctx, hipGetLastError());
    DK_HIP(ctx, hipMemcpyAsync(ctx->h_mail, ctx->d_mail, 5 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    DK_HIP(ctx, hipStreamSynchronize(st));
    *active = ctx->h_mail[0];
    *groups = ctx->h_mail[1];
    *nbig = ctx->h_mail[2];
    *nmedium = ctx->h_mail[3];
    *nbiggroups = ctx->h_mail[4];
    ctx->ws_release(mark);
    return DK_OK;
}

#include "lfirst.inc"

int suffix_array_impl(dk_ctx *ctx, const uint8_t *d_text, size_t n, uint32_t *d_sa, uint8_t *d_bwt, uint32_t *d_origin, bool *bwt_written, bool allow_lfirst);

}  // namespace

int suffix_array_device(dk_ctx *ctx, const uint8_t *d_text, size_t n, uint32_t *d_sa, uint8_t *d_bwt, uint32_t *d_origin, bool *bwt_written) {
    return suffix_array_impl(ctx, d_text, n, d_sa, d_bwt, d_origin, bwt_written, true);
}

namespace {

int suffix_array_impl(dk_ctx *ctx, const uint8_t *d_text, size_t n, uint32_t *d_sa, uint8_t *d_bwt, uint32_t *d_origin, bool *bwt_written, bool allow_lfirst) {
    if (bwt_written) *bwt_written = false;
    if (n == 0 || n > 0x7FFFFFFEull) return ctx->fail(DK_E_ARG, "suffix_array: n out of range");
    hipStream_t st = ctx->stream;
    ctx->stats.rounds = 0;
    ctx->stats.sort_passes = 0;
    ctx->stats.sorted_elements = 0;
    ctx->stats.sa_route = allow_lfirst ? 0u : static_cast<uint32_t>(DK_ROUTE_LFIRST_FALLBACK);
    uint32_t &route = ctx->stats.sa_route;
    const size_t mark = ctx->ws_mark();

    // 1. alphabet
    uint32_t *d_hist = ctx->d_mail + 16;
    DK_HIP(ctx, hipMemsetAsync(d_hist, 0, 266 * sizeof(uint32_t), st));  // (+ the run probe's two words and the period probe's eight behind the 256 counters)
    const int period_mode = DK_KNOB("DK_PERIOD", 1);  // 0 = never a period round, 1 = where the probe finds an eighth of the block periodic, 2 = wherever it finds a window (test hook)
    {
        LaunchScope ls(ctx, K_SYM_HIST, 1.0 * n);
        const size_t blocks = std::min<size_t>(div_up(n, 256 * 64), 2048);
        k_sym_hist<<<dim3(blocks), dim3(256), 0, st>>>(d_text, n, d_hist);
        if (d_bwt && allow_lfirst && n >= (1u << 16)) k_run_probe<<<dim3(div_up(div_up(n, 256), 256)), dim3(256), 0, st>>>(d_text, n, d_hist + 256);
        if (period_mode != 0 && n >= (1u << 12)) k_period_probe<<<dim3(div_up(div_up(n, 64), 256)), dim3(256), 0, st>>>(d_text, n, d_hist + 258);
    }
    DK_HIP(ctx, hipMemcpyAsync(ctx->h_mail + 16, d_hist, 266 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    DK_HIP(ctx, hipStreamSynchronize(st));
    // a run of 511 bytes, or more than 1 % of the block's 16-byte pieces inside runs: not the L-first path's kind of block
    const bool long_run = ctx->h_mail[16 + 256] != 0 || static_cast<double>(ctx->h_mail[16 + 257]) * 16.0 > 0.01 * static_cast<double>(n);
    // ... but a block with a few long runs and next to nothing else inside runs (a zero-padded header in front of text) stays with it: the run's
    // suffixes ride in the big list until the round stalls, and one token round (k_lf_tokens) places them by where the run ends
    const bool run_heavy = static_cast<double>(ctx->h_mail[16 + 257]) * 16.0 > 0.01 * static_cast<double>(n);
    uint8_t code[256];
    unsigned sigma = 0;
    for (int s = 0; s < 256; ++s) {
        code[s] = static_cast<uint8_t>(sigma);
        if (ctx->h_mail[16 + s]) ++sigma;
    }
    if (sigma <= 1) {  // one distinct symbol: suffixes sort by length
        k_sa_descending<<<dim3(div_up(n, 256)), dim3(256), 0, st>>>(d_sa, n);
        DK_HIP(ctx, hipGetLastError());
        return DK_OK;
    }
    const int bits = static_cast<int>(ceil_log2_u64(sigma));  // 1..8
    const int spk = 64 / bits;                                // symbols per key: 8 (bytes) .. 32 (ACGT) .. 64 (binary)

    uint64_t *keys = ctx->ws_alloc<uint64_t>(n), *keys_alt = ctx->ws_alloc<uint64_t>(n), *keys_3 = ctx->ws_alloc<uint64_t>(n);
    uint32_t *vals = ctx->ws_alloc<uint32_t>(n), *vals_alt = ctx->ws_alloc<uint32_t>(n);
    uint32_t *vals_3 = ctx->ws_alloc<uint32_t>(n);
    uint32_t *rank = ctx->ws_alloc<uint32_t>(n);
    uint32_t *pos = ctx->ws_alloc<uint32_t>(n), *pos_alt = ctx->ws_alloc<uint32_t>(n);
    uint32_t *gid = ctx->ws_alloc<uint32_t>(n), *gid_alt = ctx->ws_alloc<uint32_t>(n);
    uint32_t *gstart = ctx->ws_alloc<uint32_t>(n / 2 + 2), *bigidx = ctx->ws_alloc<uint32_t>(n / 2 + 2);
    uint32_t *bigoff = ctx->ws_alloc<uint32_t>(n / 32 + 2);  // one entry per big group (more than LS_MAX members each; the L-first path may draw the line at 32)
    uint8_t *d_code = reinterpret_cast<uint8_t *>(ctx->d_mail + 512);
    if (!keys || !keys_alt || !keys_3 || !vals || !vals_alt || !vals_3 || !rank || !pos || !pos_alt || !gid || !gid_alt ||
        !gstart || !bigidx || !bigoff)
        return DK_E_NOMEM;
    DK_HIP(ctx, hipMemcpyAsync(d_code, code, 256, hipMemcpyHostToDevice, st));
    const bool trace = DK_KNOB("DK_TRACE", 0) != 0;

    // 2. how long a prefix must the initial sort cover?  DK_PREFIX: 0 = always the full key, 1 = ask the sample (default),
    //    2 / 3 = always the shortest / second candidate (test hooks: every input then takes the short-prefix path; the second one is
    //    the five-pass sort whose narrow keys need the bucket starts)
    const int prefix_mode = DK_KNOB("DK_PREFIX", 1);
    int spk_sort = spk;
    double probe_big_share = 0.0;
    if (prefix_mode != 0 && (n >= (1u << 22) || prefix_mode >= 2)) {
        ProbeCands cands{0, {0, 0, 0, 0}};
        for (int passes = 4; passes <= 7 && cands.count < PP_MAX_CAND; ++passes) {
            const int k = (8 * passes) / bits;
            if (k >= 1 && k < spk && (cands.count == 0 || cands.sym[cands.count - 1] != k)) cands.sym[cands.count++] = k;
        }
        if (cands.count > 0 && prefix_mode >= 2 && prefix_mode <= 4) {  // (4: the third candidate, six passes -- an experiment hook)
            spk_sort = cands.sym[std::min(prefix_mode - 2, cands.count - 1)];
        } else if (cands.count > 0) {
            const uint32_t m = static_cast<uint32_t>(std::min<double>(n / 2.0, 10.0 * std::sqrt(static_cast<double>(n))));
            const uint32_t span = static_cast<uint32_t>(n / m);
            uint32_t table_size = 1;
            while (table_size < 2u * cands.count * m) table_size <<= 1;
            uint64_t *table = keys_3;  // free until the first big-group sort
            uint32_t *d_dups = ctx->d_mail + 300;
            DK_HIP(ctx, hipMemsetAsync(table, 0xFF, static_cast<size_t>(table_size) * sizeof(uint64_t), st));
            DK_HIP(ctx, hipMemsetAsync(d_dups, 0, PP_MAX_CAND * sizeof(uint32_t), st));
            {
                LaunchScope ls(ctx, K_PREFIX_PROBE, 16.0 * m);
                k_prefix_probe<<<dim3(div_up(m, 256)), dim3(256), 0, st>>>(d_text, n, d_code, bits, spk, m, span, cands, table, table_size - 1,
                                                                          d_dups);
            }
            DK_HIP(ctx, hipGetLastError());
            DK_HIP(ctx, hipMemcpyAsync(ctx->h_mail + 300, d_dups, PP_MAX_CAND * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            DK_HIP(ctx, hipStreamSynchronize(st));
            // m is 10 sqrt(n), so c equal pairs in the sample say that about c / 50 of ALL suffixes share their prefix with another one.
            // A radix pass over everything costs what a text round costs for about an eighth of it: up to 4 equal pairs (8 %) the
            // shorter prefix wins, its survivors go through the text round
            const uint32_t max_dups = static_cast<uint32_t>(DK_KNOB("DK_PROBE_DUPS", 4));
            for (int c = 0; c < cands.count; ++c)
                if (ctx->h_mail[300 + c] <= max_dups) { spk_sort = cands.sym[c]; break; }
            // the deepest candidate is (about) the key of the initial sort: a sample of 10 sqrt(n) suffixes meets an equal one there about as
            // often as a suffix sits in a group of more than sqrt(n) / 10 members -- the share of the big groups
            probe_big_share = static_cast<double>(ctx->h_mail[300 + cands.count - 1]) / m;
            if (trace)
                fprintf(stderr, "[dk] prefix probe: %u samples, equal pairs at %d/%d/%d/%d symbols: %u %u %u %u -> sort %d of %d symbols\n", m,
                        cands.sym[0], cands.sym[1], cands.sym[2], cands.sym[3], ctx->h_mail[300], ctx->h_mail[301], ctx->h_mail[302],
                        ctx->h_mail[303], spk_sort, spk);
        }
    }
    // BWT on the way (BwtCarry): callers that want L.  The key gives up its low byte to the code of the symbol in front of the suffix, so
    // the initial sort covers at most 56 bits = seven passes instead of eight; nothing is gathered from the text afterwards.
    // DK_BWT_CARRY=0 (test hook): sort the full key and let the caller gather L from the suffix array.
    const bool carry_enabled = DK_KNOB("DK_BWT_CARRY", 1) != 0;
    const bool carry_bwt = carry_enabled && d_bwt && d_origin && bwt_written;
    // A block that is periodic nearly everywhere (nine tenths of its 64-byte windows follow one period p <= 8: a^n b, (ab)^n, a zeroed buffer)
    // needs no more of the initial sort than the period string itself: the period round (4b) places every suffix of a stretch whatever the
    // depth, as long as the depth covers p symbols -- one or two passes instead of seven (a^n b, 1e8 bytes: 15.1 -> 7 ms).
    if (period_mode != 0 && n >= (1u << 16)) {
        const uint32_t *pc = ctx->h_mail + 16 + 258;
        uint32_t cmax = 0;
        for (int q = 1; q <= 8; ++q) cmax = std::max(cmax, pc[q - 1]);
        if (static_cast<uint64_t>(cmax) * 64 * 10 >= static_cast<uint64_t>(n) * 9) {
            int q = 1;
            while (static_cast<uint64_t>(pc[q - 1]) * 10 < static_cast<uint64_t>(cmax) * 9) ++q;
            const int passes = (q * bits + 7) / 8;
            const int k = std::min(spk, (8 * passes) / bits);
            if (k >= q && k < spk_sort) {
                spk_sort = k;
                if (trace) fprintf(stderr, "[dk] period %d nearly everywhere (%u of %zu windows): the initial sort takes %d symbols\n", q, cmax, n / 64, k);
            }
        }
    }
    const bool short_prefix = spk_sort < spk;  // the probe's verdict: few suffixes will survive the initial sort
    if (carry_bwt) spk_sort = std::min(spk_sort, 56 / bits);  // (a key merely shortened to make room for the carried byte keeps the rank path)
    uint8_t *sym = nullptr, *sym_alt = nullptr;
    uint8_t *d_inv = reinterpret_cast<uint8_t *>(ctx->d_mail + 576);
    if (carry_bwt) {
        sym = ctx->ws_alloc<uint8_t>(n);
        sym_alt = ctx->ws_alloc<uint8_t>(n);
        if (!sym || !sym_alt) return DK_E_NOMEM;
        uint8_t inv[256] = {0};
        for (int c = 255; c >= 0; --c)
            if (ctx->h_mail[16 + c]) inv[code[c]] = static_cast<uint8_t>(c);
        std::memcpy(ctx->h_mail + 576, inv, 256);
        DK_HIP(ctx, hipMemcpyAsync(d_inv, ctx->h_mail + 576, 256, hipMemcpyHostToDevice, st));
    }
    const int key_shift = carry_bwt ? 8 : 0;
    const bool narrow_keys = DK_KNOB("DK_NARROW_KEYS", 1) != 0 && bits * spk_sort <= 40;
    uint32_t *d_starts = ctx->d_mail + 720;  // 256 words

    // 3. initial sort; its first pass builds the keys from the text (no key array is ever written unsorted)
    {
        TextKeys tk;
        tk.t = d_text; tk.n = n; tk.code = d_code; tk.bits = bits; tk.spk = spk_sort; tk.with_prev = carry_bwt ? 1 : 0;
        // the last pass leaves every suffix at its slot of the suffix array itself (and the symbol in front of it in L): what the first
        // rerank finds final is final where it stands -- it stores nothing for it, and a tile without a survivor costs it nothing
        SortFinalOut fin;
        fin.vals = d_sa;
        if (carry_bwt) { fin.bwt = d_bwt; fin.inv_code = d_inv; fin.origin = d_origin; }
        // up to 40 sorted bits (short prefixes: random bytes, small alphabets): the last pass leaves 32-bit keys for the first rerank
        if (narrow_keys) { fin.narrow_shift = key_shift; fin.bucket_starts = d_starts; }
        DK_TRY(sort_pairs(ctx, keys, keys_alt, vals, vals_alt, n, key_shift, bits * spk_sort + key_shift, &tk, &fin));
    }

    if (short_prefix) route |= DK_ROUTE_SHORT_PREFIX;
    if (narrow_keys) route |= DK_ROUTE_NARROW_KEYS;
    // 3b. A caller that wants L, not the suffix array: only the groups with different symbols in front are refined, from the text alone
    //     (lfirst.inc).  DK_LFIRST: 0 = never, 1 = blocks of at least 2^16 bytes (default), 2 = always (test hook).
    const int lf_mode = DK_KNOB("DK_LFIRST", 1);
    // Not where the probe saw more than 60 % of its sample in big groups (lfirst_path would find the same after a rerank, see there), not
    // where runs of one byte value are common (k_run_probe), and not behind a shortened key: next to nothing survives such a sort, and the uniformity test costs the first rerank more than it saves
    // (2^30 random bytes: reduce 2.3 against 1.1 ms, nothing else differs; 2^28 {A,C,G,T}: 8.63 against 8.56 ms).
    // the block's dominant short period, if it has any periodic 64-byte window at all (k_period_probe): the smallest p with nearly the most windows
    int lf_period = 0;
    if (period_mode != 0) {
        const uint32_t *pc = ctx->h_mail + 16 + 258;
        uint32_t cmax = 0;
        for (int q = 1; q <= 7; ++q) cmax = std::max(cmax, pc[q - 1]);
        for (int q = 1; q <= 7 && cmax > 0 && !lf_period; ++q)
            if (static_cast<uint64_t>(pc[q - 1]) * 10 >= static_cast<uint64_t>(cmax) * 9) lf_period = q;
    }
    // with tokens to settle them in one round (lfirst.inc: LfTokens), runs may hold up to a twentieth of the block (50 MB of real text -- indentation,
    // rulers -- 1.3 %: 10.9 ms this way against 12.7; 5 MB of 90 % zero bytes, 24 %, stay with prefix doubling); without, a hundredth and no long run
    const bool lf_runs_ok = lf_period > 0 ? static_cast<double>(ctx->h_mail[16 + 257]) * 16.0 <= 0.05 * static_cast<double>(n) : !long_run;
    (void)run_heavy;
    if (carry_bwt && allow_lfirst && lf_mode != 0 && n >= 64 && (lf_mode == 2 || (n >= (1u << 16) && probe_big_share <= 0.6 && !short_prefix && lf_runs_ok))) {
        // (the arena of the deep groups: the second list buffers of the suffix-array path, which this path does not use, and the upper half of the
        //  initial keys' buffer -- the lower half holds the next-break positions of a token round)
        //  initial keys' buffer -- 8 n bytes: [0, 4 n) the next-break positions of a round with tokens, [4 n, 5 n) the arena's symbols, then the depth of every
        //  group (n / 2 + 2 words) and of every big group (n / 32 + 2 words), and a byte per big group (periodic or not)
        uint8_t *kb8 = reinterpret_cast<uint8_t *>(keys);
        uint32_t *gdepth = reinterpret_cast<uint32_t *>(kb8 + ((5 * n + 15) & ~size_t(15)));
        const LfBuffers b{keys_alt, keys_3, vals_3, vals, rank, sym_alt, vals_alt, pos, gid, sym, gstart, bigidx, bigoff, pos_alt, gid_alt, kb8 + 4 * n, gdepth, gdepth + n / 2 + 2, reinterpret_cast<uint8_t *>(gdepth + n / 2 + 2 + n / 32 + 2)};
        bool done = false, pristine = true;
        route |= DK_ROUTE_LFIRST;
        // (the initial keys are read by the first rerank only: their buffer holds the next-break positions of a token round later)
        // tokens in the first round (instead of after the first stalled one) where a thousandth of the block lies inside runs of one byte value or in
        // periodic 64-byte windows: indentation and rulers in real text
        const uint32_t *pc = ctx->h_mail + 16 + 258;
        const bool tokens_early = lf_period > 0 && DK_KNOB("DK_LF_TOKENS_EARLY", 1) != 0 &&
                                  (static_cast<double>(ctx->h_mail[16 + 257]) * 16.0 > 0.001 * static_cast<double>(n) || static_cast<double>(pc[lf_period - 1]) * 64.0 > 0.001 * static_cast<double>(n));
        DK_TRY(lfirst_path(ctx, d_text, n, keys, key_shift, narrow_keys ? d_starts : nullptr, d_sa, d_bwt, d_origin, b, static_cast<uint32_t>(spk_sort), trace, lf_mode == 2, &done, &pristine,
                           nullptr, lf_period, reinterpret_cast<uint32_t *>(keys), tokens_early));
        if (done) {
            ctx->ws_release(mark);
            *bwt_written = true;
            return DK_OK;
        }
        route = (route & ~static_cast<uint32_t>(DK_ROUTE_LFIRST | DK_ROUTE_LFIRST_BIG_ROUND | DK_ROUTE_LFIRST_DEEP | DK_ROUTE_PERIOD_ROUND)) | DK_ROUTE_LFIRST_FALLBACK;
        if (!pristine) {  // it gave up half way (L has been written to): the suffix-array path, from the start
            ctx->ws_release(mark);
            return suffix_array_impl(ctx, d_text, n, d_sa, d_bwt, d_origin, bwt_written, false);
        }
    }

    // 4. first rerank (slots are SA positions).  No rank array yet: the suffixes that survive the initial sort are first extended from
    //    the text (5a), which needs no ranks, and the rank array is built once, late, for whatever survives that (5b) -- one inverse
    //    permutation for the whole sort instead of one here plus tens of millions of random rank stores in the next round.
    size_t active = 0, groups = 0, nbig = 0, nmedium = 0, nbiggroups = 0;
    bool have_ranks = false;
    const BwtCarry first_bc{key_shift, carry_bwt ? d_bwt : nullptr, d_inv, d_origin, nullptr, sym};
    DK_TRY(rerank(ctx, keys, d_sa, nullptr, n, nullptr, nullptr, d_sa, vals_alt, pos, gid, gstart, first_bc, narrow_keys ? d_starts : nullptr));
    DK_TRY(classify_and_read(ctx, n / 2, gstart, bigidx, bigoff, &active, &groups, &nbig, &nmedium, &nbiggroups));
    std::swap(vals, vals_alt);  // vals = suffix indices of the active list

    uint64_t h = static_cast<uint64_t>(spk_sort);
    if (trace)
        fprintf(stderr, "[dk] n=%zu sigma=%u bits=%d spk=%d%s: after init sort active=%zu groups=%zu big=%zu\n", n, sigma, bits, spk_sort,
                carry_bwt ? " +prev" : "", active, groups, nbig);

    // one general round: secondary keys (ranks h further on, or the next tsym symbols of the text) -> every group sorted inside its
    // own slot range (small groups in LDS, big ones through the global sort) -> rerank.  Returns the symbols the round added (text
    // rounds: the big groups' keys may hold fewer symbols than the small groups' -- the depth every group is known to is the smaller).
    auto run_round = [&](int tsym, int *advanced, int period = 0, const uint32_t *next_break = nullptr) -> int {
        const uint32_t h_eff = static_cast<uint32_t>(std::min<uint64_t>(h, n));
        const int bsbits = nbiggroups > 1 ? static_cast<int>(ceil_log2_u64(nbiggroups)) : 1;  // bits of a big group's dense index
        int kbits, kb;
        const bool raw_text = tsym > 0 && (bits >= 5 || period > 0);  // eight raw bytes per key beat the code table unless the alphabet is small
        const int tbits = raw_text ? 8 : bits;
        const int ebits = static_cast<int>(ceil_log2_u64(static_cast<uint64_t>(n) + 1));
        if (period > 0) {
            // the period round: tokens (direction, length to the break, bytes from the break on) in 64-bit keys; the big list takes their
            // leading bits down to the first byte behind the break when the symbol in front still fits below them, else what fits.  The depth
            // every group is known to stays h: equal tokens say "equal up to the break" and the break may come right behind h.
            tsym = 8;
            kbits = 64;
            kb = 1 + ebits + 8;
            if (kb + bsbits + 8 > 64) kb = std::max(1 + ebits, 56 - bsbits);
            if (kb + bsbits > 63) kb = 63 - bsbits;
            if (advanced) *advanced = 0;
        } else if (tsym > 0) {
            if (raw_text) tsym = 8;
            kbits = tsym * tbits;
            // the big list's key: (offset in the big list) above the leading symbols of the secondary key -- at most 32 bits of it: every
            // digit of a list this short costs three launches, and what four more bytes leave unresolved the next round takes
            const int tsym_big = std::min(std::min(tsym, (63 - bsbits) / tbits), std::max(1, 32 / tbits));
            kb = tsym_big * tbits;
            if (advanced) *advanced = nbig > 0 ? tsym_big : tsym;
        } else {
            kbits = kb = static_cast<int>(ceil_log2_u64(static_cast<uint64_t>(n) + h_eff));
        }
        const TextSource ts{d_text, d_code, raw_text ? 0 : bits, tsym, period > 0 ? next_break : nullptr, period, ebits};
        route |= period > 0 ? DK_ROUTE_PERIOD_ROUND : tsym > 0 ? DK_ROUTE_TEXT_ROUND : DK_ROUTE_GENERAL_ROUND;
        if (nbig > 0) route |= DK_ROUTE_BIG_GROUPS;
        const int big_carry = carry_bwt && kb + bsbits + 8 <= 64 ? 1 : 0;  // the big list's keys have room for the symbol in front
        uint32_t *bslot = pos_alt;  // written by the rerank at the end of the round only: free until k_big_back has read it
        {
            LaunchScope ls(ctx, K_ROUND_LOCAL, 8.0 * active + 4.0 * active + 12.0 * active);
            if (tsym > 0)
                k_round_local<true><<<dim3(div_up(active, LS_TILE)), dim3(LS_BLOCK), 0, st>>>(
                    vals, gid, gstart, bigidx, bigoff, rank, static_cast<uint32_t>(n), h_eff, kbits, kb, big_carry, active, keys, vals_alt, keys_alt, vals_3, bslot, ts, sym, sym_alt);
            else
                k_round_local<false><<<dim3(div_up(active, LS_TILE)), dim3(LS_BLOCK), 0, st>>>(
                    vals, gid, gstart, bigidx, bigoff, rank, static_cast<uint32_t>(n), h_eff, kbits, kb, big_carry, active, keys, vals_alt, keys_alt, vals_3, bslot, ts, sym, sym_alt);
        }
        DK_HIP(ctx, hipGetLastError());
        if (nbig > 0) {
            uint64_t *bk = keys_alt, *bk_alt = keys_3;
            uint32_t *bv = vals_3, *bv_alt = vals;  // the round's input list has been read (k_round_local); the rerank rewrites it below
            DK_TRY(sort_pairs(ctx, bk, bk_alt, bv, bv_alt, nbig, 8 * big_carry, kb + bsbits + 8 * big_carry));
            {
                LaunchScope ls(ctx, K_BIG_BACK, 28.0 * nbig);
                k_big_back<<<dim3(div_up(nbig, 256)), dim3(256), 0, st>>>(bk, bv, bslot, nbig, keys, vals_alt, d_text, static_cast<uint32_t>(n),
                                                                          carry_bwt ? sym_alt : nullptr, big_carry);
            }
            DK_HIP(ctx, hipGetLastError());
        }
        // keys / vals_alt (/ sym_alt) now hold every group sorted by its secondary key in its own slot range
        size_t next_active = 0, next_groups = 0, next_big = 0, next_medium = 0, next_biggroups = 0;
        const BwtCarry bc{0, carry_bwt ? d_bwt : nullptr, nullptr, d_origin, sym_alt, sym};
        // SA entries of suffixes that become final after the rank array exists are read by nobody when the caller wants L (the inverse
        // permutation was the last reader): 84 M scattered 4-byte stores less on 1e8 bytes of word-like text
        uint32_t *sa_out = carry_bwt && have_ranks ? nullptr : d_sa;
        DK_TRY(rerank(ctx, keys, vals_alt, pos, active, gid, have_ranks ? rank : nullptr, sa_out, vals, pos_alt, gid_alt, gstart, bc));
        DK_TRY(classify_and_read(ctx, active / 2, gstart, bigidx, bigoff, &next_active, &next_groups, &next_big, &next_medium, &next_biggroups));
        std::swap(pos, pos_alt);
        std::swap(gid, gid_alt);
        if (trace)
            fprintf(stderr, "[dk] round %u h=%llu%s slots=%zu big=%zu key bits=%d (big %d+%d) -> active=%zu groups=%zu big=%zu (>%d: %zu)\n", ctx->stats.rounds,
                    (unsigned long long)h, period > 0 ? " (period tokens)" : tsym > 0 ? " (text)" : "", active, nbig, kbits, bsbits, kb, next_active, next_groups, next_big, PL_MAX, next_medium);
        active = next_active;
        groups = next_groups;
        nbig = next_big;
        nmedium = next_medium;
        nbiggroups = next_biggroups;
        ctx->stats.rounds += 1;
        return DK_OK;
    };

    // A caller that wants L: once the big groups hold little of what is active (doubling has broken them up, or there never were many behind
    // a shortened key), the rest is finished the L-first way (lfirst.inc: only the groups with different symbols in front, from the text,
    // inside LDS) instead of more rounds on ranks, pair chains and in-place rounds.  DK_LF_SWITCH: per cent of the active slots that may
    // still sit in big groups at the switch (0: never).  -> 1: L is complete; 0: not taken; -1: it gave up half way (start over).
    // Measured, round 4 (1e8 bytes of word-like text, suffix-array path 20.7 ms): switch at 1 % 20.4, at 5 % 19.3-20.1, at 25 % 21.2, at 40-60 %
    // 22.9, at 80 % (= L-first from the start) 20.5 ms -- and 5 % costs inputs whose big groups do not split on text (90 % zeros: 1.8 -> 2.9 ms
    // at 5 MB) more than it gains here.  Off in the product; the tuning build keeps the switch and tests/test_env_variants.py the parity.
    const int lf_switch = DK_KNOB("DK_LF_SWITCH", 0);
    auto take_over = [&](int *outcome) -> int {
        *outcome = 0;
        if (!(carry_bwt && allow_lfirst && lf_mode != 0 && lf_switch > 0 && !long_run && n >= (1u << 16) && active > 0 &&
              nbig * 100 <= active * static_cast<size_t>(lf_switch)))
            return DK_OK;
        if (nbig > 0 && nbig / nbiggroups > LF_AVG_BIG) return DK_OK;  // giant groups (periodic input) are doubling's business
        // buffers: everything but the current list is free between rounds; the key buffer that is not a sort's ping-pong partner holds the
        // big list's second suffix array and its symbols (4 n + n of its 8 n bytes); the rank array becomes the big list's positions
        uint32_t *spare = reinterpret_cast<uint32_t *>(keys);
        // (the arena of the deep groups: the caller's own list, free once the first rerank has filtered it)
        uint32_t *gdepth = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(keys) + ((5 * n + 15) & ~size_t(15)));  // (behind the big list's symbols, see above)
        const LfBuffers b{keys_alt, keys_3, vals_3, spare, rank, reinterpret_cast<uint8_t *>(spare + n), vals_alt, pos_alt, gid_alt, sym_alt, gstart, bigidx, bigoff, vals, pos, sym, gdepth, gdepth + n / 2 + 2, reinterpret_cast<uint8_t *>(gdepth + n / 2 + 2 + n / 32 + 2)};
        const LfFrom from{vals, pos, gid, sym, active};
        bool done = false, pristine = true;
        if (trace) fprintf(stderr, "[dk] %zu active, %zu of them in big groups, depth %llu: the L-first path takes over\n", active, nbig, (unsigned long long)h);
        route |= DK_ROUTE_LFIRST;
        DK_TRY(lfirst_path(ctx, d_text, n, nullptr, 0, nullptr, nullptr, d_bwt, d_origin, b, static_cast<uint32_t>(std::min<uint64_t>(h, n)), trace, true, &done, &pristine, &from));
        *outcome = done ? 1 : -1;
        return DK_OK;
    };
#define DK_TAKE_OVER()                                                                                                      \
    do {                                                                                                                    \
        int outcome_ = 0;                                                                                                   \
        DK_TRY(take_over(&outcome_));                                                                                       \
        if (outcome_ > 0) { ctx->ws_release(mark); *bwt_written = true; return DK_OK; }                                     \
        if (outcome_ < 0) {                                                                                                 \
            ctx->ws_release(mark);                                                                                          \
            return suffix_array_impl(ctx, d_text, n, d_sa, d_bwt, d_origin, bwt_written, false);                            \
        }                                                                                                                   \
    } while (0)

    // 4b. the period round (see k_period_probe): where an eighth of the block lies in stretches of one period p <= 8 (and p <= h: the members
    //     of a group must share the period string itself), every suffix inside such a stretch is placed by where the stretch ends -- a^n b,
    //     (ab)^n, zero padding: one round instead of log2(length) doubling rounds over all of them.  The rank array's buffer holds the
    //     next-break positions (it is built later).
    bool small_period_round = false;
    if (active > 0 && period_mode != 0) {
        const uint32_t *pc = ctx->h_mail + 16 + 258;
        const int pmax = static_cast<int>(std::min<uint64_t>(8, h));
        uint32_t cmax = 0;
        for (int q = 1; q <= pmax; ++q) cmax = std::max(cmax, pc[q - 1]);
        int period = 0;
        if (cmax > 0 && (period_mode == 2 || static_cast<uint64_t>(cmax) * 64 * 8 >= n))
            for (int q = 1; q <= pmax && !period; ++q)
                if (static_cast<uint64_t>(pc[q - 1]) * 10 >= static_cast<uint64_t>(cmax) * 9) period = q;
        if (period) {
            small_period_round = true;
            const size_t ptiles = div_up(n, PB_TILE);
            const size_t mark2 = ctx->ws_mark();
            uint32_t *tile_first = ctx->ws_alloc<uint32_t>(ptiles);
            if (!tile_first) return DK_E_NOMEM;
            {
                LaunchScope ls(ctx, K_PERIOD, 2.0 * n + 4.0 * n);
                k_period_first<<<dim3(ptiles), dim3(256), 0, st>>>(d_text, static_cast<uint32_t>(n), period, tile_first);
                k_period_spine<<<dim3(1), dim3(1024), 0, st>>>(tile_first, ptiles);
                k_period_fill<<<dim3(ptiles), dim3(256), 0, st>>>(d_text, static_cast<uint32_t>(n), period, tile_first, rank);
            }
            DK_HIP(ctx, hipGetLastError());
            if (trace)
                fprintf(stderr, "[dk] period probe: 64-byte windows of period 1..8: %u %u %u %u %u %u %u %u of %zu -> period %d\n", pc[0], pc[1], pc[2], pc[3], pc[4], pc[5],
                        pc[6], pc[7], n / 64, period);
            DK_TRY(run_round(8, nullptr, period, rank));
            ctx->ws_release(mark2);
        }
    }
    // 5a. extend the survivors' keys from the text: up to floor(63 / bits) further symbols per round, no ranks needed.  A second
    //     such round only when the first left a lot (otherwise what is left are long repeats, which want doubling).
    for (int t = 0; !have_ranks && active > 0 && t < 2; ++t) {
        if (t == 1 && !short_prefix && active * 8 < n) break;
        // mostly giant groups (periodic or run-dominated input): a text round would push them through the global sort for a few more
        // symbols each, where a doubling round doubles the depth for the same sort -- go straight to the ranks
        if (nbig * 2 > active) break;
        int adv = 0;
        // (small alphabets, coded keys: no more symbols than the depth reached so far -- the round doubles the depth like a doubling round
        // would, and its keys cost a table lookup per symbol: 2^28 ACGT took 31 symbols per survivor where 16 separate all but the repeats)
        DK_TRY(run_round(std::min<int>(std::min(spk, 63 / bits), static_cast<int>(std::min<uint64_t>(h, 64))), &adv));
        h += static_cast<uint64_t>(adv);
    }
    DK_TAKE_OVER();
    // 5b. survivors beyond that (long repeats): build the rank array the doubling rounds need
    if (active > 0 && !have_ranks) {
        // up to 2^27 suffixes the inverse permutation goes through LDS windows and can hand every active suffix the position of its
        // group's head by itself (marked entries of SA; pos_alt is free between rounds); above, a second pass over the active list does
        // (worth it from a quarter of all suffixes active: the marked form costs the first split 0.17 ms per 1e8 suffixes, the second pass
        // it replaces 0.14 ms per 1e7 active ones)
        const bool marked = inverse_through_windows(n) && active * 4 > n;
        route |= marked ? DK_ROUTE_ISA_MARKED : 0u;
        route |= DK_ROUTE_ISA_WINDOWS;
        uint32_t *head_pos = marked ? pos_alt : nullptr;
        {
            LaunchScope ls(ctx, K_PLACE_ACTIVE, (marked ? 28.0 : 12.0) * active);
            k_place_active<<<dim3(div_up(active, 256)), dim3(256), 0, st>>>(vals, pos, gid, gstart, active, d_sa, head_pos);
        }
        DK_TRY(inverse_permutation(ctx, d_sa, n, keys_alt, keys_3, rank, head_pos));  // rank[SA[p]] = p (keys_alt / keys_3: free between rounds)
        if (!marked) {
            LaunchScope ls(ctx, K_PLACE_ACTIVE, 16.0 * active);
            k_rank_active<<<dim3(div_up(active, 256)), dim3(256), 0, st>>>(vals, pos, gid, gstart, active, rank);
        }
        DK_HIP(ctx, hipGetLastError());
        have_ranks = true;
    }

    // 5c'. a long period (9 .. 2^18: fixed-size records, a table of one row)?  Looked for only where giant groups hold most of what is active --
    //      doubling would run over everything for log2(n / h) rounds -- and used by one period round (4b's, with the tokens of this period) as
    //      soon as the depth covers it: period 1000 x 8000 (8 MB) 21 -> 9 rounds.
    uint32_t long_period = 0;
    if (period_mode != 0 && !small_period_round && active > 0 && nbig * 2 > active && n >= (1u << 16)) {
        uint32_t *d_found = ctx->d_mail + 320;  // PS_SAMPLES words + 1 counter (words 300 .. 304 are the prefix probe's)
        const uint32_t pmax = static_cast<uint32_t>(std::min<uint64_t>(PS_MAX, n / 4));
        DK_HIP(ctx, hipMemsetAsync(d_found, 0xFF, PS_SAMPLES * sizeof(uint32_t), st));
        DK_HIP(ctx, hipMemsetAsync(d_found + PS_SAMPLES, 0, sizeof(uint32_t), st));
        {
            LaunchScope ls(ctx, K_PERIOD, 0.0);
            k_period_search<<<dim3(PS_SAMPLES), dim3(256), 0, st>>>(d_text, n, pmax, d_found);
        }
        DK_HIP(ctx, hipMemcpyAsync(ctx->h_mail + 320, d_found, PS_SAMPLES * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        DK_HIP(ctx, hipStreamSynchronize(st));
        uint32_t best = 0, best_votes = 0;  // the period most samples agree on
        for (int a = 0; a < PS_SAMPLES; ++a) {
            const uint32_t v = ctx->h_mail[320 + a];
            if (v == 0xFFFFFFFFu) continue;
            uint32_t votes = 0;
            for (int b2 = 0; b2 < PS_SAMPLES; ++b2) votes += ctx->h_mail[320 + b2] == v ? 1u : 0u;
            if (votes > best_votes || (votes == best_votes && v < best)) { best = v; best_votes = votes; }
        }
        if (best_votes >= PS_SAMPLES / 4) {
            {
                LaunchScope ls(ctx, K_PERIOD, 2.0 * n);
                k_period_count<<<dim3(div_up(div_up(n, 64), 256)), dim3(256), 0, st>>>(d_text, n, best, d_found + PS_SAMPLES);
            }
            DK_HIP(ctx, hipMemcpyAsync(ctx->h_mail + 320 + PS_SAMPLES, d_found + PS_SAMPLES, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            DK_HIP(ctx, hipStreamSynchronize(st));
            // three quarters of the block: a Fibonacci word follows "period 55" in 47 % of its windows, in stretches of a few hundred symbols that one
            // round on their ends does not settle (6.35 against 5.71 ms at 4 MB)
            if (static_cast<uint64_t>(ctx->h_mail[320 + PS_SAMPLES]) * 64 * 4 >= static_cast<uint64_t>(n) * 3) long_period = best;
            if (trace)
                fprintf(stderr, "[dk] long period: %u of %d samples say %u, %u of %zu windows follow it -> %s\n", best_votes, PS_SAMPLES, best, ctx->h_mail[320 + PS_SAMPLES], n / 64,
                        long_period ? "a period round once the depth covers it" : "not used");
        }
    }
    // 5c. doubling rounds: the general form while big groups exist, ...
    const bool plateau_enabled = DK_KNOB("DK_PLATEAU", 1) != 0;
    while (active > 0 && (nmedium > 0 || !plateau_enabled)) {
        if (ctx->stats.rounds > 44) return ctx->fail(DK_E_INTERNAL, "suffix_array: no convergence after 44 rounds");
        if (long_period && h >= long_period) {
            // (next-break positions: the suffix array's buffer when the caller wants L -- nobody reads SA once the ranks exist -- else workspace, if it has room)
            const size_t mark2 = ctx->ws_mark();
            uint32_t *nb = carry_bwt ? d_sa : ctx->ws_try_alloc<uint32_t>(n);
            const size_t ptiles = div_up(n, PB_TILE);
            uint32_t *tile_first = nb ? ctx->ws_try_alloc<uint32_t>(ptiles) : nullptr;
            const uint32_t p_now = long_period;
            long_period = 0;
            if (nb && tile_first) {
                {
                    LaunchScope ls(ctx, K_PERIOD, 2.0 * n + 4.0 * n);
                    k_period_first<<<dim3(ptiles), dim3(256), 0, st>>>(d_text, static_cast<uint32_t>(n), static_cast<int>(p_now), tile_first);
                    k_period_spine<<<dim3(1), dim3(1024), 0, st>>>(tile_first, ptiles);
                    k_period_fill<<<dim3(ptiles), dim3(256), 0, st>>>(d_text, static_cast<uint32_t>(n), static_cast<int>(p_now), tile_first, nb);
                }
                DK_HIP(ctx, hipGetLastError());
                DK_TRY(run_round(8, nullptr, static_cast<int>(p_now), nb));
                ctx->ws_release(mark2);
                continue;
            }
            ctx->ws_release(mark2);
        }
        DK_TRY(run_round(0, nullptr));
        h *= 2;
        DK_TAKE_OVER();
    }
    // ... then in place (k_plateau_sort): one sort kernel + one rank kernel per round, the live count read back one round late
    if (active > 0) {
        size_t slots = active;
        route |= DK_ROUTE_INPLACE_ROUNDS;
        uint32_t *idx_a = vals, *idx_b = vals_alt;
        uint32_t *meta_a = gid, *meta_b = gid_alt;
        uint8_t *sym_a = sym, *sym_b = sym_alt;
        uint32_t *d_live = ctx->d_mail + 700, *h_live = ctx->h_mail + 700;  // ring of 8 counters
        {
            LaunchScope ls(ctx, K_PLATEAU_RANKS, 14.0 * slots);
            k_to_inplace<<<dim3(div_up(slots, 256)), dim3(256), 0, st>>>(gid, gstart, slots, meta_b);
        }
        std::swap(meta_a, meta_b);  // gid was read, gid_alt written
        // pair chains first: the groups of two (three, four) that long repeats leave behind are settled by one comparison per chain
        size_t settled_by_chains = 0;
        if (DK_KNOB("DK_PAIR_CHAINS", 1) != 0 && slots >= 2) {
            uint64_t *rec_key = keys, *rec_key_alt = keys_alt;  // (all free at this point: nothing is sorted globally any more)
            uint32_t *rec_slot = vals_3, *rec_slot_alt = pos_alt;
            uint8_t *planes = reinterpret_cast<uint8_t *>(keys_3);  // CH_K - 1 planes of one verdict byte per slot (3 n of the buffer's 8 n bytes) ...
            uint8_t *rec_verdict = planes + static_cast<size_t>(CH_K - 1) * n;  // ... and one byte per record behind them
            const int kmax = std::min(CH_K, std::max(2, DK_KNOB("DK_CHAIN_GROUP", CH_K)));
            const int lbits = static_cast<int>(ceil_log2_u64(n));
            const uint32_t cap = static_cast<uint32_t>(std::min<size_t>(n, 0xFFFFFFF0u));  // records the buffers hold (n each); what does not fit stays with the rounds
            uint32_t *d_cnt = ctx->d_mail + 12;  // [0] records, [1] first reservation that did not fit
            DK_HIP(ctx, hipMemsetAsync(d_cnt, 0, sizeof(uint32_t), st));
            DK_HIP(ctx, hipMemsetAsync(d_cnt + 1, 0xFF, sizeof(uint32_t), st));
            DK_HIP(ctx, hipMemsetAsync(planes, 0, static_cast<size_t>(kmax - 1) * slots, st));
            {
                LaunchScope ls(ctx, K_CHAIN, 16.0 * slots);
                k_chain_extract<<<dim3(div_up(slots, 256 * CH_EXTRACT)), dim3(256), 0, st>>>(idx_a, meta_a, slots, kmax, lbits, rec_key, rec_slot, cap, d_cnt);
            }
            DK_HIP(ctx, hipGetLastError());
            DK_HIP(ctx, hipMemcpyAsync(ctx->h_mail + 12, d_cnt, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            DK_HIP(ctx, hipStreamSynchronize(st));
            const uint32_t m = std::min(ctx->h_mail[12], ctx->h_mail[13]);
            if (trace) fprintf(stderr, "[dk] pair chains: %u pairs inside groups of 2..%d among %zu slots%s (h = %llu)\n", m, kmax, slots, ctx->h_mail[13] != 0xFFFFFFFFu ? " (list full)" : "",
                               static_cast<unsigned long long>(h));
            if (m > 0) {
                route |= DK_ROUTE_PAIR_CHAINS;
                DK_TRY(sort_pairs(ctx, rec_key, rec_key_alt, rec_slot, rec_slot_alt, m, 0, 2 * lbits));
                const size_t mark2 = ctx->ws_mark();
                const uint32_t ntiles = static_cast<uint32_t>(div_up(m, CH_TILE));
                uint32_t *tile_max = ctx->ws_alloc<uint32_t>(ntiles);
                if (!tile_max) return DK_E_NOMEM;
                {
                    LaunchScope ls(ctx, K_CHAIN, 22.0 * m + 14.0 * slots);
                    k_chain_ends<<<dim3(div_up(m, 256)), dim3(256), 0, st>>>(rec_key, m, lbits, rank, static_cast<uint32_t>(n), static_cast<uint32_t>(std::min<uint64_t>(h, n)), rec_verdict);
                    k_chain_tiles<<<dim3(ntiles), dim3(256), 0, st>>>(rec_verdict, m, tile_max);
                    k_chain_spine<<<dim3(1), dim3(1024), 0, st>>>(tile_max, ntiles);
                    k_chain_verdicts<<<dim3(ntiles), dim3(256), 0, st>>>(rec_verdict, rec_key, rec_slot, m, tile_max, planes, slots);
                    k_chain_apply<<<dim3(div_up(slots, 256)), dim3(256), 0, st>>>(idx_a, meta_a, sym_a, pos, planes, slots, carry_bwt ? nullptr : d_sa, carry_bwt ? d_bwt : nullptr, d_origin, rank);
                }
                DK_HIP(ctx, hipGetLastError());
                // how many slots are left?  (counted, not summed up by the kernel above: an atomic per wave on one address was 17.7 of its 18 ms on
                // two identical halves, 1.5 of 1.6 ms on the 1e8 text block)
                const size_t ltiles = div_up(slots, RR_TILE);
                uint32_t *tile_live = ctx->ws_alloc<uint32_t>(ltiles);
                if (!tile_live) return DK_E_NOMEM;
                {
                    LaunchScope ls(ctx, K_PLATEAU_RANKS, 4.0 * slots);
                    k_plateau_count<<<dim3(ltiles), dim3(RR_BLOCK), 0, st>>>(idx_a, slots, tile_live);
                    k_plateau_scan<<<dim3(1), dim3(1024), 0, st>>>(tile_live, ltiles, d_cnt);
                }
                DK_HIP(ctx, hipGetLastError());
                DK_HIP(ctx, hipMemcpyAsync(ctx->h_mail + 12, d_cnt, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                DK_HIP(ctx, hipStreamSynchronize(st));
                settled_by_chains = slots - ctx->h_mail[12];
                if (trace) fprintf(stderr, "[dk] pair chains settled %zu of %zu slots\n", settled_by_chains, slots);
                ctx->ws_release(mark2);
            }
        }
        unsigned launched = 0, read = 0;
        size_t live = active - settled_by_chains;
        auto compact = [&]() -> int {  // live slots keep their order, so groups stay contiguous
            const size_t mark2 = ctx->ws_mark();
            const size_t ntiles = div_up(slots, RR_TILE);
            uint32_t *tile_live = ctx->ws_alloc<uint32_t>(ntiles);
            if (!tile_live) return DK_E_NOMEM;
            {
                LaunchScope ls(ctx, K_PLATEAU_RANKS, 4.0 * slots + 22.0 * live);
                k_plateau_count<<<dim3(ntiles), dim3(RR_BLOCK), 0, st>>>(idx_a, slots, tile_live);
                k_plateau_scan<<<dim3(1), dim3(1024), 0, st>>>(tile_live, ntiles, d_live);
                k_plateau_compact<<<dim3(ntiles), dim3(RR_BLOCK), 0, st>>>(idx_a, meta_a, sym_a, pos, slots, tile_live, idx_b, meta_b, sym_b, pos_alt);
            }
            DK_HIP(ctx, hipGetLastError());
            DK_HIP(ctx, hipMemcpyAsync(h_live, d_live, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            DK_HIP(ctx, hipStreamSynchronize(st));
            if (h_live[0] != live) return ctx->fail(DK_E_INTERNAL, "suffix_array: live count %u after compaction, expected %zu", h_live[0], live);
            ctx->ws_release(mark2);
            std::swap(idx_a, idx_b);
            std::swap(meta_a, meta_b);
            std::swap(sym_a, sym_b);
            std::swap(pos, pos_alt);
            slots = live;
            launched = read = 0;
            return DK_OK;
        };
        auto launch_round = [&]() -> int {
            const uint32_t h_eff = static_cast<uint32_t>(std::min<uint64_t>(h, n));
            uint32_t *cnt = d_live + (launched & 7u);
            DK_HIP(ctx, hipMemsetAsync(cnt, 0, sizeof(uint32_t), st));
            {
                LaunchScope ls(ctx, K_PLATEAU_SORT, 6.0 * slots + 4.0 * live + 10.0 * live);
                k_plateau_sort<<<dim3(div_up(slots, LS_TILE)), dim3(LS_BLOCK), 0, st>>>(idx_a, meta_a, sym_a, pos, rank, static_cast<uint32_t>(n), h_eff,
                                                                                      slots, idx_b, meta_b, sym_b, carry_bwt ? nullptr : d_sa, d_bwt, d_origin, cnt,
                                                                                      launched ? d_live + ((launched - 1) & 7u) : nullptr);
            }
            {
                LaunchScope ls(ctx, K_PLATEAU_RANKS, 6.0 * slots);
                k_plateau_ranks<<<dim3(div_up(slots, 256)), dim3(256), 0, st>>>(idx_b, meta_b, pos, slots, rank,
                                                                                launched ? d_live + ((launched - 1) & 7u) : nullptr);
            }
            DK_HIP(ctx, hipGetLastError());
            DK_HIP(ctx, hipMemcpyAsync(h_live + (launched & 7u), cnt, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            DK_HIP(ctx, hipEventRecord(ctx->round_ev[launched & 7u], st));
            std::swap(idx_a, idx_b);
            std::swap(meta_a, meta_b);
            std::swap(sym_a, sym_b);
            h = std::min<uint64_t>(h * 2, static_cast<uint64_t>(n) * 2);
            ++launched;
            return DK_OK;
        };
        auto read_round = [&]() -> int {  // the oldest round in flight: how many slots did it leave alive?
            DK_HIP(ctx, hipEventSynchronize(ctx->round_ev[read & 7u]));
            const size_t before = live;
            live = h_live[read & 7u];
            ++read;
            ctx->stats.rounds += 1;
            if (trace) fprintf(stderr, "[dk] round %u (in place) slots=%zu live %zu -> %zu\n", ctx->stats.rounds - 1, slots, before, live);
            return DK_OK;
        };
        if (live > 0 && settled_by_chains && live * 4 <= slots && slots >= (1u << 16)) DK_TRY(compact());  // the chains left mostly dead slots behind
        for (; live > 0;) {
            if (ctx->stats.rounds > 44) return ctx->fail(DK_E_INTERNAL, "suffix_array: no convergence after 44 rounds");
            DK_TRY(launch_round());
            if (launched - read < 2) continue;  // keep one round ahead of the host
            DK_TRY(read_round());
            if (live == 0) break;  // the round launched since ran over dead slots only
            if (live * 4 <= slots && slots >= (1u << 16)) {
                // mostly dead slots: drain the round in flight, then compact (live slots keep their order: groups stay contiguous)
                DK_TRY(read_round());
                if (live == 0) break;
                DK_TRY(compact());
            }
        }
        active = 0;
    }
    if (carry_bwt) *bwt_written = true;
    ctx->ws_release(mark);
    return DK_OK;
#undef DK_TAKE_OVER
}

}  // namespace

}  // namespace dk
