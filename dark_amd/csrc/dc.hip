// Distance coding of the BWT on the GPU.  Replaces compress::bwt::mtf::MTF + compress::bwt::dc::encode + EncodeIterator as
// driven by src/block/dc.rs:51-52,82-85 (in-repo analogue etc/dark-c/src/ptax.cpp:61-111).
//
// What the serial reference computes, restated without the move-to-front list: the BWT is a sequence of runs.  For a
// run of symbol s starting at i whose previous occurrence of s ended at b, the MTF rank of s at i is the number of
// DISTINCT symbols seen in (b, i) = #{c : last_c(i) > b}, and the distance stored for position b is i - b - rank - 1.
// First occurrences go to init[s] = i.  After the last position every symbol's final run end gets
// n - b - rank - 1 with rank = #{c : last_c(n) > b} (its place in the final MTF list).  One entry per run, in position
// order, is exactly what EncodeIterator yields; Context.last_rank of the entry for run r is the rank found at r's start.
//
// THE KERNELS WALK THE BWT BACKWARDS (round 5).  Forwards, the distance of a run is known when its symbol comes BACK -- a 4-byte store into the slot of
// an earlier run, scattered over the last thousand run slots on a wide alphabet: 2^30 random bytes wrote 22 GB for 6.4 GB of outputs.  The reversed
// sequence R[p'] = L[n-1-p'] has the same runs, and the number of distinct symbols between two occurrences of s is the same from either side: walking
// R, the run that STARTS at i' (ending at b = n-1-i' in L) finds its previous occurrence in R -- its NEXT one in L -- and the very same
// i' - b' - rank - 1 now belongs to the run at hand: every chunk's distances leave as whole lines, at index m-1-(run index in R).  A symbol's
// first occurrence in R is its last run in L, whose distance is the reference's final sweep, n - b - rank - 1 = i' - (distinct symbols before i'
// in R): no sweep kernel.  init[s] = n - (last position of s in R + 1) comes from the final table.  Context.last_rank of a run (the rank found at
// ITS start in L; only the rawdc dump reads it) is what the run before it in L order of its symbol computes: the one scattered (byte) store left.
// Below, "position", "previous", "last occurrence" are meant in R.
//   k_dc_summary   one wave per tile of 4096 positions: last position (+ run index) of every symbol inside the tile,
//                  number of run starts in the tile
//   k_dc_runscan   exclusive sum of the run counts                     (run index of every tile's first new run)
//   k_dc_carry_*   256-way "last non-empty" exclusive scan over tiles  (last occurrence before each tile, per symbol)
//   k_dc_main      one wave per tile, 64 positions per step (lane = position); the 256-entry last-occurrence table lives in
//                  wave-private LDS (s_pr below), ranks are found lane-parallel, results go straight to the compact arrays
//   k_dc_init      init[s] from the final table (the reference's sweep values come out of k_dc_main: see above)
// Algorithmic bytes: 3 n (three reads of L) + 6 m (dist, sym, rank per run) + 2 * 2 KiB per tile of tables.
#include "context.hpp"
#include "device_util.hpp"

namespace dk {
namespace {

constexpr int DC_BLOCK = 256;
constexpr int DC_WAVES = DC_BLOCK / 64;
constexpr int DC_TILE = 4096;        // positions per wave
constexpr int DC_PAD = 16;           // tile bytes start at offset 16 in LDS; byte 15 holds L[base-1]
constexpr int DC_MAX_CHUNKS = 512;    // workgroups of the carry scan's outer phases (phase B is one workgroup walking over all of them)
constexpr int DC_CARRY_BATCH = 8;     // tiles whose table rows a carry thread loads before it touches any of them
constexpr uint32_t DC_WINDOW = 4096;  // positions before the tile that the bitmap of a wide tile covers as well
constexpr int DC_BM_WORDS = (DC_TILE + DC_WINDOW) / 64;
constexpr int DC_FEW_A = 12;         // case-A lanes per chunk up to which each gets its own ballot instead of the shift loop
constexpr int DC_WIDE_B = 20;        // first occurrences per chunk above which (without the bitmap) every lane walks the whole table
constexpr int DC_MARKS_TILE = 4;     // distinct symbols in a tile above which the tile keeps the bitmap
constexpr int DC_MARKS_B = 6;        // first occurrences per chunk above which the bitmap route is taken (tiles that keep the bitmap)
constexpr int DC_FEW_LATER = 16;     // wide tiles: lanes per chunk that are neither the first nor the second of their symbol up to which each finds its predecessor by itself

// R[p] = L[n-1-p]
__device__ __forceinline__ uint32_t rev_at(const uint8_t *__restrict__ L, size_t n, size_t p) { return L[n - 1 - p]; }
// Stage one tile OF THE REVERSED SEQUENCE into wave-private LDS: s[PAD + j] = R[base + j]; with PAD > 0 also its neighbours R[base-1] and
// R[base+DC_TILE].  Sixteen consecutive bytes of R are sixteen bytes of L read backwards: two unaligned 8-byte loads, byte-swapped.
template <int PAD>
__device__ __forceinline__ void stage_tile(const uint8_t *__restrict__ L, size_t n, size_t base, uint8_t *s, int lane) {
    typedef uint64_t __attribute__((aligned(1))) unaligned_u64;
    typedef const unaligned_u64 __attribute__((address_space(1))) *gptr8;
    for (int j = lane * 16; j < DC_TILE; j += 64 * 16) {
        const size_t p = base + j;
        if (p + 16 <= n) {
            const size_t lo = n - 16 - p;  // L[lo .. lo + 16) = R[p + 15 .. p] backwards
            const uint64_t w0 = __builtin_bswap64(*(gptr8)(L + lo + 8)), w1 = __builtin_bswap64(*(gptr8)(L + lo));  // R[p .. p+8), R[p+8 .. p+16)
            *reinterpret_cast<uint4 *>(s + PAD + j) = make_uint4(static_cast<uint32_t>(w0), static_cast<uint32_t>(w0 >> 32), static_cast<uint32_t>(w1), static_cast<uint32_t>(w1 >> 32));
        } else {
            for (int b = 0; b < 16; ++b) s[PAD + j + b] = (p + b < n) ? static_cast<uint8_t>(rev_at(L, n, p + b)) : 0;
        }
    }
    if (PAD > 0 && lane == 0) {
        s[PAD - 1] = base > 0 ? static_cast<uint8_t>(rev_at(L, n, base - 1)) : 0;
        s[PAD + DC_TILE] = (base + DC_TILE < n) ? static_cast<uint8_t>(rev_at(L, n, base + DC_TILE)) : 0;
    }
}

__global__ __launch_bounds__(DC_BLOCK) void k_dc_summary(const uint8_t *__restrict__ L, size_t n, size_t ntiles,
                                                          uint32_t *__restrict__ tile_last, uint32_t *__restrict__ tile_lrun,
                                                          uint32_t *__restrict__ tile_runs, uint32_t *__restrict__ tile_syms,
                                                          uint32_t *__restrict__ tile_set) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[DC_WAVES][DC_PAD + DC_TILE + 16];
    // per symbol: (last position inside the tile + 1) << 13 | run index inside the tile + 1 -- both are at most DC_TILE and grow
    // together, so ONE LDS max per run end keeps the pair
    __shared__ uint32_t s_last[DC_WAVES][256];
    static_assert(DC_TILE <= 4096, "position and run index share a 32-bit word, 13 bits each");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t tile = static_cast<size_t>(blockIdx.x) * DC_WAVES + wave;
    if (tile >= ntiles) return;  // whole wave leaves together; no workgroup barrier below
    const size_t base = tile * DC_TILE;
    uint8_t *s = s_tile[wave];
    for (int k = 0; k < 4; ++k) s_last[wave][k * 64 + lane] = 0;
    stage_tile<DC_PAD>(L, n, base, s, lane);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint32_t runs = 0;
    for (int chunk = 0; chunk < DC_TILE / 64; ++chunk) {
        const int j = chunk * 64 + lane;
        const size_t p = base + j;
        const bool valid = p < n;
        const uint32_t c = s[DC_PAD + j];
        const bool start = valid && (p == 0 || c != s[DC_PAD + j - 1]);
        const uint64_t m = __ballot(start);
        // run starts up to and including this lane (v_mbcnt counts the lanes below)
        const uint32_t incl = runs + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u)) + (start ? 1u : 0u);
        // only the last position of a run inside the tile can be the tile's last occurrence of its symbol
        const bool is_end = valid && (p + 1 >= n || j == DC_TILE - 1 || s[DC_PAD + j + 1] != c);
        if (is_end) atomicMax(&s_last[wave][c], (static_cast<uint32_t>(j + 1) << 13) | incl);
        runs += static_cast<uint32_t>(__popcll(m));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t distinct = 0, set4 = 0;
    for (int k = 0; k < 4; ++k) {
        const uint32_t packed = s_last[wave][k * 64 + lane];
        tile_last[tile * 256 + k * 64 + lane] = packed ? static_cast<uint32_t>(base) + (packed >> 13) : 0u;  // position + 1
        tile_lrun[tile * 256 + k * 64 + lane] = packed & 0x1FFFu;
        uint64_t present = __ballot(packed != 0);
        while (present && distinct < 4) {  // (wave-uniform) the first four symbols of the tile, in symbol order
            set4 |= static_cast<uint32_t>(k * 64 + __builtin_ctzll(present)) << (8 * distinct);
            present &= present - 1;
            ++distinct;
        }
        distinct += static_cast<uint32_t>(__popcll(present));
    }
    if (lane == 0) {
        tile_runs[tile] = runs;
        tile_syms[tile] = distinct;  // distinct symbols of the tile: k_dc_main picks its routes by it
        tile_set[tile] = set4;       // and, when there are at most four, which
    }
}

// one workgroup: exclusive sum over tiles of the run counts; total -> mail[0]
__global__ __launch_bounds__(1024) void k_dc_runscan(uint32_t *__restrict__ tile_runs, size_t ntiles, uint32_t *__restrict__ mail) {
    __shared__ uint32_t s_tmp[16 + 1];
    const size_t per = (ntiles + 1023) / 1024;
    const size_t b0 = static_cast<size_t>(threadIdx.x) * per;
    const size_t b1 = b0 + per < ntiles ? b0 + per : ntiles;
    uint32_t sum = 0;
    for (size_t b = b0; b < b1; ++b) sum += tile_runs[b];
    uint32_t total;
    uint32_t run = block_excl_sum<16>(sum, s_tmp, &total);
    for (size_t b = b0; b < b1; ++b) {
        const uint32_t v = tile_runs[b];
        tile_runs[b] = run;
        run += v;
    }
    if (threadIdx.x == 0) mail[0] = total;
}

// carry scan, phase A: last non-empty entry of every symbol within a chunk of tiles.  tile_lrun is turned into a global
// run index + 1 on the way (tile_run_base + local).
__global__ __launch_bounds__(256) void k_dc_carry_a(const uint32_t *__restrict__ tile_last, uint32_t *__restrict__ tile_lrun,
                                                     const uint32_t *__restrict__ tile_run_base, size_t ntiles, size_t tpc,
                                                     uint32_t *__restrict__ chunk_last, uint32_t *__restrict__ chunk_lrun) {
    const size_t g = blockIdx.x, t0 = g * tpc, t1 = t0 + tpc < ntiles ? t0 + tpc : ntiles;
    const int c = threadIdx.x;
    uint32_t lp = 0, lr = 0;
    for (size_t t = t0; t < t1; t += DC_CARRY_BATCH) {  // the loads of a batch are independent of its stores: all in flight together
        uint32_t p[DC_CARRY_BATCH], r[DC_CARRY_BATCH], b[DC_CARRY_BATCH];
#pragma unroll
        for (int k = 0; k < DC_CARRY_BATCH; ++k) {
            const bool in = t + k < t1;
            p[k] = in ? tile_last[(t + k) * 256 + c] : 0u;
            r[k] = in ? tile_lrun[(t + k) * 256 + c] : 0u;
            b[k] = in ? tile_run_base[t + k] : 0u;
        }
#pragma unroll
        for (int k = 0; k < DC_CARRY_BATCH; ++k) {
            if (p[k]) {
                lp = p[k];
                lr = b[k] + r[k];
                tile_lrun[(t + k) * 256 + c] = lr;
            }
        }
    }
    chunk_last[g * 256 + c] = lp;
    chunk_lrun[g * 256 + c] = lr;
}
// phase B (one workgroup of four quarters): exclusive over chunks.  Every quarter of the chunk range first finds its own last
// non-empty entry per symbol, takes the carry of the quarters before it through LDS, and then rewrites its rows.
__global__ __launch_bounds__(1024) void k_dc_carry_b(uint32_t *__restrict__ chunk_last, uint32_t *__restrict__ chunk_lrun, size_t nchunks) {
    __shared__ uint32_t s_lp[4][256], s_lr[4][256];
    const int c = threadIdx.x & 255, q = threadIdx.x >> 8;
    const size_t per = (nchunks + 3) / 4;
    const size_t g0 = static_cast<size_t>(q) * per < nchunks ? static_cast<size_t>(q) * per : nchunks;
    const size_t g1 = g0 + per < nchunks ? g0 + per : nchunks;
    uint32_t lp = 0, lr = 0;
    for (size_t g = g0; g < g1; g += DC_CARRY_BATCH) {
        uint32_t p[DC_CARRY_BATCH], r[DC_CARRY_BATCH];
#pragma unroll
        for (int k = 0; k < DC_CARRY_BATCH; ++k) {
            const bool in = g + k < g1;
            p[k] = in ? chunk_last[(g + k) * 256 + c] : 0u;
            r[k] = in ? chunk_lrun[(g + k) * 256 + c] : 0u;
        }
#pragma unroll
        for (int k = 0; k < DC_CARRY_BATCH; ++k)
            if (p[k]) { lp = p[k]; lr = r[k]; }
    }
    s_lp[q][c] = lp;
    s_lr[q][c] = lr;
    __syncthreads();
    lp = 0;
    lr = 0;
    for (int k = 0; k < q; ++k)
        if (s_lp[k][c]) { lp = s_lp[k][c]; lr = s_lr[k][c]; }
    for (size_t g = g0; g < g1; g += DC_CARRY_BATCH) {
        uint32_t p[DC_CARRY_BATCH], r[DC_CARRY_BATCH];
#pragma unroll
        for (int k = 0; k < DC_CARRY_BATCH; ++k) {
            const bool in = g + k < g1;
            p[k] = in ? chunk_last[(g + k) * 256 + c] : 0u;
            r[k] = in ? chunk_lrun[(g + k) * 256 + c] : 0u;
        }
#pragma unroll
        for (int k = 0; k < DC_CARRY_BATCH; ++k) {
            if (g + k < g1) {
                chunk_last[(g + k) * 256 + c] = lp;
                chunk_lrun[(g + k) * 256 + c] = lr;
                if (p[k]) { lp = p[k]; lr = r[k]; }
            }
        }
    }
}
// phase C: tables become exclusive carries (last occurrence strictly before the tile)
__global__ __launch_bounds__(256) void k_dc_carry_c(uint32_t *__restrict__ tile_last, uint32_t *__restrict__ tile_lrun, size_t ntiles,
                                                     size_t tpc, const uint32_t *__restrict__ chunk_last,
                                                     const uint32_t *__restrict__ chunk_lrun) {
    const size_t g = blockIdx.x, t0 = g * tpc, t1 = t0 + tpc < ntiles ? t0 + tpc : ntiles;
    const int c = threadIdx.x;
    uint32_t lp = chunk_last[g * 256 + c], lr = chunk_lrun[g * 256 + c];
    for (size_t t = t0; t < t1; t += DC_CARRY_BATCH) {
        uint32_t p[DC_CARRY_BATCH], r[DC_CARRY_BATCH];
#pragma unroll
        for (int k = 0; k < DC_CARRY_BATCH; ++k) {
            const bool in = t + k < t1;
            p[k] = in ? tile_last[(t + k) * 256 + c] : 0u;
            r[k] = in ? tile_lrun[(t + k) * 256 + c] : 0u;
        }
#pragma unroll
        for (int k = 0; k < DC_CARRY_BATCH; ++k) {
            if (t + k < t1) {
                tile_last[(t + k) * 256 + c] = lp;
                tile_lrun[(t + k) * 256 + c] = lr;
                if (p[k]) { lp = p[k]; lr = r[k]; }
            }
        }
    }
}

// LDS table of a wave: s_pr (uint2[256]) = {last position + 1, run index + 1} of symbol c at index c.  Lane l also speaks for the
// symbols l, l+64, l+128, l+192 when the whole table is compared against one position (`mine` below).

// One v_writelane_b32: lane `sel` of `old` becomes the (wave-uniform) value.  gfx9 lets a VALU instruction read one SGPR only, so
// the lane select goes through M0 (which does not count); M0 is saved and restored.
__device__ __forceinline__ uint32_t write_lane(uint32_t value, int sel, uint32_t old) {
    uint32_t keep;
    asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                 : "+v"(old), "=&s"(keep)
                 : "s"(value), "s"(sel));
    return old;
}

// Tiles with at most FOUR distinct symbols ({A,C,G,T}; stretches of text made of long runs).  No symbol matching, no dominance count:
// one ballot per symbol of the tile gives, for every lane, the last lane before it that holds that symbol (a masked find-first-bit-high);
// with the wave-uniform carries (last occurrence of each of the four before the chunk) that is the last occurrence of every symbol before
// every lane, and the rank of a run start is the number of the other three whose last occurrence lies behind the run's own previous one.
// Only a symbol's FIRST run start inside the tile (at most four per tile) has its previous occurrence in front of the tile, where symbols
// that do not occur in the tile count as well: those lanes add what the 256-entry table of the tile's start says.
// About 70 vector instructions per chunk instead of 270.
struct FewOut {
    uint32_t *dist; uint8_t *sym; uint8_t *rank; uint32_t *run_end; uint32_t *final_last; uint32_t *final_lrun; uint32_t m;  // m: runs of the block
};
__device__ __forceinline__ void dc_tile_few(const uint8_t *s, uint32_t before_tile, size_t n, size_t base, int lane, uint32_t nsym, uint32_t set4,
                                            const uint2 *pr, uint32_t r, bool last_tile, const FewOut &o) {
    const uint32_t base32 = static_cast<uint32_t>(base);
    const uint64_t lt = lanemask_lt(lane);
    uint32_t symq[4], cpos[4], crun[4], tpos[4];  // wave-uniform: the tile's symbols; their last occurrence (+1) and its run (+1) before the
                                                   // current chunk; the same at the tile's start
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        symq[q] = (set4 >> (8 * q)) & 0xFFu;
        const uint2 e = pr[symq[q]];
        cpos[q] = tpos[q] = static_cast<uint32_t>(q) < nsym ? __builtin_amdgcn_readfirstlane(e.x) : 0u;
        crun[q] = static_cast<uint32_t>(q) < nsym ? __builtin_amdgcn_readfirstlane(e.y) : 0u;
    }
    const uint4 mine = make_uint4(pr[lane].x, pr[64 + lane].x, pr[128 + lane].x, pr[192 + lane].x);  // the table of the tile's start
    for (int chunk = 0; chunk < DC_TILE / 64; ++chunk) {
        const int j = chunk * 64 + lane;
        const size_t p = base + j;
        const bool valid = p < n;
        if (__ballot(valid) == 0) break;
        const uint32_t c = s[j];
        const uint32_t pc = j ? s[j - 1] : before_tile;
        const bool start = valid && (p == 0 || c != pc);
        const uint64_t S = __ballot(start);
        const uint32_t cb1 = base32 + static_cast<uint32_t>(chunk * 64) + 1u;  // (position of lane 0) + 1
        uint64_t M[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) M[q] = static_cast<uint32_t>(q) < nsym ? __ballot(valid && c == symq[q]) : 0ull;
        const uint32_t starts_before = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(S >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(S), 0u));
        const uint32_t run1 = r + starts_before + (start ? 1u : 0u);  // (index of the run holding this lane) + 1
        if (S != 0) {
            int kq = 0;  // which of the tile's symbols this lane holds
#pragma unroll
            for (int q = 1; q < 4; ++q) kq = (static_cast<uint32_t>(q) < nsym && c == symq[q]) ? q : kq;
            uint32_t lp1[4];  // last occurrence (+1) of every symbol of the tile before this lane
            int prevlane = -1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t m = M[q] & lt;
                const int pl = m ? 63 - __builtin_clzll(m) : -1;
                lp1[q] = pl >= 0 ? cb1 + static_cast<uint32_t>(pl) : cpos[q];
                if (q == kq) prevlane = pl;
            }
            const uint32_t b1 = kq == 0 ? lp1[0] : kq == 1 ? lp1[1] : kq == 2 ? lp1[2] : lp1[3];
            uint32_t cnt = 0;  // (a symbol's first occurrence, b1 = 0: every symbol seen so far counts -- the final sweep's rank)
#pragma unroll
            for (int q = 0; q < 4; ++q) cnt += (q != kq && lp1[q] > b1) ? 1u : 0u;
            // first run start of a symbol inside the tile: its previous occurrence (if any) lies in front of the tile
            uint64_t slow = __ballot(start && b1 <= base32);
            while (slow) {
                const int bit = __builtin_ctzll(slow);
                slow &= slow - 1;
                const uint32_t sb1 = __builtin_amdgcn_readlane(b1, bit);
                uint32_t extra = static_cast<uint32_t>(__popcll(__ballot(mine.x > sb1)) + __popcll(__ballot(mine.y > sb1)) +
                                                       __popcll(__ballot(mine.z > sb1)) + __popcll(__ballot(mine.w > sb1)));
#pragma unroll
                for (int q = 0; q < 4; ++q) extra -= tpos[q] > sb1 ? 1u : 0u;  // the tile's own symbols are counted from their occurrences
                cnt = write_lane(static_cast<uint32_t>(__builtin_amdgcn_readlane(cnt, bit)) + extra, bit, cnt);
            }
            const uint32_t prun_in = static_cast<uint32_t>(__shfl(static_cast<int>(run1), prevlane >= 0 ? prevlane : lane, 64));
            if (start) {
                const uint32_t i = base32 + static_cast<uint32_t>(j);
                const uint32_t at = o.m - 1u - (r + starts_before);  // this run's index in L order
                o.sym[at] = static_cast<uint8_t>(c);
                o.dist[at] = i - b1 - cnt;  // = i - b - rank - 1; a first occurrence (b1 = 0): i - rank, the final sweep's n - b - rank - 1
                const uint32_t prun1 = prevlane >= 0 ? prun_in : (kq == 0 ? crun[0] : kq == 1 ? crun[1] : kq == 2 ? crun[2] : crun[3]);
                if (o.rank && b1) o.rank[o.m - prun1] = static_cast<uint8_t>(cnt);  // the rank found at the start of this symbol's next run in L
                if (o.run_end) o.run_end[at] = static_cast<uint32_t>(n - 1) - i;    // where the run ends in L
            }
        }
        // carries: the last lane of every symbol of this chunk (a chunk without a run start continues the open run: same rule)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (M[q]) {  // wave-uniform
                const int l = 63 - __builtin_clzll(M[q]);
                cpos[q] = cb1 + static_cast<uint32_t>(l);
                crun[q] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(run1), l));
            }
        }
        r += static_cast<uint32_t>(__popcll(S));
    }
    if (last_tile) {  // publish the final table (init[s] = n - last position of s - 1: k_dc_init)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint2 e = pr[k * 64 + lane];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (static_cast<uint32_t>(q) < nsym && symq[q] == static_cast<uint32_t>(k * 64 + lane)) e = make_uint2(cpos[q], crun[q]);
            o.final_last[k * 64 + lane] = e.x;
            o.final_lrun[k * 64 + lane] = e.y;
        }
    }
}

// One wave per tile.  Per 64-position chunk (lane = position) the ranks are found lane-parallel:
//   case A  the run's symbol already occurred in this chunk, last at lane w: rank = number of lanes q in (w, lane) that are the
//           first of their symbol inside that window (prevsame[q] <= w) -- a dominance count on the predecessors (wave_dominance), or
//           one ballot per such lane when they are few on a small alphabet;
//   case B  first occurrence of the symbol in this chunk: previous occurrence b from the table (state before the chunk),
//           rank = #{table symbols with last position > b} + #{symbols first seen in this chunk before the lane whose table
//           position is <= b}.  Small alphabets: a short scalar loop over these lanes (at most one per distinct symbol of the chunk),
//           or, when there are many of them, every lane for itself over the whole table.  Large alphabets ("wide" tiles): a bitmap of
//           last occurrences over the tile and the DC_WINDOW positions before it + a second dominance count.
// Then every run-start lane stores its own outputs, and the last lane of every symbol updates the table (and the bitmap).
// The kernel is bound by instruction issue and by the length of its dependent chains, not by memory: a loop
// of a dozen scalar instructions per lane loses against sixty vector instructions for all lanes at once.
__global__ __launch_bounds__(DC_BLOCK) void k_dc_main(const uint8_t *__restrict__ L, size_t n, size_t ntiles,
                                                       const uint32_t *__restrict__ carry_last, const uint32_t *__restrict__ carry_lrun,
                                                       const uint32_t *__restrict__ tile_run_base, const uint32_t *__restrict__ tile_syms,
                                                       const uint32_t *__restrict__ tile_set, uint32_t *__restrict__ dist,
                                                       uint8_t *__restrict__ sym, uint8_t *__restrict__ rank, uint32_t *__restrict__ run_end,
                                                       const uint32_t *__restrict__ total_runs, uint32_t *__restrict__ final_last,
                                                       uint32_t *__restrict__ final_lrun) {
    // 8 KiB of LDS per wave, 32 KiB per workgroup: five workgroups per CU.  The tile has no padding here for that reason (the byte in
    // front of it is kept in a register).
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[DC_WAVES][DC_TILE];
    __shared__ __attribute__((aligned(16))) uint2 s_pr[DC_WAVES][256];
    // tiles that keep the bitmap (`marks`): bit j = position (window start + j) is the last occurrence of its symbol so far.  The window
    // is the tile and the DC_WINDOW positions before it (their marks come from the carry table): nearly every previous occurrence is inside.
    __shared__ __attribute__((aligned(16))) unsigned long long s_bm[DC_WAVES][DC_BM_WORDS];
    __shared__ uint32_t s_first[DC_WAVES][256];  // `wide` tiles: first lane of every symbol of the current chunk (~0 between chunks)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t tile = static_cast<size_t>(blockIdx.x) * DC_WAVES + wave;
    if (tile >= ntiles) return;
    const size_t base = tile * DC_TILE;
    uint8_t *s = s_tile[wave];
    uint2 *pr = s_pr[wave];
    unsigned long long *bm = s_bm[wave];
    uint32_t *first_lane = s_first[wave];
    bm[2 * lane] = 0;
    bm[2 * lane + 1] = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) first_lane[k * 64 + lane] = ~0u;
    // wave-uniform: `marks` = the tile has more than DC_MARKS_TILE distinct symbols (k_dc_summary counted them): the bitmap of last
    // occurrences is kept and serves the chunks with more than DC_MARKS_B first occurrences (text: 0.79 against 1.03 ms for 1e8
    // bytes; {A,C,G,T} would only pay for keeping it); `wide` = more than DC_WIDE_B distinct symbols in the tile's first chunk: the
    // symbols of a chunk are matched through the LDS table
    bool wide = false;
    const bool marks = __builtin_amdgcn_readfirstlane(tile_syms[tile]) > static_cast<uint32_t>(DC_MARKS_TILE);
    stage_tile<0>(L, n, base, s, lane);
    const uint32_t before_tile = base > 0 ? rev_at(L, n, base - 1) : 0u;
    const uint32_t m = __builtin_amdgcn_readfirstlane(*total_runs);  // runs of the block: a run's index in L order is m - 1 - (its index here)
    const uint32_t base32 = static_cast<uint32_t>(base);
    const uint32_t ws = base32 >= DC_WINDOW ? base32 - DC_WINDOW : 0u;  // window start (a multiple of 64)
    const uint32_t tw0 = (base32 - ws) >> 6;                             // bitmap word of the tile's first chunk
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t cp = carry_last[tile * 256 + k * 64 + lane], cr = carry_lrun[tile * 256 + k * 64 + lane];
        pr[k * 64 + lane] = make_uint2(cp, cr);
        if (cp > ws) atomicOr(&bm[(cp - 1u - ws) >> 6], 1ull << ((cp - 1u - ws) & 63u));  // last occurrence before the tile, inside the window
    }
    uint32_t r = __builtin_amdgcn_readfirstlane(tile_run_base[tile]);  // index of the next run to start (wave-uniform)
    const uint64_t lt = lanemask_lt(lane), le = lt | (1ull << lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t nsym = __builtin_amdgcn_readfirstlane(tile_syms[tile]);
    if (nsym <= 4u) {  // wave-uniform: the whole tile takes the four-symbol route
        const FewOut fo{dist, sym, rank, run_end, final_last, final_lrun, m};
        dc_tile_few(s, before_tile, n, base, lane, nsym, __builtin_amdgcn_readfirstlane(tile_set[tile]), pr, r, tile == ntiles - 1, fo);
        return;
    }
    for (int chunk = 0; chunk < DC_TILE / 64; ++chunk) {
        const int j = chunk * 64 + lane;
        const size_t p = base + j;
        const bool valid = p < n;
        const uint32_t c = s[j];
        const uint32_t pc = j ? s[j - 1] : before_tile;
        const bool start = valid && (p == 0 || c != pc);
        const uint64_t S = __ballot(start);
        if (S == 0) {
            // no run starts here: nothing to emit; all valid lanes continue the open run (run r-1), whose table entry must still
            // follow its last position in case the run ends exactly at this chunk's end
            if (valid && (lane == 63 || p + 1 == n)) {
                const uint32_t p1 = base32 + static_cast<uint32_t>(j) + 1u;
                if (marks) {  // the open run's symbol: its mark moves to this chunk's last position
                    const uint32_t old1 = pr[c].x;
                    if (old1 > ws) bm[(old1 - 1u - ws) >> 6] &= ~(1ull << ((old1 - 1u - ws) & 63u));
                    bm[tw0 + chunk] = 1ull << lane;
                }
                pr[c] = make_uint2(p1, r);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        // ---- previous lane with my symbol inside this chunk (-1: none), and the lanes that have such a lane after them
        int prevsame;
        uint64_t notlast;  // wave-uniform
        bool matched = false;
        if (wide) {
            // Large alphabets: nearly every lane is the first of its symbol, a few are the second, hardly any the third.  The first lane
            // of every symbol comes from one LDS min per lane, the second from another min among the rest (its predecessor is the first);
            // what is left after that looks for its predecessor lane by lane.
            if (valid) atomicMin(&first_lane[c], static_cast<uint32_t>(lane));
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t f1 = first_lane[c];
            const bool first = valid && f1 == static_cast<uint32_t>(lane);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (first) first_lane[c] = ~0u;
            const bool later = valid && !first;
            uint32_t f2 = ~0u;
            bool second = false;
            if (__ballot(later)) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (later) atomicMin(&first_lane[c], static_cast<uint32_t>(lane));
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                f2 = first_lane[c];
                second = later && f2 == static_cast<uint32_t>(lane);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (second) first_lane[c] = ~0u;
            }
            prevsame = first ? -1 : static_cast<int>(f1);  // right for the seconds; the others are corrected below
            notlast = __ballot(first && f2 != ~0u);
            uint64_t rest = __ballot(later && !second);
            if (__popcll(rest) <= DC_FEW_LATER) {
                while (rest) {
                    const int bit = __builtin_ctzll(rest);
                    rest &= rest - 1;
                    const uint32_t v = __builtin_amdgcn_readlane(c, bit);
                    const uint64_t m = __ballot(valid && c == v) & ((1ull << bit) - 1ull);  // not empty: `bit` is not the first of its symbol
                    const int w = 63 - __builtin_clzll(m);
                    prevsame = static_cast<int>(write_lane(static_cast<uint32_t>(w), bit, static_cast<uint32_t>(prevsame)));
                    notlast |= 1ull << w;
                }
                matched = true;
            }
        }
        if (!matched) {
            const uint64_t same = wave_match<8>(c, __ballot(valid)).mask();
            const uint64_t before = same & lt;
            prevsame = before ? 63 - __builtin_clzll(before) : -1;
            notlast = __ballot((same & ~le) != 0);
        }
        const uint2 tab = pr[c];                                           // table state BEFORE this chunk: {pos+1, run+1}
        const uint4 mine = make_uint4(pr[lane].x, pr[64 + lane].x, pr[128 + lane].x, pr[192 + lane].x);
        // table entries behind position b: lane l speaks for the symbols l, l+64, l+128, l+192 -- four ballots, or two while no symbol
        // above 127 has been seen (ASCII text)
        const bool upper_half_live = __ballot((mine.z | mine.w) != 0) != 0;
        auto table_after = [&](uint32_t b1) -> uint32_t {
            uint32_t k = static_cast<uint32_t>(__popcll(__ballot(mine.x > b1)) + __popcll(__ballot(mine.y > b1)));
            if (upper_half_live) k += static_cast<uint32_t>(__popcll(__ballot(mine.z > b1)) + __popcll(__ballot(mine.w > b1)));
            return k;
        };
        const bool isA = start && prevsame >= 0;
        const bool isB = start && prevsame < 0;
        // ---- case A: distinct symbols in lanes (w, lane)
        uint32_t cnt = 0;
        uint64_t mA = __ballot(isA);
        if (mA == 0) {
        } else if (!wide && __popcll(mA) <= DC_FEW_A) {
            // few such lanes: one ballot each -- the lanes that are first of their symbol after w, cut to (w, lane).  (On wide tiles the
            // dominance count below is faster even for seven lanes: 60 vector instructions against 7 x 3, but no serial scalar chain.)
            while (mA) {
                const int bit = __builtin_ctzll(mA);
                mA &= mA - 1;
                const int w = __builtin_amdgcn_readlane(prevsame, bit);
                const uint64_t firsts = __ballot(valid && prevsame <= w);
                const uint64_t window = ((1ull << bit) - 1ull) & ~((2ull << w) - 1ull);  // lanes w+1 .. bit-1
                cnt = write_lane(static_cast<uint32_t>(__popcll(firsts & window)), bit, cnt);
            }
        } else {
            // many such lanes (small alphabets): the lanes q < lane whose predecessor lies at or before w are the w + 1 lanes up to w
            // (their predecessor is before themselves) and the firsts inside the window -- a dominance count on key = predecessor + 1
            // (six bits; the keys above zero are distinct, and mine is above zero)
            const uint32_t dom = wave_dominance<6>(static_cast<uint32_t>(prevsame + 1), __ballot(valid) & lt);
            if (isA) cnt = dom - static_cast<uint32_t>(prevsame + 1);
        }
        // ---- case B: first occurrence of the symbol in this chunk; the previous one (if any) is in the table
        const bool first_here = valid && prevsame < 0;
        uint64_t mB = __ballot(isB);
        if (chunk == 0) wide = __popcll(__ballot(first_here)) > DC_WIDE_B;
        if (marks && __popcll(mB) > DC_MARKS_B) {
            // Tile that keeps the bitmap, previous occurrence b inside the window (the tile and the DC_WINDOW positions before it: on
            // random bytes all of them): the distinct symbols in (b, chunk start) are the marked positions of the bitmap in that range
            // -- a prefix popcount and three cross-lane fetches -- plus the symbols first seen in this chunk before the lane whose
            // previous occurrence is not after b (a dominance count).  Previous occurrences before the window take the scalar route.
            const uint32_t b1 = tab.x;
            const bool in_window = b1 > ws;  // previous occurrence inside the window
            const uint4 w2 = *reinterpret_cast<const uint4 *>(bm + 2 * lane);  // bitmap words 2 lane, 2 lane + 1
            const uint32_t pc0 = static_cast<uint32_t>(__popc(w2.x) + __popc(w2.y)), pc1 = static_cast<uint32_t>(__popc(w2.z) + __popc(w2.w));
            const uint32_t incl = wave_incl_sum(pc0 + pc1, lane);
            const uint32_t total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
            const uint32_t lo = in_window ? b1 - ws : 0u;  // window index of the first position after b
            const unsigned long long mw = bm[lo >> 6];
            // marks below that word: those below the pair of words of lane (word / 2), plus that lane's even word for an odd word
            const int owner = static_cast<int>(lo >> 7);
            const uint32_t below_pair = static_cast<uint32_t>(__shfl(static_cast<int>(incl - pc0 - pc1), owner, 64));
            const uint32_t even_word = static_cast<uint32_t>(__shfl(static_cast<int>(pc0), owner, 64));
            const uint32_t below = below_pair + ((lo & 64u) ? even_word : 0u) + static_cast<uint32_t>(__popcll(mw & ((1ull << (lo & 63u)) - 1ull)));
            uint32_t rk = total - below;
            // + the symbols first seen in this chunk in an EARLIER lane whose own previous occurrence is not after mine (their mark sits
            // at or before b, so the bitmap did not count them).  A dominance count over at most 64 lanes: previous occurrences before
            // the tile (or none) always qualify -- one ballot; the in-tile ones are compared bit-sliced (wave_dominance) instead of by
            // a scalar loop over ~57 lanes.
            {
                const uint64_t outside = __ballot(first_here && !in_window);
                const bool memb = first_here && in_window;
                // my previous occurrence is itself a marked position, so `below` is its place among the marks (1 .. 256): an eight-bit key
                // that orders the members as their positions do
                const uint32_t K = memb ? below - 1u : 0u;
                const uint32_t dom = static_cast<uint32_t>(__popcll(outside & lt)) + wave_dominance<8>(K, __ballot(memb) & lt);
                rk += dom;
            }
            if (isB) cnt = in_window ? rk : 0u;
            uint64_t slow = __ballot(isB && b1 != 0 && !in_window);
            while (slow) {
                const int bit = __builtin_ctzll(slow);
                slow &= slow - 1;
                const uint32_t sb1 = __builtin_amdgcn_readlane(tab.x, bit);
                uint32_t srk = table_after(sb1);
                srk += static_cast<uint32_t>(__popcll(__ballot(first_here && tab.x <= sb1) & ((1ull << bit) - 1ull)));
                cnt = write_lane(srk, bit, cnt);
            }
        } else if (__popcll(mB) > DC_WIDE_B) {
            // many first occurrences (large alphabets: random bytes have ~57 per chunk): every lane counts for itself -- the 256 table
            // positions arrive as 128 broadcast 16-byte LDS reads, then the symbols first seen earlier in this chunk
            const uint32_t b1 = tab.x;
            uint32_t rk = 0;
#pragma unroll 4
            for (int t = 0; t < 128; ++t) {
                const uint4 e = *reinterpret_cast<const uint4 *>(pr + 2 * t);  // two entries {position, run}
                rk += (e.x > b1 ? 1u : 0u) + (e.z > b1 ? 1u : 0u);
            }
            uint64_t fm = __ballot(first_here);
            while (fm) {
                const int bit = __builtin_ctzll(fm);
                fm &= fm - 1;
                const uint32_t tq = __builtin_amdgcn_readlane(tab.x, bit);
                rk += (bit < lane && tq <= b1) ? 1u : 0u;
            }
            if (isB) cnt = b1 ? rk : 0u;
        } else {
            while (mB) {
                const int bit = __builtin_ctzll(mB);
                mB &= mB - 1;
                const uint32_t b1 = __builtin_amdgcn_readlane(tab.x, bit);
                uint32_t rk = 0;
                if (b1) {
                    rk = table_after(b1);
                    const uint64_t extra = __ballot(first_here && tab.x <= b1) & ((1ull << bit) - 1ull);
                    rk += static_cast<uint32_t>(__popcll(extra));
                }
                cnt = write_lane(rk, bit, cnt);
            }
        }
        // ---- a symbol's first occurrence (at most one per symbol and block): its rank is the number of distinct symbols so far -- those in the table
        //      and those first seen in this chunk in front of it (the reference's final sweep, see the head of the file)
        for (uint64_t fz = __ballot(isB && tab.x == 0u); fz; fz &= fz - 1) {
            const int bit = __builtin_ctzll(fz);
            const uint32_t seen = table_after(0u) + static_cast<uint32_t>(__popcll(__ballot(first_here && tab.x == 0u) & ((1ull << bit) - 1ull)));
            cnt = write_lane(seen, bit, cnt);
        }
        // ---- outputs of the run-start lanes
        const uint32_t r0 = r;
        const uint32_t starts_before = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(S >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(S), 0u));
        // run starts in lanes 0 .. w, fetched from lane w (lanes without a predecessor read their own)
        const uint32_t starts_upto_w = static_cast<uint32_t>(__shfl(static_cast<int>(starts_before + (start ? 1u : 0u)), prevsame >= 0 ? prevsame : lane, 64));
        if (start) {
            const uint32_t i = base32 + static_cast<uint32_t>(j);
            const uint32_t at = m - 1u - (r0 + starts_before);  // this run's index in L order: a chunk's runs are neighbours there too
            sym[at] = static_cast<uint8_t>(c);
            // previous occurrence b (as b + 1; 0 = none) and the run it closed: lane w of this chunk, or the table's
            const uint32_t b1 = isA ? base32 + static_cast<uint32_t>(chunk * 64 + prevsame) + 1u : tab.x;
            dist[at] = i - b1 - cnt;  // = i - b - rank - 1; a first occurrence (b1 = 0): i - rank, the final sweep's n - b - rank - 1
            if (rank && b1) {  // the rank found at the start of this symbol's next run in L (the run closed at b): its Context.last_rank
                const uint32_t prun = isA ? r0 + starts_upto_w - 1u : tab.y - 1u;
                rank[m - 1u - prun] = static_cast<uint8_t>(cnt);
            }
            if (run_end) run_end[at] = static_cast<uint32_t>(n - 1) - i;  // where the run ends in L
        }
        // ---- table update: the last lane of every symbol of this chunk
        __builtin_amdgcn_wave_barrier();
        const bool last_here = valid && !__builtin_amdgcn_inverse_ballot_w64(notlast);
        if (marks) {  // marks of the symbols of this chunk move to their last lane here
            if (last_here && tab.x > ws) atomicAnd(&bm[(tab.x - 1u - ws) >> 6], ~(1ull << ((tab.x - 1u - ws) & 63u)));
            const uint64_t lm = __ballot(last_here);
            if (lane == 0) bm[tw0 + chunk] = lm;
        }
        if (last_here) {
            const uint32_t p1 = base32 + static_cast<uint32_t>(j) + 1u;
            const uint32_t run1 = r0 + starts_before + (start ? 1u : 0u);  // (index of the run holding this lane) + 1
            pr[c] = make_uint2(p1, run1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        r += static_cast<uint32_t>(__popcll(S));
    }
    if (tile == ntiles - 1) {  // publish the final table (init[s] = n - last position of s - 1: k_dc_init)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            final_last[k * 64 + lane] = pr[k * 64 + lane].x;
            final_lrun[k * 64 + lane] = pr[k * 64 + lane].y;
        }
    }
}

// init[s] = the symbol's first position in L = n - 1 - (its last position in R); n = absent (the reference's convention, src/block/dc.rs:60-66)
__global__ __launch_bounds__(256) void k_dc_init(const uint32_t *__restrict__ final_last, uint32_t n, uint32_t *__restrict__ init) {
    const uint32_t fp = final_last[threadIdx.x];  // (last position + 1, 0 = never seen)
    init[threadIdx.x] = fp ? n - fp : n;
}

}  // namespace

int dc_encode_device(dk_ctx *ctx, const uint8_t *d_bwt, size_t n, uint32_t init_host[256], uint32_t *d_dist, uint8_t *d_sym,
                     uint8_t *d_rank, uint32_t *d_run_end, size_t *m) {
    if (n == 0 || n > 0xFFFFFFF0ull) return ctx->fail(DK_E_ARG, "dc_encode: n out of range");
    hipStream_t st = ctx->stream;
    const size_t mark = ctx->ws_mark();
    const size_t ntiles = div_up(n, DC_TILE);
    const size_t nblocks = div_up(ntiles, DC_WAVES);
    const size_t tpc = div_up(ntiles, DC_MAX_CHUNKS);
    const size_t nchunks = div_up(ntiles, tpc);
    uint32_t *tile_last = ctx->ws_alloc<uint32_t>(ntiles * 256);
    uint32_t *tile_lrun = ctx->ws_alloc<uint32_t>(ntiles * 256);
    uint32_t *tile_runs = ctx->ws_alloc<uint32_t>(ntiles);
    uint32_t *tile_syms = ctx->ws_alloc<uint32_t>(ntiles);
    uint32_t *tile_set = ctx->ws_alloc<uint32_t>(ntiles);
    uint32_t *chunk_last = ctx->ws_alloc<uint32_t>(nchunks * 256);
    uint32_t *chunk_lrun = ctx->ws_alloc<uint32_t>(nchunks * 256);
    uint32_t *d_init = ctx->ws_alloc<uint32_t>(256);
    uint32_t *d_final = ctx->ws_alloc<uint32_t>(512);
    if (!tile_last || !tile_lrun || !tile_runs || !tile_syms || !tile_set || !chunk_last || !chunk_lrun || !d_init || !d_final) return DK_E_NOMEM;
    {
        LaunchScope ls(ctx, K_DC_SUMMARY, 1.0 * n + 2048.0 * ntiles);
        k_dc_summary<<<dim3(nblocks), dim3(DC_BLOCK), 0, st>>>(d_bwt, n, ntiles, tile_last, tile_lrun, tile_runs, tile_syms, tile_set);
    }
    {
        LaunchScope ls(ctx, K_DC_CARRY, 3.0 * 2048.0 * ntiles);
        k_dc_runscan<<<dim3(1), dim3(1024), 0, st>>>(tile_runs, ntiles, ctx->d_mail);
        k_dc_carry_a<<<dim3(nchunks), dim3(256), 0, st>>>(tile_last, tile_lrun, tile_runs, ntiles, tpc, chunk_last, chunk_lrun);
        k_dc_carry_b<<<dim3(1), dim3(1024), 0, st>>>(chunk_last, chunk_lrun, nchunks);
        k_dc_carry_c<<<dim3(nchunks), dim3(256), 0, st>>>(tile_last, tile_lrun, ntiles, tpc, chunk_last, chunk_lrun);
    }
    if (d_rank) DK_HIP(ctx, hipMemsetAsync(d_rank, 0, n, st));  // (a symbol's first run in L has rank 0: nobody writes it)
    {
        LaunchScope ls(ctx, K_DC_MAIN, 1.0 * n + 2048.0 * ntiles);
        k_dc_main<<<dim3(nblocks), dim3(DC_BLOCK), 0, st>>>(d_bwt, n, ntiles, tile_last, tile_lrun, tile_runs, tile_syms, tile_set, d_dist, d_sym, d_rank,
                                                            d_run_end, ctx->d_mail, d_final, d_final + 256);
    }
    {
        LaunchScope ls(ctx, K_DC_INIT, 2048.0);
        k_dc_init<<<dim3(1), dim3(256), 0, st>>>(d_final, static_cast<uint32_t>(n), d_init);
    }
    DK_HIP(ctx, hipGetLastError());
    DK_HIP(ctx, hipMemcpyAsync(ctx->h_mail, ctx->d_mail, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    DK_HIP(ctx, hipMemcpyAsync(ctx->h_mail + 16, d_init, 256 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    DK_HIP(ctx, hipStreamSynchronize(st));
    *m = ctx->h_mail[0];
    for (int s = 0; s < 256; ++s) init_host[s] = ctx->h_mail[16 + s];
    ctx->stats.dc_runs = *m;
    ctx->ws_release(mark);
    return DK_OK;
}

}  // namespace dk
