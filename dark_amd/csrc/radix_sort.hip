// Stable LSD radix sort of (u64 key, u32 value) pairs, 8 bits per pass -- the workhorse of the suffix sort.
// It takes the place of all the sorting work inside saca() (src/saca.rs:270-340): bucket placement, the induced
// sorts and the naming pass become "sort by packed prefix, then by (group, rank of the suffix h further on)".
//
// Per pass (v1, three phases, no inter-workgroup spinning):
//   k_radix_hist     one tile (4096 pairs) per workgroup -> 256-bin histogram in LDS -> tile_hist[tile][256]
//   k_radix_scan_*   digit-major exclusive scan of tile_hist (chunk sums / chunk scan / apply), rows stay coalesced
//   k_radix_scatter  re-reads the tile, ranks keys stably with wave64 ballots (match-any over the 8 digit bits),
//                    reorders the tile in LDS so that equal digits are contiguous, then writes runs to HBM
// Algorithmic bytes per pass: hist 8 B/key read; scatter 12 B/pair read + 12 B/pair written.
#include "context.hpp"
#include "device_util.hpp"

namespace dk {
namespace {

constexpr int RS_BLOCK = 256;
constexpr int RS_WAVES = RS_BLOCK / 64;
constexpr int RS_KPT = 16;                   // pairs per thread
constexpr int RS_TILE = RS_BLOCK * RS_KPT;   // 4096 pairs per workgroup
constexpr int RS_MAX_CHUNKS = 256;

__device__ __forceinline__ uint32_t digit_of(uint64_t k, int shift) { return static_cast<uint32_t>(k >> shift) & 0xFFu; }

__global__ __launch_bounds__(RS_BLOCK) void k_radix_hist(const uint64_t *__restrict__ keys, size_t n, int shift,
                                                          uint32_t *__restrict__ tile_hist) {
    __shared__ uint32_t h[256];
    const int tid = threadIdx.x;
    h[tid] = 0;
    __syncthreads();
    const size_t base = static_cast<size_t>(blockIdx.x) * RS_TILE;
#pragma unroll
    for (int k = 0; k < RS_KPT; ++k) {
        const size_t i = base + static_cast<size_t>(k) * RS_BLOCK + tid;
        if (i < n) atomicAdd(&h[digit_of(keys[i], shift)], 1u);
    }
    __syncthreads();
    tile_hist[static_cast<size_t>(blockIdx.x) * 256 + tid] = h[tid];
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
// phase B (one workgroup): digit-major exclusive scan of the chunk sums
__global__ __launch_bounds__(256) void k_radix_scan_b(uint32_t *__restrict__ chunk_sum, size_t nchunks) {
    __shared__ uint32_t s_tmp[RS_WAVES + 1];
    const int d = threadIdx.x;
    uint32_t run = 0;
#pragma unroll 8
    for (size_t g = 0; g < nchunks; ++g) {
        const uint32_t v = chunk_sum[g * 256 + d];
        chunk_sum[g * 256 + d] = run;
        run += v;
    }
    const uint32_t base = block_excl_sum<RS_WAVES>(run, s_tmp, nullptr);
#pragma unroll 8
    for (size_t g = 0; g < nchunks; ++g) chunk_sum[g * 256 + d] += base;
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

__global__ __launch_bounds__(RS_BLOCK) void k_radix_scatter(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                             uint64_t *__restrict__ kout, uint32_t *__restrict__ vout, size_t n,
                                                             int shift, const uint32_t *__restrict__ tile_offs) {
    __shared__ uint64_t s_keys[RS_TILE];          // 32 KiB: tile of keys in digit order; reused for the values
    __shared__ uint32_t s_cnt[RS_WAVES][256];     // per-wave digit counters, then exclusive over waves
    __shared__ uint32_t s_start[256];             // tile-local start of each digit
    __shared__ uint32_t s_gbase[256];             // global offset of the digit minus its tile-local start
    __shared__ uint32_t s_tmp[RS_WAVES + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t tile_base = static_cast<size_t>(blockIdx.x) * RS_TILE;
    const size_t left = n - tile_base;
    const uint32_t valid = left < static_cast<size_t>(RS_TILE) ? static_cast<uint32_t>(left) : RS_TILE;

    for (int i = tid; i < RS_WAVES * 256; i += RS_BLOCK) (&s_cnt[0][0])[i] = 0;

    // wave w owns pairs [w*1024, (w+1)*1024) of the tile, lane-striped so that loads coalesce
    uint64_t key[RS_KPT];
    uint32_t val[RS_KPT];
    const uint32_t wbase = static_cast<uint32_t>(wave) * (64 * RS_KPT);
#pragma unroll
    for (int k = 0; k < RS_KPT; ++k) {
        const uint32_t li = wbase + k * 64 + lane;
        if (li < valid) {
            key[k] = kin[tile_base + li];
            val[k] = vin[tile_base + li];
        } else {
            key[k] = ~0ull;  // padding sorts behind every real pair of the tile and is never written
            val[k] = 0;
        }
    }
    __syncthreads();

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
        const uint32_t start = block_excl_sum<RS_WAVES>(run, s_tmp, nullptr);
        s_start[d] = start;
        s_gbase[d] = tile_offs[static_cast<size_t>(blockIdx.x) * 256 + d] - start;
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

}  // namespace

// Sorts `count` pairs on key bits [begin_bit, end_bit).  keys/vals are the input buffers, *_alt equally sized scratch;
// on return `keys` and `vals` refer to whichever buffer holds the sorted data (the references are swapped per pass).
int sort_pairs(dk_ctx *ctx, uint64_t *&keys, uint64_t *&keys_alt, uint32_t *&vals, uint32_t *&vals_alt, size_t count,
               int begin_bit, int end_bit) {
    if (count <= 1 || end_bit <= begin_bit) return DK_OK;
    if (count > 0xFFFFFFFEull) return ctx->fail(DK_E_ARG, "sort_pairs: count too large");
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
            k_radix_hist<<<dim3(ntiles), dim3(RS_BLOCK), 0, st>>>(keys, count, shift, tile_hist);
        }
        {
            LaunchScope ls(ctx, K_RADIX_SCAN, 3.0 * 1024.0 * ntiles);
            k_radix_scan_a<<<dim3(nchunks), dim3(256), 0, st>>>(tile_hist, ntiles, tiles_per_chunk, chunk_sum);
            k_radix_scan_b<<<dim3(1), dim3(256), 0, st>>>(chunk_sum, nchunks);
            k_radix_scan_c<<<dim3(nchunks), dim3(256), 0, st>>>(tile_hist, ntiles, tiles_per_chunk, chunk_sum);
        }
        {
            LaunchScope ls(ctx, K_RADIX_SCATTER, 24.0 * count);
            k_radix_scatter<<<dim3(ntiles), dim3(RS_BLOCK), 0, st>>>(keys, vals, keys_alt, vals_alt, count, shift, tile_hist);
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

}  // namespace dk
