// Small wave64 / workgroup primitives shared by the kernels (gfx950: 64-wide wavefronts).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace dk {

constexpr int kWave = 64;

__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        uint32_t t = __shfl_up(v, o, kWave);
        if (lane >= o) v += t;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        uint32_t t = __shfl_up(v, o, kWave);
        if (lane >= o) v = v > t ? v : t;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        uint32_t t = __shfl_xor(v, o, kWave);
        v = v > t ? v : t;
    }
    return v;
}

// Exclusive sum over the threads of a workgroup of NW waves.  `s_tmp` must hold NW + 1 words.  Returns the exclusive
// prefix of `v`; *total (if non-null) receives the workgroup sum.  Contains two __syncthreads().
template <int NW>
__device__ __forceinline__ uint32_t block_excl_sum(uint32_t v, uint32_t *s_tmp, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_sum(v, lane);
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    uint32_t base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const uint32_t t = s_tmp[w];
        if (w < wave) base += t;
        all += t;
    }
    if (total) *total = all;
    __syncthreads();
    return base + inc - v;
}
// Exclusive running max (identity 0) over the threads of a workgroup.
template <int NW>
__device__ __forceinline__ uint32_t block_excl_max(uint32_t v, uint32_t *s_tmp, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_max(v, lane);
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    uint32_t base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const uint32_t t = s_tmp[w];
        if (w < wave) base = base > t ? base : t;
        all = all > t ? all : t;
    }
    if (total) *total = all;
    __syncthreads();
    uint32_t prev = __shfl_up(inc, 1, kWave);
    if (lane == 0) prev = 0;
    return base > prev ? base : prev;
}

}  // namespace dk
