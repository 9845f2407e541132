// Small wave64 / workgroup primitives shared by the kernels (gfx950: 64-wide wavefronts).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace dk {

constexpr int kWave = 64;

__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// DPP controls (gfx9): a lane reads the value of another lane of its row of 16, or the last lane of the row(s) before its own.  A lane
// whose source does not exist, or whose row is masked out, gets `identity`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_fetch(uint32_t v, uint32_t identity) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(identity), static_cast<int>(v), CTRL, ROW_MASK, 0xf, false));
}
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118, kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;

// inclusive scan across the 64 lanes of a wave: four steps inside the rows of 16, then the row totals (six DPP adds, no LDS traffic)
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v, int /*lane*/) {
    v += dpp_fetch<kDppRowShr1, 0xf>(v, 0);
    v += dpp_fetch<kDppRowShr2, 0xf>(v, 0);
    v += dpp_fetch<kDppRowShr4, 0xf>(v, 0);
    v += dpp_fetch<kDppRowShr8, 0xf>(v, 0);
    v += dpp_fetch<kDppRowBcast15, 0xa>(v, 0);  // rows 1 and 3 take the total of the row before
    v += dpp_fetch<kDppRowBcast31, 0xc>(v, 0);  // rows 2 and 3 take the total of rows 0-1
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v, int /*lane*/) {
    uint32_t t;
    t = dpp_fetch<kDppRowShr1, 0xf>(v, 0); v = v > t ? v : t;
    t = dpp_fetch<kDppRowShr2, 0xf>(v, 0); v = v > t ? v : t;
    t = dpp_fetch<kDppRowShr4, 0xf>(v, 0); v = v > t ? v : t;
    t = dpp_fetch<kDppRowShr8, 0xf>(v, 0); v = v > t ? v : t;
    t = dpp_fetch<kDppRowBcast15, 0xa>(v, 0); v = v > t ? v : t;
    t = dpp_fetch<kDppRowBcast31, 0xc>(v, 0); v = v > t ? v : t;
    return v;
}

// A set of lanes as two 32-bit halves in vector registers (a per-lane 64-bit value costs two instructions per operation either way;
// keeping the halves apart lets every step be the one instruction it is).
struct LaneSet {
    uint32_t lo, hi;
    __device__ __forceinline__ uint32_t count() const { return static_cast<uint32_t>(__popc(lo) + __popc(hi)); }
    // members below this lane (v_mbcnt: population count under the lane's own "lanes before me" mask)
    __device__ __forceinline__ uint32_t before() const { return __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u)); }
    __device__ __forceinline__ uint64_t mask() const { return (static_cast<uint64_t>(hi) << 32) | lo; }
};
// wave64 match-any: the lanes of `among` that hold the same BITS-bit value as this lane.  One ballot per bit; a lane keeps the lanes
// that agree with it at that bit (xnor of the ballot with its own bit spread over a word): six vector instructions per bit.
template <int BITS>
__device__ __forceinline__ LaneSet wave_match(uint32_t v, uint64_t among) {
    LaneSet s{static_cast<uint32_t>(among), static_cast<uint32_t>(among >> 32)};
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        const uint32_t ones = static_cast<uint32_t>(static_cast<int32_t>(v << (31 - b)) >> 31);  // all ones where my bit b is set
        const uint64_t B = __ballot(ones != 0);
        s.lo &= ~(static_cast<uint32_t>(B) ^ ones);
        s.hi &= ~(static_cast<uint32_t>(B >> 32) ^ ones);
    }
    return s;
}

// For every lane: how many EARLIER lanes (bit set in `earlier` = a set of lanes below mine) hold a smaller key.  The keys of the lanes that
// matter are distinct numbers below 2^BITS; a lane holding my own key is not counted.  Bit-sliced from the top: E = the earlier lanes that
// agree with me on the bits seen so far; at a bit where I have a one, those of them with a zero are smaller -- they leave E into U, the
// others that differ leave E uncounted; one population count at the end.  About ten vector instructions per bit, nothing per lane pair.
template <int BITS>
__device__ __forceinline__ uint32_t wave_dominance(uint32_t key, uint64_t earlier) {
    uint32_t e_lo = static_cast<uint32_t>(earlier), e_hi = static_cast<uint32_t>(earlier >> 32), u_lo = 0, u_hi = 0;
#pragma unroll
    for (int k = BITS - 1; k >= 0; --k) {
        const uint32_t ones = static_cast<uint32_t>(static_cast<int32_t>(key << (31 - k)) >> 31);  // all ones where my bit k is set
        const uint64_t B = __ballot(ones != 0);
        const uint32_t b_lo = static_cast<uint32_t>(B), b_hi = static_cast<uint32_t>(B >> 32);
        u_lo |= e_lo & ~b_lo & ones;
        u_hi |= e_hi & ~b_hi & ones;
        e_lo &= ~(b_lo ^ ones);
        e_hi &= ~(b_hi ^ ones);
    }
    return static_cast<uint32_t>(__popc(u_lo) + __popc(u_hi));
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
