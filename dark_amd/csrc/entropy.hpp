// Host entropy stage of the product: adaptive range coder, frequency models, the four distance models and the
// block stream layout.  Serial by construction (every coded symbol updates the state the next one reads), so it
// runs on the host, one block per core, fed by the GPU stages.
//
// Replaces: compress::entropy::ari::{Encoder,Decoder,table,bin,apm} as used by src/model/*.rs, the models
// src/model/{dark,exp,ybs,simple,raw}.rs, the stream layout of src/block/dc.rs:53-90,121-151, compress::bwt::dc::decode
// (src/block/dc.rs:146-150) and the in-repo bitwise coder src/entropy/{mod,ari}.rs.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <vector>
#include <cstring>
#include <memory>
#if defined(__SSE4_1__)
#include <immintrin.h>
#endif

#include "../../include/dark_amd.h"

namespace dk {

// ------------------------------------------------------------------------------------------------------------------
// Range coder: 32-bit [low, hi), carry-less, threshold-cut renormalisation, byte output, 4-byte big-endian tail.
// ------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kRangeThreshold = 1u << 14;  // ari::RANGE_DEFAULT_THRESHOLD (dark.rs:149, ybs.rs:69, simple.rs:35 use >>2 of it)
constexpr uint32_t kModelThreshold = kRangeThreshold >> 2;
constexpr uint32_t kTopMask = 0xFF000000u;

struct RangeState {
    uint32_t low = 0, hi = 0xFFFFFFFFu;
    // Narrow to [low + r*from, low + r*to) and renormalise.  The reference loop is
    //     loop { if top bytes differ { if hi-lo > threshold break; cut at the byte border }  emit lo>>24; lo <<= 8; hi <<= 8 }
    // i.e. "shift out every equal leading byte, then stop if the range is wide enough, else cut and go round again".
    // All equal leading bytes leave in one step here (count-leading-zeros of lo^hi), through an unconditional 4-byte store:
    // `out` must have 8 writable bytes.  Returns the number of bytes shifted out (the reference allows at most 4), or -1.
    inline int narrow(uint32_t r, uint32_t from, uint32_t to, uint8_t *out) { return renorm(low + r * from, low + r * to, out); }
    // binary decision under total 4096: [0, zero) codes 0, [zero, 4096) codes 1 -- one multiply, the bit only selects
    inline int narrow_bit12(uint32_t r, uint32_t zero, bool one, uint8_t *out) {
        const uint32_t t = r * zero;
        return renorm(low + (one ? t : 0u), low + (one ? (r << 12) : t), out);
    }
    inline int renorm(uint32_t lo, uint32_t h, uint8_t *out) {
        int shifted = 0;
        for (;;) {
            const uint32_t x = lo ^ h;
            if (__builtin_expect(x == 0, 0)) return -1;  // the interval is empty
            // equal leading bytes = clz / 8; the shift distance 8 * (clz / 8) is clz with its low three bits cleared -- one
            // instruction on the coder's dependency chain (low, hi -> r -> lo, h -> x -> shift -> low, hi) instead of two
            const unsigned sh = static_cast<unsigned>(__builtin_clz(x)) & 24u;
            const uint32_t be = __builtin_bswap32(lo);
            std::memcpy(out + shifted, &be, 4);
            shifted += static_cast<int>(sh >> 3);
            if (shifted > 4) return -1;
            lo <<= sh;
            h <<= sh;
            if (__builtin_expect(h - lo > kRangeThreshold, 1)) break;
            const uint32_t lim = h & kTopMask;
            if (h - lim >= lim - lo) lo = lim; else h = lim - 1;
        }
        low = lo;
        hi = h;
        return shifted;
    }
};

const uint64_t *reciprocal_table14();  // [t] = ceil(2^64 / t) for 2 <= t < 2^14 (entropy.cpp)

class Encoder {
public:
    Encoder(uint8_t *out, size_t cap) : out_(out), cap_(cap), inv_(reciprocal_table14()) {}
    // interval [from,to) out of total
    // The models guarantee from < to <= total < 2^14 (frequencies stay >= 1, binary probabilities stay in [1, 4095]) and the
    // coder keeps hi - low > 2^14 between symbols, so r >= 1: the hot path carries no validity branches.
    inline bool put(uint32_t total, uint32_t from, uint32_t to) {
        // span / total as the high half of span * ceil(2^64 / total) (exact for span < 2^32, total < 2^15: entropy.cpp, the pipeline's events): the
        // table entry is known long before the coder's state is, the division waited for it
        const uint32_t span = rs_.hi - rs_.low;
        const uint32_t r = (total >= 2u && total < (1u << 14)) ? static_cast<uint32_t>((static_cast<unsigned __int128>(span) * inv_[total]) >> 64) : span / total;
        return emit_narrow(r, from, to);
    }
    // same with total == 1 << shift (bin models with threshold 4096, apm::Bit): the division becomes a shift
    inline bool put_pow2(unsigned shift, uint32_t from, uint32_t to) {
        const uint32_t r = (rs_.hi - rs_.low) >> shift;
        return emit_narrow(r, from, to);
    }
    // (measured: two independent multiplies r*from, r*to beat one multiply followed by selects on the coder's critical path)
    inline bool put_bit12(uint32_t zero, bool one) { return put_pow2(12, one ? zero : 0u, one ? 4096u : zero); }
    bool finish() {  // ari::Encoder::finish: code tail = low, big-endian
        if (len_ + 4 > cap_) return fail(DK_E_CAPACITY);
        for (int i = 0; i < 4; ++i) out_[len_++] = static_cast<uint8_t>(rs_.low >> (24 - 8 * i));
        return err_ == 0;
    }
    size_t size() const { return len_; }
    int error() const { return err_; }

private:
    inline bool emit_narrow(uint32_t r, uint32_t from, uint32_t to) {
        if (__builtin_expect(len_ + 8 > cap_, 0)) return slow_narrow(r, from, to);
        const int k = rs_.narrow(r, from, to, out_ + len_);
        if (__builtin_expect(k < 0, 0)) return fail(DK_E_INTERNAL);
        len_ += static_cast<size_t>(k);
        return true;
    }
    bool slow_narrow(uint32_t r, uint32_t from, uint32_t to) {  // within 8 bytes of the end of the caller's buffer
        uint8_t tmp[8];
        const int k = rs_.narrow(r, from, to, tmp);
        if (k < 0) return fail(DK_E_INTERNAL);
        if (len_ + static_cast<size_t>(k) > cap_) return fail(DK_E_CAPACITY);
        std::memcpy(out_ + len_, tmp, static_cast<size_t>(k));
        len_ += static_cast<size_t>(k);
        return true;
    }
    bool fail(int e) { if (!err_) err_ = e; return false; }
    RangeState rs_;
    uint8_t *out_;
    size_t cap_, len_ = 0;
    const uint64_t *inv_;
    int err_ = 0;
};

class Decoder {
public:
    Decoder(const uint8_t *in, size_t len) : in_(in), len_(len) {}
    // The reference computes offset = (code - low) / r and searches the model for it.  floor(x / r) >= c  <=>  x >= c * r, so the
    // search can compare x = code - low with border * r instead: same symbol, no second division.
    inline void begin(uint32_t total) {
        feed();
        r_ = (rs_.hi - rs_.low) / total;
        x_ = code_ - rs_.low;
        if (r_ == 0 || static_cast<uint64_t>(x_) >= static_cast<uint64_t>(r_) * total) fail(DK_E_STREAM);
    }
    inline void begin_pow2(unsigned shift) {
        feed();
        r_ = (rs_.hi - rs_.low) >> shift;
        x_ = code_ - rs_.low;
        if (r_ == 0 || (static_cast<uint64_t>(x_) >> shift) >= r_) fail(DK_E_STREAM);
    }
    // true when the decoded offset lies below `border` (i.e. offset < border)
    inline bool below(uint32_t border) const { return static_cast<uint64_t>(x_) < static_cast<uint64_t>(r_) * border; }
    // binary decision under total 4096 in one step: offset >= zero  <=>  x >= r * zero
    inline bool bit12(uint32_t zero, bool &one) {
        feed();
        const uint32_t r = (rs_.hi - rs_.low) >> 12;
        const uint32_t x = code_ - rs_.low;
        if (__builtin_expect(r == 0 || (x >> 12) >= r, 0)) return fail(DK_E_STREAM);
        one = x >= r * zero;
        uint8_t scratch[8];
        const int k = rs_.narrow_bit12(r, zero, one, scratch);
        if (__builtin_expect(k < 0, 0)) return fail(DK_E_STREAM);
        pending_ = k;
        return true;
    }
    uint32_t x() const { return x_; }
    uint32_t r() const { return r_; }
    inline bool take(uint32_t from, uint32_t to) {
        uint8_t scratch[8];
        const int k = rs_.narrow(r_, from, to, scratch);
        if (k < 0) return fail(DK_E_STREAM);
        pending_ = k;
        return err_ == 0;
    }
    bool finish() { feed(); return err_ == 0; }  // ari::Decoder::finish consumes the pending tail bytes
    size_t consumed() const { return pos_; }      // bytes read so far: after finish() = exactly what the encoder wrote
    int error() const { return err_; }
    bool fail(int e) { if (!err_) err_ = e; return false; }

private:
    inline void feed() {
        const unsigned k = static_cast<unsigned>(pending_);
        // k == 0 (no byte left the coder in the last step) is frequent and irregular: it goes through the same arithmetic
        // (shift by 32 - 0) rather than through a branch the predictor cannot learn
        if (__builtin_expect(pos_ + 4 <= len_, 1)) {  // take k (0..4) bytes out of one unconditional 4-byte big-endian load
            uint32_t be;
            std::memcpy(&be, in_ + pos_, 4);
            const uint64_t w = __builtin_bswap32(be);
            code_ = static_cast<uint32_t>(((static_cast<uint64_t>(code_) << 32) | w) >> (32 - 8 * k));
            pos_ += k;
            pending_ = 0;
            return;
        }
        while (pending_) {
            uint8_t b = 0;
            if (pos_ < len_) b = in_[pos_++]; else fail(DK_E_STREAM);  // read_u8 Err -> the reference unwraps (panics)
            code_ = (code_ << 8) + b;
            --pending_;
        }
    }
    RangeState rs_;
    const uint8_t *in_;
    size_t len_, pos_ = 0;
    uint32_t code_ = 0, r_ = 0, x_ = 0;
    int pending_ = 4;
    int err_ = 0;
};

// ------------------------------------------------------------------------------------------------------------------
// Two-thread encode: the models never read the coder's state, so one thread can run the (serial) modelling and hand the
// resulting intervals to a second thread that runs the (serial) range coder.  Same intervals in the same order -> the same
// bytes.  EventSink has the put/put_pow2 interface of Encoder; EventPipe is the single-producer single-consumer ring.
// ------------------------------------------------------------------------------------------------------------------
// The interval stream between the two threads, in 16-bit units: a binary decision under total 4096 (4 of 5 events) is ONE unit
// 1 b 00 zzzzzzzzzzzz (b = coded bit, z = probability of zero); any other interval is three units from, to, total (< 2^15).
class EventPipe;
class EventSink {
public:
    explicit EventSink(EventPipe &p);
    inline bool put(uint32_t total, uint32_t from, uint32_t to) {
        if (!(from < to && to <= total && total < 32768u)) return fail(DK_E_INTERNAL);
        if (fill_ + 3 > cap_) flush();
        cur_[fill_] = static_cast<uint16_t>(from);
        cur_[fill_ + 1] = static_cast<uint16_t>(to);
        cur_[fill_ + 2] = static_cast<uint16_t>(total);
        fill_ += 3;
        return true;
    }
    inline bool put_pow2(unsigned shift, uint32_t from, uint32_t to) { return put(1u << shift, from, to); }  // same quotient as the shift
    inline bool put_bit12(uint32_t zero, bool one) {
        if (fill_ + 1 > cap_) flush();
        cur_[fill_++] = static_cast<uint16_t>(0x8000u | (one ? 0x4000u : 0u) | zero);
        return true;
    }
    bool finish();  // flush the last batch and mark the end of the stream
    int error() const { return err_; }
    bool fail(int e) { if (!err_) err_ = e; return false; }

private:
    void flush();
    EventPipe &pipe_;
    uint16_t *cur_ = nullptr;
    uint32_t fill_ = 0, cap_ = 0;
    int slot_ = 0;
    int err_ = 0;
};

// ------------------------------------------------------------------------------------------------------------------
// Frequency models: ari::table::Model (u16 counts, cut threshold, halving with round-up), ari::bin::Model
// (probability of zero out of a fixed total, shift-rate adaptation), their SumProxy mixers, apm::Bit.
// ------------------------------------------------------------------------------------------------------------------
template <int N>
struct FreqTable {
    uint32_t total;
    uint16_t f[N];
    void flat() { for (int i = 0; i < N; ++i) f[i] = 1; total = N; while (total >= kModelThreshold) halve(); }
    void halve() { total = 0; for (int i = 0; i < N; ++i) { f[i] = static_cast<uint16_t>((f[i] + 1) >> 1); total += f[i]; } }
    inline void bump(size_t v, unsigned add_log, uint32_t add_const) {
        const uint32_t add = (total >> add_log) + add_const;
        f[v] = static_cast<uint16_t>(f[v] + add);
        total += add;
        if (total >= kModelThreshold) halve();
    }
    inline uint32_t below(size_t v) const { uint32_t s = 0; for (size_t i = 0; i < v; ++i) s += f[i]; return s; }
    template <class E> bool encode(E &e, size_t v) const { const uint32_t lo = below(v); return e.put(total, lo, lo + f[v]); }
    bool decode(Decoder &d, size_t &v) const {
        d.begin(total);
        if (d.error()) return false;
        uint32_t lo = 0, hi = f[0];
        size_t k = 0;
        while (!d.below(hi)) {
            if (++k >= static_cast<size_t>(N)) return d.fail(DK_E_STREAM);
            lo = hi;
            hi += f[k];
        }
        v = k;
        return d.take(lo, hi);
    }
};

#if defined(__SSE4_1__)
// The 8-entry tables of the dark model as one 16-byte vector.  Same arithmetic, but every update is a whole-vector store and every
// read a whole-vector load: the same table is updated for one distance and read again for the next, and a 16-byte load that follows
// a 2-byte store to the same line cannot be forwarded from the store buffer (it waits for the store to retire).
namespace simd8 {
alignas(16) inline constexpr uint16_t kOneHot[8][8] = {{0xFFFF, 0, 0, 0, 0, 0, 0, 0}, {0, 0xFFFF, 0, 0, 0, 0, 0, 0}, {0, 0, 0xFFFF, 0, 0, 0, 0, 0},
                                                      {0, 0, 0, 0xFFFF, 0, 0, 0, 0}, {0, 0, 0, 0, 0xFFFF, 0, 0, 0}, {0, 0, 0, 0, 0, 0xFFFF, 0, 0},
                                                      {0, 0, 0, 0, 0, 0, 0xFFFF, 0}, {0, 0, 0, 0, 0, 0, 0, 0xFFFF}};
alignas(16) inline constexpr uint16_t kBelow[9][8] = {{0, 0, 0, 0, 0, 0, 0, 0},
                                                     {0xFFFF, 0, 0, 0, 0, 0, 0, 0},
                                                     {0xFFFF, 0xFFFF, 0, 0, 0, 0, 0, 0},
                                                     {0xFFFF, 0xFFFF, 0xFFFF, 0, 0, 0, 0, 0},
                                                     {0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0, 0, 0, 0},
                                                     {0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0, 0, 0},
                                                     {0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0, 0},
                                                     {0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0},
                                                     {0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF}};
inline __m128i one_hot(size_t v) { return _mm_load_si128(reinterpret_cast<const __m128i *>(kOneHot[v])); }
inline __m128i below(size_t v) { return _mm_load_si128(reinterpret_cast<const __m128i *>(kBelow[v])); }
// sum of the eight 16-bit lanes; every lane must be below 2^15 (all frequencies here are below 2^14)
inline uint32_t hsum(__m128i x) {
    __m128i s = _mm_madd_epi16(x, _mm_set1_epi16(1));
    s = _mm_add_epi32(s, _mm_shuffle_epi32(s, 0x4E));
    s = _mm_add_epi32(s, _mm_shuffle_epi32(s, 0xB1));
    return static_cast<uint32_t>(_mm_cvtsi128_si32(s));
}
}  // namespace simd8

template <>
struct FreqTable<8> {
    alignas(16) uint16_t f[8];
    uint32_t total;
    __m128i vec() const { return _mm_load_si128(reinterpret_cast<const __m128i *>(f)); }
    void set(__m128i v) { _mm_store_si128(reinterpret_cast<__m128i *>(f), v); }
    void flat() { set(_mm_set1_epi16(1)); total = 8; while (total >= kModelThreshold) halve(); }
    void halve() {
        const __m128i h = _mm_srli_epi16(_mm_add_epi16(vec(), _mm_set1_epi16(1)), 1);
        set(h);
        total = simd8::hsum(h);
    }
    inline void bump(size_t v, unsigned add_log, uint32_t add_const) {
        const uint32_t add = (total >> add_log) + add_const;
        set(_mm_add_epi16(vec(), _mm_and_si128(simd8::one_hot(v), _mm_set1_epi16(static_cast<short>(add)))));
        total += add;
        if (total >= kModelThreshold) halve();
    }
    inline uint32_t below(size_t v) const { return simd8::hsum(_mm_and_si128(vec(), simd8::below(v))); }
    template <class E> bool encode(E &e, size_t v) const { const uint32_t lo = below(v); return e.put(total, lo, lo + f[v]); }
    bool decode(Decoder &d, size_t &v) const {
        d.begin(total);
        if (d.error()) return false;
        uint32_t lo = 0, hi = f[0];
        size_t k = 0;
        while (!d.below(hi)) {
            if (++k >= 8) return d.fail(DK_E_STREAM);
            lo = hi;
            hi += f[k];
        }
        v = k;
        return d.take(lo, hi);
    }
};

template <class E>
inline bool encode_mix12(E &e, const FreqTable<8> &a, const FreqTable<8> &b, size_t v) {
    const __m128i bv = b.vec();
    const __m128i w = _mm_add_epi16(a.vec(), _mm_add_epi16(bv, bv));  // a + 2 b per symbol, below 3 * 2^12 + slack
    const uint32_t lo = simd8::hsum(_mm_and_si128(w, simd8::below(v)));
    const uint32_t fv = simd8::hsum(_mm_and_si128(w, simd8::one_hot(v)));
    return e.put(a.total + 2u * b.total, lo, lo + fv);
}
inline bool decode_mix12(Decoder &d, const FreqTable<8> &a, const FreqTable<8> &b, size_t &v) {
    d.begin(a.total + 2u * b.total);
    if (d.error()) return false;
    const __m128i bv = b.vec();
    __m128i w = _mm_add_epi16(a.vec(), _mm_add_epi16(bv, bv));
    w = _mm_add_epi16(w, _mm_slli_si128(w, 2));  // inclusive prefix sums of the eight lanes (the last one is the total, < 2^16)
    w = _mm_add_epi16(w, _mm_slli_si128(w, 4));
    w = _mm_add_epi16(w, _mm_slli_si128(w, 8));
    alignas(16) uint16_t incl[8];
    _mm_store_si128(reinterpret_cast<__m128i *>(incl), w);
    // symbol = number of cumulative borders at or below the offset: x >= r * incl[i], i = 0..6 (x < r * incl[7] was checked by begin())
    const uint64_t x = d.x(), r = d.r();
    size_t k = 0;
#pragma GCC unroll 8
    for (int i = 0; i < 7; ++i) k += x >= r * incl[i] ? 1u : 0u;
    v = k;
    return d.take(k ? incl[k - 1] : 0u, incl[k]);
}
#endif

// table::SumProxy::new(1, a, 2, b, 0): the only weighting the reference uses (dark.rs:194,243)
template <int N, class E>
inline bool encode_mix12(E &e, const FreqTable<N> &a, const FreqTable<N> &b, size_t v) {
    uint32_t lo = 0;
#pragma GCC unroll 8
    for (size_t i = 0; i < static_cast<size_t>(N); ++i) lo += i < v ? a.f[i] + 2u * b.f[i] : 0u;  // fixed trip count: no data-dependent branch
    return e.put(a.total + 2u * b.total, lo, lo + a.f[v] + 2u * b.f[v]);
}
template <int N>
inline bool decode_mix12(Decoder &d, const FreqTable<N> &a, const FreqTable<N> &b, size_t &v) {
    d.begin(a.total + 2u * b.total);
    if (d.error()) return false;
    // symbol = number of cumulative borders at or below the offset; all N compares are independent (no search loop to mispredict)
    uint32_t cum[N + 1];
    cum[0] = 0;
#pragma GCC unroll 8
    for (int i = 0; i < N; ++i) cum[i + 1] = cum[i] + a.f[i] + 2u * b.f[i];
    const uint64_t x = d.x(), r = d.r();
    size_t k = 0;
#pragma GCC unroll 8
    for (int i = 1; i < N; ++i) k += x >= r * cum[i] ? 1u : 0u;  // x < r * cum[N] was checked by begin()
    v = k;
    return d.take(cum[k], cum[k + 1]);
}

struct BinFreq {  // total is always kModelThreshold = 1 << 12 in the reference's models
    static constexpr unsigned kShift = 12;
    uint32_t zero;
    void flat() { zero = kModelThreshold >> 1; }
    template <unsigned RATE> inline void learn(bool one) {  // branch-free: the bit is data, not control
        const uint32_t down = zero >> RATE, up = (kModelThreshold - zero) >> RATE;
        zero += one ? (0u - down) : up;
    }
};
static_assert((1u << BinFreq::kShift) == kModelThreshold, "bin total must be a power of two");
template <class E>
inline bool encode_bit_p(E &e, uint32_t zero, bool one) {
    return e.put_bit12(zero, one);
}
inline bool decode_bit_p(Decoder &d, uint32_t zero, bool &one) {
    return d.bit12(zero, one);
}

// ------------------------------------------------------------------------------------------------------------------
// Distance models.  Interface = src/model/mod.rs:32-39 (reset / encode / decode); only Context.symbol matters to
// them (dark.rs:184, exp.rs:61, ybs.rs:98).
// ------------------------------------------------------------------------------------------------------------------
inline unsigned bit_length(uint32_t v) { return v ? 32u - static_cast<unsigned>(__builtin_clz(v)) : 0u; }

class DarkModel {  // src/model/dark.rs
public:
    DarkModel() { reset(); }
    void reset();
    template <class E> bool encode(uint32_t dist, uint8_t symbol, E &e);
    bool decode(uint8_t symbol, Decoder &d, uint32_t &dist);
    // encode() = exponent, modelled mantissa bits, flat mantissa bits.  The last part reads no model state at all, so a second
    // thread that knows the distance can produce those decisions by itself (two-thread encode, entropy.cpp).
    template <class E> bool encode_exponent(uint32_t dist, uint8_t symbol, E &e);
    template <class E> bool encode_mantissa_modelled(uint32_t dist, E &e);
    // encode_exponent() in two halves that share no state (five-stage pipeline, entropy.cpp): the table decision mixes a per-symbol table a
    // and a global table b as a + 2 b, which is linear -- one thread can emit (total, below, frequency) of a, another of b, and whoever adds
    // them up gets the decision encode_exponent() would have coded.  Both halves follow avg_dist of every symbol by themselves (adapt()).
    // E::half(total, below, freq) / E::half_bit(zero) take what the halves produce.
    template <class E> bool encode_exponent_symbol_half(uint32_t dist, uint8_t symbol, E &e);
    template <class E> bool encode_exponent_global_half(uint32_t dist, uint8_t symbol, E &e);
    template <class E> static bool encode_mantissa_flat(uint32_t dist, E &e);  // stateless
    // number of binary decisions encode_exponent emits after its one table decision
    static unsigned exponent_bits(uint32_t dist) { const unsigned log = bit_length(dist + 1); return log >= 8 ? log - 7 : 0; }
    static unsigned modelled_mantissa_bits(uint32_t dist) { const unsigned log = bit_length(dist + 1); return log > 3 ? 3 : log - 1; }

private:
    struct PerSymbol { int64_t avg_dist; FreqTable<8> log_freq; BinFreq extra[32]; };
    inline void adapt(PerSymbol &c, uint32_t dist, int log_diff);
    FreqTable<8> log_global_[12][3];
    BinFreq log_bits_[2][32];
    BinFreq mantissa_[32][4];
    PerSymbol sym_[256];
    unsigned last_token_;
};

class ExpModel {  // src/model/exp.rs
public:
    ExpModel() { reset(); }
    void reset();
    template <class E> bool encode(uint32_t dist, uint8_t symbol, E &e);
    bool decode(uint8_t symbol, Decoder &d, uint32_t &dist);

private:
    static uint32_t log_fixed(uint32_t d);
    uint32_t avg_log_[256];
    uint16_t prob_[10][24];
};

class YbsModel {  // src/model/ybs.rs
public:
    YbsModel() { reset(); }
    void reset();
    template <class E> bool encode(uint32_t dist, uint8_t symbol, E &e);
    bool decode(uint8_t symbol, Decoder &d, uint32_t &dist);

private:
    struct PerSymbol { uint32_t avg_log, last_diff; };
    static void track(PerSymbol &c, uint32_t log);
    FreqTable<14> low_[13];
    FreqTable<19> high_;
    BinFreq rest_[3];
    PerSymbol sym_[256];
};

class SimpleModel {  // src/model/simple.rs
public:
    SimpleModel() { reset(); }
    void reset();
    template <class E> bool encode(uint32_t dist, uint8_t symbol, E &e);
    bool decode(uint8_t symbol, Decoder &d, uint32_t &dist);

private:
    FreqTable<256> freq_[4];
};

// ------------------------------------------------------------------------------------------------------------------
// Block stream (src/block/dc.rs) and friends
// ------------------------------------------------------------------------------------------------------------------
struct DcStream {          // what the GPU DC stage hands to the entropy stage
    size_t n = 0;          // block size
    const uint32_t *init = nullptr;     // [256] first position per symbol, n = absent
    const uint32_t *dist = nullptr;     // [m]
    const uint8_t *sym = nullptr;       // [m]
    const uint8_t *rank = nullptr;      // [m] Context.last_rank   (RAWDC only)
    const uint32_t *run_end = nullptr;  // [m] position of each entry (RAWDC only: distance_limit = n - pos)
    size_t m = 0;
    uint32_t origin = 0;
    // may be null: dist / sym are still arriving from the GPU -- *ready = the number of entries that are there (grows to m); every
    // thread that walks the stream waits at this frontier (write_stream)
    const std::atomic<size_t> *ready = nullptr;
    unsigned stall_ms = 0;  // a frontier that does not move for this long ends the pass with DK_E_HIP (0 = 20 s)
};
constexpr size_t DC_STREAM_POISON = ~static_cast<size_t>(0);  // *ready: the producer gave up, nothing more will arrive
// src/block/dc.rs:53-90: init-table RLE header, distances, origin, finish
// host_threads: 0 = automatic (two threads for large blocks when a partner core sharing the L3 can be pinned), 1 = one thread
int encode_block_stream(int model_id, const DcStream &s, uint8_t *out, size_t cap, size_t *out_len, int host_threads = 0);
int last_entropy_threads();  // threads the calling thread's last encode_block_stream used
int last_entropy_group();    // and the last-level-cache group (lowest cpu number in it) that pass claimed; -1: none (one thread)
int last_entropy_group_numa();  // ... and that group's memory node (-1: none claimed / unknown)
void set_preferred_numa(int node);  // calling thread: the memory node its next coding passes should claim their L3 group on (-1: no preference)
std::vector<size_t> l3_claim_order(const std::vector<int> &numa, size_t own, int preferred);  // (entropy.cpp: the order groups are tried in)
int set_entropy_thread_mode(int mode);  // 0 automatic | 1 | 2 | 4, process-wide (overrides DK_ENTROPY_THREADS)
int host_l3_groups(int min_cores);      // L3 groups with at least min_cores cores usable by the calling thread
// src/block/dc.rs:121-151: header, dc::decode pulling model.decode, origin.  *single = 1 when the block has a
// one-symbol alphabet (the reference then mis-reads origin; see DESIGN.md "Reference quirks").
int decode_block_stream(int model_id, const uint8_t *in, size_t in_len, size_t n, uint8_t *bwt_out,
                        uint32_t *origin, int *single, size_t *consumed = nullptr);
// compress::bwt::dc::decode fed from an array
int dc_decode_array(const uint32_t init[256], const uint32_t *dist, size_t m, uint8_t *bwt_out, size_t n, size_t *consumed);
// model-level helpers (src/model/mod.rs:59-76)
int model_encode_stream(int model_id, const uint32_t *dist, const uint8_t *sym, size_t m, uint8_t *out, size_t cap, size_t *out_len);
int model_decode_stream(int model_id, const uint8_t *in, size_t in_len, const uint8_t *sym, size_t m, uint32_t *dist);
// src/entropy/{mod,ari}.rs
int bitcoder_encode(const uint8_t *bits, const uint16_t *flat, size_t nbits, uint8_t *out, size_t cap, size_t *out_len);
int bitcoder_decode(const uint8_t *in, size_t in_len, const uint16_t *flat, size_t nbits, uint8_t *bits);
// block::raw with the bbb model (src/block/raw.rs, src/model/bbb.rs; bbb.cpp -- gates restated after etc/bbb/main.cpp, PARITY UNPINNED)
int raw_bbb_encode_stream(const uint8_t *bwt, size_t n, uint32_t origin, uint8_t *out, size_t cap, size_t *out_len);
int raw_bbb_decode_stream(const uint8_t *in, size_t in_len, size_t n, uint8_t *bwt, uint32_t *origin, size_t *consumed = nullptr);
// largest n a model can code without losing bits (0 = unknown model)
uint64_t model_max_block(int model_id);

}  // namespace dk
