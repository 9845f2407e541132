// `bbb` coding model (src/model/bbb.rs) for block::raw::{Encoder,Decoder} (src/block/raw.rs:35-104): the BWT bytes themselves are coded,
// bit by bit, MSB first, each under a probability refined by five gate stages.
//
// What is pinned and what is not: the model's structure -- the bit-history state map, the five contexts, the way
// the stages are mixed, the update order -- is src/model/bbb.rs, followed here.  The gates themselves (compress::entropy::ari::apm::
// Gate / Bit) are not in the reference tree; they are built after the model's in-repo analogue, Mahoney's bbb (etc/bbb/main.cpp:348-460):
// logistic stretch / squash with a 33-point table, 33 bins of 16-bit probabilities per gate, linear interpolation, both bins updated.
// Assumptions G1-G5 are listed in oracle/dark_oracle.c; the oracle makes the same ones, so agreement with it says nothing about the
// Rust crate.  PARITY UNPINNED: streams made here decode here.
#include <cstdlib>
#include <memory>
#include <new>

#include "entropy.hpp"

namespace dk {
namespace {

constexpr uint8_t kStates[256 * 4] = {
#include "bbb_states.inc"
};
inline uint8_t next_state(unsigned state, unsigned bit) { return kStates[state * 4 + bit]; }

// logistic tables: squash(d) = 4096 / (1 + e^-d) for d in 1/256 units (33-point interpolation), stretch = its inverse on 12-bit inputs
struct Logistic {
    int16_t stretch[4096];
    static int squash(int d) {
        static const int16_t knots[33] = {1, 2, 3, 6, 10, 16, 27, 45, 73, 120, 194, 310, 488, 747, 1101, 1546, 2047, 2549, 2994, 3348, 3607,
                                          3785, 3901, 3975, 4022, 4050, 4068, 4079, 4085, 4089, 4092, 4093, 4094};
        if (d > 2047) return 4095;
        if (d < -2047) return 0;
        const int frac = d & 127, k = (d >> 7) + 16;
        return (knots[k] * (128 - frac) + knots[k + 1] * frac + 64) >> 7;
    }
    Logistic() {
        int done = 0;
        for (int x = -2047; x <= 2047; ++x) {
            const int p = squash(x);
            while (done <= p) stretch[done++] = static_cast<int16_t>(x);
        }
        stretch[4095] = 2047;
    }
};
const Logistic &logistic() {
    static const Logistic t;
    return t;
}

struct Gate {  // 33 bins across the stretched domain
    uint16_t bin[33];
    struct Where { uint8_t lo, weight; };
    uint32_t pass(uint32_t flat, Where &w) const {
        const int s = logistic().stretch[flat];
        w.weight = static_cast<uint8_t>(s & 127);
        w.lo = static_cast<uint8_t>((s + 2048) >> 7);
        return (static_cast<uint32_t>(bin[w.lo]) * (128u - w.weight) + static_cast<uint32_t>(bin[w.lo + 1]) * w.weight) >> 11;
    }
    void learn(bool one, Where w, int rate) {
        const int shift = rate + 4;
        const int target = one ? 0 : (65536 + (1 << shift) - 2);  // probabilities are of a ZERO bit
        for (int k = 0; k < 2; ++k) {
            uint16_t &b = bin[w.lo + k];
            b = static_cast<uint16_t>(b + ((target - static_cast<int>(b)) >> shift));
        }
    }
};

class BbbModel {
public:
    BbbModel() {
        Gate fresh;
        for (int j = 0; j < 33; ++j) fresh.bin[j] = static_cast<uint16_t>(Logistic::squash((j - 16) * 128) * 16);
        for (auto &g : g1a_) g = fresh;
        for (auto &g : g1b_) g = fresh;
        for (auto &g : g2_) g = fresh;
        for (auto &g : g3_) g = fresh;
        for (auto &g : g4_) g = fresh;
        for (auto &g : g5_) g = fresh;
        for (int i = 0; i < 256; ++i) {  // bbb.rs:108-126
            size_t n0 = kStates[i * 4 + 2], n1 = kStates[i * 4 + 3];
            if (n0 == 0) n1 <<= 7;
            if (n1 == 0) n0 <<= 7;
            prob_of_state_[i] = static_cast<uint16_t>(((n0 + 1) << 16) / (n0 + n1 + 2));
            state_of_ctx_[i] = 0;
        }
    }
    template <class E> bool put(uint8_t sym, E &e) {  // bbb.rs:288-297
        for (int i = 7; i >= 0; --i) {
            const bool one = (sym >> i) & 1u;
            Trail t;
            const uint32_t zero = predict(t);
            if (!encode_bit_p(e, zero, one)) return false;
            learn(one, i == 0, t);
        }
        return true;
    }
    bool get(Decoder &d, uint8_t &sym) {  // bbb.rs:299-310
        unsigned v = 0;
        for (int i = 7; i >= 0; --i) {
            Trail t;
            const uint32_t zero = predict(t);
            bool one;
            if (!decode_bit_p(d, zero, one)) return false;
            v = (v << 1) | (one ? 1u : 0u);
            learn(one, i == 0, t);
        }
        sym = static_cast<uint8_t>(v);
        return true;
    }

private:
    struct Trail { Gate::Where w1a, w1b, w2, w3, w4, w5; uint32_t c2, c3, c4, c5; };
    uint32_t predict(Trail &t) const {  // bbb.rs:232-280
        const uint32_t p0 = prob_of_state_[state_of_ctx_[ctx_]] >> 4;
        const uint32_t p1 = (g1a_[partial_].pass(p0, t.w1a) + g1b_[partial_].pass(p0, t.w1b) + 1) >> 1;
        t.c2 = partial_ | ((history_ & 0xFFu) << 8);
        const uint32_t p2 = g2_[t.c2].pass(p1, t.w2);
        t.c3 = (history_ & 0xFFu) | run_ctx_;
        const uint32_t p3 = g3_[t.c3].pass(p2, t.w3);
        t.c4 = partial_ | (history_ & 0x1F00u);
        const uint32_t p4 = (g4_[t.c4].pass(p3, t.w4) * 3 + p3 + 2) >> 2;
        t.c5 = static_cast<uint32_t>(((static_cast<uint64_t>(partial_ ^ (history_ & 0xFFFFFFu)) * 123456791ull) & 0xFFFFFFFFull) >> 18);
        const uint32_t p5 = (g5_[t.c5].pass(p4, t.w5) + p4 + 1) >> 1;
        return p5 < 2048 ? p5 + 1 : p5;  // bbb.rs:268-274 (G5)
    }
    void learn(bool one, bool byte_done, const Trail &t) {  // bbb.rs:197-230
        const unsigned bit = one ? 1u : 0u;
        const uint32_t c1 = partial_;
        partial_ = ((partial_ & 0x7Fu) << 1) | bit;
        if (byte_done) {
            history_ = ((history_ & 0xFFFFFFu) << 8) | partial_;
            partial_ = 1;
            if (((history_ ^ (history_ >> 8)) & 0xFFu) == 0) {
                if (run_len_ < 0xFFFFu) ++run_len_;
                if (run_len_ == 1 || run_len_ == 2 || run_len_ == 4) run_ctx_ += 0x100u;
            } else {
                run_len_ = 0;
                run_ctx_ = 0;
            }
        }
        const unsigned state = state_of_ctx_[ctx_];  // bbb.rs:128-137
        state_of_ctx_[ctx_] = next_state(state, bit);
        prob_of_state_[state] = static_cast<uint16_t>((0xFFu * prob_of_state_[state] + ((1u - bit) << 16) + 0x80u) >> 8);
        ctx_ = partial_;
        g1a_[c1].learn(one, t.w1a, 1);
        g1b_[c1].learn(one, t.w1b, 5);
        g2_[t.c2].learn(one, t.w2, 3);
        g3_[t.c3].learn(one, t.w3, 4);
        g4_[t.c4].learn(one, t.w4, 3);
        g5_[t.c5].learn(one, t.w5, 3);
    }
    uint32_t ctx_ = 0, partial_ = 1, history_ = 0, run_len_ = 0, run_ctx_ = 0;
    uint8_t state_of_ctx_[256];
    uint16_t prob_of_state_[256];
    Gate g1a_[0x100], g1b_[0x100], g2_[0x10000], g3_[0x400], g4_[0x2000], g5_[0x4000];
};

}  // namespace

// block::raw::Encoder::encode after the BWT (src/block/raw.rs:45-58): origin as four symbols, then every BWT byte, finish
int raw_bbb_encode_stream(const uint8_t *bwt, size_t n, uint32_t origin, uint8_t *out, size_t cap, size_t *out_len) {
    std::unique_ptr<BbbModel> m(new (std::nothrow) BbbModel());
    if (!m) return DK_E_NOMEM;
    Encoder e(out, cap);
    bool ok = true;
    for (int k = 3; k >= 0 && ok; --k) ok = m->put(static_cast<uint8_t>(origin >> (8 * k)), e);
    for (size_t i = 0; i < n && ok; ++i) ok = m->put(bwt[i], e);
    ok = ok && e.finish();
    if (!ok) return e.error() ? e.error() : DK_E_INTERNAL;
    *out_len = e.size();
    return DK_OK;
}
// block::raw::Decoder::decode up to the BWT (src/block/raw.rs:85-97)
int raw_bbb_decode_stream(const uint8_t *in, size_t in_len, size_t n, uint8_t *bwt, uint32_t *origin, size_t *consumed) {
    std::unique_ptr<BbbModel> m(new (std::nothrow) BbbModel());
    if (!m) return DK_E_NOMEM;
    Decoder d(in, in_len);
    uint32_t o = 0;
    uint8_t b;
    for (int k = 0; k < 4; ++k) {
        if (!m->get(d, b)) return DK_E_STREAM;
        o = (o << 8) | b;
    }
    for (size_t i = 0; i < n; ++i)
        if (!m->get(d, bwt[i])) return DK_E_STREAM;
    if (!d.finish()) return DK_E_STREAM;
    *origin = o;
    if (consumed) *consumed = d.consumed();
    return DK_OK;
}

}  // namespace dk
