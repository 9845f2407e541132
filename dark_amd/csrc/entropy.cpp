// Host entropy stage: see entropy.hpp for what each piece replaces in the reference.
#include "entropy.hpp"

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>
#if defined(__linux__)
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>
#endif

namespace dk {

// ------------------------------------------------------------------------------------------------------------------
// dark model (src/model/dark.rs)
// ------------------------------------------------------------------------------------------------------------------
namespace {
constexpr unsigned kMaxLogCode = 8;      // dark.rs:46
constexpr unsigned kMaxLogContext = 11;  // dark.rs:47
constexpr unsigned kMaxBitContext = 3;   // dark.rs:49
constexpr int kAdaptPowers[9] = {6, 5, 4, 3, 2, 1, 4, 6, 4};  // dark.rs:50
constexpr unsigned kLogGlobalRate = 12, kLogSymbolRate = 5;   // dark.rs:141-142
constexpr uint32_t kLogAdd = 5;                               // dark.rs:143
inline unsigned token_of(unsigned log) { return log < 2 ? 0u : (log < 8 ? 1u : 2u); }  // dark.rs:214
}  // namespace

void DarkModel::reset() {  // dark.rs:121-146,160-178
    for (auto &row : log_global_) for (auto &t : row) t.flat();
    for (auto &row : log_bits_) for (auto &b : row) b.flat();
    for (auto &row : mantissa_) for (auto &b : row) b.flat();
    for (auto &c : sym_) {
        c.avg_dist = 1000;
        c.log_freq.flat();
        for (auto &b : c.extra) b.flat();
    }
    last_token_ = 1;
}

inline void DarkModel::adapt(PerSymbol &c, uint32_t dist, int log_diff) {  // dark.rs:94-101
    const int64_t a = log_diff < -6 ? 7 : (log_diff >= 3 ? 3 : kAdaptPowers[6 + log_diff]);
    c.avg_dist += (a * (static_cast<int64_t>(dist) - c.avg_dist)) >> 3;  // floor, like isize >> 3
}

template <class E>
bool DarkModel::encode_exponent(uint32_t dist, uint8_t symbol, E &e) {  // dark.rs:180-214,229-231
    if (dist >= 0x7FFFFFFFu) return false;  // log would reach 32: freq_mantissa has 32 rows
    const unsigned log = bit_length(dist + 1);
    PerSymbol &c = sym_[symbol];
    const unsigned avg_log = std::min(kMaxLogContext, bit_length(static_cast<uint32_t>(c.avg_dist)));
    {
        FreqTable<8> &g = log_global_[avg_log][last_token_];
        const size_t code = std::min(log, kMaxLogCode) - 1;
        if (!encode_mix12(e, c.log_freq, g, code)) return false;
        c.log_freq.bump(code, kLogSymbolRate, kLogAdd);
        g.bump(code, kLogGlobalRate, kLogAdd);
    }
    if (log >= kMaxLogCode) {  // unary extension, one mixed binary decision per extra bit of exponent
        BinFreq *gb = log_bits_[avg_log == kMaxLogContext ? 1 : 0];
        for (unsigned i = 0; i + kMaxLogCode < log; ++i) {
            if (!encode_bit_p(e, (c.extra[i].zero + gb[i].zero) >> 1, true)) return false;
            c.extra[i].learn<3>(true);
            gb[i].learn<2>(true);
        }
        const unsigned i = log - kMaxLogCode;
        if (!encode_bit_p(e, (c.extra[i].zero + gb[i].zero) >> 1, false)) return false;
        c.extra[i].learn<3>(false);
        gb[i].learn<2>(false);
    }
    last_token_ = token_of(log);
    adapt(c, dist, static_cast<int>(log) - static_cast<int>(avg_log));
    return true;
}
static_assert(kMaxLogCode == 8, "DarkModel::exponent_bits assumes MAX_LOG_CODE = 8");

template <class E>
bool DarkModel::encode_exponent_symbol_half(uint32_t dist, uint8_t symbol, E &e) {  // the per-symbol state of encode_exponent(): log_freq, extra, avg_dist
    if (dist >= 0x7FFFFFFFu) return false;
    const unsigned log = bit_length(dist + 1);
    PerSymbol &c = sym_[symbol];
    const unsigned avg_log = std::min(kMaxLogContext, bit_length(static_cast<uint32_t>(c.avg_dist)));
    const size_t code = std::min(log, kMaxLogCode) - 1;
    e.half(c.log_freq.total, c.log_freq.below(code), c.log_freq.f[code]);
    c.log_freq.bump(code, kLogSymbolRate, kLogAdd);
    if (log >= kMaxLogCode) {
        for (unsigned i = 0; i + kMaxLogCode <= log; ++i) {  // log - 7 decisions: ones, then a zero
            e.half_bit(c.extra[i].zero);
            c.extra[i].learn<3>(i + kMaxLogCode < log);
        }
    }
    adapt(c, dist, static_cast<int>(log) - static_cast<int>(avg_log));
    return true;
}
template <class E>
bool DarkModel::encode_exponent_global_half(uint32_t dist, uint8_t symbol, E &e) {  // the shared state: log_global_, log_bits_, last_token_ (+ its own avg_dist)
    if (dist >= 0x7FFFFFFFu) return false;
    const unsigned log = bit_length(dist + 1);
    PerSymbol &c = sym_[symbol];  // (only avg_dist of it is used here)
    const unsigned avg_log = std::min(kMaxLogContext, bit_length(static_cast<uint32_t>(c.avg_dist)));
    FreqTable<8> &g = log_global_[avg_log][last_token_];
    const size_t code = std::min(log, kMaxLogCode) - 1;
    e.half(g.total, g.below(code), g.f[code]);
    g.bump(code, kLogGlobalRate, kLogAdd);
    if (log >= kMaxLogCode) {
        BinFreq *gb = log_bits_[avg_log == kMaxLogContext ? 1 : 0];
        for (unsigned i = 0; i + kMaxLogCode <= log; ++i) {
            e.half_bit(gb[i].zero);
            gb[i].learn<2>(i + kMaxLogCode < log);
        }
    }
    last_token_ = token_of(log);
    adapt(c, dist, static_cast<int>(log) - static_cast<int>(avg_log));
    return true;
}

template <class E>
bool DarkModel::encode_mantissa_modelled(uint32_t dist, E &e) {  // dark.rs:216-224: the first three bits below the leading one
    if (dist >= 0x7FFFFFFFu) return false;
    const uint32_t v = dist + 1;
    const unsigned log = bit_length(v);
    BinFreq *mc = mantissa_[log];
    const unsigned modelled = log > kMaxBitContext ? kMaxBitContext : log - 1;
    for (unsigned i = 1; i <= modelled; ++i) {
        const bool bit = (v >> (log - i - 1)) & 1u;
        if (!encode_bit_p(e, mc[i - 1].zero, bit)) return false;
        mc[i - 1].learn<8>(bit);
    }
    return true;
}
template <class E>
bool DarkModel::encode_mantissa_flat(uint32_t dist, E &e) {  // dark.rs:225-227: the rest through the never-updated 4th model
    const uint32_t v = dist + 1;
    const unsigned log = bit_length(v);
    for (unsigned i = kMaxBitContext + 1; i < log; ++i)
        if (!encode_bit_p(e, kModelThreshold >> 1, (v >> (log - i - 1)) & 1u)) return false;
    return true;
}
static_assert(kMaxBitContext == 3, "DarkModel::modelled_mantissa_bits assumes MAX_BIT_CONTEXT = 3");

template <class E>
bool DarkModel::encode(uint32_t dist, uint8_t symbol, E &e) {  // dark.rs:180-232
    return encode_exponent(dist, symbol, e) && encode_mantissa_modelled(dist, e) && encode_mantissa_flat(dist, e);
}

bool DarkModel::decode(uint8_t symbol, Decoder &d, uint32_t &dist) {  // dark.rs:234-287
    PerSymbol &c = sym_[symbol];
    const unsigned avg_log = std::min(kMaxLogContext, bit_length(static_cast<uint32_t>(c.avg_dist)));
    unsigned log;
    {
        FreqTable<8> &g = log_global_[avg_log][last_token_];
        size_t code;
        if (!decode_mix12(d, c.log_freq, g, code)) return false;
        c.log_freq.bump(code, kLogSymbolRate, kLogAdd);
        g.bump(code, kLogGlobalRate, kLogAdd);
        log = static_cast<unsigned>(code) + 1;
    }
    if (log >= kMaxLogCode) {
        BinFreq *gb = log_bits_[avg_log == kMaxLogContext ? 1 : 0];
        unsigned count = 0;
        for (;;) {
            if (count >= 32) return d.fail(DK_E_STREAM);
            bool bit;
            if (!decode_bit_p(d, (c.extra[count].zero + gb[count].zero) >> 1, bit)) return false;
            c.extra[count].learn<3>(bit);
            gb[count].learn<2>(bit);
            if (!bit) break;
            ++count;
        }
        log += count;
    }
    if (log >= 32) return d.fail(DK_E_STREAM);
    last_token_ = token_of(log);
    BinFreq *mc = mantissa_[log];
    uint32_t v = 1;
    for (unsigned i = 1; i < log; ++i) {
        bool bit;
        if (i > kMaxBitContext) {
            if (!decode_bit_p(d, mc[kMaxBitContext].zero, bit)) return false;
        } else {
            if (!decode_bit_p(d, mc[i - 1].zero, bit)) return false;
            mc[i - 1].learn<8>(bit);
        }
        v = (v << 1) + (bit ? 1u : 0u);
    }
    dist = v - 1;
    adapt(c, dist, static_cast<int>(log) - static_cast<int>(avg_log));
    return d.error() == 0;
}

// ------------------------------------------------------------------------------------------------------------------
// exp model (src/model/exp.rs) on apm::Bit (12-bit probability of zero; update(value, 5, 0))
// ------------------------------------------------------------------------------------------------------------------
namespace {
constexpr uint32_t kFixedBase = 8, kFixedMask = (1u << kFixedBase) - 1;
inline void apm_learn(uint16_t &p, bool one) {  // apm::Bit::update(value, rate = 5, bias = 0)
    if (one) p = static_cast<uint16_t>(p - (p >> 5)); else p = static_cast<uint16_t>(p + ((4096 - p) >> 5));
}
}  // namespace

void ExpModel::reset() {  // exp.rs:47-56
    for (auto &a : avg_log_) a = 1u << kFixedBase;
    for (auto &row : prob_) for (auto &p : row) p = 2048;
}
uint32_t ExpModel::log_fixed(uint32_t d) {  // exp.rs:34-43 with its integer quirks (8/3 == 2, 8>>12 == 0)
    if (d <= 2) return d << kFixedBase;
    if (d <= 4) return (3u << kFixedBase) + (d & 1) * (kFixedBase >> 1);
    if (d <= 7) return (4u << kFixedBase) + ((d - 5) % 3) * (kFixedBase / 3);
    if (d <= 12) return (5u << kFixedBase) + (d & 3) * (kFixedBase >> 2);
    return (6u << kFixedBase) + (d - 12) * (kFixedBase >> 12);
}
template <class E>
bool ExpModel::encode(uint32_t dist, uint8_t symbol, E &e) {  // exp.rs:58-78
    const uint32_t log = avg_log_[symbol];
    const uint32_t w2 = log & kFixedMask, w1 = kFixedMask + 1 - w2;
    uint16_t *m1 = prob_[log >> kFixedBase], *m2 = prob_[(log >> kFixedBase) + 1];
    for (int i = 23; i >= 0; --i) {  // only 24 bits are coded: higher bits of dist are dropped (exp.rs:22,67)
        const bool bit = (dist >> i) & 1u;
        const uint32_t flat = (w1 * m1[i] + w2 * m2[i]) >> kFixedBase;
        if (!encode_bit_p(e, flat, bit)) return false;
        apm_learn(m1[i], bit);
        apm_learn(m2[i], bit);
    }
    avg_log_[symbol] = (3 * log + log_fixed(dist)) >> 2;
    return true;
}
bool ExpModel::decode(uint8_t symbol, Decoder &d, uint32_t &dist) {  // exp.rs:80-101
    const uint32_t log = avg_log_[symbol];
    const uint32_t w2 = log & kFixedMask, w1 = kFixedMask + 1 - w2;
    uint16_t *m1 = prob_[log >> kFixedBase], *m2 = prob_[(log >> kFixedBase) + 1];
    uint32_t v = 0;
    for (int i = 23; i >= 0; --i) {
        const uint32_t flat = (w1 * m1[i] + w2 * m2[i]) >> kFixedBase;
        bool bit;
        if (!decode_bit_p(d, flat, bit)) return false;
        apm_learn(m1[i], bit);
        apm_learn(m2[i], bit);
        v += v + (bit ? 1u : 0u);
    }
    avg_log_[symbol] = (3 * log + log_fixed(v)) >> 2;
    dist = v;
    return true;
}

// ------------------------------------------------------------------------------------------------------------------
// ybs model (src/model/ybs.rs)
// ------------------------------------------------------------------------------------------------------------------
void YbsModel::reset() {  // ybs.rs:54-65,75-87
    for (auto &t : low_) t.flat();
    high_.flat();
    for (auto &b : rest_) b.flat();
    for (auto &c : sym_) c.avg_log = c.last_diff = 0;
}
void YbsModel::track(PerSymbol &c, uint32_t log) {  // ybs.rs:33-38
    const uint32_t a = c.last_diff > 3 ? 2 : 1;
    c.last_diff = log > c.avg_log ? log - c.avg_log : c.avg_log - log;
    c.avg_log = (a * log + c.avg_log) / (a + 1);
}
template <class E>
bool YbsModel::encode(uint32_t dist, uint8_t symbol, E &e) {  // ybs.rs:89-127
    constexpr uint32_t kMaxLow = 12;
    const uint32_t group = dist < 4 ? dist : bit_length(dist) + 1;
    PerSymbol &c = sym_[symbol];
    FreqTable<14> &t = low_[std::min(c.avg_log, kMaxLow)];
    const uint32_t coded = std::min(group, kMaxLow);
    if (!t.encode(e, coded)) return false;
    t.bump(coded, 10, 1);
    track(c, coded);
    if (group < 4) return true;
    if (group >= kMaxLow) {
        const uint32_t add = group - kMaxLow;
        if (add >= 19) return false;  // table_high has 32-13 entries: the reference would index out of bounds
        if (!high_.encode(e, add)) return false;
        high_.bump(add, 10, 1);
    }
    const uint32_t log = group - 1;
    for (uint32_t i = 1; i < log; ++i) {
        const bool bit = (dist >> (log - i - 1)) & 1u;
        if (i >= 3) {
            if (!encode_bit_p(e, rest_[2].zero, bit)) return false;  // un-updated last bin (ybs.rs:119)
        } else {
            if (!encode_bit_p(e, rest_[i - 1].zero, bit)) return false;
            rest_[i - 1].learn<5>(bit);
        }
    }
    return true;
}
bool YbsModel::decode(uint8_t symbol, Decoder &d, uint32_t &dist) {  // ybs.rs:129-165
    constexpr uint32_t kMaxLow = 12;
    PerSymbol &c = sym_[symbol];
    FreqTable<14> &t = low_[std::min(c.avg_log, kMaxLow)];
    size_t coded;
    if (!t.decode(d, coded)) return false;
    track(c, static_cast<uint32_t>(coded));
    t.bump(coded, 10, 1);
    if (coded < 4) { dist = static_cast<uint32_t>(coded); return true; }
    uint32_t group = static_cast<uint32_t>(coded);
    if (coded == kMaxLow) {
        size_t add;
        if (!high_.decode(d, add)) return false;
        high_.bump(add, 10, 1);
        group = kMaxLow + static_cast<uint32_t>(add);
    }
    const uint32_t log = group - 1;
    uint32_t v = 1;
    for (uint32_t i = 1; i < log; ++i) {
        bool bit;
        if (i >= 3) {
            if (!decode_bit_p(d, rest_[2].zero, bit)) return false;
        } else {
            if (!decode_bit_p(d, rest_[i - 1].zero, bit)) return false;
            rest_[i - 1].learn<5>(bit);
        }
        v = (v << 1) + (bit ? 1u : 0u);
    }
    dist = v;
    return true;
}

// ------------------------------------------------------------------------------------------------------------------
// simple model (src/model/simple.rs)
// ------------------------------------------------------------------------------------------------------------------
namespace { constexpr unsigned kSimpleUp[4] = {10, 8, 7, 6}; }  // simple.rs:38
void SimpleModel::reset() { for (auto &t : freq_) t.flat(); }
template <class E>
bool SimpleModel::encode(uint32_t dist, uint8_t, E &e) {  // simple.rs:51-64
    const uint32_t head = std::min<uint32_t>(0xFF, dist);
    if (!freq_[0].encode(e, head)) return false;
    freq_[0].bump(head, kSimpleUp[0], 1);
    if (head == 0xFF) {
        const uint32_t rest = dist - 0xFF;
        for (int i = 0; i < 3; ++i) {
            const uint32_t b = (rest >> (8 * i)) & 0xFF;
            if (!freq_[i + 1].encode(e, b)) return false;
            freq_[i + 1].bump(b, kSimpleUp[i + 1], 1);
        }
    }
    return true;
}
bool SimpleModel::decode(uint8_t, Decoder &d, uint32_t &dist) {  // simple.rs:66-80
    size_t head;
    if (!freq_[0].decode(d, head)) return false;
    freq_[0].bump(head, kSimpleUp[0], 1);
    uint32_t v = static_cast<uint32_t>(head);
    if (head == 0xFF) {
        for (int i = 0; i < 3; ++i) {
            size_t b;
            if (!freq_[i + 1].decode(d, b)) return false;
            freq_[i + 1].bump(b, kSimpleUp[i + 1], 1);
            v += static_cast<uint32_t>(b) << (8 * i);
        }
    }
    dist = v;
    return true;
}

uint64_t model_max_block(int model_id) {
    switch (model_id) {
    case DK_MODEL_DARK: return 0x7FFFFFFEull;         // dist + 1 must have < 32 significant bits
    case DK_MODEL_EXP: return 1ull << 24;             // 24 coded bits (exp.rs:67)
    case DK_MODEL_YBS: return 1ull << 29;             // group - 12 < 19 (ybs.rs:59,110-113)
    case DK_MODEL_SIMPLE: return (1ull << 24) + 254;  // three extra bytes (simple.rs:56-61)
    case DK_MODEL_RAWDC: return 0xFFFFFFFEull;
    default: return 0;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Block stream layout (src/block/dc.rs)
// ------------------------------------------------------------------------------------------------------------------
namespace {

// A stream that is still arriving from the GPU: wait until more than k entries are there (*have = the frontier).  The frontier is moved
// by host functions on the context's HIP stream; DC_STREAM_POISON means the producer gave up (a copy failed, the stream reported an
// error), and a frontier that stands still for stall_ms (DcStream::stall_ms, default 20 s: a D2H piece of a 2 GiB block takes 40 ms)
// means the stream is hung or has faulted: DK_E_HIP either way -- no coder thread spins for ever.
__attribute__((noinline)) int wait_for_entries(const DcStream &s, size_t k, size_t *have) {
    size_t h = s.ready->load(std::memory_order_acquire);
    std::chrono::steady_clock::time_point since;
    bool timing = false;
    const double limit_ms = static_cast<double>(s.stall_ms ? s.stall_ms : 20000u);
    for (unsigned spins = 0; h <= k; h = s.ready->load(std::memory_order_acquire)) {
        if (++spins < (1u << 14)) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
            continue;
        }
        std::this_thread::yield();
        if (!timing) { since = std::chrono::steady_clock::now(); timing = true; }
        else if ((spins & 0xFFu) == 0 && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - since).count() > limit_ms) return DK_E_HIP;
    }
    if (h == DC_STREAM_POISON) return DK_E_HIP;
    *have = h < s.m ? h : s.m;
    return DK_OK;
}

template <class M, class E, class = void> struct has_bulk : std::false_type {};
template <class M, class E>
struct has_bulk<M, E, decltype(void(std::declval<M &>().bulk(static_cast<const uint32_t *>(nullptr), static_cast<const uint8_t *>(nullptr), size_t(0), std::declval<E &>())))>
    : std::true_type {};

// src/block/dc.rs:54-90 with any model M
template <class M, class E>
int write_stream(M &model, const DcStream &s, E &e) {
    const size_t n = s.n;
    auto code = [&](uint32_t v, uint8_t sym) { return model.encode(v, sym, e); };
    auto fail = [&] { return e.error() ? e.error() : DK_E_MODEL; };  // the sink's own error (capacity) wins over "the model refused"
    // Init-table RLE: alternating present / absent run lengths over symbols 0..254; every present symbol is followed
    // by its first position.  The first present-run length is raw, all later lengths are len-1.  Symbol 0xFF is
    // never visited (all loops stop at 0xFF): blocks containing byte 0xFF encode, but cannot be decoded.
    bool active = true;
    size_t i = 0;
    while (i < 0xFF) {
        const size_t base = i;
        if (active) {
            while (i < 0xFF && s.init[i] < n) ++i;
            if (!code(static_cast<uint32_t>(base == 0 ? i : i - base - 1), 0)) return fail();
            for (size_t c = base; c < i; ++c)
                if (!code(s.init[c], static_cast<uint8_t>(c))) return fail();
            active = false;
        } else {
            do { ++i; } while (i < 0xFF && s.init[i] == n);
            if (!code(static_cast<uint32_t>(i - base - 1), 0)) return fail();
            active = true;
        }
    }
    // src/block/dc.rs:82-85.  A stream that is still arriving from the GPU (DcStream::ready) is walked stretch by stretch: the wait sits
    // between the stretches, the loop over a stretch is the loop over a finished stream (a frontier check per distance inside it cost the
    // four-stage pipeline 3 % on the GPU box: one more live value in loops that are short of registers).
    for (size_t k = 0; k < s.m;) {
        size_t have = s.m;
        if (s.ready) {
            const int rc = wait_for_entries(s, k, &have);
            if (rc != DK_OK) return rc;
        }
        if constexpr (has_bulk<M, E>::value) {  // a stage that walks a whole stretch with its cursors in registers (the four-stage pipeline's merger)
            if (!model.bulk(s.dist + k, s.sym + k, have - k, e)) return fail();
            k = have;
        } else {
            for (; k < have; ++k)
                if (!code(s.dist[k], s.sym[k])) return fail();
        }
    }
    if (!code(s.origin, 0)) return fail();  // src/block/dc.rs:88 under CTX_0
    if (!e.finish()) return e.error();
    return DK_OK;
}

// src/model/raw.rs:12-44: every model.encode call becomes one 10-byte little-endian record
struct RecordSink {
    uint8_t *out; size_t cap, len = 0; bool full = false;
    void rec(uint32_t d, uint8_t sym, uint8_t rank, uint32_t limit) {
        if (len + 10 > cap) { full = true; return; }
        uint8_t *p = out + len;
        for (int i = 0; i < 4; ++i) p[i] = static_cast<uint8_t>(d >> (8 * i));
        p[4] = sym; p[5] = rank;
        for (int i = 0; i < 4; ++i) p[6 + i] = static_cast<uint8_t>(limit >> (8 * i));
        len += 10;
    }
};
int write_records(const DcStream &s, uint8_t *out, size_t cap, size_t *out_len) {
    if (!s.rank || !s.run_end) return DK_E_ARG;
    RecordSink r{out, cap};
    const size_t n = s.n;
    bool active = true;
    size_t i = 0;
    while (i < 0xFF) {  // same control flow as write_stream; CTX_0 = {0, 0, 0x101} (src/block/dc.rs:16-18)
        const size_t base = i;
        if (active) {
            while (i < 0xFF && s.init[i] < n) ++i;
            r.rec(static_cast<uint32_t>(base == 0 ? i : i - base - 1), 0, 0, 0x101);
            for (size_t c = base; c < i; ++c) r.rec(s.init[c], static_cast<uint8_t>(c), 0, static_cast<uint32_t>(n));
            active = false;
        } else {
            do { ++i; } while (i < 0xFF && s.init[i] == n);
            r.rec(static_cast<uint32_t>(i - base - 1), 0, 0, 0x101);
            active = true;
        }
    }
    for (size_t k = 0; k < s.m; ++k) r.rec(s.dist[k], s.sym[k], s.rank[k], static_cast<uint32_t>(n - s.run_end[k]));
    r.rec(s.origin, 0, 0, 0x101);
    *out_len = r.len;
    return r.full ? DK_E_CAPACITY : DK_OK;
}

// compress::bwt::dc::decode: rebuilds the BWT from first positions + distances while keeping the symbols ordered by
// their next known position (the decoder's view of the MTF list).  `next_dist(symbol, &d)` supplies distances.
inline void fill_run(uint8_t *out, size_t n, size_t i, size_t stop, uint8_t sym) {
    if (stop - i <= 16 && i + 16 <= n) {  // BWT runs are short: one unconditional 16-byte splat beats a memset call
        uint8_t splat[16];
        std::memset(splat, sym, 16);
        std::memcpy(out + i, splat, 16);
    } else {
        std::memset(out + i, sym, stop - i);
    }
}

template <class F>
int dc_rebuild(uint32_t const init[256], uint8_t *out, size_t n, F &&next_dist, int *single) {
    uint64_t next[256];
    uint8_t order[256 + 16];
    size_t alpha = 0;
    for (int s = 0; s < 256; ++s) {
        next[s] = init[s];
        if (init[s] < n) {  // insertion by first position
            size_t j = alpha++;
            while (j > 0 && next[order[j - 1]] > next[s]) { order[j] = order[j - 1]; --j; }
            order[j] = static_cast<uint8_t>(s);
        }
    }
    if (single) *single = alpha <= 1;
    if (alpha == 0) return DK_E_STREAM;  // no symbol at all: a block of nothing but 0xFF bytes, which the header cannot carry
    if (alpha == 1) {  // "redundant alphabet": filled without reading any distance
        std::memset(out, order[0], n);
        return DK_OK;
    }
    // the positions are kept in list order beside the symbols (no next[order[r]] double load inside the sinking loop)
    uint64_t pos[256 + 1];
    for (size_t j = 0; j < alpha; ++j) pos[j] = next[order[j]];
    pos[alpha] = ~0ull;  // sentinel: ends the sinking loop
    size_t i = 0;
    while (i < n) {
        const uint8_t sym = order[0];
        const uint64_t stop = pos[1];
        if (stop > n || stop < i) return DK_E_STREAM;
        fill_run(out, n, i, stop, sym);
        i = stop;
        uint32_t d;
        if (int rc = next_dist(sym, &d)) return rc;
        const uint64_t future = stop + d;
        if (future > n) return DK_E_STREAM;
        // (the sinking loop's exit is mispredicted for most distances -- the symbol sinks a few places, a different number every time.  Counting the
        // places with two vector compares over pos[j] - j and moving them with blends was built in round 4 and is 13 % slower: the next step's
        // loads then wait for vector stores they only partly overlap, which costs more than the misprediction.  Also without effect on the GPU box:
        // the decoder's division replaced by the encoder's reciprocal multiply, and a prefetch of the next symbol's model state -- the symbol of
        // the next distance is known before the current one is decoded -- 30.0-30.3 against 29.8-30.7 ns per distance.)
        size_t r = 1;
        while (future + r > pos[r]) { order[r - 1] = order[r]; pos[r - 1] = pos[r]; ++r; }
        order[r - 1] = sym;
        pos[r - 1] = future + r - 1;
    }
    for (size_t j = 0; j < alpha; ++j)
        if (pos[j] < n || pos[j] >= n + alpha) return DK_E_STREAM;
    return DK_OK;
}

// src/block/dc.rs:121-151
template <class M>
int read_stream(M &model, Decoder &d, size_t n, uint8_t *bwt_out, uint32_t *origin, int *single) {
    uint32_t init[256];
    for (auto &v : init) v = static_cast<uint32_t>(n);
    bool active = true;
    size_t i = 0;
    while (i < 0xFF) {
        uint32_t v;
        if (!model.decode(0, d, v)) return DK_E_STREAM;
        const size_t num = static_cast<size_t>(v) + ((i == 0 && active) ? 0 : 1);
        if (active) {
            for (size_t c = i; c < i + num && c < 0x100; ++c) {
                if (!model.decode(static_cast<uint8_t>(c), d, v)) return DK_E_STREAM;
                init[c] = v;
            }
        }
        active = !active;
        i += num;
    }
    int rc = dc_rebuild(init, bwt_out, n, [&](uint8_t sym, uint32_t *out) {
        return model.decode(sym, d, *out) ? DK_OK : DK_E_STREAM;
    }, single);
    if (rc) return rc;
    if (single && *single) {
        // One-symbol block: bwt::dc::decode returns without asking for a distance, but the encoder wrote one (the end-of-block sweep
        // entry of that symbol, src/block/dc.rs:82-85).  The reference goes on to read `origin` from it (DESIGN.md "Reference
        // quirks"); here it is decoded and dropped, so that origin is the real one and consumed() is the length the encoder wrote --
        // a following [u32 n][stream] record starts exactly there.
        uint32_t sweep;
        if (!model.decode(bwt_out[0], d, sweep)) return DK_E_STREAM;
    }
    if (!model.decode(0, d, *origin)) return DK_E_STREAM;
    if (!d.finish()) return DK_E_STREAM;
    return DK_OK;
}

template <class F>
int with_model(int model_id, F &&f) {
    switch (model_id) {
    case DK_MODEL_DARK: { auto m = std::make_unique<DarkModel>(); return f(*m); }
    case DK_MODEL_EXP: { auto m = std::make_unique<ExpModel>(); return f(*m); }
    case DK_MODEL_YBS: { auto m = std::make_unique<YbsModel>(); return f(*m); }
    case DK_MODEL_SIMPLE: { auto m = std::make_unique<SimpleModel>(); return f(*m); }
    default: return DK_E_MODEL;
    }
}

}  // namespace


// ------------------------------------------------------------------------------------------------------------------
// EventPipe: ring of batches between the modelling thread (producer) and the coding thread (consumer)
// ------------------------------------------------------------------------------------------------------------------
class EventPipe {
public:
    static constexpr uint32_t kBatch = 1u << 14;  // 16-bit units per batch (32 KiB: stays in the consumer's L2)
    static constexpr int kSlots = 16;
    static constexpr uint32_t kEnd = 0x80000000u;  // OR-ed into the count of the last batch
    EventPipe() : buf_(new uint16_t[static_cast<size_t>(kSlots) * kBatch]) {
        for (auto &c : ready_) c.store(0, std::memory_order_relaxed);
    }
    uint16_t *slot(int i) { return buf_.get() + static_cast<size_t>(i) * kBatch; }
    void acquire_free(int i) {  // producer: wait until slot i has been consumed
        unsigned spins = 0;
        while (ready_[i].load(std::memory_order_acquire) != 0) backoff(spins);
        producer_spins += spins;
    }
    void publish(int i, uint32_t count_and_flags) { ready_[i].store(count_and_flags | kFull, std::memory_order_release); }
    uint32_t acquire_full(int i) {  // consumer: wait until slot i is full; returns count | kEnd
        unsigned spins = 0;
        uint32_t v;
        while ((v = ready_[i].load(std::memory_order_acquire)) == 0) backoff(spins);
        consumer_spins += spins;
        return v & ~kFull;
    }
    void release(int i) { ready_[i].store(0, std::memory_order_release); }
    uint64_t producer_spins = 0, consumer_spins = 0;  // pause iterations each side spent waiting (DK_TRACE)

private:
    static constexpr uint32_t kFull = 0x40000000u;
    static void backoff(unsigned &spins) {
        // busy-wait: a hand-off is expected every few microseconds; yielding to the scheduler costs far more than that
        if (++spins < (1u << 14)) {  // ~0.5 ms of pauses, then let the scheduler run whoever we are waiting for
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        } else {
            std::this_thread::yield();
        }
    }
    std::unique_ptr<uint16_t[]> buf_;
    alignas(64) std::atomic<uint32_t> ready_[kSlots];
};

EventSink::EventSink(EventPipe &p) : pipe_(p) {
    pipe_.acquire_free(0);
    cur_ = pipe_.slot(0);
    cap_ = EventPipe::kBatch;
}
void EventSink::flush() {
    pipe_.publish(slot_, fill_);
    slot_ = (slot_ + 1) % EventPipe::kSlots;
    pipe_.acquire_free(slot_);
    cur_ = pipe_.slot(slot_);
    fill_ = 0;
}
bool EventSink::finish() {
    if (!cur_) return err_ == 0;  // already finished
    pipe_.publish(slot_, fill_ | EventPipe::kEnd);
    fill_ = 0;
    cur_ = nullptr;
    return err_ == 0;
}

namespace {

// consumer side: the range coder over the recorded intervals
__attribute__((noinline)) int drain_pipe(EventPipe &pipe, uint8_t *out, size_t cap, size_t *out_len) {
    RangeState rs;
    size_t len = 0;
    int err = DK_OK;
    for (int slot = 0;; slot = (slot + 1) % EventPipe::kSlots) {
        const uint32_t v = pipe.acquire_full(slot);
        const uint32_t count = v & ~EventPipe::kEnd;
        const uint16_t *ev = pipe.slot(slot);
        if (!err && len + 4 * static_cast<size_t>(count) + 12 > cap) err = DK_E_CAPACITY;  // keep draining so the producer can end
        if (!err) {
            uint8_t *p = out + len;
            for (uint32_t k = 0; k < count;) {
                const uint32_t u = ev[k];
                const uint32_t span = rs.hi - rs.low;
                int nb;
                if (u & 0x8000u) {  // binary decision under total 4096
                    const uint32_t zero = u & 0x0FFFu, r = span >> 12;
                    const bool one = (u & 0x4000u) != 0;
                    nb = rs.narrow(r, one ? zero : 0u, one ? 4096u : zero, p);
                    k += 1;
                } else {
                    const uint32_t to = ev[k + 1], total = ev[k + 2];
                    const uint32_t r = span / total;
                    nb = r ? rs.narrow(r, u, to, p) : -1;
                    k += 3;
                }
                if (nb < 0) { err = DK_E_INTERNAL; break; }
                p += nb;
            }
            len = static_cast<size_t>(p - out);
        }
        pipe.release(slot);
        if (v & EventPipe::kEnd) break;
    }
    if (!err) {
        if (len + 4 > cap) err = DK_E_CAPACITY;
        else for (int i = 0; i < 4; ++i) out[len++] = static_cast<uint8_t>(rs.low >> (24 - 8 * i));  // ari::Encoder::finish
    }
    *out_len = len;
    return err;
}

// consumer side of the pipe when the consumer knows which kind of decision comes next
class PipeReader {
public:
    explicit PipeReader(EventPipe &p) : pipe_(p) { next_batch(); }
    // table decision (3 units, never split across batches) / binary decision (1 unit)
    inline bool table(uint32_t &from, uint32_t &to, uint32_t &total) {
        if (pos_ == count_ && !refill()) return false;
        from = cur_[pos_]; to = cur_[pos_ + 1]; total = cur_[pos_ + 2];
        pos_ += 3;
        return true;
    }
    inline bool bit(uint32_t &zero, bool &one) {
        if (pos_ == count_ && !refill()) return false;
        const uint32_t u = cur_[pos_++];
        zero = u & 0x0FFFu;
        one = (u & 0x4000u) != 0;
        return true;
    }
    void drain() {  // consume whatever the producer still sends, so that it can end
        while (!last_) { pipe_.release(slot_); slot_ = (slot_ + 1) % EventPipe::kSlots; next_batch(); }
        pipe_.release(slot_);
    }

private:
    bool refill() {
        while (pos_ == count_) {
            if (last_) return false;  // the producer ended early (it failed)
            pipe_.release(slot_);
            slot_ = (slot_ + 1) % EventPipe::kSlots;
            next_batch();
        }
        return true;
    }
    void next_batch() {
        const uint32_t v = pipe_.acquire_full(slot_);
        last_ = (v & EventPipe::kEnd) != 0;
        count_ = v & ~EventPipe::kEnd;
        cur_ = pipe_.slot(slot_);
        pos_ = 0;
    }
    EventPipe &pipe_;
    const uint16_t *cur_ = nullptr;
    uint32_t pos_ = 0, count_ = 0;
    int slot_ = 0;
    bool last_ = false;
};

// The dark model as the two threads see it: every decision that needs model state comes from the second thread through the pipe;
// the stateless tail of the mantissa (probability 1/2 under total 4096, about a third of all decisions of a text block) is produced
// by the coding thread itself from the distance, which takes that traffic off the pipe and evens out the two sides.
struct DarkModelSide {  // second thread
    DarkModel &m;
    bool encode(uint32_t dist, uint8_t symbol, EventSink &e) {
        return m.encode_exponent(dist, symbol, e) && m.encode_mantissa_modelled(dist, e);
    }
};
struct DarkCoderSide {  // calling thread
    PipeReader &rd;
    bool encode(uint32_t dist, uint8_t, Encoder &e) {
        if (dist >= 0x7FFFFFFFu) return false;
        uint32_t from, to, total;
        if (!rd.table(from, to, total) || !e.put(total, from, to)) return false;
        for (unsigned k = DarkModel::exponent_bits(dist) + DarkModel::modelled_mantissa_bits(dist); k; --k) {
            uint32_t zero; bool one;
            if (!rd.bit(zero, one) || !e.put_bit12(zero, one)) return false;
        }
        return DarkModel::encode_mantissa_flat(dist, e);
    }
};

#if defined(__linux__)
// "0-7,128-135" -> {0..7, 128..135}
std::vector<int> read_cpu_list(const std::string &path) {
    std::vector<int> out;
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return out;
    char buf[4096];
    if (std::fgets(buf, sizeof buf, f)) {
        const char *p = buf;
        while (*p && *p != '\n') {
            char *end;
            long a = std::strtol(p, &end, 10), b = a;
            if (end == p) break;
            if (*end == '-') { p = end + 1; b = std::strtol(p, &end, 10); }
            for (long c = a; c <= b && out.size() < 4096; ++c) out.push_back(static_cast<int>(c));
            p = (*end == ',') ? end + 1 : end;
        }
    }
    std::fclose(f);
    return out;
}
// The machine's last-level-cache groups (one per CCX on EPYC), read once from sysfs.
struct L3Group { int id; cpu_set_t cpus; cpu_set_t primary; int numa; };  // primary: one hardware thread (the lowest numbered) of every core; numa: its memory node (-1: unknown)
struct Topology { std::vector<L3Group> groups; size_t threads_per_core = 1; };
const Topology &topology() {
    static const Topology topo = [] {
        Topology t;
        cpu_set_t seen;
        CPU_ZERO(&seen);
        int misses = 0;
        for (int cpu = 0; cpu < CPU_SETSIZE && misses < 64; ++cpu) {  // (cpu numbers can have holes; 64 in a row is the end)
            if (CPU_ISSET(cpu, &seen)) continue;
            const std::string base = "/sys/devices/system/cpu/cpu" + std::to_string(cpu);
            const std::vector<int> l3 = read_cpu_list(base + "/cache/index3/shared_cpu_list");
            if (l3.empty()) { ++misses; continue; }  // no such cpu / no L3 information
            misses = 0;
            if (t.groups.empty()) t.threads_per_core = std::max<size_t>(1, read_cpu_list(base + "/topology/thread_siblings_list").size());
            L3Group g;
            g.id = l3.front();
            g.numa = -1;
            CPU_ZERO(&g.cpus);
            CPU_ZERO(&g.primary);
            for (int c : l3) {
                if (c < 0 || c >= CPU_SETSIZE) continue;
                CPU_SET(c, &g.cpus);
                CPU_SET(c, &seen);
                const std::vector<int> sib = read_cpu_list("/sys/devices/system/cpu/cpu" + std::to_string(c) + "/topology/thread_siblings_list");
                if (sib.empty() || *std::min_element(sib.begin(), sib.end()) == c) CPU_SET(c, &g.primary);
            }
            t.groups.push_back(g);
        }
        for (int node = 0, missing = 0; node < 1024 && missing < 8; ++node) {  // the memory node of every group (node numbers can have holes)
            const std::vector<int> cpus = read_cpu_list("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
            if (cpus.empty()) { ++missing; continue; }
            missing = 0;
            for (L3Group &g : t.groups)
                if (std::find(cpus.begin(), cpus.end(), g.id) != cpus.end()) g.numa = node;
        }
        return t;
    }();
    return topo;
}
#endif

// Host threads for one block.  Unconfined, helper and caller tend to land on different CCDs and every hand-off costs more than the
// split saves (EPYC 9575F, ns per distance of a text block: one thread 22; two threads unconfined 38, confined to one L3 16-19).
// The encode_* functions below return DK_E_NODEVICE (reused as "not available") when the cores cannot be had; the caller then falls
// back to a narrower form.
// The helper threads of one encode call and its caller are confined, for the duration of the call, to ONE last-level-cache group with
// enough cores.  Groups are claimed across processes with an advisory lock (a file per group under /dev/shm, released by close or
// process death): ranks started by one launcher tend to sit in the same CCX when their first block is ready, and two pipelines
// spinning in one CCX would halve each other.  The caller's own group is tried first, then the others by distance; when every
// group is taken the call gets none (and codes on one thread).  The caller's affinity mask is restored on release.
#if defined(__linux__)
// Lock file of one L3 group.  It lives in a directory only this user can write ($XDG_RUNTIME_DIR, else /dev/shm/dark_amd.<uid>, mode 0700,
// checked to be a real directory of ours), is opened without following links and must be a regular single-link file of ours: a name
// planted by somebody else in a world-writable directory is never opened, chmod'ed or locked.  Coordination is therefore among the
// processes of one user -- the ranks one launcher started, which is the case that matters.
int open_group_lock(int group_id) {
    std::string dir;
    const char *xdg = getenv("XDG_RUNTIME_DIR");
    struct stat st;
    if (xdg && xdg[0] == '/' && lstat(xdg, &st) == 0 && S_ISDIR(st.st_mode) && st.st_uid == geteuid() && (st.st_mode & 022) == 0) {
        dir = xdg;
    } else {
        dir = "/dev/shm/dark_amd." + std::to_string(static_cast<unsigned long>(geteuid()));
        if (mkdir(dir.c_str(), 0700) != 0 && errno != EEXIST) return -1;
        if (lstat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & 077) != 0) return -1;
    }
    const std::string path = dir + "/dark_amd.l3." + std::to_string(group_id) + ".lock";
    const int fd = open(path.c_str(), O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0600);
    if (fd < 0) return -1;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_nlink != 1 || st.st_uid != geteuid()) { close(fd); return -1; }
    return fd;
}
#endif

void note_group(int id, int numa);  // remembers the L3 group the calling thread's coding pass claimed (dk_last_entropy_info)
int preferred_numa();               // the memory node the calling thread's block comes from (its GPU's): groups there are tried first
struct ThreadPair {
    int me = -1;
#if defined(__linux__)
    cpu_set_t saved, group;
    int lock_fd = -1;
    bool acquire(int cores_needed = 2) {
        if (pthread_getaffinity_np(pthread_self(), sizeof(saved), &saved) != 0) return false;
        const int cpu = sched_getcpu();
        const Topology &topo = topology();
        if (cpu < 0 || cpu >= CPU_SETSIZE || topo.groups.empty()) return false;
        size_t own = topo.groups.size();
        for (size_t g = 0; g < topo.groups.size(); ++g)
            if (CPU_ISSET(cpu, &topo.groups[g].cpus)) own = g;
        if (own == topo.groups.size()) return false;
        std::vector<int> numa_of(topo.groups.size());
        for (size_t g = 0; g < topo.groups.size(); ++g) numa_of[g] = topo.groups[g].numa;
        for (const size_t gidx : l3_claim_order(numa_of, own, preferred_numa())) {  // the GPU's memory node first, there and elsewhere by distance from the caller's group
            const long gi = static_cast<long>(gidx);
            const L3Group &g = topo.groups[gidx];
            // one hardware thread per core where the caller's mask allows it: two stages of one pipeline on the two hardware
            // threads of one core would share its execution units (seen as run-to-run swings of several percent)
            cpu_set_t usable;
            CPU_AND(&usable, &g.primary, &saved);
            if (CPU_COUNT(&usable) < cores_needed) {
                CPU_AND(&usable, &g.cpus, &saved);
                const size_t cores = (static_cast<size_t>(CPU_COUNT(&usable)) + topo.threads_per_core - 1) / topo.threads_per_core;
                if (cores < static_cast<size_t>(cores_needed)) continue;
            }
            const int fd = open_group_lock(g.id);
            if (fd >= 0) {
                if (flock(fd, LOCK_EX | LOCK_NB) != 0) { close(fd); continue; }  // another pipeline lives here
            } else if (gi != static_cast<long>(own)) {
                continue;  // cannot coordinate: stay at home rather than crowd somebody else's group
            }
            if (pthread_setaffinity_np(pthread_self(), sizeof(usable), &usable) != 0) { if (fd >= 0) close(fd); continue; }
            group = usable;
            lock_fd = fd;
            me = g.id;
            note_group(me, g.numa);
            return true;
        }
        return false;
    }
    void pin_partner() const { (void)pthread_setaffinity_np(pthread_self(), sizeof(group), &group); }
    void release() {
        (void)pthread_setaffinity_np(pthread_self(), sizeof(saved), &saved);
        if (lock_fd >= 0) close(lock_fd);
        lock_fd = -1;
    }
#else
    bool acquire(int = 2) { return false; }
    void pin_partner() const {}
    void release() {}
#endif
};
void trace_pair(const ThreadPair &tp, const EventPipe &pipe) {
    if (getenv("DK_TRACE"))
        fprintf(stderr, "[dark_amd] entropy: two threads inside the L3 group of cpu %d; waits (pause iterations): models %llu, coder %llu\n",
                tp.me, (unsigned long long)pipe.producer_spins, (unsigned long long)pipe.consumer_spins);
}

template <class M>
int encode_two_threads(M &model, const DcStream &s, uint8_t *out, size_t cap, size_t *out_len) {
    ThreadPair tp;
    if (!tp.acquire()) return DK_E_NODEVICE;
    EventPipe pipe;
    int producer_rc = DK_OK;
    std::thread producer;
    try {
        producer = std::thread([&] {
            tp.pin_partner();
            EventSink sink(pipe);
            producer_rc = write_stream(model, s, sink);
            if (producer_rc != DK_OK) sink.finish();  // write_stream finishes the sink itself on success
        });
    } catch (...) { tp.release(); return DK_E_NODEVICE; }
    int rc = drain_pipe(pipe, out, cap, out_len);
    producer.join();
    tp.release();
    trace_pair(tp, pipe);
    return producer_rc ? producer_rc : rc;
}

// dark model: all stateful modelling on the second thread; stateless mantissa tail + range coder on the calling one
int encode_two_threads(DarkModel &model, const DcStream &s, uint8_t *out, size_t cap, size_t *out_len) {
    ThreadPair tp;
    if (!tp.acquire()) return DK_E_NODEVICE;
    EventPipe pipe;
    int producer_rc = DK_OK;
    std::thread producer;
    try {
        producer = std::thread([&] {
            tp.pin_partner();
            EventSink sink(pipe);
            DarkModelSide side{model};
            producer_rc = write_stream(side, s, sink);
            if (producer_rc != DK_OK) sink.finish();
        });
    } catch (...) { tp.release(); return DK_E_NODEVICE; }
    int rc;
    {
        PipeReader rd(pipe);
        DarkCoderSide side{rd};
        Encoder e(out, cap);
        rc = write_stream(side, s, e);
        *out_len = e.size();
        rd.drain();
    }
    producer.join();
    tp.release();
    trace_pair(tp, pipe);
    return producer_rc ? producer_rc : rc;
}

// ------------------------------------------------------------------------------------------------------------------
// Four-stage pipeline for the dark model (large single blocks, four cores of one L3):
//     exponent model  ->  ring E  \
//                                   merger  ->  ring U  ->  range coder (calling thread)
//     mantissa model  ->  ring M  /
// Every decision travels as a UNIFORM 16-byte event (from, to, ceil(2^64 / total)): the coder computes r = span / total as the high
// half of one 64 x 64 multiply -- exact for span < 2^32, total < 2^15 (the error term span * (inv * total - 2^64) stays below 2^64 /
// total), and 2^52 for the binary decisions makes it the shift by 12 -- so its loop has no division and no branch that depends on
// the kind of decision: 12.1 ns per distance against 13.9 for the two-kind loop and ~16 when it also has to find its way through
// two input rings.  The expensive model half (one table decision per distance, 11 ns) and the mantissa half (12.5 ns) run side by
// side; the merger only copies events in the order the reference emits them (counts follow from the distance's bit length).
// ------------------------------------------------------------------------------------------------------------------
struct UEvent { uint64_t inv; uint32_t from, to; };
static_assert(sizeof(UEvent) == 16, "uniform event is 16 bytes");

inline uint64_t reciprocal64(uint32_t total) {  // ceil(2^64 / total), total >= 2
    return (total & (total - 1)) ? (~0ull / total + 1) : ((1ull << 63) / total * 2);
}
const uint64_t *reciprocal_table() {  // totals of the dark model's mixed tables stay below 3 * 2^12
    static const std::vector<uint64_t> table = [] {
        std::vector<uint64_t> t(1u << 14);
        t[0] = t[1] = 0;
        for (uint32_t d = 2; d < (1u << 14); ++d) t[d] = reciprocal64(d);
        return t;
    }();
    return table.data();
}

class URing {
public:
    static constexpr uint32_t kBatch = 2048;  // events per batch (32 KiB)
    static constexpr int kSlots = 16;
    static constexpr uint32_t kEnd = 0x80000000u;
    URing() : buf_(new UEvent[static_cast<size_t>(kSlots) * kBatch + 4]) {  // + 4: the merger copies four events at a time
        for (auto &c : ready_) c.store(0, std::memory_order_relaxed);
    }
    UEvent *slot(int i) { return buf_.get() + static_cast<size_t>(i) * kBatch; }
    void acquire_free(int i) { unsigned spins = 0; while (ready_[i].load(std::memory_order_acquire) != 0) backoff(spins); producer_spins += spins; }
    void publish(int i, uint32_t count_and_flags) { ready_[i].store(count_and_flags | kFull, std::memory_order_release); }
    uint32_t acquire_full(int i) {
        unsigned spins = 0;
        uint32_t v;
        while ((v = ready_[i].load(std::memory_order_acquire)) == 0) backoff(spins);
        consumer_spins += spins;
        return v & ~kFull;
    }
    void release(int i) { ready_[i].store(0, std::memory_order_release); }
    uint64_t producer_spins = 0, consumer_spins = 0;

private:
    static constexpr uint32_t kFull = 0x40000000u;
    static void backoff(unsigned &spins) {
        if (++spins < (1u << 14)) {  // ~0.5 ms of pauses, then let the scheduler run whoever we are waiting for
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        } else {
            std::this_thread::yield();
        }
    }
    std::unique_ptr<UEvent[]> buf_;
    alignas(64) std::atomic<uint32_t> ready_[kSlots];
};

class USink {  // what the model halves (and the merger) write into
public:
    explicit USink(URing &r) : ring_(r), inv_(reciprocal_table()) { ring_.acquire_free(0); cur_ = ring_.slot(0); }
    inline bool raw(const UEvent &ev) {
        if (fill_ == URing::kBatch) flush();
        cur_[fill_++] = ev;
        return true;
    }
    // bulk interface (merger): at least `want` (<= 4) free places at tail(), then advance(count <= want)
    inline UEvent *tail(uint32_t want) {
        if (fill_ + want > URing::kBatch) flush();
        return cur_ + fill_;
    }
    inline void advance(uint32_t count) { fill_ += count; }
    // cursor interface (merger's bulk loop): the free places of the current batch as a pointer range; done(p) = everything before p is written
    inline UEvent *cursor(UEvent **end) { *end = cur_ + URing::kBatch; return cur_ + fill_; }
    inline void done(UEvent *p) { fill_ = static_cast<uint32_t>(p - cur_); }
    inline bool put(uint32_t total, uint32_t from, uint32_t to) {
        if (!(from < to && to <= total && total < 32768u && total >= 2u)) return fail(DK_E_INTERNAL);
        return raw(UEvent{total < (1u << 14) ? inv_[total] : reciprocal64(total), from, to});
    }
    inline bool put_pow2(unsigned shift, uint32_t from, uint32_t to) { return raw(UEvent{1ull << (64 - shift), from, to}); }
    inline bool put_bit12(uint32_t zero, bool one) { return raw(UEvent{1ull << 52, one ? zero : 0u, one ? 4096u : zero}); }
    // half of a mixed decision (five-stage pipeline): from = what lies below the symbol, to = frequency << 16 | total; half of a mixed bit: from = zero
    inline bool half(uint32_t total, uint32_t below, uint32_t freq) { return raw(UEvent{0, below, (freq << 16) | total}); }
    inline bool half_bit(uint32_t zero) { return raw(UEvent{0, zero, 0}); }
    bool finish() {
        if (!cur_) return err_ == 0;
        ring_.publish(slot_, fill_ | URing::kEnd);
        cur_ = nullptr;
        return err_ == 0;
    }
    int error() const { return err_; }
    bool fail(int e) { if (!err_) err_ = e; return false; }

private:
    void flush() {
        ring_.publish(slot_, fill_);
        slot_ = (slot_ + 1) % URing::kSlots;
        ring_.acquire_free(slot_);
        cur_ = ring_.slot(slot_);
        fill_ = 0;
    }
    URing &ring_;
    const uint64_t *inv_;
    UEvent *cur_ = nullptr;
    uint32_t fill_ = 0;
    int slot_ = 0;
    int err_ = 0;
};

class UReader {
public:
    explicit UReader(URing &r) : ring_(r) { next_batch(); }
    inline const UEvent *next() {  // nullptr: the producer ended early
        if (pos_ == count_ && !refill()) return nullptr;
        return cur_ + pos_++;
    }
    // bulk interface (merger): events left in the current batch (0: ask refill_batch()), a pointer to them, skip(count)
    inline uint32_t available() const { return count_ - pos_; }
    inline const UEvent *head() const { return cur_ + pos_; }
    inline void skip(uint32_t count) { pos_ += count; }
    inline bool refill_batch() { return refill(); }
    inline const UEvent *cursor(const UEvent **end) const { *end = cur_ + count_; return cur_ + pos_; }
    inline void done(const UEvent *p) { pos_ = static_cast<uint32_t>(p - cur_); }
    void drain() {
        while (!last_) { ring_.release(slot_); slot_ = (slot_ + 1) % URing::kSlots; next_batch(); }
        ring_.release(slot_);
    }

private:
    bool refill() {
        while (pos_ == count_) {
            if (last_) return false;
            ring_.release(slot_);
            slot_ = (slot_ + 1) % URing::kSlots;
            next_batch();
        }
        return true;
    }
    void next_batch() {
        const uint32_t v = ring_.acquire_full(slot_);
        last_ = (v & URing::kEnd) != 0;
        count_ = v & ~URing::kEnd;
        cur_ = ring_.slot(slot_);
        pos_ = 0;
    }
    URing &ring_;
    const UEvent *cur_ = nullptr;
    uint32_t pos_ = 0, count_ = 0;
    int slot_ = 0;
    bool last_ = false;
};

struct DarkExponentSide {
    DarkModel &m;
    bool encode(uint32_t dist, uint8_t symbol, USink &e) { return m.encode_exponent(dist, symbol, e); }
};
// (only the three MODELLED mantissa bits: the flat tail below them goes through a model that never learns, dark.rs:225-227 -- no state, so
// the merger makes those events itself.  With the tail on this thread it was the pipeline's slowest stage: it never waited, the other
// three did; profiles/r04_entropy_ab.json)
// a sink that is nothing but a pointer into the current batch: a local of the caller, so it lives in a register (USink's fill count is a
// member the compiler reloads after every 16-byte store it cannot tell apart from it: one store-to-load round trip per decision)
struct UCursor {
    UEvent *o;
    inline bool put_bit12(uint32_t zero, bool one) { *o++ = UEvent{1ull << 52, one ? zero : 0u, one ? 4096u : zero}; return true; }
    inline bool half(uint32_t total, uint32_t below, uint32_t freq) { *o++ = UEvent{0, below, (freq << 16) | total}; return true; }
    inline bool half_bit(uint32_t zero) { *o++ = UEvent{0, zero, 0}; return true; }
};
struct DarkMantissaSide {
    DarkModel &m;
    bool encode(uint32_t dist, uint8_t, USink &e) { return m.encode_mantissa_modelled(dist, e); }
    bool bulk(const uint32_t *dist, const uint8_t *, size_t count, USink &e) {  // at most three decisions per distance
        UEvent *o_end;
        UCursor c{e.cursor(&o_end)};
        for (size_t k = 0; k < count; ++k) {
            if (__builtin_expect(c.o + 3 <= o_end, 1)) {
                if (!m.encode_mantissa_modelled(dist[k], c)) { e.done(c.o); return false; }
            } else {
                e.done(c.o);
                if (!m.encode_mantissa_modelled(dist[k], e)) return false;
                c.o = e.cursor(&o_end);
            }
        }
        e.done(c.o);
        return true;
    }
};
struct DarkMergeSide {  // the order of dark.rs:180-232: table decision, unary extension, mantissa bits
    UReader &exponent;
    UReader &mantissa;
    // `count` events from one ring to the other, four (64 bytes, two vector copies) at a time whenever both sides have four places:
    // what is copied beyond `count` is overwritten by the next copy, and no loop exit depends on the small counts of the usual case
    static inline bool move(UReader &from, USink &to, unsigned count) {
        while (count) {
            uint32_t have = from.available();
            if (have == 0) {
                if (!from.refill_batch()) return false;
                have = from.available();
            }
            if (have >= 4) {
                const unsigned c = count < 4 ? count : 4;
                std::memcpy(static_cast<void *>(to.tail(4)), from.head(), 4 * sizeof(UEvent));
                to.advance(c);
                from.skip(c);
                count -= c;
            } else {  // end of an input batch: one by one
                to.raw(*from.head());
                from.skip(1);
                count -= 1;
            }
        }
        return true;
    }
    // the mantissa bits below the three modelled ones, most significant first: a decision between two halves of 4096 each (the
    // never-updated model of dark.rs:225-227 stays at its initial zero = 2048), four events at a time like move()
    static inline void flat_tail(USink &to, uint32_t v, unsigned log) {
        for (unsigned left = log - 4; left;) {  // bits left - 1 .. 0 of v are still to go
            UEvent *out = to.tail(4);
            const unsigned c = left < 4 ? left : 4;
            for (unsigned j = 0; j < 4; ++j) {  // (what is written beyond c is overwritten by the next events)
                const uint32_t one = (v >> ((left - 1 - j) & 31u)) & 1u;
                out[j] = UEvent{1ull << 52, one << 11, 2048u + (one << 11)};
            }
            to.advance(c);
            left -= c;
        }
    }
    bool encode(uint32_t dist, uint8_t, USink &e) {
        if (dist >= 0x7FFFFFFFu) return false;
        const unsigned log = bit_length(dist + 1);
        if (!move(exponent, e, 1 + DarkModel::exponent_bits(dist)) || !move(mantissa, e, DarkModel::modelled_mantissa_bits(dist))) return false;
        if (log > 4) flat_tail(e, dist + 1, log);
        return true;
    }
    // A stretch of distances with the three rings' cursors in registers.  Through encode() every copy goes through the sink's and the readers'
    // members, which the compiler reloads after each 64-byte copy (the copy might have written them, for all it knows): three store-to-load
    // round trips in a row on the sink's fill count alone, per distance -- the merger had become the pipeline's slowest stage.  The common
    // distance (at most four exponent decisions, at most four flat bits, four events left in both input batches, twelve places in the output
    // batch) is copied here with plain pointers; anything else takes encode() between two synchronisations of the cursors.
    bool bulk(const uint32_t *dist, const uint8_t *, size_t count, USink &e) {
        const UEvent *pe_end, *pm_end, *pe = exponent.cursor(&pe_end), *pm = mantissa.cursor(&pm_end);
        UEvent *o_end, *o = e.cursor(&o_end);
        for (size_t k = 0; k < count; ++k) {
            const uint32_t d = dist[k];
            const unsigned log = bit_length(d + 1);  // (d = 2^32 - 1 gives 0 and goes the slow way, which refuses it)
            const unsigned ne = 1 + (log >= 8 ? log - 7 : 0), nm = log > 3 ? 3 : log - 1;
            if (__builtin_expect(log - 1u <= 7u && pe + 4 <= pe_end && pm + 4 <= pm_end && o + 12 <= o_end, 1)) {  // log 1 .. 8: ne <= 2, flat bits <= 4
                std::memcpy(static_cast<void *>(o), pe, 4 * sizeof(UEvent));
                o += ne;
                pe += ne;
                std::memcpy(static_cast<void *>(o), pm, 4 * sizeof(UEvent));
                o += nm;
                pm += nm;
                const uint32_t v = d + 1;
                const unsigned left = log > 4 ? log - 4 : 0;  // flat bits left - 1 .. 0 of v, most significant first
#pragma GCC unroll 4
                for (unsigned j = 0; j < 4; ++j) {
                    const uint32_t one = (v >> ((left - 1 - j) & 31u)) & 1u;
                    o[j] = UEvent{1ull << 52, one << 11, 2048u + (one << 11)};
                }
                o += left;
                continue;
            }
            exponent.done(pe);
            mantissa.done(pm);
            e.done(o);
            if (!encode(d, 0, e)) return false;
            pe = exponent.cursor(&pe_end);
            pm = mantissa.cursor(&pm_end);
            o = e.cursor(&o_end);
        }
        exponent.done(pe);
        mantissa.done(pm);
        e.done(o);
        return true;
    }
};

// Five stages: the exponent model itself in two halves (DarkModel::encode_exponent_symbol_half / _global_half), added up by the merger --
//     per-symbol tables -> ring A \
//     global tables     -> ring B  >  merger  ->  ring U  ->  range coder
//     mantissa model    -> ring M /
// With the coder's chain shortened and the merger's cursors in registers the exponent model was the slowest stage (it never waited, the
// other three did): its table decision is a + 2 b of a per-symbol and a global table, its extra bits the mean of a per-symbol and a global
// probability, and neither side reads the other's state.
struct DarkSymbolHalfSide {
    DarkModel &m;
    bool encode(uint32_t dist, uint8_t symbol, USink &e) { return m.encode_exponent_symbol_half(dist, symbol, e); }
    bool bulk(const uint32_t *dist, const uint8_t *sym, size_t count, USink &e) {  // at most 1 + 25 half decisions per distance
        UEvent *o_end;
        UCursor c{e.cursor(&o_end)};
        for (size_t k = 0; k < count; ++k) {
            if (__builtin_expect(c.o + 32 <= o_end, 1)) {
                if (!m.encode_exponent_symbol_half(dist[k], sym[k], c)) { e.done(c.o); return false; }
            } else {
                e.done(c.o);
                if (!m.encode_exponent_symbol_half(dist[k], sym[k], e)) return false;
                c.o = e.cursor(&o_end);
            }
        }
        e.done(c.o);
        return true;
    }
};
struct DarkGlobalHalfSide {
    DarkModel &m;
    bool encode(uint32_t dist, uint8_t symbol, USink &e) { return m.encode_exponent_global_half(dist, symbol, e); }
    bool bulk(const uint32_t *dist, const uint8_t *sym, size_t count, USink &e) {
        UEvent *o_end;
        UCursor c{e.cursor(&o_end)};
        for (size_t k = 0; k < count; ++k) {
            if (__builtin_expect(c.o + 32 <= o_end, 1)) {
                if (!m.encode_exponent_global_half(dist[k], sym[k], c)) { e.done(c.o); return false; }
            } else {
                e.done(c.o);
                if (!m.encode_exponent_global_half(dist[k], sym[k], e)) return false;
                c.o = e.cursor(&o_end);
            }
        }
        e.done(c.o);
        return true;
    }
};
struct DarkMergeSide5 {
    UReader &half_a, &half_b, &mantissa;
    const uint64_t *inv;
    static inline UEvent mixed(const UEvent &a, const UEvent &b, const uint64_t *inv) {  // table::SumProxy::new(1, a, 2, b, 0) of the two halves
        const uint32_t total = (a.to & 0xFFFFu) + 2u * (b.to & 0xFFFFu), from = a.from + 2u * b.from, freq = (a.to >> 16) + 2u * (b.to >> 16);
        static_assert(3u * kModelThreshold <= (1u << 14), "a + 2 b below the reciprocal table's 2^14 entries: the mask must never bite");
        return UEvent{inv[total & 0x3FFFu], from, from + freq};  // (totals stay below 3 * 2^12)
    }
    static inline UEvent mixed_bit(const UEvent &a, const UEvent &b, bool one) {
        const uint32_t zero = (a.from + b.from) >> 1;
        return UEvent{1ull << 52, one ? zero : 0u, one ? 4096u : zero};
    }
    bool encode(uint32_t dist, uint8_t, USink &e) {
        if (dist >= 0x7FFFFFFFu) return false;
        const unsigned log = bit_length(dist + 1);
        const UEvent *a = half_a.next(), *b = half_b.next();
        if (!a || !b) return false;
        if (!e.raw(mixed(*a, *b, inv))) return false;
        for (unsigned i = 0; i + kMaxLogCode <= log; ++i) {
            a = half_a.next();
            b = half_b.next();
            if (!a || !b) return false;
            if (!e.raw(mixed_bit(*a, *b, i + kMaxLogCode < log))) return false;
        }
        if (!DarkMergeSide::move(mantissa, e, DarkModel::modelled_mantissa_bits(dist))) return false;
        if (log > 4) DarkMergeSide::flat_tail(e, dist + 1, log);
        return true;
    }
    bool bulk(const uint32_t *dist, const uint8_t *, size_t count, USink &e) {  // (see DarkMergeSide::bulk)
        const UEvent *pa_end, *pb_end, *pm_end, *pa = half_a.cursor(&pa_end), *pb = half_b.cursor(&pb_end), *pm = mantissa.cursor(&pm_end);
        UEvent *o_end, *o = e.cursor(&o_end);
        for (size_t k = 0; k < count; ++k) {
            const uint32_t d = dist[k];
            const unsigned log = bit_length(d + 1);
            const unsigned nm = log > 3 ? 3 : log - 1;
            if (__builtin_expect(log - 1u <= 6u && pa < pa_end && pb < pb_end && pm + 4 <= pm_end && o + 12 <= o_end, 1)) {  // log 1 .. 7: one mixed decision, flat bits <= 3
                *o++ = mixed(*pa++, *pb++, inv);
                std::memcpy(static_cast<void *>(o), pm, 4 * sizeof(UEvent));
                o += nm;
                pm += nm;
                const uint32_t v = d + 1;
                const unsigned left = log > 4 ? log - 4 : 0;
#pragma GCC unroll 4
                for (unsigned j = 0; j < 4; ++j) {
                    const uint32_t one = (v >> ((left - 1 - j) & 31u)) & 1u;
                    o[j] = UEvent{1ull << 52, one << 11, 2048u + (one << 11)};
                }
                o += left;
                continue;
            }
            half_a.done(pa);
            half_b.done(pb);
            mantissa.done(pm);
            e.done(o);
            if (!encode(d, 0, e)) return false;
            pa = half_a.cursor(&pa_end);
            pb = half_b.cursor(&pb_end);
            pm = mantissa.cursor(&pm_end);
            o = e.cursor(&o_end);
        }
        half_a.done(pa);
        half_b.done(pb);
        mantissa.done(pm);
        e.done(o);
        return true;
    }
};

// the calling thread: the range coder over ring U
// (kept out of line: inlined into the callers' large bodies its loop lost registers to spills -- `low` went through the stack)
__attribute__((noinline)) int code_uniform(URing &ring, uint8_t *out, size_t cap, size_t *out_len) {
    RangeState rs;
    size_t len = 0;
    int err = DK_OK;
    for (int slot = 0;; slot = (slot + 1) % URing::kSlots) {
        const uint32_t v = ring.acquire_full(slot);
        const uint32_t count = v & ~URing::kEnd;
        const UEvent *ev = ring.slot(slot);
        if (!err && len + 4 * static_cast<size_t>(count) + 12 <= cap) {  // room for the worst case of this batch: unchecked stores
            // The coder's state as (low, span = hi - low): the new span is r * (to - from) shifted like the borders -- computed beside the
            // two border products instead of by a subtraction after the shift -- so the chain from one decision to the next is
            //     span -> r (mulx) -> low + r * from (imul, add) -> xor with the upper border -> clz & 24 -> shifted span.
            uint8_t *p = out + len;
            uint32_t low = rs.low, span = rs.hi - rs.low;
            // The quotient of the NEXT decision is asked for before this one's byte shift is known: five decisions in six shift nothing out and
            // nearly all the others one byte, so r' = (s * inv') >> 64 and ((s << 8) * inv') >> 64 are both started as soon as the new span s is
            // there, and the count of leading zeros only picks one of them.  The chain from one decision to the next is then
            //     r -> s = r * width (imul) -> r' (mulx) -> select,
            // eight cycles where the straight form (mulx -> imul, add -> xor -> clz & 24 -> shift -> mulx) has twelve; a shift of two or three
            // bytes, or a threshold cut, computes its quotient again (rare, predicted).
            uint32_t r = count ? static_cast<uint32_t>((static_cast<unsigned __int128>(span) * ev[0].inv) >> 64) : 0u;
            for (uint32_t k = 0; k < count; ++k) {
                const uint64_t inv_next = ev[k + 1 < count ? k + 1 : k].inv;  // (the last decision of a batch: its own again, never used)
                uint32_t lo = low + r * ev[k].from, h = low + r * ev[k].to, s = r * (ev[k].to - ev[k].from);
                const uint32_t r_stay = static_cast<uint32_t>((static_cast<unsigned __int128>(s) * inv_next) >> 64);
                uint32_t r_byte = static_cast<uint32_t>((static_cast<unsigned __int128>(static_cast<uint64_t>(s) << 8) * inv_next) >> 64);
                // (both products, then the choice: left alone the compiler chooses between s and s << 8 first and multiplies once -- the shift
                // count back on the chain in front of the multiply)
                uint32_t r_stay_ = r_stay;
                __asm__("" : "+r"(r_stay_), "+r"(r_byte));
                const uint32_t x = lo ^ h;
                if (__builtin_expect(r == 0 || x == 0, 0)) { err = DK_E_INTERNAL; break; }
                const unsigned sh = static_cast<unsigned>(__builtin_clz(x)) & 24u;
                const uint32_t be = __builtin_bswap32(lo);
                std::memcpy(p, &be, 4);
                p += sh >> 3;
                lo <<= sh;
                s <<= sh;
                r = sh ? r_byte : r_stay_;
                if (__builtin_expect(sh > 8u || s <= kRangeThreshold, 0)) {
                    if (s <= kRangeThreshold) {  // threshold cut: the reference's loop from its second round on
                        h = lo + s;
                        int shifted = static_cast<int>(sh >> 3);
                        for (;;) {
                            const uint32_t lim = h & kTopMask;
                            if (h - lim >= lim - lo) lo = lim; else h = lim - 1;
                            const uint32_t x2 = lo ^ h;
                            if (x2 == 0) { err = DK_E_INTERNAL; break; }
                            const unsigned sh2 = static_cast<unsigned>(__builtin_clz(x2)) & 24u;
                            const uint32_t be2 = __builtin_bswap32(lo);
                            std::memcpy(p, &be2, 4);
                            p += sh2 >> 3;
                            shifted += static_cast<int>(sh2 >> 3);
                            if (shifted > 4) { err = DK_E_INTERNAL; break; }
                            lo <<= sh2;
                            h <<= sh2;
                            if (h - lo > kRangeThreshold) break;
                        }
                        if (err) break;
                        s = h - lo;
                    }
                    r = static_cast<uint32_t>((static_cast<unsigned __int128>(s) * inv_next) >> 64);
                }
                low = lo;
                span = s;
            }
            rs.low = low;
            rs.hi = low + span;
            len = static_cast<size_t>(p - out);
        } else if (!err) {  // close to the end of the caller's buffer: every event through a scratch word, exact check
            for (uint32_t k = 0; k < count; ++k) {
                uint8_t tmp[8];
                const uint32_t span = rs.hi - rs.low;
                const uint32_t r = static_cast<uint32_t>((static_cast<unsigned __int128>(span) * ev[k].inv) >> 64);
                const int nb = r ? rs.narrow(r, ev[k].from, ev[k].to, tmp) : -1;
                if (nb < 0) { err = DK_E_INTERNAL; break; }
                if (len + static_cast<size_t>(nb) > cap) { err = DK_E_CAPACITY; break; }  // (keep draining so the helpers can end)
                std::memcpy(out + len, tmp, static_cast<size_t>(nb));
                len += static_cast<size_t>(nb);
            }
        }
        ring.release(slot);
        if (v & URing::kEnd) break;
    }
    if (!err) {
        if (len + 4 > cap) err = DK_E_CAPACITY;
        else for (int i = 0; i < 4; ++i) out[len++] = static_cast<uint8_t>(rs.low >> (24 - 8 * i));  // ari::Encoder::finish
    }
    *out_len = len;
    return err;
}

// Five stages where the L3 group has five cores for it.  DK_E_NODEVICE: no such group (or a helper thread could not be started).
int encode_five_stages(DarkModel &model, const DcStream &s, uint8_t *out, size_t cap, size_t *out_len) {
    ThreadPair tp;
    if (!tp.acquire(5)) return DK_E_NODEVICE;
    const uint64_t *inv = reciprocal_table();
    URing ring_a, ring_b, ring_m, ring_u;
    int rc_a = DK_OK, rc_b = DK_OK, rc_m = DK_OK, rc_u = DK_OK;
    auto global_model = std::make_unique<DarkModel>(), mantissa_model = std::make_unique<DarkModel>();  // (own objects: no cache line shared between the threads' states)
    std::thread t_a, t_b, t_m, t_u;
    auto run_a = [&] { tp.pin_partner(); USink sink(ring_a); DarkSymbolHalfSide side{model}; rc_a = write_stream(side, s, sink); sink.finish(); };
    auto run_b = [&] { tp.pin_partner(); USink sink(ring_b); DarkGlobalHalfSide side{*global_model}; rc_b = write_stream(side, s, sink); sink.finish(); };
    auto run_m = [&] { tp.pin_partner(); USink sink(ring_m); DarkMantissaSide side{*mantissa_model}; rc_m = write_stream(side, s, sink); sink.finish(); };
    auto run_u = [&] {
        tp.pin_partner();
        UReader rd_a(ring_a), rd_b(ring_b), rd_m(ring_m);
        USink sink(ring_u);
        DarkMergeSide5 side{rd_a, rd_b, rd_m, inv};
        rc_u = write_stream(side, s, sink);
        sink.finish();
        rd_a.drain();
        rd_b.drain();
        rd_m.drain();
    };
    int started = 0;
    try {
        t_a = std::thread(run_a); ++started;
        t_b = std::thread(run_b); ++started;
        t_m = std::thread(run_m); ++started;
        t_u = std::thread(run_u); ++started;
    } catch (...) {
        // let whatever did start run to its end: somebody has to empty the rings it fills
        if (started >= 1) { UReader rd(ring_a); rd.drain(); t_a.join(); }
        if (started >= 2) { UReader rd(ring_b); rd.drain(); t_b.join(); }
        if (started >= 3) { UReader rd(ring_m); rd.drain(); t_m.join(); }
        tp.release();
        return DK_E_NODEVICE;
    }
    const int rc = code_uniform(ring_u, out, cap, out_len);
    t_u.join();
    t_a.join();
    t_b.join();
    t_m.join();
    tp.release();
    if (getenv("DK_TRACE"))
        fprintf(stderr, "[dark_amd] entropy: five stages inside the L3 group of cpu %d; waits (pause iterations): symbol half %llu, global half %llu, "
                        "mantissa model %llu, merger in %llu + %llu + %llu out %llu, coder %llu\n", tp.me, (unsigned long long)ring_a.producer_spins,
                (unsigned long long)ring_b.producer_spins, (unsigned long long)ring_m.producer_spins, (unsigned long long)ring_a.consumer_spins,
                (unsigned long long)ring_b.consumer_spins, (unsigned long long)ring_m.consumer_spins, (unsigned long long)ring_u.producer_spins,
                (unsigned long long)ring_u.consumer_spins);
    return rc_a ? rc_a : (rc_b ? rc_b : (rc_m ? rc_m : (rc_u ? rc_u : rc)));
}
template <class M>
int encode_five_stages(M &, const DcStream &, uint8_t *, size_t, size_t *) { return DK_E_NODEVICE; }

// DK_E_NODEVICE: fewer than four cores in the caller's L3 group (or a helper thread could not be started)
int encode_four_stages(DarkModel &model, const DcStream &s, uint8_t *out, size_t cap, size_t *out_len) {
    ThreadPair tp;
    if (!tp.acquire(4)) return DK_E_NODEVICE;
    (void)reciprocal_table();
    URing ring_e, ring_m, ring_u;
    int rc_e = DK_OK, rc_m = DK_OK, rc_u = DK_OK;
    auto mantissa_model = std::make_unique<DarkModel>();  // its own object: no cache line shared with the exponent thread's state
    std::thread t_e, t_m, t_u;
    auto run_e = [&] {
        tp.pin_partner();
        USink sink(ring_e);
        DarkExponentSide side{model};
        rc_e = write_stream(side, s, sink);
        sink.finish();
    };
    auto run_m = [&] {
        tp.pin_partner();
        USink sink(ring_m);
        DarkMantissaSide side{*mantissa_model};
        rc_m = write_stream(side, s, sink);
        sink.finish();
    };
    auto run_u = [&] {
        tp.pin_partner();
        UReader rd_e(ring_e), rd_m(ring_m);
        USink sink(ring_u);
        DarkMergeSide side{rd_e, rd_m};
        rc_u = write_stream(side, s, sink);
        sink.finish();
        rd_e.drain();
        rd_m.drain();
    };
    int started = 0;
    try {
        t_e = std::thread(run_e); ++started;
        t_m = std::thread(run_m); ++started;
        t_u = std::thread(run_u); ++started;
    } catch (...) {
        // let whatever did start run to its end: somebody has to empty the rings it fills
        if (started >= 1) { UReader rd(ring_e); rd.drain(); t_e.join(); }
        if (started >= 2) { UReader rd(ring_m); rd.drain(); t_m.join(); }
        tp.release();
        return DK_E_NODEVICE;
    }
    const int rc = code_uniform(ring_u, out, cap, out_len);
    t_u.join();
    t_e.join();
    t_m.join();
    tp.release();
    if (getenv("DK_TRACE"))
        fprintf(stderr, "[dark_amd] entropy: four stages inside the L3 group of cpu %d; waits (pause iterations): exponent model %llu, mantissa "
                        "model %llu, merger in %llu + %llu out %llu, coder %llu\n", tp.me, (unsigned long long)ring_e.producer_spins,
                (unsigned long long)ring_m.producer_spins, (unsigned long long)ring_e.consumer_spins, (unsigned long long)ring_m.consumer_spins,
                (unsigned long long)ring_u.producer_spins, (unsigned long long)ring_u.consumer_spins);
    return rc_e ? rc_e : (rc_m ? rc_m : (rc_u ? rc_u : rc));
}
template <class M>
int encode_four_stages(M &, const DcStream &, uint8_t *, size_t, size_t *) { return DK_E_NODEVICE; }  // only the dark model splits this way

// CPUs this process may burn according to its cgroup (v2 cpu.max, v1 cfs quota); a huge number when unlimited or unknown.  The
// pipeline stages busy-wait, so a pipeline wider than the quota would only be throttled.
int cgroup_cpu_budget() {
    static const int budget = [] {
        long long quota = -1, period = 100000;
        if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64];
            if (std::fscanf(f, "%63s %lld", q, &period) >= 1 && q[0] != 'm') quota = std::atoll(q);
            std::fclose(f);
        } else if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (std::fscanf(g, "%lld", &quota) != 1) quota = -1;
            std::fclose(g);
            if (FILE *h = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (std::fscanf(h, "%lld", &period) != 1) period = 100000;
                std::fclose(h);
            }
        }
        if (quota <= 0 || period <= 0) return 1 << 20;
        return static_cast<int>(std::max<long long>(1, quota / period));
    }();
    return budget;
}

// 0 = automatic, 1 = one thread, 2 = models | coder, 4 = the four-stage pipeline (dark model), 5 = five stages (the exponent model in two
// halves), each whenever the cores exist (else the next narrower form).
// Process-wide: DK_ENTROPY_THREADS at load time, dk_set_entropy_threads() afterwards (a launcher that has compared what every rank of a
// node can claim sets the same form on all of them, so that the slowest rank is not decided by who wins the race for an L3 group).
std::atomic<int> g_thread_mode{[] { const char *e = getenv("DK_ENTROPY_THREADS"); return e ? atoi(e) : 0; }()};
int entropy_thread_mode() { return g_thread_mode.load(std::memory_order_relaxed); }

}  // namespace

const uint64_t *reciprocal_table14() { return reciprocal_table(); }

static thread_local int t_last_threads = 1;
static thread_local int t_last_group = -1, t_last_group_numa = -1, t_pref_numa = -1;
int last_entropy_threads() { return t_last_threads; }
namespace { void note_group(int id, int numa) { t_last_group = id; t_last_group_numa = numa; } int preferred_numa() { return t_pref_numa; } }
int last_entropy_group() { return t_last_threads > 1 ? t_last_group : -1; }
int last_entropy_group_numa() { return t_last_threads > 1 ? t_last_group_numa : -1; }
void set_preferred_numa(int node) { t_pref_numa = node; }
// Order in which a coding pass tries to claim the machine's L3 groups (indices into the topology's list; numa[g] = memory node of group g, -1
// unknown): the groups on the preferred node first -- the node the block's GPU hangs on: the pinned staging the coder reads was allocated by
// the rank's first touch, next to that GPU, and with eight ranks on two sockets a group across the socket link reads all of it remotely -- and
// within each class the caller's own group, then the others by distance (own + 1, own - 1, own + 2, ...).  preferred < 0, or no group there: by
// distance alone.
std::vector<size_t> l3_claim_order(const std::vector<int> &numa, size_t own, int preferred) {
    std::vector<size_t> by_distance;
    const size_t count = numa.size();
    if (own >= count) return by_distance;
    for (size_t step = 0; step < 2 * count; ++step) {
        const long off = (step & 1) ? static_cast<long>((step + 1) / 2) : -static_cast<long>(step / 2);
        const long gi = static_cast<long>(own) + off;
        if (gi < 0 || gi >= static_cast<long>(count) || (step > 0 && off == 0)) continue;
        by_distance.push_back(static_cast<size_t>(gi));
    }
    if (preferred < 0) return by_distance;
    std::vector<size_t> order;
    for (const size_t g : by_distance) if (numa[g] == preferred) order.push_back(g);
    for (const size_t g : by_distance) if (numa[g] != preferred) order.push_back(g);
    return order;
}
int set_entropy_thread_mode(int mode) {
    if (mode != 0 && mode != 1 && mode != 2 && mode != 4 && mode != 5) return DK_E_ARG;
    g_thread_mode.store(mode, std::memory_order_relaxed);
    return DK_OK;
}
// last-level-cache groups in which the calling thread's affinity mask leaves at least min_cores cores (what ThreadPair::acquire looks for)
int host_l3_groups(int min_cores) {
#if defined(__linux__)
    cpu_set_t mask;
    if (min_cores < 1 || pthread_getaffinity_np(pthread_self(), sizeof(mask), &mask) != 0) return 0;
    const Topology &topo = topology();
    int count = 0;
    for (const L3Group &g : topo.groups) {
        cpu_set_t usable;
        CPU_AND(&usable, &g.primary, &mask);
        size_t cores = static_cast<size_t>(CPU_COUNT(&usable));
        if (cores < static_cast<size_t>(min_cores)) {
            CPU_AND(&usable, &g.cpus, &mask);
            cores = (static_cast<size_t>(CPU_COUNT(&usable)) + topo.threads_per_core - 1) / topo.threads_per_core;
        }
        count += cores >= static_cast<size_t>(min_cores) ? 1 : 0;
    }
    return count;
#else
    (void)min_cores;
    return 0;
#endif
}

int encode_block_stream(int model_id, const DcStream &s, uint8_t *out, size_t cap, size_t *out_len, int host_threads) {
    if (!s.init || (!s.dist && s.m) || (!s.sym && s.m) || !out || !out_len) return DK_E_ARG;
    if (model_id == DK_MODEL_RAWDC) return write_records(s, out, cap, out_len);
    if (s.n > model_max_block(model_id)) return DK_E_MODEL;
    return with_model(model_id, [&](auto &model) {
        model.reset();  // Encoder::new resets the model (src/block/dc.rs:31)
        const int mode = host_threads ? host_threads : entropy_thread_mode();
        t_last_threads = 1;
        t_last_group = -1;
        const bool large = s.m >= (1u << 21);
        const int budget = cgroup_cpu_budget();
        if (mode == 5 || (mode == 0 && large && budget >= 5)) {
            const int rc5 = encode_five_stages(model, s, out, cap, out_len);
            if (rc5 != DK_E_NODEVICE) { t_last_threads = 5; return rc5; }
            model.reset();  // not this model, or fewer than five cores: try four stages
        }
        if (mode == 4 || mode == 5 || (mode == 0 && large && budget >= 4)) {
            const int rc4 = encode_four_stages(model, s, out, cap, out_len);
            if (rc4 != DK_E_NODEVICE) { t_last_threads = 4; return rc4; }
            model.reset();  // not this model, or fewer than four cores: try two threads
        }
        if (mode == 2 || mode == 4 || mode == 5 || (mode == 0 && large && budget >= 2)) {
            const int rc2 = encode_two_threads(model, s, out, cap, out_len);
            if (rc2 != DK_E_NODEVICE) { t_last_threads = 2; return rc2; }
            model.reset();  // no partner core: fall through to the single-thread coder
        }
        Encoder e(out, cap);
        int rc = write_stream(model, s, e);
        *out_len = e.size();
        return rc;
    });
}

int decode_block_stream(int model_id, const uint8_t *in, size_t in_len, size_t n, uint8_t *bwt_out,
                        uint32_t *origin, int *single, size_t *consumed) {
    if (!in || !bwt_out || !origin || n == 0) return DK_E_ARG;
    return with_model(model_id, [&](auto &model) {
        model.reset();  // src/block/dc.rs:108
        Decoder d(in, in_len);
        int rc = read_stream(model, d, n, bwt_out, origin, single);
        if (consumed) *consumed = d.consumed();
        return rc;
    });
}

int dc_decode_array(const uint32_t init[256], const uint32_t *dist, size_t m, uint8_t *bwt_out, size_t n, size_t *consumed) {
    size_t k = 0;
    int rc = dc_rebuild(init, bwt_out, n, [&](uint8_t, uint32_t *out) {
        if (k >= m) return DK_E_STREAM;
        *out = dist[k++];
        return DK_OK;
    }, nullptr);
    if (consumed) *consumed = k;
    return rc;
}

int model_encode_stream(int model_id, const uint32_t *dist, const uint8_t *sym, size_t m, uint8_t *out, size_t cap, size_t *out_len) {
    return with_model(model_id, [&](auto &model) {
        model.reset();
        Encoder e(out, cap);
        for (size_t k = 0; k < m; ++k)
            if (!model.encode(dist[k], sym[k], e)) { *out_len = e.size(); return e.error() ? e.error() : DK_E_MODEL; }
        bool ok = e.finish();
        *out_len = e.size();
        return ok ? DK_OK : e.error();
    });
}

int model_decode_stream(int model_id, const uint8_t *in, size_t in_len, const uint8_t *sym, size_t m, uint32_t *dist) {
    return with_model(model_id, [&](auto &model) {
        model.reset();
        Decoder d(in, in_len);
        for (size_t k = 0; k < m; ++k)
            if (!model.decode(sym[k], d, dist[k])) return DK_E_STREAM;
        return DK_OK;
    });
}

// ------------------------------------------------------------------------------------------------------------------
// Bitwise coder of src/entropy/ari.rs (Range) driven as src/entropy/mod.rs does
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct BitRange {
    uint32_t lo = 0, hi = 0xFFFFFFFFu;
    inline uint32_t mid(uint16_t p) const {  // ari.rs:21-30: split point for a 12-bit probability of zero
        const uint32_t f = p + 1u - (p >> 11);
        const uint32_t diff = hi - lo;
        return lo + (diff >> 12) * f + (((diff & 0xFFFu) * f) >> 12);
    }
    template <class Emit> inline size_t roll(Emit &&emit) {  // ari.rs:32-42
        size_t k = 0;
        while (((lo ^ hi) & kTopMask) == 0) { emit(static_cast<uint8_t>(lo >> 24)); ++k; lo <<= 8; hi = (hi << 8) | 0xFF; }
        return k;
    }
};
}  // namespace

int bitcoder_encode(const uint8_t *bits, const uint16_t *flat, size_t nbits, uint8_t *out, size_t cap, size_t *out_len) {
    BitRange r;
    size_t len = 0;
    for (size_t k = 0; k < nbits; ++k) {
        const uint32_t m = r.mid(flat[k]);
        if (bits[k] == 0) r.hi = m; else r.lo = m + 1;  // ari.rs:44-53
        if (len + 4 > cap) return DK_E_CAPACITY;
        r.roll([&](uint8_t b) { out[len++] = b; });
    }
    if (len + 4 > cap) return DK_E_CAPACITY;
    for (int i = 0; i < 4; ++i) out[len++] = static_cast<uint8_t>(r.lo >> (24 - 8 * i));  // post_encode ari.rs:67-73
    *out_len = len;
    return DK_OK;
}

int bitcoder_decode(const uint8_t *in, size_t in_len, const uint16_t *flat, size_t nbits, uint8_t *bits) {
    BitRange r;
    uint32_t code = 0;
    size_t pos = 0, pending = 4;  // entropy/mod.rs:52-68
    for (size_t k = 0; k < nbits; ++k) {
        while (pending) { if (pos >= in_len) return DK_E_STREAM; code = (code << 8) + in[pos++]; --pending; }
        const uint32_t m = r.mid(flat[k]);  // ari.rs:55-65
        if (code <= m) { r.hi = m; bits[k] = 0; } else { r.lo = m + 1; bits[k] = 1; }
        pending = r.roll([](uint8_t) {});
    }
    return DK_OK;
}

}  // namespace dk
