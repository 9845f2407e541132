/* TEST INFRASTRUCTURE ONLY -- see dark_oracle.h for the scope and the pinning status.
 *
 * Plain-C restatement of the reference CPU path.  Every function cites the reference lines it follows
 * (paths are relative to /root/reference).  Functions whose bodies live in the un-vendored crate
 * `compress` 0.1 are marked [compress] and restate that crate's published algorithm; they are anchored on the
 * reference's call sites and on the in-repo C++ analogue etc/dark-c (same author, same log strings).
 */
#include "dark_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define SUF_INVALID 0xFFFFFFFFu /* saca.rs:22 */

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static __thread double g_stage[4];
void orc_last_stage_seconds(double out[4]) { memcpy(out, g_stage, sizeof g_stage); }

/* ------------------------------------------------------------------------------------------------
 * Suffix array
 * ---------------------------------------------------------------------------------------------- */
#define SYM uint32_t
#define FN(x) x##_u32
static int saca_u32(const uint32_t *input, size_t n, size_t alphabet_size, uint32_t *storage, size_t storage_len);
#include "sais_body.inc"
#undef SYM
#undef FN
#define SYM uint8_t
#define FN(x) x##_u8
#include "sais_body.inc"
#undef SYM
#undef FN

/* saca.rs:351-354 Constructor::new */
size_t orc_saca_storage_words(size_t n) {
    size_t extra_2s = ((size_t)1 << 15) + ((size_t)1 << 7);
    size_t a = n / 4, b = extra_2s < n / 2 ? extra_2s : n / 2;
    size_t extra = 0x100 + (a > b ? a : b);
    return n + extra;
}

/* saca.rs:368-378 Constructor::compute */
int orc_sa_sais(const uint8_t *t, size_t n, uint32_t *sa) {
    if (n == 0 || n >= 0xFFFFFFFFu) return -1; /* reference panics at saca.rs:107 for n == 0 */
    size_t words = orc_saca_storage_words(n);
    uint32_t *storage = (uint32_t *)calloc(words, sizeof(uint32_t));
    if (!storage) return -9;
    int rc = saca_u8(t, n, 0x100, storage, words);
    if (!rc) memcpy(sa, storage, n * sizeof(uint32_t));
    free(storage);
    return rc;
}

/* saca.rs:25-35 sort_direct: slices compare lexicographically, a proper prefix is smaller */
static const uint8_t *g_naive_t;
static size_t g_naive_n;
static int naive_cmp(const void *pa, const void *pb) {
    uint32_t a = *(const uint32_t *)pa, b = *(const uint32_t *)pb;
    size_t la = g_naive_n - a, lb = g_naive_n - b;
    size_t l = la < lb ? la : lb;
    int c = memcmp(g_naive_t + a, g_naive_t + b, l);
    if (c) return c;
    return la < lb ? -1 : (la > lb ? 1 : 0);
}
int orc_sa_naive(const uint8_t *t, size_t n, uint32_t *sa) {
    if (n == 0) return -1;
    for (size_t i = 0; i < n; i++) sa[i] = (uint32_t)i;
    g_naive_t = t;
    g_naive_n = n;
    qsort(sa, n, sizeof(uint32_t), naive_cmp);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * BWT  [compress] bwt::TransformIterator / bwt::decode
 * ---------------------------------------------------------------------------------------------- */
/* Contract forced by saca.rs:411-412: L[i] = T[SA[i]-1], or T[n-1] where SA[i]==0; origin = that i. */
int orc_bwt_forward(const uint8_t *t, size_t n, const uint32_t *sa, uint8_t *bwt, uint32_t *origin) {
    int seen = 0;
    for (size_t i = 0; i < n; i++) {
        uint32_t p = sa[i];
        if (p == 0) {
            if (seen) return -1; /* TransformIterator asserts origin.is_none() */
            seen = 1;
            *origin = (uint32_t)i;
            bwt[i] = t[n - 1];
        } else {
            bwt[i] = t[p - 1];
        }
    }
    return seen ? 0 : -1;
}

/* [compress] bwt::compute_inversion_table + InverseIterator.  The origin element is placed FIRST in its
 * symbol class, then positions 0..origin, then origin+1..n (archon3.cpp:74-81 does the same); the walk
 * starts at origin and follows table[cur]-1 n times (archon3.cpp:82-85). */
int orc_bwt_inverse(const uint8_t *bwt, size_t n, uint32_t origin, uint8_t *out) {
    if (n == 0 || origin >= n) return -1;
    uint32_t *table = (uint32_t *)malloc(n * sizeof(uint32_t));
    if (!table) return -9;
    size_t freq[257];
    memset(freq, 0, sizeof freq);
    for (size_t i = 0; i < n; i++) freq[bwt[i]] += 1;       /* Radix::gather */
    {                                                       /* Radix::accumulate */
        size_t acc = 0;
        for (int s = 0; s <= 256; s++) { size_t f = freq[s]; freq[s] = acc; acc += f; }
    }
    table[freq[bwt[origin]]++] = 0;
    for (size_t i = 0; i < origin; i++) table[freq[bwt[i]]++] = (uint32_t)(i + 1);
    for (size_t i = (size_t)origin + 1; i < n; i++) table[freq[bwt[i]]++] = (uint32_t)(i + 1);
    size_t cur = origin, k = 0;
    int rc = -1;
    while (k < n) {
        cur = (size_t)table[cur] - 1; /* wraps to (size_t)-1 on the terminal entry */
        if (cur == (size_t)-1) {
            out[k++] = bwt[origin];
            rc = (k == n) ? 0 : -1;
            break;
        }
        out[k++] = bwt[cur];
    }
    free(table);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * MTF + DC  [compress] bwt::mtf::MTF, bwt::dc::{encode, EncodeIterator, decode, Context}
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint8_t symbols[256]; } MTF;

/* [compress] MTF::encode: rank of sym, then move to front */
static unsigned mtf_encode(MTF *m, uint8_t sym) {
    uint8_t next = m->symbols[0];
    if (next == sym) return 0;
    unsigned rank = 1;
    for (;;) {
        uint8_t t = m->symbols[rank];
        m->symbols[rank] = next;
        next = t;
        if (next == sym) break;
        rank += 1;
        if (rank >= 256) return 256; /* assert in the crate */
    }
    m->symbols[0] = sym;
    return rank;
}

typedef struct { uint8_t symbol; uint8_t last_rank; size_t distance_limit; } Ctx; /* dc::Context */

/* [compress] dc::encode: call site block/dc.rs:52; analogue ptax.cpp:61-111 (`r[lp] = cp-lp-arm-1` :97) */
static int dc_encode_sparse(const uint8_t *input, size_t n, uint32_t *distances, size_t init[256], MTF *mtf) {
    size_t last[256];
    size_t num_unique = 0;
    for (int s = 0; s < 256; s++) { last[s] = n; init[s] = n; }
    for (size_t i = 0; i < n; i++) {
        uint8_t sym = input[i];
        distances[i] = (uint32_t)n; /* filler */
        size_t base = last[sym];
        last[sym] = i;
        if (base == n) {
            mtf->symbols[num_unique] = sym;
            mtf_encode(mtf, sym); /* == num_unique */
            init[sym] = i;
            num_unique += 1;
        } else {
            size_t rank = mtf_encode(mtf, sym);
            if (rank >= 256) return -1;
            if (rank > 0) {
                if (i < base + rank + 1) return -1;
                distances[base] = (uint32_t)(i - base - rank - 1);
            }
        }
    }
    for (size_t rank = 0; rank < num_unique; rank++) { /* final sweep, in MTF order */
        uint8_t sym = mtf->symbols[rank];
        size_t base = last[sym];
        if (n < base + rank + 1) return -1;
        distances[base] = (uint32_t)(n - base - rank - 1);
    }
    return 0;
}

int orc_dc_encode(const uint8_t *bwt, size_t n, uint32_t *dist_sparse, uint32_t init_out[256],
                  uint32_t *d, uint8_t *sym, uint8_t *rank, uint32_t *limit, size_t *m) {
    if (n == 0 || n >= 0xFFFFFFFFu) return -1;
    MTF mtf;
    memset(&mtf, 0, sizeof mtf);
    size_t init[256];
    int rc = dc_encode_sparse(bwt, n, dist_sparse, init, &mtf);
    if (rc) return rc;
    for (int s = 0; s < 256; s++) init_out[s] = (uint32_t)init[s];
    /* [compress] EncodeIterator::next */
    size_t pos[256];
    memcpy(pos, init, sizeof pos);
    size_t last_active = 0, k = 0;
    for (size_t i = 0; i < n; i++) {
        if (dist_sparse[i] == (uint32_t)n) continue;
        uint8_t s = bwt[i];
        size_t r = last_active - pos[s];
        if (r >= 256) return -2;
        last_active = i + 1;
        pos[s] = i + 1 + dist_sparse[i];
        if (d) d[k] = dist_sparse[i];
        if (sym) sym[k] = s;
        if (rank) rank[k] = (uint8_t)r;
        if (limit) limit[k] = (uint32_t)(n - i);
        k++;
    }
    *m = k;
    return 0;
}

typedef int (*dist_fn)(void *user, const Ctx *ctx, size_t *out);

/* [compress] dc::decode: call site block/dc.rs:146-150; analogue ptax.cpp:113-146 */
static int dc_decode(size_t next[256], uint8_t *output, size_t n, MTF *mtf, dist_fn fn, void *user) {
    size_t i = 0;
    for (int sym = 0; sym < 256; sym++) { /* insertion sort of present symbols by first position */
        size_t d = next[sym];
        if (d < n) {
            size_t j = i;
            while (j > 0 && next[mtf->symbols[j - 1]] > d) {
                mtf->symbols[j] = mtf->symbols[j - 1];
                j -= 1;
            }
            mtf->symbols[j] = (uint8_t)sym;
            i += 1;
        }
    }
    if (i <= 1) { /* redundant alphabet case: no distance is read */
        memset(output, mtf->symbols[0], n);
        return 0;
    }
    size_t alphabet_size = i;
    uint8_t ranks[256];
    memset(ranks, 0, sizeof ranks);
    i = 0;
    while (i < n) {
        uint8_t sym = mtf->symbols[0];
        size_t stop = next[mtf->symbols[1]];
        if (stop > n) return -1;
        while (i < stop) output[i++] = sym;
        Ctx ctx = {sym, ranks[sym], n + 1 - i};
        size_t dd;
        int rc = fn(user, &ctx, &dd);
        if (rc) return rc;
        size_t future = stop + dd;
        if (future > n) return -1; /* assert!(future <= n) */
        size_t rank = 1;
        while (rank < alphabet_size && future + rank > next[mtf->symbols[rank]]) {
            mtf->symbols[rank - 1] = mtf->symbols[rank];
            rank += 1;
        }
        mtf->symbols[rank - 1] = sym;
        next[sym] = future + rank - 1;
        ranks[sym] = (uint8_t)(rank - 1);
    }
    for (int s = 0; s < 256; s++) {
        if (next[s] < n || next[s] >= n + alphabet_size) return -1; /* final assert of the crate */
    }
    return i == n ? 0 : -1;
}

typedef struct { const uint32_t *d; size_t m, k; } ArrSrc;
static int arr_dist(void *user, const Ctx *ctx, size_t *out) {
    (void)ctx;
    ArrSrc *a = (ArrSrc *)user;
    if (a->k >= a->m) return -3;
    *out = a->d[a->k++];
    return 0;
}
int orc_dc_decode(const uint32_t init[256], const uint32_t *d, size_t m, uint8_t *bwt, size_t n, size_t *consumed) {
    size_t next[256];
    for (int s = 0; s < 256; s++) next[s] = init[s];
    MTF mtf;
    memset(&mtf, 0, sizeof mtf);
    ArrSrc a = {d, m, 0};
    int rc = dc_decode(next, bwt, n, &mtf, arr_dist, &a);
    if (consumed) *consumed = a.k;
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Range coder  [compress] entropy::ari::{RangeEncoder, Encoder, Decoder}
 * analogue: etc/dark-c/src/low.cpp:25-42 (ark::parse); README.md:20 says the Rust one is "improved"
 * ---------------------------------------------------------------------------------------------- */
#define RANGE_DEFAULT_THRESHOLD (1u << 14)
#define BORDER_SYMBOL_MASK 0xFF000000u

typedef struct { uint32_t low, hi, threshold; } RangeEncoder;
static void range_new(RangeEncoder *r) { r->low = 0; r->hi = 0xFFFFFFFFu; r->threshold = RANGE_DEFAULT_THRESHOLD; }

/* [compress] RangeEncoder::process(total, from, to, output) -> number of bytes shifted out */
static int range_process(RangeEncoder *r, uint32_t total, uint32_t from, uint32_t to, uint8_t *output) {
    if (!(from < to && to <= total)) return -1;
    uint32_t range = (r->hi - r->low) / total;
    if (range == 0) return -1;
    uint32_t lo = r->low + range * from;
    uint32_t hi = r->low + range * to;
    int num_shift = 0;
    for (;;) {
        if (((lo ^ hi) & BORDER_SYMBOL_MASK) != 0) {
            if (hi - lo > r->threshold) break;
            uint32_t lim = hi & BORDER_SYMBOL_MASK;
            if (hi - lim >= lim - lo) lo = lim; else hi = lim - 1;
        }
        if (num_shift >= 4) return -1; /* the crate's output buffer holds BORDER_BYTES = 4 */
        output[num_shift++] = (uint8_t)(lo >> 24);
        lo <<= 8;
        hi <<= 8;
    }
    r->low = lo;
    r->hi = hi;
    return num_shift;
}
/* [compress] RangeEncoder::query */
static uint32_t range_query(const RangeEncoder *r, uint32_t total, uint32_t code) {
    uint32_t range = (r->hi - r->low) / total;
    return (code - r->low) / range;
}

typedef struct { RangeEncoder range; uint8_t *out; size_t cap, len; int err; int sink; } Enc;
static void enc_new(Enc *e, uint8_t *out, size_t cap) { range_new(&e->range); e->out = out; e->cap = cap; e->len = 0; e->err = 0; e->sink = 0; }
/* [compress] ari::Encoder::encode: model.get_range + get_denominator -> process -> write */
static int enc_put(Enc *e, uint32_t total, uint32_t lo, uint32_t hi) {
    uint8_t buf[4];
    int num = range_process(&e->range, total, lo, hi, buf);
    if (num < 0) { e->err = -1; return -1; }
    if (e->len + (size_t)num > e->cap) { e->err = -2; return -2; }
    memcpy(e->out + e->len, buf, (size_t)num);
    e->len += (size_t)num;
    return 0;
}
/* [compress] ari::Encoder::finish: code tail = low as u32 big-endian */
static int enc_finish(Enc *e) {
    uint32_t code = e->range.low;
    if (e->len + 4 > e->cap) { e->err = -2; return -2; }
    for (int i = 0; i < 4; i++) e->out[e->len++] = (uint8_t)(code >> (24 - 8 * i));
    return e->err;
}

typedef struct { RangeEncoder range; const uint8_t *in; size_t len, pos; uint32_t code; int pending; int err; } Dec;
static void dec_new(Dec *d, const uint8_t *in, size_t len) { range_new(&d->range); d->in = in; d->len = len; d->pos = 0; d->code = 0; d->pending = 4; d->err = 0; }
/* [compress] ari::Decoder::feed */
static void dec_feed(Dec *d) {
    while (d->pending) {
        uint8_t b = 0;
        if (d->pos < d->len) b = d->in[d->pos++]; else d->err = -4; /* read_u8 error -> unwrap panics in the reference */
        d->code = (d->code << 8) + b;
        d->pending -= 1;
    }
}
/* [compress] ari::Decoder::decode, first half: offset under `total` */
static uint32_t dec_offset(Dec *d, uint32_t total) {
    dec_feed(d);
    return range_query(&d->range, total, d->code);
}
/* second half: consume the found interval */
static int dec_consume(Dec *d, uint32_t total, uint32_t lo, uint32_t hi) {
    uint8_t buf[4];
    int num = range_process(&d->range, total, lo, hi, buf);
    if (num < 0) { d->err = -1; return -1; }
    d->pending = num;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Frequency models  [compress] ari::table::{Model,SumProxy}, ari::bin::{Model,SumProxy}, ari::apm::Bit
 * analogues: low.cpp:55-64 (table update/downscale), low.cpp:119-126 (binary update)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint32_t total; uint32_t cut_threshold; int n; uint16_t table[256]; } Table;

static void table_downscale(Table *t) { /* cut_shift = 1: (f+1)>>1 keeps frequencies positive */
    t->total = 0;
    for (int i = 0; i < t->n; i++) {
        t->table[i] = (uint16_t)((t->table[i] + 1) >> 1);
        t->total += t->table[i];
    }
}
static void table_new_flat(Table *t, int num_values, uint32_t threshold) {
    t->n = num_values;
    t->cut_threshold = threshold;
    for (int i = 0; i < num_values; i++) t->table[i] = 1;
    t->total = (uint32_t)num_values;
    while (t->total >= threshold) table_downscale(t);
}
static void table_reset_flat(Table *t) {
    for (int i = 0; i < t->n; i++) t->table[i] = 1;
    t->total = (uint32_t)t->n;
}
static void table_update(Table *t, size_t value, unsigned add_log, uint32_t add_const) {
    uint32_t add = (t->total >> add_log) + add_const;
    t->table[value] = (uint16_t)(t->table[value] + add);
    t->total += add;
    if (t->total >= t->cut_threshold) table_downscale(t);
}
static int table_encode(Enc *e, const Table *t, size_t value) {
    uint32_t lo = 0;
    for (size_t i = 0; i < value; i++) lo += t->table[i];
    return enc_put(e, t->total, lo, lo + t->table[value]);
}
static int table_decode(Dec *d, const Table *t, size_t *value) {
    uint32_t offset = dec_offset(d, t->total);
    if (offset >= t->total) { d->err = -5; return -5; }
    size_t v = 0;
    uint32_t lo = 0, hi;
    while ((hi = lo + t->table[v]) <= offset) { lo = hi; v++; }
    *value = v;
    return dec_consume(d, t->total, lo, hi);
}
/* table::SumProxy::new(wa, fa, wb, fb, shift): every border/total = (wa*x1 + wb*x2) >> shift */
static int tablesum_encode(Enc *e, uint32_t wa, const Table *a, uint32_t wb, const Table *b, unsigned ws, size_t value) {
    uint32_t lo0 = 0, lo1 = 0;
    for (size_t i = 0; i < value; i++) { lo0 += a->table[i]; lo1 += b->table[i]; }
    uint32_t hi0 = lo0 + a->table[value], hi1 = lo1 + b->table[value];
    uint32_t total = (wa * a->total + wb * b->total) >> ws;
    return enc_put(e, total, (wa * lo0 + wb * lo1) >> ws, (wa * hi0 + wb * hi1) >> ws);
}
static int tablesum_decode(Dec *d, uint32_t wa, const Table *a, uint32_t wb, const Table *b, unsigned ws, size_t *value) {
    uint32_t total = (wa * a->total + wb * b->total) >> ws;
    uint32_t offset = dec_offset(d, total);
    if (offset >= total) { d->err = -5; return -5; }
    size_t v = 0;
    uint32_t lo = 0, hi;
    while ((hi = lo + ((wa * a->table[v] + wb * b->table[v]) >> ws)) <= offset) {
        lo = hi;
        v++;
        if ((int)v >= a->n) { d->err = -5; return -5; }
    }
    *value = v;
    return dec_consume(d, total, lo, hi);
}

typedef struct { uint32_t zero, total, rate; } Bin;
static void bin_new_flat(Bin *b, uint32_t threshold, uint32_t rate) { b->zero = threshold >> 1; b->total = threshold; b->rate = rate; }
static void bin_reset_flat(Bin *b) { b->zero = b->total >> 1; }
static void bin_update(Bin *b, int value) {
    if (value) b->zero -= b->zero >> b->rate;              /* update_one */
    else b->zero += (b->total - b->zero) >> b->rate;       /* update_zero */
}
static int binraw_encode(Enc *e, uint32_t zero, uint32_t total, int value) {
    return value ? enc_put(e, total, zero, total) : enc_put(e, total, 0, zero);
}
static int binraw_decode(Dec *d, uint32_t zero, uint32_t total, int *value) {
    uint32_t offset = dec_offset(d, total);
    if (offset >= total) { d->err = -5; return -5; }
    if (offset < zero) { *value = 0; return dec_consume(d, total, 0, zero); }
    *value = 1;
    return dec_consume(d, total, zero, total);
}
static int bin_encode(Enc *e, const Bin *b, int value) { return binraw_encode(e, b->zero, b->total, value); }
static int bin_decode(Dec *d, const Bin *b, int *value) { return binraw_decode(d, b->zero, b->total, value); }
/* bin::SumProxy::new(wa,a,wb,b,shift) */
static int binsum_encode(Enc *e, uint32_t wa, const Bin *a, uint32_t wb, const Bin *b, unsigned ws, int value) {
    return binraw_encode(e, (wa * a->zero + wb * b->zero) >> ws, (wa * a->total + wb * b->total) >> ws, value);
}
static int binsum_decode(Dec *d, uint32_t wa, const Bin *a, uint32_t wb, const Bin *b, unsigned ws, int *value) {
    return binraw_decode(d, (wa * a->zero + wb * b->zero) >> ws, (wa * a->total + wb * b->total) >> ws, value);
}

/* [compress] apm::Bit: 12-bit flat probability of ZERO (FLAT_BITS = 12; entropy/ari.rs:22-30 relies on it) */
#define FLAT_BITS 12
#define FLAT_TOTAL (1 << FLAT_BITS)
static void apmbit_update(uint16_t *fp, int value, int rate, int bias) {
    if (!value) { int one = FLAT_TOTAL - bias - (int)*fp; *fp = (uint16_t)(*fp + (one >> rate)); }
    else { int zero = (int)*fp - bias; *fp = (uint16_t)(*fp - (zero >> rate)); }
}

/* ------------------------------------------------------------------------------------------------
 * Distance models (src/model/{dark,exp,ybs,simple,raw}.rs -- fully in the reference)
 * ---------------------------------------------------------------------------------------------- */
typedef struct ModelVT {
    void (*reset)(void *st);
    int (*encode)(void *st, uint32_t dist, const Ctx *ctx, Enc *e);
    int (*decode)(void *st, const Ctx *ctx, Dec *d, uint32_t *out);
} ModelVT;

/* ---- dark (src/model/dark.rs) ---- */
#define MAX_LOG_CODE 8     /* dark.rs:46 */
#define MAX_LOG_CONTEXT 11 /* dark.rs:47 */
#define NUM_LAST_LOGS 3    /* dark.rs:48 */
#define MAX_BIT_CONTEXT 3  /* dark.rs:49 */
static const long long ADAPT_POWERS[9] = {6, 5, 4, 3, 2, 1, 4, 6, 4}; /* dark.rs:50 */

typedef struct { long long avg_dist; Table freq_log; Bin freq_extra[32]; } DarkSym; /* dark.rs:73-77 */
typedef struct {
    Table freq_log[MAX_LOG_CONTEXT + 1][NUM_LAST_LOGS]; /* dark.rs:107 */
    Bin freq_log_bits[2][32];                           /* dark.rs:108 */
    Bin freq_mantissa[32][MAX_BIT_CONTEXT + 1];         /* dark.rs:109 */
    DarkSym contexts[256];
    size_t last_log_token;
    unsigned update_log_global, update_log_power;
    uint32_t update_log_add;
} DarkModel;

static void dark_new(DarkModel *m) { /* dark.rs:121-150 */
    uint32_t threshold = RANGE_DEFAULT_THRESHOLD >> 2;
    for (int i = 0; i <= MAX_LOG_CONTEXT; i++)
        for (int j = 0; j < NUM_LAST_LOGS; j++) table_new_flat(&m->freq_log[i][j], MAX_LOG_CODE, threshold);
    for (int k = 0; k < 2; k++)
        for (int i = 0; i < 32; i++) bin_new_flat(&m->freq_log_bits[k][i], threshold, 2);
    for (int i = 0; i < 32; i++)
        for (int j = 0; j <= MAX_BIT_CONTEXT; j++) bin_new_flat(&m->freq_mantissa[i][j], threshold, 8);
    for (int s = 0; s < 256; s++) {
        m->contexts[s].avg_dist = 1000;
        table_new_flat(&m->contexts[s].freq_log, MAX_LOG_CODE, threshold);
        for (int i = 0; i < 32; i++) bin_new_flat(&m->contexts[s].freq_extra[i], threshold, 3);
    }
    m->last_log_token = 1;
    m->update_log_global = 12;
    m->update_log_power = 5;
    m->update_log_add = 5;
}
static void dark_reset(void *st) { /* dark.rs:160-178 */
    DarkModel *m = (DarkModel *)st;
    for (int i = 0; i <= MAX_LOG_CONTEXT; i++)
        for (int j = 0; j < NUM_LAST_LOGS; j++) table_reset_flat(&m->freq_log[i][j]);
    for (int k = 0; k < 2; k++)
        for (int i = 0; i < 32; i++) bin_reset_flat(&m->freq_log_bits[k][i]);
    for (int i = 0; i < 32; i++)
        for (int j = 0; j <= MAX_BIT_CONTEXT; j++) bin_reset_flat(&m->freq_mantissa[i][j]);
    for (int s = 0; s < 256; s++) {
        m->contexts[s].avg_dist = 1000;
        table_reset_flat(&m->contexts[s].freq_log);
        for (int i = 0; i < 32; i++) bin_reset_flat(&m->contexts[s].freq_extra[i]);
    }
    m->last_log_token = 1;
}
static size_t isize_log(uint32_t d) { /* dark.rs:152-156 */
    size_t log = 0;
    while (log < 32 && (d >> log) != 0) log++;
    return log;
}
static void darksym_update(DarkSym *c, uint32_t dist, long long log_diff) { /* dark.rs:94-101 */
    long long adapt = log_diff < -6 ? 7 : (log_diff >= 3 ? 3 : ADAPT_POWERS[6 + log_diff]);
    c->avg_dist += (adapt * ((long long)dist - c->avg_dist)) >> 3; /* arithmetic shift on isize */
}
static int dark_encode(void *st, uint32_t dist, const Ctx *ctx, Enc *e) { /* dark.rs:180-232 */
    DarkModel *m = (DarkModel *)st;
    if (dist == 0xFFFFFFFFu) return -1;
    dist += 1;
    size_t log = isize_log(dist);
    if (log >= 32) return -1; /* freq_mantissa has 32 rows (dark.rs:132) */
    DarkSym *context = &m->contexts[ctx->symbol];
    size_t avg_log = isize_log((uint32_t)context->avg_dist);
    size_t avg_log_capped = avg_log < MAX_LOG_CONTEXT ? avg_log : MAX_LOG_CONTEXT;
    { /* base part of the exponent */
        Table *sym_freq = &context->freq_log;
        size_t log_capped = (log < MAX_LOG_CODE ? log : MAX_LOG_CODE) - 1;
        Table *global_freq = &m->freq_log[avg_log_capped][m->last_log_token];
        if (tablesum_encode(e, 1, sym_freq, 2, global_freq, 0, log_capped)) return -1;
        table_update(sym_freq, log_capped, m->update_log_power, m->update_log_add);
        table_update(global_freq, log_capped, m->update_log_global, m->update_log_add);
    }
    if (log >= MAX_LOG_CODE) { /* unary extension */
        Bin *flb = m->freq_log_bits[avg_log_capped == MAX_LOG_CONTEXT ? 1 : 0];
        for (size_t i = MAX_LOG_CODE; i < log; i++) {
            Bin *bc = &context->freq_extra[i - MAX_LOG_CODE], *fc = &flb[i - MAX_LOG_CODE];
            if (binsum_encode(e, 1, bc, 1, fc, 1, 1)) return -1;
            bin_update(bc, 1);
            bin_update(fc, 1);
        }
        size_t i = log - MAX_LOG_CODE;
        Bin *bc = &context->freq_extra[i], *fc = &flb[i];
        if (binsum_encode(e, 1, bc, 1, fc, 1, 0)) return -1;
        bin_update(bc, 0);
        bin_update(fc, 0);
    }
    m->last_log_token = log < 2 ? 0 : (log < 8 ? 1 : 2);
    Bin *mantissa_context = m->freq_mantissa[log];
    for (size_t i = 1; i < log; i++) {
        int bit = (int)((dist >> (log - i - 1)) & 1);
        if (i > MAX_BIT_CONTEXT) {
            if (bin_encode(e, &mantissa_context[MAX_BIT_CONTEXT], bit)) return -1; /* never updated */
        } else {
            Bin *bc = &mantissa_context[i - 1];
            if (bin_encode(e, bc, bit)) return -1;
            bin_update(bc, bit);
        }
    }
    long long log_diff = (long long)log - (long long)avg_log_capped;
    darksym_update(context, dist - 1, log_diff);
    return 0;
}
static int dark_decode(void *st, const Ctx *ctx, Dec *d, uint32_t *out) { /* dark.rs:234-287 */
    DarkModel *m = (DarkModel *)st;
    DarkSym *context = &m->contexts[ctx->symbol];
    size_t avg_log = isize_log((uint32_t)context->avg_dist);
    size_t avg_log_capped = avg_log < MAX_LOG_CONTEXT ? avg_log : MAX_LOG_CONTEXT;
    size_t log_pre;
    {
        Table *sym_freq = &context->freq_log;
        Table *global_freq = &m->freq_log[avg_log_capped][m->last_log_token];
        size_t lg;
        if (tablesum_decode(d, 1, sym_freq, 2, global_freq, 0, &lg)) return -1;
        table_update(sym_freq, lg, m->update_log_power, m->update_log_add);
        table_update(global_freq, lg, m->update_log_global, m->update_log_add);
        log_pre = lg + 1;
    }
    size_t log = log_pre;
    if (log_pre >= MAX_LOG_CODE) {
        size_t count = 0;
        Bin *flb = m->freq_log_bits[avg_log_capped == MAX_LOG_CONTEXT ? 1 : 0];
        for (;;) {
            if (count >= 32) return -1;
            Bin *bc = &context->freq_extra[count], *fc = &flb[count];
            int bit;
            if (binsum_decode(d, 1, bc, 1, fc, 1, &bit)) return -1;
            bin_update(bc, bit);
            bin_update(fc, bit);
            if (!bit) break;
            count += 1;
        }
        log = log_pre + count;
    }
    if (log >= 32) return -1;
    m->last_log_token = log < 2 ? 0 : (log < 8 ? 1 : 2);
    Bin *mantissa_context = m->freq_mantissa[log];
    uint32_t dist = 1;
    for (size_t i = 1; i < log; i++) {
        int bit;
        if (i > MAX_BIT_CONTEXT) {
            if (bin_decode(d, &mantissa_context[MAX_BIT_CONTEXT], &bit)) return -1;
        } else {
            Bin *bc = &mantissa_context[i - 1];
            if (bin_decode(d, bc, &bit)) return -1;
            bin_update(bc, bit);
        }
        dist = (dist << 1) + (uint32_t)bit;
    }
    long long log_diff = (long long)log - (long long)avg_log_capped;
    dist -= 1;
    darksym_update(context, dist, log_diff);
    *out = dist;
    return 0;
}
static const ModelVT DARK_VT = {dark_reset, dark_encode, dark_decode};

/* ---- exp (src/model/exp.rs) ---- */
#define FIXED_BASE 8u
#define FIXED_MASK ((1u << FIXED_BASE) - 1)
#define LOG_LIMIT 10
#define LOG_DEFAULT (1u << FIXED_BASE)
#define BIT_UPDATE 5
typedef struct { uint32_t avg_log[256]; uint16_t prob[LOG_LIMIT][24]; } ExpModel;
static void exp_reset(void *st) { /* exp.rs:47-56 */
    ExpModel *m = (ExpModel *)st;
    for (int i = 0; i < 256; i++) m->avg_log[i] = LOG_DEFAULT;
    for (int i = 0; i < LOG_LIMIT; i++)
        for (int j = 0; j < 24; j++) m->prob[i][j] = FLAT_TOTAL >> 1; /* Bit::new_equal */
}
static uint32_t exp_get_log(uint32_t d) { /* exp.rs:34-43 (integer quirks kept: 8/3 = 2, 8>>12 = 0) */
    uint32_t du = d;
    if (d <= 2) return du << FIXED_BASE;
    if (d <= 4) return (3u << FIXED_BASE) + (du & 1) * (FIXED_BASE >> 1);
    if (d <= 7) return (4u << FIXED_BASE) + ((du - 5) % 3) * (FIXED_BASE / 3);
    if (d <= 12) return (5u << FIXED_BASE) + (du & 3) * (FIXED_BASE >> 2);
    return (6u << FIXED_BASE) + (du - 12) * (FIXED_BASE >> 12);
}
static int exp_encode(void *st, uint32_t dist, const Ctx *ctx, Enc *e) { /* exp.rs:58-78 */
    ExpModel *m = (ExpModel *)st;
    uint32_t log = m->avg_log[ctx->symbol];
    uint32_t w2 = log & FIXED_MASK, w1 = FIXED_MASK + 1 - w2;
    uint16_t *m1 = m->prob[log >> FIXED_BASE], *m2 = m->prob[(log >> FIXED_BASE) + 1];
    for (int i = 23; i >= 0; i--) {
        int value = (dist & (1u << i)) != 0;
        uint32_t flat = (w1 * m1[i] + w2 * m2[i]) >> FIXED_BASE;
        if (value ? enc_put(e, FLAT_TOTAL, flat, FLAT_TOTAL) : enc_put(e, FLAT_TOTAL, 0, flat)) return -1;
        apmbit_update(&m1[i], value, BIT_UPDATE, 0);
        apmbit_update(&m2[i], value, BIT_UPDATE, 0);
    }
    m->avg_log[ctx->symbol] = (3 * log + exp_get_log(dist)) >> 2;
    return 0;
}
static int exp_decode(void *st, const Ctx *ctx, Dec *d, uint32_t *out) { /* exp.rs:80-101 */
    ExpModel *m = (ExpModel *)st;
    uint32_t log = m->avg_log[ctx->symbol];
    uint32_t w2 = log & FIXED_MASK, w1 = FIXED_MASK + 1 - w2;
    uint16_t *m1 = m->prob[log >> FIXED_BASE], *m2 = m->prob[(log >> FIXED_BASE) + 1];
    uint32_t dist = 0;
    for (int i = 23; i >= 0; i--) {
        uint32_t flat = (w1 * m1[i] + w2 * m2[i]) >> FIXED_BASE;
        int value;
        if (binraw_decode(d, flat, FLAT_TOTAL, &value)) return -1;
        apmbit_update(&m1[i], value, BIT_UPDATE, 0);
        apmbit_update(&m2[i], value, BIT_UPDATE, 0);
        dist += dist + (uint32_t)value;
    }
    m->avg_log[ctx->symbol] = (3 * log + exp_get_log(dist)) >> 2;
    *out = dist;
    return 0;
}
static const ModelVT EXP_VT = {exp_reset, exp_encode, exp_decode};

/* ---- ybs (src/model/ybs.rs) ---- */
typedef struct { size_t avg_log, last_diff; } YbsSym;
typedef struct { Table table_log[13]; Table table_high; Bin bin_rest[3]; YbsSym contexts[256]; } YbsModel;
static void ybs_new(YbsModel *m) { /* ybs.rs:54-65 */
    uint32_t threshold = RANGE_DEFAULT_THRESHOLD >> 2;
    for (int i = 0; i < 13; i++) table_new_flat(&m->table_log[i], 14, threshold);
    table_new_flat(&m->table_high, 32 - 13, threshold);
    for (int i = 0; i < 3; i++) bin_new_flat(&m->bin_rest[i], threshold, 5);
    memset(m->contexts, 0, sizeof m->contexts);
}
static void ybs_reset(void *st) { /* ybs.rs:75-87 */
    YbsModel *m = (YbsModel *)st;
    for (int i = 0; i < 13; i++) table_reset_flat(&m->table_log[i]);
    table_reset_flat(&m->table_high);
    for (int i = 0; i < 3; i++) bin_reset_flat(&m->bin_rest[i]);
    memset(m->contexts, 0, sizeof m->contexts);
}
static void ybssym_update(YbsSym *c, size_t log) { /* ybs.rs:33-38 */
    size_t a = c->last_diff > 3 ? 2 : 1, b = 1;
    long long diff = (long long)log - (long long)c->avg_log;
    c->last_diff = (size_t)(diff < 0 ? -diff : diff);
    c->avg_log = (a * log + b * c->avg_log) / (a + b);
}
static int ybs_encode(void *st, uint32_t dist, const Ctx *ctx, Enc *e) { /* ybs.rs:89-127 */
    YbsModel *m = (YbsModel *)st;
    const size_t max_low_log = 12;
    size_t group;
    if (dist < 4) group = dist;
    else { size_t log = 3; while (log < 32 && (dist >> log) != 0) log++; group = log + 1; }
    YbsSym *context = &m->contexts[ctx->symbol];
    size_t con_log = context->avg_log < max_low_log ? context->avg_log : max_low_log;
    Table *freq_log = &m->table_log[con_log];
    size_t log_encoded = group < max_low_log ? group : max_low_log;
    if (table_encode(e, freq_log, log_encoded)) return -1;
    table_update(freq_log, log_encoded, 10, 1);
    ybssym_update(context, log_encoded);
    if (group < 4) return 0;
    if (group >= max_low_log) {
        size_t add = group - max_low_log;
        if ((int)add >= m->table_high.n) return -1; /* index out of bounds in the reference */
        if (table_encode(e, &m->table_high, add)) return -1;
        table_update(&m->table_high, add, 10, 1);
    }
    size_t log = group - 1;
    for (size_t i = 1; i < log; i++) {
        int bit = (int)((dist >> (log - i - 1)) & 1);
        if (i >= 3) {
            if (bin_encode(e, &m->bin_rest[2], bit)) return -1;
        } else {
            Bin *bc = &m->bin_rest[i - 1];
            if (bin_encode(e, bc, bit)) return -1;
            bin_update(bc, bit);
        }
    }
    return 0;
}
static int ybs_decode(void *st, const Ctx *ctx, Dec *d, uint32_t *out) { /* ybs.rs:129-165 */
    YbsModel *m = (YbsModel *)st;
    const size_t max_low_log = 12;
    YbsSym *context = &m->contexts[ctx->symbol];
    size_t con_log = context->avg_log < max_low_log ? context->avg_log : max_low_log;
    Table *freq_log = &m->table_log[con_log];
    size_t log_decoded;
    if (table_decode(d, freq_log, &log_decoded)) return -1;
    ybssym_update(context, log_decoded);
    table_update(freq_log, log_decoded, 10, 1);
    if (log_decoded < 4) { *out = (uint32_t)log_decoded; return 0; }
    size_t group = log_decoded;
    if (log_decoded == max_low_log) {
        size_t add;
        if (table_decode(d, &m->table_high, &add)) return -1;
        table_update(&m->table_high, add, 10, 1);
        group = max_low_log + add;
    }
    size_t log = group - 1;
    uint32_t dist = 1;
    for (size_t i = 1; i < log; i++) {
        int bit;
        if (i >= 3) {
            if (bin_decode(d, &m->bin_rest[2], &bit)) return -1;
        } else {
            Bin *bc = &m->bin_rest[i - 1];
            if (bin_decode(d, bc, &bit)) return -1;
            bin_update(bc, bit);
        }
        dist = (dist << 1) + (uint32_t)bit;
    }
    *out = dist;
    return 0;
}
static const ModelVT YBS_VT = {ybs_reset, ybs_encode, ybs_decode};

/* ---- simple (src/model/simple.rs) ---- */
typedef struct { Table freq[4]; unsigned up[4]; } SimpleModel;
static void simple_new(SimpleModel *m) { /* simple.rs:33-41 */
    uint32_t threshold = RANGE_DEFAULT_THRESHOLD >> 2;
    for (int i = 0; i < 4; i++) table_new_flat(&m->freq[i], 0x100, threshold);
    m->up[0] = 10; m->up[1] = 8; m->up[2] = 7; m->up[3] = 6;
}
static void simple_reset(void *st) {
    SimpleModel *m = (SimpleModel *)st;
    for (int i = 0; i < 4; i++) table_reset_flat(&m->freq[i]);
}
static int simple_encode(void *st, uint32_t dist, const Ctx *ctx, Enc *e) { /* simple.rs:51-64 */
    (void)ctx;
    SimpleModel *m = (SimpleModel *)st;
    size_t val = dist < 0xFF ? dist : 0xFF;
    if (table_encode(e, &m->freq[0], val)) return -1;
    table_update(&m->freq[0], val, m->up[0], 1);
    if (val == 0xFF) {
        size_t rest = (size_t)(dist - 0xFF);
        for (int i = 0; i < 3; i++) {
            size_t b = (rest >> (i * 8)) & 0xFF;
            if (table_encode(e, &m->freq[i + 1], b)) return -1;
            table_update(&m->freq[i + 1], b, m->up[i + 1], 1);
        }
    }
    return 0;
}
static int simple_decode(void *st, const Ctx *ctx, Dec *d, uint32_t *out) { /* simple.rs:66-80 */
    (void)ctx;
    SimpleModel *m = (SimpleModel *)st;
    size_t base;
    if (table_decode(d, &m->freq[0], &base)) return -1;
    table_update(&m->freq[0], base, m->up[0], 1);
    size_t u = base;
    if (base == 0xFF) {
        for (int i = 0; i < 3; i++) {
            size_t b;
            if (table_decode(d, &m->freq[i + 1], &b)) return -1;
            table_update(&m->freq[i + 1], b, m->up[i + 1], 1);
            u += b << (i * 8);
        }
    }
    *out = (uint32_t)u;
    return 0;
}
static const ModelVT SIMPLE_VT = {simple_reset, simple_encode, simple_decode};

/* ---- rawdc dump (src/model/raw.rs:12-44): 10-byte LE records (u32 d, u8 sym, u8 last_rank, u32 limit) ---- */
typedef struct { uint8_t *out; size_t cap, len; int err; } RawDc;
static void rawdc_reset(void *st) { (void)st; }
static int rawdc_encode(void *st, uint32_t d, const Ctx *c, Enc *e) {
    (void)e;
    RawDc *r = (RawDc *)st;
    if (r->len + 10 > r->cap) { r->err = -2; return -2; }
    uint8_t *p = r->out + r->len;
    uint32_t lim = (uint32_t)c->distance_limit;
    p[0] = (uint8_t)d; p[1] = (uint8_t)(d >> 8); p[2] = (uint8_t)(d >> 16); p[3] = (uint8_t)(d >> 24);
    p[4] = c->symbol; p[5] = c->last_rank;
    p[6] = (uint8_t)lim; p[7] = (uint8_t)(lim >> 8); p[8] = (uint8_t)(lim >> 16); p[9] = (uint8_t)(lim >> 24);
    r->len += 10;
    return 0;
}
static int rawdc_decode(void *st, const Ctx *c, Dec *d, uint32_t *out) { (void)st; (void)c; (void)d; *out = 0; return 0; }
static const ModelVT RAWDC_VT = {rawdc_reset, rawdc_encode, rawdc_decode};

typedef struct { const ModelVT *vt; void *st; } ModelBox;
static int model_make(int model_id, ModelBox *b) {
    switch (model_id) {
    case ORC_MODEL_DARK: { DarkModel *m = (DarkModel *)malloc(sizeof *m); if (!m) return -9; dark_new(m); b->vt = &DARK_VT; b->st = m; return 0; }
    case ORC_MODEL_EXP: { ExpModel *m = (ExpModel *)malloc(sizeof *m); if (!m) return -9; exp_reset(m); b->vt = &EXP_VT; b->st = m; return 0; }
    case ORC_MODEL_YBS: { YbsModel *m = (YbsModel *)malloc(sizeof *m); if (!m) return -9; ybs_new(m); b->vt = &YBS_VT; b->st = m; return 0; }
    case ORC_MODEL_SIMPLE: { SimpleModel *m = (SimpleModel *)malloc(sizeof *m); if (!m) return -9; simple_new(m); b->vt = &SIMPLE_VT; b->st = m; return 0; }
    default: return -10;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Block codec (src/block/dc.rs)
 * ---------------------------------------------------------------------------------------------- */
static const Ctx CTX_0 = {0, 0, 0x101}; /* block/dc.rs:16-18 */

/* block/dc.rs:52-90: DC + init-table RLE header + distances + origin + finish */
static int block_dc_encode_from_bwt(const ModelVT *vt, void *st, const uint8_t *output, size_t block_size,
                                    uint32_t origin, Enc *eh) {
    double t0 = now_s();
    uint32_t *suf = (uint32_t *)malloc(block_size * sizeof(uint32_t));
    if (!suf) return -9;
    MTF mtf;
    memset(&mtf, 0, sizeof mtf);
    size_t init[256];
    int rc = dc_encode_sparse(output, block_size, suf, init, &mtf);
    double t1 = now_s();
    g_stage[2] = t1 - t0;
    if (rc) { free(suf); return rc; }
    { /* encode init distances: block/dc.rs:54-80.  Loops stop at 0xFF: symbol 255 is never transmitted. */
        int cur_active = 1;
        size_t i = 0;
        while (i < 0xFF) {
            size_t base = i;
            if (cur_active) {
                while (i < 0xFF && init[i] < block_size) i += 1;
                uint32_t num = (uint32_t)(base == 0 ? i : i - base - 1);
                if (vt->encode(st, num, &CTX_0, eh)) { free(suf); return -1; }
                for (size_t sym = base; sym < i; sym++) {
                    Ctx ctx = {(uint8_t)sym, 0, block_size};
                    if (vt->encode(st, (uint32_t)init[sym], &ctx, eh)) { free(suf); return -1; }
                }
                cur_active = 0;
            } else {
                do { i += 1; } while (i < 0xFF && init[i] == block_size);
                uint32_t num = (uint32_t)(i - base - 1);
                if (vt->encode(st, num, &CTX_0, eh)) { free(suf); return -1; }
                cur_active = 1;
            }
        }
    }
    { /* encode distances: block/dc.rs:82-85 driving EncodeIterator::next */
        size_t pos[256];
        memcpy(pos, init, sizeof pos);
        size_t last_active = 0;
        for (size_t i = 0; i < block_size; i++) {
            if (suf[i] == (uint32_t)block_size) continue;
            uint8_t s = output[i];
            size_t r = last_active - pos[s];
            last_active = i + 1;
            pos[s] = i + 1 + suf[i];
            Ctx ctx = {s, (uint8_t)r, block_size - i};
            if (vt->encode(st, suf[i], &ctx, eh)) { free(suf); return -1; }
        }
    }
    free(suf);
    if (vt->encode(st, origin, &CTX_0, eh)) return -1; /* block/dc.rs:88 */
    rc = enc_finish(eh);                               /* block/dc.rs:90 */
    g_stage[3] = now_s() - t1;
    return rc;
}

int orc_block_dc_encode_bwt(int model_id, const uint8_t *bwt, size_t n, uint32_t origin,
                            uint8_t *out, size_t cap, size_t *out_len) {
    if (n == 0 || n >= 0xFFFFFFFFu) return -1;
    Enc eh;
    int rc;
    if (model_id == ORC_MODEL_RAWDC) {
        uint8_t tail[16];
        enc_new(&eh, tail, sizeof tail);
        RawDc r = {out, cap, 0, 0};
        rc = block_dc_encode_from_bwt(&RAWDC_VT, &r, bwt, n, origin, &eh);
        *out_len = r.len;
        return rc ? rc : r.err;
    }
    ModelBox mb;
    rc = model_make(model_id, &mb);
    if (rc) return rc;
    mb.vt->reset(mb.st); /* block/dc.rs:31 */
    enc_new(&eh, out, cap);
    rc = block_dc_encode_from_bwt(mb.vt, mb.st, bwt, n, origin, &eh);
    free(mb.st);
    *out_len = eh.len;
    return rc ? rc : eh.err;
}

/* block/dc.rs:41-91 */
int orc_block_dc_encode(int model_id, const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len) {
    if (n == 0 || n >= 0xFFFFFFFFu) return -1;
    memset(g_stage, 0, sizeof g_stage);
    double t0 = now_s();
    uint32_t *sa = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint8_t *bwt = (uint8_t *)malloc(n);
    if (!sa || !bwt) { free(sa); free(bwt); return -9; }
    int rc = orc_sa_sais(in, n, sa);
    double t1 = now_s();
    uint32_t origin = 0;
    if (!rc) rc = orc_bwt_forward(in, n, sa, bwt, &origin);
    double t2 = now_s();
    free(sa);
    if (!rc) rc = orc_block_dc_encode_bwt(model_id, bwt, n, origin, out, cap, out_len);
    free(bwt);
    g_stage[0] = t1 - t0;
    g_stage[1] = t2 - t1;
    return rc;
}

typedef struct { const ModelVT *vt; void *st; Dec *dh; } DecSrc;
static int model_dist(void *user, const Ctx *ctx, size_t *out) { /* closure at block/dc.rs:146-150 */
    DecSrc *s = (DecSrc *)user;
    uint32_t d;
    if (s->vt->decode(s->st, ctx, s->dh, &d)) return -1;
    *out = d;
    return s->dh->err;
}

/* block/dc.rs:119-160 */
int orc_block_dc_decode(int model_id, const uint8_t *in, size_t in_len, size_t n, uint8_t *out) {
    if (n == 0 || n >= 0xFFFFFFFFu) return -1;
    ModelBox mb;
    int rc = model_make(model_id, &mb);
    if (rc) return rc;
    mb.vt->reset(mb.st); /* block/dc.rs:108 */
    Dec dh;
    dec_new(&dh, in, in_len);
    size_t init[256];
    for (int s = 0; s < 256; s++) init[s] = n;
    { /* block/dc.rs:123-144 */
        int cur_active = 1;
        size_t i = 0;
        while (i < 0xFF) {
            size_t add = (i == 0 && cur_active) ? 0 : 1;
            uint32_t v;
            if (mb.vt->decode(mb.st, &CTX_0, &dh, &v) || dh.err) { free(mb.st); return -1; }
            size_t num = (size_t)v + add;
            if (cur_active) {
                for (size_t sym = i; sym < i + num && sym < 0x100; sym++) { /* .skip(i).take(num) over 256 entries */
                    Ctx ctx = {(uint8_t)sym, 0, n};
                    if (mb.vt->decode(mb.st, &ctx, &dh, &v) || dh.err) { free(mb.st); return -1; }
                    init[sym] = v;
                }
                cur_active = 0;
            } else {
                cur_active = 1;
            }
            i += num;
        }
    }
    uint8_t *bwt = (uint8_t *)malloc(n);
    if (!bwt) { free(mb.st); return -9; }
    MTF mtf;
    memset(&mtf, 0, sizeof mtf);
    DecSrc src = {mb.vt, mb.st, &dh};
    rc = dc_decode(init, bwt, n, &mtf, model_dist, &src);
    uint32_t origin = 0;
    if (!rc) rc = (mb.vt->decode(mb.st, &CTX_0, &dh, &origin) || dh.err) ? -1 : 0; /* block/dc.rs:151 */
    if (!rc) {                                                                       /* block/dc.rs:154-156 */
        rc = orc_bwt_inverse(bwt, n, origin, out);
        /* Reference quirk: for a one-symbol block dc::decode returns before reading the single sweep
         * distance the encoder wrote, so `origin` above is that distance (0), and bwt::decode then yields
         * 1 byte instead of n.  Reported as +1 ("reference output would be truncated"); `out` holds the
         * first byte only. */
        if (rc && n > 1) {
            int single = 1;
            for (size_t i = 1; i < n && single; i++) single = bwt[i] == bwt[0];
            if (single) { out[0] = bwt[0]; rc = 1; }
        }
    }
    if (!rc) { dec_feed(&dh); if (dh.err) rc = -4; }                                 /* dh.finish() block/dc.rs:158 */
    free(bwt);
    free(mb.st);
    return rc;
}

/* model/mod.rs:59-76 roundtrip_dc: reset, encode the stream, finish */
int orc_model_encode(int model_id, const uint32_t *d, const uint8_t *sym, size_t m,
                     uint8_t *out, size_t cap, size_t *out_len) {
    ModelBox mb;
    int rc = model_make(model_id, &mb);
    if (rc) return rc;
    mb.vt->reset(mb.st);
    Enc eh;
    enc_new(&eh, out, cap);
    for (size_t k = 0; k < m && !rc; k++) {
        Ctx ctx = {sym[k], 0, 0};
        rc = mb.vt->encode(mb.st, d[k], &ctx, &eh);
    }
    if (!rc) rc = enc_finish(&eh);
    *out_len = eh.len;
    free(mb.st);
    return rc;
}
int orc_model_decode(int model_id, const uint8_t *in, size_t in_len, const uint8_t *sym, size_t m, uint32_t *d) {
    ModelBox mb;
    int rc = model_make(model_id, &mb);
    if (rc) return rc;
    mb.vt->reset(mb.st);
    Dec dh;
    dec_new(&dh, in, in_len);
    for (size_t k = 0; k < m && !rc; k++) {
        Ctx ctx = {sym[k], 0, 0};
        rc = mb.vt->decode(mb.st, &ctx, &dh, &d[k]);
        if (!rc && dh.err) rc = dh.err;
    }
    free(mb.st);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * entropy::ari::Range + entropy::{Encoder,Decoder} (src/entropy/ari.rs, src/entropy/mod.rs) -- in-repo
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint32_t lo, hi; } BitRange;
static uint32_t bitrange_mid(const BitRange *r, uint16_t model_flat) { /* entropy/ari.rs:21-30 */
    uint32_t flat = (uint32_t)model_flat + 1 - ((uint32_t)model_flat >> (FLAT_BITS - 1));
    uint32_t diff = r->hi - r->lo;
    uint32_t flat_mask = FLAT_TOTAL - 1;
    return r->lo + (diff >> FLAT_BITS) * flat + (((diff & flat_mask) * flat) >> FLAT_BITS);
}
static size_t bitrange_roll(BitRange *r, uint8_t *out) { /* entropy/ari.rs:32-42 */
    size_t count = 0;
    while (((r->lo ^ r->hi) & 0xFF000000u) == 0) {
        out[count++] = (uint8_t)(r->lo >> 24);
        r->lo <<= 8;
        r->hi = (r->hi << 8) | 0xFF;
    }
    return count;
}
int orc_bitcoder_encode(const uint8_t *bits, const uint16_t *flat, size_t nbits, uint8_t *out, size_t cap, size_t *out_len) {
    BitRange r = {0, 0xFFFFFFFFu}; /* entropy/ari.rs:14-19 */
    size_t len = 0;
    for (size_t k = 0; k < nbits; k++) { /* entropy/ari.rs:44-53 + entropy/mod.rs:28-32 */
        uint32_t mid = bitrange_mid(&r, flat[k]);
        if (bits[k] == 0) r.hi = mid; else r.lo = mid + 1;
        uint8_t buf[8];
        size_t num = bitrange_roll(&r, buf);
        if (len + num > cap) return -2;
        memcpy(out + len, buf, num);
        len += num;
    }
    if (len + 4 > cap) return -2;
    for (int i = 0; i < 4; i++) out[len++] = (uint8_t)(r.lo >> (24 - i * 8)); /* post_encode entropy/ari.rs:67-73 */
    *out_len = len;
    return 0;
}
int orc_bitcoder_decode(const uint8_t *in, size_t in_len, const uint16_t *flat, size_t nbits, uint8_t *bits) {
    BitRange r = {0, 0xFFFFFFFFu};
    uint32_t code = 0;
    size_t pos = 0, pending = 4; /* entropy/mod.rs:52-68 */
    for (size_t k = 0; k < nbits; k++) {
        while (pending) { if (pos >= in_len) return -4; code = (code << 8) + in[pos++]; pending--; }
        uint32_t mid = bitrange_mid(&r, flat[k]); /* entropy/ari.rs:55-65 */
        uint8_t tmp[8];
        if (code <= mid) { r.hi = mid; bits[k] = 0; } else { r.lo = mid + 1; bits[k] = 1; }
        pending = bitrange_roll(&r, tmp);
    }
    return 0;
}

/* ================================================================================================
 * `bbb` coding model (src/model/bbb.rs) over block::raw (src/block/raw.rs:35-104)
 *
 * PARITY UNPINNED, and more weakly anchored than anything above.  bbb.rs itself is in the reference and is
 * followed line by line (state map, contexts, the five gate stages, their mixing, the update order).  But every
 * gate is compress::entropy::ari::apm::Gate and every probability an apm::Bit, whose arithmetic lives in the
 * un-vendored crate.  They are restated after the model's in-repo ANALOGUE, Mahoney's bbb
 * (etc/bbb/main.cpp: squash :348-359, Stretch :361-382, class APM :425-460), under these assumptions:
 *   G1  Bit = 12-bit probability of a ZERO bit (as for the exp model above; src/entropy/ari.rs:22-30 relies on it);
 *       a true bit is coded in [flat, 4096), a false one in [0, flat)
 *   G2  Bit::to_wide / from_wide = Mahoney's stretch / squash (d scaled by 8 bits, 33-point table)
 *   G3  Gate = 33 bins of 16-bit probabilities, initialised to squash((j-16)*128)*16, pass = linear interpolation
 *       between the two bins around stretch(p), BinCoords = (bin, weight)
 *   G4  Gate::update(bit, coords, rate, 0) moves BOTH bins towards the observed bit with shift rate+4: bbb.rs passes
 *       1,5,3,4,3,3 where main.cpp:502-507 passes 5,9,7,8,7,7 to the same six stages
 *   G5  Bit::predict() is "flat >= 2048" (bbb.rs:268-274 then mirrors main.cpp:542 `p+=p<2048`)
 * None of G2-G5 can be checked in this image; a stream made here round-trips here and nowhere else is claimed.
 * ============================================================================================== */
/* the oracle's OWN copy of the state table, generated from the reference's literal by oracle/make_bbb_table.py (the product keeps a
 * separate one, dark_amd/csrc/bbb_states.inc; tests/test_oracle.py compares both with src/model/bbb.rs:34-99) */
static const uint8_t BBB_STATE_ROWS[256][4] = {
#include "bbb_state_table.inc"
};
#define BBB_STATE(state, col) BBB_STATE_ROWS[(state)][(col)]
const uint8_t *orc_bbb_state_table(void) { return &BBB_STATE_ROWS[0][0]; }

static int bbb_squash(int d) { /* etc/bbb/main.cpp:348-359 */
    static const int t[33] = {1, 2, 3, 6, 10, 16, 27, 45, 73, 120, 194, 310, 488, 747, 1101, 1546, 2047, 2549, 2994, 3348, 3607, 3785, 3901,
                              3975, 4022, 4050, 4068, 4079, 4085, 4089, 4092, 4093, 4094};
    if (d > 2047) return 4095;
    if (d < -2047) return 0;
    int w = d & 127;
    d = (d >> 7) + 16;
    return (t[d] * (128 - w) + t[d + 1] * w + 64) >> 7;
}
static short g_bbb_stretch[4096];
static int g_bbb_stretch_ready;
static void bbb_stretch_init(void) { /* etc/bbb/main.cpp:373-382 */
    if (g_bbb_stretch_ready) return;
    int pi = 0;
    for (int x = -2047; x <= 2047; ++x) {
        int i = bbb_squash(x);
        for (int j = pi; j <= i; ++j) g_bbb_stretch[j] = (short)x;
        pi = i + 1;
    }
    g_bbb_stretch[4095] = 2047;
    g_bbb_stretch_ready = 1;
}

typedef struct { uint16_t t[33]; } BbbGate;            /* [compress] apm::Gate (G3) */
typedef struct { int bin, weight; } BbbCoords;          /* [compress] apm::BinCoords */
static void bbbgate_new(BbbGate *g) { for (int j = 0; j < 33; ++j) g->t[j] = (uint16_t)(bbb_squash((j - 16) * 128) * 16); }
static uint32_t bbbgate_pass(const BbbGate *g, uint32_t flat, BbbCoords *bc) { /* main.cpp:442-450 */
    int s = g_bbb_stretch[flat];
    bc->weight = s & 127;
    bc->bin = (s + 2048) >> 7;
    return (uint32_t)((g->t[bc->bin] * (128 - bc->weight) + g->t[bc->bin + 1] * bc->weight) >> 11);
}
static void bbbgate_update(BbbGate *g, int bit, BbbCoords bc, int rate) { /* main.cpp:444-446, G4 */
    int r = rate + 4, z = bit ? 0 : 1;
    int target = (z << 16) + (z << r) - z - z;
    g->t[bc.bin] = (uint16_t)(g->t[bc.bin] + ((target - (int)g->t[bc.bin]) >> r));
    g->t[bc.bin + 1] = (uint16_t)(g->t[bc.bin + 1] + ((target - (int)g->t[bc.bin + 1]) >> r));
}

typedef struct { /* bbb.rs:150-173 */
    size_t ctx_id;
    uint8_t ctx2state[256];
    uint16_t sm_table[256];
    uint8_t bit_context;
    uint32_t last_bytes;
    uint16_t run_count, run_context;
    BbbGate gate1a[0x100], gate1b[0x100], gate2[0x10000], gate3[0x400], gate4[0x2000], gate5[0x4000];
} BbbModel;
typedef struct { BbbCoords b11, b12, b2, b3, b4, b5; size_t c1, c2, c3, c4, c5; } BbbCookie; /* bbb.rs:175-187 */

static void bbb_reset(BbbModel *m) { /* bbb.rs:191-205 via reset() :284-286, StateMap::new :108-126 */
    bbb_stretch_init();
    m->ctx_id = 0;
    memset(m->ctx2state, 0, sizeof m->ctx2state);
    for (int i = 0; i < 256; ++i) {
        size_t n0 = BBB_STATE(i, 2), n1 = BBB_STATE(i, 3);
        if (n0 == 0) n1 <<= 7;
        if (n1 == 0) n0 <<= 7;
        m->sm_table[i] = (uint16_t)(((n0 + 1) << 16) / (n0 + n1 + 2));
    }
    m->bit_context = 1;
    m->last_bytes = 0;
    m->run_count = 0;
    m->run_context = 0;
    BbbGate proto;
    bbbgate_new(&proto);
    for (size_t i = 0; i < 0x100; ++i) { m->gate1a[i] = proto; m->gate1b[i] = proto; }
    for (size_t i = 0; i < 0x10000; ++i) m->gate2[i] = proto;
    for (size_t i = 0; i < 0x400; ++i) m->gate3[i] = proto;
    for (size_t i = 0; i < 0x2000; ++i) m->gate4[i] = proto;
    for (size_t i = 0; i < 0x4000; ++i) m->gate5[i] = proto;
}
static uint32_t bbb_predict(const BbbModel *m, BbbCookie *ck) { /* bbb.rs:232-280 */
    uint32_t p0 = (uint32_t)(m->sm_table[m->ctx2state[m->ctx_id]] >> (16 - FLAT_BITS)); /* StateMap::predict :139-143 */
    size_t bit_context = m->bit_context, last_bytes = m->last_bytes;
    ck->c1 = bit_context;
    uint32_t p11 = bbbgate_pass(&m->gate1a[ck->c1], p0, &ck->b11);
    uint32_t p12 = bbbgate_pass(&m->gate1b[ck->c1], p0, &ck->b12);
    uint32_t p1 = (p11 + p12 + 1) >> 1;
    ck->c2 = bit_context | ((last_bytes & 0xFF) << 8);
    uint32_t p2 = bbbgate_pass(&m->gate2[ck->c2], p1, &ck->b2);
    ck->c3 = (last_bytes & 0xFF) | (size_t)m->run_context;
    uint32_t p3 = bbbgate_pass(&m->gate3[ck->c3], p2, &ck->b3);
    ck->c4 = bit_context | (last_bytes & 0x1F00);
    uint32_t p4x = bbbgate_pass(&m->gate4[ck->c4], p3, &ck->b4);
    uint32_t p4 = (p4x * 3 + p3 + 2) >> 2;
    size_t c5y = bit_context ^ (last_bytes & 0xFFFFFF);
    ck->c5 = ((c5y * 123456791u) & 0xFFFFFFFFu) >> 18;
    uint32_t p5x = bbbgate_pass(&m->gate5[ck->c5], p4, &ck->b5);
    uint32_t p5 = (p5x + p4 + 1) >> 1;
    return p5 >= (FLAT_TOTAL >> 1) ? p5 : p5 + 1; /* :268-274, G5 */
}
static void bbb_update(BbbModel *m, int bit, int reset, const BbbCookie *ck) { /* bbb.rs:197-230 */
    m->bit_context = (uint8_t)(((m->bit_context & 0x7F) << 1) | bit);
    if (reset) {
        m->last_bytes = ((m->last_bytes & 0xFFFFFF) << 8) | m->bit_context;
        m->bit_context = 1;
        if (((m->last_bytes ^ (m->last_bytes >> 8)) & 0xFF) == 0) {
            if (m->run_count < 0xFFFF) m->run_count += 1;
            if (m->run_count == 1 || m->run_count == 2 || m->run_count == 4) m->run_context += 0x100;
        } else {
            m->run_count = 0;
            m->run_context = 0;
        }
    }
    { /* StateMap::update :128-137 */
        size_t state = m->ctx2state[m->ctx_id];
        m->ctx2state[m->ctx_id] = BBB_STATE(state, bit);
        size_t top = (size_t)(1 - bit) << 16, old = m->sm_table[state];
        m->sm_table[state] = (uint16_t)((0xFF * old + top + 0x80) >> 8);
    }
    m->ctx_id = m->bit_context;
    bbbgate_update(&m->gate1a[ck->c1], bit, ck->b11, 1);
    bbbgate_update(&m->gate1b[ck->c1], bit, ck->b12, 5);
    bbbgate_update(&m->gate2[ck->c2], bit, ck->b2, 3);
    bbbgate_update(&m->gate3[ck->c3], bit, ck->b3, 4);
    bbbgate_update(&m->gate4[ck->c4], bit, ck->b4, 3);
    bbbgate_update(&m->gate5[ck->c5], bit, ck->b5, 3);
}
static int bbb_encode_sym(BbbModel *m, uint8_t sym, Enc *e) { /* bbb.rs:288-297 */
    for (int i = 7; i >= 0; --i) {
        int bit = (sym >> i) & 1;
        BbbCookie ck;
        uint32_t flat = bbb_predict(m, &ck);
        if (binraw_encode(e, flat, FLAT_TOTAL, bit)) return -1;
        bbb_update(m, bit, i == 0, &ck);
    }
    return 0;
}
static int bbb_decode_sym(BbbModel *m, Dec *d, uint8_t *sym) { /* bbb.rs:299-310 */
    uint8_t s = 0;
    for (int i = 7; i >= 0; --i) {
        BbbCookie ck;
        uint32_t flat = bbb_predict(m, &ck);
        int bit;
        if (binraw_decode(d, flat, FLAT_TOTAL, &bit)) return -1;
        s |= (uint8_t)(bit << i);
        bbb_update(m, bit, i == 0, &ck);
    }
    *sym = s;
    return 0;
}

/* block::raw::Encoder::encode (src/block/raw.rs:35-59) with the bbb model, from a given BWT + origin */
int orc_raw_bbb_encode_bwt(const uint8_t *bwt, size_t n, uint32_t origin, uint8_t *out, size_t cap, size_t *out_len) {
    BbbModel *m = (BbbModel *)malloc(sizeof *m);
    if (!m) return -2;
    bbb_reset(m);
    Enc eh;
    enc_new(&eh, out, cap);
    int rc = 0;
    rc = rc || bbb_encode_sym(m, (uint8_t)(origin >> 24), &eh);
    rc = rc || bbb_encode_sym(m, (uint8_t)(origin >> 16), &eh);
    rc = rc || bbb_encode_sym(m, (uint8_t)(origin >> 8), &eh);
    rc = rc || bbb_encode_sym(m, (uint8_t)origin, &eh);
    for (size_t i = 0; i < n && !rc; ++i) rc = bbb_encode_sym(m, bwt[i], &eh);
    if (!rc) rc = enc_finish(&eh);
    free(m);
    if (rc || eh.err) return eh.err ? eh.err : -1;
    *out_len = eh.len;
    return 0;
}
/* block::raw::Decoder::decode (src/block/raw.rs:84-104) up to the BWT + origin */
int orc_raw_bbb_decode_bwt(const uint8_t *in, size_t in_len, size_t n, uint8_t *bwt, uint32_t *origin) {
    BbbModel *m = (BbbModel *)malloc(sizeof *m);
    if (!m) return -2;
    bbb_reset(m);
    Dec dh;
    dec_new(&dh, in, in_len);
    uint8_t b[4];
    int rc = 0;
    for (int k = 0; k < 4 && !rc; ++k) rc = bbb_decode_sym(m, &dh, &b[k]);
    *origin = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
    for (size_t i = 0; i < n && !rc; ++i) rc = bbb_decode_sym(m, &dh, &bwt[i]);
    free(m);
    return (rc || dh.err) ? -5 : 0;
}
