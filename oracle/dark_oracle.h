/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU oracle: a plain-C restatement of the hot path of kvark/dark (suffix sort -> BWT -> DC/MTF ->
 * adaptive range coder).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this; the product (dark_amd/) never links, imports or calls it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - SA, BWT+origin, inverse BWT: pinned by the reference's own known answers
 *     (/root/reference/src/saca.rs:411-412) and cross-checked against orc_sa_naive.
 *   - entropy::ari::Range bitwise coder: restated from in-repo source (src/entropy/ari.rs).
 *   - bwt::mtf, bwt::dc, entropy::ari::{Encoder,Decoder,table,bin,apm}: these live in the
 *     un-vendored crate `compress` 0.1 (Cargo.toml:18, no lock file).  Restated from its published
 *     algorithm and from the reference's call sites; PARITY UNPINNED (no golden bytes exist in the
 *     reference for these stages).
 */
#ifndef DARK_ORACLE_H
#define DARK_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_MODEL_DARK = 0, ORC_MODEL_EXP = 1, ORC_MODEL_YBS = 2, ORC_MODEL_SIMPLE = 3, ORC_MODEL_RAWDC = 4 };

/* saca.rs:351-354: words of storage Constructor::new(max_n) allocates */
size_t orc_saca_storage_words(size_t n);

/* ground truth: comparison sort of all suffixes, "shorter is smaller" (saca.rs:25-35 sort_direct) */
int orc_sa_naive(const uint8_t *t, size_t n, uint32_t *sa);
/* saca.rs:344-384 Constructor::compute -> SA-IS (saca.rs:270-340) */
int orc_sa_sais(const uint8_t *t, size_t n, uint32_t *sa);

/* compress::bwt::TransformIterator (call site block/dc.rs:47-49; known answers saca.rs:411-412) */
int orc_bwt_forward(const uint8_t *t, size_t n, const uint32_t *sa, uint8_t *bwt, uint32_t *origin);
/* compress::bwt::decode (call site block/dc.rs:154; analogue etc/dark-c/src/archon3.cpp:70-86) */
int orc_bwt_inverse(const uint8_t *bwt, size_t n, uint32_t origin, uint8_t *out);

/* compress::bwt::dc::encode + EncodeIterator (call site block/dc.rs:52,82; analogue ptax.cpp:61-111).
 * dist_sparse: n words, filler == n.  Compacted stream (position order): d, sym, last_rank, limit. */
int orc_dc_encode(const uint8_t *bwt, size_t n, uint32_t *dist_sparse, uint32_t init[256],
                  uint32_t *d, uint8_t *sym, uint8_t *rank, uint32_t *limit, size_t *m);
/* compress::bwt::dc::decode (call site block/dc.rs:146-150; analogue ptax.cpp:113-146), fed from an array */
int orc_dc_decode(const uint32_t init[256], const uint32_t *d, size_t m, uint8_t *bwt, size_t n, size_t *consumed);

/* block::dc::Encoder::encode (block/dc.rs:41-91) -- stream without the n header of main.rs:102 */
int orc_block_dc_encode(int model_id, const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len);
/* block::dc::Decoder::decode (block/dc.rs:119-160) */
int orc_block_dc_decode(int model_id, const uint8_t *in, size_t in_len, size_t n, uint8_t *out);
/* same, but starting from a given BWT+origin (stage-boundary entry used by parity tests) */
int orc_block_dc_encode_bwt(int model_id, const uint8_t *bwt, size_t n, uint32_t origin,
                            uint8_t *out, size_t cap, size_t *out_len);

/* model-level roundtrip helpers (model/mod.rs:59-76 roundtrip_dc): a stream of (dist, ctx.symbol) */
int orc_model_encode(int model_id, const uint32_t *d, const uint8_t *sym, size_t m,
                     uint8_t *out, size_t cap, size_t *out_len);
int orc_model_decode(int model_id, const uint8_t *in, size_t in_len, const uint8_t *sym, size_t m, uint32_t *d);

/* entropy::Encoder/Decoder over entropy::ari::Range (src/entropy/mod.rs, src/entropy/ari.rs):
 * bits[i] in {0,1}, flat[i] = 12-bit probability of zero (apm::Bit::to_flat) */
int orc_bitcoder_encode(const uint8_t *bits, const uint16_t *flat, size_t nbits, uint8_t *out, size_t cap, size_t *out_len);
int orc_bitcoder_decode(const uint8_t *in, size_t in_len, const uint16_t *flat, size_t nbits, uint8_t *bits);

/* block::raw::{Encoder,Decoder} with model::bbb::Model (src/block/raw.rs:35-104, src/model/bbb.rs), entered after / left before the
 * BWT.  The gates of the model are restated after etc/bbb/main.cpp (assumptions G1-G5 in dark_oracle.c): PARITY UNPINNED. */
int orc_raw_bbb_encode_bwt(const uint8_t *bwt, size_t n, uint32_t origin, uint8_t *out, size_t cap, size_t *out_len);
int orc_raw_bbb_decode_bwt(const uint8_t *in, size_t in_len, size_t n, uint8_t *bwt, uint32_t *origin);

/* stage timers of the last orc_block_dc_encode call on this thread (seconds): sa, bwt, dc, entropy */
/* the 256 x 4 bit-history state table the bbb model above was compiled with (oracle/bbb_state_table.inc) */
const uint8_t *orc_bbb_state_table(void);
void orc_last_stage_seconds(double out[4]);

#ifdef __cplusplus
}
#endif
#endif
