/* dark_amd.h -- C ABI of the MI355X-native BWT compression path that drops in for kvark/dark's
 * src/saca.rs + src/block + src/model + src/entropy.
 *
 * Every entry point names the reference interface it replaces (paths relative to the reference root).
 * Conventions (mirroring the reference's `&mut self` objects):
 *   - a dk_ctx is owned by one host thread and one GPU; it is NOT thread-safe; distinct contexts may run
 *     concurrently on distinct threads / GPUs;
 *   - all functions return DK_OK (0) or a negative DK_E_* code and never unwind or abort across the boundary;
 *     dk_last_error() gives the text of the last failure on that context;
 *   - "host" entry points take caller-owned host pointers valid for the call; "dk_dev_" entry points take
 *     caller-owned DEVICE pointers (hipMalloc'ed on the context's GPU, e.g. torch tensor .data_ptr());
 *   - stream contract of the dk_dev_ entry points: the library works on its own non-blocking HIP stream and synchronises it
 *     before returning, so outputs are complete on return.  It does NOT order its work after the caller's streams: device
 *     INPUTS must be complete (the producing stream synchronised) before the call.  dark_amd/context.py does that for
 *     torch tensors (torch.cuda.current_stream().synchronize());
 *   - n == 0 is an error (the reference panics at src/saca.rs:107); n must be <= dk_capacity();
 *   - there is no CPU fallback: without a GPU dk_ctx_create fails.  Suffix sorting, BWT, DC distances and
 *     the inverse BWT run on the GPU; the adaptive range coder and dc::decode are serial by construction
 *     (every symbol updates the model the next one reads) and run on the host inside this library.
 */
#ifndef DARK_AMD_H
#define DARK_AMD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DK_OK 0
#define DK_E_ARG (-1)        /* null pointer, n == 0, n > capacity, origin out of range ... */
#define DK_E_NOMEM (-2)      /* host or device allocation failed */
#define DK_E_HIP (-3)        /* HIP runtime error (text in dk_last_error) */
#define DK_E_CAPACITY (-4)   /* caller's output buffer is too small */
#define DK_E_MODEL (-5)      /* unknown model id, or the model cannot represent a value at this block size */
#define DK_E_STREAM (-6)     /* corrupt / truncated stream, or a stream the reference format cannot decode */
#define DK_E_INTERNAL (-7)   /* invariant violated (bug) */
#define DK_E_NODEVICE (-8)   /* no usable GPU: this library has no CPU backend */

/* model ids = the -m names of src/main.rs:73-82 */
#define DK_MODEL_DARK 0   /* src/model/dark.rs   (valid for every block size < 2^31) */
#define DK_MODEL_EXP 1    /* src/model/exp.rs    (CLI default; distances truncated to 24 bits: n <= 2^24) */
#define DK_MODEL_YBS 2    /* src/model/ybs.rs    (distances < 2^29) */
#define DK_MODEL_SIMPLE 3 /* src/model/simple.rs (distances < 2^24 + 255) */
#define DK_MODEL_RAWDC 4  /* src/model/raw.rs:12-44 DcOut: 10-byte records instead of a coded stream */

typedef struct dk_ctx dk_ctx;

/* saca::Constructor::new(max_n) src/saca.rs:351-360 + block::dc::{Encoder,Decoder}::new src/block/dc.rs:30-37,106-115.
 * Allocates the device workspace for blocks of up to max_n bytes on GPU `hip_device` (>= 0). */
int dk_ctx_create(int hip_device, size_t max_n, dk_ctx **out);
void dk_ctx_destroy(dk_ctx *ctx);
/* saca::Constructor::capacity src/saca.rs:363-365 */
size_t dk_capacity(const dk_ctx *ctx);
const char *dk_last_error(const dk_ctx *ctx);
/* library / build identification, e.g. "dark_amd 0.1 gfx950" */
const char *dk_version(void);

/* ---- host-pointer entry points ------------------------------------------------------------------------- */
/* saca::Constructor::compute src/saca.rs:368-378: sa_out[0..n) = suffix array of in[0..n)
 * (no sentinel; a suffix that is a prefix of another sorts first). */
int dk_suffix_array(dk_ctx *ctx, const uint8_t *in, size_t n, uint32_t *sa_out);
/* compress::bwt::TransformIterator as driven by src/block/dc.rs:45-50: bwt_out[i] = in[SA[i]-1] (in[n-1] where
 * SA[i]==0), *origin = the i with SA[i]==0.  Known answers src/saca.rs:411-412. */
int dk_bwt_forward(dk_ctx *ctx, const uint8_t *in, size_t n, uint8_t *bwt_out, uint32_t *origin);
/* compress::bwt::decode as driven by src/block/dc.rs:154-156 */
int dk_bwt_inverse(dk_ctx *ctx, const uint8_t *bwt, size_t n, uint32_t origin, uint8_t *out);
/* compress::bwt::dc::encode + EncodeIterator as driven by src/block/dc.rs:52,82-85, compacted: one entry per run
 * of the BWT, in position order.  init[s] = first position of s, or n if absent.  dist/sym/rank hold up to n
 * entries (rank = Context.last_rank, may be NULL); *m = number of entries. */
int dk_dc_encode(dk_ctx *ctx, const uint8_t *bwt, size_t n, uint32_t init[256],
                 uint32_t *dist, uint8_t *sym, uint8_t *rank, size_t *m);
/* compress::bwt::dc::decode as driven by src/block/dc.rs:146-150 (host, serial).  *consumed (may be NULL) = number of
 * distances read. */
int dk_dc_decode(dk_ctx *ctx, const uint32_t init[256], const uint32_t *dist, size_t m,
                 uint8_t *bwt_out, size_t n, size_t *consumed);
/* block::Encoder::encode for block::dc::Encoder<M> src/block/dc.rs:41-91.  Output = the coded stream WITHOUT the
 * u32 n header that src/main.rs:102 writes in front of it. */
int dk_block_encode(dk_ctx *ctx, int model_id, const uint8_t *in, size_t n,
                    uint8_t *out, size_t out_cap, size_t *out_len);
/* block::Decoder::decode for block::dc::Decoder<M> src/block/dc.rs:119-160 */
int dk_block_decode(dk_ctx *ctx, int model_id, const uint8_t *in, size_t in_len, size_t n, uint8_t *out);
/* block::raw::{Encoder,Decoder}<M: RawModel> src/block/raw.rs:17-105: SA -> BWT, then origin as four symbols (bits 31..24 first,
 * src/block/raw.rs:48-51) and every BWT byte through the RawModel.  RawModels of the reference:
 *   DK_RAWMODEL_OUT  model::raw::Out src/model/raw.rs:46-76: every symbol is appended to ./out.raw, nothing reaches the coder.  Here the
 *                    symbols come back in `dump` (n + 4 bytes) and `out` receives what the idle coder's finish() writes (4 zero bytes).
 *                    Decoding "is not supported" in the reference (raw.rs:71-75 returns symbol 0 for everything): n zero bytes, DK_OK.
 *   DK_RAWMODEL_BBB  model::bbb::Model src/model/bbb.rs: the BWT bytes are coded bit by bit into `out` (dump is not used, may be NULL).
 *                    The model's structure follows bbb.rs; its gates (compress::entropy::ari::apm::Gate, not in the reference tree)
 *                    are restated after the in-repo analogue etc/bbb/main.cpp:348-460 -- PARITY UNPINNED: a stream made here decodes
 *                    here; compatibility with the Rust crate's bytes is not claimed. */
#define DK_RAWMODEL_OUT 0
#define DK_RAWMODEL_BBB 1
int dk_raw_block_encode(dk_ctx *ctx, int raw_model, const uint8_t *in, size_t n, uint8_t *out, size_t out_cap, size_t *out_len,
                        uint8_t *dump, size_t dump_cap, size_t *dump_len);
int dk_raw_block_decode(dk_ctx *ctx, int raw_model, const uint8_t *in, size_t in_len, size_t n, uint8_t *out);
/* Bytes of `in` the last dk_block_decode / dk_dev_block_decode on this context consumed (= the length the encoder wrote:
 * 4 priming bytes + one per renormalisation shift).  Lets a caller walk concatenated [u32 n][stream] records -- the
 * multi-block extension of the single-block file of src/main.rs:70,102. */
size_t dk_last_consumed(const dk_ctx *ctx);
/* Properties of the block the last dk_block_encode / dk_dev_block_encode on this context coded (0 after any other call):
 *   DK_FLAG_HAS_FF         the block contains byte 0xFF.  The stream is bit-exact with the reference's, and like the reference's it
 *                          cannot be decoded: the init-table header never transmits symbol 0xFF (src/block/dc.rs:57,60,73,127).
 *                          A front end should refuse or warn (dark_amd/cli.py does) instead of writing an archive that is lost.
 *   DK_FLAG_SINGLE_SYMBOL  one distinct symbol: the reference's decoder mis-reads `origin` for such a block;
 *                          this library's decoder returns all n bytes. */
#define DK_FLAG_HAS_FF 1u
#define DK_FLAG_SINGLE_SYMBOL 2u
unsigned dk_last_block_flags(const dk_ctx *ctx);

/* ---- device-resident entry points (inputs already in HBM; used by pipelines and by bench.py) ----------------- */
int dk_dev_suffix_array(dk_ctx *ctx, const uint8_t *d_in, size_t n, uint32_t *d_sa_out);
int dk_dev_bwt_forward(dk_ctx *ctx, const uint8_t *d_in, size_t n, uint8_t *d_bwt_out, uint32_t *origin);
int dk_dev_bwt_inverse(dk_ctx *ctx, const uint8_t *d_bwt, size_t n, uint32_t origin, uint8_t *d_out);
/* d_dist/d_sym/d_rank: device arrays of n entries (d_rank may be NULL); init and m are returned on the host */
int dk_dev_dc_encode(dk_ctx *ctx, const uint8_t *d_bwt, size_t n, uint32_t init[256],
                     uint32_t *d_dist, uint8_t *d_sym, uint8_t *d_rank, size_t *m);
/* whole forward path from a device-resident block to the coded stream in host memory */
int dk_dev_block_encode(dk_ctx *ctx, int model_id, const uint8_t *d_in, size_t n,
                        uint8_t *out, size_t out_cap, size_t *out_len);
/* `count` independent blocks on one GPU, pipelined: the device stages of block i+1 run while up to `host_threads` host threads
 * code the distance streams of earlier blocks (the reference treats every block as self-contained: fresh model and coder per
 * Encoder::new, src/block/dc.rs:30-37,53).  out[i] / out_len[i] are exactly what dk_dev_block_encode gives for block i. */
int dk_dev_batch_encode(dk_ctx *ctx, int model_id, size_t count, const uint8_t *const *d_in, const size_t *n,
                        uint8_t *const *out, const size_t *out_cap, size_t *out_len, int host_threads);
/* The same pipeline fed one block at a time (blocks that arrive from a file): begin starts `host_threads` coding threads;
 * push runs the device stages of one more block on the calling thread -- on return d_in may be reused -- and queues its coding
 * (it waits while every staging slot is busy); *out_len is written when that block's coding ends; finish waits for all coders,
 * frees the batch and returns the first failure.  Between begin and finish the context serves this batch only: every other entry point
 * returns DK_E_ARG, a second begin included.  A failed push leaves the batch open (earlier blocks are still being coded into the
 * caller's `out` / `out_len`): the caller must still call finish before it frees those buffers.  dk_ctx_destroy finishes a batch that
 * was left open, so the coding threads never outlive the staging memory they read -- and with that the dk_batch handle is GONE: after
 * dk_ctx_destroy it must not be passed to dk_batch_finish (or anything else) any more; the result of that implicit finish is dropped. */
typedef struct dk_batch dk_batch;
int dk_batch_begin(dk_ctx *ctx, int model_id, int host_threads, dk_batch **out);
int dk_batch_push(dk_batch *batch, const uint8_t *d_in, size_t n, uint8_t *out, size_t out_cap, size_t *out_len);
int dk_batch_finish(dk_batch *batch);
/* inverse of dk_dev_batch_encode: host threads decode the streams while the GPU inverts the BWTs that are ready */
int dk_dev_batch_decode(dk_ctx *ctx, int model_id, size_t count, const uint8_t *const *in, const size_t *in_len, const size_t *n,
                        uint8_t *const *d_out, int host_threads);
/* whole inverse path from a coded stream in host memory to a device-resident block */
int dk_dev_block_decode(dk_ctx *ctx, int model_id, const uint8_t *in, size_t in_len, size_t n, uint8_t *d_out);

/* ---- several GPUs in one call ------------------------------------------------------------------------------------------------------
 * Blocks are independent (fresh model and coder per block, src/block/dc.rs:30-37,53): block i goes to GPU devices[i mod ndev].  Inside
 * the call every listed device gets one host thread with its own dk_ctx ("one host thread + one context per GPU",
 * no inter-GPU traffic); each runs the pipelined batch path on its blocks with host_threads_per_gpu coding threads.  `devices` may name
 * a GPU twice (two contexts on one GPU).  in / out are host pointers; out[i] / out_len[i] are exactly what dk_block_encode gives for
 * block i.  err (may be NULL) receives the text of the first failure. */
int dk_multi_block_encode(const int *devices, int ndev, int model_id, size_t count, const uint8_t *const *in, const size_t *n,
                          uint8_t *const *out, const size_t *out_cap, size_t *out_len, int host_threads_per_gpu, char *err, size_t err_cap);
int dk_multi_block_decode(const int *devices, int ndev, int model_id, size_t count, const uint8_t *const *in, const size_t *in_len,
                          const size_t *n, uint8_t *const *out, int host_threads_per_gpu, char *err, size_t err_cap);

/* ---- model / coder level (host; what src/model/mod.rs:59-76 and src/entropy/ari.rs:76-107 exercise) ----------- */
/* model.reset(); for k: model.encode(dist[k], Context{symbol: sym[k]}, eh); eh.finish()   (src/model/mod.rs:59-66) */
int dk_model_encode(int model_id, const uint32_t *dist, const uint8_t *sym, size_t m,
                    uint8_t *out, size_t out_cap, size_t *out_len);
/* model.reset(); for k: dist[k] = model.decode(Context{symbol: sym[k]}, dh)              (src/model/mod.rs:67-75) */
int dk_model_decode(int model_id, const uint8_t *in, size_t in_len, const uint8_t *sym, size_t m, uint32_t *dist);
/* entropy::Encoder over entropy::ari::Range (src/entropy/mod.rs:11-41, src/entropy/ari.rs:8-74):
 * bits[k] in {0,1}; flat[k] = apm::Bit::to_flat() = 12-bit probability of a zero */
int dk_bitcoder_encode(const uint8_t *bits, const uint16_t *flat, size_t nbits,
                       uint8_t *out, size_t out_cap, size_t *out_len);
int dk_bitcoder_decode(const uint8_t *in, size_t in_len, const uint16_t *flat, size_t nbits, uint8_t *bits);

/* Host entropy stage on its own (what the GPU stages feed): src/block/dc.rs:53-90 from (init, dist[m], sym[m], origin) to the
 * coded stream, and src/block/dc.rs:121-151 back to (BWT, origin).  rank/run_end (Context.last_rank and the position of each
 * entry) are only read by DK_MODEL_RAWDC and may be NULL otherwise.  *single_symbol = 1 flags a one-symbol block, for which the
 * reference mis-reads `origin` (DESIGN.md "Reference quirks"). */
int dk_stream_encode(int model_id, size_t n, const uint32_t init[256], const uint32_t *dist, const uint8_t *sym,
                     const uint8_t *rank, const uint32_t *run_end, size_t m, uint32_t origin,
                     uint8_t *out, size_t out_cap, size_t *out_len);
int dk_stream_decode(int model_id, const uint8_t *in, size_t in_len, size_t n, uint8_t *bwt_out, uint32_t *origin,
                     int *single_symbol, size_t *consumed /* may be NULL: bytes of `in` read = the length the encoder wrote */);

/* Host half of block::raw on its own (what the GPU BWT feeds / is fed by): src/block/raw.rs:45-58 from (L, origin) to the coded stream and
 * src/block/raw.rs:85-97 back.  raw_model must be a coding model (DK_RAWMODEL_BBB). */
int dk_raw_stream_encode(int raw_model, const uint8_t *bwt, size_t n, uint32_t origin, uint8_t *out, size_t out_cap, size_t *out_len);
int dk_raw_stream_decode(int raw_model, const uint8_t *in, size_t in_len, size_t n, uint8_t *bwt_out, uint32_t *origin, size_t *consumed);

/* ---- measurement ------------------------------------------------------------------------------------------ */
#define DK_NUM_KERNEL_SLOTS 32
typedef struct dk_stats {
    /* wall-clock stage times of the last block call on this context, milliseconds */
    double ms_h2d, ms_sa, ms_bwt, ms_dc, ms_d2h, ms_entropy, ms_ibwt, ms_total;
    /* ms_d2h of a large single block (>= 2^21 distances, dk_block_encode / dk_dev_block_encode): the distance stream leaves the GPU in pieces
     * while the host coder already runs, so ms_d2h = the time to enqueue the copies + the time the call waits for the stream after coding
     * (normally microseconds: the transfer hides behind the coder); the coder's own waits at the frontier are part of ms_entropy. */
    uint32_t rounds;          /* prefix-doubling rounds executed by the last suffix sort */
    uint32_t sort_passes;     /* radix passes executed by the last suffix sort */
    uint64_t sorted_elements; /* sum over passes of elements moved */
    uint64_t dc_runs;         /* m of the last dc encode */
    uint32_t entropy_threads; /* host threads the last range-coder pass used: 1, 2 (models | coder), 4 (exponent model, mantissa model, merger, coder) or 5 (the exponent model in two halves) */
    int32_t entropy_l3_group; /* the last-level-cache group that pass claimed for its threads (lowest cpu number in it); -1: none */
    /* per-kernel HIP-event timings accumulated since dk_stats_reset (only while profiling is enabled) */
    uint32_t kernel_launches[DK_NUM_KERNEL_SLOTS];
    double kernel_ms[DK_NUM_KERNEL_SLOTS];
    double kernel_bytes[DK_NUM_KERNEL_SLOTS]; /* algorithmic bytes summed over launches */
    uint32_t sa_route;        /* DK_ROUTE_* bits: which ways through the suffix sort the last call took (tests assert the route they are named after) */
    int16_t entropy_l3_numa;  /* memory node of that group (-1: none claimed / unknown) ... */
    int16_t gpu_numa;         /* ... and of the context's GPU (/sys/bus/pci/devices/<bdf>/numa_node; -1: unknown): the coder looks for its group there first */
    uint64_t ws_peak_bytes;   /* most the context's device workspace has held at once since dk_ctx_create ... */
    uint64_t ws_size_bytes;   /* ... and its size (about 69.4 x the capacity + 64 MiB) */
} dk_stats;
#define DK_ROUTE_SHORT_PREFIX 0x1u      /* the prefix probe shortened the initial sort's key */
#define DK_ROUTE_NARROW_KEYS 0x2u       /* ... and its last pass left 32-bit keys */
#define DK_ROUTE_TEXT_ROUND 0x4u        /* a text-extension round ran */
#define DK_ROUTE_ISA_WINDOWS 0x8u       /* rank array through LDS windows */
#define DK_ROUTE_ISA_MARKED 0x10u       /* ... with the active suffixes' head positions handed over by marked SA entries */
#define DK_ROUTE_ISA_BUCKETS 0x20u      /* reserved (rounds 1-4: rank array by a bucketed store for blocks above 2^27 suffixes; every block goes through the windows now) */
#define DK_ROUTE_GENERAL_ROUND 0x40u    /* a doubling round in its general form ran */
#define DK_ROUTE_BIG_GROUPS 0x80u       /* ... with groups of more than 1024 members through the global sort */
#define DK_ROUTE_INPLACE_ROUNDS 0x100u  /* in-place (plateau) rounds ran */
#define DK_ROUTE_PAIR_CHAINS 0x200u     /* ... after pair chains settled groups of two to four */
#define DK_ROUTE_LFIRST 0x400u          /* BWT callers: only groups with different symbols in front were refined (no suffix array) */
#define DK_ROUTE_LFIRST_BIG_ROUND 0x800u  /* ... with at least one global-sort round of big groups */
#define DK_ROUTE_LFIRST_DEEP 0x1000u    /* ... and groups that went the way of long repeats (common extension measured directly) */
#define DK_ROUTE_LFIRST_GIANT 0x4000u    /* ... and common extensions longer than 64 KiB, measured by the whole grid (copies of whole files) */
#define DK_ROUTE_LFIRST_FALLBACK 0x2000u  /* the L-first path gave up (giant groups / over-long common extensions): suffix-array path from the start */
#define DK_ROUTE_PERIOD_ROUND 0x8000u    /* a period round ran: suffixes inside stretches of one short period (runs, (ab)^n, zero padding) placed by where the stretch ends */
#define DK_ROUTE_PACKED_PAIRS 0x10000u   /* the initial sort moved packed pairs: key, carried code and position in one 64-bit word (at most 32 key bits, small alphabets) */
/* enable (1) / disable (0) HIP-event bracketing of every kernel launch on the context's stream */
int dk_set_profiling(dk_ctx *ctx, int enabled);
int dk_stats_reset(dk_ctx *ctx);
int dk_get_stats(const dk_ctx *ctx, dk_stats *out);
/* name of kernel slot i (NULL past the end) */
const char *dk_kernel_name(int slot);

/* ---- host coding threads (operations; nothing in the reference corresponds: its coder is one thread) ------------------------------------
 * A large single block is coded by a pipeline of 2 or 4 host threads confined to one last-level-cache group; groups are
 * claimed per process, and a process that finds none free codes on one thread.  A launcher of several ranks per node can make that
 * deterministic: ask every rank how many groups it could claim, and set the same form everywhere. */
/* thread form of the host coding pass, process-wide: 0 automatic (default; or DK_ENTROPY_THREADS at load time) | 1 | 2 | 4 | 5 */
int dk_set_entropy_threads(int mode);
/* number of last-level-cache groups in which the calling thread may use at least min_cores cores */
int dk_host_l3_groups(int min_cores);
/* the calling thread's last host coding pass: threads used (1 / 2 / 4 / 5) and the L3 group claimed (-1: none).  Either may be NULL. */
void dk_last_entropy_info(int *threads, int *l3_group);
/* memory node the calling thread's coding passes should claim their L3 group on first (-1: none; the block entry points set it to their
 * context's GPU's node themselves -- for callers of dk_stream_encode that know where their staging memory lives) */
void dk_set_entropy_numa_node(int node);
/* the order in which a coding pass tries the machine's L3 groups, on a topology given by the caller (group_numa[g] = memory node of group g,
 * own = the caller's group, preferred_numa as above): indices into order_out[ngroups], returns how many.  For tests of the policy on
 * machines the test host is not (two sockets, eight GPUs). */
int dk_dbg_l3_claim_order(const int *group_numa, int ngroups, int own, int preferred_numa, int *order_out);

/* ---- stage-level debug entry points used by the parity tests ------------------------------------------------ */
/* dk_stream_encode on a distance stream that is still ARRIVING, the way dk_dev_block_encode feeds its host coder while the D2H copies
 * of a large block run: *ready (read atomically; another thread of the caller moves it) = entries of dist / sym that are there.  The
 * coder waits at that frontier; (size_t)-1 there = "the producer gave up", and a frontier that stands still for stall_ms milliseconds
 * (0 = 20 s) = "the stream hangs": DK_E_HIP in both cases instead of a coder that spins for ever.  host_threads: 0 automatic | 1 | 2 | 4 | 5. */
int dk_dbg_stream_encode_gated(int model_id, size_t n, const uint32_t init[256], const uint32_t *dist, const uint8_t *sym, size_t m,
                               uint32_t origin, uint8_t *out, size_t out_cap, size_t *out_len, const size_t *ready, unsigned stall_ms,
                               int host_threads);
/* stable LSD radix sort of (u64 key, u32 value) pairs on bits [begin_bit, end_bit) -- the workhorse of the suffix sort */
int dk_dbg_sort_pairs(dk_ctx *ctx, uint64_t *keys, uint32_t *vals, size_t count, int begin_bit, int end_bit);
/* the same on device arrays, in place (measurement: tools/local_sort_bench.py) */
int dk_dbg_dev_sort_pairs(dk_ctx *ctx, uint64_t *d_keys, uint32_t *d_vals, size_t count, int begin_bit, int end_bit);
/* every 8192-pair tile of a device array sorted by itself inside one workgroup's LDS (the local pass an MSD-first sort would end with:
 * an experiment of round 4) */
int dk_dbg_dev_local_sort(dk_ctx *ctx, uint64_t *d_keys, uint32_t *d_vals, size_t count, int begin_bit, int end_bit);

#ifdef __cplusplus
}
#endif
#endif /* DARK_AMD_H */
