/*
 * seqkit_hip.h — C-ABI of the MI355X (gfx950) implementation of seqkit's per-read hot path.
 *
 * The reference (annalam/seqkit v0.8.0, Rust) has no FFI/plugin interface: every command is a
 * `while read_line` loop with the arithmetic inlined.  This header is the seam a
 * host (the reference's Rust `fasta`/`sam` binaries, or this repo's C++ ones) binds to replace the
 * inlined per-read arithmetic.  Each entry point cites the reference lines it replaces; paths are
 * relative to the reference tree.  INTEGRATION.md shows the Rust `extern "C"` block.
 *
 * Conventions
 *  - plain pointers and sizes only; every function returns SK_OK (0) or a negative SK_ERR_* code and
 *    never throws or aborts; sk_last_error() gives the message for the last failure on that ctx.
 *  - one sk_ctx per GPU; a ctx is used by one host thread at a time; a ctx owns one HIP stream.
 *  - read batches are fixed-stride SoA: row r of a byte matrix lives at base + r*stride, only its first
 *    len[r] bytes are data (len == NULL: every row holds exactly `stride` bytes).  Bytes past len[r]
 *    are ignored on input; on output (masked sequence) they are unspecified.
 *  - functions without suffix take HOST pointers and are synchronous (they stage through device
 *    workspace owned by the ctx).  `_dev` functions take DEVICE pointers, enqueue on the ctx stream
 *    and return immediately; sk_sync() waits.  Device byte matrices must be 16-byte aligned, and readable up
 *    to the next 4-byte boundary behind their last row (the kernels read whole dwords: when n * stride is not a
 *    multiple of 4, up to 3 bytes past the matrix are fetched and ignored — any allocation of a whole number of
 *    dwords, every hipMalloc, provides them).  Device OUTPUT columns of a demultiplex call are written with wide
 *    stores: assign must be 16-byte aligned, lowest_diff 4-byte, first_idx / last_idx 8-byte (any hipMalloc'ed base
 *    is; a column sliced at an odd row is refused with SK_ERR_INVALID).
 *  - there is no CPU fallback: without a usable GPU sk_create() fails and nothing else can be called.
 */
#ifndef SEQKIT_HIP_H
#define SEQKIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SK_OK            0
#define SK_ERR_INVALID  (-1)   /* bad argument (NULL, negative size, misaligned device pointer, ...) */
#define SK_ERR_HIP      (-2)   /* HIP runtime failure; sk_last_error() has hipGetErrorString          */
#define SK_ERR_NO_DEVICE (-3)  /* no gfx950-capable device at that index                              */
#define SK_ERR_STATE    (-4)   /* call order (e.g. demux before sk_set_barcodes)                      */
#define SK_ERR_NOMEM    (-5)
#define SK_ERR_COMM     (-6)   /* RCCL: library not loadable, or a collective failed                   */

/* assignment codes written by the demultiplex entry points (src/fasta_demultiplex.rs:168-194) */
#define SK_ASSIGN_NONE      (-1)   /* lowest_diff > max_diff                                   */
#define SK_ASSIGN_AMBIGUOUS (-2)   /* lowest_diff <= max_diff but best != equally_fine sample  */

#define SK_MAX_BARCODE_LEN 255
#define SK_MAX_SAMPLES     32767

typedef struct sk_ctx sk_ctx;

/* ---- lifetime ------------------------------------------------------------------------------ */
int  sk_version(void);                                   /* 0x00MMmmpp */
int  sk_device_count(void);                              /* number of visible HIP devices (>= 0) */
int  sk_create(int device_id, sk_ctx **out);
void sk_destroy(sk_ctx *ctx);
const char *sk_last_error(const sk_ctx *ctx);            /* ctx may be NULL: last sk_create failure */
int  sk_sync(sk_ctx *ctx);                               /* wait for everything enqueued on the ctx stream */
void *sk_stream(sk_ctx *ctx);                            /* the ctx's hipStream_t, for interop */

/* ---- device / pinned memory for hosts that do not link HIP themselves ------------------------ */
int sk_malloc_device(sk_ctx *ctx, size_t bytes, void **out);
int sk_free_device(sk_ctx *ctx, void *p);
/* the two pinned calls may come from any thread while another runs a pass on the ctx: they write nothing to it (a failure is
 * the return code only; sk_last_error is unchanged) and bind the ctx's device to the calling thread */
int sk_malloc_pinned(sk_ctx *ctx, size_t bytes, void **out);
int sk_free_pinned(sk_ctx *ctx, void *p);
int sk_copy_h2d(sk_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);   /* async on ctx stream */
int sk_copy_d2h(sk_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);   /* async on ctx stream */

/* ---- sample sheet -> barcode table -------------------------------------------------------------
 * Replaces the per-read walk over `samples[s].barcode` (src/fasta_demultiplex.rs:157-158, Sample
 * :23-28, sheet rules :58-104 stay on the host).  table = S rows of L bytes, sheet order.  Candidate
 * bytes 'N' and 'U' are wildcards (src/fasta_demultiplex.rs:273).  max_diff is the reference's
 * MAX_BARCODE_DIFFERENCE = 1 (:168).  Resets the counters.                                        */
int sk_set_barcodes(sk_ctx *ctx, const uint8_t *table, int S, int L, int max_diff);

/* ---- D1+D2+D3: barcode_diff + best-match loop + decision ----------------------------------------
 * src/fasta_demultiplex.rs:269-277, :154-166, :168-194.  bc = n observed barcodes of exactly L bytes,
 * row stride bc_stride >= L.  Outputs (each nullable except assign): assign[r] = sample index, or
 * SK_ASSIGN_NONE / SK_ASSIGN_AMBIGUOUS; lowest_diff[r] (saturates at 255; 255 when S == 0);
 * first_idx[r] = best_sample, last_idx[r] = equally_fine_sample.  Adds to the ctx counters.         */
int sk_demux_assign(sk_ctx *ctx, const uint8_t *bc, int bc_stride, int64_t n,
                    int32_t *assign, uint8_t *lowest_diff, int16_t *first_idx, int16_t *last_idx);
int sk_demux_assign_dev(sk_ctx *ctx, const uint8_t *bc, int bc_stride, int64_t n,
                        int32_t *assign, uint8_t *lowest_diff, int16_t *first_idx, int16_t *last_idx,
                        uint64_t *counts /* device u64[S+3], NULL = ctx counters */);

/* Which rows the detail columns (lowest_diff / first_idx / last_idx) are defined for.  The reference looks at them in one
 * place only, the ambiguity warning (src/fasta_demultiplex.rs:184-188), i.e. for reads with lowest_diff <= max_diff; the
 * branch of a read that matched nothing (:190-194) never does.
 *   SK_DETAIL_FULL    (default) every row, as the loop :154-166 leaves them.
 *   SK_DETAIL_MATCHED rows with assign != SK_ASSIGN_NONE; the detail of SK_ASSIGN_NONE rows is unspecified (255 / -1 /
 *                     -1 when the lookup table answers).  A demultiplex-alone call with max_diff <= 1 is then ONE table
 *                     lookup per read — the sheet's rows and their one-substitution neighbours, decided on the host with
 *                     the same loop — instead of S x L compares, for sheets of <= 1021 samples, <= 20 columns, <= 7
 *                     letters (8 to 15 letters — a sheet typed partly in lower case — when its rows, or the two halves
 *                     beside a separator, are at most 8 columns; a row with `N` / `U` where other rows hold a letter is entered once per class of that
 *                     column: a few such columns per row, up to 220 000 keys in all); a dual-index sheet whose table would not fit a
 *                     workgroup's LDS is looked up half by half when that is exact (its half-barcodes at least 3
 *                     apart); other sheets run the matchers and fill every row.  The decision-only form (all three
 *                     pointers NULL) takes the table under either mode, and so does the barcode phase of a fused call
 *                     whose sheet is beyond the tile pass's own matcher (128 samples).
 * Applies to sk_demux_assign(_dev) and sk_fused_pass(_dev) of this ctx from the next call on.                        */
#define SK_DETAIL_FULL    0
#define SK_DETAIL_MATCHED 1
int sk_set_detail_mode(sk_ctx *ctx, int mode);
/* What serves the calls described above for the current sheet (the table is built here if no call has asked for it yet;
 * src/fasta_demultiplex.rs:154-194 is what it answers either way): *kind = SK_TABLE_NONE (the S x L matchers), SK_TABLE_FULL_KEY
 * or SK_TABLE_FACTORED, or-ed with SK_TABLE_WIDE_CLASSES for a sheet of 8 to 15 letters; *keys = its entries, *bytes = its size
 * on the device.  Any of the three pointers may be NULL.  A host logs it; the tests assert which path they exercise.       */
#define SK_TABLE_NONE         0
#define SK_TABLE_FULL_KEY     1
#define SK_TABLE_FACTORED     2
#define SK_TABLE_WIDE_CLASSES 4
int sk_barcode_table_info(sk_ctx *ctx, int *kind, int64_t *keys, int64_t *bytes);

/* ---- T1: 3' running-sum quality trim --------------------------------------------------------------
 * src/fasta_trim_by_quality.rs:28-42.  qual rows are the quality line after trim_end(); writes
 * lowest_k[r] in [0, len[r]] (0 means the reference emits "N\n+\n!\n", :44-45).                      */
int sk_trim_by_quality(sk_ctx *ctx, const uint8_t *qual, const uint16_t *len, int stride, int64_t n,
                       uint8_t min_baseq, uint16_t *lowest_k);
int sk_trim_by_quality_dev(sk_ctx *ctx, const uint8_t *qual, const uint16_t *len, int stride, int64_t n,
                           uint8_t min_baseq, uint16_t *lowest_k);

/* ---- M1: mask bases whose Phred+33 quality is below min_baseq ------------------------------------
 * src/fasta_mask_by_quality.rs:40-43 (byte form, i.e. ASCII lines; the host keeps non-ASCII lines).
 * out_seq may equal seq (in place).                                                                  */
int sk_mask_by_quality(sk_ctx *ctx, uint8_t *seq /* in/out */, const uint8_t *qual, const uint16_t *len,
                       int stride, int64_t n, uint8_t min_baseq);
int sk_mask_by_quality_dev(sk_ctx *ctx, const uint8_t *seq, const uint8_t *qual, int stride, int64_t n,
                           uint8_t min_baseq, uint8_t *out_seq);

/* ---- fused per-read pass: demultiplex (+ add barcode) + trim by quality + mask by quality ---------
 * One pass over a batch of n clusters with 1 or 2 mates.  `add barcode` (src/fasta_add_barcode.rs:
 * 19-44) is realised by handing the index read's bases over as the bc column instead of round-tripping
 * them through the header text; header handling (D4/D5) stays on the host.  Any of the three parts can
 * be switched off by leaving its outputs NULL:
 *   demultiplex : bc != NULL           -> assign (+ optional lowest_diff/first_idx/last_idx, counters)
 *   trim        : mate.lowest_k != NULL
 *   mask        : mate.out_seq  != NULL (may alias mate.seq)
 * The same struct serves the host and the _dev entry point.                                          */
typedef struct {
	const uint8_t  *seq;        /* n x stride  (needed only for mask)           */
	const uint8_t  *qual;       /* n x stride                                   */
	const uint16_t *len;        /* n, or NULL = all rows are `stride` long      */
	uint8_t        *out_seq;    /* n x stride masked bases, or NULL             */
	uint16_t       *lowest_k;   /* n, or NULL                                   */
} sk_mate;

typedef struct {
	int64_t  n;                 /* clusters in the batch                        */
	int      n_mates;           /* 1 (single end) or 2 (paired end)             */
	int      stride;            /* row stride of seq/qual/out_seq, both mates   */
	uint8_t  min_baseq;         /* threshold for trim and mask                  */
	sk_mate  mate[2];
	const uint8_t *bc;          /* n x bc_stride observed barcodes, or NULL     */
	int      bc_stride;
	int32_t *assign;            /* n                                            */
	uint8_t *lowest_diff;       /* n or NULL                                    */
	int16_t *first_idx;         /* n or NULL                                    */
	int16_t *last_idx;          /* n or NULL                                    */
	uint64_t *counts;           /* _dev only: device u64[S+3] or NULL = ctx's   */
} sk_fused_args;

int sk_fused_pass(sk_ctx *ctx, const sk_fused_args *args);
int sk_fused_pass_dev(sk_ctx *ctx, const sk_fused_args *args);
/* Many independent batches in one call: what n_batches calls of sk_fused_pass_dev compute (same outputs; counters are sums, so
 * their order does not matter) — checked once, the sheet's table uploaded once.  Batches that are barcode assignment alone
 * and whose sheet is served by a table in LDS (sk_barcode_table_info) run as ONE launch whose waves walk the steps of all
 * batches (at most 256 batches a launch; the table staged once, ramp and tail paid once); every other shape is the
 * batches' launches back to back on the ctx stream — on the ctx's two streams in turn when no batch has a barcode phase (trim /
 * mask alone): sk_sync() waits for both — (src/fasta_demultiplex.rs:154-194, src/fasta_trim_by_quality.rs:28-42
 * work per read: reads are independent, so are batches).  Asynchronous like the _dev calls; sk_sync() waits.  The two
 * conveniences below build the argument blocks. */
int sk_fused_pass_many_dev(sk_ctx *ctx, const sk_fused_args *batches, int n_batches);
typedef struct sk_demux_batch {
	const uint8_t *bc;          /* n x bc_stride */
	int64_t n;
	int32_t *assign;            /* n */
	uint8_t *lowest_diff;       /* n or NULL */
	int16_t *first_idx, *last_idx;
} sk_demux_batch;
int sk_demux_assign_many_dev(sk_ctx *ctx, const sk_demux_batch *batches, int n_batches, int bc_stride);
typedef struct sk_trim_batch {
	const uint8_t *qual;        /* n x stride */
	const uint16_t *len;        /* n or NULL */
	int64_t n;
	uint16_t *lowest_k;         /* n */
} sk_trim_batch;
int sk_trim_by_quality_many_dev(sk_ctx *ctx, const sk_trim_batch *batches, int n_batches, int stride, uint8_t min_baseq);

/* ---- placement tuning for resident batches ---------------------------------------------------------------------------
 * WHERE the pages of a device buffer lie moves the fused pass by up to 12 % on one and the same GPU: measured on MI355X,
 * the same kernel on the same bytes takes 9.35 ms with one set of allocations and 10.6 ms with another, each reproducible
 * to 0.1 % for the life of the buffers, bimodal, and decided by how the buffers of the streams that are active together
 * relate to each other — not by their addresses' alignment, padding or order (tools/soa_placement_search.py,
 * tools/arena_sweep.py).  A host whose batch buffers live long (a resident shard that is processed many
 * times, a ring of staging buffers) can therefore choose once: it allocates k candidate device buffers for each big
 * matrix of the pass, fills the input candidates with the same bytes, and this call times the pass while it swaps one
 * matrix at a time for its other candidates (coordinate descent, `sweeps` rounds, 1 + 2 launches per probe) and leaves
 * the fastest combination in `args`.  The losers are the caller's to free.  Counters are not touched (the probes count
 * into a scratch vector).  ms_before / ms_after: the pass on the first candidates and on the chosen ones.            */
#define SK_MAX_CANDIDATES 8
typedef struct {
	int k;                                              /* candidates per matrix, 1..SK_MAX_CANDIDATES            */
	const uint8_t *seq[2][SK_MAX_CANDIDATES];           /* per mate; entries of an unused mate / matrix are NULL  */
	const uint8_t *qual[2][SK_MAX_CANDIDATES];
	uint8_t *out_seq[2][SK_MAX_CANDIDATES];
} sk_fused_candidates;
int sk_fused_tune_placement_dev(sk_ctx *ctx, sk_fused_args *args, const sk_fused_candidates *cands, int sweeps,
                                float *ms_before, float *ms_after, int *n_probes);

/* ---- the fused pass over a TILE-BLOCKED batch -----------------------------------------------------------------
 * Same arithmetic and the same reference lines as sk_fused_pass; only where the bytes sit differs.  A batch is cut
 * into tiles of 64 consecutive clusters.  Everything tile t READS is one contiguous block of `in_block` bytes at
 * in + t*in_block, everything it WRITES one block of `out_block` bytes at out + t*out_block; inside a block every
 * array of the 64 clusters is one segment: row (r mod 64) of mate m's qualities at  in_qual[m] + (r mod 64)*stride,
 * its bases at in_seq[m] + ..., the observed barcode at in_bc + (r mod 64)*bc_stride, the optional u16 length at
 * in_len[m] + 2*(r mod 64); outputs likewise (out_seq[m], u16 out_lowest_k[m], i32 out_assign, and with
 * SK_BLK_DETAIL u8 out_lowest_diff, i16 out_first_idx, i16 out_last_idx).  Offsets of absent segments are -1.
 * Both buffers hold WHOLE blocks also for the last, partial tile ((n+63)/64 blocks; the padding rows are read and
 * their outputs are unspecified).  Why: a device wave streams one read range and one write range per tile, and a
 * host moves a batch with ONE copy per direction.  sk_blocked_layout_init fills the struct from the shape; the
 * packer (host) owns the layout, so no existing file format is touched.  Served shapes: stride <= 960; with
 * barcodes, sheets of <= 128 samples over <= 7 distinct non-wildcard bytes, L <= 31, bc_stride <= 32 — everything
 * else takes sk_fused_pass(_dev) (SK_ERR_INVALID here).                                                          */
#define SK_BLK_MASK   1     /* out_seq segments (mask by quality)                 */
#define SK_BLK_TRIM   2     /* out_lowest_k segments (trim by quality)            */
#define SK_BLK_LEN    4     /* in_len segments (ragged rows); else every row is `stride` long */
#define SK_BLK_DETAIL 8     /* lowest_diff / first_idx / last_idx beside assign   */
typedef struct {
	int32_t n_mates, stride, bc_stride, flags;   /* bc_stride 0 = no barcodes (no demultiplex)                  */
	int32_t in_block, out_block;                 /* bytes per tile of 64 clusters (multiples of 128)            */
	int32_t in_qual[2], in_seq[2], in_len[2], in_bc;
	int32_t out_seq[2], out_lowest_k[2], out_assign, out_lowest_diff, out_first_idx, out_last_idx;
} sk_blocked_layout;
int sk_blocked_layout_init(sk_blocked_layout *lay, int n_mates, int stride, int bc_stride, int flags);
/* in/out: device buffers of ((n+63)/64) * in_block / out_block bytes, 16-byte aligned; counts as in sk_fused_args */
int sk_fused_pass_blocked_dev(sk_ctx *ctx, const sk_blocked_layout *lay, const uint8_t *in, uint8_t *out, int64_t n,
                              uint8_t min_baseq, uint64_t *counts);

/* ---- counters: sample.total_reads[S], total_reads, identified_reads (+ ambiguous) -----------------
 * src/fasta_demultiplex.rs:108-109,169,177-178.  Layout u64[S+3]: per-sample counts, then [S] total,
 * [S+1] identified, [S+2] ambiguous.  These are the only cross-shard state of the path; a multi-GPU
 * host sums them with one all-reduce (sk_counts_device_ptr gives the device buffer to reduce).       */
int sk_counts_reset(sk_ctx *ctx);
int sk_counts_get(sk_ctx *ctx, uint64_t *counts /* host, S+3 */);
/* The ctx's device u64[S+3].  It holds what was enqueued on the ctx BEFORE this call (once the ctx stream has got there):
 * ask again after later demultiplex calls, do not keep the pointer across them.                              */
void *sk_counts_device_ptr(sk_ctx *ctx);

/* ---- (e) multi-GPU: the count reduce over RCCL / xGMI ---------------------------------------------------------
 * Reads shard trivially (every cluster is independent); the additive counters above — and the BAM counters and
 * histogram — are the only state that crosses shards (src/fasta_demultiplex.rs:108-109,169,177-178 are plain `+= 1`
 * on one thread in the reference).  RCCL (librccl.so.1, loaded on first use) sums them; there is no other collective.
 *  - one process driving several GPUs (the C++ hosts): sk_counts_allreduce(ctxs, n) sums the u64[S+3] counters of
 *    all n ctxs in place, so that every ctx ends with the totals.  Ctxs that share a device are summed on that device
 *    first; distinct devices then take part in one ncclAllReduce(ncclSum, ncclUint64) on their ctx streams (a
 *    communicator per device set is created with ncclCommInitAll on first use).  Synchronous when n > 1.
 *  - one process per GPU (bench.py, torchrun): rank 0 calls sk_comm_get_unique_id, the host hands the 128 bytes to
 *    every rank (any side channel), every rank calls sk_comm_init_rank; afterwards sk_counts_allreduce(&ctx, 1) and
 *    sk_allreduce_u64_dev sum across the ranks, enqueued on the ctx stream (asynchronous; sk_sync waits).          */
#define SK_COMM_ID_BYTES 128
/* rank-local, talks to nobody: SK_OK when this ctx could join a communicator (librccl loadable, device bindable).
 * sk_comm_init_rank blocks until every rank has called it, so hosts exchange these answers first and only then join. */
int sk_comm_ready(sk_ctx *ctx);
int sk_comm_get_unique_id(uint8_t id[SK_COMM_ID_BYTES]);
int sk_comm_init_rank(sk_ctx *ctx, const uint8_t id[SK_COMM_ID_BYTES], int rank, int n_ranks);
int sk_comm_destroy(sk_ctx *ctx);
int sk_counts_allreduce(sk_ctx **ctxs, int n_ctx);
/* in-place sum of a device u64 vector over the ranks of sk_comm_init_rank (BAM counters + histogram, or counters
 * a host keeps in its own device buffer); a ctx without a communicator is its own world: nothing to do.           */
int sk_allreduce_u64_dev(sk_ctx *ctx, uint64_t *buf, size_t count);

/* ---- S1 + H1: BAM flag counters and |TLEN| histogram -----------------------------------------------
 * src/sam_statistics.rs:63-69 (counters[0]=total, [1]=aligned, [2]=duplicate) and
 * src/sam_fragment_lengths.rs:29-43 (hist[0..max_frag], *hist_total = records histogrammed).  Inputs
 * are the BAM fixed-core fields as SoA columns (flag@14, refID@0, next_refID@20, tlen@28 of the 32-byte
 * core).  Results are ADDED to the caller's arrays.  The --reads=N early stop (:42) is order dependent
 * and stays on the host.  counters or hist may be NULL to skip that half.                             */
int sk_bam_flag_tlen(sk_ctx *ctx, const uint16_t *flag, const int32_t *tid, const int32_t *mtid,
                     const int32_t *tlen, int64_t n, int32_t max_frag,
                     uint64_t counters[3], uint64_t *hist, uint64_t *hist_total);
/* out = device u64[3 + 1 + (max_frag+1)]: counters, hist_total, hist; ADDED to. */
int sk_bam_flag_tlen_dev(sk_ctx *ctx, const uint16_t *flag, const int32_t *tid, const int32_t *mtid,
                         const int32_t *tlen, int64_t n, int32_t max_frag, uint64_t *out);

/* ---- B1 on the device: BGZF inflate and the BAM record walk ------------------------------------
 * src/common.rs:121-157: the reference reads a BAM through htslib, which inflates every BGZF block (SAMv1 §4.1: a gzip
 * member of at most 64 KiB, self-contained), checks its CRC-32 and walks the records (block_size + 32-byte core + ...).
 * Here the compressed file crosses PCIe and the device does all three.
 *
 * sk_bgzf_inflate_dev: blocks[i] says where block i's raw DEFLATE payload lies in comp (in_off, in_len: what is between
 * the gzip header and the 8-byte trailer), how many bytes it inflates to (out_len = the trailer's ISIZE), where they go
 * in out (out_off) and the trailer's CRC32.  comp must be readable up to the next 4-byte boundary behind the last
 * payload; out 16-byte aligned.  status[i] (device u32): 0 = block i was inflated (and, with check_crc, its CRC
 * matched); 1..8 = the decoder gave the block up (an irregular code, a distance before the block, sizes that do not
 * match: what zlib would call a data error, and a few legal rarities) — nothing is decided here: the caller inflates
 * such a block with zlib, whose verdict stands; bit 8 (0x100) = CRC mismatch.  All pointers are device pointers.
 *
 * sk_bam_walk_dev: stream = the inflated blocks back to back (out above, stream_len bytes, readable 8 bytes beyond),
 * block_end[c] = where block c ends in it (device u64[n], ascending), first_record = where the first record begins
 * (behind the BAM header), n_ref = the header's number of references (or -1: not used by the guesses).  Every block is
 * walked from a guessed entry — its first byte; for the blocks where that is no record's first byte, the first offset at
 * which three records in a row look like records — and the guesses are verified against the predecessors' exits until
 * nothing changes (at most max_rounds rounds).  *verified = 1: entry[c] (device u64[n]) is
 * where the first record that begins in block c begins, for every c — the chain from first_record, proven block by
 * block — and it ends exactly at stream_len; *n_records = records in the stream.  *verified = 0: the stream is not a
 * well-formed sequence of records (or did not settle): the caller's record-at-a-time path reports it as the reference
 * would.  exit_scratch (device u64[n]) and nrec_scratch (device u32[n + 1]) are work space.
 *
 * sk_bam_walk_reduce_dev: S1 + H1 (sk_bam_flag_tlen's predicates) over the records of a verified chain, read straight
 * from the inflated bytes.  out = device u64[3 + 1 + (max_frag + 1)] as sk_bam_flag_tlen_dev, ADDED to.
 *
 * sk_bam_file_reduce: the three above over a whole BAM file: reads it (pread into pinned buffers), ships the compressed
 * bytes, inflates, walks, reduces; blocks the device gave up are inflated with zlib on the host.  counters / hist as
 * sk_bam_flag_tlen (ADDED to; either may be NULL).  *handled = 0 (and nothing added): the file is not one this path
 * serves — not a regular file, not BGZF, cut short, a record chain that does not verify, a block zlib rejects too —
 * and the caller falls back to its record-at-a-time reader, which produces the reference's output and messages.
 * The device and page-locked buffers of a call (the compressed file; for the inflated stream a reserved address range of six
 * times the file's size into which memory is mapped as far as the file inflates) stay with the ctx for the next call;
 * sk_destroy frees them.
 * info (may be NULL): [0] compressed bytes, [1] inflated bytes, [2] BGZF blocks, [3] records, [4] blocks inflated by
 * zlib on the host, [5] walk rounds, [6] ms reading + copying, [7] ms of device work behind the last copy.          */
typedef struct sk_bgzf_block {
	uint64_t in_off;               /* of the DEFLATE payload in comp */
	uint32_t in_len, out_len;
	uint64_t out_off;
	uint32_t crc32, reserved;
} sk_bgzf_block;
int sk_bgzf_inflate_dev(sk_ctx *ctx, const uint8_t *comp, const sk_bgzf_block *blocks, int64_t n_blocks, uint8_t *out,
                        uint32_t *status, int check_crc);
int sk_bam_walk_dev(sk_ctx *ctx, const uint8_t *stream, uint64_t stream_len, const uint64_t *block_end, int64_t n,
                    uint64_t first_record, int32_t n_ref, uint64_t *entry, uint64_t *exit_scratch, uint32_t *nrec_scratch,
                    int max_rounds, int *verified, uint64_t *n_records, int *rounds);
int sk_bam_walk_reduce_dev(sk_ctx *ctx, const uint8_t *stream, uint64_t stream_len, const uint64_t *block_end,
                           const uint64_t *entry, int64_t n, int32_t max_frag, int want_counters, int want_hist, uint64_t *out);
int sk_bam_file_reduce(sk_ctx *ctx, const char *path, int32_t max_frag, uint64_t counters[3], uint64_t *hist,
                       uint64_t *hist_total, int *handled, double info[8]);

/* ---- F2 on the device: the gzip writers' DEFLATE ------------------------------------------------
 * src/common.rs:49-81: every output file of the reference is a pipe into a gzip / pigz child; what a test can hold it to is
 * the decompressed stream.  sk_bgzf_deflate compresses n independent blocks of at most SK_DEFLATE_MAX_IN bytes (in +
 * blocks[i].in_off, in_len bytes; host pointers) into n complete BGZF members (SAMv1 §4.1: gzip members with the BC
 * subfield, CRC32 and ISIZE) written back to back into out: member i is out[out_off[i] .. out_off[i + 1]).  The device
 * finds the matches (a hash of four bytes, greedy), builds a Huffman code per block and writes the bits; a block that does
 * not shrink is framed as a stored block.  out_cap >= n * SK_DEFLATE_MAX_MEMBER always suffices.  in must be readable to the
 * next 4-byte boundary behind its last byte.
 * sk_bgzf_deflate_dev is the device half alone: slots[i * slot_stride ..] receives block i's DEFLATE payload (slot_stride
 * >= SK_DEFLATE_SLOT, a multiple of 4), result[2 i] its byte count, crc[i] the CRC-32 of the block's input; tokens is
 * scratch of n * SK_DEFLATE_MAX_IN dwords.  All pointers device pointers.                                             */
#define SK_DEFLATE_MAX_IN     0xff00
#define SK_DEFLATE_MAX_MEMBER 65536
#define SK_DEFLATE_SLOT       81920
typedef struct sk_deflate_block {
	uint64_t in_off;
	uint32_t in_len, reserved;
} sk_deflate_block;
int sk_bgzf_deflate(sk_ctx *ctx, const uint8_t *in, size_t in_bytes, const sk_deflate_block *blocks, int64_t n_blocks,
                    uint8_t *out, size_t out_cap, uint64_t *out_off);
int sk_bgzf_deflate_dev(sk_ctx *ctx, const uint8_t *in, const sk_deflate_block *blocks, int64_t n_blocks, uint8_t *slots,
                        uint32_t slot_stride, uint32_t *tokens, uint32_t *result, uint32_t *crc);

/* ---- f2: `sam fragments` record filter ---------------------------------------------------------------------
 * src/sam_fragments.rs:27-38: keep the forward mate of a converging, mapped, primary, non-duplicate, QC-passing pair on
 * one reference whose |tlen| lies in [min_size, max_size].  keep_bits: (n+7)/8 bytes, bit j of byte k <=> record 8k+j;
 * *kept is ADDED to.  The BED line of a kept record (:41) is text and stays on the host.  Columns must be 16-byte
 * aligned for the _dev form.                                                                                */
int sk_bam_fragments(sk_ctx *ctx, const uint16_t *flag, const int32_t *tid, const int32_t *mtid, const int32_t *tlen,
                     int64_t n, int64_t min_size, int64_t max_size, uint8_t *keep_bits, uint64_t *kept);
int sk_bam_fragments_dev(sk_ctx *ctx, const uint16_t *flag, const int32_t *tid, const int32_t *mtid, const int32_t *tlen,
                         int64_t n, int64_t min_size, int64_t max_size, uint8_t *keep_bits, uint64_t *kept);

/* ---- f2 (second half): `sam count` ---------------------------------------------------------------------------
 * src/sam_count.rs:44-127.  sk_count_set_regions loads the BED regions grouped by BAM reference: regions of
 * reference c are entries chr_off[c] .. chr_off[c+1]-1 of rstart/rend (0-based half-open, any order inside a
 * group; they are sorted by start here, as :63 does) and ridx gives each entry's index in the caller's list of
 * n_regions regions (n_entries <= n_regions: regions on chromosomes the BAM does not have are simply not entered);
 * the n_regions counters (u32, like the reference's) are cleared.  sk_count_add runs, for every record, the filter
 * chain (:46-50, :78-94), the fragment interval in the reference's u32 arithmetic (:75,97-107) and adds 1 to every
 * region of the record's reference that the interval overlaps (:122-126).  Columns: flag, mapq, refID, next_refID,
 * pos, next_pos, tlen of the BAM core and, for single_end only, cigar end_pos (NULL otherwise).  What depends on
 * record order — the "not coordinate sorted" error (:70-72) and chr_names[tid] (:55) — is the caller's.  The counts
 * do not depend on the order of the records; the speed does: the region search of a record starts from its
 * predecessor's answer, which is two probes in a coordinate-sorted batch and more than a plain search otherwise.      */
int sk_count_set_regions(sk_ctx *ctx, int n_chr, const int32_t *chr_off, const uint32_t *rstart, const uint32_t *rend,
                         const int32_t *ridx /* NULL = 0,1,2,... */, int64_t n_entries, int64_t n_regions);
int sk_count_add(sk_ctx *ctx, const uint16_t *flag, const uint8_t *mapq, const int32_t *tid, const int32_t *mtid,
                 const int32_t *pos, const int32_t *mpos, const int32_t *tlen, const int32_t *end_pos, int64_t n,
                 uint8_t min_mapq, uint32_t max_frag_len, int single_end, int center);
int sk_count_add_dev(sk_ctx *ctx, const uint16_t *flag, const uint8_t *mapq, const int32_t *tid, const int32_t *mtid,
                     const int32_t *pos, const int32_t *mpos, const int32_t *tlen, const int32_t *end_pos, int64_t n,
                     uint8_t min_mapq, uint32_t max_frag_len, int single_end, int center);
int sk_count_get(sk_ctx *ctx, uint32_t *region_frags /* n_regions */);

/* ---- `fasta gc content` ------------------------------------------------------------------------------------------
 * src/fasta_gc_content.rs:41-46.  sk_gc_set_genome copies the concatenated sequences to the device once (it stays
 * there until the next call or sk_destroy); sk_gc_count then returns, for region i = bytes [start[i], start[i]+len[i])
 * of that buffer, gc[i] = bytes that are C, G, c or g and total[i] = bytes that are neither N nor n.  Region bounds are
 * the caller's to check (the reference's `chr_seq.get(start..stop)`, :41).                                          */
int sk_gc_set_genome(sk_ctx *ctx, const uint8_t *genome, int64_t genome_len);
int sk_gc_count(sk_ctx *ctx, const int64_t *start, const int64_t *len, int64_t n_regions, uint64_t *gc, uint64_t *total);

/* ---- f4: `sam to fastq` sequence() ---------------------------------------------------------------------------
 * src/sam_to_fastq.rs:31-59: the bases of BAM records as ASCII — codes 1,2,4,8 -> A,C,G,T, anything else N; records
 * with flag & 0x10 come out reverse-complemented; a base whose quality is below min_baseq (the reference passes 10,
 * :103) is N.  SoA layout: row r of seq4 at seq4 + r*seq4_stride holds the record's packed bases exactly as BAM stores
 * them (two per byte, even positions in the high nibble); rows of qual (raw phred bytes) and out at + r*stride.
 * stride and seq4_stride must be multiples of 4 with 2*seq4_stride >= stride; len NULL = every row holds `stride`
 * bases; flag is the BAM flag column.  Bytes of out past a row's length are unspecified.  Pairing mates by qname
 * and the text around the bases (:96-149) stay on the host.                                                     */
int sk_bam_sequence(sk_ctx *ctx, const uint8_t *seq4, int seq4_stride, const uint8_t *qual, int stride,
                    const uint16_t *len, const uint16_t *flag, int64_t n, uint8_t min_baseq, uint8_t *out);
int sk_bam_sequence_dev(sk_ctx *ctx, const uint8_t *seq4, int seq4_stride, const uint8_t *qual, int stride,
                        const uint16_t *len, const uint16_t *flag, int64_t n, uint8_t min_baseq, uint8_t *out);

/* ---- f3: barcode census -----------------------------------------------------------------------------------
 * The HashMap<String, u64> of src/fasta_demultiplex.rs:190-194 (dry run: barcodes that matched no sample) and of
 * src/fasta_statistics.rs:23-27 (every BC: field), kept as a hash table in HBM that lives in the ctx.
 * bc is the SoA barcode matrix of sk_demux_assign: row r at bc + r*bc_stride, the barcode in its first L bytes,
 * ended early by a NUL.  L <= 31, bc_stride <= 64, alphabet ACGTNacgtn+ (what the reference's regexes admit); a row holding any
 * other byte is not counted and is reported in stats[2] (the host keeps such rows in its own map).  When assign is
 * not NULL only rows with assign[r] == SK_ASSIGN_NONE are counted (:190).  row_base + r is remembered as the first
 * occurrence of a barcode, so that results can be listed in first-seen order whatever the launch order was.    */
typedef struct sk_census_entry {
	char barcode[32];              /* NUL-terminated */
	uint64_t count;
	int64_t first_row;
} sk_census_entry;
int sk_census_reset(sk_ctx *ctx);                                 /* empty census (creates it on first use) */
int sk_census_add(sk_ctx *ctx, const uint8_t *bc, int bc_stride, int L, int64_t n, const int32_t *assign, int64_t row_base);
int sk_census_add_dev(sk_ctx *ctx, const uint8_t *bc, int bc_stride, int L, int64_t n, const int32_t *assign, int64_t row_base);
/* stats[0] distinct barcodes, [1] rows counted, [2] rows rejected, [3] table slots */
int sk_census_stats(sk_ctx *ctx, uint64_t stats[4]);
/* hist[b] = number of distinct barcodes whose count c has floor(log2(c)) == b: lets a caller pick min_count */
int sk_census_count_hist(sk_ctx *ctx, uint64_t hist[64]);
/* barcodes with count >= min_count in first-seen order; at most cap are written, *total = how many qualify */
int sk_census_entries(sk_ctx *ctx, uint64_t min_count, sk_census_entry *out, uint64_t cap, uint64_t *total);

/* ---- timing on the ctx stream (hipEvents), so a host without HIP can time device work -------------- */
int sk_timer_start(sk_ctx *ctx);
int sk_timer_stop(sk_ctx *ctx, float *ms);               /* records, synchronises, returns elapsed ms */

#ifdef __cplusplus
}
#endif
#endif
