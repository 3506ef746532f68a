/*
 * seqkit_oracle.h — CPU ORACLE for the seqkit per-read hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (seqkit_amd/, the
 * C-ABI library in include/seqkit_hip.h) never links, imports or executes anything
 * under oracle/.
 *
 * PARITY UNPINNED: the reference (annalam/seqkit v0.8.0) is Rust, cannot be built in
 * this image (no cargo/rustc, no crates), and ships no tests, golden vectors or
 * fixtures for this path.  This file is a plain-C restatement of
 * the reference's loops, written from the cited source lines; it is checked against
 * the hand-derived known-answer vectors in tests/golden/ (SURVEY.md Appendix A), not
 * against outputs of the reference binary.
 *
 * Citations are relative to /root/reference/.  Release-mode Rust semantics are
 * assumed throughout (u8 subtraction wraps), which is what `cargo install` builds
 * (README.md:28-30).
 */
#ifndef SEQKIT_ORACLE_H
#define SEQKIT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- T1: src/fasta_trim_by_quality.rs:28-42 ------------------------------------ */
/* qual = the quality line after Rust str::trim_end(); returns lowest_k in [0, n]. */
uint32_t orc_trim_lowest_k(const uint8_t *qual, uint32_t n, uint8_t min_baseq);

/* ---- M1: src/fasta_mask_by_quality.rs:40-43 (byte form; valid for ASCII lines) -- */
void orc_mask_bytes(uint8_t *seq, const uint8_t *qual, uint32_t n, uint8_t min_baseq);

/* ---- D1: src/fasta_demultiplex.rs:269-277 -------------------------------------- */
size_t orc_barcode_diff(const uint8_t *observed, const uint8_t *candidate, size_t len);

/* ---- D2: src/fasta_demultiplex.rs:154-166 -------------------------------------- */
typedef struct {
	uint64_t lowest_diff;      /* UINT64_MAX when S == 0 (usize::MAX) */
	uint32_t best_sample;      /* first index attaining the minimum   */
	uint32_t equally_fine;     /* last index attaining the minimum    */
} orc_match;
orc_match orc_best_match(const uint8_t *observed, const uint8_t *table /* S*L row-major */,
                         int S, int L);

/* ---- D3: src/fasta_demultiplex.rs:168-194 -------------------------------------- */
/* assignment code: >= 0 sample index, ORC_NONE no sample within max_diff,
 * ORC_AMBIGUOUS two or more samples equally close (read dropped).                  */
#define ORC_NONE      (-1)
#define ORC_AMBIGUOUS (-2)
int32_t orc_decide(orc_match m, uint64_t max_diff);

/* ---- batch (SoA, fixed stride) forms: same arithmetic over packed buffers ------- */
/* row r lives at base + r*stride; only the first len[r] bytes are data; len == NULL
 * means every row holds exactly `stride` bytes of data.                             */
void orc_trim_batch(const uint8_t *qual, const uint16_t *len, int stride, int64_t n,
                    uint8_t min_baseq, uint16_t *lowest_k);
void orc_mask_batch(uint8_t *seq, const uint8_t *qual, const uint16_t *len, int stride,
                    int64_t n, uint8_t min_baseq);
/* counts: S per-sample, then [S]=total, [S+1]=identified, [S+2]=ambiguous; ADDED to. */
void orc_demux_batch(const uint8_t *table, int S, int L, int max_diff,
                     const uint8_t *bc, int bc_stride, int64_t n,
                     int32_t *assign, uint8_t *lowest_diff /* saturated at 255, nullable */,
                     int16_t *first_idx /* nullable */, int16_t *last_idx /* nullable */,
                     uint64_t *counts /* nullable */);

/* ---- S1 + H1: src/sam_statistics.rs:63-69, src/sam_fragment_lengths.rs:29-43 ---- */
/* counters[0]=total, [1]=aligned, [2]=duplicate (S1); hist[0..max_frag] (H1), and
 * *hist_total = number of records that entered the histogram.  All ADDED to.        */
void orc_bam_flag_tlen(const uint16_t *flag, const int32_t *tid, const int32_t *mtid,
                       const int32_t *tlen, int64_t n, int32_t max_frag,
                       uint64_t counters[3], uint64_t *hist, uint64_t *hist_total);
/* H1 with the --reads=N early stop (src/sam_fragment_lengths.rs:42): returns the
 * number of records consumed from the input before stopping.                        */
int64_t orc_fragment_lengths_stop(const uint16_t *flag, const int32_t *tid, const int32_t *mtid,
                                  const int32_t *tlen, int64_t n, int32_t max_frag,
                                  uint64_t stop_after, uint64_t *hist, uint64_t *hist_total);

/* ---- f2: src/sam_fragments.rs:27-41 ------------------------------------------------ */
/* keep[i] = 1 when record i is the forward mate of a converging, mapped, primary, non-duplicate, QC-passing pair on
 * one reference whose |tlen| lies in [min_size, max_size]; returns the number kept.  */
int64_t orc_fragments_keep(const uint16_t *flag, const int32_t *tid, const int32_t *mtid, const int32_t *tlen,
                           int64_t n, int64_t min_size, int64_t max_size, uint8_t *keep);

/* ---- f4: src/sam_to_fastq.rs:31-59 sequence() --------------------------------------- */
/* seq4 = the BAM record's packed bases (two per byte, base k in the HIGH nibble of byte
 * k/2 when k is even: rust-htslib Seq::encoded_base), qual = its l raw phred bytes,
 * reverse = flag & 0x10.  out gets l ASCII bases: 1,2,4,8 -> A,C,G,T (anything else N);
 * a reverse-strand record is emitted reverse-complemented; a base whose quality is below
 * min_baseq is N.  (The caller appends qualities in STORED order even for reverse reads,
 * src/sam_to_fastq.rs:107-110 — that is the reference's behaviour and is kept.)          */
void orc_bam_sequence(const uint8_t *seq4, const uint8_t *qual, uint32_t l, int reverse,
                      uint8_t min_baseq, uint8_t *out);
/* SoA batch: row r of seq4 at seq4 + r*seq4_stride, of qual/out at + r*stride; len NULL =
 * every row holds `stride` bases; flag = the BAM flag column.                             */
void orc_bam_sequence_batch(const uint8_t *seq4, int seq4_stride, const uint8_t *qual, int stride,
                            const uint16_t *len, const uint16_t *flag, int64_t n,
                            uint8_t min_baseq, uint8_t *out);

/* ---- f2 (second half): src/sam_count.rs:44-127, one record ---------------------------- */
typedef struct {
	uint8_t  min_mapq;          /* --min-mapq      */
	uint32_t max_frag_len;      /* --max-frag-len  */
	int      single_end;        /* --single-end    */
	int      count_centers;     /* --center        */
} orc_count_params;
/* The per-record part of the loop: the filter chain (:46-50, :78-94), the fragment interval
 * with the reference's u32 arithmetic (:75,97,99-107) — returns 0 when the record is skipped,
 * else 1 with the 0-based half-open [*start, *end).  The chromosome bookkeeping and the
 * sortedness check (:52-73) are the caller's.  end_pos = cigar().end_pos().               */
int orc_count_interval(uint16_t flag, uint8_t mapq, int32_t tid, int32_t mtid, int32_t pos, int32_t mpos,
                       int32_t tlen, int32_t end_pos, const orc_count_params *p, uint32_t *start, uint32_t *end);
/* The overlap sweep of :113-126 over one chromosome's regions (indices sorted by start):
 * region_frags[idx[i]] += 1 for every region with start < end and end > start_f.           */
void orc_count_overlaps(const uint32_t *rstart, const uint32_t *rend, const int64_t *idx, int64_t nreg,
                        uint32_t start, uint32_t end, uint32_t *region_frags);

/* The whole loop body of src/sam_count.rs:45-126 for one record, with the state the loop
 * carries (prev_chr, prev_pos, the deque of the current chromosome's regions).  rchr[r] is
 * the index of the BAM reference whose name equals region r's chromosome, or -1.  Returns
 * 0, ORC_COUNT_UNSORTED (:71 "Input BAM file is not coordinate sorted.") or
 * ORC_COUNT_BAD_TID (:55 chr_names[tid] out of bounds: a Rust panic).                     */
#define ORC_COUNT_UNSORTED 1
#define ORC_COUNT_BAD_TID  2
typedef struct { int32_t prev_chr; int64_t prev_pos; int64_t *deque; int64_t front, len, cap; } orc_count_state;
void orc_count_state_init(orc_count_state *st);
void orc_count_state_free(orc_count_state *st);
int orc_count_record(orc_count_state *st, uint16_t flag, uint8_t mapq, int32_t tid, int32_t mtid, int32_t pos,
                     int32_t mpos, int32_t tlen, int32_t end_pos, const orc_count_params *p, int32_t n_chr,
                     const int32_t *rchr, const uint32_t *rstart, const uint32_t *rend, int64_t n_regions,
                     uint32_t *region_frags);
/* n records through orc_count_record; returns 0 or the first non-zero code (*where = its record) */
int orc_count_batch(orc_count_state *st, const uint16_t *flag, const uint8_t *mapq, const int32_t *tid,
                    const int32_t *mtid, const int32_t *pos, const int32_t *mpos, const int32_t *tlen,
                    const int32_t *end_pos, int64_t n, const orc_count_params *p, int32_t n_chr,
                    const int32_t *rchr, const uint32_t *rstart, const uint32_t *rend, int64_t n_regions,
                    uint32_t *region_frags, int64_t *where);

/* ---- fasta gc content: src/fasta_gc_content.rs:41-46 --------------------------------- */
/* gc = bytes that are C, G, c or g; total = bytes that are neither N nor n. */
void orc_gc_count(const uint8_t *seq, size_t n, uint64_t *gc, uint64_t *total);

/* ---- text helpers that restate Rust std behaviour used on the path -------------- */
/* str::trim_end(): length of s after removing trailing Unicode White_Space chars
 * (U+0009..000D, 0020, 0085, 00A0, 1680, 2000..200A, 2028, 2029, 202F, 205F, 3000);
 * s must be valid UTF-8.  Note 0x1C..0x1F are NOT whitespace for Rust.               */
size_t orc_trim_end_len(const uint8_t *s, size_t n);
/* str::trim(): returns new start offset, writes new length.                          */
size_t orc_trim_start_off(const uint8_t *s, size_t n);
/* 1 when s[0..n) is well-formed UTF-8 (what BufRead::read_line demands,
 * src/common.rs:106-112).                                                            */
int orc_utf8_valid(const uint8_t *s, size_t n);
/* regex " BC:[ACGTNacgtn+]+" leftmost-first, greedy (src/fasta_demultiplex.rs:38,140):
 * returns 1 and [*start,*end) of the whole match, else 0.                            */
int orc_find_bc_field(const uint8_t *hdr, size_t n, size_t *start, size_t *end);

/* regex " UMI:[^\s]*" leftmost (src/fasta_simplify_read_ids.rs:26,43) */
int orc_find_umi_field(const uint8_t *hdr, size_t n, size_t *start, size_t *end);
/* " BC:[ACGTNacgtn]+" — the `fasta statistics` variant without '+' (src/fasta_statistics.rs:16). */
int orc_find_bc_field_stats(const uint8_t *hdr, size_t n, size_t *start, size_t *end);

/* ---- f3: HashMap<String, u64> census --------------------------------------------------
 * src/fasta_demultiplex.rs:190-194 and src/fasta_statistics.rs:23-27: one
 * `*map.entry(barcode).or_insert(0) += 1` per read.  Rows are bc + r*stride; the barcode is
 * the first L bytes, ended early by a NUL.  With assign != NULL only rows whose code is
 * ORC_NONE are counted (:190).  Returns the number of distinct barcodes and, in *out, one
 * malloc()ed entry per barcode in first-seen order (free with orc_free).  Rust's HashMap
 * iterates in a random order, so the order among equal counts in the reference's printed
 * tables is unspecified; first-seen order is this repository's canonical choice.          */
typedef struct {
	char barcode[32];          /* NUL-terminated; L <= 31 */
	uint64_t count;
	int64_t first_row;
} orc_census_entry;
int64_t orc_census(const uint8_t *bc, int stride, int L, int64_t n, const int32_t *assign,
                   int64_t row_base, orc_census_entry **out);
void orc_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
