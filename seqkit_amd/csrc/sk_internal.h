// sk_internal.h — shared between the C-ABI layer (sk_capi.hip) and the gfx950 kernels (sk_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sk_lut.h"

namespace sk {

constexpr int kWave = 64;             // gfx950 wavefront; one read tile = 64 reads = one wave
constexpr int kTileRows = 64;
constexpr int kLdsPad = 16;           // bytes kept free before and after the LDS tile image
constexpr int kMaxTileStride = 960;   // 64 rows x 960 B + pads + histogram <= 64 KiB of LDS
constexpr int kMaxOneHotLen = 32;     // barcode length served by the one-hot popcount matcher (W <= 8 dwords)
constexpr int kKeyBits = 11;          // trim scan packs (U << 11 | j); rows up to 2047 bytes
constexpr int kMaxLdsHist = 1024;     // per-wave LDS histogram entries (S+3) before falling back to global atomics
constexpr int kAssignNone = -1;       // SK_ASSIGN_NONE
constexpr int kAssignAmbiguous = -2;  // SK_ASSIGN_AMBIGUOUS

// Quality threshold as packed-byte constants (see sk_kernels.hip: lowq_flags).
struct QualConsts {
	uint32_t cl2;      // splat of (256 - t2) & 0x7f
	uint32_t c72;      // 0x80808080 when (256 - t2) >= 128 else 0
	int mode;          // 0: min_baseq == 0 (never masks); 1: g1 & ~g2; 2: g1; 3: g1 | ~g2
	int min_baseq;
};
QualConsts make_qual_consts(int min_baseq);

struct MateDev {
	const uint8_t *seq;
	const uint8_t *qual;
	const uint16_t *len;
	uint8_t *out_seq;
	uint16_t *lowest_k;
};

struct BarcodeDev {
	const uint8_t *raw;        // S x L sheet bytes
	const uint32_t *onehot;    // S x W candidate codes (one-hot class bit, 0x80 wildcard), or nullptr
	const uint8_t *lut;        // L x 256 observed byte -> (class bit | 0x80), or nullptr
	// bit-sliced matcher tables (one blob, copied to LDS by every workgroup), or nullptr:
	//   [0,256)            byte -> class id 0..6 (a byte the sheet uses) or 7 (any other byte)
	//   [256, 256+4G)      valid-candidate mask, G = ceil(S/32) dwords (padded to 16 bytes)
	//   [mm_off, ...)      MM[k][class][g]: bit s%32 of dword g=s/32 set <=> candidate s counts a mismatch at
	//                      position k when the observed byte has that class
	const uint8_t *bs;
	int bs_bytes, bs_mm_off, G;
	int S, L, W, max_diff;
	// neighbourhood table (sk_lut.h; sk_kernels.hip: demux_lut_kernel); nbr.tab == nullptr: the sheet has none
	LutDev nbr;
	// spread counters (sk_kernels.hip: flush_counts_spread), or nullptr: kCountReplicas rows of count_rep_pitch u64, all zero
	// between launches
	unsigned long long *count_rep;
	int count_rep_pitch;
};
constexpr int kCountReplicas = 16;
constexpr int kMaxBitSlicedLen = 31;     // 5 counter planes
constexpr int kMaxBitSlicedBytes = 24 * 1024;

// One batch of a many-batch lookup launch (TileArgs::many).  The launch numbers the steps (256 rows each) of all batches in a row:
// batch b's are [q_begin_b, q_end).  Its pointers are BIASED by its first step — bc - q_begin * 256 * bc_stride, assign - q_begin *
// 256 and so on — so that the kernel's address of launch step gq, row r lands on the batch's own step gq - q_begin; its rows end at
// launch row row_end = q_begin * 256 + n (the launch's rows stay below 2^31).
struct ManyBatch {
	const uint8_t *bc;
	int32_t *assign;
	uint8_t *lowest_diff;
	int16_t *first_idx, *last_idx;
	uint32_t row_end;
	int32_t q_end;
};

struct TileArgs {
	int64_t n;
	int n_mates;
	int stride;
	QualConsts qc;
	MateDev mate[2];
	const uint8_t *bc;
	int bc_stride;
	BarcodeDev table;
	int32_t *assign;
	uint8_t *lowest_diff;
	int16_t *first_idx;
	int16_t *last_idx;
	int detail_matched;             // SK_DETAIL_MATCHED: the detail columns of SK_ASSIGN_NONE rows are unspecified
	unsigned long long *counts;     // device u64[S+3]
	unsigned long long *counts_wide;    // or nullptr: copies of the ctx's counters that the lookup kernel adds to instead of `counts`; the ctx folds
	                                    // them before anything reads.  counts_wide_rows == 0: kCountReplicas copies with one 128-byte line per
	                                    // counter (counter i of copy r at [(r * (S+3) + i) << kCountWideShift]: a few counters that hundreds of
	                                    // workgroups add to at once); > 0 (sheets of more than kCountDenseFrom counters): that many dense rows of
	                                    // S + 3, a workgroup adds to row blockIdx.x % rows — a thousand counters at a line each are a thousand
	                                    // uncoalesced atomics per workgroup, 17 us at the end of a 10 M-pair call of a 1 000-sample sheet
	int counts_wide_rows;
	const ManyBatch *many;          // or nullptr.  Device array of n_many batches: the lookup kernels then walk the steps of all of them in one launch
	int n_many, many_quads;         // (n = the rows of all batches, for the launch's shape; bc / assign / detail pointers above are not used)
};
constexpr int kCountWideShift = 4;
constexpr int kCountDenseFrom = 160;     // S + 3 above this: dense rows
constexpr int kCountDenseRows = 512;
hipError_t launch_counts_fold_wide(unsigned long long *wide, int nc, int dense_rows, unsigned long long *counts, hipStream_t st);

// Tile-blocked batch (include/seqkit_hip.h: sk_blocked_layout): tile t of 64 clusters reads the ONE byte range
// in + t*in_block .. +in_block and writes out + t*out_block .. +out_block; the offsets say where each segment of the
// tile sits inside its block (-1 = absent).  Both buffers hold whole blocks, also for the last, partial tile.
struct BlockedArgs {
	const uint8_t *in;
	uint8_t *out;
	int64_t n;
	int n_mates, stride, bc_stride;
	int in_block, out_block;
	int in_qual[2], in_seq[2], in_len[2], in_bc;
	int out_seq[2], out_lowest_k[2], out_assign, out_lowest_diff, out_first_idx, out_last_idx;
	QualConsts qc;
	BarcodeDev table;
	unsigned long long *counts;     // device u64[S+3]
};
hipError_t launch_tile_blocked(const BlockedArgs &a, int n_cu, hipStream_t st);
// true when the blocked kernel serves this shape (rows fit an LDS tile and the packed scan key; with barcodes: the
// bit-sliced matcher with <= 4 groups and a tile of barcodes in two 1 KiB chunks)
bool blocked_shape_ok(const BlockedArgs &a);

// launchers (all asynchronous on `st`); return hipError_t of the launch
hipError_t launch_tile_pass(const TileArgs &a, int n_cu, hipStream_t st);
// whether launch_tile_pass matches the barcodes inside the tile pass (bit-sliced matcher) instead of with a launch of their own
bool tile_pass_fuses_demux(bool has_bc, bool any_mate, int stride, bool has_bitsliced, int G, int S, int bc_stride);
hipError_t launch_mask_flat(const uint8_t *seq, const uint8_t *qual, uint8_t *out, int64_t bytes,
                            const QualConsts &qc, int n_cu, hipStream_t st);
hipError_t launch_bam_flag_tlen(const uint16_t *flag, const int32_t *tid, const int32_t *mtid, const int32_t *tlen,
                                int64_t n, int32_t max_frag, unsigned long long *out, int want_counters, int want_hist,
                                int n_cu, hipStream_t st);

hipError_t launch_bam_fragments(const uint16_t *flag, const int32_t *tid, const int32_t *mtid, const int32_t *tlen, int64_t n,
                                int64_t min_size, int64_t max_size, uint8_t *keep_bits, unsigned long long *kept, int n_cu, hipStream_t st);

// `sam count`: record columns + the region tables of sk_count_set_regions
struct CountArgs {
	const uint16_t *flag;
	const uint8_t *mapq;
	const int32_t *tid, *mtid, *pos, *mpos, *tlen, *end_pos;
	int64_t n;
	uint32_t min_mapq, max_frag_len;
	int single_end, center;
	int n_chr;
	const int32_t *chr_off;        // n_chr + 1 offsets into the sorted region arrays
	const uint32_t *rstart, *rend, *rpmax;
	const int32_t *ridx;           // sorted position -> region index of the caller
	uint32_t *counts;
};
hipError_t launch_bam_count(const CountArgs &a, int n_cu, hipStream_t st);

hipError_t launch_gc_count(const uint8_t *genome, const int64_t *seg_start, const int32_t *seg_len, const int32_t *seg_region, int64_t nseg,
                           unsigned long long *out, int n_cu, hipStream_t st);

hipError_t launch_bam_sequence(const uint8_t *seq4, int seq4_stride, const uint8_t *qual, int stride, const uint16_t *len, const uint16_t *flag,
                               int64_t n, int min_baseq, uint8_t *out, int n_cu, hipStream_t st);

// ---- BGZF inflate and the BAM record walk on the device (sk_inflate.hip) ----
// blocks: device array of sk_bgzf_block; status: device u32 per block (0 = inflated; 1..8 the decoder gave up; | 0x100 CRC mismatch)
bool tile_pass_demux_by_table(const TileArgs &b);
hipError_t launch_bgzf_inflate(const uint8_t *comp, const void *blocks, int64_t n_blocks, uint8_t *out, uint32_t *status, int check_crc, int n_cu, hipStream_t st);
hipError_t launch_bam_walk(const uint8_t *stream, uint64_t stream_len, const uint64_t *bend, uint64_t *entry, uint64_t *exitp, uint32_t *nrec, int64_t n,
                           uint64_t first, uint32_t *changed, int fix_round /* 0 first walk, 1 fix, 2 guess */, int32_t n_ref, hipStream_t st);
hipError_t launch_bam_walk_reduce(const uint8_t *stream, uint64_t stream_len, const uint64_t *bend, const uint64_t *entry, int64_t n, int32_t max_frag,
                                  int want_counters, int want_hist, unsigned long long *out, hipStream_t st);

// ---- BGZF deflate on the device (sk_deflate.hip) ----
// blocks: device array of sk_deflate_block; out: n_blocks slots of out_stride bytes (a block's payload from the slot's first byte on);
// tokens: device u32[n_blocks * deflate_tokens_per_block()] scratch; result: device u32[2 n] (payload bytes, tokens); crc: device u32[n]
hipError_t launch_bgzf_deflate(const uint8_t *in, const void *blocks, int64_t n_blocks, uint8_t *out, uint32_t out_stride, uint32_t *tokens, uint32_t *result,
                               uint32_t *crc, int n_cu, hipStream_t st);
size_t deflate_tokens_per_block();

// ---- barcode census (sk_census.hip) ----
struct Census;
struct CensusEntry {            // == sk_census_entry of include/seqkit_hip.h
	char barcode[32];           // NUL-terminated
	uint64_t count;
	int64_t first_row;
};
hipError_t census_create(Census **out, hipStream_t st);
void census_destroy(Census *cs);
hipError_t census_reset(Census *cs, int n_cu, hipStream_t st);
hipError_t census_add(Census *cs, const uint8_t *bc, int bc_stride, int L, int64_t n, const int32_t *assign, int64_t row_base,
                      int n_cu, hipStream_t st);
hipError_t census_stats(Census *cs, uint64_t out[4], hipStream_t st);
uint64_t census_slots(const Census *cs);
hipError_t census_count_hist(Census *cs, uint64_t hist[64], int n_cu, hipStream_t st);
hipError_t census_entries(Census *cs, uint64_t min_count, CensusEntry *out, uint64_t cap, uint64_t *total, int n_cu, hipStream_t st);
constexpr int kMaxCensusLen = 31;

}  // namespace sk

// ---- what sk_bamfile.cpp needs of a ctx (sk_capi.hip owns the struct) ----
struct sk_ctx;
namespace sk {
hipStream_t ctx_stream(sk_ctx *c);
hipStream_t ctx_stream2(sk_ctx *c);
int ctx_n_cu(sk_ctx *c);
void *ctx_keep(sk_ctx *c, int slot, size_t bytes, bool pinned, int *rc);      // a buffer that stays with the ctx from call to call (freed by sk_destroy)
size_t ctx_kept_bytes(sk_ctx *c, int slot);
void *ctx_ext(sk_ctx *c);                                       // an object kept with the ctx (freed by sk_destroy through free_fn)
void ctx_set_ext(sk_ctx *c, void *p, void (*free_fn)(void *));
int ctx_bind(sk_ctx *c);                                        // hipSetDevice; SK_OK or an error code with the message set
int ctx_fail(sk_ctx *c, int code, const char *fmt, ...);       // sets sk_last_error, returns code
}  // namespace sk
