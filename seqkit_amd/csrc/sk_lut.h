// sk_lut.h — the sheet's neighbourhood table: D1 + D2 + D3 (src/fasta_demultiplex.rs:154-194, :269-277) as ONE lookup.
//
// With max_diff <= 1 the observed barcodes that get a sample (or the ambiguity verdict) are few: every sheet row and
// its one-substitution neighbours.  The host enumerates them, decides each with the reference's loop and stores
//     key -> (lowest_diff, first argmin, last argmin)
// in a table small enough for a workgroup's LDS; a read is then: classify its bytes, pack, hash, two probes.
//
// Key.  Every byte of a counting column (not a wildcard in EVERY row of the sheet, not the separator) becomes a 3-bit class: the
// index of the sheet letter it equals, or `other` (an index no letter has) for any byte the sheet never uses.  The
// classes of up to 20 columns are packed into two words without a carry or a multiply (lut_pack).  Columns that are a
// wildcard in EVERY row (UMI columns), the separator and the bytes past L are masked out (keepA / keepB).  A row that has a
// wildcard in a counting column matches every byte there (src/fasta_demultiplex.rs:272-273): the host enters it once per
// class of that column — (letters + 1)^w variants for w such columns, refused beyond kMaxKeys keys in all.
// A separator is a column that holds the same letter in every row, a letter no other column uses ('+' of `i7+i5`):
// a mismatch there adds one to the distance of EVERY row, so it is compared on its own and not part of the key — the
// argmin set does not depend on it, and the table is a quarter smaller.  The two segments beside it are read on their
// own (W1 + W2 dwords, W1 == W2): the key has no hole, 8 + 8 columns are four dwords of classes instead of five.
//
// Table.  Two-choice cuckoo, one 8-byte entry per slot, two tables of 2^nb slots.  (A, B) -> (X, B) with
// X = mix(A ^ f(B) ^ seed) is a bijection (f any function, mix invertible), so an entry does not hold the key: the
// slot index is nb bits of X, the entry keeps the other 32 - nb bits of X and B — equality of those IS equality of the
// key (quotienting).  Table 1 is indexed by the top nb bits of X, table 2 by the nb bits below them (X rotated left by nb).
//     w0 = B (31 bits, bit 7 of every byte is never set) | lowest_diff << 31
//     w1 = tag (the low 33 - nb bits of X, or of X rotated) | idx << idx_shift | ambiguous << 31
// The tag is one bit wider than quotienting needs: its top bit repeats the slot's lowest bit, and a FREE slot holds the other
// value there (w1 = (~slot & 1) << (32 - nb), w0 = kLutFree) — no key that hashes to the slot can match it, so a key whose B is
// always zero (at most 8 columns, no separator) is looked up in w1 alone.
// idx = the sample (first == last), or for an ambiguous key the index of its (first, last) pair in a side list: 7 bits for
// sheets of at most 128 samples (idx_shift 24, nb >= 9), 10 bits up to kLutMaxSamples (idx_shift 21, nb >= 12).
// A free slot has w0 = kLutFree (bit 7 set: equals no key).
//
// The factored form (LutDev::pair): 384 dual-index samples x (16 x 4 + 1) keys are 25 k entries — 512 KiB, served from L2 at
// half the rate of a table in LDS.  A sheet `i7+i5` with a separator is then looked up HALF BY HALF: one small table per
// half (its distinct half-barcodes and their one-substitution neighbours -> half id h, distance d) and one table of the
// sheet's (h7, h5) pairs -> (first, last) sample.  An observed barcode is within max_diff of a row iff both halves are
// found, d7 + d5 + (separator differs) <= max_diff and the pair is a row of the sheet.  That is the reference's loop
// (src/fasta_demultiplex.rs:154-194) exactly when every half key lies within distance 1 of ONE half-barcode only — the
// host checks it while it enumerates (half-barcodes at distance >= 3 of each other always pass) and builds the full-key
// table otherwise.  Entries are 8 bytes: half tables {key word, h | d << 16}, pair table {h7 | h5 << 10, first | last << 16}.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>

#if defined(__HIPCC__)
#define SK_HD __host__ __device__
#else
#define SK_HD
#endif

namespace sk {

constexpr int kLutMaxLen = 20;           // columns the packing holds
constexpr int kLutMaxSamples = 1021;     // idx is 7 or 10 bits; S + 3 counters fit the kernels' LDS histogram (kMaxLdsHist)
constexpr int kLutMinBits = 9;           // tag + idx + flag must fit 32 bits: nb >= idx bits + 2
constexpr uint32_t kLutFree = 0x00000080u;
constexpr uint32_t kLutPairFree = 0xFFFFFFFFu;

// the factored form: three cuckoo tables of 8-byte entries in one blob (half 1, half 2, pairs), each 2 x 2^nb slots
struct LutPairDev {
	const uint32_t *tab;       // or nullptr: the sheet is served by the full-key table (or by none)
	int nb1, nb2, nbp;         // slot bits
	uint32_t off2, offp;       // where the second half's table and the pair table begin, in entries
	uint32_t seed1, seed2, seedp;
	uint32_t keep1, keep2;     // class bits of the counting columns of each half's packed word
	int bytes;                 // of the blob
};

struct LutDev {
	const uint32_t *tab;       // 2 tables x (mask + 1) slots x 2 dwords, or nullptr: the sheet has no table
	const int16_t *amb;        // (first, last) pairs of the ambiguous keys
	int W1, W2;                // key dwords: the row's first W1 dwords, and (with a separator) W2 dwords from the byte after it
	int nb, mask;              // slot bits, slots per table - 1
	uint32_t seed, tag_mask;   // tag_mask = (1 << (33 - nb)) - 1
	int sh;                    // byte -> letter index: (b >> sh) & 7
	uint32_t tab_lo, tab_hi;   // the letters by index (v_perm table); an unused index holds a byte with another index
	uint32_t other;            // the class of "a byte the sheet never uses", in every byte
	uint32_t keepA, keepB;     // class bits of the counting columns in the packed words
	int sep_off;               // separator: its byte offset in the row (-1 = none; then W2 == 0) ...
	uint32_t sep_val;          // ... and its letter
	int max_diff;              // 0 or 1
	int idx_shift;             // w1's idx field: 24 (7 bits) or 21 (10 bits) ...
	uint32_t idx_mask;         // ... and its mask
	LutPairDev pair;           // the factored form (then tab above is nullptr)
};

// classes of 4 consecutive columns (one per byte, 3 bits each) x 5 dwords -> two words; no two fields overlap
SK_HD inline void lut_pack(const uint32_t (&c)[5], uint32_t &A, uint32_t &B)
{
	A = c[0] | (c[1] << 3) | ((c[4] & 0x03030303u) << 6);
	B = c[2] | (c[3] << 3) | ((c[4] & 0x04040404u) << 4);
}

SK_HD inline uint32_t lut_mix(uint32_t A, uint32_t B, uint32_t seed)
{
	uint32_t t = B + (B << 10);
	t ^= t >> 6;
	uint32_t x = (A ^ t ^ seed) * 0x85EBCA6Bu;     // odd multipliers and an xor-shift: every step is invertible,
	x ^= x >> 15;                                  // (A, B) -> (x, B) is a bijection; the TOP bits of x are the best mixed
	return x * 0x9E3779B1u;
}
// Table 1 is indexed by the top nb bits of x, table 2 by the nb bits below them: side 2 looks at x rotated left by nb, and for
// either side v the slot is v's top nb bits and the tag its low 32 - nb bits (nb in [1, 31]).
// v_mul_lo_u32 issues at the rate of an add on gfx950 (tools/micro/mulrate_exp.hip), so this mix is 5 instructions where the
// shift-and-add mix of rounds 2-3 was 9, and the tag compare needs no shift: 6 of the 63 VALU instructions the 8-column
// lookup spent per row, in a kernel bound by instruction issue.  Tables come out the same size or smaller
// (40 random sheets of each of 8 shapes; a single multiply without the xor-shift made them 8 x larger).
SK_HD inline uint32_t lut_side2(uint32_t x, int nb)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_amdgcn_alignbit(x, x, (uint32_t)(32 - nb));      // one instruction where shift, shift, or were three
#else
	return (x << nb) | (x >> (32 - nb));
#endif
}
SK_HD inline uint32_t lut_slot(uint32_t v, int nb) { return v >> (32 - nb); }

// one half's key word: the classes of its (at most 8) columns, dwords 0 and 1 of the half interleaved as in lut_pack
SK_HD inline uint32_t lut_pack_half(uint32_t c0, uint32_t c1) { return c0 | (c1 << 3); }

// Host side: what sk_set_barcodes' sheet becomes.  false = this sheet has no table (the matchers serve it).
// A table of more than lds_budget bytes is served from L2; a sheet with a separator then gets the factored form instead when
// its half tables exist and fit (lds_budget 0: never factored).
struct LutHost {
	LutDev dev{};                        // tab / amb / pair.tab are null here
	std::vector<uint32_t> slots;         // the two tables (full-key form) or the blob of three (factored form: dev.pair.bytes != 0)
	std::vector<int16_t> amb;            // pairs
	size_t n_keys = 0;
};
bool lut_build(const uint8_t *sheet, int S, int L, int max_diff, LutHost &out, int lds_budget = 128 << 10);

}  // namespace sk
