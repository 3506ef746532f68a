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
// Wide classes.  A sheet with 8 to 15 letters in its counting columns (one typed partly in lower case: the reference compares raw
// bytes, src/fasta_demultiplex.rs:273-274) gets 4-bit classes when some 3-bit window of a byte plus ONE more bit — a function g
// of another 3-bit window — tells its letters apart: class = (b >> sh) & 7 | g[(b >> sh2) & 7] << 3.  Eight columns then fill a
// word, so the forms are: at most 8 columns and no separator (the full-key table, B == 0 for every key), or two halves of at most
// 8 columns beside a separator in the factored form; any other shape keeps the matchers.
//
// The factored form (LutDev::pair): 384 dual-index samples x (16 x 4 + 1) keys are 25 k entries — 512 KiB, served from L2 at
// half the rate of a table in LDS.  A sheet `i7+i5` with a separator is then looked up HALF BY HALF: one small table per
// half (its distinct half-barcodes and their one-substitution neighbours -> half id h, distance d) and one table of the
// sheet's (h7, h5) pairs -> (first, last) sample.  An observed barcode is within max_diff of a row iff both halves are
// found, d7 + d5 + (separator differs) <= max_diff and the pair is a row of the sheet.  That is the reference's loop
// (src/fasta_demultiplex.rs:154-194) exactly when every half key lies within distance 1 of ONE half-barcode only — the
// host checks it while it enumerates (half-barcodes at distance >= 3 of each other always pass) and builds the full-key
// table otherwise.  Entries are 8 bytes: half tables {key word, h | d << 16}, pair table {h7 | h5 << 10, first | last << 16}.
// A free slot of a half table holds {a word that is no key of the table, d = 0xFFFF}: matching it is a miss (tot > max_diff).
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
constexpr uint32_t kLutPairFree = 0xFFFFFFFFu;     // a free slot's key word in the factored form (the builder steps down from here past any real key)
constexpr uint32_t kLutPairFreeVal = 0xFFFF0000u;  // ... and its value word in the two half tables: distance 65 535, so a half that equals the free word is a miss

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
	// wide classes (8 to 15 letters: a sheet typed in both cases; sk_lut.h "Wide classes"): the class is 4 bits, its top bit
	// g[(b >> sh2) & 7]; classes 8 ... 15 have their letters in tab2; the key is one word (at most 8 columns per word)
	int wide, sh2;
	uint32_t g_lo, g_hi;       // eight bytes, 0x00 or 0x08
	uint32_t tab2_lo, tab2_hi;
	uint32_t keepA, keepB;     // class bits of the counting columns in the packed words
	int sep_off;               // separator: its byte offset in the row (-1 = none; then W2 == 0) ...
	uint32_t sep_val;          // ... and its letter
	int max_diff;              // 0 or 1
	int idx_shift;             // w1's idx field: 24 (7 bits) or 21 (10 bits) ...
	uint32_t idx_mask;         // ... and its mask
	LutPairDev pair;           // the factored form (then tab above is nullptr)
};

// classes of 4 consecutive columns (one per byte, 3 bits each) x 5 dwords -> two words; no two fields overlap.
// Wide classes (4 bits): eight columns at most, one word.
SK_HD inline void lut_pack(const uint32_t (&c)[5], uint32_t &A, uint32_t &B, int wide = 0)
{
	if (wide) { A = c[0] | (c[1] << 4); B = 0u; return; }
	A = c[0] | (c[1] << 3) | ((c[4] & 0x03030303u) << 6);
	B = c[2] | (c[3] << 3) | ((c[4] & 0x04040404u) << 4);
}

// eight bytes (hi : lo) selected by the four bytes of sel, each 0 ... 7 (v_perm_b32)
SK_HD inline uint32_t lut_perm8(uint32_t hi, uint32_t lo, uint32_t sel)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_amdgcn_perm(hi, lo, sel);
#else
	uint32_t out = 0;
	for (int j = 0; j < 4; j++) {
		const uint32_t q = (sel >> (8 * j)) & 0xffu;
		out |= ((q < 4 ? lo >> (8 * q) : hi >> (8 * (q - 4))) & 0xffu) << (8 * j);
	}
	return out;
#endif
}
// 0xff in every byte of df that is not zero.  On the device one v_perm_b32 whose selector rule does the work (12 -> 0x00, 13
// and above -> 0xff): (df & 0x7f) + 12 is 12 for a byte in {0, 0x80} and 13 ... 0x8b otherwise (no carry leaves the byte);
// or-ing df back in lifts 0x80 to 0x8c.
SK_HD inline uint32_t lut_ne_mask(uint32_t df)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_amdgcn_perm(0u, 0u, ((df & 0x7f7f7f7fu) + 0x0c0c0c0cu) | df);
#else
	const uint32_t nz = (((df & 0x7f7f7f7fu) + 0x7f7f7f7fu) | df) & 0x80808080u;
	return nz - (nz >> 7) + nz;                     // 0x80 -> 0xff per byte
#endif
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
SK_HD inline uint32_t lut_pack_half(uint32_t c0, uint32_t c1, int wide = 0) { return c0 | (c1 << (wide ? 4 : 3)); }

// The class of each of a dword's four bytes: the index of the sheet letter the byte equals, `other` for a byte the sheet never
// uses.  8 instructions on the device (16 with wide classes); the kernels and the CPU model of tests/cpp/lut_test.cpp share it.
SK_HD inline uint32_t lut_classes(uint32_t dw, const LutDev &t)
{
	const uint32_t sel = (dw >> t.sh) & 0x07070707u;
	uint32_t letter = lut_perm8(t.tab_hi, t.tab_lo, sel), cls = sel;
	if (t.wide) {
		const uint32_t g = lut_perm8(t.g_hi, t.g_lo, (dw >> t.sh2) & 0x07070707u);      // 0x00 or 0x08 per byte
		const uint32_t m8 = (g >> 3) * 0xffu;
		letter = (lut_perm8(t.tab2_hi, t.tab2_lo, sel) & m8) | (letter & ~m8);
		cls = sel | g;
	}
	const uint32_t m = lut_ne_mask(dw ^ letter);                                       // 0xff in every byte that is not its candidate letter
	return (m & t.other) | (~m & cls);
}

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
