// sk_inflate.hip — B1 on the device: BGZF blocks inflated by the GPU, BAM records walked there.
//
// Reference behaviour served: src/common.rs:121-157 (BamReader: htslib inflates every BGZF block of the file, checks its
// CRC-32 and hands out records one by one), in front of src/sam_statistics.rs:63-69 and src/sam_fragment_lengths.rs:29-43.
// Through round 5 the hosts inflated on CPU threads: `sam statistics` on a 32 M-record BAM was 14 CPU-seconds of inflate for
// 0.45 ms of kernel (profiles/r05_bam_host.txt).  A BGZF block (SAMv1 §4.1) is a gzip member of at most 64 KiB whose DEFLATE
// stream (RFC 1951) depends on nothing outside it, and its header says how long it is: the file is thousands of independent
// streams, and the compressed bytes are a third of what crosses PCIe when the host inflates.
//
// bgzf_inflate_kernel — ONE WAVE PER BLOCK.  DEFLATE decoding is a chain: a symbol's first bit is known only when the one
// before it has been decoded.  So a wave decodes like one thread — the decoder's state (bit buffer, positions) is wave-uniform
// and lives in scalar registers — and brings its 64 lanes to bear on what IS parallel.  Throughput comes from thousands of
// waves decoding their own blocks at once (16 per CU), and — since the scalar unit is one per CU — from keeping the scalar
// instructions per symbol few.
//   * input: 64 dwords of the compressed stream sit in ONE vector register (lane i: dword i of the window), the next window
//     in a second one, loaded with one coalesced raw-buffer load each (clipped by its descriptor: beyond the payload it
//     reads zeros); the bit buffer takes its next dword with v_readlane — no memory access per symbol on the input side.
//   * tables (LDS, per wave): the literal/length code resolved by its first 10 bits, the distance code by its first 8;
//     longer codes (rare by construction) are walked bit by bit against the canonical code's first-code / count arrays.  A
//     block's tables are built by the whole wave: ranks of the symbols within their code length by ballots, then every lane
//     resolves the table indices it owns by the canonical rule.
//   * symbols, a GROUP per table access (inf_symbols_groups): lane j decodes the symbol that would begin at bit j of the
//     buffer — a literal, or a whole match: length code, extra bits, distance code, extra bits — and a short scalar chain
//     (v_readlane of the symbol's bit count at the offset the symbol before ended at) finds which lanes are real symbol
//     starts: one LDS latency per ~6 symbols (the buffer holds up to 128 bits) instead of one (or two) per symbol, 8 scalar
//     instructions a symbol.  The symbols'
//     places in the output are a prefix sum (DPP) of the bytes they make; literals are stored by their lanes.
//   * matches, a BATCH per copy: decoding does not need a match's bytes, so matches are noted as tokens while decoding goes
//     on (up to 64, within one unit of the ring) and copied afterwards — those whose source was flushed long ago all at
//     once (a lane, or eight, per match: ONE trip to memory per batch where the first form of this kernel made one per
//     match and sat it out: four matches in five of a literal-heavy BAM block reach further back than the ring), the others in
//     order through the ring.
//   * output: the wave keeps the last 1-1.5 KiB it produced in an LDS ring addressed by the OUTPUT address (mod the ring), four
//     units of 512 bytes: the one being written, two of history, one free for a group's literals to run into; a unit that is
//     complete leaves for HBM as coalesced 16-byte stores (a 16-byte-aligned piece of the output, by the addressing).
// Anything irregular — a code that is over-subscribed or incomplete, a distance before the block's first byte, output or
// input that ends early or late — gives the block a non-zero status, and the caller inflates THAT block on the CPU with
// zlib, whose verdict stands: the device never decides that a file is corrupt.
//
// bgzf_crc_kernel — CRC-32 of every inflated block (RFC 1952), a wave per block: each lane the CRC of its 1/64, combined
// with carry-less multiplication by x^(8 n) mod P.
//
// bam_walk_kernel — the records of the inflated stream.  Record i + 1 begins where record i's block_size says: a chain
// through the whole file.  It is cut at the BGZF blocks: every block is walked by ONE LANE from a GUESSED entry offset
// (htslib never splits a record across blocks when it fits one, so the guess "at the block's first byte" is right for the
// files it wrote), and yields the offset at which the walk leaves the block.  The guesses are then VERIFIED: block c + 1's
// entry must equal block c's exit, and block 0's the end of the header; blocks whose entry was wrong take their
// predecessor's exit and walk again (bam_walk_fix_kernel).  When every entry equals its predecessor's exit the chain is
// THE chain — by induction from the header, not by heuristics — and a last pass reduces flag / refID / next_refID / tlen of
// every record straight from the inflated bytes (bam_record of sk_kernels.hip: the same predicate as the SoA kernel).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <cstring>

#include "sk_internal.h"

namespace sk {

typedef uint32_t u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

constexpr int kInfWaves = 4;             // waves per workgroup; each inflates blocks of its own, no barrier anywhere
constexpr int kInfLitBits = 10, kInfDistBits = 8, kInfPreBits = 7;
#ifndef SK_INF_RING
#define SK_INF_RING 2048
#endif
// build knobs (A/B builds only): the second form of the symbol loop (groups and batches), its group loop in assembly, nt stores for the flush
#ifndef SK_INF_GROUPS
#define SK_INF_GROUPS 1
#endif
#ifndef SK_INF_ASM_GROUPS
#define SK_INF_ASM_GROUPS 1
#endif
#ifndef SK_INF_FLUSH_NT
#define SK_INF_FLUSH_NT 1
#endif
constexpr u32 kRing = SK_INF_RING, kHalf = kRing / 2;
static_assert((kRing & (kRing - 1)) == 0 && kHalf >= 1024, "ring: a power of two, a half holds a longest match");

// table entry: bits 0-3 the code's length, 4-7 extra bits, 8-10 kind, 16-31 literal / base value
enum : u32 { kKindLiteral = 0u, kKindBase = 1u, kKindEob = 2u, kKindLong = 3u, kKindBad = 4u };
__device__ __forceinline__ u32 inf_entry(u32 kind, u32 len, u32 extra, u32 value) { return len | (extra << 4) | (kind << 8) | (value << 16); }

struct InfLds {                          // one wave's
	u32 lit[1 << kInfLitBits];
	u32 dist[1 << kInfDistBits];
	uint16_t pre[1 << kInfPreBits];      // code-length code: length | symbol << 4
	uint16_t sorted[288 + 32];           // the symbols by (code length, symbol): literal/length alphabet, then distances
	uint16_t first[2][16], offs[2][16], count[2][16];      // per alphabet and code length: first code, where its symbols begin in `sorted`, how many
	uint8_t lens[320];
	uint8_t ring[kRing] __attribute__((aligned(16)));
};

__device__ const uint16_t kInfLenBase[32] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258, 0, 0, 0};
__device__ const uint8_t kInfLenExtra[32] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0, 0, 0, 0};
__device__ const uint16_t kInfDistBase[32] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577, 0, 0};
__device__ const uint8_t kInfDistExtra[32] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 0, 0};
__device__ const uint8_t kInfPreOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// what a symbol of alphabet `which` (0 literal/length, 1 distance, 2 code lengths) decodes to, its code being `len` bits
__device__ __forceinline__ u32 inf_symbol_entry(int which, u32 sym, u32 len)
{
	if (which == 0) {
		if (sym < 256u) return inf_entry(kKindLiteral, len, 0u, sym);
		if (sym == 256u) return inf_entry(kKindEob, len, 0u, 0u);
		if (sym <= 285u) return inf_entry(kKindBase, len, kInfLenExtra[sym - 257u], kInfLenBase[sym - 257u]);
		return inf_entry(kKindBad, len, 0u, 0u);                       // 286, 287: in the fixed code, never valid in data
	}
	if (which == 1) return sym < 30u ? inf_entry(kKindBase, len, kInfDistExtra[sym], kInfDistBase[sym]) : inf_entry(kKindBad, len, 0u, 0u);
	return len | (sym << 4);
}

__device__ __forceinline__ void inf_lds_fence()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ u32 inf_uniform(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }

// Build the decoding table of one alphabet from its code lengths lens[0 .. nsym) (LDS), by the whole wave.
// P: index bits of the table; MAXL: longest code of the alphabet.  Returns 0, or 3 for a set of lengths that is no prefix code
// (over-subscribed; incomplete other than the two cases DEFLATE's writers produce: no distance code at all, one distance code).
template <int WHICH, int P, int MAXL>
__device__ __forceinline__ u32 inf_build(InfLds &L, const uint8_t *lens, int nsym, int lane)
{
	const int ngroups = (nsym + 63) >> 6;
	u32 cnt[MAXL + 1];
#pragma unroll
	for (int l = 0; l <= MAXL; l++) cnt[l] = 0u;
	for (int g = 0; g < ngroups; g++) {
		const int s = g * 64 + lane;
		const u32 ls = s < nsym ? lens[s] : 0u;
#pragma unroll
		for (int l = 1; l <= MAXL; l++) cnt[l] += (u32)__popcll(__ballot(ls == (u32)l));
	}
	u32 total = 0u, maxlen = 0u;
	int left = 1;
	bool over = false;
#pragma unroll
	for (int l = 1; l <= MAXL; l++) {
		left = (left << 1) - (int)cnt[l];
		over = over || left < 0;
		total += cnt[l];
		if (cnt[l]) maxlen = (u32)l;
	}
	if (over) return 3u;
	if (left != 0) {
		// incomplete: legal for the distance alphabet when it has no code (a block of literals) or one code of one bit
		if (!(WHICH == 1 && (total == 0u || (total == 1u && cnt[1] == 1u)))) return 3u;
	}
	u32 first[MAXL + 2], offs[MAXL + 2];
	{
		u32 code = 0u, o = 0u;
		first[0] = 0u; offs[0] = 0u;
#pragma unroll
		for (int l = 1; l <= MAXL; l++) {
			code = (code + (l > 1 ? cnt[l - 1] : 0u)) << 1;
			first[l] = code;
			offs[l] = o;
			o += cnt[l];
		}
	}
	constexpr int A = WHICH == 1 ? 1 : 0;                              // (the code-length alphabet has no long codes: nothing of it is kept)
	constexpr u32 sbase = WHICH == 1 ? 288u : 0u;
	uint16_t *const sorted = WHICH == 2 ? reinterpret_cast<uint16_t *>(L.dist) : L.sorted + sbase;      // (the code-length alphabet borrows the distance table's room: it is built later)
	if (WHICH != 2) {
#pragma unroll
		for (int l = 1; l <= MAXL; l++)
			if (lane == l) { L.first[A][l] = (uint16_t)first[l]; L.offs[A][l] = (uint16_t)offs[l]; L.count[A][l] = (uint16_t)cnt[l]; }
	}
	// the symbols in the order of the canonical code: by length, then by symbol
	u32 run[MAXL + 1];
#pragma unroll
	for (int l = 0; l <= MAXL; l++) run[l] = 0u;
	for (int g = 0; g < ngroups; g++) {
		const int s = g * 64 + lane;
		const u32 ls = s < nsym ? lens[s] : 0u;
#pragma unroll
		for (int l = 1; l <= MAXL; l++) {
			const u64 m = __ballot(ls == (u32)l);
			if (ls == (u32)l) sorted[offs[l] + run[l] + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u))] = (uint16_t)s;
			run[l] += (u32)__popcll(m);
		}
	}
	inf_lds_fence();
	// every index of the table: the first l for which its first l bits (the code, most significant bit first) are a code of length l
	for (int i0 = 0; i0 < (1 << P); i0 += 64) {
		const u32 i = (u32)(i0 + lane);
		u32 code = 0u, e = maxlen > (u32)P ? inf_entry(kKindLong, 0u, 0u, 0u) : inf_entry(kKindBad, 0u, 0u, 0u);
		bool found = false;
#pragma unroll
		for (int l = 1; l <= (MAXL < P ? MAXL : P); l++) {
			code = (code << 1) | ((i >> (l - 1)) & 1u);
			const u32 idx = code - first[l];
			if (!found && idx < cnt[l]) {
				found = true;
				e = inf_symbol_entry(WHICH, sorted[offs[l] + idx], (u32)l);
			}
		}
		if (WHICH == 0) L.lit[i] = e;
		else if (WHICH == 1) L.dist[i] = e;
		else L.pre[i] = (uint16_t)(found ? e : 0u);
	}
	inf_lds_fence();
	return 0u;
}

// the bit reader: everything wave-uniform except the two windows
struct InfBits {
	u64 bb;                  // bits not yet consumed, the next one lowest
	u64 bx;                  // ... and bits 64 and up: the assembly's group loop keeps up to 128 (everything else refills to 64 at most and finds bx zero, or shifts it along)
	u32 cnt;                 // how many
	u32 vin, vnext;          // (vector) the window the buffer is fed from, and the one behind it
	u32 widx;                // the next dword of vin
	u32 next_off;            // byte offset of the window after vnext
	u32 taken;               // dwords taken out of the windows
	__amdgpu_buffer_rsrc_t rs;
};
__device__ __forceinline__ void inf_refill(InfBits &b, int lane)
{
	if (b.cnt <= 32u) {
		const u32 w = (u32)__builtin_amdgcn_readlane((int)b.vin, (int)b.widx);
		b.bb |= (u64)w << b.cnt;
		b.cnt += 32u;
		b.widx++;
		b.taken++;
		if (b.widx == 64u) {
			b.vin = b.vnext;
			b.vnext = __builtin_amdgcn_raw_buffer_load_b32(b.rs, (int)(b.next_off + 4u * (u32)lane), 0, 0);
			b.next_off += 256u;
			b.widx = 0u;
		}
	}
}
__device__ __forceinline__ u32 inf_take(InfBits &b, u32 n)
{
	const u32 v = (u32)b.bb & ((1u << n) - 1u);
	if (n != 0u) {                                                      // (n <= 16 here)
		b.bb = (b.bb >> n) | (b.bx << (64u - n));
		b.bx >>= n;
		b.cnt -= n;
	}
	return v;
}
// position the reader at byte `at` of the descriptor's range
__device__ __forceinline__ void inf_seek(InfBits &b, u32 at, int lane)
{
	const u32 a4 = at & ~3u;
	b.vin = __builtin_amdgcn_raw_buffer_load_b32(b.rs, (int)(a4 + 4u * (u32)lane), 0, 0);
	b.vnext = __builtin_amdgcn_raw_buffer_load_b32(b.rs, (int)(a4 + 256u + 4u * (u32)lane), 0, 0);
	b.next_off = a4 + 512u;
	b.widx = 0u;
	b.bb = 0ull;
	b.bx = 0ull;
	b.cnt = 0u;
	b.taken = a4 >> 2;
	inf_refill(b, lane);
	const u32 skip = 8u * (at & 3u);
	b.bb >>= skip;
	b.cnt -= skip;
	inf_refill(b, lane);
}
// bytes of the range consumed so far (rounded up to the byte the next bit lies in)
__device__ __forceinline__ u32 inf_consumed(const InfBits &b) { return (b.taken * 32u - b.cnt + 7u) >> 3; }

// a code longer than the table's index bits, walked bit by bit against the canonical code (returns kKindBad's entry when no code matches)
template <int WHICH, int P>
__device__ __forceinline__ u32 inf_long_code(const InfLds &L, u64 bb)
{
	constexpr int A = WHICH == 1 ? 1 : 0;
	constexpr u32 sbase = WHICH == 1 ? 288u : 0u;
	u32 code = 0u;
#pragma unroll
	for (int l = 1; l <= P; l++) code = (code << 1) | ((u32)(bb >> (l - 1)) & 1u);
	for (int l = P + 1; l <= 15; l++) {
		code = (code << 1) | ((u32)(bb >> (l - 1)) & 1u);
		const u32 idx = code - inf_uniform(L.first[A][l]);
		if (idx < inf_uniform(L.count[A][l])) return inf_symbol_entry(WHICH, inf_uniform(L.sorted[sbase + inf_uniform(L.offs[A][l]) + idx]), (u32)l);
	}
	return inf_entry(kKindBad, 1u, 0u, 0u);
}

// dst[lo, hi) (dst = where the block's output begins; positions within the block) leaves the ring for HBM.  a0 = the block's first
// output address modulo 2^32.
__device__ __forceinline__ void inf_flush(const InfLds &L, uint8_t *dst, u32 a0, u32 lo, u32 hi, int lane)
{
	if (hi <= lo) return;
	inf_lds_fence();
	const u32 alo = a0 + lo, ahi = a0 + hi;                            // output addresses (mod 2^32: only their low bits are used, and differences)
	u32 body_lo = (alo + 15u) & ~15u, body_hi = ahi & ~15u;
	if (body_hi < body_lo) { body_lo = ahi; body_hi = ahi; }           // no aligned 16 bytes inside: all head
	// head and tail: byte by byte (at most 15 + 15)
	const u32 nhead = body_lo - alo, ntail = ahi - body_hi;
	if ((u32)lane < nhead) dst[lo + (u32)lane] = L.ring[(alo + (u32)lane) & (kRing - 1u)];
	if ((u32)lane < ntail) dst[hi - ntail + (u32)lane] = L.ring[(body_hi + (u32)lane) & (kRing - 1u)];
	for (u32 a = body_lo + 16u * (u32)lane; a < body_hi; a += 1024u) {
		const u32x4 v = *reinterpret_cast<const u32x4 *>(&L.ring[a & (kRing - 1u)]);
#if SK_INF_FLUSH_NT
		__builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(dst + (lo + (a - alo))));
#else
		*reinterpret_cast<u32x4 *>(dst + (lo + (a - alo))) = v;
#endif
	}
}

extern __shared__ __attribute__((aligned(16))) uint8_t inf_smem[];      // kInfWaves x InfLds

// The symbols of one DEFLATE block: the decoder's inner loop, kept out of line.  Inlined into inf_block it shared the scalar
// registers with everything that function keeps alive across it (the block table, the header's fields, three table builders):
// the allocator kept the bit buffer's count, the window index and the output position in lanes of a spill VGPR and moved them
// in and out with v_readlane / v_writelane on every symbol (a 64 KiB block of literals took 1 300 cycles per symbol).  Arguments
// of a device function arrive in vector registers: the state is made scalar again on entry (it is wave-uniform by construction).
// what crosses the call: plain words (the buffer descriptor is made again inside: a value of its type does not travel through a call's registers)
struct InfRun {
	u64 bb, bx;
	u32 cnt, vin, vnext, widx, next_off, taken, op, flushed, err;
	__device__ __forceinline__ void put(const InfBits &b) { bb = b.bb; bx = b.bx; cnt = b.cnt; vin = b.vin; vnext = b.vnext; widx = b.widx; next_off = b.next_off; taken = b.taken; }
	__device__ __forceinline__ void get(InfBits &b) const { b.bb = bb; b.bx = bx; b.cnt = cnt; b.vin = vin; b.vnext = vnext; b.widx = widx; b.next_off = next_off; b.taken = taken; }
};
__device__ __forceinline__ u64 inf_uniform64(u64 v) { return (u64)inf_uniform((u32)v) | ((u64)inf_uniform((u32)(v >> 32)) << 32); }
__device__ __attribute__((noinline)) InfRun inf_symbols(int wave, InfRun r, const uint8_t *in_base, u32 in_range, u32 a0, u32 out_len, uint8_t *dst, int lane)
{
	InfLds &L = reinterpret_cast<InfLds *>(inf_smem)[inf_uniform((u32)wave)];      // (by index: a pointer argument would arrive as a generic address, and every table read as a flat load)
	InfBits b;
	r.get(b);
	b.bb = inf_uniform64(b.bb); b.bx = inf_uniform64(b.bx); b.cnt = inf_uniform(b.cnt); b.widx = inf_uniform(b.widx); b.next_off = inf_uniform(b.next_off); b.taken = inf_uniform(b.taken);
	in_base = reinterpret_cast<const uint8_t *>((uintptr_t)inf_uniform64((u64)(uintptr_t)in_base));
	b.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(in_base), 0, (int)inf_uniform(in_range), 0x00020000);      // (the descriptor, made again from scalars)
	u32 op = inf_uniform(r.op), flushed = inf_uniform(r.flushed), err = 0u;
	a0 = inf_uniform(a0); out_len = inf_uniform(out_len);
	dst = reinterpret_cast<uint8_t *>((uintptr_t)inf_uniform64((u64)(uintptr_t)dst));
	for (;;) {
		inf_refill(b, lane);
		u32 e = inf_uniform(L.lit[(u32)b.bb & ((1u << kInfLitBits) - 1u)]);
		if (((e >> 8) & 7u) == kKindLong) e = inf_long_code<0, kInfLitBits>(L, b.bb);
		const u32 kind = (e >> 8) & 7u;
		inf_take(b, e & 15u);
		if (kind == kKindLiteral) {
			if (op >= out_len) { err = 6u; break; }
			if (lane == 0) L.ring[(a0 + op) & (kRing - 1u)] = (uint8_t)(e >> 16);
			op++;
			if (((a0 + op) & (kHalf - 1u)) == 0u) { inf_flush(L, dst, a0, flushed, op, lane); flushed = op; }
			continue;
		}
		if (kind == kKindEob) break;
		if (kind != kKindBase) { err = 4u; break; }
		const u32 len = (e >> 16) + inf_take(b, (e >> 4) & 15u);
		inf_refill(b, lane);
		u32 d = inf_uniform(L.dist[(u32)b.bb & ((1u << kInfDistBits) - 1u)]);
		if (((d >> 8) & 7u) == kKindLong) d = inf_long_code<1, kInfDistBits>(L, b.bb);
		if (((d >> 8) & 7u) != kKindBase) { err = 4u; break; }
		inf_take(b, d & 15u);
		const u32 dist = (d >> 16) + inf_take(b, (d >> 4) & 15u);
		if (dist > op) { err = 5u; break; }
		if (len > out_len - op) { err = 6u; break; }
		// the copy, cut where the output crosses a half of the ring
		const u32 m0 = op;                                             // where the match begins
		for (u32 done = 0u; done < len;) {
			const u32 room = kHalf - ((a0 + op) & (kHalf - 1u));
			const u32 seg = min(len - done, room);
			// what the ring holds: this half and the one before it (and nothing before the block's first byte)
			const u32 half_base = (a0 + op) & ~(kHalf - 1u);
			const u32 floor_addr = half_base - kHalf;
			inf_lds_fence();
			for (u32 k = (u32)lane; k < seg; k += 64u) {
				const u32 j = done + k;                                   // byte j of the match
				const u32 sj = dist >= len ? j : j % dist;               // (an overlapping match repeats its first dist bytes)
				const u32 sp = m0 - dist + sj;                            // source position within the block
				const u32 sa = a0 + sp;
				uint8_t v;
				// in the ring <=> its address is at or above the floor; addresses wrap at 2^32, differences do not (a block is 64 KiB)
				if ((int)(sa - floor_addr) >= 0) v = L.ring[sa & (kRing - 1u)];
				else v = __hip_atomic_load(dst + sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // flushed long ago: from L2, past the vector cache
				L.ring[(a0 + op + k) & (kRing - 1u)] = v;
			}
			op += seg; done += seg;
			if (((a0 + op) & (kHalf - 1u)) == 0u) { inf_flush(L, dst, a0, flushed, op, lane); flushed = op; }
		}
	}
	r.put(b); r.op = op; r.flushed = flushed; r.err = err;
	return r;
}

// The symbols of one DEFLATE block (round 6, second form): GROUPS of symbols per table access, BATCHES of matches per copy.
//
// What the first form (inf_symbols, kept for -DSK_INF_GROUPS=0) spent its time on, measured: a block of BAM records with uniformly
// drawn bases and qualities is 26 500 literals and 9 200 matches of 4 bytes on average, and four matches in five reach further back
// than the 1-2 KiB the ring holds: each of those was a load from memory (~2 us) that the wave sat out — 7 300 of them are the
// block's 14 ms.  And decoding one symbol per table access is a chain of latencies of its own (LDS read, v_readfirstlane, scalar
// arithmetic, the byte's ds_write: a hand-written loop of 13 scalar instructions instead of the compiler's 45 gained 16 %).
//   * groups: the tables are read ONCE for every bit offset of the buffer — lane j looks up the 10 bits at offset j, and the distance
//     table's 8 bits there — and the symbols of the buffer's bits (33-64 of them here, 97-128 in the assembly's loop) follow from registers (v_readlane at the offset the symbol
//     before ended at), no memory access in between.  A code is taken only if all its bits are in the buffer (o + length <= cnt): a
//     lane that looked at bits beyond the buffer's end is never believed.  The literals of a group are written together: the lanes
//     at which a literal began (a mask built by the chain) store their bytes at ring positions given by their rank in the mask.  The
//     chain is gfx950 assembly: 1 vector + 5 scalar instructions and two branches a literal.
//   * batches: decoding does not need the matches' bytes, so a match is only NOTED (position, length, distance: lane k of three
//     registers holds token k) while decoding goes on, up to 64 of them within one unit of the ring.  Then the matches whose source was
//     flushed long ago — final bytes, no dependence on anything pending — are copied all at once, a lane per match: one trip to
//     memory per batch instead of one per match; the others follow in order through the ring.
// The ring is four units of kRing / 4: the unit being written, two of history (a match source is in the ring iff it is not below the
// current unit's base - 2 units), and one that is free, so that a group's literals may run past the unit's end (by at most 63 bytes)
// while matches of the batch are still pending.
// -DSK_INF_STAMPS (diagnostic builds only, tools/r06/inflate_stamps.py): shader cycles (s_memtime) per phase, summed over all waves
#ifdef SK_INF_STAMPS
__device__ u64 g_inf_stamps[16];
#define SK_ISTAMP(i) do { const u64 now_ = __builtin_amdgcn_s_memtime(); ist_acc[i] += now_ - ist_last; ist_last = now_; } while (0)
#else
#define SK_ISTAMP(i) do { } while (0)
#endif
constexpr u32 kUnit = kRing / 4u;
static_assert(kUnit >= 512u, "a unit holds a longest match and a group's literals");
__device__ __forceinline__ u64 inf_shr(u64 v, u32 n) { return n >= 64u ? 0ull : v >> n; }      // (n may be all 64 bits: the hardware would shift modulo 64)
__device__ __forceinline__ void inf_drop(InfBits &b, u32 n)          // n of the buffer's (up to 128) bits are done with
{
	if (n == 0u) return;
	if (n < 64u) { b.bb = (b.bb >> n) | (b.bx << (64u - n)); b.bx >>= n; }
	else { b.bb = inf_shr(b.bx, n - 64u); b.bx = 0ull; }
	b.cnt -= n;
}
// inclusive prefix sum over the wave's 64 lanes on DPP (gfx9: row_shr 1, 2, 4, 8 inside rows of 16, then row_bcast 15 and 31 across them)
__device__ __forceinline__ u32 inf_wave_scan(u32 x)
{
	x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
	x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
	x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
	x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
	x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
	x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
	return x;
}

__device__ __attribute__((noinline)) InfRun inf_symbols_groups(int wave, InfRun r, const uint8_t *in_base, u32 in_range, u32 a0, u32 out_len, uint8_t *dst, int lane)
{
	InfLds &L = reinterpret_cast<InfLds *>(inf_smem)[inf_uniform((u32)wave)];
	InfBits b;
	r.get(b);
	b.bb = inf_uniform64(b.bb); b.bx = inf_uniform64(b.bx); b.cnt = inf_uniform(b.cnt); b.widx = inf_uniform(b.widx); b.next_off = inf_uniform(b.next_off); b.taken = inf_uniform(b.taken);
	in_base = reinterpret_cast<const uint8_t *>((uintptr_t)inf_uniform64((u64)(uintptr_t)in_base));
	b.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(in_base), 0, (int)inf_uniform(in_range), 0x00020000);
	u32 op = inf_uniform(r.op), flushed = inf_uniform(r.flushed), err = 0u;
	a0 = inf_uniform(a0); out_len = inf_uniform(out_len);
	dst = reinterpret_cast<uint8_t *>((uintptr_t)inf_uniform64((u64)(uintptr_t)dst));
#ifdef SK_INF_STAMPS
	u64 ist_acc[16] = {};
	u64 ist_last = __builtin_amdgcn_s_memtime();
#endif
	// the batch's tokens (position | length << 16, distance) live where the block's header was decoded: nobody reads that now
	u32 *tokA = reinterpret_cast<u32 *>(L.pre), *tokB = reinterpret_cast<u32 *>(L.lens);
	static_assert(sizeof(L.pre) >= 256 && sizeof(L.lens) >= 256, "64 tokens");
	// the output is in another unit of the ring than what was flushed last: the units behind leave
	auto flush_crossed = [&]() {
		const u32 boundary = ((a0 + op) & ~(kUnit - 1u)) - a0;                 // position of the current unit's first byte ("negative" in the block's first unit: nothing crossed)
		if ((int)(boundary - flushed) > 0) { SK_ISTAMP(9); inf_flush(L, dst, a0, flushed, boundary, lane); flushed = boundary; SK_ISTAMP(6); }
	};
	auto unit_end = [&]() { return (((a0 + op) | (kUnit - 1u)) + 1u) - a0; };  // position behind the current unit's last byte
	// bytes [j0, j0 + n) of the match that begins at position m0 (all of them in one unit): from the ring, or from what was flushed
	auto copy_part = [&](u32 m0, u32 len, u32 dist, u32 j0, u32 n) {
		const u32 floor_addr = ((a0 + m0 + j0) & ~(kUnit - 1u)) - 2u * kUnit;
		const u32 src0 = m0 - dist;
		inf_lds_fence();
		if (dist >= len) {
			for (u32 k = (u32)lane; k < n; k += 64u) {
				const u32 sp = src0 + j0 + k, sa = a0 + sp;
				uint8_t v;
				if ((int)(sa - floor_addr) >= 0) v = L.ring[sa & (kRing - 1u)];
				else v = __hip_atomic_load(dst + sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // flushed long ago: from L2, past the vector cache
				L.ring[(a0 + m0 + j0 + k) & (kRing - 1u)] = v;
			}
		} else {                                                           // an overlapping match repeats its first dist bytes (dist < 258: in the ring)
			for (u32 k = (u32)lane; k < n; k += 64u) {
				const u32 sp = src0 + (j0 + k) % dist, sa = a0 + sp;
				uint8_t v;
				if ((int)(sa - floor_addr) >= 0) v = L.ring[sa & (kRing - 1u)];
				else v = __hip_atomic_load(dst + sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				L.ring[(a0 + m0 + j0 + k) & (kRing - 1u)] = v;
			}
		}
	};
	u32 ntok = 0u;
	u32 ue = unit_end();
	auto finish_batch = [&]() {
		SK_ISTAMP(8);
		if (ntok != 0u) {
			inf_lds_fence();
			u32 tpos = 0u, tlen = 0u, tdist = 0u;
			if ((u32)lane < ntok) { const u32 w = tokA[lane]; tpos = w & 0xffffu; tlen = w >> 16; tdist = tokB[lane]; }
			// the matches whose whole source was flushed, all at once: the short ones a lane each (one trip to memory for all of them) ...
			const u32 fl = tpos - tdist + tlen <= flushed ? tlen : 0u;
			if (__builtin_amdgcn_ballot_w64(fl != 0u && fl <= 8u) != 0ull) {
				const uint8_t *src = dst + (tpos - tdist);
				const u32 n = fl <= 8u ? fl : 0u;
				uint8_t v[8];
#pragma unroll
				for (u32 i = 0u; i < 8u; i++) v[i] = i < n ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint8_t)0;
#pragma unroll
				for (u32 i = 0u; i < 8u; i++) if (i < n) L.ring[(a0 + tpos + i) & (kRing - 1u)] = v[i];
			}
			// ... the longer ones eight lanes each, eight matches a round, 32 bytes of each a trip (a lane each, the longest of 64 matches
			// set the number of trips: 13 for reads in position order, where a match is 23 bytes on average and some are 100)
			const u64 longm = __builtin_amdgcn_ballot_w64(fl > 8u);
			if (longm != 0ull) {
				uint8_t *idx = L.lens + 256;                                   // (the tokens' 64 x 4 bytes end there)
				if (fl > 8u) idx[__builtin_amdgcn_mbcnt_hi((u32)(longm >> 32), __builtin_amdgcn_mbcnt_lo((u32)longm, 0u))] = (uint8_t)lane;
				inf_lds_fence();
				const u32 nlong = (u32)__builtin_popcountll(longm);
				for (u32 r0 = 0u; r0 < nlong; r0 += 8u) {
					const u32 r = r0 + ((u32)lane >> 3);
					if (r < nlong) {
						const u32 k = idx[r], w = tokA[k], p = w & 0xffffu, n = w >> 16;
						const uint8_t *src = dst + (p - tokB[k]);
						for (u32 j = (u32)lane & 7u; j < n; j += 32u) {
							uint8_t v[4];
#pragma unroll
							for (u32 i = 0u; i < 4u; i++) v[i] = j + 8u * i < n ? __hip_atomic_load(src + j + 8u * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint8_t)0;
#pragma unroll
							for (u32 i = 0u; i < 4u; i++) if (j + 8u * i < n) L.ring[(a0 + p + j + 8u * i) & (kRing - 1u)] = v[i];
						}
					}
				}
			}
			if (fl != 0u) tlen = 0u;
			SK_ISTAMP(4);
			// the others in order, through the ring
			u64 todo = __builtin_amdgcn_ballot_w64(tlen != 0u);
			while (todo != 0ull) {
				const u32 k = (u32)__builtin_ctzll(todo);
				todo &= todo - 1ull;
				copy_part((u32)__builtin_amdgcn_readlane((int)tpos, (int)k), (u32)__builtin_amdgcn_readlane((int)tlen, (int)k), (u32)__builtin_amdgcn_readlane((int)tdist, (int)k), 0u,
				          (u32)__builtin_amdgcn_readlane((int)tlen, (int)k));
			}
			ntok = 0u;
			SK_ISTAMP(5);
		}
		flush_crossed();
		ue = unit_end();
	};
	// a match of len bytes from dist back, at op: a token of the batch, or — across the unit's end — at once, in two parts
	auto emit_match = [&](u32 len, u32 dist) {
		if (op + len > ue) {
			finish_batch();                                                  // (what is pending first; the unit may have been left by literals: then it is flushed and ue moves on)
		}
		if (op + len > ue) {
			const u32 m0 = op, part = ue - op;
			copy_part(m0, len, dist, 0u, part);
			op += part;
			flush_crossed();
			copy_part(m0, len, dist, part, len - part);
			op += len - part;
			ue = unit_end();
			SK_ISTAMP(7);
		} else {
			if (ntok == 64u) finish_batch();
			if (lane == 0) { tokA[ntok] = op | (len << 16); tokB[ntok] = dist; }
			op += len;
			ntok++;
			if (op >= ue) finish_batch();
		}
	};
	bool done = false;
#if SK_INF_ASM_GROUPS
	// The groups in gfx950 assembly (the C++ statement of the same steps is the #else branch below: as compiled, its scalar state
	// was copied from register to register at every branch — 13 s_mov at the loop's edges alone — and the scalar unit, one per CU for
	// its 16 waves, is what this kernel waits for).  One pass of the loop is one group: refill, the per-lane decode (34 vector
	// instructions, two table reads), the chain of symbol starts, the symbols' places (a prefix sum on DPP), the matches' tokens and
	// the literals' bytes stored, the buffer moved on.  It leaves the loop (why) for what is rare: 1 the window is used up, 2 the
	// symbol at the buffer's first bit is not one this path takes (a long code, the end of the block, a match that ends behind the
	// unit: bit by bit below), 3 the unit is full or the batch is, 4 / 5 a distance or a length the block cannot have.
	static_assert(kRing == 2048u && kInfLitBits == 10 && kInfDistBits == 8, "the assembly below has these as literals");
	static_assert(offsetof(InfLds, dist) - offsetof(InfLds, lit) == 4096 && offsetof(InfLds, lens) - offsetof(InfLds, pre) == 1088, "and these offsets");
	const u32 lit_base = (u32)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)&L.lit[0];
	const u32 tok_base = (u32)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)&L.pre[0];
	const u32 ring_base = (u32)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)&L.ring[0];
	while (!done && !err) {
		u32 why;
		u64 bx_s = inf_uniform64(b.bx);                                    // (the compiler does not take b.bx for uniform on every path into here)
		asm volatile(
			"s_mov_b64 s[94:95], exec\n\t"
			"v_sub_u32 v57, 64, %[vlane]\n\t"                    // (lane 0 never takes part in the shift that uses it: its v62:63 stay zero)
			"v_mov_b32 v62, 0\n\t"
			"v_mov_b32 v63, 0\n"
			"1:\n\t"                                          // the buffer: s84:85 bits 0-63, s82:83 bits 64-127; filled to more than 96 bits, a dword at a time
			"s_cmp_gt_u32 %[cnt], 96\n\t"
			"s_cbranch_scc1 2f\n\t"
			"v_readlane_b32 s86, %[vin], %[widx]\n\t"
			"s_mov_b32 s87, 0\n\t"
			"s_cmp_lt_u32 %[cnt], 64\n\t"
			"s_cbranch_scc0 10f\n\t"
			"s_lshl_b64 s[88:89], s[86:87], %[cnt]\n\t"          // the dword's bits below bit 64 ...
			"s_or_b64 s[84:85], s[84:85], s[88:89]\n\t"
			"s_cmp_le_u32 %[cnt], 32\n\t"
			"s_cbranch_scc1 11f\n\t"
			"s_sub_u32 s88, 64, %[cnt]\n\t"                      // ... and those above it (cnt 33-63: a shift by 31-1)
			"s_lshr_b32 s88, s86, s88\n\t"
			"s_or_b32 s82, s82, s88\n\t"
			"s_branch 11f\n"
			"10:\n\t"
			"s_sub_u32 s90, %[cnt], 64\n\t"
			"s_lshl_b64 s[88:89], s[86:87], s90\n\t"
			"s_or_b64 s[82:83], s[82:83], s[88:89]\n"
			"11:\n\t"
			"s_add_u32 %[cnt], %[cnt], 32\n\t"
			"s_add_u32 %[taken], %[taken], 1\n\t"
			"s_add_u32 %[widx], %[widx], 1\n\t"
			"s_cmp_eq_u32 %[widx], 64\n\t"
			"s_cbranch_scc1 20f\n\t"
			"s_branch 1b\n"
			"2:\n\t"                                          // lane j: v60:61 = the 64 bits from bit j on; v41 entry, v42 code bits, v43 extra bits, v44 length
			"v_lshrrev_b64 v[60:61], %[vlane], s[84:85]\n\t"
			"s_mov_b64 exec, -2\n\t"
			"v_lshlrev_b64 v[62:63], v57, s[82:83]\n\t"
			"s_mov_b64 exec, s[94:95]\n\t"
			"v_or_b32 v60, v60, v62\n\t"
			"v_or_b32 v61, v61, v63\n\t"
			"v_and_b32 v40, 0x3ff, v60\n\t"
			"v_lshl_add_u32 v40, v40, 2, %[lit]\n\t"
			"ds_read_b32 v41, v40\n\t"
			"s_waitcnt lgkmcnt(0)\n\t"
			"v_and_b32 v42, 15, v41\n\t"
			"v_bfe_u32 v43, v41, 4, 4\n\t"
			"v_alignbit_b32 v40, v61, v60, v42\n\t"
			"v_bfe_u32 v40, v40, 0, v43\n\t"
			"v_lshrrev_b32 v44, 16, v41\n\t"
			"v_add_u32 v44, v44, v40\n\t"
			"v_add_u32 v45, v42, v43\n\t"
			"v_alignbit_b32 v40, v61, v60, v45\n\t"
			"v_and_b32 v40, 0xff, v40\n\t"
			"v_lshl_add_u32 v40, v40, 2, %[lit]\n\t"
			"ds_read_b32 v46, v40 offset:4096\n\t"              // v46 distance entry, v47 code bits, v48 extra bits, v50 distance, v49 the match's bits
			"s_waitcnt lgkmcnt(0)\n\t"
			"v_and_b32 v47, 15, v46\n\t"
			"v_bfe_u32 v48, v46, 4, 4\n\t"
			"v_add_u32 v49, v45, v47\n\t"
			"v_alignbit_b32 v40, v61, v60, v49\n\t"
			"v_bfe_u32 v40, v40, 0, v48\n\t"
			"v_lshrrev_b32 v50, 16, v46\n\t"
			"v_add_u32 v50, v50, v40\n\t"
			"v_add_u32 v49, v49, v48\n\t"
			"v_and_b32 v40, 0x700, v41\n\t"
			"v_cmp_eq_u32_e64 s[90:91], 0, v40\n\t"              // s90:91 the lanes that see a literal
			"v_and_b32 v56, 0x700, v46\n\t"
			"v_lshl_or_b32 v40, v56, 4, v40\n\t"
			"v_cmp_eq_u32_e32 vcc, 0x1100, v40\n\t"              // a length and a distance, both from their tables
			"v_mov_b32 v51, 0x100\n\t"
			"v_cndmask_b32_e32 v51, v51, v49, vcc\n\t"
			"v_cndmask_b32_e64 v51, v51, v42, s[90:91]\n\t"      // v51 the symbol's bits (0x100: not for this path), v52 the bytes it makes
			"v_cndmask_b32_e64 v52, v44, 1, s[90:91]\n\t"
			"s_mov_b32 s60, 0\n\t"
			"s_mov_b64 s[88:89], 0\n"
			"3:\n\t"                                          // the chain: s60 the offset, s88:89 the lanes at which a symbol begins
			"v_readlane_b32 s61, v51, s60\n\t"
			"s_add_u32 s61, s61, s60\n\t"
			"s_cmp_gt_u32 s61, %[cnt]\n\t"
			"s_cbranch_scc1 4f\n\t"
			"s_bitset1_b64 s[88:89], s60\n\t"
			"s_mov_b32 s60, s61\n\t"
			"s_cmp_lt_u32 s60, 64\n\t"                           // (a lane per offset: 64 of them)
			"s_cbranch_scc1 3b\n"
			"4:\n\t"
			"s_cmp_eq_u64 s[88:89], 0\n\t"
			"s_cbranch_scc1 21f\n\t"
			"v_cndmask_b32_e64 v53, 0, v52, s[88:89]\n\t"        // the symbols' places: v54 inclusive prefix sum of their bytes, v55 position
			"v_mov_b32 v54, v53\n\t"
			"s_nop 1\n\t"
			"v_add_u32_dpp v54, v54, v54 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
			"s_nop 1\n\t"
			"v_add_u32_dpp v54, v54, v54 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
			"s_nop 1\n\t"
			"v_add_u32_dpp v54, v54, v54 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
			"s_nop 1\n\t"
			"v_add_u32_dpp v54, v54, v54 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
			"s_nop 1\n\t"
			"v_add_u32_dpp v54, v54, v54 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
			"s_nop 1\n\t"
			"v_add_u32_dpp v54, v54, v54 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
			"v_sub_u32 v55, v54, v53\n\t"
			"v_add_u32 v55, %[op], v55\n\t"
			"v_readlane_b32 s62, v54, 63\n\t"                    // s62 the group's bytes
			"v_add_u32 v40, v55, v52\n\t"
			"v_cmp_lt_u32_e32 vcc, %[ue], v40\n\t"
			"s_andn2_b64 s[92:93], s[88:89], s[90:91]\n\t"       // s92:93 the matches among the symbols
			"s_mov_b32 s64, 0\n\t"
			"s_and_b64 s[86:87], vcc, s[92:93]\n\t"              // ... that end behind the unit: the first of them ends the group
			"s_cbranch_scc0 5f\n\t"
			"s_ff1_i32_b64 s63, s[86:87]\n\t"
			"s_lshl_b64 s[86:87], 1, s63\n\t"
			"s_add_u32 s86, s86, -1\n\t"
			"s_addc_u32 s87, s87, -1\n\t"
			"s_and_b64 s[88:89], s[88:89], s[86:87]\n\t"
			"s_and_b64 s[92:93], s[92:93], s[86:87]\n\t"
			"v_readlane_b32 s62, v55, s63\n\t"
			"s_sub_u32 s62, s62, %[op]\n\t"
			"s_mov_b32 s60, s63\n\t"
			"s_mov_b32 s64, 1\n"
			"5:\n\t"
			"s_add_u32 s61, %[op], s62\n\t"
			"s_cmp_gt_u32 s61, %[olen]\n\t"
			"s_cbranch_scc1 25f\n\t"
			"s_cmp_eq_u64 s[92:93], 0\n\t"
			"s_cbranch_scc1 6f\n\t"
			"v_cmp_gt_u32_e32 vcc, v50, v55\n\t"
			"s_and_b64 s[86:87], vcc, s[92:93]\n\t"
			"s_cbranch_scc1 24f\n\t"
			"s_bcnt1_i32_b64 s65, s[92:93]\n\t"
			"s_add_u32 s66, %[ntok], s65\n\t"
			"s_cmp_gt_u32 s66, 64\n\t"
			"s_cbranch_scc1 22f\n\t"
			"s_mov_b64 exec, s[92:93]\n\t"                       // the matches' tokens: position | length << 16, distance
			"v_mbcnt_lo_u32_b32 v56, s92, 0\n\t"
			"v_mbcnt_hi_u32_b32 v56, s93, v56\n\t"
			"v_add_u32 v56, %[ntok], v56\n\t"
			"v_lshl_add_u32 v56, v56, 2, %[tok]\n\t"
			"v_lshl_or_b32 v40, v44, 16, v55\n\t"
			"ds_write_b32 v56, v40\n\t"
			"ds_write_b32 v56, v50 offset:1088\n\t"
			"s_mov_b64 exec, s[94:95]\n\t"
			"s_mov_b32 %[ntok], s66\n"
			"6:\n\t"
			"s_and_b64 s[86:87], s[88:89], s[90:91]\n\t"         // the literals' bytes
			"s_cbranch_scc0 7f\n\t"
			"s_mov_b64 exec, s[86:87]\n\t"
			"v_add_u32 v40, %[a0], v55\n\t"
			"v_and_b32 v40, 0x7ff, v40\n\t"
			"v_add_u32 v40, %[ring], v40\n\t"
			"ds_write_b8_d16_hi v40, v41\n\t"
			"s_mov_b64 exec, s[94:95]\n"
			"7:\n\t"
			"s_add_u32 %[op], %[op], s62\n\t"
			"s_cmp_lt_u32 s60, 64\n\t"                           // the buffer moves on by s60 bits (up to 99)
			"s_cbranch_scc1 8f\n\t"
			"s_sub_u32 s61, s60, 64\n\t"
			"s_lshr_b64 s[84:85], s[82:83], s61\n\t"
			"s_mov_b64 s[82:83], 0\n\t"
			"s_branch 9f\n"
			"8:\n\t"
			"s_cmp_eq_u32 s60, 0\n\t"
			"s_cbranch_scc1 9f\n\t"
			"s_lshr_b64 s[84:85], s[84:85], s60\n\t"
			"s_sub_u32 s61, 64, s60\n\t"
			"s_lshl_b64 s[86:87], s[82:83], s61\n\t"
			"s_or_b64 s[84:85], s[84:85], s[86:87]\n\t"
			"s_lshr_b64 s[82:83], s[82:83], s60\n"
			"9:\n\t"
			"s_sub_u32 %[cnt], %[cnt], s60\n\t"
			"s_cmp_lg_u32 s64, 0\n\t"
			"s_cbranch_scc1 21f\n\t"
			"s_cmp_ge_u32 %[op], %[ue]\n\t"
			"s_cbranch_scc1 22f\n\t"
			"s_branch 1b\n"
			"20:\n\t"
			"s_mov_b32 %[why], 1\n\t"
			"s_branch 29f\n"
			"21:\n\t"
			"s_mov_b32 %[why], 2\n\t"
			"s_branch 29f\n"
			"22:\n\t"
			"s_mov_b32 %[why], 3\n\t"
			"s_branch 29f\n"
			"24:\n\t"
			"s_mov_b32 %[why], 4\n\t"
			"s_branch 29f\n"
			"25:\n\t"
			"s_mov_b32 %[why], 5\n"
			"29:\n\t"
			"s_mov_b64 exec, s[94:95]\n\t"
			"s_waitcnt lgkmcnt(0)"
			: [bb] "+{s[84:85]}"(b.bb), [bx] "+{s[82:83]}"(bx_s), [cnt] "+s"(b.cnt), [widx] "+s"(b.widx), [taken] "+s"(b.taken), [op] "+s"(op), [ntok] "+s"(ntok), [why] "=&s"(why)
			: [vin] "v"(b.vin), [vlane] "v"(lane), [ue] "s"(ue), [olen] "s"(out_len), [a0] "s"(a0), [lit] "s"(lit_base), [tok] "s"(tok_base), [ring] "s"(ring_base)
			: "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95",
			  "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v60", "v61", "v62", "v63",
			  "vcc", "scc", "memory");
		b.bx = bx_s;
		SK_ISTAMP(1);
		if (why == 1u) {                                                    // the window behind, and the one after it on its way
			b.vin = b.vnext;
			b.vnext = __builtin_amdgcn_raw_buffer_load_b32(b.rs, (int)(b.next_off + 4u * (u32)lane), 0, 0);
			b.next_off += 256u;
			b.widx = 0u;
			continue;
		}
		if (why == 4u) { err = 5u; break; }
		if (why == 5u) { err = 6u; break; }
		if (why == 3u || op >= ue) finish_batch();
		if (why != 2u) continue;
		{
			// the symbol at the buffer's first bit, bit by bit: the end of the block, a code longer than its table's index or no code at
			// all, a match that ends behind the unit (or one whose 36 bits the buffer did not hold)
			inf_refill(b, lane);
			u32 e1 = inf_uniform(L.lit[(u32)b.bb & ((1u << kInfLitBits) - 1u)]);
			if (((e1 >> 8) & 7u) == kKindLong) e1 = inf_long_code<0, kInfLitBits>(L, b.bb);
			const u32 k1 = (e1 >> 8) & 7u;
			inf_take(b, e1 & 15u);
			if (k1 == kKindLiteral) {
				if (op >= out_len) { err = 6u; break; }
				if (lane == 0) L.ring[(a0 + op) & (kRing - 1u)] = (uint8_t)(e1 >> 16);
				op++;
				if (op >= ue) finish_batch();
				continue;
			}
			if (k1 == kKindEob) { done = true; break; }
			if (k1 != kKindBase) { err = 4u; break; }
			const u32 len = (e1 >> 16) + inf_take(b, (e1 >> 4) & 15u);
			inf_refill(b, lane);
			u32 d = inf_uniform(L.dist[(u32)b.bb & ((1u << kInfDistBits) - 1u)]);
			if (((d >> 8) & 7u) == kKindLong) d = inf_long_code<1, kInfDistBits>(L, b.bb);
			if (((d >> 8) & 7u) != kKindBase) { err = 4u; break; }
			inf_take(b, d & 15u);
			inf_refill(b, lane);
			const u32 dist = (d >> 16) + inf_take(b, (d >> 4) & 15u);
			if (dist > op) { err = 5u; break; }
			if (len > out_len - op) { err = 6u; break; }
			emit_match(len, dist);
			SK_ISTAMP(3);
		}
	}
#else
	while (!done && !err) {
		inf_refill(b, lane);
		// ---- the group: lane j decodes the symbol that would begin at bit j of the buffer — a literal, or a whole match (length code,
		// extra bits, distance code, extra bits: at most 36 bits when both codes are within their tables' index bits)
		const u64 sh = b.bb >> (u32)lane;
		const u32 w0 = (u32)sh, w1 = (u32)(sh >> 32);
		const u32 e = L.lit[w0 & ((1u << kInfLitBits) - 1u)];
		const u32 ll = e & 15u, x = (e >> 4) & 15u, kind = (e >> 8) & 7u;
		const u32 len_v = (e >> 16) + __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(w1, w0, ll), 0u, x);
		const u32 s2 = ll + x;                                              // <= 15: the table's index bits + 5
		const u32 de = L.dist[__builtin_amdgcn_alignbit(w1, w0, s2) & ((1u << kInfDistBits) - 1u)];
		const u32 dl = de & 15u, dx = (de >> 4) & 15u, dkind = (de >> 8) & 7u;
		const u32 s3 = s2 + dl;                                             // <= 23
		const u32 dist_v = (de >> 16) + __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(w1, w0, s3), 0u, dx);
		const bool is_lit = kind == kKindLiteral, is_match = kind == kKindBase && dkind == kKindBase;
		const u32 vtot = is_lit ? ll : (is_match ? s3 + dx : 0x100u);        // the symbol's bits; "longer than any buffer" for what this path does not take
		const u32 vout = is_lit ? 1u : len_v;
		u32 o = 0u;
#ifdef SK_INF_STAMPS
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		SK_ISTAMP(1);
		ist_acc[10]++;
#endif
		for (;;) {
			// ---- where the symbols begin, from offset o on, until one whose bits are not all in the buffer (or that this path does not take)
			u64 mask;
			u32 t;
			asm volatile(
				"s_mov_b64 %[mask], 0\n"
				"1:\n\t"
				"v_readlane_b32 %[t], %[vtot], %[o]\n\t"
				"s_add_u32 %[t], %[t], %[o]\n\t"
				"s_cmp_gt_u32 %[t], %[cnt]\n\t"
				"s_cbranch_scc1 2f\n\t"
				"s_bitset1_b64 %[mask], %[o]\n\t"
				"s_mov_b32 %[o], %[t]\n\t"
				"s_branch 1b\n"
				"2:"
				: [mask] "=&s"(mask), [t] "=&s"(t), [o] "+s"(o)
				: [vtot] "v"(vtot), [cnt] "s"(b.cnt)
				: "scc");
			SK_ISTAMP(2);
			if (mask != 0ull) {
				// ---- their places in the output: a prefix sum of the bytes they make
				const bool sym = (mask >> (u32)lane) & 1ull;
				const u32 incl = inf_wave_scan(sym ? vout : 0u);
				u32 pos = op + incl - (sym ? vout : 0u);
				u32 total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
				const bool m_here = sym && !is_lit;
				// a match that would end behind the unit ends the group before it; it is taken alone, below
				const u64 crossing = __builtin_amdgcn_ballot_w64(m_here && pos + vout > ue);
				bool alone = false;
				u32 c = 0u;
				if (crossing != 0ull) {
					c = (u32)__builtin_ctzll(crossing);
					alone = true;
					mask &= (1ull << c) - 1ull;
					total = (u32)__builtin_amdgcn_readlane((int)pos, (int)c) - op;
				}
				const bool take = (mask >> (u32)lane) & 1ull;
				const bool m_take = take && !is_lit;
				if (op + total > out_len) { err = 6u; break; }
				if (__builtin_amdgcn_ballot_w64(m_take && dist_v > pos) != 0ull) { err = 5u; break; }
				const u64 mm = __builtin_amdgcn_ballot_w64(m_take);
				const u32 nm = (u32)__builtin_popcountll(mm);
				if (ntok + nm > 64u) finish_batch();
				if (take && is_lit) L.ring[(a0 + pos) & (kRing - 1u)] = (uint8_t)(e >> 16);
				if (m_take) {
					const u32 k = ntok + __builtin_amdgcn_mbcnt_hi((u32)(mm >> 32), __builtin_amdgcn_mbcnt_lo((u32)mm, 0u));
					tokA[k] = pos | (len_v << 16);
					tokB[k] = dist_v;
				}
				ntok += nm;
				op += total;
				SK_ISTAMP(3);
				if (op >= ue) finish_batch();
				if (alone) {
					const u32 len = (u32)__builtin_amdgcn_readlane((int)len_v, (int)c), dist = (u32)__builtin_amdgcn_readlane((int)dist_v, (int)c);
					if (dist > op) { err = 5u; break; }
					if (len > out_len - op) { err = 6u; break; }
					emit_match(len, dist);
					o = c + (u32)__builtin_amdgcn_readlane((int)vtot, (int)c);
					continue;                                                  // (the group's registers stand: on with the symbol behind it)
				}
			}
			// ---- the chain stopped at o: out of bits (the next group), or a symbol for the long way
			if (o != 0u) break;
			{
				// the end of the block, a code longer than its table's index, or no code at all: this symbol bit by bit, and the next group behind it
				u32 e1 = inf_uniform(L.lit[(u32)b.bb & ((1u << kInfLitBits) - 1u)]);
				if (((e1 >> 8) & 7u) == kKindLong) e1 = inf_long_code<0, kInfLitBits>(L, b.bb);
				const u32 k1 = (e1 >> 8) & 7u;
				inf_take(b, e1 & 15u);
				if (k1 == kKindLiteral) {
					if (op >= out_len) { err = 6u; break; }
					if (lane == 0) L.ring[(a0 + op) & (kRing - 1u)] = (uint8_t)(e1 >> 16);
					op++;
					if (op >= ue) finish_batch();
					break;
				}
				if (k1 == kKindEob) { done = true; break; }
				if (k1 != kKindBase) { err = 4u; break; }
				const u32 len = (e1 >> 16) + inf_take(b, (e1 >> 4) & 15u);
				inf_refill(b, lane);
				u32 d = inf_uniform(L.dist[(u32)b.bb & ((1u << kInfDistBits) - 1u)]);
				if (((d >> 8) & 7u) == kKindLong) d = inf_long_code<1, kInfDistBits>(L, b.bb);
				if (((d >> 8) & 7u) != kKindBase) { err = 4u; break; }
				inf_take(b, d & 15u);
				inf_refill(b, lane);
				const u32 dist = (d >> 16) + inf_take(b, (d >> 4) & 15u);
				if (dist > op) { err = 5u; break; }
				if (len > out_len - op) { err = 6u; break; }
				emit_match(len, dist);
				break;
			}
		}
		if (o != 0u) inf_drop(b, o);
		SK_ISTAMP(3);
	}
#endif
	if (!err) finish_batch();
#ifdef SK_INF_STAMPS
	if (lane == 0) for (int i = 0; i < 16; i++) atomicAdd(&g_inf_stamps[i], ist_acc[i]);
#endif
	r.put(b); r.op = op; r.flushed = flushed; r.err = err;
	return r;
}

struct InfBlock { u64 in_off; u32 in_len, out_len; u64 out_off; u32 crc, pad; };
static_assert(sizeof(InfBlock) == 32, "== sk_bgzf_block");

// One block.  Returns its status (0 = inflated, out_len bytes written).
__device__ __forceinline__ u32 inf_block(InfLds &L, int wave, const uint8_t *comp, const InfBlock &blk, uint8_t *out, int lane)
{
	InfBits b;
	const u64 in4 = blk.in_off & ~3ull;
	const u32 head = (u32)(blk.in_off - in4);
	const u32 range = (head + blk.in_len + 3u) & ~3u;
	b.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(comp + in4), 0, (int)range, 0x00020000);
	inf_seek(b, head, lane);
	// the block's first output address, modulo 2^20: what is used of it are its low bits (the ring's index, the halves' and the
	// 16-byte boundaries) and differences inside the block's 64 KiB — which must not wrap (a block that straddled 2^32 in the
	// inflated stream of a 3.6 GB file came out with its flushes cut short: its CRC told, zlib did it again)
	const u32 a0 = (u32)blk.out_off & 0xFFFFFu;
	const u32 out_len = blk.out_len;
	u32 op = 0u, flushed = 0u;
	u32 err = 0u;
	for (;;) {
		inf_refill(b, lane);
		const u32 bfinal = inf_take(b, 1u), btype = inf_take(b, 2u);
		if (btype == 0u) {
			// stored: LEN / NLEN behind the next byte boundary, then LEN bytes as they are
			inf_take(b, b.cnt & 7u);
			inf_refill(b, lane);
			const u32 len = inf_take(b, 16u);
			inf_refill(b, lane);
			const u32 nlen = inf_take(b, 16u);
			if ((len ^ nlen) != 0xffffu) { err = 2u; break; }
			if (len > out_len - op) { err = 6u; break; }
			const u32 src0 = b.taken * 4u - (b.cnt >> 3);                 // byte offset (in the descriptor's range) of the first stored byte
			if (src0 + len > head + blk.in_len) { err = 8u; break; }
			for (u32 done = 0u; done < len;) {
				const u32 room = kHalf - ((a0 + op) & (kHalf - 1u));
				const u32 seg = min(len - done, room);
				for (u32 k = (u32)lane; k < seg; k += 64u) L.ring[(a0 + op + k) & (kRing - 1u)] = comp[in4 + src0 + done + k];
				op += seg; done += seg;
				if (((a0 + op) & (kHalf - 1u)) == 0u) { inf_flush(L, out + blk.out_off, a0, flushed, op, lane); flushed = op; }
			}
			inf_seek(b, src0 + len, lane);
		} else if (btype == 1u || btype == 2u) {
			int hlit = 288, hdist = 32;
			if (btype == 1u) {
				for (int s = lane; s < 320; s += 64) L.lens[s] = (uint8_t)(s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : (s < 288 ? 8 : 5))));
				inf_lds_fence();
			} else {
				hlit = (int)inf_take(b, 5u) + 257;
				hdist = (int)inf_take(b, 5u) + 1;
				const int hclen = (int)inf_take(b, 4u) + 4;
				if (hlit > 286 || hdist > 30) { err = 3u; break; }
				if (lane < 19) L.lens[lane] = 0;
				inf_lds_fence();
				for (int i = 0; i < hclen; i++) {
					inf_refill(b, lane);
					const u32 v = inf_take(b, 3u);
					if (lane == 0) L.lens[kInfPreOrder[i]] = (uint8_t)v;
				}
				inf_lds_fence();
				err = inf_build<2, kInfPreBits, 7>(L, L.lens, 19, lane);
				if (err) break;
				const int total = hlit + hdist;
				int n = 0;
				u32 prev = 0u;
				while (n < total) {
					inf_refill(b, lane);
					const u32 e = inf_uniform(L.pre[(u32)b.bb & ((1u << kInfPreBits) - 1u)]);
					const u32 l = e & 15u, sym = e >> 4;
					if (l == 0u) { err = 4u; break; }
					inf_take(b, l);
					if (sym < 16u) {
						if (lane == 0) L.lens[n] = (uint8_t)sym;
						prev = sym;
						n++;
						continue;
					}
					u32 rep, val = 0u;
					if (sym == 16u) { if (n == 0) { err = 4u; break; } val = prev; rep = 3u + inf_take(b, 2u); }
					else if (sym == 17u) rep = 3u + inf_take(b, 3u);
					else rep = 11u + inf_take(b, 7u);
					if (n + (int)rep > total) { err = 4u; break; }
					for (u32 k = (u32)lane; k < rep; k += 64u) L.lens[n + (int)k] = (uint8_t)val;
					prev = val;
					n += (int)rep;
				}
				if (err) break;
				inf_lds_fence();
				if (inf_uniform(L.lens[256]) == 0u) { err = 3u; break; }      // no end-of-block code
			}
			// (the distance lengths are moved behind a literal/length alphabet of full size, so that `sorted` and the builders see fixed places)
			err = inf_build<0, kInfLitBits, 15>(L, L.lens, hlit, lane);
			if (err) break;
			err = inf_build<1, kInfDistBits, 15>(L, L.lens + hlit, hdist, lane);
			if (err) break;
			// ---- the symbols of the block (a function of its own: see inf_symbols)
			{
				InfRun r;
				r.put(b); r.op = op; r.flushed = flushed; r.err = 0u;
#if SK_INF_GROUPS
				r = inf_symbols_groups(wave, r, comp + in4, range, a0, out_len, out + blk.out_off, lane);
#else
				r = inf_symbols(wave, r, comp + in4, range, a0, out_len, out + blk.out_off, lane);
#endif
				r.get(b); op = r.op; flushed = r.flushed; err = r.err;
			}
			if (err) break;
		} else {
			err = 1u;
			break;
		}
		if (bfinal) break;
	}
	if (err) return err;
	inf_flush(L, out + blk.out_off, a0, flushed, op, lane);
	if (op != out_len) return 7u;
	if (inf_consumed(b) > head + blk.in_len) return 8u;
	return 0u;
}

__global__ __launch_bounds__(kInfWaves * 64) void bgzf_inflate_kernel(const uint8_t *comp, const InfBlock *blocks, int64_t n_blocks, uint8_t *out, u32 *status)
{
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	InfLds &L = reinterpret_cast<InfLds *>(inf_smem)[wave];
	for (int64_t bi = (int64_t)blockIdx.x * kInfWaves + wave; bi < n_blocks; bi += (int64_t)gridDim.x * kInfWaves) {
		InfBlock blk;
		{	// (uniform: scalar loads)
			const u64 *p = reinterpret_cast<const u64 *>(blocks + bi);
			const u64 w0 = p[0], w1 = p[1], w2 = p[2];
			blk.in_off = w0;
			blk.in_len = (u32)__builtin_amdgcn_readfirstlane((int)(u32)w1);
			blk.out_len = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(w1 >> 32));
			blk.out_off = w2;
			blk.in_off = (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)w0) | ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(w0 >> 32)) << 32);
			blk.out_off = (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)w2) | ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(w2 >> 32)) << 32);
			blk.crc = 0u; blk.pad = 0u;
		}
		u32 st = 0u;
#ifdef SK_INF_STAMPS
		const u64 t_blk = __builtin_amdgcn_s_memtime();
#endif
		if (blk.out_len != 0u || blk.in_len != 0u) st = inf_block(L, wave, comp, blk, out, lane);
#ifdef SK_INF_STAMPS
		if (lane == 0) { atomicAdd(&g_inf_stamps[15], __builtin_amdgcn_s_memtime() - t_blk); atomicAdd(&g_inf_stamps[14], 1ull); }
#endif
		if (lane == 0) status[bi] = st;
		inf_lds_fence();
	}
}

// ---- CRC-32 -------------------------------------------------------------------------------------------------------
// reflected polynomial 0xEDB88320 (RFC 1952 §8).  A lane's piece 16 bytes per load, a dword per step through four 256-entry
// tables in LDS (slicing by four); the pieces are joined with crc(A ++ B) = crc(A) * x^(8 |B|) + crc(B) in GF(2)[x] / P (bit-reflected arithmetic).
__device__ __forceinline__ u32 crc_mul(u32 a, u32 b)                     // a * b mod P, both bit-reflected (bit 31 = x^0)
{
	u32 r = 0u;
	for (int i = 0; i < 32; i++) {
		if (a & 0x80000000u) r ^= b;
		a <<= 1;
		b = (b >> 1) ^ ((b & 1u) ? 0xEDB88320u : 0u);
	}
	return r;
}
__device__ __forceinline__ u32 crc_xpow8(u32 nbytes, const u32 *pw)     // x^(8 n) mod P: pw[k] = x^(8 * 2^k)
{
	u32 r = 0x80000000u;                                                 // x^0
	for (int k = 0; nbytes != 0u; k++, nbytes >>= 1)
		if (nbytes & 1u) r = crc_mul(r, pw[k]);
	return r;
}

__global__ __launch_bounds__(256) void bgzf_crc_kernel(const uint8_t *out, const InfBlock *blocks, int64_t n_blocks, u32 *status)
{
	// four tables (slicing by four: a dword of the message per step), 4 KiB of LDS
	__shared__ u32 tab[4][256];
	__shared__ u32 pw[20];
	{
		u32 c = threadIdx.x;
		for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1u) ? 0xEDB88320u : 0u);
		tab[0][threadIdx.x] = c;
		__syncthreads();
		u32 t = c;
		for (int k = 1; k < 4; k++) { t = (t >> 8) ^ tab[0][t & 0xffu]; tab[k][threadIdx.x] = t; }
		if (threadIdx.x == 0) {
			u32 p = 0x00800000u;                                         // x^8
			for (int k = 0; k < 20; k++) { pw[k] = p; p = crc_mul(p, p); }
		}
	}
	__syncthreads();
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (int64_t bi = (int64_t)blockIdx.x * 4 + wave; bi < n_blocks; bi += (int64_t)gridDim.x * 4) {
		const InfBlock blk = blocks[bi];
		const u32 n = blk.out_len;
		// a lane's piece: 1/64 of the block rounded up to 16 bytes, from a 16-byte boundary of the output ADDRESS on (the first lane
		// also takes the bytes before the first boundary)
		const uint8_t *p = out + blk.out_off;
		const u32 mis = (u32)((16u - ((uintptr_t)p & 15u)) & 15u);        // bytes before the first boundary
		const u32 headn = mis < n ? mis : n;
		const u32 body = n - headn;
		const u32 piece = (((body + 63u) >> 6) + 15u) & ~15u;
		const u32 lo = headn + min(body, piece * (u32)lane), hi = headn + min(body, piece * (u32)lane + piece);
		u32 c = lane == 0 ? 0xFFFFFFFFu : 0u;                            // (the initial value rides through the first piece; the others start from 0: the register's linearity)
		u32 len = hi - lo;
		if (lane == 0) {
			for (u32 i = 0; i < headn; i++) c = tab[0][(c ^ p[i]) & 0xffu] ^ (c >> 8);
			len += headn;
		}
		u32 i = lo;
		for (; i + 16u <= hi; i += 16u) {
			const u32x4 w = *reinterpret_cast<const u32x4 *>(p + i);
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const u32 x = c ^ w[k];
				c = tab[3][x & 0xffu] ^ tab[2][(x >> 8) & 0xffu] ^ tab[1][(x >> 16) & 0xffu] ^ tab[0][x >> 24];
			}
		}
		for (; i < hi; i++) c = tab[0][(c ^ p[i]) & 0xffu] ^ (c >> 8);
		// join: lane l takes over its right neighbour at distance 1, 2, 4 ...: c = c * x^(8 len_right) + c_right
		for (int o = 1; o < 64; o <<= 1) {
			const u32 cr = __shfl_down(c, o), lr = __shfl_down(len, o);
			if ((lane & (2 * o - 1)) == 0 && lane + o < 64) {
				c = crc_mul(c, crc_xpow8(lr, pw)) ^ cr;
				len += lr;
			}
		}
		if (lane == 0 && (c ^ 0xFFFFFFFFu) != blk.crc) atomicOr(&status[bi], 0x100u);
	}
}

// ---- the records of the inflated stream ----------------------------------------------------------------------------
__device__ __forceinline__ u32 bam_le32(const uint8_t *p)
{
	// (any alignment: two aligned dwords and a byte shift)
	const uintptr_t a = (uintptr_t)p;
	const u32 *q = reinterpret_cast<const u32 *>(a & ~(uintptr_t)3);
	const u32 sh = (u32)(a & 3u);
	const u32 lo = q[0];
	if (sh == 0u) return lo;
	return __builtin_amdgcn_alignbyte(q[1], lo, sh);
}

// Walk block c from entry[c] (a position of the stream) to the first record that begins at or behind the block's end
// (bend[c] = where block c + 1 begins): exitp[c] = that position, nrec[c] = records begun inside.  A record whose block_size
// no record can have (< 32) or that reaches beyond the stream ends the walk with exitp = ~0 - (1 or 2): the caller's CPU path
// reports such files.  One lane per block.
struct WalkArgs {
	const uint8_t *stream;
	u64 stream_len;
	const u64 *bend;         // [n]: end of block c in the stream (== beginning of block c + 1)
	u64 *entry;              // [n + 1]
	u64 *exitp;              // [n]
	u32 *nrec;               // [n]
	int64_t n;
	u64 first;               // where the first record begins (behind the BAM header)
	u32 *changed;            // bam_walk_fix_kernel: number of entries it replaced
};
constexpr u64 kWalkBadRecord = ~0ull - 1ull, kWalkTruncated = ~0ull - 2ull;

__device__ __forceinline__ void bam_walk_one(const WalkArgs &a, int64_t c)
{
	u64 o = a.entry[c];
	const u64 end = a.bend[c];
	u32 n = 0u;
	while (o < end) {
		if (o + 4 > a.stream_len) { o = kWalkTruncated; break; }
		const u32 bs = bam_le32(a.stream + o);
		if (bs < 32u) { o = kWalkBadRecord; break; }
		if (o + 4 + (u64)bs > a.stream_len) { o = kWalkTruncated; break; }
		n++;
		o += 4 + (u64)bs;
	}
	a.exitp[c] = o;
	a.nrec[c] = n;
}

// first round: every block from its first byte (block 0 from the end of the header; a block the header covers wholly is entered where the header ends too)
__global__ __launch_bounds__(256) void bam_walk_kernel(const WalkArgs a)
{
	const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= a.n) return;
	const u64 begin = c == 0 ? 0ull : a.bend[c - 1];
	a.entry[c] = begin < a.first ? a.first : begin;
	bam_walk_one(a, c);
}
// later rounds: a block whose entry is not its predecessor's exit takes that exit and is walked again
__global__ __launch_bounds__(256) void bam_walk_fix_kernel(const WalkArgs a)
{
	const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= a.n || c == 0) return;
	u64 want = a.exitp[c - 1];
	if (want >= kWalkTruncated) return;                                  // (the predecessor ran into a bad record: nothing to propagate; the verdict is the caller's)
	if (want < a.first) want = a.first;
	if (a.entry[c] == want) return;
	a.entry[c] = want;
	bam_walk_one(a, c);
	atomicAdd(a.changed, 1u);
}

// A block whose entry is not its predecessor's exit — a file whose writer cut its blocks anywhere — would be put right one block per
// round (its predecessor's exit is wrong until the one before THAT is right: the chain again).  So such a block first GUESSES: a wave
// looks, 64 offsets at a time from the block's first byte on, for the first offset at which three records in a row look like records
// (sizes that fit the stream, reference ids the header has, a read name that ends in NUL, a variable part that holds what the core
// says it holds), and walks from there.  A guess is only a guess: the rounds that follow verify every entry against its
// predecessor's exit as before, and replace the ones that fooled the test.
__device__ __forceinline__ bool bam_plausible(const uint8_t *s, u64 len, u64 o, int32_t n_ref)
{
	for (int d = 0; d < 3; d++) {
		if (o == len) return true;
		if (o + 36 > len) return false;
		const u32 bs = bam_le32(s + o);
		if (bs < 32u || bs > (1u << 28) || o + 4 + (u64)bs > len) return false;
		const int32_t tid = (int32_t)bam_le32(s + o + 4), pos = (int32_t)bam_le32(s + o + 8);
		const u32 w8 = bam_le32(s + o + 12), w12 = bam_le32(s + o + 16);
		const int32_t l_seq = (int32_t)bam_le32(s + o + 20), mtid = (int32_t)bam_le32(s + o + 24), mpos = (int32_t)bam_le32(s + o + 28);
		const u32 l_name = w8 & 0xffu, n_cigar = w12 & 0xffffu;
		if (tid < -1 || mtid < -1 || pos < -1 || mpos < -1 || l_seq < 0 || l_name == 0u) return false;
		if (n_ref >= 0 && (tid >= n_ref || mtid >= n_ref)) return false;
		if (32ull + l_name + 4ull * n_cigar + (((u64)l_seq + 1) >> 1) + (u64)l_seq > (u64)bs) return false;
		if (s[o + 36 + l_name - 1] != 0) return false;
		o += 4 + (u64)bs;
	}
	return true;
}
__global__ __launch_bounds__(256) void bam_walk_guess_kernel(const WalkArgs a, int32_t n_ref)
{
	const int lane = threadIdx.x & 63;
	const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (c >= a.n || c == 0) return;
	u64 want = a.exitp[c - 1];
	if (want < kWalkTruncated) {
		if (want < a.first) want = a.first;
		if (a.entry[c] == want) return;                                    // (already its predecessor's exit)
	}
	const u64 begin0 = a.bend[c - 1], end = a.bend[c];
	const u64 begin = begin0 < a.first ? a.first : begin0;
	u64 found = end;                                                       // no record begins in this block (one longer than the block covers it)
	for (u64 base = begin; base < end; base += 64) {
		const u64 o = base + (u64)lane;
		const bool ok = o < end && bam_plausible(a.stream, a.stream_len, o, n_ref);
		const u64 m = __ballot(ok);
		if (m) { found = base + (u64)(__ffsll((unsigned long long)m) - 1); break; }
	}
	if (lane == 0) {
		a.entry[c] = found;
		bam_walk_one(a, c);
	}
}

// S1 + H1 over the records of the verified chain, straight from the inflated bytes (sk_kernels.hip: bam_flag_tlen_kernel's
// predicate, src/sam_statistics.rs:63-69, src/sam_fragment_lengths.rs:29-43).  out = u64[3 counters][1 hist_total][max_frag + 1 bins].
// One lane per block; counters per lane, summed per workgroup; histogram bins by atomics on an LDS copy when it fits.
constexpr int kWalkLdsBins = 8192;
__global__ __launch_bounds__(256) void bam_walk_reduce_kernel(const WalkArgs a, int32_t max_frag, int want_counters, int want_hist, unsigned long long *out)
{
	__shared__ u32 lh[kWalkLdsBins];
	__shared__ u32 red[4];
	const bool lds_hist = want_hist && max_frag + 1 <= kWalkLdsBins;
	if (lds_hist) for (int i = threadIdx.x; i <= max_frag; i += blockDim.x) lh[i] = 0u;
	if (threadIdx.x < 4) red[threadIdx.x] = 0u;
	__syncthreads();
	const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	u32 total = 0u, aligned = 0u, dup = 0u, hist = 0u;
	if (c < a.n) {
		u64 o = a.entry[c];
		const u64 end = a.bend[c];
		while (o < end) {
			const uint8_t *r = a.stream + o;
			const u32 bs = bam_le32(r);
			const u32 w14 = bam_le32(r + 4 + 12);                           // n_cigar_op (low half), flag (high half)
			const u32 f = w14 >> 16;
			if (want_counters) {
				const bool primary = (f & (0x100u | 0x800u)) == 0u;            // src/sam_statistics.rs:64
				const bool mapped = primary && !(f & 0x4u);                    // :66
				total += primary ? 1u : 0u;
				aligned += mapped ? 1u : 0u;
				dup += (mapped && (f & 0x400u)) ? 1u : 0u;                     // :69
			}
			if (want_hist && (f & (0x1u | 0x40u | 0x4u | 0x8u | 0x400u | 0x100u | 0x800u)) == (0x1u | 0x40u)) {      // src/sam_fragment_lengths.rs:30-35
				const int32_t tid = (int32_t)bam_le32(r + 4), mtid = (int32_t)bam_le32(r + 4 + 20), tl = (int32_t)bam_le32(r + 4 + 28);
				if (tid == mtid) {                                             // :36
					const u32 fl = tl < 0 ? 0u - (u32)tl : (u32)tl;             // |tlen| as the reference widens it (i32::MIN -> 2^31)
					if (fl <= (u32)max_frag) {                                  // :38
						hist++;
						if (lds_hist) atomicAdd(&lh[fl], 1u);
						else atomicAdd(&out[4 + fl], 1ull);
					}
				}
			}
			o += 4 + (u64)bs;
		}
	}
	for (int s = 32; s > 0; s >>= 1) {
		total += __shfl_xor(total, s); aligned += __shfl_xor(aligned, s); dup += __shfl_xor(dup, s); hist += __shfl_xor(hist, s);
	}
	if ((threadIdx.x & 63) == 0) { atomicAdd(&red[0], total); atomicAdd(&red[1], aligned); atomicAdd(&red[2], dup); atomicAdd(&red[3], hist); }
	__syncthreads();
	if (threadIdx.x < 4 && red[threadIdx.x]) atomicAdd(&out[threadIdx.x], (unsigned long long)red[threadIdx.x]);
	if (lds_hist) for (int i = threadIdx.x; i <= max_frag; i += blockDim.x) if (lh[i]) atomicAdd(&out[4 + i], (unsigned long long)lh[i]);
}

// ---- launchers -------------------------------------------------------------------------------------------------------
hipError_t launch_bgzf_inflate(const uint8_t *comp, const void *blocks, int64_t n_blocks, uint8_t *out, uint32_t *status, int check_crc, int n_cu, hipStream_t st)
{
	if (n_blocks <= 0) return hipSuccess;
	static bool attr_set = false;
	const size_t lds = sizeof(InfLds) * kInfWaves;
	if (!attr_set) {
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(bgzf_inflate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) return e;
		attr_set = true;
	}
	const int per_cu = std::max(1, (int)((160 * 1024) / lds));
	int64_t grid = (n_blocks + kInfWaves - 1) / kInfWaves;
	const int64_t cap = (int64_t)n_cu * per_cu * 4;                       // a few rounds of resident workgroups: blocks differ in how long they take
	if (grid > cap) grid = cap;
	bgzf_inflate_kernel<<<dim3((unsigned)grid), dim3(kInfWaves * 64), lds, st>>>(comp, reinterpret_cast<const InfBlock *>(blocks), n_blocks, out, status);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess || !check_crc) return e;
	int64_t cgrid = (n_blocks + 3) / 4;
	if (cgrid > (int64_t)n_cu * 16) cgrid = (int64_t)n_cu * 16;
	bgzf_crc_kernel<<<dim3((unsigned)cgrid), dim3(256), 0, st>>>(out, reinterpret_cast<const InfBlock *>(blocks), n_blocks, status);
	return hipGetLastError();
}

hipError_t launch_bam_walk(const uint8_t *stream, uint64_t stream_len, const uint64_t *bend, uint64_t *entry, uint64_t *exitp, uint32_t *nrec, int64_t n,
                           uint64_t first, uint32_t *changed, int fix_round, int32_t n_ref, hipStream_t st)
{
	if (n <= 0) return hipSuccess;
	WalkArgs a;
	a.stream = stream; a.stream_len = stream_len; a.bend = reinterpret_cast<const u64 *>(bend); a.entry = reinterpret_cast<u64 *>(entry);
	a.exitp = reinterpret_cast<u64 *>(exitp); a.nrec = nrec; a.n = n; a.first = first; a.changed = changed;
	const unsigned grid = (unsigned)((n + 255) / 256);
	if (fix_round == 2) bam_walk_guess_kernel<<<(unsigned)((n + 3) / 4), 256, 0, st>>>(a, n_ref);
	else if (fix_round) bam_walk_fix_kernel<<<grid, 256, 0, st>>>(a);
	else bam_walk_kernel<<<grid, 256, 0, st>>>(a);
	return hipGetLastError();
}

hipError_t launch_bam_walk_reduce(const uint8_t *stream, uint64_t stream_len, const uint64_t *bend, const uint64_t *entry, int64_t n, int32_t max_frag,
                                  int want_counters, int want_hist, unsigned long long *out, hipStream_t st)
{
	if (n <= 0) return hipSuccess;
	WalkArgs a;
	a.stream = stream; a.stream_len = stream_len; a.bend = reinterpret_cast<const u64 *>(bend); a.entry = const_cast<u64 *>(reinterpret_cast<const u64 *>(entry));
	a.exitp = nullptr; a.nrec = nullptr; a.n = n; a.first = 0; a.changed = nullptr;
	bam_walk_reduce_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(a, max_frag, want_counters, want_hist, out);
	return hipGetLastError();
}

}  // namespace sk

#ifdef SK_INF_STAMPS
extern "C" int sk_debug_inflate_stamps(unsigned long long *out, int reset)
{
	hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(sk::g_inf_stamps), sizeof(unsigned long long) * 16);
	if (e == hipSuccess && reset) {
		unsigned long long z[16] = {};
		e = hipMemcpyToSymbol(HIP_SYMBOL(sk::g_inf_stamps), z, sizeof z);
	}
	return (int)e;
}
#endif
