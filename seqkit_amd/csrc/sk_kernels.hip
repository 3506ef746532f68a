// sk_kernels.hip — gfx950 (MI355X, CDNA4) kernels for seqkit's per-read hot path.
//
// Design (DESIGN.md has the long form):
//  * read batches are row-major fixed-stride byte matrices.  A tile = 64 consecutive rows = ONE
//    contiguous, 16-byte aligned byte range (64*stride), whatever the stride.  One 64-lane wavefront
//    owns a tile: it streams the range with 16 B/lane loads (1 KiB per wave instruction, fully
//    coalesced), applies the quality mask on the packed dwords as they pass (elementwise, no row
//    structure needed), stores the masked bases with 16 B/lane stores, and drops the quality bytes
//    into a wave-private LDS image of the tile.  The LDS image is the transposition point: afterwards
//    lane r walks row r from its 3' end (dword LDS reads + v_alignbyte for the row's misalignment).
//  * a workgroup is four wavefronts that share only the read-only matcher tables and the per-sample
//    histogram in LDS; there is no barrier inside the tile loop (the lanes of ONE wave hand data to each
//    other through the wave's private LDS tile).  2-4 workgroups are resident per CU depending on what
//    the pass is bound by (launch_tile_pass), persistent over tiles.
//  * barcode matching is lane-per-read and bit-sliced: walk the L positions, not the S candidates; the
//    per-candidate mismatch counts are column sums kept as bit planes (demux_row_bitsliced).  Sheets
//    the bit-sliced tables cannot hold fall back to a one-hot popcount or a byte-compare matcher.
//  * two batch layouts: row-major SoA matrices (tile_pass_kernel) and the tile-blocked layout, where
//    everything a 64-cluster tile reads is ONE contiguous range and everything it writes another
//    (tile_blocked_kernel).
//  * everything is integer/byte work bounded by HBM; there is no MFMA in this file by design.
//
// Reference semantics restated per function: see the citations (paths relative to the reference tree).
#include "sk_internal.h"

#include <cstdlib>
#include <mutex>
#include <vector>

namespace sk {

#ifndef SK_STRAGGLER_FROM
#define SK_STRAGGLER_FROM 6
#endif
#ifndef SK_STRAGGLER_ROWS
#define SK_STRAGGLER_ROWS 8
#endif
#ifndef SK_STRAGGLER_LEFT
#define SK_STRAGGLER_LEFT 6
#endif

typedef uint32_t u32;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

typedef const __attribute__((address_space(4))) uint32_t *const_u32_ptr;   // AMDGPU constant address space

constexpr u32 kLo7 = 0x7f7f7f7fu;
constexpr u32 kHi1 = 0x80808080u;

QualConsts make_qual_consts(int m)
{
	QualConsts q{};
	q.min_baseq = m;
	q.cl2 = 0; q.c72 = 0;
	if (m == 0) { q.mode = 0; return q; }
	int t2;
	if (33 + m <= 255) { q.mode = 1; t2 = 33 + m; }          // mask iff 33 <= q < 33+m
	else if (33 + m == 256) { q.mode = 3; t2 = 1; }           // mask iff q >= 33
	else { q.mode = 4; t2 = 33 + m - 256; }                   // mask iff q >= 33 or q < t2 (wrapped interval)
	int c2 = 256 - t2;
	q.cl2 = (u32)(c2 & 0x7f) * 0x01010101u;
	q.c72 = (c2 & 0x80) ? kHi1 : 0u;
	if (q.mode == 1 && !(c2 & 0x80)) q.mode = 2;              // bit 7 of the addend decides or / and in the carry
	return q;
}

// ---------------------------------------------------------------------------------------------------
// Packed-byte quality arithmetic.  For a dword q of four Phred+33 bytes:
//   ql = q & 0x7f7f7f7f
//   t1 = ql + 0x5f5f5f5f           low 7 bits of (q-33) mod 256 per byte; bit 7 = carry out of bit 6
//   vq = t1 ^ (~q & 0x80808080)    (q - 33) mod 256 per byte   -- `qual - 33u8` wrapping, release Rust
//   g1 = q | t1                    bit 7 = [q >= 33]   (carry of q + 223; 223 has bit 7 set)
//   g2                             bit 7 = [q >= t2]   (carry of q + (256 - t2), generic majority form)
// mask flag F (bit 7 of each byte) = [(q-33) mod 256 < min_baseq]   src/fasta_mask_by_quality.rs:42
// ---------------------------------------------------------------------------------------------------
// MODE: 0 never masks (min_baseq 0) | 1 g1 & ~g2, c72 set | 2 g1 & ~g2, c72 clear | 3 g1 | 4 g1 | ~g2 (c72 set)
template <int MODE>
__device__ __forceinline__ u32 lowq_flags(u32 q, u32 ql, u32 t1, u32 cl2)
{
	if (MODE == 0) return 0u;
	u32 g1 = q | t1;
	if (MODE == 3) return g1;
	u32 t2 = ql + cl2;
	u32 g2 = (MODE == 2) ? (q & t2) : (q | t2);
	if (MODE == 4) return g1 | ~g2;
	return g1 & ~g2;
}

// out byte i = flagged ? 'N' : s byte i, via one v_perm_b32: selector i picks s, 4+i picks 'N'.
__device__ __forceinline__ u32 mask_select(u32 s, u32 F)
{
	u32 sel = ((F >> 5) & 0x04040404u) | 0x03020100u;
	return __builtin_amdgcn_perm(0x4e4e4e4eu, s, sel);
}

__device__ __forceinline__ u32 sub33(u32 q, u32 t1) { return t1 ^ (~q & kHi1); }

template <int MODE>
__device__ __forceinline__ void mask_dword4(const u32x4 &q, const u32x4 &s, u32 cl2, u32x4 &out, u32x4 &vq)
{
#pragma unroll
	for (int i = 0; i < 4; i++) {
		u32 ql = q[i] & kLo7;
		u32 t1 = ql + 0x5f5f5f5fu;
		vq[i] = sub33(q[i], t1);
		out[i] = mask_select(s[i], lowq_flags<MODE>(q[i], ql, t1, cl2));
	}
}

// 16-byte accesses of the read-once / write-once byte matrices (SK_NT: bit 0 = nontemporal loads, bit 1 = stores)
#ifndef SK_NT
#define SK_NT 3
#endif
__device__ __forceinline__ u32x4 stream_load(const uint8_t *p)
{
#if SK_NT & 1
	return __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
#else
	return *reinterpret_cast<const u32x4 *>(p);
#endif
}
__device__ __forceinline__ void stream_store(uint8_t *p, const u32x4 &v)
{
#if SK_NT & 2
	__builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
#else
	*reinterpret_cast<u32x4 *>(p) = v;
#endif
}

// byte-granular load/store of a 16-byte chunk that crosses the end of the matrix
__device__ __forceinline__ u32x4 load_tail(const uint8_t *p, int valid)
{
	u32x4 v = {0u, 0u, 0u, 0u};
	for (int b = 0; b < valid; b++) v[b >> 2] |= (u32)p[b] << (8 * (b & 3));
	return v;
}
__device__ __forceinline__ void store_tail(uint8_t *p, const u32x4 &v, int valid)
{
	for (int b = 0; b < valid; b++) p[b] = (uint8_t)(v[b >> 2] >> (8 * (b & 3)));
}

// ---------------------------------------------------------------------------------------------------
// M1 flat: mask by quality over the whole matrix as one byte stream (no row structure involved).
// ---------------------------------------------------------------------------------------------------
// A workgroup takes 16 KiB spans (4 x 256 chunks of 16 bytes); a thread's four loads of each input are issued before
// the first is used, so 8 KiB per wave are in flight, and both the loads and the stores bypass the caches' retention
// (read-once / write-once data).
template <int MODE>
__global__ __launch_bounds__(256) void mask_flat_kernel(const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
                                                        uint8_t *__restrict__ out, int64_t bytes, u32 cl2)
{
#ifndef SK_MASK_UNROLL
#define SK_MASK_UNROLL 4
#endif
	constexpr int kUnroll = SK_MASK_UNROLL;
	const int64_t nchunk = (bytes + 15) >> 4;
	const int64_t nspan = (nchunk + 256 * kUnroll - 1) / (256 * kUnroll);
	for (int64_t sp = blockIdx.x; sp < nspan; sp += gridDim.x) {
		// every wave owns 4 KiB of the span: chunks wave*256 + k*64 + lane
		const int64_t c0 = sp * (256 * kUnroll) + (threadIdx.x >> 6) * (64 * kUnroll) + (threadIdx.x & 63);
		if ((sp + 1) * (256 * kUnroll) * 16 <= bytes) {                     // the whole span is inside the matrix
			u32x4 q[kUnroll], sv[kUnroll];
#pragma unroll
			for (int k = 0; k < kUnroll; k++) {
				const int64_t off = (c0 + k * 64) << 4;
				q[k] = stream_load(qual + off);
				sv[k] = stream_load(seq + off);
			}
#pragma unroll
			for (int k = 0; k < kUnroll; k++) {
				u32x4 o, vq;
				mask_dword4<MODE>(q[k], sv[k], cl2, o, vq);
				stream_store(out + ((c0 + k * 64) << 4), o);
			}
		} else {
			for (int k = 0; k < kUnroll; k++) {
				const int64_t c = c0 + k * 64;
				if (c >= nchunk) break;
				const int64_t off = c << 4;
				u32x4 q, s, o, vq;
				if (off + 16 <= bytes) {
					q = *reinterpret_cast<const u32x4 *>(qual + off);
					s = *reinterpret_cast<const u32x4 *>(seq + off);
					mask_dword4<MODE>(q, s, cl2, o, vq);
					*reinterpret_cast<u32x4 *>(out + off) = o;
				} else {
					int valid = (int)(bytes - off);
					q = load_tail(qual + off, valid);
					s = load_tail(seq + off, valid);
					mask_dword4<MODE>(q, s, cl2, o, vq);
					store_tail(out + off, o, valid);
				}
			}
		}
	}
}

hipError_t launch_mask_flat(const uint8_t *seq, const uint8_t *qual, uint8_t *out, int64_t bytes,
                            const QualConsts &qc, int n_cu, hipStream_t st)
{
	if (bytes <= 0) return hipSuccess;
	int64_t nchunk = (bytes + 15) >> 4;
	int64_t want = (nchunk + 256 * SK_MASK_UNROLL - 1) / (256 * SK_MASK_UNROLL);
	// Fewer waves with more bytes each in flight stream better than many: measured at 16 M x 150 (450 B/read),
	// workgroups per CU 1: 5.8-6.0, 2: 5.6, 3: 5.9-6.0, 4: 5.5, 8: 5.3 TB/s.
	static const int wgs_per_cu = getenv("SK_MASK_WGS") ? atoi(getenv("SK_MASK_WGS")) : 3;
	int grid = (int)(want < (int64_t)n_cu * wgs_per_cu ? want : (int64_t)n_cu * wgs_per_cu);
	switch (qc.mode) {
	case 0: mask_flat_kernel<0><<<grid, 256, 0, st>>>(seq, qual, out, bytes, qc.cl2); break;
	case 1: mask_flat_kernel<1><<<grid, 256, 0, st>>>(seq, qual, out, bytes, qc.cl2); break;
	case 2: mask_flat_kernel<2><<<grid, 256, 0, st>>>(seq, qual, out, bytes, qc.cl2); break;
	case 3: mask_flat_kernel<3><<<grid, 256, 0, st>>>(seq, qual, out, bytes, qc.cl2); break;
	default: mask_flat_kernel<4><<<grid, 256, 0, st>>>(seq, qual, out, bytes, qc.cl2); break;
	}
	return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// T1: trim scan of one row held in the wave's LDS tile image (bytes already turned into
// v = (q-33) mod 256).  src/fasta_trim_by_quality.rs:28-42:
//     total = lowest_total = -50; k = lowest_k = n
//     while k > 0 { k -= 1; total += v[k] - m; if total > 0 {break}
//                   if total < lowest_total { lowest_total = total; lowest_k = k } }
// With j = n - k (bytes consumed), T_j = sum of the last j v's, U_j = T_j - j*m = total + 50:
//     break    <=> U_j > 50        <=> T_j > 50 + j*m          (j*m is wave-uniform: a scalar)
//     update   <=> U_j < lowest_U  (strict: the earliest j wins ties)
// Packing K_j = U_j*2^11 + j turns "strictly smaller U, earliest j" into one signed min; K_0 = 0 is the
// initial state (lowest_total = -50, lowest_k = n).  |U| <= 255*2047 keeps K inside int32.
// ---------------------------------------------------------------------------------------------------
// The running sums inside a dword come from v_dot4_u32_u8 with byte-select multipliers, so they are independent of each
// other (no add chain) and cost one VALU op per byte.
//
// Since K_j = U_j*2^11 + j with j < 2^11, the break condition U_j > 50 is K_j >= 51*2^11 — one constant for every j.
// An 8-byte step is therefore handled whole: its eight keys give Kmin and Kmax (v_min3 / v_max3 trees); while
// Kmax < 51*2^11 nothing broke and Kmin is merged into the running best.  The step in which a lane stops — a key at or
// above the limit, or the row's last, partial step — is left to a replay after the loop, byte by byte; the loop itself
// (trim_scan_packed, SK_SCAN_ASM_*) has no per-byte compare / mask / select chain.
constexpr int kBreakKey = 51 << kKeyBits;
constexpr int kStragglerFrom = SK_STRAGGLER_FROM;   // 8-byte steps before the hand-over is considered
constexpr int kStragglerLeft = SK_STRAGGLER_LEFT;   // ... and at least this many steps of the longest row are left
constexpr int kStragglerRows = SK_STRAGGLER_ROWS;   // hand over when at most this many rows are still being scanned

// ---- scan / reduce inside groups of 8 lanes, on DPP (gfx9 control row_shr:n = 0x110+n; a row is 16 lanes = 2 groups) ----
// `u` is the lane's index in its group; a lane whose source would lie outside the group takes `idle` instead.
__device__ __forceinline__ int group8_inclusive_sum(int x, int u)
{
	int t;
	t = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false); x += u >= 1 ? t : 0;
	t = __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false); x += u >= 2 ? t : 0;
	t = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false); x += u >= 4 ? t : 0;
	return x;
}
__device__ __forceinline__ int group8_min(int x)       // every lane of the group ends with the group's minimum (xor butterfly: 1, 2, mirror)
{
	x = min(x, __builtin_amdgcn_update_dpp(x, x, 0xb1, 0xf, 0xf, false));      // quad_perm [1,0,3,2]
	x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x4e, 0xf, 0xf, false));      // quad_perm [2,3,0,1]
	x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 0xf, false));     // row_half_mirror: the other quad of the group
	return x;
}

// Stragglers.  The lane-per-row loop below costs the same whether 64 rows are still being scanned or one, and one read
// of a tile that never breaks (a read of '#', say — every run has some) keeps the whole wave in it for all ceil(L/8)
// steps.  When at most 8 rows are left they are finished by the whole wave together: row g goes to lanes 8g..8g+7, lane
// u of the group takes the 4 bytes u-th next in scan order (32 bytes of every row per round), the running sums come from
// a prefix sum inside the group, the first lane of a group that sees the break (or the row's end) cuts the range, and the
// minimum key before the cut is a minimum over the group.  ~60 VALU per round for all rows, against ~30 per 8-byte step.
// In: the rows of `todo` (<= 8) have consumed `j0` bytes each (whole steps, none of them stopped); T / best are the
// per-lane running sum and best key.  Out: best of those lanes (src/fasta_trim_by_quality.rs:33-41 for the rest of the row).
template <bool RAW = false>
__device__ __forceinline__ void trim_finish_stragglers(const uint8_t *tile, unsigned long long todo, int j0, int step, int lane,
                                                       int end, int len, u32 T, int &best)
{
	const int g = lane >> 3, u = lane & 7;
	// which row does this lane's group work on: the g-th set bit of todo (groups beyond the number of rows idle)
	int row = -1;
	{
		unsigned long long m = todo;
		for (int k = 0; k < 8 && m; k++) {                        // scalar loop
			const int r = (int)__builtin_ctzll(m);
			m &= m - 1;
			row = g == k ? r : row;
		}
	}
	const bool have = row >= 0;
	const int src = (have ? row : lane) << 2;                     // ds_bpermute addresses lanes in bytes
	const int end_r = __builtin_amdgcn_ds_bpermute(src, end), len_r = __builtin_amdgcn_ds_bpermute(src, len);
	u32 T_r = (u32)__builtin_amdgcn_ds_bpermute(src, (int)T);
	int best_r = 0x7fffffff;
	const u32 sh = (u32)end_r & 3u;                               // (end_r - j) & 3 for every j that is a multiple of 4
	bool going = have;
	for (int jr = j0; __ballot(going && jr < len_r) != 0ull; jr += 32) {
		const int e = end_r - jr - 4 * u;                          // this lane's 4 bytes: addresses [e-4, e), scanned downwards
		const int a = max(e & ~3, 0);                              // lanes past the row's start read the front of the image: masked below
		const u32 hi = *reinterpret_cast<const u32 *>(tile + a), lo = *reinterpret_cast<const u32 *>(tile + a - 4);
		u32 d = __builtin_amdgcn_alignbyte(hi, lo, sh);
		if (RAW) d = sub33(d, (d & kLo7) + 0x5f5f5f5fu);             // the image holds raw quality bytes
		const int s0 = (int)__builtin_amdgcn_udot4(d, 0x01000000u, 0u, false), s1 = (int)__builtin_amdgcn_udot4(d, 0x01010000u, 0u, false);
		const int s2 = (int)__builtin_amdgcn_udot4(d, 0x01010100u, 0u, false), s3 = (int)__builtin_amdgcn_udot4(d, 0x01010101u, 0u, false);
		const int incl = group8_inclusive_sum(s3, u);
		const u32 Tb = T_r + (u32)(incl - s3);                     // sum before this lane's first byte
		const int j1 = jr + 4 * u + 1;                             // bytes consumed after this lane's first byte
		const int c1 = j1 * step;
		const int K0 = (int)((Tb + (u32)s0) << kKeyBits) + c1, K1 = (int)((Tb + (u32)s1) << kKeyBits) + (c1 + step);
		const int K2 = (int)((Tb + (u32)s2) << kKeyBits) + (c1 + 2 * step), K3 = (int)((Tb + (u32)s3) << kKeyBits) + (c1 + 3 * step);
		// the scan stops at the first byte past the row's end or whose total is positive (:36); keys from there on do not count
		const bool ok0 = going && j1 <= len_r && K0 < kBreakKey;
		const bool ok1 = ok0 && j1 + 1 <= len_r && K1 < kBreakKey;
		const bool ok2 = ok1 && j1 + 2 <= len_r && K2 < kBreakKey;
		const bool ok3 = ok2 && j1 + 3 <= len_r && K3 < kBreakKey;
		const u32 stops = (u32)(__ballot(!ok3) >> (8 * g)) & 0xffu;           // the lanes of this group that hold a stop
		const int f = stops ? (int)__builtin_ctz(stops) : 8;                 // the first of them
		int k = min(min(ok0 ? K0 : 0x7fffffff, ok1 ? K1 : 0x7fffffff), min(ok2 ? K2 : 0x7fffffff, ok3 ? K3 : 0x7fffffff));
		k = u <= f ? k : 0x7fffffff;
		best_r = min(best_r, k);
		going = going && stops == 0u;
		T_r += (u32)__builtin_amdgcn_ds_bpermute((lane | 7) << 2, incl);      // the group's total of this round
	}
	best_r = group8_min(best_r);                                                // row g's minimum over all rounds, in every lane of group g
	{
		unsigned long long m = todo;
		for (int k = 0; m; k++) {
			const int r = (int)__builtin_ctzll(m);
			m &= m - 1;
			const int v = __builtin_amdgcn_readlane(best_r, 8 * k);
			best = lane == r ? min(best, v) : best;
		}
	}
}

// The scan loop of trim_scan_packed (see there), two 8-byte steps per iteration.  v_cmpx: exec &= no key of the step is at
// or above the limit.  The two steps use the two register pairs in turn — the dword a step's upper v_alignbyte needs is
// the lower one of the pair the step before consumed, so nothing is moved — and their sixteen key constants are relative
// to the iteration's first byte: the best key and the limit move on once per iteration.  The tail is the hand-over test
// (few rows left, past where reads usually break).  The wait at the end is for the loads the loop issued last: the
// compiler does not know of them and would hand their registers to something else while they are still on their way.
// Registers by name: v[126:127] / v[124:125] the two pairs (ds_read2_b32 needs a pair and an asm operand cannot name its
// halves); s[80:83] the byte-select multipliers, s79 = 0x21212121, s[84:99] the key constants 1 .. 16 times `step`.
#define SK_SCAN_ASM_PROLOGUE \
	"s_mov_b64 %[sv], exec\n\t" \
	"ds_read2_b32 v[126:127], %[q] offset0:2 offset1:3\n\t" \
	"ds_read_b32 v124, %[q] offset:16\n\t" \
	"s_mov_b32 s80, 0x01000000\n\t" \
	"s_mov_b32 s81, 0x01010000\n\t" \
	"s_mov_b32 s82, 0x01010100\n\t" \
	"s_mov_b32 s83, 0x01010101\n\t" \
	"s_mov_b32 s79, 0x21212121\n\t" \
	"s_mov_b32 s84, %[step]\n\t" \
	"s_add_i32 s85, s84, %[step]\n\t" \
	"s_add_i32 s86, s85, %[step]\n\t" \
	"s_add_i32 s87, s86, %[step]\n\t" \
	"s_add_i32 s88, s87, %[step]\n\t" \
	"s_add_i32 s89, s88, %[step]\n\t" \
	"s_add_i32 s90, s89, %[step]\n\t" \
	"s_add_i32 s91, s90, %[step]\n\t" \
	"s_add_i32 s92, s91, %[step]\n\t" \
	"s_add_i32 s93, s92, %[step]\n\t" \
	"s_add_i32 s94, s93, %[step]\n\t" \
	"s_add_i32 s95, s94, %[step]\n\t" \
	"s_add_i32 s96, s95, %[step]\n\t" \
	"s_add_i32 s97, s96, %[step]\n\t" \
	"s_add_i32 s98, s97, %[step]\n\t" \
	"s_add_i32 s99, s98, %[step]\n\t" \
	"s_mov_b32 %[lim], %[brk]\n\t" \
	"s_mov_b32 %[jj], 0\n\t" \
	"s_mov_b32 %[jj1], 1\n\t" \
	"s_mov_b64 %[strag], 0\n" \
	"1:\n\t"
// one step: HI / L1 / L0 the three dwords it scans (descending addresses), PAIR where the next step's two are loaded,
// C1 .. C8 its key constants; RAWCHECK = SK_SCAN_ASM_RAWCHECK or nothing
#define SK_SCAN_ASM_STEP(HI, L1, L0, PAIR, C1, C2, C3, C4, C5, C6, C7, C8, RAWCHECK) \
	"s_waitcnt lgkmcnt(0)\n\t" \
	"v_alignbyte_b32 %[d1], " HI ", " L1 ", %[sh]\n\t" \
	"v_alignbyte_b32 %[d0], " L1 ", " L0 ", %[sh]\n\t" \
	"ds_read2_b32 " PAIR ", %[q] offset1:1\n\t" \
	"v_dot4_u32_u8 %[k0], %[d1], s80, %[T]\n\t" \
	"v_dot4_u32_u8 %[k1], %[d1], s81, %[T]\n\t" \
	"v_dot4_u32_u8 %[k2], %[d1], s82, %[T]\n\t" \
	"v_dot4_u32_u8 %[k3], %[d1], s83, %[T]\n\t" \
	"v_dot4_u32_u8 %[k4], %[d0], s80, %[k3]\n\t" \
	"v_dot4_u32_u8 %[k5], %[d0], s81, %[k3]\n\t" \
	"v_dot4_u32_u8 %[k6], %[d0], s82, %[k3]\n\t" \
	"v_dot4_u32_u8 %[T], %[d0], s83, %[k3]\n\t" \
	"v_lshl_add_u32 %[k0], %[k0], 11, " C1 "\n\t" \
	"v_lshl_add_u32 %[k1], %[k1], 11, " C2 "\n\t" \
	"v_lshl_add_u32 %[k2], %[k2], 11, " C3 "\n\t" \
	"v_lshl_add_u32 %[k3], %[k3], 11, " C4 "\n\t" \
	"v_lshl_add_u32 %[k4], %[k4], 11, " C5 "\n\t" \
	"v_lshl_add_u32 %[k5], %[k5], 11, " C6 "\n\t" \
	"v_lshl_add_u32 %[k6], %[k6], 11, " C7 "\n\t" \
	"v_lshl_add_u32 %[k7], %[T], 11, " C8 "\n\t" \
	RAWCHECK \
	"v_max3_i32 %[d1], %[k0], %[k1], %[k2]\n\t" \
	"v_max3_i32 %[d0], %[k3], %[k4], %[k5]\n\t" \
	"v_max_i32 %[kx], %[k6], %[k7]\n\t" \
	"v_max3_i32 %[d1], %[d1], %[d0], %[kx]\n\t" \
	"v_cmpx_gt_i32_e32 vcc, %[lim], %[d1]\n\t" \
	"s_cbranch_execz 2f\n\t" \
	"v_min3_i32 %[k0], %[k0], %[k1], %[k2]\n\t" \
	"v_min3_i32 %[k3], %[k3], %[k4], %[k5]\n\t" \
	"v_min3_i32 %[k6], %[k6], %[k7], %[best]\n\t" \
	"v_min3_i32 %[best], %[k0], %[k3], %[k6]\n\t" \
	"v_add_u32_e32 %[q], -8, %[q]\n\t"
#define SK_SCAN_ASM_RAWCHECK \
	"v_sad_u8 %[A], %[d1], s79, %[A]\n\t" \
	"v_sad_u8 %[A], %[d0], s79, %[A]\n\t"
#define SK_SCAN_ASM_STEP_A(RAWCHECK) SK_SCAN_ASM_STEP("v124", "v127", "v126", "v[124:125]", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", RAWCHECK)
#define SK_SCAN_ASM_STEP_B(RAWCHECK) SK_SCAN_ASM_STEP("v126", "v125", "v124", "v[126:127]", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", RAWCHECK)
// a ragged row leaves at its last, partial step (or its end): that step is replayed
#define SK_SCAN_ASM_RAGGED_TOP(JJ) \
	"v_cmpx_lt_i32_e32 vcc, " JJ ", %[mf]\n\t" \
	"s_cbranch_execz 2f\n\t"
// rows of one length: an odd number of whole steps ends after a first half
#define SK_SCAN_ASM_UNIFORM_MID \
	"s_cmp_ge_i32 %[jj1], %[nfull]\n\t" \
	"s_cbranch_scc1 2f\n\t"
#define SK_SCAN_ASM_TAIL \
	"v_subrev_u32_e32 %[best], s99, %[best]\n\t" \
	"s_sub_i32 %[lim], %[lim], s99\n\t" \
	"s_add_i32 %[jj], %[jj], 2\n\t" \
	"s_add_i32 %[jj1], %[jj1], 2\n\t" \
	"s_cmp_ge_i32 %[jj], %[nfull]\n\t" \
	"s_cbranch_scc1 2f\n\t" \
	"s_sub_i32 %[tmp], %[jj], %[from1]\n\t" \
	"s_cmp_ge_u32 %[tmp], %[span]\n\t" \
	"s_cbranch_scc1 1b\n\t" \
	"s_bcnt1_i32_b64 %[tmp], exec\n\t" \
	"s_cmp_gt_i32 %[tmp], %[rows]\n\t" \
	"s_cbranch_scc1 1b\n\t" \
	"s_mov_b64 %[strag], exec\n" \
	"2:\n\t" \
	"s_mov_b64 exec, %[sv]\n\t" \
	"s_waitcnt lgkmcnt(0)\n\t"
#define SK_SCAN_ASM_OPERANDS \
	: [T] "+v"(T), [best] "+v"(best), [q] "+v"(q), [A] "+v"(A), [d1] "=&v"(d1), [d0] "=&v"(d0), [k0] "=&v"(k0), [k1] "=&v"(k1), \
	[k2] "=&v"(k2), [k3] "=&v"(k3), [k4] "=&v"(k4), [k5] "=&v"(k5), [k6] "=&v"(k6), [k7] "=&v"(k7), [kx] "=&v"(kx), \
	[lim] "=&s"(lim), [jj] "=&s"(jj), [jj1] "=&s"(jj1), [tmp] "=&s"(tmp), [sv] "=&s"(sv), [strag] "=&s"(strag) \
	: [sh] "v"(sh), [mf] "v"(my_full), [step] "s"(__builtin_amdgcn_readfirstlane(step)), [nfull] "s"(__builtin_amdgcn_readfirstlane(nfull)), \
	[span] "s"(__builtin_amdgcn_readfirstlane((int)strag_span)), [brk] "n"(kBreakKey), [from1] "n"(kStragglerFrom + 1), [rows] "n"(kStragglerRows) \
	: "memory", "vcc", "scc", "v124", "v125", "v126", "v127", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", \
	"s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99"

// RAW: the image holds the quality bytes as they came from memory, not v = (q - 33) mod 256 (the trim-alone pass: its
// stream phase then has no arithmetic at all).  While no byte of a row is below 33, v = q - 33 and the keys are the same
// with m + 33 in the place of m; whether that held is checked on the way — sum |q - 33| against sum q - 33 j, two v_sad_u8
// per step — and a row that fails (a byte below '!': nothing a sequencer writes) is scanned again byte by byte.
template <bool UNIFORM_LEN, bool RAW = false>
__device__ __forceinline__ int trim_scan_packed(const uint8_t *tile, int row_start, int len, int maxlen, int m, bool active)
{
	// EIGHT bytes per step: the two dwords come from one LDS instruction (ds_read2_b32), their eight keys give one
	// minimum and one maximum, and the step's bookkeeping is paid once per eight bytes.
	// The loop is what a pass over long scans is bound by — by the instructions it issues, vector AND scalar
	// (EXPERIMENTS.md A.4, A.5) — so it is a plain divergent loop: a lane whose step holds a break (or whose row has no whole step left) LEAVES
	// it, and the hardware's execution mask does what selects and ballots did before (a lane that is out writes nothing:
	// its sum and its best key stay what they were; "is anybody left" is the loop's own exec test).  The step a lane
	// stopped in is replayed byte by byte after the loop.
	// Keys are built with the SAME constants in every iteration — K' = K minus the key offset of the iteration's first byte
	// — the running best is kept relative as well (one subtraction per iteration moves it on) and the break limit is a
	// scalar that moves with it.  |K'| < 2^31 for rows of up to 960 bytes.
	const int end = row_start + len;
	const u32 sh = (u32)end & 3u;
	const int step_true = 1 - m * (1 << kKeyBits);            // C_j = j - j*m*2^11 = j * step
	const int step = RAW ? step_true - 33 * (1 << kKeyBits) : step_true;     // the loop's: on raw bytes m + 33 stands for m (|K'| < 2^31 still)
	const int step8 = 8 * step;
	const int nfull = maxlen >> 3;                            // whole steps of the longest row (wave-uniform)
	const int nst = (maxlen + 7) >> 3;
	u32 T = 0;
	u32 A = 0;                                                // RAW: sum of |q - 33| over the bytes T covers
	int best = 0;                                             // relative to the step the lane is in
	// Every lane runs the loop, rows past the end of the last tile too (their image is whatever the tile before left
	// there and their result is clipped by the store).  No address is clamped: a lane only reads one step past the last one
	// it executes, i.e. at most 11 bytes before its row — the previous row of the image, or the pad in front of it
	// (kLdsPad) for row 0.
	//
	// The loop is written out in gfx950 assembly (SK_SCAN_ASM_* above): what the compiler made of the same loop in C++ was
	// 33 vector and 22 scalar instructions per step (the structurizer's mask bookkeeping, a 64-bit register move for the
	// prefetched pair, a per-lane step counter); this is 30.5 and 5 (8 inside the hand-over window), two more vector ones
	// on raw bytes.  v_cmpx narrows exec to the lanes the step did not stop, s_cbranch_execz is "nobody left", the lane's
	// count of whole steps is read off its LDS pointer afterwards (only survivors move it on), and the next step's two
	// dwords are loaded into the registers the v_alignbyte pair has just consumed.
	(void)active;
	static_assert(kLdsPad >= 16, "the scan reads up to 16 bytes in front of the tile image");
	const int my_full = len >> 3;
	// hand-over window as one unsigned compare: kStragglerFrom <= jj < nst - kStragglerLeft for the step jj just completed
	const u32 strag_span = nst - kStragglerLeft > kStragglerFrom ? (u32)(nst - kStragglerLeft - kStragglerFrom) : 0u;
	const u32 q0 = (u32)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)(tile + (end & ~3) - 16);
	u32 q = q0;
	unsigned long long strag = 0ull;                          // rows handed to trim_finish_stragglers
	if (nfull > 0) {
		u32 d1, d0, k0, k1, k2, k3, k4, k5, k6, k7, kx;
		int lim, jj, jj1, tmp;
		unsigned long long sv;
		// (the four forms differ in two places: a ragged row leaves at the top of a step; raw bytes are checked)
		if (UNIFORM_LEN && !RAW)
			asm volatile(SK_SCAN_ASM_PROLOGUE SK_SCAN_ASM_STEP_A("") SK_SCAN_ASM_UNIFORM_MID SK_SCAN_ASM_STEP_B("") SK_SCAN_ASM_TAIL SK_SCAN_ASM_OPERANDS);
		else if (UNIFORM_LEN)
			asm volatile(SK_SCAN_ASM_PROLOGUE SK_SCAN_ASM_STEP_A(SK_SCAN_ASM_RAWCHECK) SK_SCAN_ASM_UNIFORM_MID SK_SCAN_ASM_STEP_B(SK_SCAN_ASM_RAWCHECK) SK_SCAN_ASM_TAIL SK_SCAN_ASM_OPERANDS);
		else if (!RAW)
			asm volatile(SK_SCAN_ASM_PROLOGUE SK_SCAN_ASM_RAGGED_TOP("%[jj]") SK_SCAN_ASM_STEP_A("") SK_SCAN_ASM_RAGGED_TOP("%[jj1]") SK_SCAN_ASM_STEP_B("") SK_SCAN_ASM_TAIL SK_SCAN_ASM_OPERANDS);
		else
			asm volatile(SK_SCAN_ASM_PROLOGUE SK_SCAN_ASM_RAGGED_TOP("%[jj]") SK_SCAN_ASM_STEP_A(SK_SCAN_ASM_RAWCHECK) SK_SCAN_ASM_RAGGED_TOP("%[jj1]") SK_SCAN_ASM_STEP_B(SK_SCAN_ASM_RAWCHECK) SK_SCAN_ASM_TAIL SK_SCAN_ASM_OPERANDS);
	}
	const int ws = (int)(q0 - q) >> 3;                        // whole steps this lane got through: only survivors of a step move q on
	const bool handed = (strag >> (threadIdx.x & (kWave - 1))) & 1ull;
	const bool broke = !handed && ws < (UNIFORM_LEN ? nfull : my_full);     // stopped at a break: the stop step's eight bytes are in its sum already
	bool wrapped = false;
	if (RAW) {                                                // from sums of q to sums of v, if the row allows it
		const u32 off = 33u * 8u * (u32)(ws + (broke ? 1 : 0));
		wrapped = A != T - off;
		T -= off;
	}
	best += (ws & ~1) * step8;                                // the true key again: it was relative to the iteration's first byte
	if (strag) {
		const int strag_j0 = 8 * __builtin_amdgcn_readlane(ws, (int)__builtin_ctzll(strag));
		trim_finish_stragglers<RAW>(tile, strag, strag_j0, step_true, (int)(threadIdx.x & (kWave - 1)), end, len, T, best);
	}
	if (!handed) {                                            // replay the step the lane stopped in: src/fasta_trim_by_quality.rs:33-41 byte by byte
		const uint8_t *qp = tile + ((end & ~3) - 8 * (ws + 1));
		const u32 h2 = *reinterpret_cast<const u32 *>(qp + 8), l1 = *reinterpret_cast<const u32 *>(qp + 4), l0 = *reinterpret_cast<const u32 *>(qp);
		u32 stop_d1 = __builtin_amdgcn_alignbyte(h2, l1, sh), stop_d0 = __builtin_amdgcn_alignbyte(l1, l0, sh);
		if (RAW) { stop_d1 = sub33(stop_d1, (stop_d1 & kLo7) + 0x5f5f5f5fu); stop_d0 = sub33(stop_d0, (stop_d0 & kLo7) + 0x5f5f5f5fu); }
		u32 t = T;
		if (broke) t -= __builtin_amdgcn_udot4(stop_d1, 0x01010101u, __builtin_amdgcn_udot4(stop_d0, 0x01010101u, 0u, false), false);
#pragma unroll
		for (int i = 0; i < 8; i++) {
			const int j = 8 * ws + i + 1;
			t += ((i < 4 ? stop_d1 : stop_d0) >> (8 * (3 - (i & 3)))) & 0xFFu;
			const int K = (int)(t << kKeyBits) + j * step_true;
			if (j > len || K >= kBreakKey) break;
			best = min(best, K);
		}
	}
	int low_j = best & ((1 << kKeyBits) - 1);
	if (RAW && __builtin_amdgcn_ballot_w64(wrapped) != 0ull) {
		if (wrapped) {                                        // src/fasta_trim_by_quality.rs:28-42 as it stands, on this row alone
			int total = 0, low = 0;
			low_j = 0;
			for (int j = 1; j <= len; j++) {
				total += (int)((tile[end - j] - 33u) & 0xFFu) - m;
				if (total > 50) break;
				if (total < low) { low = total; low_j = j; }
			}
		}
	}
	return len - low_j;
}

// same scan with separate (lowest_U, lowest_j); for rows longer than 2047 bytes
template <bool RAW = false>
__device__ __forceinline__ int trim_scan_wide(const uint8_t *tile, int row_start, int len, int maxlen, int m, bool active)
{
	const int end = row_start + len;
	const u32 sh = (u32)end & 3u;
	int a = end & ~3;
	u32 hi = *reinterpret_cast<const u32 *>(tile + a);
	int T = 0, lowU = 0, lowj = 0;
	bool alive = active;
	const int ndw = (maxlen + 3) >> 2;
	for (int jj = 0; jj < ndw; jj++) {
		a = max(a - 4, -4);
		u32 lo = *reinterpret_cast<const u32 *>(tile + a);
		u32 d = __builtin_amdgcn_alignbyte(hi, lo, sh);
		if (RAW) d = sub33(d, (d & kLo7) + 0x5f5f5f5fu);
		hi = lo;
#pragma unroll
		for (int i = 3; i >= 0; i--) {
			const int j = 4 * jj + (4 - i);
			const int jm = j * m;
			T += (int)((d >> (8 * i)) & 0xffu);
			alive = alive && (j <= len) && (T <= 50 + jm);
			int U = T - jm;
			bool upd = alive && (U < lowU);
			lowU = upd ? U : lowU;
			lowj = upd ? j : lowj;
		}
		if (__ballot(alive) == 0ull) break;
	}
	return len - lowj;
}

// ---------------------------------------------------------------------------------------------------
// D1+D2: masked Hamming distance against every sheet barcode, first/last argmin.
// src/fasta_demultiplex.rs:269-277 (barcode_diff) and :154-166 (best / equally_fine).
//
// One-hot form.  Per barcode position k the sheet uses at most 7 distinct non-wildcard bytes; class c of
// position k gets bit c.  Observed byte b at k is re-coded as (class bit of b at k, or 0 if the sheet
// never uses b there) | 0x80; a candidate byte is its class bit, or 0x80 for the wildcards 'N'/'U'.
// Then popcount(obs_code & cand_code) == 1 exactly when the position does NOT count as a mismatch
// (equal bytes, or wildcard), so mismatches = L - popcount over the W code dwords.
// first argmin = max over s of (matches<<16 | 0xffff-s); last argmin = max of (matches<<16 | s).
// The candidate codes sit in the workgroup's LDS (one copy for its 8 waves, rows padded to WP dwords so
// the wave-uniform reads are aligned ds_read_b128/b64 broadcasts).
// ---------------------------------------------------------------------------------------------------
template <int W, int WP>
__device__ __forceinline__ void match_onehot(const u32 *cand_lds, int S, const u32 (&o)[W], u32 &keyF, u32 &keyL)
{
	keyF = 0u; keyL = 0u;
#pragma unroll 4
	for (int s = 0; s < S; s++) {
		const u32 *c = cand_lds + s * WP;
		u32 pc = 0;
#pragma unroll
		for (int w = 0; w < W; w++) pc += (u32)__builtin_popcount(o[w] & c[w]);
		keyF = max(keyF, (pc << 16) | (0xffffu - (u32)s));
		keyL = max(keyL, (pc << 16) | (u32)s);
	}
}

__host__ __device__ constexpr int padded_w(int w) { return w <= 1 ? 1 : w <= 2 ? 2 : w <= 4 ? 4 : 8; }

template <int W>
__device__ __forceinline__ void demux_row_onehot(const uint8_t *tile, int row_start, const BarcodeDev &t, const u32 *cand_lds,
                                                 int &diff, int &first, int &last)
{
	// raw observed bytes: W dwords starting at a misaligned LDS address
	const u32 sh = (u32)row_start & 3u;
	int a = row_start & ~3;
	u32 lo = *reinterpret_cast<const u32 *>(tile + a);
	u32 o[W];
#pragma unroll
	for (int w = 0; w < W; w++) {
		u32 hi = *reinterpret_cast<const u32 *>(tile + a + 4 * (w + 1));
		u32 raw = __builtin_amdgcn_alignbyte(hi, lo, sh);
		lo = hi;
		u32 code = 0;
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const int k = 4 * w + i;
			if (k < t.L) code |= (u32)t.lut[k * 256 + (int)((raw >> (8 * i)) & 0xffu)] << (8 * i);
		}
		o[w] = code;
	}
	u32 keyF, keyL;
	match_onehot<W, padded_w(W)>(cand_lds, t.S, o, keyF, keyL);
	diff = t.L - (int)(keyF >> 16);
	first = (int)(0xffffu - (keyF & 0xffffu));
	last = (int)(keyL & 0xffffu);
}

// ---------------------------------------------------------------------------------------------------
// Bit-sliced form of D1+D2 (the fast path).  Instead of walking the S candidates per read, walk the L
// positions: MM[k][class of the observed byte at k] is an S-bit vector with bit s set when candidate s
// counts a mismatch there (its byte is not a wildcard and differs).  The per-candidate mismatch counts
// are then the column sums of L such vectors, kept as 5 bit planes per 32 candidates and accumulated
// three positions at a time with carry-save adders (v_xor / v_bfi / v_and).  The minimum and its first /
// last candidate come out of the planes MSB-first: keep the candidates whose plane bit is 0 when any
// such candidate survives.  ~13 VALU ops per 3 positions per 32 candidates instead of ~14 per candidate.
// ---------------------------------------------------------------------------------------------------
#define SK_MAJ(a, b, c) ((((a) ^ (b)) & (c)) | (~((a) ^ (b)) & (a)))      /* -> v_xor + v_bfi */

template <int GB>
__device__ __forceinline__ void demux_row_bitsliced(const uint8_t *row, const uint8_t *bs, int mm_off, int G, int g0, int L,
                                                    int &best, int &first, int &last)
{
	const uint8_t *cls = bs;
	const u32 *valid = reinterpret_cast<const u32 *>(bs + 256);
	const u32 *mm = reinterpret_cast<const u32 *>(bs + mm_off);
	u32 P0[GB], P1[GB], P2[GB], P3[GB], P4[GB];
#pragma unroll
	for (int g = 0; g < GB; g++) P0[g] = P1[g] = P2[g] = P3[g] = P4[g] = 0u;
	for (int k = 0; k < L; k += 3) {
		const u32 *r0 = mm + ((k * 8 + (int)cls[row[k]]) * G + g0);
		const bool h1 = k + 1 < L, h2 = k + 2 < L;                       // wave-uniform
		const u32 *r1 = mm + (((k + 1) * 8 + (int)cls[row[h1 ? k + 1 : k]]) * G + g0);
		const u32 *r2 = mm + (((k + 2) * 8 + (int)cls[row[h2 ? k + 2 : k]]) * G + g0);
#pragma unroll
		for (int g = 0; g < GB; g++) {
			const u32 x0 = r0[g], x1 = h1 ? r1[g] : 0u, x2 = h2 ? r2[g] : 0u;
			const u32 sm = x0 ^ x1 ^ x2, cr = SK_MAJ(x0, x1, x2);          // weights 1 and 2
			const u32 c0 = P0[g] & sm;  P0[g] ^= sm;
			const u32 c1 = SK_MAJ(P1[g], c0, cr);  P1[g] ^= c0 ^ cr;
			const u32 c2 = P2[g] & c1;  P2[g] ^= c1;
			const u32 c3 = P3[g] & c2;  P3[g] ^= c2;
			P4[g] ^= c3;
		}
	}
#pragma unroll
	for (int g = 0; g < GB; g++) {
		u32 cand = valid[g0 + g];
		int d = 0;
		u32 t;
		t = cand & ~P4[g]; d = 2 * d + (t ? 0 : 1); cand = t ? t : cand;
		t = cand & ~P3[g]; d = 2 * d + (t ? 0 : 1); cand = t ? t : cand;
		t = cand & ~P2[g]; d = 2 * d + (t ? 0 : 1); cand = t ? t : cand;
		t = cand & ~P1[g]; d = 2 * d + (t ? 0 : 1); cand = t ? t : cand;
		t = cand & ~P0[g]; d = 2 * d + (t ? 0 : 1); cand = t ? t : cand;
		const int f = (g0 + g) * 32 + __builtin_ctz(cand);
		const int l = (g0 + g) * 32 + 31 - __builtin_clz(cand);
		if (d < best) { best = d; first = f; last = l; }
		else if (d == best) last = l;
	}
}

// byte-for-byte form: any sheet alphabet, any length; sheet bytes come from the LDS copy
__device__ __forceinline__ void demux_row_bytes(const uint8_t *tile, int row_start, const BarcodeDev &t, const uint8_t *raw_lds,
                                                int &diff, int &first, int &last)
{
	int lowest = 0x7fffffff; first = 0; last = 0;
	for (int s = 0; s < t.S; s++) {
		const uint8_t *c = raw_lds + s * t.L;
		int d = 0;
		for (int k = 0; k < t.L; k++) {
			uint8_t cb = c[k];
			if (cb == 'N' || cb == 'U') continue;
			d += (tile[row_start + k] != cb) ? 1 : 0;
		}
		if (d < lowest) { lowest = d; first = s; last = s; }
		else if (d == lowest) last = s;
	}
	diff = lowest;
}

// ---------------------------------------------------------------------------------------------------
// The tile pass: one wavefront per 64-row tile, persistent over tiles; a workgroup is NW wavefronts that
// share nothing but the read-only matcher tables and the per-sample histogram in LDS.
// ---------------------------------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) uint8_t sk_smem[];

// Hand-off between the lanes of ONE wave through its private LDS tile.  DS operations of a wave are
// executed in issue order, so no s_barrier is needed (and none is wanted: the waves of a workgroup must
// drift apart so that their stream / scan / match phases overlap); the fence only pins compiler order.
__device__ __forceinline__ void wave_lds_fence()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct LdsPlan {
	int table_bytes;     // matcher tables (bit-sliced blob, one-hot codes or raw sheet bytes), rounded to 16
	int hist_off;        // u32[S+3]
	int tiles_off;       // first wave's tile slot (includes the front pad)
	int tile_slot;       // bytes per wave: front pad + 64*row_bytes rounded to 16 + back pad
	int use_lds_hist;
};

// copy the matcher tables into the workgroup's LDS and clear its histogram (all threads; ends with a barrier)
__device__ __forceinline__ void stage_tables(const BarcodeDev &t, const LdsPlan &lp, u32 *hist)
{
	if (t.bs != nullptr) {
		const u32 *src = reinterpret_cast<const u32 *>(t.bs);
		u32 *dst = reinterpret_cast<u32 *>(sk_smem);
		for (int i = threadIdx.x; i < (t.bs_bytes >> 2); i += blockDim.x) dst[i] = src[i];
	} else if (t.onehot != nullptr) {
		const int wp = padded_w(t.W);
		u32 *dst = reinterpret_cast<u32 *>(sk_smem);
		for (int i = threadIdx.x; i < t.S * wp; i += blockDim.x) {
			const int s = i / wp, w = i - s * wp;
			dst[i] = (w < t.W) ? t.onehot[s * t.W + w] : 0u;
		}
	} else {
		for (int i = threadIdx.x; i < t.S * t.L; i += blockDim.x) sk_smem[i] = t.raw[i];
	}
	if (lp.use_lds_hist)
		for (int i = threadIdx.x; i < t.S + 3; i += blockDim.x) hist[i] = 0u;
	__syncthreads();
}

// D3 (src/fasta_demultiplex.rs:168-194) + outputs + counters for one lane's read
struct WaveCounts { u32 total, ident, ambig; };

__device__ __forceinline__ void demux_commit(const TileArgs &a, const LdsPlan &lp, u32 *hist, int64_t r, bool active,
                                             int diff, int first, int last, WaveCounts &wc)
{
	const int S = a.table.S;
	int code = kAssignNone;
	if (S > 0 && diff <= a.table.max_diff) code = (first == last) ? first : kAssignAmbiguous;
	if (active) {
		a.assign[r] = code;
		if (a.lowest_diff) a.lowest_diff[r] = (uint8_t)(diff > 255 ? 255 : diff);
		if (a.first_idx) a.first_idx[r] = (int16_t)first;
		if (a.last_idx) a.last_idx[r] = (int16_t)last;
		if (code >= 0) {
			if (lp.use_lds_hist) atomicAdd(&hist[code], 1u);
			else atomicAdd(&a.counts[code], 1ull);
		}
	}
	wc.total += (u32)__popcll(__ballot(active));
	wc.ident += (u32)__popcll(__ballot(active && code >= 0));
	wc.ambig += (u32)__popcll(__ballot(active && code == kAssignAmbiguous));
}

// The lookup kernels keep no per-wave `identified` / `ambiguous` tallies: two ballots and two popcounts per tile were 5-7 % of
// their time.  An ambiguous row adds to hist[S + 2] where an identified one adds to its sample; identified = the sum of the
// samples' bins, taken by the kernel that folds the counter copies behind the launch (counts_fold_wide_kernel,
// counts_fold_kernel) — or here, when the workgroup ends, if the launch adds to the caller's vector directly (a barrier and
// a reduction at the end of every workgroup: 0.7 us of cfg 3's 27 at 10 M rows, which is why the folds do it).
static_assert(kLutMaxSamples + 3 <= kMaxLdsHist, "a sheet the table serves has its histogram in LDS, and a fold kernel behind every launch that spreads its counters");
__device__ __forceinline__ void lut_identified_from_hist(int S, u32 *hist, int lane)
{
	__syncthreads();
	u32 sum = 0;
	for (int i = threadIdx.x; i < S; i += blockDim.x) sum += hist[i];
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) sum += (u32)__shfl_xor((int)sum, d);
	if (lane == 0 && sum) atomicAdd(&hist[S + 1], sum);
}

// shift: counter i lives at counts[i << shift] (4: one 128-byte line per counter, TileArgs::counts_shift)
__device__ __forceinline__ void flush_counts(int S, unsigned long long *counts, const LdsPlan &lp, u32 *hist, int lane, const WaveCounts &wc, int shift = 0)
{
	if (lp.use_lds_hist) {
		if (lane == 0) {
			if (wc.total) atomicAdd(&hist[S], wc.total);
			if (wc.ident) atomicAdd(&hist[S + 1], wc.ident);
			if (wc.ambig) atomicAdd(&hist[S + 2], wc.ambig);
		}
		__syncthreads();
		for (int i = threadIdx.x; i < S + 3; i += blockDim.x) {
			u32 c = hist[i];
			if (c) atomicAdd(&counts[(size_t)i << shift], (unsigned long long)c);
		}
	} else if (lane == 0) {
		if (wc.total) atomicAdd(&counts[(size_t)S << shift], (unsigned long long)wc.total);
		if (wc.ident) atomicAdd(&counts[(size_t)(S + 1) << shift], (unsigned long long)wc.ident);
		if (wc.ambig) atomicAdd(&counts[(size_t)(S + 2) << shift], (unsigned long long)wc.ambig);
	}
}

// The counters of a demultiplex-alone launch.  Such a call can be a few tens of microseconds, and thousands of workgroups
// each adding S + 3 numbers to the SAME S + 3 addresses took longer than the lookups (16 single-index, 1 M reads: 34 us
// with, 8 us without; 10 M: 51 / 37 — about 10 ns per addition to an address that others add to).
// So a workgroup adds to one of kCountReplicas copies of the counters (different lines), and a one-workgroup kernel behind
// the launch (counts_fold_kernel) moves the copies into the counters and leaves them zero for the next launch.
// (Folding inside the kernel — the last workgroup found by tickets — was tried: a ticket is an atomic WITH return, and
// 2 048 of them cost 6 us at 10 M reads but 90 us at 100 M; an agent-scope release fence per workgroup cost 65 us.)
__device__ __forceinline__ void flush_counts_spread(const BarcodeDev &tb, unsigned long long *counts, const LdsPlan &lp, u32 *hist, int lane,
                                                    const WaveCounts &wc)
{
	const bool spread = tb.count_rep != nullptr && lp.use_lds_hist;
	flush_counts(tb.S, spread ? tb.count_rep + (size_t)(blockIdx.x & (kCountReplicas - 1)) * tb.count_rep_pitch : counts, lp, hist, lane, wc);
}
// the ctx's wide counters (one line each) into its u64[S+3], leaving them zero (sk_capi.hip folds before anything reads)
__global__ __launch_bounds__(256) void counts_fold_wide_kernel(unsigned long long *__restrict__ wide, int nc, unsigned long long *__restrict__ counts)
{
	__shared__ unsigned long long identified;           // only the lookup kernels add to these copies: see lut_identified_from_hist
	if (threadIdx.x == 0) identified = 0;
	__syncthreads();
	unsigned long long mine = 0;
	for (int i = threadIdx.x; i < nc; i += blockDim.x) {
		unsigned long long sum = 0;
#pragma unroll
		for (int r = 0; r < kCountReplicas; r++) {
			unsigned long long *p = wide + (((size_t)r * nc + i) << kCountWideShift);
			const unsigned long long v = *p;
			if (v) { sum += v; *p = 0; }
		}
		// `identified` IS the sum of the samples' bins (src/fasta_demultiplex.rs:177-178 add to both or to neither): it is taken
		// from them, and whatever a kernel may have added to the copies' own `identified` is dropped, not added on top
		if (sum && i != nc - 2) counts[i] += sum;
		if (i < nc - 3) mine += sum;
	}
	if (mine) atomicAdd(&identified, mine);
	__syncthreads();
	if (threadIdx.x == 0 && identified) counts[nc - 2] += identified;
}
// the dense rows (TileArgs::counts_wide_rows): block (x, y) sums counters 256 x ... of rows y, y + gridDim.y, ... — a thread's loads
// are independent, neighbouring threads read neighbouring counters
__global__ __launch_bounds__(256) void counts_fold_dense_kernel(unsigned long long *__restrict__ wide, int nc, int rows, unsigned long long *__restrict__ counts)
{
	__shared__ unsigned long long identified;
	if (threadIdx.x == 0) identified = 0;
	__syncthreads();
	const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	unsigned long long sum = 0;
	if (i < nc) {
		for (int r = (int)blockIdx.y; r < rows; r += (int)gridDim.y) {
			unsigned long long *p = wide + (size_t)r * nc + i;
			const unsigned long long v = *p;
			if (v) { sum += v; *p = 0; }
		}
		if (sum && i != nc - 2) atomicAdd(&counts[i], sum);          // (as above: `identified` comes from the samples' bins alone)
	}
	if (i < nc - 3 && sum) atomicAdd(&identified, sum);
	__syncthreads();
	if (threadIdx.x == 0 && identified) atomicAdd(&counts[nc - 2], identified);
}
hipError_t launch_counts_fold_wide(unsigned long long *wide, int nc, int dense_rows, unsigned long long *counts, hipStream_t st)
{
	if (dense_rows > 0) counts_fold_dense_kernel<<<dim3((unsigned)((nc + 255) / 256), 16), 256, 0, st>>>(wide, nc, dense_rows, counts);
	else counts_fold_wide_kernel<<<1, 256, 0, st>>>(wide, nc, counts);
	return hipGetLastError();
}
// derive: the copies were filled by a lookup kernel, which leaves `identified` to be summed from the samples' bins here
__global__ __launch_bounds__(256) void counts_fold_kernel(unsigned long long *__restrict__ rep, int pitch, int nc, unsigned long long *__restrict__ counts, int derive)
{
	__shared__ unsigned long long identified;
	if (threadIdx.x == 0) identified = 0;
	__syncthreads();
	unsigned long long mine = 0;
	for (int i = threadIdx.x; i < nc; i += blockDim.x) {
		unsigned long long sum = 0;
#pragma unroll
		for (int r = 0; r < kCountReplicas; r++) { sum += rep[(size_t)r * pitch + i]; rep[(size_t)r * pitch + i] = 0; }
		if (sum && !(derive && i == nc - 2)) atomicAdd(&counts[i], sum);      // (derive: `identified` comes from the samples' bins alone)
		if (i < nc - 3) mine += sum;
	}
	if (!derive) return;
	if (mine) atomicAdd(&identified, mine);
	__syncthreads();
	if (threadIdx.x == 0 && identified) atomicAdd(&counts[nc - 2], identified);
}
// what launch_tile_pass puts behind a demultiplex-alone kernel: the same condition as flush_counts_spread's
static hipError_t launch_counts_fold(const TileArgs &b, bool by_table, hipStream_t st)
{
	if (b.table.count_rep == nullptr || b.table.S + 3 > kMaxLdsHist) return hipSuccess;
	counts_fold_kernel<<<1, 256, 0, st>>>(b.table.count_rep, b.table.count_rep_pitch, b.table.S + 3, b.counts, by_table ? 1 : 0);
	return hipGetLastError();
}

// One (tile, mate) item is streamed as 1 KiB chunks.  The loads of two chunks are always in flight per wave
// (register slots R0/R1), INCLUDING across the scan and barcode phases: the last two issue slots of an item
// already fetch the first two chunks of the wave's next item, so HBM requests keep flowing while the wave
// does its LDS/VALU work.  Only loads run ahead; the LDS image is written when a chunk is consumed, so one
// LDS tile per wave is enough.
//
// Every global access in this kernel is an UNCONDITIONAL raw-buffer instruction: each (tile, array) gets its
// own buffer descriptor whose num_records is the tile's valid byte count, and the hardware range check clips
// what falls outside (per dword for 16-byte accesses, measured on gfx950: tools/micro/bufclip.hip).  Loads
// that would run past the tile return zeros, stores are dropped, absent optional arrays get a zero-record
// descriptor.  With no branch around any VMEM instruction the compiler's s_waitcnt vmcnt(N) counts are exact,
// which is what keeps the two register slots genuinely in flight.  Loads use records rounded UP to a dword
// (they may read up to 3 bytes past the end of a matrix), 16-byte stores use records rounded DOWN to a dword;
// the <= 3 bytes of a matrix whose size is not a multiple of 4 are finished by mask_tail_kernel.
//   MODE  : packed-compare mode of the quality threshold (QualConsts::mode)
//   DEMUX : the barcode phase is part of the pass, bit-sliced matcher with G <= 4 (S <= 128) and
//           64*bc_stride <= 2048; every other demultiplex shape runs as demux_tile_kernel beside this one
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#ifndef SK_SLOTS
#define SK_SLOTS 2
#endif
#ifndef SK_SLOTS1
#define SK_SLOTS1 4
#endif
constexpr int kSlots = SK_SLOTS;                      // 1 KiB chunk loads in flight per wave and stream
constexpr int kAuxStream = (SK_NT & 1) ? 2 : 0;      // nt on the read-once streams
#ifdef SK_ST_AUX
constexpr int kAuxStreamSt = SK_ST_AUX;              // tuning: cache-policy bits of the streaming stores (bit 0 sc0, bit 1 nt, bit 4 sc1)
#else
constexpr int kAuxStreamSt = (SK_NT & 2) ? 2 : 0;
#endif

// The kernel's first argument as it lies in the kernel-argument segment, behind an offset of zero that is produced by
// an asm statement at the point of use: a load through it is a scalar load that cannot be hoisted out of the loop it
// stands in.  For arguments that are needed once per tile and would otherwise occupy scalar registers all the time.
template <class T>
__device__ __forceinline__ const T __attribute__((address_space(4))) *kernel_args_now()
{
	u32 z;
	asm volatile("s_mov_b32 %0, 0" : "=s"(z));
	return reinterpret_cast<const T __attribute__((address_space(4))) *>(
		(const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + z);
}

__device__ __forceinline__ rsrc_t make_rsrc(const void *p, int64_t byte_off, int records)
{
	return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(static_cast<const uint8_t *>(p)) + (p ? byte_off : 0), 0, p ? records : 0, 0x00020000);
}

template <int MODE, bool DEMUX, int SLOTS>
__global__ __launch_bounds__(256, SLOTS > 4 ? 2 : 4) void tile_pass_kernel(const TileArgs a, const LdsPlan lp)
{
	// SLOTS == 10 is the trim-alone pass (launch_tile_pass): one input stream, no bases, nothing stored but lowest_k — the
	// seq registers, the mask arithmetic and the (dropped) stores are left out of it at compile time
	constexpr bool kTrimOnly = SLOTS == 10;
	const int lane = threadIdx.x & (kWave - 1);
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform by construction: make it an SGPR
	const int nwave = blockDim.x >> 6;
	uint8_t *tile = sk_smem + lp.tiles_off + wave * lp.tile_slot + kLdsPad;
	u32 *hist = reinterpret_cast<u32 *>(sk_smem + lp.hist_off);
	if (DEMUX) stage_tables(a.table, lp, hist);
	WaveCounts wc = {0u, 0u, 0u};

	const int ntiles = (int)((a.n + kTileRows - 1) / kTileRows);      // tile indices are 32-bit here (launch_tile_pass checks): five 64-bit counters were ten scalar registers
	const int stride = a.stride;
	const int m = a.qc.min_baseq;
	const u32 cl2 = a.qc.cl2;

	// Active mates (those with any output).  A mate's five pointers are fetched from the kernel arguments when its item
	// starts (a scalar load indexed by k): held in registers for the whole kernel they were twenty of the SGPRs this
	// kernel does not have.
	int nam = 0, mate0 = 0;
	{
		const bool act0 = a.n_mates > 0 && (a.mate[0].out_seq || a.mate[0].lowest_k);
		const bool act1 = a.n_mates > 1 && (a.mate[1].out_seq || a.mate[1].lowest_k);
		if (act0 && act1) nam = 2;
		else if (act0) nam = 1;
		else if (act1) { nam = 1; mate0 = 1; }
	}
	const int tstep = (int)gridDim.x * nwave;
	const int nchunkp = (((kTileRows * stride + 1023) >> 10) + SLOTS - 1) / SLOTS * SLOTS;   // chunks per full tile, rounded up to the slot count
	const int voff = lane * 16;

	// descriptors of the two input streams of item (t, k); a tile past the end gets zero records
	auto in_rsrc = [&](int t, int k, rsrc_t &rq, rsrc_t &rs) {
		const MateDev &mt = a.mate[mate0 + k];
		const bool ok = t < ntiles;
		const int64_t row0 = ok ? (int64_t)t * kTileRows : 0;
		const int rows = ok ? (int)((a.n - row0) < kTileRows ? (a.n - row0) : kTileRows) : 0;
		const int nb = (rows * stride + 3) & ~3;
		rq = make_rsrc(mt.qual, row0 * (int64_t)stride, nb);
		rs = make_rsrc(mt.out_seq ? mt.seq : nullptr, row0 * (int64_t)stride, nb);
	};

	int t = (int)blockIdx.x * nwave + wave;                 // tiles are dealt round-robin to the resident waves
	rsrc_t rq, rs;
	in_rsrc(t, 0, rq, rs);
	u32x4 qv[SLOTS], sv[kTrimOnly ? 1 : SLOTS];
#pragma unroll
	for (int i = 0; i < SLOTS; i++) {
		qv[i] = __builtin_amdgcn_raw_buffer_load_b128(rq, voff + i * 1024, 0, kAuxStream);
		if (!kTrimOnly) sv[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + i * 1024, 0, kAuxStream);
	}

	for (; t < ntiles; t += tstep) {
		const int64_t row0 = (int64_t)t * kTileRows;
		const int rows = (int)((a.n - row0) < kTileRows ? (a.n - row0) : kTileRows);
		const bool active = lane < rows;

		// the tile's observed barcodes are fetched now and only looked at in the barcode phase
		u32x4 bcv0 = {0u, 0u, 0u, 0u}, bcv1 = bcv0;
		if (DEMUX) {
			const TileArgs __attribute__((address_space(4))) *ka = kernel_args_now<TileArgs>();
			const int bstride = ka->bc_stride;
			const rsrc_t rb = make_rsrc(ka->bc, row0 * (int64_t)bstride, (rows * bstride + 3) & ~3);
			bcv0 = __builtin_amdgcn_raw_buffer_load_b128(rb, voff, 0, 0);
			bcv1 = __builtin_amdgcn_raw_buffer_load_b128(rb, voff + 1024, 0, 0);
		}

		for (int k = 0; k < nam; k++) {
			const MateDev &mt = a.mate[mate0 + k];
			const bool do_trim = mt.lowest_k != nullptr;
			const int nb = rows * stride;
			const rsrc_t ro = make_rsrc(mt.out_seq, row0 * (int64_t)stride, nb & ~3);
			const rsrc_t rl = make_rsrc(do_trim ? mt.len : nullptr, row0 * 2, rows * 2);
			// the item after this one: the other mate of this tile, or the first mate of the wave's next tile
			const int tn = (k + 1 < nam) ? t : t + tstep;
			const int kn = (k + 1 < nam) ? k + 1 : 0;
			rsrc_t nq, ns;
			in_rsrc(tn, kn, nq, ns);
			int len_ld = (int)__builtin_amdgcn_raw_buffer_load_b16(rl, lane * 2, 0, 0);
			// Materialise the length NOW: it is the youngest load at this point and the slots R0/R1 it queues
			// behind are needed immediately anyway.  Left to its first use after the chunk loop, the wait would be
			// an uncounted vmcnt(0) that drains the next item's prefetch right before the scan.
			asm volatile("" : "+v"(len_ld));

			// ---- stream phase: 16 B per lane, 1 KiB per wave instruction, two chunks in flight ------------
			if (kTrimOnly) {
				// Trim alone moves bytes from memory to the image and does nothing else with them, so the phase is written
				// without per-chunk decisions: only the tile's last group of chunks can reach past the image — there a lane
				// beyond it writes to the 16 bytes of pad behind the image instead of being masked off (one v_min against a
				// compare, an exec save, a branch and a restore) — and the group that fetches the NEXT item is its own copy
				// of the loop instead of five selects per chunk.  (Rows past the end of the last tile get the zeros the
				// descriptor returns; nothing reads them but lanes whose result is clipped.)
				const int image16 = (kTileRows * stride + 15) & ~15;
				int c = 0;
				for (; c + SLOTS < nchunkp; c += SLOTS) {
#pragma unroll
					for (int i = 0; i < SLOTS; i++) {
						const int off = (c + i) * 1024 + voff;
						*reinterpret_cast<u32x4 *>(tile + off) = qv[i];
						qv[i] = __builtin_amdgcn_raw_buffer_load_b128(rq, off + SLOTS * 1024, 0, kAuxStream);
					}
				}
#pragma unroll
				for (int i = 0; i < SLOTS; i++) {
					const int off = (c + i) * 1024 + voff;
					*reinterpret_cast<u32x4 *>(tile + min(off, image16)) = qv[i];
					qv[i] = __builtin_amdgcn_raw_buffer_load_b128(nq, voff + i * 1024, 0, kAuxStream);
				}
			} else
			for (int c = 0; c < nchunkp; c += SLOTS) {
				const bool last = c + SLOTS >= nchunkp;
#pragma unroll
				for (int i = 0; i < SLOTS; i++) {
					const int off = (c + i) * 1024 + voff;
					u32x4 o, vq;
					mask_dword4<MODE>(qv[i], sv[i], cl2, o, vq);
					__builtin_amdgcn_raw_buffer_store_b128(o, ro, off, 0, kAuxStreamSt);
					if (do_trim && off < nb) *reinterpret_cast<u32x4 *>(tile + off) = vq;
					qv[i] = __builtin_amdgcn_raw_buffer_load_b128(last ? nq : rq, last ? voff + i * 1024 : off + SLOTS * 1024, 0, kAuxStream);
					sv[i] = __builtin_amdgcn_raw_buffer_load_b128(last ? ns : rs, last ? voff + i * 1024 : off + SLOTS * 1024, 0, kAuxStream);
				}
			}
			rq = nq; rs = ns;

			// ---- scan phase: lane r walks row r of the LDS image from its 3' end --------------------------
			if (do_trim) {
				wave_lds_fence();
				const int len = (mt.len != nullptr) ? len_ld : stride;
				int kk;
				if (stride >= (1 << kKeyBits)) kk = trim_scan_wide<kTrimOnly>(tile, lane * stride, len, stride, m, active);
				else if (mt.len != nullptr) kk = trim_scan_packed<false, kTrimOnly>(tile, lane * stride, len, stride, m, active);
				else kk = trim_scan_packed<true, kTrimOnly>(tile, lane * stride, len, stride, m, active);
				// (the descriptor of lowest_k is made here, from the pointer as it lies in the kernel arguments: four scalar
				// registers less across the chunk loop and the scan)
				__builtin_amdgcn_raw_buffer_store_b16((unsigned short)kk, make_rsrc(kernel_args_now<TileArgs>()->mate[mate0 + k].lowest_k, row0 * 2, rows * 2), lane * 2, 0, 0);
				wave_lds_fence();
			}
		}

		// ---- barcode phase (bit-sliced matcher) -----------------------------------------------------------
		if (DEMUX) {
			// (what this phase needs of the kernel arguments — the matcher's shape, the four column pointers — is read from
			// them here, once per tile, through an offset the compiler cannot see through: as loop invariants they would
			// sit in scalar registers for the whole kernel)
			const TileArgs __attribute__((address_space(4))) *ka = kernel_args_now<TileArgs>();
			const int bstride = ka->bc_stride;
			if (voff < rows * bstride) *reinterpret_cast<u32x4 *>(tile + voff) = bcv0;
			if (1024 + voff < rows * bstride) *reinterpret_cast<u32x4 *>(tile + 1024 + voff) = bcv1;
			wave_lds_fence();
			int best = 0x7fffffff, first = 0, last = 0;
			const uint8_t *row = tile + lane * bstride;
			const int mm_off = ka->table.bs_mm_off, tL = ka->table.L;
			switch (ka->table.G) {   // wave-uniform
			case 1: demux_row_bitsliced<1>(row, sk_smem, mm_off, 1, 0, tL, best, first, last); break;
			case 2: demux_row_bitsliced<2>(row, sk_smem, mm_off, 2, 0, tL, best, first, last); break;
			case 3: demux_row_bitsliced<3>(row, sk_smem, mm_off, 3, 0, tL, best, first, last); break;
			default: demux_row_bitsliced<4>(row, sk_smem, mm_off, 4, 0, tL, best, first, last); break;
			}
			// D3 (src/fasta_demultiplex.rs:168-194) with descriptor-clipped stores
			int code = kAssignNone;
			if (best <= ka->table.max_diff) code = (first == last) ? first : kAssignAmbiguous;
			__builtin_amdgcn_raw_buffer_store_b32((u32)code, make_rsrc(ka->assign, row0 * 4, rows * 4), lane * 4, 0, 0);
			__builtin_amdgcn_raw_buffer_store_b8((uint8_t)(best > 255 ? 255 : best), make_rsrc(ka->lowest_diff, row0, rows), lane, 0, 0);
			__builtin_amdgcn_raw_buffer_store_b16((unsigned short)first, make_rsrc(ka->first_idx, row0 * 2, rows * 2), lane * 2, 0, 0);
			__builtin_amdgcn_raw_buffer_store_b16((unsigned short)last, make_rsrc(ka->last_idx, row0 * 2, rows * 2), lane * 2, 0, 0);
			if (active && code >= 0) atomicAdd(&hist[code], 1u);
			wc.total += (u32)__popcll(__ballot(active));
			wc.ident += (u32)__popcll(__ballot(active && code >= 0));
			wc.ambig += (u32)__popcll(__ballot(active && code == kAssignAmbiguous));
			wave_lds_fence();
		}
	}
	if (DEMUX) {
		const TileArgs __attribute__((address_space(4))) *ka = kernel_args_now<TileArgs>();
		flush_counts(ka->table.S, ka->counts, lp, hist, lane, wc);
	}
}

// ---------------------------------------------------------------------------------------------------
// tile_blocked_kernel — the same pass over the TILE-BLOCKED layout (BlockedArgs): everything a tile reads is one
// contiguous range of `in` (qual / seq of each mate, the observed barcodes, optional lengths) and everything it
// writes one contiguous range of `out`.  A wave therefore streams ONE advancing read range and ONE write range
// instead of five and four far-apart ones, the host moves a batch with one copy per direction, and the kernel needs
// two buffer descriptors per tile (block base + scalar segment offsets) instead of one per array.  Blocks are
// whole also for the last tile, so nothing is clipped: rows past n are computed and land in the block's padding;
// only the counters look at `active`.  The LDS image takes whole chunks (its slot is nchunkp KiB).
// ---------------------------------------------------------------------------------------------------
// Every global access is an unconditional raw-buffer instruction whose VGPR offset carries the segment offset, so
// the hardware range check (VGPR offset + immediate against num_records = the block size) covers it; absent
// segments get an offset past every block (kNoSeg): their loads return zeros without memory traffic, their stores
// are dropped, and no branch sits around a VMEM instruction (exact s_waitcnt counts, as in tile_pass_kernel).
constexpr int kNoSeg = 0x40000000;

template <int MODE, bool DEMUX, int SLOTS>
__global__ __launch_bounds__(256, 4) void tile_blocked_kernel(const BlockedArgs a, const LdsPlan lp)
{
	const int lane = threadIdx.x & (kWave - 1);
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int nwave = blockDim.x >> 6;
	uint8_t *tile = sk_smem + lp.tiles_off + wave * lp.tile_slot + kLdsPad;
	u32 *hist = reinterpret_cast<u32 *>(sk_smem + lp.hist_off);
	if (DEMUX) stage_tables(a.table, lp, hist);
	WaveCounts wc = {0u, 0u, 0u};

	const int ntiles = (int)((a.n + kTileRows - 1) / kTileRows);      // 32-bit tile indices (launch_tile_blocked checks)
	const int stride = a.stride;
	const u32 cl2 = a.qc.cl2;
	const bool do_trim = a.out_lowest_k[0] >= 0, ragged = a.in_len[0] >= 0;
	const int seg = kTileRows * stride;
	const int nchunkp = (((seg + 1023) >> 10) + SLOTS - 1) / SLOTS * SLOTS;
	const int voff = lane * 16;
	const int lim = seg - voff;                  // chunk offsets below this are inside the lane's part of a segment
	const int tstep = (int)gridDim.x * nwave;
	auto seg_off = [](int o) { return o >= 0 ? o : kNoSeg; };
	// A segment's last chunk runs past the segment (64*stride is rarely a multiple of 1 KiB): those lanes must not
	// touch memory — inside the block they would fetch the NEXT segment's bytes a second time (+7 % read traffic at
	// 150 bp) — so their offset is sent out of range, where the hardware answers with zeros and no request.
	auto at = [&](int base, int coff) { return coff < lim ? base + coff : kNoSeg; };

	int t = (int)blockIdx.x * nwave + wave;                     // tiles are dealt round-robin to the resident waves
	// descriptor of a tile's input block; past the last tile: zero records (the prefetch of the item after the last)
	auto in_block = [&](int tt) {
		const BlockedArgs __attribute__((address_space(4))) *ka = kernel_args_now<BlockedArgs>();
		const int bytes = ka->in_block;
		return make_rsrc(tt < ntiles ? ka->in : nullptr, tt * (int64_t)bytes, bytes);
	};
	rsrc_t rin = in_block(t);
	int vq = voff + a.in_qual[0], vs = voff + seg_off(a.in_seq[0]);      // lane offsets of the item's two streams inside the block
	u32x4 qv[SLOTS], sv[SLOTS];
#pragma unroll
	for (int i = 0; i < SLOTS; i++) {
		qv[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, at(vq, i * 1024), 0, kAuxStream);
		sv[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, at(vs, i * 1024), 0, kAuxStream);
	}
	const int bc_lim = kTileRows * a.bc_stride - voff;

	for (; t < ntiles; t += tstep) {
		u32x4 bcv0 = {0u, 0u, 0u, 0u}, bcv1 = bcv0;
		if (DEMUX) {
			const int in_bc = kernel_args_now<BlockedArgs>()->in_bc;      // once per tile: read where it is used (as tile_pass_kernel does)
			bcv0 = __builtin_amdgcn_raw_buffer_load_b128(rin, 0 < bc_lim ? voff + in_bc : kNoSeg, 0, 0);
			bcv1 = __builtin_amdgcn_raw_buffer_load_b128(rin, 1024 < bc_lim ? voff + in_bc + 1024 : kNoSeg, 0, 0);
		}
		const BlockedArgs __attribute__((address_space(4))) *kt = kernel_args_now<BlockedArgs>();
		uint8_t *const out_base = kt->out;
		const int out_block = kt->out_block;
		const rsrc_t rout = make_rsrc(out_base, t * (int64_t)out_block, out_block);

		for (int k = 0; k < a.n_mates; k++) {
			// masked bases go through their own descriptor: it clips the item's last chunk at the segment's end
			const int so = a.out_seq[k];
			const rsrc_t ro = make_rsrc(so >= 0 ? out_base : nullptr, t * (int64_t)out_block + so, seg);
			// rin is still this tile's block here (it moves on in the last chunks of the tile's last mate)
			int len_ld = (int)__builtin_amdgcn_raw_buffer_load_b16(rin, lane * 2 + seg_off(a.in_len[k]), 0, 0);
			asm volatile("" : "+v"(len_ld));                      // materialise now (see tile_pass_kernel)

			// ---- stream phase: SLOTS chunks of each stream in flight ------------------------------------------------
			int c = 0;
			for (; c < nchunkp - SLOTS; c += SLOTS) {
#pragma unroll
				for (int i = 0; i < SLOTS; i++) {
					const int off = (c + i) * 1024;
					u32x4 o, w;
					mask_dword4<MODE>(qv[i], sv[i], cl2, o, w);
					__builtin_amdgcn_raw_buffer_store_b128(o, ro, voff + off, 0, kAuxStreamSt);
					*reinterpret_cast<u32x4 *>(tile + voff + off) = w;
					qv[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, at(vq, off + SLOTS * 1024), 0, kAuxStream);
					sv[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, at(vs, off + SLOTS * 1024), 0, kAuxStream);
				}
			}
			// the item's last chunks; their slots already fetch the NEXT item — the other mate of this tile, or the
			// first mate of the wave's next tile — so loads stay in flight across the scan and barcode phases
			{
				const bool last_mate = k + 1 >= a.n_mates;
				if (last_mate) rin = in_block(t + tstep);
				vq = voff + (a.in_qual[last_mate ? 0 : 1]);
				vs = voff + seg_off(a.in_seq[last_mate ? 0 : 1]);
#pragma unroll
				for (int i = 0; i < SLOTS; i++) {
					const int off = (c + i) * 1024;
					u32x4 o, w;
					mask_dword4<MODE>(qv[i], sv[i], cl2, o, w);
					__builtin_amdgcn_raw_buffer_store_b128(o, ro, voff + off, 0, kAuxStreamSt);
					*reinterpret_cast<u32x4 *>(tile + voff + off) = w;
					qv[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, at(vq, i * 1024), 0, kAuxStream);
					sv[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, at(vs, i * 1024), 0, kAuxStream);
				}
			}

			// ---- scan phase --------------------------------------------------------------------------------------
			if (do_trim) {
				wave_lds_fence();
				const bool active = lane < a.n - t * kTileRows;
				const int len = ragged ? min(len_ld, stride) : stride;      // rows past n hold padding: keep the scan inside the image
				int kk;
				if (ragged) kk = trim_scan_packed<false>(tile, lane * stride, len, stride, a.qc.min_baseq, active);
				else kk = trim_scan_packed<true>(tile, lane * stride, len, stride, a.qc.min_baseq, active);
				__builtin_amdgcn_raw_buffer_store_b16((unsigned short)kk, rout, lane * 2 + (a.out_lowest_k[k]), 0, 0);
				wave_lds_fence();
			}
		}

		// ---- barcode phase (bit-sliced matcher) --------------------------------------------------------------------
		if (DEMUX) {
			const int bstride = a.bc_stride;
			*reinterpret_cast<u32x4 *>(tile + voff) = bcv0;
			*reinterpret_cast<u32x4 *>(tile + 1024 + voff) = bcv1;
			wave_lds_fence();
			int best = 0x7fffffff, first = 0, last = 0;
			const uint8_t *row = tile + lane * bstride;
			switch (a.table.G) {   // wave-uniform
			case 1: demux_row_bitsliced<1>(row, sk_smem, a.table.bs_mm_off, 1, 0, a.table.L, best, first, last); break;
			case 2: demux_row_bitsliced<2>(row, sk_smem, a.table.bs_mm_off, 2, 0, a.table.L, best, first, last); break;
			case 3: demux_row_bitsliced<3>(row, sk_smem, a.table.bs_mm_off, 3, 0, a.table.L, best, first, last); break;
			default: demux_row_bitsliced<4>(row, sk_smem, a.table.bs_mm_off, 4, 0, a.table.L, best, first, last); break;
			}
			int code = kAssignNone;                                  // D3: src/fasta_demultiplex.rs:168-194
			if (best <= a.table.max_diff) code = (first == last) ? first : kAssignAmbiguous;
			const BlockedArgs __attribute__((address_space(4))) *ka = kernel_args_now<BlockedArgs>();
			__builtin_amdgcn_raw_buffer_store_b32((u32)code, rout, lane * 4 + ka->out_assign, 0, 0);
			__builtin_amdgcn_raw_buffer_store_b8((uint8_t)(best > 255 ? 255 : best), rout, lane + seg_off(ka->out_lowest_diff), 0, 0);
			__builtin_amdgcn_raw_buffer_store_b16((unsigned short)first, rout, lane * 2 + seg_off(ka->out_first_idx), 0, 0);
			__builtin_amdgcn_raw_buffer_store_b16((unsigned short)last, rout, lane * 2 + seg_off(ka->out_last_idx), 0, 0);
			const bool active = lane < a.n - t * kTileRows;
			if (active && code >= 0) atomicAdd(&hist[code], 1u);
			wc.total += (u32)__popcll(__ballot(active));
			wc.ident += (u32)__popcll(__ballot(active && code >= 0));
			wc.ambig += (u32)__popcll(__ballot(active && code == kAssignAmbiguous));
			wave_lds_fence();
		}
	}
	if (DEMUX) flush_counts(a.table.S, a.counts, lp, hist, lane, wc);
}

// the last (n*stride mod 4) bytes of a matrix, which the dword-clipped 16-byte stores of the tile pass leave out
__global__ void mask_tail_kernel(const uint8_t *seq, const uint8_t *qual, uint8_t *out, int64_t total, int m)
{
	const int64_t i = (total & ~(int64_t)3) + threadIdx.x;
	if (i < total) out[i] = ((uint8_t)(qual[i] - 33) < m) ? (uint8_t)'N' : seq[i];
}

// Standalone demultiplex pass: every matcher (bit-sliced with any group count, one-hot popcount, byte
// compare), any bc_stride the LDS tile can hold.  One wave per 64-row tile as above.
__global__ __launch_bounds__(256) void demux_tile_kernel(const TileArgs a, const LdsPlan lp)
{
	const int lane = threadIdx.x & (kWave - 1);
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform by construction: make it an SGPR
	const int nwave = blockDim.x >> 6;
	uint8_t *tile = sk_smem + lp.tiles_off + wave * lp.tile_slot + kLdsPad;
	u32 *hist = reinterpret_cast<u32 *>(sk_smem + lp.hist_off);
	const u32 *cand_lds = reinterpret_cast<const u32 *>(sk_smem);
	stage_tables(a.table, lp, hist);
	WaveCounts wc = {0u, 0u, 0u};
	const int S = a.table.S;
	const int64_t ntiles = (a.n + kTileRows - 1) / kTileRows;
	const int bstride = a.bc_stride;
	const int64_t bc_total = a.n * (int64_t)bstride;
	for (int64_t t = (int64_t)blockIdx.x * nwave + wave; t < ntiles; t += (int64_t)gridDim.x * nwave) {
		const int64_t row0 = t * kTileRows;
		const int rows = (int)((a.n - row0) < kTileRows ? (a.n - row0) : kTileRows);
		const bool active = lane < rows;
		const int64_t base = row0 * (int64_t)bstride;
		const int tile_bytes = rows * bstride;
		for (int off = lane * 16; off < tile_bytes; off += kWave * 16) {
			const int64_t g = base + off;
			u32x4 v;
			if (g + 16 <= bc_total) v = *reinterpret_cast<const u32x4 *>(a.bc + g);
			else v = load_tail(a.bc + g, (int)(bc_total - g));
			*reinterpret_cast<u32x4 *>(tile + off) = v;
		}
		wave_lds_fence();
		int diff = 255, first = 0, last = 0;
		if (S > 0) {
			const int rs = lane * bstride;
			if (a.table.bs != nullptr) {
				int best = 0x7fffffff;
				const uint8_t *row = tile + rs;
				switch (a.table.G) {   // wave-uniform
				case 1: demux_row_bitsliced<1>(row, sk_smem, a.table.bs_mm_off, 1, 0, a.table.L, best, first, last); break;
				case 2: demux_row_bitsliced<2>(row, sk_smem, a.table.bs_mm_off, 2, 0, a.table.L, best, first, last); break;
				case 3: demux_row_bitsliced<3>(row, sk_smem, a.table.bs_mm_off, 3, 0, a.table.L, best, first, last); break;
				case 4: demux_row_bitsliced<4>(row, sk_smem, a.table.bs_mm_off, 4, 0, a.table.L, best, first, last); break;
				default:
					for (int g0 = 0; g0 < a.table.G; g0++)
						demux_row_bitsliced<1>(row, sk_smem, a.table.bs_mm_off, a.table.G, g0, a.table.L, best, first, last);
					break;
				}
				diff = best;
			} else if (a.table.onehot != nullptr) {
				switch (a.table.W) {   // wave-uniform
				case 1: demux_row_onehot<1>(tile, rs, a.table, cand_lds, diff, first, last); break;
				case 2: demux_row_onehot<2>(tile, rs, a.table, cand_lds, diff, first, last); break;
				case 3: demux_row_onehot<3>(tile, rs, a.table, cand_lds, diff, first, last); break;
				case 4: demux_row_onehot<4>(tile, rs, a.table, cand_lds, diff, first, last); break;
				case 5: demux_row_onehot<5>(tile, rs, a.table, cand_lds, diff, first, last); break;
				case 6: demux_row_onehot<6>(tile, rs, a.table, cand_lds, diff, first, last); break;
				case 7: demux_row_onehot<7>(tile, rs, a.table, cand_lds, diff, first, last); break;
				default: demux_row_onehot<8>(tile, rs, a.table, cand_lds, diff, first, last); break;
				}
			} else {
				demux_row_bytes(tile, rs, a.table, sk_smem, diff, first, last);
			}
		}
		demux_commit(a, lp, hist, row0 + lane, active, diff, first, last, wc);
		wave_lds_fence();
	}
	flush_counts_spread(a.table, a.counts, lp, hist, lane, wc);
}

// Tuning knobs of the lookup kernels (tools/lut_cold_ab.py builds variants and times them with their rows coming from HBM).
// Round 5's cold measurements, 100 M rows in one call: tiles of a wave on their way 1 / 2 / 4 — the same time (the launch waits
// for the memory side, not for latency); nt on the result stores -2.5 ... -3 % for every shape; nt on the row loads -5 % for the
// coalesced 16-byte loads of demux_lut8x2_kernel and +3 ... +5 % for gathered rows (a line serves two or three wave instructions
// there; nt drops it after the first).
#ifndef SK_LUT_DEPTH
#define SK_LUT_DEPTH 2
#endif
#ifndef SK_LUT_PAIR_DEPTH
#define SK_LUT_PAIR_DEPTH 4
#endif
#ifndef SK_LUT_AUX
#define SK_LUT_AUX 0                     // cache policy of gathered and one-row-per-lane loads (2 = nt)
#endif
#ifndef SK_LUT8_AUX
#define SK_LUT8_AUX 2                    // of demux_lut8x2_kernel's coalesced 16-byte loads
#endif
#ifndef SK_LUT_ST_AUX
#define SK_LUT_ST_AUX 2                  // of the result stores
#endif
constexpr int kLutAux = SK_LUT_AUX, kLut8Aux = SK_LUT8_AUX, kLutStAux = SK_LUT_ST_AUX;

// N consecutive dwords at a dword-aligned byte offset of a raw buffer (clipped per dword): the widest loads that cover them
template <int N> __device__ __forceinline__ void gather_dwords(rsrc_t rb, int off, u32 (&d)[N])
{
	typedef u32 u32x2_t __attribute__((ext_vector_type(2)));
	typedef u32 u32x3_t __attribute__((ext_vector_type(3)));
	static_assert(N >= 2 && N <= 6, "a key segment of one to five dwords and the dword behind it");
	if constexpr (N == 2) { const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rb, off, 0, kLutAux); d[0] = v[0]; d[1] = v[1]; }
	else if constexpr (N == 3) { const u32x3_t v = __builtin_amdgcn_raw_buffer_load_b96(rb, off, 0, kLutAux); d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; }
	else {
		const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rb, off, 0, kLutAux);
		d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
		if constexpr (N == 5) d[4] = __builtin_amdgcn_raw_buffer_load_b32(rb, off + 16, 0, kLutAux);
		if constexpr (N == 6) { const u32x2_t w = __builtin_amdgcn_raw_buffer_load_b64(rb, off + 16, 0, kLutAux); d[4] = w[0]; d[5] = w[1]; }
	}
}

// (the class of each of a dword's four bytes — the index of the sheet letter the byte equals, `other` for a byte the sheet never
// uses — is sk_lut.h's lut_classes: the CPU model of the table shares it)

// ---------------------------------------------------------------------------------------------------
// D1+D2+D3 by table lookup (sk_lut.h): demultiplex alone.  With max_diff <= 1 the barcodes that get a sample or the
// ambiguity verdict are the sheet rows and their one-substitution neighbours — 96 x (16 x 4 + 1) = 6 240 keys for the 96
// dual-index sheet.  The host decides each with the reference's loop (src/fasta_demultiplex.rs:154-194) and stores
// lowest_diff / first / last with it; a read is: its bytes classified (v_perm letter table, packed-byte compare),
// packed to two words, mixed, and looked up in two slots.  An 8-byte entry holds the quotient of the key instead of the
// key, which is what makes the dual-index table 128 KiB — it lives in the workgroup's LDS (sixteen waves share one copy),
// so the two probes of a read are two ds_read_b64 instead of two trips to L2.  Not found = nothing within max_diff =
// SK_ASSIGN_NONE (its detail columns then say 255 / -1 / -1: SK_DETAIL_MATCHED of include/seqkit_hip.h).
//   W1, W2 : key dwords — the row's first W1 dwords and, with a separator, W2 == W1 dwords from the byte after it
//   DIRECT : rows start on dword boundaries, W1 <= 2, no separator: lane r loads row r as it lies (8 B/lane, coalesced, for
//            8 bp); otherwise lane r gathers the dwords its row's segments lie in, at any alignment (see GATHER below) —
//            either way a row goes from memory to its lane's registers, nothing passes through LDS but the table
//   LDSTAB : the table fits the workgroup's LDS (<= 128 KiB); else it is read through the vector cache
//   DETAIL : lowest_diff / first_idx / last_idx are written too
// Two tiles per wave are always in flight (register slots), so that a CU of sixteen waves has ~34 KiB on the way.  There
// is no branch around a VMEM instruction in the loop but the narrow columns of a call's last, partial quad: the waits for the
// register slots are counted, not drained.
// ---------------------------------------------------------------------------------------------------
//   PAIR   : the factored form of sk_lut.h (a sheet `i7+i5` whose full-key table would not fit the LDS): one lookup per half
//            (-> half id, distance) and one of the pair of ids (-> first / last sample); three small tables in one LDS blob
template <int W1, int W2, bool DIRECT, bool LDSTAB, bool DETAIL, bool PAIR = false, bool MANY = false>
__global__ __launch_bounds__(LDSTAB ? 1024 : 256) void demux_lut_kernel(const TileArgs a, const LdsPlan lp)
{
	typedef u32 u32x2_t __attribute__((ext_vector_type(2)));
	constexpr int W = W1 + W2;
	static_assert(W >= 1 && W <= 5 && (W2 == 0 || W2 == W1) && !(DIRECT && (W2 > 0 || W1 > 2)), "shape");
	static_assert(!PAIR || (W2 == W1 && W1 <= 2 && LDSTAB && !DIRECT), "the factored form: two halves of one or two dwords, tables in LDS");
	const int lane = threadIdx.x & (kWave - 1);
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int nwave = blockDim.x >> 6;
	u32 *hist = reinterpret_cast<u32 *>(sk_smem + lp.hist_off);
	const LutDev &t = a.table.nbr;
	const int S = a.table.S;
	const int64_t ntiles = MANY ? 4 * (int64_t)a.many_quads : (a.n + kTileRows - 1) / kTileRows;      // (MANY: demux_lut8x2_kernel's note; a quad here is four tiles)
	const int bstride = a.bc_stride;
	// tiles of a wave on their way: two (a quad's four tiles take the register slots in turn: 1, 2 or 4; replayed on on-die rows
	// at 10 M pairs more were slower in round 4, from HBM 1 / 2 / 4 take the same time); four for the factored form, whose three
	// lookups per row leave a tile's loads more time to hide in
	constexpr int kDepth = PAIR ? SK_LUT_PAIR_DEPTH : SK_LUT_DEPTH;
	static_assert(kDepth == 1 || kDepth == 2 || kDepth == 4, "a quad's four tiles take the register slots in turn");
	u32 raw[kDepth][W];
	// GATHER (every shape but DIRECT): a lane fetches the dwords its row's key lies in straight from memory, at any alignment —
	// W1 + 1 dwords from the one that holds the row's first byte and, with a separator, W2 + 1 from the one that holds the byte
	// after it.  (Through round 4's first sessions such tiles went through an image in LDS: 2 writes and 5 reads at a 17-byte
	// pitch per dual-index tile kept the LDS pipe busier than HBM — 42.3 us against 35.8 at 10 M pairs.)  The separator itself is
	// one byte of the first window's last two dwords (v_perm_b32).
	constexpr int G2 = W2 > 0 ? W2 + 1 : 2;
	u32 g1[kDepth][W1 + 1], g2[kDepth][G2];
	const int rs = lane * bstride, rs2 = rs + t.sep_off + 1;            // where the lane's row lies in a tile does not depend on the tile
	const u32 sh1 = (u32)rs & 3u, sh2 = (u32)rs2 & 3u;
	const u32 sep_sel = 0x0c0c0c00u | (sh1 + (u32)(t.sep_off - 4 * (W1 - 1)));      // the separator among the 8 bytes of dwords W1 - 1 and W1: 1 ... 7
	// tiles are counted in 32 bits (launch_tile_pass checks); a tile past the last clips to nothing (zero-record descriptors)
	const int nt32 = (int)ntiles, last_rows = (int)(a.n - (ntiles - 1) * kTileRows);
	auto rows_of = [&](int ti) { return ti < nt32 - 1 ? kTileRows : (ti == nt32 - 1 ? last_rows : 0); };
	auto rows_many = [&](int ti, u32 row_end) {
		const int left = (int)(row_end - (u32)ti * (u32)kTileRows);
		return left <= 0 ? 0 : (left < kTileRows ? left : kTileRows);
	};
	struct BatchView { const uint8_t *bc; u32 row_end; int q_end; };      // (read when a cursor moves, not per tile: demux_lut8x2_kernel's note)
	int cb = 0, fb = 0;                                                  // MANY: the batch of the quad at hand / of the quad behind it in the wave's sequence
	BatchView cv = {nullptr, 0u, 0}, fv = {nullptr, 0u, 0};
	int32_t *c_assign = nullptr;
	uint8_t *c_low = nullptr;
	int16_t *c_first = nullptr, *c_last = nullptr;
	auto view_of = [&](int i) { const ManyBatch &m = a.many[i]; return BatchView{m.bc, m.row_end, m.q_end}; };
	auto seek = [&](int gq, int &cur, BatchView &v) {
		bool moved = false;
		while (cur + 1 < a.n_many && gq >= v.q_end) { cur++; v = view_of(cur); moved = true; }
		return moved;
	};
	if (MANY) {
		cv = fv = view_of(0);
		c_assign = a.many[0].assign; c_low = a.many[0].lowest_diff; c_first = a.many[0].first_idx; c_last = a.many[0].last_idx;
	}
	int cur_quad = 0;                                                    // MANY: fetch() takes a tile of this quad from batch cb, any other from fb
	auto fetch = [&](int ti, int s) {
		const bool mine = (ti >> 2) == cur_quad;
		const u32 m_end = mine ? cv.row_end : fv.row_end;
		const uint8_t *m_bc = mine ? cv.bc : fv.bc;
		const int rows = MANY ? rows_many(ti, m_end) : rows_of(ti);
		const rsrc_t rb = make_rsrc(MANY ? m_bc : a.bc, (int64_t)(rows ? ti : 0) * (kTileRows * bstride), (rows * bstride + 3) & ~3);
		if (DIRECT) {
			if (W == 2) {
				const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rb, lane * bstride, 0, kLutAux);
				raw[s][0] = v[0]; raw[s][W - 1] = v[1];
			} else {
				raw[s][0] = __builtin_amdgcn_raw_buffer_load_b32(rb, lane * bstride, 0, kLutAux);
			}
		} else {
			gather_dwords<W1 + 1>(rb, rs & ~3, g1[s]);
			if constexpr (W2 > 0) gather_dwords<W2 + 1>(rb, rs2 & ~3, g2[s]);
		}
	};
	// A wave's unit of work is a QUAD: four consecutive tiles = 256 consecutive rows, whose codes leave as ONE 16-byte store per
	// lane.  (Through round 5's first session a tile's codes left as a 4-byte store per lane: at 100 M pairs the kernel took
	// 410 us with those stores and 293 without them, and the same 410 with the lookups taken out — the narrow stores were what
	// the launch waited for, tools/lut_cold_ab.py.)  The four codes of a lane belong to rows 64 j + lane; they change places
	// through 1 KiB of LDS per wave (four ds_write_b32, one ds_read_b128, all conflict-free; a wave's LDS operations execute in
	// order, so nothing waits but the read's consumer).
	const int qstep = (int)gridDim.x * nwave;
	int qb = (int)blockIdx.x * nwave + wave;
	if (MANY) { seek(qb, cb, cv); fb = cb; fv = cv; seek(qb + qstep, fb, fv); cur_quad = qb; if (cb != 0) { const ManyBatch &mb = a.many[cb]; c_assign = mb.assign; c_low = mb.lowest_diff; c_first = mb.first_idx; c_last = mb.last_idx; } }
	for (int s = 0; s < kDepth; s++) fetch(4 * qb + s, s);
	
	// the table and the histogram while the first tiles are on their way
	if (lp.use_lds_hist) for (int i = threadIdx.x; i < S + 3; i += blockDim.x) hist[i] = 0u;
	const int ltab_off = lp.tiles_off - lp.table_bytes;                  // LDSTAB: the table sits behind the histogram
	if (LDSTAB) {
		// eight loads per thread on their way before the first is written: the 128 KiB of a dual-index table are one round trip
		// to L2 per workgroup, not eight
		const int n16 = lp.table_bytes >> 4;
		for (int i0 = threadIdx.x; i0 < n16; i0 += 8 * (int)blockDim.x) {
			u32x4 tv[8];
#pragma unroll
			for (int j = 0; j < 8; j++) {
				const int i = i0 + j * (int)blockDim.x;
				if (i < n16) tv[j] = reinterpret_cast<const u32x4 *>(PAIR ? t.pair.tab : t.tab)[i];
			}
#pragma unroll
			for (int j = 0; j < 8; j++) {
				const int i = i0 + j * (int)blockDim.x;
				if (i < n16) *reinterpret_cast<u32x4 *>(sk_smem + ltab_off + i * 16) = tv[j];
			}
		}
	}
	__syncthreads();
	auto entry = [&](u32 slot) {
		if (LDSTAB) return *reinterpret_cast<const u32x2_t *>(sk_smem + ltab_off + (int)slot * 8);
		return *reinterpret_cast<const u32x2_t *>(t.tab + (size_t)slot * 2);
	};
	const int ltab_off2 = ltab_off + (t.mask + 1) * 8;           // table 2 behind table 1: its base is a constant, not an add per row
	auto entry2 = [&](u32 slot) {
		if (LDSTAB) return *reinterpret_cast<const u32x2_t *>(sk_smem + ltab_off2 + (int)slot * 8);
		return *reinterpret_cast<const u32x2_t *>(t.tab + (size_t)(t.mask + 1) * 2 + (size_t)slot * 2);
	};
	auto entry_w1 = [&](u32 slot) {                              // the second word alone: all a key without B needs (sk_lut.h)
		if (LDSTAB) return *reinterpret_cast<const u32 *>(sk_smem + ltab_off + 4 + (int)slot * 8);
		return t.tab[(size_t)slot * 2 + 1];
	};
	auto entry2_w1 = [&](u32 slot) {
		if (LDSTAB) return *reinterpret_cast<const u32 *>(sk_smem + ltab_off2 + 4 + (int)slot * 8);
		return t.tab[(size_t)(t.mask + 1) * 2 + (size_t)slot * 2 + 1];
	};
	const u32 idx_bits = (u32)__builtin_popcount(t.idx_mask);
	u32 n_total = 0;
	u32 *tp = reinterpret_cast<u32 *>(sk_smem + lp.tiles_off + wave * lp.tile_slot);
	const int nq32 = (nt32 + 3) >> 2;
	for (; qb < nq32; qb += qstep) {
		int code[4], dtot[4], dfirst[4], dlast[4];
		u32 q_row_end = (u32)a.n;
		int32_t *q_assign = a.assign;
		uint8_t *q_low = a.lowest_diff;
		int16_t *q_first = a.first_idx, *q_last = a.last_idx;
		if (MANY) {
			if (seek(qb, cb, cv)) { const ManyBatch &mb = a.many[cb]; c_assign = mb.assign; c_low = mb.lowest_diff; c_first = mb.first_idx; c_last = mb.last_idx; }
			cur_quad = qb;
			q_row_end = cv.row_end; q_assign = c_assign; q_low = c_low; q_first = c_first; q_last = c_last;
			seek(qb + qstep, fb, fv);
		}
#pragma unroll
		for (int j = 0; j < 4; j++) {
			constexpr int kD = kDepth;
			const int s = j % kD;
			const int ti = 4 * qb + j;
			const int rows = MANY ? rows_many(ti, q_row_end) : rows_of(ti);
			const bool active = lane < rows;
			const int nj = j + kD;                                       // the tile that takes this one's register slot
			const int tnext = nj < 4 ? 4 * qb + nj : 4 * (qb + qstep) + (nj - 4);
			u32 d[W];
			u32 sepbad = 0u;
			if (DIRECT) {
#pragma unroll
				for (int w = 0; w < W; w++) d[w] = raw[s][w];
				fetch(tnext, s);
			} else {
#pragma unroll
				for (int w = 0; w < W1; w++) d[w] = __builtin_amdgcn_alignbyte(g1[s][w + 1], g1[s][w], sh1);
				if constexpr (W2 > 0) {
#pragma unroll
					for (int w = 0; w < W2; w++) d[W1 + w] = __builtin_amdgcn_alignbyte(g2[s][w + 1], g2[s][w], sh2);
					sepbad = __builtin_amdgcn_perm(g1[s][W1], g1[s][W1 - 1], sep_sel) != t.sep_val ? 1u : 0u;
				}
				fetch(tnext, s);
			}
			// classes: the index of the sheet letter a byte equals, `other` for every byte the sheet never uses
			u32 c[5] = {0u, 0u, 0u, 0u, 0u};
#pragma unroll
			for (int w = 0; w < W; w++) {
				c[w] = lut_classes(d[w], t);
			}
			bool found, amb;
			int tot = 0, idx, pfirst = 0, plast = 0;
			if constexpr (PAIR) {
				// each half on its own: its word -> (half id, distance); then the pair of ids -> (first, last) sample.  All three
				// tables are two-choice cuckoo tables of 8-byte entries that hold their key whole.
				const LutPairDev &pr = t.pair;
				constexpr int k2 = W1 < 4 ? W1 : 0, k3 = W1 + 1 < 5 ? W1 + 1 : 0;      // (PAIR: W1 <= 2)
				const u32 A1 = (W1 == 1 ? c[0] : lut_pack_half(c[0], c[1], t.wide)) & pr.keep1;
				const u32 A2 = (W1 == 1 ? c[k2] : lut_pack_half(c[k2], c[k3], t.wide)) & pr.keep2;
				const u32 x1 = lut_mix(A1, 0u, pr.seed1), x2 = lut_mix(A2, 0u, pr.seed2);
				// (the six tables' places in LDS are constants of the launch; a table-2 slot is the nb bits below the top nb: one v_bfe)
				auto at = [&](int base, u32 slot) { return *reinterpret_cast<const u32x2_t *>(sk_smem + base + (int)slot * 8); };
				const int b11 = ltab_off, b12 = b11 + (8 << pr.nb1), b21 = ltab_off + (int)pr.off2 * 8, b22 = b21 + (8 << pr.nb2);
				const int bp1 = ltab_off + (int)pr.offp * 8, bp2 = bp1 + (8 << pr.nbp);
				const u32x2_t e11 = at(b11, lut_slot(x1, pr.nb1)), e12 = at(b12, __builtin_amdgcn_ubfe(x1, 32u - 2u * (u32)pr.nb1, (u32)pr.nb1));
				const u32x2_t e21 = at(b21, lut_slot(x2, pr.nb2)), e22 = at(b22, __builtin_amdgcn_ubfe(x2, 32u - 2u * (u32)pr.nb2, (u32)pr.nb2));
				const bool f1 = e11[0] == A1 || e12[0] == A1, f2 = e21[0] == A2 || e22[0] == A2;
				const u32 v1 = e11[0] == A1 ? e11[1] : e12[1], v2 = e21[0] == A2 ? e21[1] : e22[1];
				tot = (int)(v1 >> 16) + (int)(v2 >> 16) + (int)sepbad;
				const u32 pk = (v1 & 0x3ffu) | ((v2 & 0x3ffu) << 10);
				const u32 xp = lut_mix(pk, 0u, pr.seedp);
				const u32x2_t ep1 = at(bp1, lut_slot(xp, pr.nbp)), ep2 = at(bp2, __builtin_amdgcn_ubfe(xp, 32u - 2u * (u32)pr.nbp, (u32)pr.nbp));
				const bool fp = ep1[0] == pk || ep2[0] == pk;
				const u32 pv = ep1[0] == pk ? ep1[1] : ep2[1];
				pfirst = (int)(pv & 0xffffu); plast = (int)(pv >> 16);
				found = f1 && f2 && fp && tot <= t.max_diff;
				amb = pfirst != plast;
				idx = pfirst;
			} else {
				u32 A, B;
				lut_pack(c, A, B, t.wide);
				A &= t.keepA; B &= t.keepB;
				const u32 x = lut_mix(A, B, t.seed);
				const u32 y = lut_side2(x, t.nb);                             // table 2 takes the next nb bits
				// at most 8 columns and no separator: B is zero for every key, and the tag compare alone decides (a free slot's
				// tag matches no key of its slot); the first word is fetched only for the distance SK_DETAIL wants
				constexpr bool kNoB = W2 == 0 && W1 <= 2;
				u32x2_t e1, e2;
				if (kNoB && !DETAIL) { e1[0] = e2[0] = 0u; e1[1] = entry_w1(lut_slot(x, t.nb)); e2[1] = entry2_w1(lut_slot(y, t.nb)); }
				else { e1 = entry(lut_slot(x, t.nb)); e2 = entry2(lut_slot(y, t.nb)); }
				const u32 m1 = (kNoB ? 0u : ((e1[0] ^ B) & 0x7fffffffu)) | ((e1[1] ^ x) & t.tag_mask);
				const u32 m2 = (kNoB ? 0u : ((e2[0] ^ B) & 0x7fffffffu)) | ((e2[1] ^ y) & t.tag_mask);
				const u32 w0 = m1 == 0u ? e1[0] : e2[0], w1 = m1 == 0u ? e1[1] : e2[1];
				found = m1 == 0u || m2 == 0u;
				// the table holds no key beyond max_diff: only a differing separator can push a hit over it
				if (W2 > 0) { tot = (int)(w0 >> 31) + (int)sepbad; found = found && tot <= t.max_diff; }
				else if (DETAIL) tot = (int)(w0 >> 31);
				idx = (int)__builtin_amdgcn_ubfe(w1, (u32)t.idx_shift, idx_bits);
				amb = (int)w1 < 0;
			}
			code[j] = found ? (amb ? kAssignAmbiguous : idx) : kAssignNone;
			if (DETAIL) {
				int first = idx, last = idx;
				if (PAIR) { first = pfirst; last = plast; }
				else if (found && amb) { first = t.amb[2 * idx]; last = t.amb[2 * idx + 1]; }
				dtot[j] = found ? tot : 255; dfirst[j] = found ? first : -1; dlast[j] = found ? last : -1;
			}
			if (active && found) atomicAdd(amb ? &hist[S + 2] : &hist[idx], 1u);   // S + 3 <= kMaxLdsHist: the histogram is always in LDS
			n_total += (u32)rows;
		}
		// the quad's outputs
		const int64_t ro = (int64_t)qb * (4 * kTileRows);
		const int rows_q = MANY ? rows_many(4 * qb, q_row_end) + rows_many(4 * qb + 1, q_row_end) + rows_many(4 * qb + 2, q_row_end) + rows_many(4 * qb + 3, q_row_end)
		                        : (int)(a.n - ro < 4 * kTileRows ? a.n - ro : 4 * kTileRows);
#pragma unroll
		for (int j = 0; j < 4; j++) tp[kTileRows * j + lane] = (u32)code[j];
		wave_lds_fence();
		const u32x4 cv = *reinterpret_cast<const u32x4 *>(tp + 4 * lane);
		__builtin_amdgcn_raw_buffer_store_b128(cv, make_rsrc(q_assign, ro * 4, rows_q * 4), lane * 16, 0, kLutStAux);
		if (DETAIL) {
			if (rows_q == 4 * kTileRows) {
				// (the same exchange for the three narrow columns, one after the other through the same 1 KiB: 4 bytes, 8 and 8 per lane)
				wave_lds_fence();
				uint8_t *tp8 = reinterpret_cast<uint8_t *>(tp);
#pragma unroll
				for (int j = 0; j < 4; j++) tp8[kTileRows * j + lane] = (uint8_t)dtot[j];
				wave_lds_fence();
				const u32 dv = tp[lane];
				__builtin_amdgcn_raw_buffer_store_b32(dv, make_rsrc(q_low, ro, rows_q), lane * 4, 0, kLutStAux);
				wave_lds_fence();
				unsigned short *tp16 = reinterpret_cast<unsigned short *>(tp);
#pragma unroll
				for (int j = 0; j < 4; j++) tp16[kTileRows * j + lane] = (unsigned short)dfirst[j];
				wave_lds_fence();
				const u32x2_t fv = *reinterpret_cast<const u32x2_t *>(tp + 2 * lane);
				__builtin_amdgcn_raw_buffer_store_b64(fv, make_rsrc(q_first, ro * 2, rows_q * 2), lane * 8, 0, kLutStAux);
				wave_lds_fence();
#pragma unroll
				for (int j = 0; j < 4; j++) tp16[kTileRows * j + lane] = (unsigned short)dlast[j];
				wave_lds_fence();
				const u32x2_t lv = *reinterpret_cast<const u32x2_t *>(tp + 2 * lane);
				__builtin_amdgcn_raw_buffer_store_b64(lv, make_rsrc(q_last, ro * 2, rows_q * 2), lane * 8, 0, kLutStAux);
				wave_lds_fence();
			} else {
				// the call's last, partial quad: a descriptor clips whole elements, so the narrow columns go row by row
#pragma unroll
				for (int j = 0; j < 4; j++) {
					const int rj = rows_q - kTileRows * j < 0 ? 0 : (rows_q - kTileRows * j > kTileRows ? kTileRows : rows_q - kTileRows * j);
					const int64_t rt = ro + kTileRows * j;
					__builtin_amdgcn_raw_buffer_store_b8((uint8_t)dtot[j], make_rsrc(q_low, rj ? rt : 0, rj), lane, 0, 0);
					__builtin_amdgcn_raw_buffer_store_b16((unsigned short)dfirst[j], make_rsrc(q_first, (rj ? rt : 0) * 2, rj * 2), lane * 2, 0, 0);
					__builtin_amdgcn_raw_buffer_store_b16((unsigned short)dlast[j], make_rsrc(q_last, (rj ? rt : 0) * 2, rj * 2), lane * 2, 0, 0);
				}
			}
		}
	}
	if (!a.counts_wide && !a.table.count_rep) lut_identified_from_hist(S, hist, lane);      // nobody folds behind this launch
	const WaveCounts wc = {n_total, 0u, 0u};
	// a few hundred workgroups end together, and an addition to an address that others add to takes about 10 ns: 512 x 19
	// of them into two lines were 4.5 of cfg 3's 31 us at 10 M reads.  When the counters are the ctx's they go to one of
	// sixteen copies with a line per counter (folded before anything reads them), else to the caller's vector as it is
	if (a.counts_wide && a.counts_wide_rows) flush_counts(S, a.counts_wide + (size_t)(blockIdx.x % (unsigned)a.counts_wide_rows) * (S + 3), lp, hist, lane, wc, 0);
	else if (a.counts_wide) flush_counts(S, a.counts_wide + (((size_t)(blockIdx.x & (kCountReplicas - 1)) * (S + 3)) << kCountWideShift), lp, hist, lane, wc, kCountWideShift);
	else flush_counts_spread(a.table, a.counts, lp, hist, lane, wc);
}

// The same lookup for rows of exactly 8 bytes on an 8-byte pitch (cfg 3), TWO rows per lane: a tile is 128 rows, a lane loads
// its two rows with one 16-byte load (1 KiB per wave instruction instead of 512 B, nt: read once); a wave's step is two
// consecutive tiles, whose four codes per lane leave as one 16-byte store after changing lanes through LDS; the per-tile scalar
// work (descriptors, clipping, counters) is paid once per 128 rows.  Everything per row is as above.
template <bool LDSTAB, bool DETAIL, bool MANY = false>
__global__ __launch_bounds__(LDSTAB ? 1024 : 256) void demux_lut8x2_kernel(const TileArgs a, const LdsPlan lp)
{
	typedef u32 u32x2_t __attribute__((ext_vector_type(2)));
	constexpr int kRows = 2 * kTileRows;
	const int lane = threadIdx.x & (kWave - 1);
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int nwave = blockDim.x >> 6;
	u32 *hist = reinterpret_cast<u32 *>(sk_smem + lp.hist_off);
	const LutDev &t = a.table.nbr;
	const int S = a.table.S;
	// MANY (round 6; TileArgs::many): the steps of SEVERAL batches in one launch — a batch's pointers are biased so that step gq of the
	// launch addresses the batch's own rows with the launch's indices (ManyBatch), its rows end at row_end; a wave's steps go up, so
	// two cursors (the batch of the step at hand, the batch of the step being fetched) only move forward.  The table is staged
	// once, no wave waits at a batch's end, and the next batch's first rows are on their way while the last of this one are looked up.
	const int64_t ntiles = MANY ? 2 * (int64_t)a.many_quads : (a.n + kRows - 1) / kRows;
	const int nt32 = (int)ntiles, last_rows = (int)(a.n - (ntiles - 1) * kRows);
	auto rows_of = [&](int ti) { return ti < nt32 - 1 ? kRows : (ti == nt32 - 1 ? last_rows : 0); };
	auto rows_many = [&](int ti, u32 row_end) {
		const int left = (int)(row_end - (u32)ti * (u32)kRows);
		return left <= 0 ? 0 : (left < kRows ? left : kRows);
	};
	// (a batch's fields are read when a cursor moves — once per batch and wave —, not per step: a scalar load per fetch waited for the
	// LDS as well, and the many-batch launch ran at 0.47 of the HBM peak where the calls back to back did 0.58)
	struct BatchView { const uint8_t *bc; u32 row_end; int q_end; };
	int cb = 0, fb = 0;                                                  // MANY: the batch of the step at hand / of the step being fetched
	BatchView cv = {nullptr, 0u, 0}, fv = {nullptr, 0u, 0};
	int32_t *c_assign = nullptr;
	uint8_t *c_low = nullptr;
	int16_t *c_first = nullptr, *c_last = nullptr;
	auto view_of = [&](int i) { const ManyBatch &m = a.many[i]; return BatchView{m.bc, m.row_end, m.q_end}; };
	auto seek = [&](int gq, int &cur, BatchView &v) {
		bool moved = false;
		while (cur + 1 < a.n_many && gq >= v.q_end) { cur++; v = view_of(cur); moved = true; }
		return moved;
	};
	if (MANY) {
		cv = fv = view_of(0);
		c_assign = a.many[0].assign; c_low = a.many[0].lowest_diff; c_first = a.many[0].first_idx; c_last = a.many[0].last_idx;
	}
	constexpr int kDepth = 2;                                            // halves of a step on their way (one step = 2 KiB per wave)
	u32x4 raw[kDepth];
	auto fetch = [&](int ti, int s) {
		if (MANY) {
			const int rows = rows_many(ti, fv.row_end);
			raw[s] = __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(fv.bc, (int64_t)(rows ? ti : 0) * (kRows * 8), rows * 8), lane * 16, 0, kLut8Aux);
			return;
		}
		const int rows = rows_of(ti);
		raw[s] = __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(a.bc, (int64_t)(rows ? ti : 0) * (kRows * 8), rows * 8), lane * 16, 0, kLut8Aux);
	};
	// a wave's step is two consecutive tiles = 256 consecutive rows, whose four codes per lane leave as ONE 16-byte store (as
	// demux_lut_kernel's quads: two ds_write_b64, one ds_read_b128 through 1 KiB of LDS per wave)
	const int qstep = (int)gridDim.x * nwave;
	int qb = (int)blockIdx.x * nwave + wave;
	if (MANY) seek(qb, fb, fv);
#pragma unroll
	for (int s = 0; s < kDepth; s++) fetch(2 * qb + s, s);
	for (int i = threadIdx.x; i < S + 3; i += blockDim.x) hist[i] = 0u;
	const int ltab_off = lp.tiles_off - lp.table_bytes;
	if (LDSTAB) {
		const int n16 = lp.table_bytes >> 4;
		for (int i0 = threadIdx.x; i0 < n16; i0 += 8 * (int)blockDim.x) {
			u32x4 tv[8];
#pragma unroll
			for (int j = 0; j < 8; j++) {
				const int i = i0 + j * (int)blockDim.x;
				if (i < n16) tv[j] = reinterpret_cast<const u32x4 *>(t.tab)[i];
			}
#pragma unroll
			for (int j = 0; j < 8; j++) {
				const int i = i0 + j * (int)blockDim.x;
				if (i < n16) *reinterpret_cast<u32x4 *>(sk_smem + ltab_off + i * 16) = tv[j];
			}
		}
	}
	__syncthreads();
	auto entry = [&](u32 slot) {
		if (LDSTAB) return *reinterpret_cast<const u32x2_t *>(sk_smem + ltab_off + (int)slot * 8);
		return *reinterpret_cast<const u32x2_t *>(t.tab + (size_t)slot * 2);
	};
	const int ltab_off2 = ltab_off + (t.mask + 1) * 8;           // table 2 behind table 1: its base is a constant, not an add per row
	auto entry2 = [&](u32 slot) {
		if (LDSTAB) return *reinterpret_cast<const u32x2_t *>(sk_smem + ltab_off2 + (int)slot * 8);
		return *reinterpret_cast<const u32x2_t *>(t.tab + (size_t)(t.mask + 1) * 2 + (size_t)slot * 2);
	};
	auto entry_w1 = [&](u32 slot) {                              // the second word alone: all a key without B needs (sk_lut.h)
		if (LDSTAB) return *reinterpret_cast<const u32 *>(sk_smem + ltab_off + 4 + (int)slot * 8);
		return t.tab[(size_t)slot * 2 + 1];
	};
	auto entry2_w1 = [&](u32 slot) {
		if (LDSTAB) return *reinterpret_cast<const u32 *>(sk_smem + ltab_off2 + 4 + (int)slot * 8);
		return t.tab[(size_t)(t.mask + 1) * 2 + (size_t)slot * 2 + 1];
	};
	const u32 idx_bits = (u32)__builtin_popcount(t.idx_mask);
	u32 n_total = 0;
	u32 *tp = reinterpret_cast<u32 *>(sk_smem + lp.tiles_off + wave * lp.tile_slot);
	const int nq32 = (nt32 + 1) >> 1;
	for (; qb < nq32; qb += qstep) {
		int code[2][2], tot[2][2], first[2][2], last[2][2];
		// where this step's rows end and its outputs go (MANY: the batch's; else the call's)
		u32 q_row_end = (u32)a.n;
		int32_t *q_assign = a.assign;
		uint8_t *q_low = a.lowest_diff;
		int16_t *q_first = a.first_idx, *q_last = a.last_idx;
		if (MANY) {
			if (seek(qb, cb, cv)) { const ManyBatch &mb = a.many[cb]; c_assign = mb.assign; c_low = mb.lowest_diff; c_first = mb.first_idx; c_last = mb.last_idx; }
			q_row_end = cv.row_end; q_assign = c_assign; q_low = c_low; q_first = c_first; q_last = c_last;
			seek(qb + qstep, fb, fv);
		}
#pragma unroll
		for (int s = 0; s < 2; s++) {
			const int ti = 2 * qb + s;
			const int rows = MANY ? rows_many(ti, q_row_end) : rows_of(ti);
			const u32x4 d = raw[s];
			fetch(2 * (qb + qstep) + s, s);
#pragma unroll
			for (int r = 0; r < 2; r++) {
				u32 c[5] = {0u, 0u, 0u, 0u, 0u};
#pragma unroll
				for (int w = 0; w < 2; w++) {
					c[w] = lut_classes(d[2 * r + w], t);
				}
				u32 A, B;
				lut_pack(c, A, B, t.wide);
				A &= t.keepA; B &= t.keepB;
				const u32 x = lut_mix(A, B, t.seed);
				const u32 y = lut_side2(x, t.nb);
				// 8 columns: B is zero for every key and the tag compare alone decides (a free slot's tag matches no key of its
				// slot, sk_lut.h); the first word is fetched only for the distance SK_DETAIL wants
				u32x2_t e1, e2;
				if (!DETAIL) { e1[0] = e2[0] = 0u; e1[1] = entry_w1(lut_slot(x, t.nb)); e2[1] = entry2_w1(lut_slot(y, t.nb)); }
				else { e1 = entry(lut_slot(x, t.nb)); e2 = entry2(lut_slot(y, t.nb)); }
				const u32 m1 = (e1[1] ^ x) & t.tag_mask, m2 = (e2[1] ^ y) & t.tag_mask;
				const u32 w0 = m1 == 0u ? e1[0] : e2[0], w1 = m1 == 0u ? e1[1] : e2[1];
				const bool found = m1 == 0u || m2 == 0u;
				const int idx = (int)__builtin_amdgcn_ubfe(w1, (u32)t.idx_shift, idx_bits);
				const bool amb = (int)w1 < 0;
				code[s][r] = found ? (amb ? kAssignAmbiguous : idx) : kAssignNone;
				if (DETAIL) {
					int f = idx, l = idx;
					if (found && amb) { f = t.amb[2 * idx]; l = t.amb[2 * idx + 1]; }
					tot[s][r] = found ? (int)(w0 >> 31) : 255;       // (the table holds no key beyond max_diff, and these rows have no separator)
					first[s][r] = found ? f : -1; last[s][r] = found ? l : -1;
				}
				const bool active = 2 * lane + r < rows;
				if (active && found) atomicAdd(amb ? &hist[S + 2] : &hist[idx], 1u);
			}
			n_total += (u32)rows;
		}
		// the step's outputs: lane l holds rows 2 l, 2 l + 1 of either tile; it leaves with rows 4 l ... 4 l + 3 of the step
		const int64_t ro = (int64_t)qb * (2 * kRows);
		const int rows_q = MANY ? rows_many(2 * qb, q_row_end) + rows_many(2 * qb + 1, q_row_end) : (int)(a.n - ro < 2 * kRows ? a.n - ro : 2 * kRows);
#pragma unroll
		for (int s = 0; s < 2; s++) {
			u32x2_t v;
			v[0] = (u32)code[s][0]; v[1] = (u32)code[s][1];
			*reinterpret_cast<u32x2_t *>(tp + kRows * s + 2 * lane) = v;
		}
		wave_lds_fence();
		const u32x4 cv = *reinterpret_cast<const u32x4 *>(tp + 4 * lane);
		__builtin_amdgcn_raw_buffer_store_b128(cv, make_rsrc(q_assign, ro * 4, rows_q * 4), lane * 16, 0, kLutStAux);
		if (DETAIL) {
			if (rows_q == 2 * kRows) {
				wave_lds_fence();
				unsigned short *tp16 = reinterpret_cast<unsigned short *>(tp);
#pragma unroll
				for (int s = 0; s < 2; s++) tp16[(kRows * s) / 2 + lane] = (unsigned short)((tot[s][0] & 0xff) | (tot[s][1] << 8));
				wave_lds_fence();
				const u32 dv = tp[lane];
				__builtin_amdgcn_raw_buffer_store_b32(dv, make_rsrc(q_low, ro, rows_q), lane * 4, 0, kLutStAux);
				wave_lds_fence();
#pragma unroll
				for (int s = 0; s < 2; s++) tp[(kRows * s) / 2 + lane] = ((u32)first[s][0] & 0xffffu) | ((u32)first[s][1] << 16);
				wave_lds_fence();
				const u32x2_t fv = *reinterpret_cast<const u32x2_t *>(tp + 2 * lane);
				__builtin_amdgcn_raw_buffer_store_b64(fv, make_rsrc(q_first, ro * 2, rows_q * 2), lane * 8, 0, kLutStAux);
				wave_lds_fence();
#pragma unroll
				for (int s = 0; s < 2; s++) tp[(kRows * s) / 2 + lane] = ((u32)last[s][0] & 0xffffu) | ((u32)last[s][1] << 16);
				wave_lds_fence();
				const u32x2_t lv = *reinterpret_cast<const u32x2_t *>(tp + 2 * lane);
				__builtin_amdgcn_raw_buffer_store_b64(lv, make_rsrc(q_last, ro * 2, rows_q * 2), lane * 8, 0, kLutStAux);
				wave_lds_fence();
			} else {
				// the call's last, partial step: a descriptor clips whole elements, so the narrow columns go row by row
#pragma unroll
				for (int s = 0; s < 2; s++) {
					const int rj = rows_q - kRows * s < 0 ? 0 : (rows_q - kRows * s > kRows ? kRows : rows_q - kRows * s);
					const int64_t rt = rj ? ro + kRows * s : 0;
					const rsrc_t rd = make_rsrc(q_low, rt, rj), rf = make_rsrc(q_first, rt * 2, rj * 2), rl = make_rsrc(q_last, rt * 2, rj * 2);
#pragma unroll
					for (int r = 0; r < 2; r++) {
						__builtin_amdgcn_raw_buffer_store_b8((uint8_t)tot[s][r], rd, lane * 2 + r, 0, 0);
						__builtin_amdgcn_raw_buffer_store_b16((unsigned short)first[s][r], rf, lane * 4 + 2 * r, 0, 0);
						__builtin_amdgcn_raw_buffer_store_b16((unsigned short)last[s][r], rl, lane * 4 + 2 * r, 0, 0);
					}
				}
			}
		}
	}
	if (!a.counts_wide && !a.table.count_rep) lut_identified_from_hist(S, hist, lane);      // nobody folds behind this launch
	const WaveCounts wc = {n_total, 0u, 0u};
	if (a.counts_wide && a.counts_wide_rows) flush_counts(S, a.counts_wide + (size_t)(blockIdx.x % (unsigned)a.counts_wide_rows) * (S + 3), lp, hist, lane, wc, 0);
	else if (a.counts_wide) flush_counts(S, a.counts_wide + (((size_t)(blockIdx.x & (kCountReplicas - 1)) * (S + 3)) << kCountWideShift), lp, hist, lane, wc, kCountWideShift);
	else flush_counts_spread(a.table, a.counts, lp, hist, lane, wc);
}

template <bool LDSTAB, bool DETAIL>
static const void *demux_lut_fn(int W1, int W2, bool direct)
{
	if (direct) return W1 == 1 ? reinterpret_cast<const void *>(demux_lut_kernel<1, 0, true, LDSTAB, DETAIL>)
	                           : reinterpret_cast<const void *>(demux_lut_kernel<2, 0, true, LDSTAB, DETAIL>);
	if (W2 > 0) return W1 == 1 ? reinterpret_cast<const void *>(demux_lut_kernel<1, 1, false, LDSTAB, DETAIL>)
	                           : reinterpret_cast<const void *>(demux_lut_kernel<2, 2, false, LDSTAB, DETAIL>);
	switch (W1) {
	case 1: return reinterpret_cast<const void *>(demux_lut_kernel<1, 0, false, LDSTAB, DETAIL>);
	case 2: return reinterpret_cast<const void *>(demux_lut_kernel<2, 0, false, LDSTAB, DETAIL>);
	case 3: return reinterpret_cast<const void *>(demux_lut_kernel<3, 0, false, LDSTAB, DETAIL>);
	case 4: return reinterpret_cast<const void *>(demux_lut_kernel<4, 0, false, LDSTAB, DETAIL>);
	default: return reinterpret_cast<const void *>(demux_lut_kernel<5, 0, false, LDSTAB, DETAIL>);
	}
}

// rows too long for an LDS tile: one thread per row straight from global memory (correct, not fast)
__global__ __launch_bounds__(256) void trim_rows_global_kernel(const uint8_t *__restrict__ qual, const uint16_t *__restrict__ len,
                                                               int stride, int64_t n, int m, uint16_t *__restrict__ lowest_k)
{
	for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
		const uint8_t *q = qual + r * (int64_t)stride;
		int l = len ? (int)len[r] : stride;
		int total = -50, lowest_total = -50, k = l, lowest_k_ = l;
		while (k > 0) {
			k -= 1;
			total += (int)(uint8_t)(q[k] - 33) - m;
			if (total > 0) break;
			if (total < lowest_total) { lowest_total = total; lowest_k_ = k; }
		}
		lowest_k[r] = (uint16_t)lowest_k_;
	}
}

// LDS layout + workgroup shape for a tile kernel; the grid comes from the occupancy API so that every
// workgroup of the persistent grid is resident (a queued workgroup would run its whole share of tiles late).
struct LaunchShape { LdsPlan lp; int grid, block, lds; };

static hipError_t plan_shape(const void *fn, const BarcodeDev &table, int64_t n, int image_bytes /* LDS image per wave */, bool with_tables, int max_nw,
                             int n_cu, int images, int max_wg, LaunchShape &out)
{
	static const int kLdsPerCu = 160 * 1024;
	LdsPlan lp{};
	lp.table_bytes = 0;
	if (with_tables) {
		int tb = table.bs ? table.bs_bytes : table.onehot ? table.S * padded_w(table.W) * 4 : table.S * table.L;
		lp.table_bytes = (tb + 15) & ~15;
	}
	lp.use_lds_hist = (with_tables && table.S + 3 <= kMaxLdsHist) ? 1 : 0;
	lp.hist_off = lp.table_bytes;
	lp.tiles_off = (lp.hist_off + (lp.use_lds_hist ? (table.S + 3) * 4 : 0) + 15) & ~15;
	lp.tile_slot = kLdsPad + ((image_bytes + 15) & ~15) + kLdsPad;
	// the shape search (occupancy queries, LDS opt-in) is cached per (device, kernel, LDS layout): small batches from
	// the command-line hosts launch thousands of times with the same shape
	struct Shape { int dev; const void *fn; int tiles_off, tile_slot, max_nw, max_wg, nw, wg; };
	static std::mutex cache_m;
	static std::vector<Shape> cache;
	// tuning knobs (tools/ablate.py, tools/waves_exp.py): read per launch so that one process can compare shapes on the same buffers
	const int env_nw = getenv("SK_TILE_WAVES") ? atoi(getenv("SK_TILE_WAVES")) : 0;
	const int env_wg = getenv("SK_TILE_WGS") ? atoi(getenv("SK_TILE_WGS")) : 0;
	const bool use_cache = env_nw == 0 && env_wg == 0;
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess) return e;
	int best_nw = 0, best_wg = 0, best_waves = 0;
	if (use_cache) {
		std::lock_guard<std::mutex> lk(cache_m);
		for (const Shape &c : cache)
			if (c.dev == dev && c.fn == fn && c.tiles_off == lp.tiles_off && c.tile_slot == lp.tile_slot * images && c.max_nw == max_nw && c.max_wg == max_wg) {
				best_nw = c.nw; best_wg = c.wg; best_waves = c.nw * c.wg;
				break;
			}
	}
	if (best_waves == 0) {
		e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsPerCu);
		if (e != hipSuccess) return e;
		for (int nw = (env_nw > 0 && env_nw <= 8) ? env_nw : max_nw; nw >= 1; nw >>= 1) {
			const int lds = lp.tiles_off + nw * images * lp.tile_slot;
			if (lds > kLdsPerCu) continue;
			int wg = 0;
			e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg, fn, kWave * nw, (size_t)lds);
			if (e != hipSuccess) return e;
			if (env_wg > 0) { if (wg > env_wg) wg = env_wg; }
			else if (max_wg > 0 && wg > max_wg) wg = max_wg;
			if (wg * nw > best_waves) { best_waves = wg * nw; best_nw = nw; best_wg = wg; }
		}
		if (best_waves > 0 && use_cache) {
			std::lock_guard<std::mutex> lk(cache_m);
			cache.push_back({dev, fn, lp.tiles_off, lp.tile_slot * images, max_nw, max_wg, best_nw, best_wg});
		}
	}
	if (best_waves == 0) return hipErrorInvalidValue;
	out.lds = lp.tiles_off + best_nw * images * lp.tile_slot;
	int64_t ntiles = (n + kTileRows - 1) / kTileRows;
	int64_t want = (ntiles + best_nw - 1) / best_nw;
	int64_t cap = (int64_t)n_cu * best_wg;
	out.grid = (int)(want < cap ? want : cap);
	out.block = kWave * best_nw;
	out.lp = lp;
	return hipSuccess;
}

static hipError_t plan_and_launch(const void *fn, const TileArgs &b, int row_bytes, bool with_tables, int max_nw, int n_cu, hipStream_t st, int images = 1,
                                  int max_wg = 0)
{
	LaunchShape sh;
	hipError_t e = plan_shape(fn, b.table, b.n, kTileRows * row_bytes, with_tables, max_nw, n_cu, images, max_wg, sh);
	if (e != hipSuccess) return e;
	TileArgs bb = b;
	void *kargs[] = {(void *)&bb, (void *)&sh.lp};
	return hipLaunchKernel(fn, dim3(sh.grid), dim3(sh.block), kargs, sh.lds, st);
}

template <bool DEMUX, int SLOTS>
static const void *tile_blocked_fn(int mode)
{
	switch (mode) {
	case 0: return reinterpret_cast<const void *>(tile_blocked_kernel<0, DEMUX, SLOTS>);
	case 1: return reinterpret_cast<const void *>(tile_blocked_kernel<1, DEMUX, SLOTS>);
	case 2: return reinterpret_cast<const void *>(tile_blocked_kernel<2, DEMUX, SLOTS>);
	case 3: return reinterpret_cast<const void *>(tile_blocked_kernel<3, DEMUX, SLOTS>);
	default: return reinterpret_cast<const void *>(tile_blocked_kernel<4, DEMUX, SLOTS>);
	}
}

bool blocked_shape_ok(const BlockedArgs &a)
{
	if (a.n_mates < 1 || a.n_mates > 2 || a.stride < 1 || a.stride > kMaxTileStride) return false;
	if (a.out_seq[0] < 0 && a.out_lowest_k[0] < 0) return false;
	if (a.in_bc >= 0) {
		if (!a.table.bs || a.table.G > 4 || a.table.S <= 0 || kTileRows * a.bc_stride > 2048) return false;
	}
	return true;
}

hipError_t launch_tile_blocked(const BlockedArgs &a, int n_cu, hipStream_t st)
{
	if (a.n <= 0) return hipSuccess;
	if (!blocked_shape_ok(a)) return hipErrorInvalidValue;
	if ((a.n + kTileRows - 1) / kTileRows > (int64_t)1 << 30) return hipErrorInvalidValue;      // tiles are counted in 32 bits
	const bool demux = a.in_bc >= 0;
	// chunks in flight per stream: 2 / 3 / 4 measure the same on the two-mate pass (10.56-10.64 ms), 5 is slower
	const int slots = a.n_mates == 1 ? SK_SLOTS1 : kSlots;
	const void *fn;
	if (a.n_mates == 1) fn = demux ? tile_blocked_fn<true, SK_SLOTS1>(a.qc.mode) : tile_blocked_fn<false, SK_SLOTS1>(a.qc.mode);
	else fn = demux ? tile_blocked_fn<true, kSlots>(a.qc.mode) : tile_blocked_fn<false, kSlots>(a.qc.mode);
	const int nchunkp = (((kTileRows * a.stride + 1023) >> 10) + slots - 1) / slots * slots;
	int image = nchunkp * 1024;                                       // the LDS image takes whole chunks, trim or not
	if (demux && image < 2048) image = 2048;
	const int64_t ntiles = (a.n + kTileRows - 1) / kTileRows;
	const int max_wg = a.n_mates == 1 ? 3 : (ntiles < (int64_t)n_cu * 4 * 2 * 16 ? 3 : 2);   // as launch_tile_pass
	LaunchShape sh;
	hipError_t e = plan_shape(fn, a.table, a.n, image, demux, 4, n_cu, 1, max_wg, sh);
	if (e != hipSuccess) return e;
	BlockedArgs bb = a;
	void *kargs[] = {(void *)&bb, (void *)&sh.lp};
	return hipLaunchKernel(fn, dim3(sh.grid), dim3(sh.block), kargs, sh.lds, st);
}

template <bool DEMUX, int SLOTS>
static const void *tile_pass_fn(int mode)
{
	switch (mode) {
	case 0: return reinterpret_cast<const void *>(tile_pass_kernel<0, DEMUX, SLOTS>);
	case 1: return reinterpret_cast<const void *>(tile_pass_kernel<1, DEMUX, SLOTS>);
	case 2: return reinterpret_cast<const void *>(tile_pass_kernel<2, DEMUX, SLOTS>);
	case 3: return reinterpret_cast<const void *>(tile_pass_kernel<3, DEMUX, SLOTS>);
	default: return reinterpret_cast<const void *>(tile_pass_kernel<4, DEMUX, SLOTS>);
	}
}

// ONE statement of when the barcode phase rides in the tile pass: launch_tile_pass decides with it, and the C-ABI layer asks it
// whether a call needs the neighbourhood table built (a call that does not fuse takes the table for its barcode launch; were
// the two to disagree, no table would be there and the S x L matchers would run instead: correct, 10-50 x slower).
// any_mate = some mate has work; a stride beyond kMaxTileStride sends the mates to the fallback kernels, which fuse nothing.
// SK_NO_FUSED_DEMUX (tests, tools) is read once per process.
bool tile_pass_fuses_demux(bool has_bc, bool any_mate, int stride, bool has_bitsliced, int G, int S, int bc_stride)
{
	static const bool env_no_fuse = getenv("SK_NO_FUSED_DEMUX") != nullptr;
	return has_bc && any_mate && stride <= kMaxTileStride && has_bitsliced && G <= 4 && S > 0 && kTileRows * bc_stride <= 2048 && !env_no_fuse;
}

// does a barcode-phase launch of these arguments take the sheet's lookup table (one lookup per read; with the ctx's wide counters its
// adds are atomics into lines of their own and nothing is folded per launch)?
bool tile_pass_demux_by_table(const TileArgs &b)
{
	const bool env_no_hash = getenv("SK_NO_HASH_DEMUX") != nullptr;
	const bool want_detail = b.lowest_diff || b.first_idx || b.last_idx;
	return (b.table.nbr.tab || b.table.nbr.pair.tab) && (!want_detail || b.detail_matched) && kTileRows * b.bc_stride <= 2048 && !env_no_hash;
}

hipError_t launch_tile_pass(const TileArgs &a, int n_cu, hipStream_t st)
{
	if (a.n <= 0) return hipSuccess;
	if ((a.n + kTileRows - 1) / kTileRows > (int64_t)1 << 30) return hipErrorInvalidValue;      // the kernels count tiles in 32 bits (2^36 rows: no device holds them)
	TileArgs b = a;
	bool any_mate = false;
	for (int mi = 0; mi < b.n_mates; mi++) any_mate = any_mate || b.mate[mi].out_seq || b.mate[mi].lowest_k;
	// rows that do not fit an LDS tile: trim falls back to the row-per-thread kernel, mask to the flat one
	if (any_mate && b.stride > kMaxTileStride) {
		for (int mi = 0; mi < b.n_mates; mi++) {
			MateDev &mt = b.mate[mi];
			if (mt.out_seq) {
				hipError_t e = launch_mask_flat(mt.seq, mt.qual, mt.out_seq, b.n * (int64_t)b.stride, b.qc, n_cu, st);
				if (e != hipSuccess) return e;
			}
			if (mt.lowest_k) {
				int64_t want = (b.n + 255) / 256;
				int grid = (int)(want < (int64_t)n_cu * 8 ? want : (int64_t)n_cu * 8);
				trim_rows_global_kernel<<<grid, 256, 0, st>>>(mt.qual, mt.len, b.stride, b.n, b.qc.min_baseq, mt.lowest_k);
				hipError_t e = hipGetLastError();
				if (e != hipSuccess) return e;
			}
			mt.out_seq = nullptr; mt.lowest_k = nullptr;
		}
		any_mate = false;
	}
	// the barcode phase rides in the tile pass when the bit-sliced matcher applies (S <= 128) and the tile's
	// barcodes fit two 1 KiB register chunks; otherwise it is its own launch beside the mate pass
	const bool fuse_demux = tile_pass_fuses_demux(b.bc != nullptr, any_mate, b.stride, b.table.bs != nullptr, b.table.G, b.table.S, b.bc_stride);
	if (b.bc && !fuse_demux) {
		// the sheet has a neighbourhood table and the call wants the decision alone, or the detail columns of matched rows
		// only (TileArgs::detail_matched): one lookup per read
		const bool env_no_hash = getenv("SK_NO_HASH_DEMUX") != nullptr;
		const bool want_detail = b.lowest_diff || b.first_idx || b.last_idx;
		const bool by_table = tile_pass_demux_by_table(b);
		hipError_t e;
		TileArgs bb = b;
		if (b.many && !by_table) return hipErrorNotSupported;      // (many batches in one launch: the table's kernels only)
		if (by_table) {
			const LutDev &t = b.table.nbr;
			// rows on dword boundaries whose key is one or two dwords: lane r loads row r as it lies (one 8 B/lane load for 8-byte
			// rows); every other shape gathers (demux_lut_kernel).  (SK_DEMUX_DIRECT=0 / SK_DEMUX_LDSTAB=0 force the other forms:
			// the tests run every kernel on the same inputs)
			const bool pair = t.pair.tab != nullptr;                  // the factored form (sk_lut.h): its three tables always in LDS
			const char *env_direct = getenv("SK_DEMUX_DIRECT");
			const bool direct = !pair && (b.bc_stride & 3) == 0 && t.W2 == 0 && t.W1 <= 2 && 4 * t.W1 <= b.bc_stride && (!env_direct || atoi(env_direct) != 0);
			const char *env_ldstab = getenv("SK_DEMUX_LDSTAB");
			const int table_bytes = pair ? t.pair.bytes : (t.mask + 1) * 2 * 8;
			const bool ldstab = pair || (table_bytes <= (128 << 10) && (!env_ldstab || atoi(env_ldstab) != 0));
			// rows of exactly 8 key bytes on an 8-byte pitch: two rows per lane (demux_lut8x2_kernel) — always when the detail columns
			// are written too, and for the decision alone from 1.5 M rows (after round 4's instruction diet: 10 M rows 20.4 us
			// against 23.6 with one row per lane, 2 M 8.3 / 9.5, 6 M 14.6 / 16.9, 100 M the same; 1 M 7.8 / 7.3: a workgroup's
			// first tile is its whole share there).  SK_DEMUX_ROWS2=0 / 1 force the choice (tools/demux_ab.py, tests).
			const char *env_rows2 = getenv("SK_DEMUX_ROWS2");
			const bool rows2 = direct && t.W1 == 2 && b.bc_stride == 8 && (env_rows2 ? atoi(env_rows2) != 0 : (want_detail || b.n >= 1500000));
			const void *fn = pair ? (t.W1 == 1 ? (want_detail ? reinterpret_cast<const void *>(demux_lut_kernel<1, 1, false, true, true, true>) : reinterpret_cast<const void *>(demux_lut_kernel<1, 1, false, true, false, true>))
			                                   : (want_detail ? reinterpret_cast<const void *>(demux_lut_kernel<2, 2, false, true, true, true>) : reinterpret_cast<const void *>(demux_lut_kernel<2, 2, false, true, false, true>)))
			                 : rows2 ? (ldstab ? (want_detail ? reinterpret_cast<const void *>(demux_lut8x2_kernel<true, true>) : reinterpret_cast<const void *>(demux_lut8x2_kernel<true, false>))
			                                 : (want_detail ? reinterpret_cast<const void *>(demux_lut8x2_kernel<false, true>) : reinterpret_cast<const void *>(demux_lut8x2_kernel<false, false>)))
			                 : ldstab ? (want_detail ? demux_lut_fn<true, true>(t.W1, t.W2, direct) : demux_lut_fn<true, false>(t.W1, t.W2, direct))
			                          : (want_detail ? demux_lut_fn<false, true>(t.W1, t.W2, direct) : demux_lut_fn<false, false>(t.W1, t.W2, direct));
			if (b.many) {
				// many batches in one launch: the two kernels that serve the benchmark's sheets have the form; any other shape is the caller's loop
				if (!ldstab) return hipErrorNotSupported;
				// (the gathered-row kernel has the form too and keeps it for A/B — SK_LUT_MANY_GATHER=1 — but its many-batch launch is SLOWER than its
				// launches back to back, which is what the caller's loop does: 96 dual-index, 4 x 10 M rows from HBM, 0.51 of the HBM peak against 0.56)
				const char *mg = getenv("SK_LUT_MANY_GATHER");
				const bool many_gather = mg && atoi(mg) != 0;
				if (rows2) fn = want_detail ? reinterpret_cast<const void *>(demux_lut8x2_kernel<true, true, true>) : reinterpret_cast<const void *>(demux_lut8x2_kernel<true, false, true>);
				else if (many_gather && !direct && t.W1 == 2 && t.W2 == 2)
					fn = pair ? (want_detail ? reinterpret_cast<const void *>(demux_lut_kernel<2, 2, false, true, true, true, true>) : reinterpret_cast<const void *>(demux_lut_kernel<2, 2, false, true, false, true, true>))
					          : (want_detail ? reinterpret_cast<const void *>(demux_lut_kernel<2, 2, false, true, true, false, true>) : reinterpret_cast<const void *>(demux_lut_kernel<2, 2, false, true, false, false, true>));
				else return hipErrorNotSupported;
			}
			LdsPlan lp{};
			lp.use_lds_hist = 1;                                      // S + 3 <= kMaxLdsHist
			lp.hist_off = 0;
			lp.table_bytes = ldstab ? table_bytes : 0;
			lp.tiles_off = (((b.table.S + 3) * 4 + 15) & ~15) + lp.table_bytes;
			lp.tile_slot = 4 * kTileRows * 4;                         // rows come straight from memory: no image; 1 KiB per wave in which a quad's codes change lanes
			// sixteen waves share a table copy; eight once the call is long and the table leaves room for one workgroup per CU only
			// (rows from HBM, 100 M pairs of the 96 dual-index sheet: 381 us against 398; shorter calls want the waves)
			int nw = ldstab ? ((lp.table_bytes > (64 << 10) && b.n >= 4000000) ? 8 : 16) : 4;
			// tuning knobs (tools/lut_cold_ab.py), read once per process; the kernels that keep their table in the vector cache are
			// compiled for 256 threads, so they take 4 waves at most
			static const int env_nw = [] { const char *v = getenv("SK_LUT_NW"); return v ? atoi(v) : 0; }();
			static const int env_wg = [] { const char *v = getenv("SK_LUT_WG"); return v ? atoi(v) : 0; }();
			if (env_nw >= 1 && env_nw <= (ldstab ? 16 : 4)) nw = env_nw;
			while (nw > 4 && lp.tiles_off + nw * lp.tile_slot > 160 * 1024) nw >>= 1;
			const int lds = lp.tiles_off + nw * lp.tile_slot;
			struct Occ { int dev; const void *fn; int lds, nw, wg; };
			static std::mutex occ_m;
			static std::vector<Occ> occ;
			int dev = 0, wg = 0;
			e = hipGetDevice(&dev);
			if (e != hipSuccess) return e;
			{
				std::lock_guard<std::mutex> lk(occ_m);
				for (const Occ &o : occ) if (o.dev == dev && o.fn == fn && o.lds == lds && o.nw == nw) wg = o.wg;
			}
			if (wg == 0) {
				e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
				if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg, fn, kWave * nw, (size_t)lds);
				if (e != hipSuccess) return e;
				if (wg < 1) return hipErrorInvalidValue;
				std::lock_guard<std::mutex> lk(occ_m);
				occ.push_back({dev, fn, lds, nw, wg});
			}
			// sixteen waves per CU whatever the table leaves room for: with its rows coming from HBM a 10 M-row call of the cfg 3 sheet
			// took 25.3 us on sixteen and 26.7 on thirty-two (each wave's first load is a full HBM round trip before anything
			// moves, and twice the waves end in twice the stragglers); 100 M rows the same (tools/lut_cold_ab.py)
			if (ldstab && wg > 1) wg = 1;
			if (env_wg >= 1) wg = env_wg;
			const int tile_rows = 4 * kTileRows;                      // a wave's unit: 256 rows in either kernel
			const int64_t ntiles = b.many ? (int64_t)b.many_quads : (b.n + tile_rows - 1) / tile_rows, want = (ntiles + nw - 1) / nw, cap = (int64_t)n_cu * wg;
			const int64_t grid = want < cap ? want : cap;
			// a few hundred workgroups add to the counters directly; thousands go through the spread copies and the fold
			if (grid <= 512 || bb.counts_wide) bb.table.count_rep = nullptr;
			void *kargs[] = {(void *)&bb, (void *)&lp};
			e = hipLaunchKernel(fn, dim3((unsigned)grid), dim3(kWave * nw), kargs, lds, st);
		} else {
			e = plan_and_launch(reinterpret_cast<const void *>(demux_tile_kernel), b, b.bc_stride, true, 4, n_cu, st);
		}
		if (e == hipSuccess && !(by_table && bb.counts_wide)) e = launch_counts_fold(bb, by_table, st);      // (the wide counters are folded when somebody reads them)
		if (e != hipSuccess) return e;
	}
	if (!any_mate) return hipSuccess;
	int row_bytes = 0;
	for (int mi = 0; mi < b.n_mates; mi++)
		if (b.mate[mi].lowest_k) row_bytes = b.stride;          // only the trim scan needs the LDS image
	const int64_t total = b.n * (int64_t)b.stride;
	if (total & 3)
		for (int mi = 0; mi < b.n_mates; mi++)
			if (b.mate[mi].out_seq) mask_tail_kernel<<<1, 4, 0, st>>>(b.mate[mi].seq, b.mate[mi].qual, b.mate[mi].out_seq, total, b.qc.min_baseq);
	if (fuse_demux && b.bc_stride > row_bytes) row_bytes = b.bc_stride;      // the barcode phase puts the tile's barcodes into the wave's LDS slot
	// Shape of the phase kernel: workgroups of four waves, and how many of them a CU holds depends on what the pass is
	// bound by (tools/waves_exp.py, tools/small_n.py; same buffers, one process):
	//  * two mates, streaming-bound: TWO per CU (eight resident waves).  More waves keep more requests in flight than
	//    HBM serves well: 62.5 M clusters 9.52 ms with 16 waves, 9.31 ms with 8; at 16 M clusters +5 % for the fused
	//    pass, +14 % for mask + trim of two mates.  Short inputs (fewer than 16 tiles per wave) take three, to shorten
	//    the tail: 1 M clusters 176 -> 160 us;
	//  * one mate with the mask: three (1 M reads 101 -> 89 us, no difference at 16 M);
	//  * trim alone, bound by the scan's VALU work: four (16 M reads of full-length scans 1.19 -> 0.80 ms).
	const int64_t ntiles = (b.n + kTileRows - 1) / kTileRows;
	const int wg_two_mates = ntiles < (int64_t)n_cu * 4 * 2 * 16 ? 3 : 2;
	if (fuse_demux) return plan_and_launch(tile_pass_fn<true, kSlots>(b.qc.mode), b, row_bytes, true, 4, n_cu, st, 1, wg_two_mates);
	b.bc = nullptr;
	// One mate = two read streams at most: keep four chunks per stream in flight instead of two (mask + trim of one
	// mate 4.9 -> 5.1 TB/s at 16 M x 150; no gain with two mates or the barcode phase).  Trim alone reads ONE stream and
	// spends most of its time in the scan: there the whole next tile (ten chunks at 150 bp) is kept in flight
	// (3.6 -> 3.9 -> 4.3 TB/s with 2 / 4 / 10 slots on read-like qualities).
	int active = 0;
	bool any_mask = false;
	for (int mi = 0; mi < b.n_mates; mi++) {
		active += (b.mate[mi].out_seq || b.mate[mi].lowest_k) ? 1 : 0;
		any_mask = any_mask || b.mate[mi].out_seq;
	}
	if (active == 1 && !any_mask) return plan_and_launch(tile_pass_fn<false, 10>(b.qc.mode), b, row_bytes, false, 4, n_cu, st, 1, 4);
	if (active == 1) return plan_and_launch(tile_pass_fn<false, SK_SLOTS1>(b.qc.mode), b, row_bytes, false, 4, n_cu, st, 1, 3);
	return plan_and_launch(tile_pass_fn<false, kSlots>(b.qc.mode), b, row_bytes, false, 4, n_cu, st, 1, wg_two_mates);
}

// ---------------------------------------------------------------------------------------------------
// S1 + H1: BAM flag counters and |TLEN| histogram.
// src/sam_statistics.rs:63-69; src/sam_fragment_lengths.rs:29-43.
// out = u64[3 counters][1 hist_total][max_frag+1 bins]
// ---------------------------------------------------------------------------------------------------
// per-record work shared by both BAM kernels
struct BamAcc { u32 total, aligned, dup, hist; };

__device__ __forceinline__ void bam_record(u32 f, int32_t tid, int32_t mtid, int32_t tl, bool valid, int32_t max_frag, int want_counters,
                                           int want_hist, int lds_bins, u32 *lh, unsigned long long *out, BamAcc &acc)
{
	if (want_counters) {
		const bool primary = valid && (f & (0x100u | 0x800u)) == 0u;                       // src/sam_statistics.rs:64
		const bool mapped = primary && !(f & 0x4u);                                         // :66
		acc.total += primary ? 1u : 0u;
		acc.aligned += mapped ? 1u : 0u;
		acc.dup += (mapped && (f & 0x400u)) ? 1u : 0u;                                      // :69
	}
	if (want_hist) {
		// paired, first, both mapped, not dup/secondary/supplementary, same reference: src/sam_fragment_lengths.rs:30-37
		const bool flags_ok = valid && (f & (0x1u | 0x40u | 0x4u | 0x8u | 0x400u | 0x100u | 0x800u)) == (0x1u | 0x40u);
		// |tlen| as insert_size().abs() on i64: INT32_MIN maps to 2^31, above any max_frag
		const u32 af = tl < 0 ? (u32)0 - (u32)tl : (u32)tl;
		if (flags_ok && tid == mtid && af <= (u32)max_frag) {
			acc.hist += 1u;
			if ((int)af < lds_bins) atomicAdd(&lh[af], 1u);
			else atomicAdd(&out[4 + af], 1ull);
		}
	}
}

__device__ __forceinline__ void bam_flush(const BamAcc &acc, u32 *wg_cnt, u32 *lh, int lds_bins, unsigned long long *out)
{
	if (acc.total) atomicAdd(&wg_cnt[0], acc.total);
	if (acc.aligned) atomicAdd(&wg_cnt[1], acc.aligned);
	if (acc.dup) atomicAdd(&wg_cnt[2], acc.dup);
	if (acc.hist) atomicAdd(&wg_cnt[3], acc.hist);
	__syncthreads();
	if (threadIdx.x < 4 && wg_cnt[threadIdx.x]) atomicAdd(&out[threadIdx.x], (unsigned long long)wg_cnt[threadIdx.x]);
	for (int i = threadIdx.x; i < lds_bins; i += blockDim.x) {
		u32 c = lh[i];
		if (c) atomicAdd(&out[4 + i], (unsigned long long)c);
	}
}

// S1 + H1, 8 records per lane and iteration: the four columns arrive as 16-byte buffer loads (1 + 2 + 2 + 2 per
// lane = 7 KiB per wave in flight), clipped by per-iteration descriptors, so there is no tail branch around a load.
// Columns must be 16-byte aligned (anything else takes bam_flag_tlen_scalar_kernel).
__global__ __launch_bounds__(256) void bam_flag_tlen_kernel(const uint16_t *__restrict__ flag, const int32_t *__restrict__ tid,
                                                            const int32_t *__restrict__ mtid, const int32_t *__restrict__ tlen,
                                                            int64_t n, int32_t max_frag, unsigned long long *__restrict__ out,
                                                            int want_counters, int want_hist, int lds_bins)
{
	u32 *lh = reinterpret_cast<u32 *>(sk_smem);
	__shared__ u32 wg_cnt[4];
	if (threadIdx.x < 4) wg_cnt[threadIdx.x] = 0u;
	for (int i = threadIdx.x; i < lds_bins; i += blockDim.x) lh[i] = 0u;
	__syncthreads();
	BamAcc acc = {0u, 0u, 0u, 0u};
	const int64_t per_it = (int64_t)blockDim.x * 8;                       // records per workgroup iteration
	for (int64_t base = (int64_t)blockIdx.x * per_it; base < n; base += (int64_t)gridDim.x * per_it) {
		const int64_t left = n - base;
		const int nrec = (int)(left < per_it ? left : per_it);
		const rsrc_t rf = make_rsrc(flag, base * 2, (nrec * 2 + 3) & ~3);
		const rsrc_t rt = make_rsrc(want_hist ? tid : nullptr, base * 4, nrec * 4);
		const rsrc_t rm = make_rsrc(want_hist ? mtid : nullptr, base * 4, nrec * 4);
		const rsrc_t rl = make_rsrc(want_hist ? tlen : nullptr, base * 4, nrec * 4);
		const int r0 = threadIdx.x * 8;
		const u32x4 f = __builtin_amdgcn_raw_buffer_load_b128(rf, r0 * 2, 0, 0);
		const u32x4 t0 = __builtin_amdgcn_raw_buffer_load_b128(rt, r0 * 4, 0, 0), t1 = __builtin_amdgcn_raw_buffer_load_b128(rt, r0 * 4 + 16, 0, 0);
		const u32x4 m0 = __builtin_amdgcn_raw_buffer_load_b128(rm, r0 * 4, 0, 0), m1 = __builtin_amdgcn_raw_buffer_load_b128(rm, r0 * 4 + 16, 0, 0);
		const u32x4 l0 = __builtin_amdgcn_raw_buffer_load_b128(rl, r0 * 4, 0, 0), l1 = __builtin_amdgcn_raw_buffer_load_b128(rl, r0 * 4 + 16, 0, 0);
#pragma unroll
		for (int j = 0; j < 8; j++) {
			const u32 fj = (f[j >> 1] >> (16 * (j & 1))) & 0xffffu;
			const int32_t tj = (int32_t)(j < 4 ? t0[j & 3] : t1[j & 3]);
			const int32_t mj = (int32_t)(j < 4 ? m0[j & 3] : m1[j & 3]);
			const int32_t lj = (int32_t)(j < 4 ? l0[j & 3] : l1[j & 3]);
			bam_record(fj, tj, mj, lj, r0 + j < nrec, max_frag, want_counters, want_hist, lds_bins, lh, out, acc);
		}
	}
	bam_flush(acc, wg_cnt, lh, lds_bins, out);
}

// the same reduction with one record per lane and plain loads: columns of any alignment
__global__ __launch_bounds__(256) void bam_flag_tlen_scalar_kernel(const uint16_t *__restrict__ flag, const int32_t *__restrict__ tid,
                                                                   const int32_t *__restrict__ mtid, const int32_t *__restrict__ tlen,
                                                                   int64_t n, int32_t max_frag, unsigned long long *__restrict__ out,
                                                                   int want_counters, int want_hist, int lds_bins)
{
	u32 *lh = reinterpret_cast<u32 *>(sk_smem);
	__shared__ u32 wg_cnt[4];
	if (threadIdx.x < 4) wg_cnt[threadIdx.x] = 0u;
	for (int i = threadIdx.x; i < lds_bins; i += blockDim.x) lh[i] = 0u;
	__syncthreads();
	BamAcc acc = {0u, 0u, 0u, 0u};
	const int64_t step = (int64_t)gridDim.x * blockDim.x;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step)
		bam_record(flag[i], want_hist ? tid[i] : 0, want_hist ? mtid[i] : 0, want_hist ? tlen[i] : 0, true, max_frag, want_counters, want_hist,
		           lds_bins, lh, out, acc);
	bam_flush(acc, wg_cnt, lh, lds_bins, out);
}

// f2: src/sam_fragments.rs:27-38 — the forward mate of a converging, mapped, primary, non-duplicate, QC-passing pair on
// one reference with min_size <= |tlen| <= max_size.  Output is a bit mask (bit j of byte k <=> record 8k+j): each lane
// decides its 8 consecutive records and stores one byte; the host turns set bits into BED lines (:41).
__global__ __launch_bounds__(256) void bam_fragments_kernel(const uint16_t *__restrict__ flag, const int32_t *__restrict__ tid,
                                                            const int32_t *__restrict__ mtid, const int32_t *__restrict__ tlen,
                                                            int64_t n, u32 min_size, u32 max_size, uint8_t *__restrict__ keep_bits,
                                                            unsigned long long *__restrict__ kept)
{
	__shared__ u32 wg_kept;
	if (threadIdx.x == 0) wg_kept = 0u;
	__syncthreads();
	u32 count = 0;
	const int64_t per_it = (int64_t)blockDim.x * 8;
	for (int64_t base = (int64_t)blockIdx.x * per_it; base < n; base += (int64_t)gridDim.x * per_it) {
		const int64_t left = n - base;
		const int nrec = (int)(left < per_it ? left : per_it);
		const rsrc_t rf = make_rsrc(flag, base * 2, (nrec * 2 + 3) & ~3);
		const rsrc_t rt = make_rsrc(tid, base * 4, nrec * 4);
		const rsrc_t rm = make_rsrc(mtid, base * 4, nrec * 4);
		const rsrc_t rl = make_rsrc(tlen, base * 4, nrec * 4);
		const rsrc_t ro = make_rsrc(keep_bits, base >> 3, (nrec + 7) >> 3);
		const int r0 = threadIdx.x * 8;
		const u32x4 f = __builtin_amdgcn_raw_buffer_load_b128(rf, r0 * 2, 0, 0);
		const u32x4 t0 = __builtin_amdgcn_raw_buffer_load_b128(rt, r0 * 4, 0, 0), t1 = __builtin_amdgcn_raw_buffer_load_b128(rt, r0 * 4 + 16, 0, 0);
		const u32x4 m0 = __builtin_amdgcn_raw_buffer_load_b128(rm, r0 * 4, 0, 0), m1 = __builtin_amdgcn_raw_buffer_load_b128(rm, r0 * 4 + 16, 0, 0);
		const u32x4 l0 = __builtin_amdgcn_raw_buffer_load_b128(rl, r0 * 4, 0, 0), l1 = __builtin_amdgcn_raw_buffer_load_b128(rl, r0 * 4 + 16, 0, 0);
		u32 bits = 0;
#pragma unroll
		for (int j = 0; j < 8; j++) {
			const u32 fj = (f[j >> 1] >> (16 * (j & 1))) & 0xffffu;
			const int32_t tj = (int32_t)(j < 4 ? t0[j & 3] : t1[j & 3]);
			const int32_t mj = (int32_t)(j < 4 ? m0[j & 3] : m1[j & 3]);
			const int32_t lj = (int32_t)(j < 4 ? l0[j & 3] : l1[j & 3]);
			// paired, both mapped, not dup / secondary / supplementary / QC-fail, this mate forward, mate reverse
			const bool flags_ok = (fj & (0x1u | 0x4u | 0x8u | 0x400u | 0x100u | 0x800u | 0x10u | 0x20u | 0x200u)) == (0x1u | 0x20u);
			const u32 af = lj < 0 ? (u32)0 - (u32)lj : (u32)lj;
			const bool keep = (r0 + j < nrec) && flags_ok && tj == mj && af >= min_size && af <= max_size;
			bits |= keep ? (1u << j) : 0u;
		}
		__builtin_amdgcn_raw_buffer_store_b8((uint8_t)bits, ro, threadIdx.x, 0, 0);
		count += (u32)__builtin_popcount(bits);
	}
	// one addition per workgroup: every one of them goes to the same address, about 10 ns each — per wave (8 192 of them) that was
	// 23 us of a 2 M-record call that streams in 5, and of the 481 us of 200 M records
	for (int off = 32; off > 0; off >>= 1) count += __shfl_down(count, off, 64);
	if ((threadIdx.x & 63) == 0 && count) atomicAdd(&wg_kept, count);
	__syncthreads();
	if (threadIdx.x == 0 && wg_kept) atomicAdd(kept, (unsigned long long)wg_kept);
}

// ---------------------------------------------------------------------------------------------------
// f2 (second half): `sam count` (src/sam_count.rs:44-127) — per record the filter chain and the fragment interval
// in the reference's u32 arithmetic, then one count for every region of the record's chromosome that the interval
// overlaps.  The reference walks a deque of the chromosome's regions sorted by start (:122-126: stop at the first
// region that starts at or after the fragment's end, skip regions that end at or before its start); here that is a
// binary search for the stop point and a backward walk that ends as soon as the running maximum of the region ends
// (rpmax) says nothing further left can reach the fragment.  The deque's pop_front (:116-119) only drops regions
// that can no longer be hit in a coordinate-sorted file, which the host checks record by record (:70-72).
// ---------------------------------------------------------------------------------------------------
// first index i of [b, e) with rstart[i] >= key, or e
__device__ __forceinline__ int region_lower_bound(const uint32_t *rstart, int b, int e, u32 key)
{
	while (b < e) {
		const int mid = (b + e) >> 1;
		if (rstart[mid] >= key) e = mid; else b = mid + 1;
	}
	return b;
}

// the same over [lo, hi), starting from a guess p in [lo, hi]: doubling steps away from p until the answer is bracketed,
// then a binary search inside the bracket — two probes when the answer is p or p + 1
__device__ __forceinline__ int region_gallop(const uint32_t *rstart, int lo, int hi, int p, u32 key)
{
	// (three doublings at most — the answer within eight regions of the guess —, then the rest of that side is searched
	// as a whole: records in no particular order cost a few probes more than a plain search, not twice as many)
	if (p < hi && rstart[p] < key) {                           // to the right of p: everything before l is below the key
		int l = p + 1, h = p + 1, step = 1;
		while (h < hi && step <= 4 && rstart[h] < key) { l = h + 1; h += step; step <<= 1; }
		if (step > 4 && h < hi && rstart[h] < key) { l = h + 1; h = hi; }
		return region_lower_bound(rstart, l, h < hi ? h : hi, key);
	}
	int h = p, l = p - 1, step = 1;                            // p or to its left: rstart[h] >= key (or h == hi)
	while (l >= lo && step <= 4 && rstart[l] >= key) { h = l; l -= step; step <<= 1; }
	if (step > 4 && l >= lo && rstart[l] >= key) { h = l; l = lo - 1; }
	return region_lower_bound(rstart, l + 1 > lo ? l + 1 : lo, h, key);
}

// Four CONSECUTIVE records per thread and iteration.  Their columns arrive as one wide load each (16 bytes of a 4-byte
// column, 8 of the flags, 4 of the mapping qualities) instead of four narrow ones — the record-per-thread kernel spent
// its time issuing 28 small loads per four records — and only the first of the four is searched for from scratch (see
// below).  VEC = the columns are 16-byte aligned.
constexpr int kCountIlp = 4;
template <bool VEC>
__global__ __launch_bounds__(256) void bam_count_kernel(const CountArgs a)
{
	const int64_t ngroups = (a.n + kCountIlp - 1) / kCountIlp;
	for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (int64_t)gridDim.x * blockDim.x) {
		const int64_t r0 = g * kCountIlp;
		const bool whole = r0 + kCountIlp <= a.n;
		u32 fl[kCountIlp], mq[kCountIlp];
		int32_t posv[kCountIlp], tidv[kCountIlp], mtidv[kCountIlp], mposv[kCountIlp], tlenv[kCountIlp], endv[kCountIlp];
		if (VEC && whole) {
			const u32x2 f2 = *reinterpret_cast<const u32x2 *>(a.flag + r0);
			const u32 m4 = *reinterpret_cast<const u32 *>(a.mapq + r0);
			const u32x4 p4 = *reinterpret_cast<const u32x4 *>(a.pos + r0), t4 = *reinterpret_cast<const u32x4 *>(a.tid + r0);
			u32x4 e4 = {0u, 0u, 0u, 0u}, mt4 = e4, mp4 = e4, tl4 = e4;
			if (a.single_end) e4 = *reinterpret_cast<const u32x4 *>(a.end_pos + r0);
			else {
				mt4 = *reinterpret_cast<const u32x4 *>(a.mtid + r0);
				mp4 = *reinterpret_cast<const u32x4 *>(a.mpos + r0);
				tl4 = *reinterpret_cast<const u32x4 *>(a.tlen + r0);
			}
#pragma unroll
			for (int u = 0; u < kCountIlp; u++) {
				fl[u] = (f2[u >> 1] >> (16 * (u & 1))) & 0xffffu;
				mq[u] = (m4 >> (8 * u)) & 0xffu;
				posv[u] = (int32_t)p4[u]; tidv[u] = (int32_t)t4[u];
				mtidv[u] = (int32_t)mt4[u]; mposv[u] = (int32_t)mp4[u]; tlenv[u] = (int32_t)tl4[u]; endv[u] = (int32_t)e4[u];
			}
		} else {
#pragma unroll
			for (int u = 0; u < kCountIlp; u++) {
				const int64_t rc = r0 + u < a.n ? r0 + u : r0;                     // past the end: the group's first record again, dropped below
				fl[u] = a.flag[rc]; mq[u] = a.mapq[rc]; posv[u] = a.pos[rc]; tidv[u] = a.tid[rc];
				endv[u] = a.single_end ? a.end_pos[rc] : 0;
				mtidv[u] = a.single_end ? 0 : a.mtid[rc]; mposv[u] = a.single_end ? 0 : a.mpos[rc]; tlenv[u] = a.single_end ? 0 : a.tlen[rc];
			}
		}
		u32 start[kCountIlp], end[kCountIlp];
		int lo[kCountIlp], b[kCountIlp], e[kCountIlp];
#pragma unroll
		for (int u = 0; u < kCountIlp; u++) {
			const u32 f = fl[u];
			bool ok = r0 + u < a.n;
			ok = ok && !(f & (0x4u | 0x400u | 0x100u | 0x800u));                // :46-48
			ok = ok && mq[u] >= a.min_mapq;                                     // :49
			const int32_t pos = posv[u], tid = tidv[u];
			u32 st = (u32)pos, en;                                              // :75
			if (a.single_end) {
				en = (u32)endv[u];                                              // :77
			} else {
				ok = ok && (f & 0x1u) && !(f & 0x8u);                           // :79-80
				ok = ok && tid == mtidv[u];                                     // :81
				const int32_t mpos = mposv[u];
				ok = ok && !(pos > mpos || (pos == mpos && !(f & 0x40u)));      // :91
				const int32_t tl = tlenv[u];
				const u32 ins = tl < 0 ? 0u - (u32)tl : (u32)tl;                // :93
				ok = ok && ins >= 20u;                                          // :94
				en = st + ins;                                                  // :96
			}
			ok = ok && en - st <= a.max_frag_len;                               // :99
			if (a.center) { st += (en - st) / 2u; en = st + 1u; }               // :103-107
			ok = ok && tid >= 0 && tid < a.n_chr;                               // the host has raised chr_names[tid] already
			const int tc = ok ? tid : 0;
			lo[u] = a.chr_off[tc];
			b[u] = lo[u];
			e[u] = ok ? a.chr_off[tc + 1] : lo[u];                              // a dropped record searches an empty range
			start[u] = st;
			end[u] = en;
		}
		// First region with rstart >= end.  The lane's first record is searched for in its chromosome's whole range; the
		// next ones start from their predecessor's answer and gallop: the host demands a coordinate-sorted file, where
		// consecutive records land on the same region or the next — two probes instead of fifteen (144 against 103 G
		// records/s with four plain searches in lockstep; on records in no order, which only a caller of the C-ABI can
		// hand in, 48 against 82: the answers are the same either way).
		int hi_prev = -1;
#pragma unroll
		for (int u = 0; u < kCountIlp; u++) {
			const int hi = e[u];
			if (u > 0 && lo[u] == lo[u - 1] && hi == hi_prev && hi > lo[u]) b[u] = region_gallop(a.rstart, lo[u], hi, b[u - 1], end[u]);
			else b[u] = region_lower_bound(a.rstart, lo[u], hi, end[u]);
			hi_prev = hi;
		}
#pragma unroll
		for (int u = 0; u < kCountIlp; u++) {
			for (int i = b[u] - 1; i >= lo[u]; i--) {
				if (a.rpmax[i] <= start[u]) break;
				if (a.rend[i] > start[u]) atomicAdd(&a.counts[a.ridx[i]], 1u);
			}
		}
	}
}

hipError_t launch_bam_count(const CountArgs &a, int n_cu, hipStream_t st)
{
	if (a.n <= 0) return hipSuccess;
	const int64_t want = (a.n + 256 * kCountIlp - 1) / (256 * kCountIlp);
	const int grid = (int)(want < (int64_t)n_cu * 8 ? want : (int64_t)n_cu * 8);
	uintptr_t bits = (uintptr_t)a.flag | (uintptr_t)a.mapq | (uintptr_t)a.pos | (uintptr_t)a.tid;
	bits |= a.single_end ? (uintptr_t)a.end_pos : ((uintptr_t)a.mtid | (uintptr_t)a.mpos | (uintptr_t)a.tlen);
	if ((bits & 15u) == 0) bam_count_kernel<true><<<grid, 256, 0, st>>>(a);
	else bam_count_kernel<false><<<grid, 256, 0, st>>>(a);
	return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// `fasta gc content` (src/fasta_gc_content.rs:41-46): per region of a genome held in HBM, gc = bytes that are C, G, c or
// g and total = bytes that are neither N nor n.  The host cuts regions into segments of at most 64 KiB; one wave
// counts one segment with 16-byte loads (aligned body, byte-wise head and tail) and adds its two sums to the region's
// counters.  1 byte read per base: HBM-bound.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 zero_bytes(u32 x)       // 0x80 in every byte of x that is 0
{
	return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}

__device__ __forceinline__ void gc_dword(u32 w, u32 &gc, u32 &non_n)
{
	const u32 up = w & 0xDFDFDFDFu;          // clear bit 5: only 'c'/'C', 'g'/'G', 'n'/'N' land on the three letters tested below
	gc += (u32)__popc(zero_bytes(up ^ 0x43434343u) | zero_bytes(up ^ 0x47474747u));
	non_n += 4u - (u32)__popc(zero_bytes(up ^ 0x4E4E4E4Eu));
}

__device__ __forceinline__ void gc_byte(u32 b, u32 &gc, u32 &non_n)
{
	const u32 up = b & 0xDFu;
	gc += (up == 0x43u || up == 0x47u) ? 1u : 0u;
	non_n += (up != 0x4Eu) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void gc_count_kernel(const uint8_t *__restrict__ genome, const int64_t *__restrict__ seg_start,
                                                       const int32_t *__restrict__ seg_len, const int32_t *__restrict__ seg_region, int64_t nseg,
                                                       unsigned long long *__restrict__ out /* [2 * regions]: gc, total */)
{
	const int lane = threadIdx.x & 63;
	const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
	for (int64_t s = wave; s < nseg; s += nwaves) {
		const uint8_t *p = genome + seg_start[s];
		const int n = seg_len[s];
		u32 gc = 0, non_n = 0;
		int head = (int)((16 - ((uintptr_t)p & 15)) & 15);
		if (head > n) head = n;
		if (lane < head) gc_byte(p[lane], gc, non_n);
		const int body = (n - head) & ~15;
		const u32x4 *v = reinterpret_cast<const u32x4 *>(p + head);
		const int nchunk = body >> 4;
		int i = lane;
		for (; i + 192 < nchunk; i += 256) {                                  // four 1 KiB rows of the wave in flight
			u32x4 w[4];
#pragma unroll
			for (int k = 0; k < 4; k++) w[k] = stream_load(reinterpret_cast<const uint8_t *>(v + i + 64 * k));
#pragma unroll
			for (int k = 0; k < 4; k++) { gc_dword(w[k][0], gc, non_n); gc_dword(w[k][1], gc, non_n); gc_dword(w[k][2], gc, non_n); gc_dword(w[k][3], gc, non_n); }
		}
		for (; i < nchunk; i += 64) {
			const u32x4 w = stream_load(reinterpret_cast<const uint8_t *>(v + i));
			gc_dword(w[0], gc, non_n); gc_dword(w[1], gc, non_n); gc_dword(w[2], gc, non_n); gc_dword(w[3], gc, non_n);
		}
		const int tail0 = head + body;
		if (tail0 + lane < n) gc_byte(p[tail0 + lane], gc, non_n);
		for (int o = 32; o > 0; o >>= 1) { gc += __shfl_xor(gc, o); non_n += __shfl_xor(non_n, o); }
		if (lane == 0) {
			const int r = seg_region[s];
			if (gc) atomicAdd(&out[2 * r], (unsigned long long)gc);
			if (non_n) atomicAdd(&out[2 * r + 1], (unsigned long long)non_n);
		}
	}
}

hipError_t launch_gc_count(const uint8_t *genome, const int64_t *seg_start, const int32_t *seg_len, const int32_t *seg_region, int64_t nseg,
                           unsigned long long *out, int n_cu, hipStream_t st)
{
	if (nseg <= 0) return hipSuccess;
	const int64_t want = (nseg + 3) / 4;
	const int grid = (int)(want < (int64_t)n_cu * 8 ? want : (int64_t)n_cu * 8);
	gc_count_kernel<<<grid, 256, 0, st>>>(genome, seg_start, seg_len, seg_region, nseg, out);
	return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// f4: `sam to fastq` sequence() (src/sam_to_fastq.rs:31-59) — BAM 4-bit bases -> ASCII, reverse-complemented for
// reverse-strand records, 'N' where the quality is below min_baseq.  One thread per 16 output bytes; a workgroup
// walks tiles of 64 rows, whose lengths and strands it keeps in LDS.
// ---------------------------------------------------------------------------------------------------
constexpr u32 kN4 = 0x4E4E4E4Eu;      // "NNNN"

// four 4-bit codes, one per byte -> ASCII: codes 1,2,4,8 -> A C G T (or their complements T G C A), anything else 'N'
__device__ __forceinline__ u32 bases_from_codes(u32 nib, bool complement)
{
	// 8-entry byte tables indexed by the low 3 bits of the code; codes 8..15 use the second pair
	const u32 lo = complement ? 0x4E47544Eu : 0x4E43414Eu;        // codes 0..3: N c1 c2 N    (c1 c2 = "AC" / "TG")
	const u32 hi = complement ? 0x4E4E4E43u : 0x4E4E4E47u;        // codes 4..7: c4 N N N     (c4 = 'G' / 'C')
	const u32 lo8 = complement ? 0x4E4E4E41u : 0x4E4E4E54u;       // codes 8..11: c8 N N N    (c8 = 'T' / 'A')
	const u32 sel = nib & 0x07070707u;
	const u32 r_low = __builtin_amdgcn_perm(hi, lo, sel);
	const u32 r_high = __builtin_amdgcn_perm(kN4, lo8, sel);
	const u32 is_high = ((nib >> 3) & 0x01010101u) * 0xFFu;
	return (r_high & is_high) | (r_low & ~is_high);
}

// 0xFF in every byte where q < m (unsigned), else 0x00; SMALL_M: m < 128, so a byte with its top bit set is never below
template <bool SMALL_M>
__device__ __forceinline__ u32 bytes_below(u32 q, u32 m4)
{
	const u32 d = (q | kHi1) - (m4 & ~kHi1);                       // per byte, no borrow: bit 7 = (q & 0x7f) >= (m & 0x7f)
	const u32 lt = SMALL_M ? (~(q | d) & kHi1) : (((~q & m4) | (~(q ^ m4) & ~d)) & kHi1);
	return (lt >> 7) * 0xFFu;
}

// 16 bits, four codes with the first in the top nibble -> one code per byte, the FIRST code in byte 3
__device__ __forceinline__ u32 spread_nibbles(u32 h)
{
	u32 x = (h | (h << 8)) & 0x00FF00FFu;
	return (x | (x << 4)) & 0x0F0F0F0Fu;
}

// Both strands run the same instructions.  EIGHT output bytes per thread: output bytes 8j..8j+7 of a row come from source
// positions s0..s0+7 with s0 = 8j read forwards (stored order, :47-57) or s0 = len-8-8j read backwards (reverse
// complement, :36-46).  The qualities and the base codes from s0 on are fetched as aligned dwords through per-tile buffer
// descriptors — two- and one-dword loads of the qualities (three dwords cover any 8 bytes), one two-dword load of the
// packed bases (8 bytes from the dword that holds the first code cover any 8 codes) — and funnel-shifted into place; a
// byte permute whose selector depends on the strand puts them in output order; one two-dword store.  Where s0 < 0 (the
// last, partial unit of a reverse row) or s0 + 7 >= len, the bytes that fall outside the read land in output positions
// >= len, which are unspecified; a dword that would start before the row is replaced by the row's first one (the offset
// register must not go negative: the hardware range check adds the instruction's immediate to it without wrapping), and
// what lies past the tile reads as 0.  A row pitch that is an odd multiple of 4 has a last unit of four bytes: the same
// instructions, and only its first dword is stored (the second would be the next row's first).
// (Four bytes per thread was the first form: ten memory instructions per 8 bytes instead of four, 52-57 % of the HBM peak
// against 60-66 %.  Sixteen bytes per thread was measured and dropped: fewer VALU per byte, but its 16-byte lane pitch
// quarters the coalescing of the loads and stores.)
#ifndef SK_SEQ_UNROLL
#define SK_SEQ_UNROLL 1                  // elements per thread and iteration: 1 / 2 / 4 -> 60.4 / 57.5 / 50.4 % of the HBM peak (tools/seq_ab.py)
#endif
constexpr int kSeqUnroll = SK_SEQ_UNROLL;
template <bool SMALL_M>
__global__ __launch_bounds__(256) void bam_sequence8_kernel(const uint8_t *__restrict__ seq4, int seq4_stride, const uint8_t *__restrict__ qual,
                                                            int stride, const uint16_t *__restrict__ len, const uint16_t *__restrict__ flag,
                                                            int64_t n, u32 m4, u32 inv_upr, uint8_t *__restrict__ out)
{
	__shared__ u32 row_info[64];                                   // len | reverse << 16
	const int upr = (stride + 7) >> 3;                             // 8-byte units per row; the last one is half a unit when stride % 8 == 4
	const bool half_tail = (stride & 7) != 0;
	const int64_t ntiles = (n + 63) / 64;
	for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
		const int64_t row0 = t * 64;
		const int rows = (int)((n - row0) < 64 ? (n - row0) : 64);
		__syncthreads();
		if ((int)threadIdx.x < rows) {
			const int64_t r = row0 + threadIdx.x;
			const u32 l = len ? (u32)len[r] : (u32)stride;
			row_info[threadIdx.x] = l | (((u32)flag[r] >> 4) & 1u) << 16;
		}
		__syncthreads();
		const rsrc_t rq = make_rsrc(qual, row0 * (int64_t)stride, rows * stride);
		const rsrc_t rs = make_rsrc(seq4, row0 * (int64_t)seq4_stride, rows * seq4_stride);
		const rsrc_t ro = make_rsrc(out, row0 * (int64_t)stride, rows * stride);
		const u32 total = (u32)rows * (u32)upr;
		// kSeqUnroll elements per thread and iteration, all their loads issued before the first is used
		for (u32 e = threadIdx.x; e < total; e += blockDim.x * kSeqUnroll) {
			u32 rlv[kSeqUnroll], w0v[kSeqUnroll];
			int jv[kSeqUnroll], s0v[kSeqUnroll];
			bool revv[kSeqUnroll];
			u32x2 w12v[kSeqUnroll], ddv[kSeqUnroll];
#pragma unroll
			for (int u = 0; u < kSeqUnroll; u++) {
				const u32 eu = e + (u32)u * blockDim.x;
				const u32 ec = eu < total ? eu : total - 1;           // past the tile: recompute the last element, store nothing
				const u32 rl = inv_upr ? __umulhi(ec, inv_upr) : ec / (u32)upr;
				const int j = (int)(ec - rl * (u32)upr);
				const u32 info = row_info[rl];
				const int l = (int)(info & 0xFFFFu);
				const bool rev = (info >> 16) != 0u;
				int s0 = rev ? l - 8 - 8 * j : 8 * j;
				if (s0 < -8) s0 = -8;                                  // every byte of this unit is past the read already
				// qualities s0..s0+7 in source order: three dwords, funnel-shifted below (a dword that would start before the row
				// is replaced by the row's first one: what it contributes lies before the read's first base)
				const int qrow = (int)rl * stride, qd = s0 >> 2;
				w0v[u] = __builtin_amdgcn_raw_buffer_load_b32(rq, qrow + 4 * (qd < 0 ? 0 : qd), 0, 0);
				w12v[u] = __builtin_amdgcn_raw_buffer_load_b64(rq, qrow + 4 * (qd + 1 < 0 ? 0 : qd + 1), 0, 0);
				// base codes s0..s0+7: the 8 bytes from the dword that holds byte a = s0 >> 1 cover bytes a..a+4
				const int sd = (s0 >> 1) >> 2;
				ddv[u] = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)rl * seq4_stride + 4 * (sd < 0 ? 0 : sd), 0, 0);
				rlv[u] = rl; jv[u] = j; s0v[u] = s0; revv[u] = rev;
			}
#pragma unroll
			for (int u = 0; u < kSeqUnroll; u++) {
				const int s0 = s0v[u];
				const bool rev = revv[u];
				const int qd = s0 >> 2;
				const u32 sh = (u32)s0 & 3u;
				const u32 w0 = w0v[u], w1 = w12v[u][0], w2 = qd + 1 < 0 ? w12v[u][0] : w12v[u][1];      // qd = -2: the third dword is the row's first
				const u32 qa = __builtin_amdgcn_alignbyte(w1, w0, sh), qb = __builtin_amdgcn_alignbyte(w2, w1, sh);
				const int a = s0 >> 1, sd = a >> 2;
				const u32 d0 = ddv[u][0], d1 = sd < 0 ? ddv[u][0] : ddv[u][1];                           // sd = -1: the second dword is the row's first
				const u32 b03 = __builtin_amdgcn_alignbyte(d1, d0, (u32)a & 3u);          // bytes a..a+3
				const u32 b4 = __builtin_amdgcn_alignbyte(0u, d1, (u32)a & 3u) & 0xFFu;   // byte a+4
				const u32 be = __builtin_bswap32(b03);                                    // codes 2a..2a+7, the first in the top nibble
				const u32 n8 = (s0 & 1) ? ((be << 4) | (b4 >> 4)) : be;                   // codes s0..s0+7, likewise
				const u32 c_first = spread_nibbles(n8 >> 16), c_last = spread_nibbles(n8 & 0xFFFFu);      // byte b = code at position +3-b of its four
				u32 q_lo, q_hi, n_lo, n_hi;
				if (rev) {          // output byte k = source position s0+7-k
					q_lo = __builtin_amdgcn_perm(0u, qb, 0x00010203u); q_hi = __builtin_amdgcn_perm(0u, qa, 0x00010203u);
					n_lo = c_last; n_hi = c_first;
				} else {
					q_lo = qa; q_hi = qb;
					n_lo = __builtin_amdgcn_perm(0u, c_first, 0x00010203u); n_hi = __builtin_amdgcn_perm(0u, c_last, 0x00010203u);
				}
				const u32 lo_low = bytes_below<SMALL_M>(q_lo, m4), hi_low = bytes_below<SMALL_M>(q_hi, m4);
				u32x2 o;
				o[0] = (kN4 & lo_low) | (bases_from_codes(n_lo, rev) & ~lo_low);
				o[1] = (kN4 & hi_low) | (bases_from_codes(n_hi, rev) & ~hi_low);
				if (e + (u32)u * blockDim.x < total) {
					if (half_tail && jv[u] == upr - 1) __builtin_amdgcn_raw_buffer_store_b32(o[0], ro, (int)rlv[u] * stride + 8 * jv[u], 0, 0);
					else __builtin_amdgcn_raw_buffer_store_b64(o, ro, (int)rlv[u] * stride + 8 * jv[u], 0, 0);
				}
			}
		}
	}
}

// The same decoding for rows of a pitch that is a multiple of 4 (8 through round 5; 148 = 2 x 74 bases is the common case of the other
// residue: a row then ends in half a unit) and at most NQ KiB / 64 (NQ = 10: 150-base reads at pitch 152 or 160) — through an LDS
// image of the tile.  A WAVE owns a tile of 64 rows: its qualities and packed bases arrive as NQ + NS
// fully coalesced 16-byte loads per lane (1 KiB per wave instruction, nt: read once) and are written to the wave's LDS slot;
// the NEXT tile's loads are issued right behind (the whole next tile is in flight while this one is worked on: 15 KiB per
// wave, 120 KiB per CU); lane l then computes units l, l + 64, ... of the tile (a unit = 8 output bytes, exactly
// bam_sequence8_kernel's arithmetic with its dwords read from LDS — two lanes per bank), keeps them in registers until every
// lane has read what it needs, writes them over the quality image, and the tile leaves as NQ coalesced 16-byte stores per
// lane.  No workgroup barrier: a wave's LDS operations execute in order.  (bam_sequence8_kernel asks memory for 4-, 8- and
// 8-byte pieces per lane and stores 8: 60-65 % of the HBM peak whatever was tried on it — EXPERIMENTS.md A.7; it stays for
// every other pitch.)
#ifndef SK_SEQ_TWO
#define SK_SEQ_TWO 0
#endif
template <bool SMALL_M, int NQ, int ROWS>
__global__ __launch_bounds__(256, ROWS == 64 ? 2 : 4) void bam_sequence_tile_kernel(const uint8_t *__restrict__ seq4, int seq4_stride, const uint8_t *__restrict__ qual,
                                                                   int stride, const uint16_t *__restrict__ len, const uint16_t *__restrict__ flag,
                                                                   int64_t n, u32 m4, u32 inv_upr, uint8_t *__restrict__ out)
{
	constexpr int NS = (NQ + 1) / 2;
	constexpr int kSlot = (NQ + NS) * 1024 + 256;
	constexpr int kMaxU = NQ * 2;                                  // units per lane at most: ROWS x (pitch / 8) units / 64 lanes
	static_assert(ROWS == 64 || ROWS == 32, "a tile is one or half a row per lane");
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	uint8_t *img = sk_smem + wave * kSlot;                         // the qualities (then the output) ...
	uint8_t *simg = img + NQ * 1024;                               // ... the packed bases ...
	u32 *info = reinterpret_cast<u32 *>(img + (NQ + NS) * 1024);   // ... len | reverse << 16 per row
	const int upr = (stride + 7) >> 3;                             // units per row; a pitch that is 4 mod 8 (148: 2 x 74 bases) ends in half a unit
	const bool half_tail = (stride & 4) != 0;
	const int64_t ntiles = (n + ROWS - 1) / ROWS;
	const int64_t tstep = (int64_t)gridDim.x * 4;
	auto fetch = [&](int64_t t, u32x4 (&rq)[NQ], u32x4 (&rs)[NS], u32 &rinfo) {
		const int rows = t < ntiles ? (int)((n - t * ROWS) < ROWS ? (n - t * ROWS) : ROWS) : 0;
		const int64_t row0 = rows ? t * ROWS : 0;
		const rsrc_t dq = make_rsrc(qual, row0 * (int64_t)stride, rows * stride), ds = make_rsrc(seq4, row0 * (int64_t)seq4_stride, (rows * seq4_stride + 3) & ~3);
#pragma unroll
		for (int c = 0; c < NQ; c++) rq[c] = __builtin_amdgcn_raw_buffer_load_b128(dq, c * 1024 + lane * 16, 0, kAuxStream);
#pragma unroll
		for (int c = 0; c < NS; c++) rs[c] = __builtin_amdgcn_raw_buffer_load_b128(ds, c * 1024 + lane * 16, 0, kAuxStream);
		const u32 lv = (u32)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(make_rsrc(len, row0 * 2, rows * 2), lane * 2, 0, 0);
		const u32 fv = (u32)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(make_rsrc(flag, row0 * 2, rows * 2), lane * 2, 0, 0);
		rinfo = (len ? lv : (u32)stride) | ((fv >> 4) & 1u) << 16;
	};
	// one tile: its registers into the wave's LDS slot, the tile `ahead` steps on asked for into the same registers, then the work
	auto process = [&](int64_t t, int ahead, u32x4 (&rq)[NQ], u32x4 (&rs)[NS], u32 &rinfo) {
		const int64_t row0 = t * ROWS;
		const int rows = (int)((n - row0) < ROWS ? (n - row0) : ROWS);
		const u32 units = (u32)(ROWS * upr);
#pragma unroll
		for (int c = 0; c < NQ; c++) *reinterpret_cast<u32x4 *>(img + c * 1024 + lane * 16) = rq[c];
#pragma unroll
		for (int c = 0; c < NS; c++) *reinterpret_cast<u32x4 *>(simg + c * 1024 + lane * 16) = rs[c];
		info[lane] = rinfo;
		wave_lds_fence();
		fetch(t + ahead * tstep, rq, rs, rinfo);                       // a later tile: on its way while this one is worked on
		u32x2 ov[kMaxU];
#pragma unroll
		for (int i = 0; i < kMaxU; i++) {
			ov[i][0] = ov[i][1] = 0u;
			if (64u * (u32)i < units) {                                // (uniform)
				const u32 ee = (u32)lane + 64u * (u32)i;
				const u32 e = ee < units ? ee : units - 1u;               // (half a wave in the last turn of a 32-row tile: recompute the last unit)
				const u32 rl = inv_upr ? __umulhi(e, inv_upr) : e / (u32)upr;
				const int j = (int)(e - rl * (u32)upr);
				const u32 inf = info[rl];
				const int l = (int)(inf & 0xFFFFu);
				const bool rev = (inf >> 16) != 0u;
				int s0 = rev ? l - 8 - 8 * j : 8 * j;
				if (s0 < -8) s0 = -8;                                  // every byte of this unit is past the read already
				// qualities s0 .. s0 + 7 in source order: three dwords of the row's image, funnel-shifted (a dword that would start
				// before the row is replaced by the row's first one: what it contributes lies before the read's first base)
				const int qrow = (int)rl * stride, qd = s0 >> 2;
				const u32 *qw = reinterpret_cast<const u32 *>(img + qrow);
				const u32 w0 = qw[qd < 0 ? 0 : qd], w1 = qw[qd + 1 < 0 ? 0 : qd + 1], w2 = qw[qd + 2 < 0 ? 0 : qd + 2];
				const u32 sh = (u32)s0 & 3u;
				const u32 qa = __builtin_amdgcn_alignbyte(w1, w0, sh), qb = __builtin_amdgcn_alignbyte(w2, w1, sh);
				// base codes s0 .. s0 + 7: the 8 bytes from the dword that holds byte a = s0 >> 1 cover bytes a .. a + 4
				const int a = s0 >> 1, sd = a >> 2;
				const u32 *sw = reinterpret_cast<const u32 *>(simg + (int)rl * seq4_stride);
				const u32 d0 = sw[sd < 0 ? 0 : sd], d1 = sw[sd + 1 < 0 ? 0 : sd + 1];
				const u32 b03 = __builtin_amdgcn_alignbyte(d1, d0, (u32)a & 3u);          // bytes a .. a + 3
				const u32 b4 = __builtin_amdgcn_alignbyte(0u, d1, (u32)a & 3u) & 0xFFu;   // byte a + 4
				const u32 be = __builtin_bswap32(b03);                                    // codes 2a .. 2a + 7, the first in the top nibble
				const u32 n8 = (s0 & 1) ? ((be << 4) | (b4 >> 4)) : be;                   // codes s0 .. s0 + 7, likewise
				const u32 c_first = spread_nibbles(n8 >> 16), c_last = spread_nibbles(n8 & 0xFFFFu);
				u32 q_lo, q_hi, n_lo, n_hi;
				if (rev) {          // output byte k = source position s0 + 7 - k
					q_lo = __builtin_amdgcn_perm(0u, qb, 0x00010203u); q_hi = __builtin_amdgcn_perm(0u, qa, 0x00010203u);
					n_lo = c_last; n_hi = c_first;
				} else {
					q_lo = qa; q_hi = qb;
					n_lo = __builtin_amdgcn_perm(0u, c_first, 0x00010203u); n_hi = __builtin_amdgcn_perm(0u, c_last, 0x00010203u);
				}
				const u32 lo_low = bytes_below<SMALL_M>(q_lo, m4), hi_low = bytes_below<SMALL_M>(q_hi, m4);
				ov[i][0] = (kN4 & lo_low) | (bases_from_codes(n_lo, rev) & ~lo_low);
				ov[i][1] = (kN4 & hi_low) | (bases_from_codes(n_hi, rev) & ~hi_low);
			}
		}
		wave_lds_fence();                                              // every lane has read what it needs: the units go over the quality image
		if (!half_tail) {
#pragma unroll
			for (int i = 0; i < kMaxU; i++)
				if ((u32)lane + 64u * (u32)i < units) *reinterpret_cast<u32x2 *>(img + 8 * (lane + 64 * i)) = ov[i];
		} else {
			// unit j of row rl lies at rl * stride + 8 j (4-byte aligned); the row's last unit is its first four bytes only
#pragma unroll
			for (int i = 0; i < kMaxU; i++) {
				const u32 e = (u32)lane + 64u * (u32)i;
				if (e < units) {
					const u32 rl = inv_upr ? __umulhi(e, inv_upr) : e / (u32)upr;
					const int j = (int)(e - rl * (u32)upr);
					u32 *dst = reinterpret_cast<u32 *>(img + (int)rl * stride + 8 * j);
					dst[0] = ov[i][0];
					if (j != upr - 1) dst[1] = ov[i][1];
				}
			}
		}
		wave_lds_fence();
		const rsrc_t dout = make_rsrc(out, row0 * (int64_t)stride, rows * stride);
#pragma unroll
		for (int c = 0; c < NQ; c++) {
			const u32x4 v = *reinterpret_cast<const u32x4 *>(img + c * 1024 + lane * 16);
			__builtin_amdgcn_raw_buffer_store_b128(v, dout, c * 1024 + lane * 16, 0, kAuxStreamSt);
		}
		wave_lds_fence();
	};
	int64_t t = (int64_t)blockIdx.x * 4 + wave;
#if SK_SEQ_TWO
	// TWO register sets: while a tile is worked on, the next one AND the one behind it are on their way (30 KiB per wave)
	u32x4 rqa[NQ], rsa[NS], rqb[NQ], rsb[NS];
	u32 ia = 0u, ib = 0u;
	fetch(t, rqa, rsa, ia);
	fetch(t + tstep, rqb, rsb, ib);
	for (; t < ntiles; t += 2 * tstep) {
		process(t, 2, rqa, rsa, ia);
		if (t + tstep < ntiles) process(t + tstep, 2, rqb, rsb, ib);
	}
#else
	u32x4 rq[NQ], rs[NS];
	u32 rinfo = 0u;
	fetch(t, rq, rs, rinfo);
	for (; t < ntiles; t += tstep) process(t, 1, rq, rs, rinfo);
#endif
}

hipError_t launch_bam_sequence(const uint8_t *seq4, int seq4_stride, const uint8_t *qual, int stride, const uint16_t *len, const uint16_t *flag,
                               int64_t n, int min_baseq, uint8_t *out, int n_cu, hipStream_t st)
{
	if (n <= 0) return hipSuccess;
	const int64_t ntiles = (n + 63) / 64;
	static const int wgs = getenv("SK_SEQ_WGS") ? atoi(getenv("SK_SEQ_WGS")) : 8;
	const int grid = (int)(ntiles < (int64_t)n_cu * wgs ? ntiles : (int64_t)n_cu * wgs);
	const u32 m4 = (u32)(min_baseq & 0xFF) * 0x01010101u;
	// rl = e / upr as a multiply-high: exact while e * upr < 2^32, and e < 64 * upr
	const u32 upr = ((u32)stride + 7u) >> 3;
	const u32 inv8 = (upr > 1 && upr < 8192u) ? (u32)(((1ull << 32) + upr - 1) / upr) : 0u;
	// pitches the LDS-tile kernel serves: a multiple of 4, a tile's qualities in ten 1 KiB chunks and its packed bases in five
	// (150-base reads at pitch 152 / 160); matrices 16-byte aligned (SK_SEQ_TILE=0: the other kernel, for A/B and the tests)
	const char *env_tile = getenv("SK_SEQ_TILE");                      // (read per launch: the tests run both kernels in one process)
	const bool env_no_tile = env_tile && atoi(env_tile) == 0;
	if (!env_no_tile && (stride & 3) == 0 && stride >= 8 && 64 * stride <= 10 * 1024 && (seq4_stride & 3) == 0 && 64 * seq4_stride <= 5 * 1024 &&
	    ((uintptr_t)seq4 & 15) == 0 && ((uintptr_t)qual & 15) == 0 && ((uintptr_t)out & 15) == 0) {
		// (tiles of 32 rows with twice the waves were measured too: 62.1 % of the HBM peak against 63.2 % on the box where the
		// 8-byte kernel did 60.0 %; one workgroup per CU 53.8 %)
		const bool small = (min_baseq & 0xFF) < 128;
		constexpr int kNQ = 10;
		const int lds = 4 * ((kNQ + (kNQ + 1) / 2) * 1024 + 256);
		const void *fn = small ? reinterpret_cast<const void *>(bam_sequence_tile_kernel<true, kNQ, 64>) : reinterpret_cast<const void *>(bam_sequence_tile_kernel<false, kNQ, 64>);
		hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
		if (e != hipSuccess) return e;
		const int64_t want = (ntiles + 3) / 4;
		const int g2 = (int)(want < (int64_t)n_cu * 2 ? want : (int64_t)n_cu * 2);
		void *kargs[] = {(void *)&seq4, (void *)&seq4_stride, (void *)&qual, (void *)&stride, (void *)&len, (void *)&flag, (void *)&n, (void *)&m4, (void *)&inv8, (void *)&out};
		return hipLaunchKernel(fn, dim3((unsigned)g2), dim3(256), kargs, (size_t)lds, st);
	}
	if ((min_baseq & 0xFF) < 128)
		bam_sequence8_kernel<true><<<grid, 256, 0, st>>>(seq4, seq4_stride, qual, stride, len, flag, n, m4, inv8, out);
	else
		bam_sequence8_kernel<false><<<grid, 256, 0, st>>>(seq4, seq4_stride, qual, stride, len, flag, n, m4, inv8, out);
	return hipGetLastError();
}

hipError_t launch_bam_fragments(const uint16_t *flag, const int32_t *tid, const int32_t *mtid, const int32_t *tlen, int64_t n,
                                int64_t min_size, int64_t max_size, uint8_t *keep_bits, unsigned long long *kept, int n_cu, hipStream_t st)
{
	if (n <= 0) return hipSuccess;
	// |tlen| <= 2^31: clamp the i64 window of the reference into that range (an empty window keeps nothing)
	const int64_t lo = min_size < 0 ? 0 : min_size, hi = max_size > 0x80000000ll ? 0x80000000ll : max_size;
	const u32 ulo = lo > 0x80000000ll ? 0xffffffffu : (u32)lo, uhi = hi < 0 ? 0u : (u32)hi;
	const bool empty = hi < 0 || lo > hi;
	int64_t want = (n + 2047) / 2048;
	static const int wgs = getenv("SK_BAM_WGS") ? atoi(getenv("SK_BAM_WGS")) : 2;      // 1/2/3/4/8 per CU: 48 (flag+tlen) / 76 / 75 / 71 / 69 % of HBM peak
	int grid = (int)(want < (int64_t)n_cu * wgs ? want : (int64_t)n_cu * wgs);
	bam_fragments_kernel<<<grid, 256, 0, st>>>(flag, tid, mtid, tlen, n, empty ? 1u : ulo, empty ? 0u : uhi, keep_bits, kept);
	return hipGetLastError();
}

hipError_t launch_bam_flag_tlen(const uint16_t *flag, const int32_t *tid, const int32_t *mtid, const int32_t *tlen,
                                int64_t n, int32_t max_frag, unsigned long long *out, int want_counters, int want_hist,
                                int n_cu, hipStream_t st)
{
	if (n <= 0) return hipSuccess;
	int64_t bins = want_hist ? (int64_t)max_frag + 1 : 0;
	int lds_bins = (int)(bins < 15360 ? bins : 15360);     // 60 KiB of LDS at most
	const bool aligned = (((uintptr_t)flag | (want_hist ? ((uintptr_t)tid | (uintptr_t)mtid | (uintptr_t)tlen) : 0)) & 15u) == 0;
	if (aligned) {
		int64_t want = (n + 2047) / 2048;
		static const int wgs = getenv("SK_BAM_WGS") ? atoi(getenv("SK_BAM_WGS")) : 2;      // 1/2/3/4/8 per CU: 48 (flag+tlen) / 76 / 75 / 71 / 69 % of HBM peak
		int grid = (int)(want < (int64_t)n_cu * wgs ? want : (int64_t)n_cu * wgs);
		bam_flag_tlen_kernel<<<grid, 256, lds_bins * 4, st>>>(flag, tid, mtid, tlen, n, max_frag, out, want_counters, want_hist, lds_bins);
	} else {
		int64_t want = (n + 255) / 256;
		int grid = (int)(want < (int64_t)n_cu * 4 ? want : (int64_t)n_cu * 4);
		bam_flag_tlen_scalar_kernel<<<grid, 256, lds_bins * 4, st>>>(flag, tid, mtid, tlen, n, max_frag, out, want_counters, want_hist, lds_bins);
	}
	return hipGetLastError();
}

}  // namespace sk
