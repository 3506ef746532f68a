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
//  * workgroup = one wavefront, so the tile hand-off needs no cross-wave barrier; up to 16 tiles are
//    resident per CU (9.6 KiB of LDS each at 150 bp) and interleave their stream / scan phases.
//  * barcode matching is lane-per-read against a one-hot re-coding of the sheet: matches =
//    popcount(obs & cand) over W dwords, candidates fetched through the scalar cache (wave-uniform).
//  * everything is integer/byte work bounded by HBM; there is no MFMA in this file by design.
//
// Reference semantics restated per function: see the citations (paths relative to the reference tree).
#include "sk_internal.h"

namespace sk {

typedef uint32_t u32;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

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
	else if (33 + m == 256) { q.mode = 2; t2 = 1; }           // mask iff q >= 33
	else { q.mode = 3; t2 = 33 + m - 256; }                   // mask iff q >= 33 or q < t2 (wrapped interval)
	int c2 = 256 - t2;
	q.cl2 = (u32)(c2 & 0x7f) * 0x01010101u;
	q.c72 = (c2 & 0x80) ? kHi1 : 0u;
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
template <int MODE>
__device__ __forceinline__ u32 lowq_flags(u32 q, u32 ql, u32 t1, u32 cl2, u32 c72)
{
	if (MODE == 0) return 0u;
	u32 g1 = q | t1;
	if (MODE == 2) return g1;
	u32 t2 = ql + cl2;
	u32 g2 = (q & c72) | ((q | c72) & t2);
	if (MODE == 1) return g1 & ~g2;
	return g1 | ~g2;
}

// out byte i = flagged ? 'N' : s byte i, via one v_perm_b32: selector i picks s, 4+i picks 'N'.
__device__ __forceinline__ u32 mask_select(u32 s, u32 F)
{
	u32 sel = ((F >> 5) & 0x04040404u) | 0x03020100u;
	return __builtin_amdgcn_perm(0x4e4e4e4eu, s, sel);
}

__device__ __forceinline__ u32 sub33(u32 q, u32 t1) { return t1 ^ (~q & kHi1); }

template <int MODE>
__device__ __forceinline__ void mask_dword4(const u32x4 &q, const u32x4 &s, u32 cl2, u32 c72, u32x4 &out, u32x4 &vq)
{
#pragma unroll
	for (int i = 0; i < 4; i++) {
		u32 ql = q[i] & kLo7;
		u32 t1 = ql + 0x5f5f5f5fu;
		vq[i] = sub33(q[i], t1);
		out[i] = mask_select(s[i], lowq_flags<MODE>(q[i], ql, t1, cl2, c72));
	}
}

__device__ __forceinline__ void mask_dword4_rt(int mode, const u32x4 &q, const u32x4 &s, u32 cl2, u32 c72, u32x4 &out, u32x4 &vq)
{
	switch (mode) {   // wave-uniform
	case 0: mask_dword4<0>(q, s, cl2, c72, out, vq); break;
	case 1: mask_dword4<1>(q, s, cl2, c72, out, vq); break;
	case 2: mask_dword4<2>(q, s, cl2, c72, out, vq); break;
	default: mask_dword4<3>(q, s, cl2, c72, out, vq); break;
	}
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
template <int MODE>
__global__ __launch_bounds__(256) void mask_flat_kernel(const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
                                                        uint8_t *__restrict__ out, int64_t bytes, u32 cl2, u32 c72)
{
	const int64_t nchunk = (bytes + 15) >> 4;
	const int64_t step = (int64_t)gridDim.x * blockDim.x;
	for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunk; c += step) {
		const int64_t off = c << 4;
		u32x4 q, s, o, vq;
		if (off + 16 <= bytes) {
			q = *reinterpret_cast<const u32x4 *>(qual + off);
			s = *reinterpret_cast<const u32x4 *>(seq + off);
			mask_dword4<MODE>(q, s, cl2, c72, o, vq);
			*reinterpret_cast<u32x4 *>(out + off) = o;
		} else {
			int valid = (int)(bytes - off);
			q = load_tail(qual + off, valid);
			s = load_tail(seq + off, valid);
			mask_dword4<MODE>(q, s, cl2, c72, o, vq);
			store_tail(out + off, o, valid);
		}
	}
}

hipError_t launch_mask_flat(const uint8_t *seq, const uint8_t *qual, uint8_t *out, int64_t bytes,
                            const QualConsts &qc, int n_cu, hipStream_t st)
{
	if (bytes <= 0) return hipSuccess;
	int64_t nchunk = (bytes + 15) >> 4;
	int64_t want = (nchunk + 255) / 256;
	int grid = (int)(want < (int64_t)n_cu * 8 ? want : (int64_t)n_cu * 8);
	switch (qc.mode) {
	case 0: mask_flat_kernel<0><<<grid, 256, 0, st>>>(seq, qual, out, bytes, qc.cl2, qc.c72); break;
	case 1: mask_flat_kernel<1><<<grid, 256, 0, st>>>(seq, qual, out, bytes, qc.cl2, qc.c72); break;
	case 2: mask_flat_kernel<2><<<grid, 256, 0, st>>>(seq, qual, out, bytes, qc.cl2, qc.c72); break;
	default: mask_flat_kernel<3><<<grid, 256, 0, st>>>(seq, qual, out, bytes, qc.cl2, qc.c72); break;
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
__device__ __forceinline__ int trim_scan_packed(const uint8_t *tile, int row_start, int len, int maxlen, int m, bool active)
{
	const int end = row_start + len;
	const u32 sh = (u32)end & 3u;
	int a = end & ~3;
	u32 hi = *reinterpret_cast<const u32 *>(tile + a);
	int T = 0, best = 0;
	bool alive = active;
	const int ndw = (maxlen + 3) >> 2;
	for (int jj = 0; jj < ndw; jj++) {
		a = max(a - 4, -4);                                   // rows shorter than the scan stay inside the front pad
		u32 lo = *reinterpret_cast<const u32 *>(tile + a);
		u32 d = __builtin_amdgcn_alignbyte(hi, lo, sh);       // bytes [end-4(jj+1), end-4jj) of the image
		hi = lo;
#pragma unroll
		for (int i = 3; i >= 0; i--) {
			const int j = 4 * jj + (4 - i);                   // bytes consumed so far, wave-uniform
			const int jm = j * m;                             // scalar
			T += (int)((d >> (8 * i)) & 0xffu);
			alive = alive && (j <= len) && (T <= 50 + jm);
			int K = T * (1 << kKeyBits) + (j - jm * (1 << kKeyBits));
			best = alive ? min(best, K) : best;
		}
		if (__ballot(alive) == 0ull) break;
	}
	return len - (best & ((1 << kKeyBits) - 1));
}

// same scan with separate (lowest_U, lowest_j); for rows longer than 2047 bytes
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
// ---------------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void match_onehot(const u32 *__restrict__ cand, int S, const u32 (&o)[W], u32 &keyF, u32 &keyL)
{
	keyF = 0u; keyL = 0u;
	for (int s = 0; s < S; s++) {
		const u32 *c = cand + (size_t)s * W;       // wave-uniform address: scalar loads
		u32 pc = 0;
#pragma unroll
		for (int w = 0; w < W; w++) pc += (u32)__builtin_popcount(o[w] & c[w]);
		keyF = max(keyF, (pc << 16) | (0xffffu - (u32)s));
		keyL = max(keyL, (pc << 16) | (u32)s);
	}
}

template <int W>
__device__ __forceinline__ void demux_row_onehot(const uint8_t *tile, int row_start, const BarcodeDev &t,
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
	match_onehot<W>(t.onehot, t.S, o, keyF, keyL);
	diff = t.L - (int)(keyF >> 16);
	first = (int)(0xffffu - (keyF & 0xffffu));
	last = (int)(keyL & 0xffffu);
}

// byte-for-byte form: any sheet alphabet, any length
__device__ __forceinline__ void demux_row_bytes(const uint8_t *tile, int row_start, const BarcodeDev &t,
                                                int &diff, int &first, int &last)
{
	int lowest = 0x7fffffff; first = 0; last = 0;
	for (int s = 0; s < t.S; s++) {
		const uint8_t *c = t.raw + (size_t)s * t.L;
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
// The tile pass: one wavefront per 64-row tile, persistent over tiles.
// ---------------------------------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) uint8_t sk_smem[];

__device__ __forceinline__ void wave_lds_fence()
{
	// single-wave workgroup: LDS operations of one wave complete in order; this keeps the compiler from
	// moving LDS reads above the LDS writes of other lanes and waits for the writes to land.
	__syncthreads();
}

__global__ __launch_bounds__(64) void tile_pass_kernel(const TileArgs a, int hist_off, int use_lds_hist)
{
	const int lane = threadIdx.x;
	uint8_t *tile = sk_smem + kLdsPad;
	u32 *hist = reinterpret_cast<u32 *>(sk_smem + hist_off);
	const int S = a.table.S;
	const bool do_demux = a.bc != nullptr;

	if (do_demux && use_lds_hist) {
		for (int i = lane; i < S; i += kWave) hist[i] = 0u;
	}
	u32 n_total = 0, n_ident = 0, n_ambig = 0;      // wave-uniform

	const int64_t ntiles = (a.n + kTileRows - 1) / kTileRows;
	const int stride = a.stride;
	const int64_t total_bytes = a.n * (int64_t)stride;
	const int m = a.qc.min_baseq;
	const u32 cl2 = a.qc.cl2, c72 = a.qc.c72;
	const int mode = a.qc.mode;

	for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
		const int64_t row0 = t * kTileRows;
		const int rows = (int)((a.n - row0) < kTileRows ? (a.n - row0) : kTileRows);
		const bool active = lane < rows;

		for (int mi = 0; mi < a.n_mates; mi++) {
			const MateDev &mt = a.mate[mi];
			const bool do_mask = mt.out_seq != nullptr;
			const bool do_trim = mt.lowest_k != nullptr;
			if (!do_mask && !do_trim) continue;
			const int64_t base = row0 * (int64_t)stride;
			const int tile_bytes = rows * stride;

			// ---- stream phase: 16 B per lane, 1 KiB per wave instruction -------------------------
			for (int off = lane * 16; off < tile_bytes; off += kWave * 16) {
				const int64_t g = base + off;
				u32x4 q, s = {0u, 0u, 0u, 0u}, o, vq;
				const bool full = (g + 16 <= total_bytes);
				if (full) {
					q = *reinterpret_cast<const u32x4 *>(mt.qual + g);
					if (do_mask) s = *reinterpret_cast<const u32x4 *>(mt.seq + g);
				} else {
					const int valid = (int)(total_bytes - g);
					q = load_tail(mt.qual + g, valid);
					if (do_mask) s = load_tail(mt.seq + g, valid);
				}
				mask_dword4_rt(mode, q, s, cl2, c72, o, vq);
				if (do_mask) {
					if (full) *reinterpret_cast<u32x4 *>(mt.out_seq + g) = o;
					else store_tail(mt.out_seq + g, o, (int)(total_bytes - g));
				}
				if (do_trim) *reinterpret_cast<u32x4 *>(tile + off) = vq;
			}

			// ---- scan phase: lane r walks row r of the LDS image from its 3' end -------------------
			if (do_trim) {
				wave_lds_fence();
				int len = stride;
				if (mt.len != nullptr && active) len = (int)mt.len[row0 + lane];
				int k;
				if (stride < (1 << kKeyBits)) k = trim_scan_packed(tile, lane * stride, len, stride, m, active);
				else k = trim_scan_wide(tile, lane * stride, len, stride, m, active);
				if (active) mt.lowest_k[row0 + lane] = (uint16_t)k;
				wave_lds_fence();
			}
		}

		// ---- barcode phase ---------------------------------------------------------------------------
		if (do_demux) {
			const int bstride = a.bc_stride;
			const int64_t base = row0 * (int64_t)bstride;
			const int tile_bytes = rows * bstride;
			const int64_t bc_total = a.n * (int64_t)bstride;
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
				if (a.table.onehot != nullptr) {
					switch (a.table.W) {   // wave-uniform
					case 1: demux_row_onehot<1>(tile, rs, a.table, diff, first, last); break;
					case 2: demux_row_onehot<2>(tile, rs, a.table, diff, first, last); break;
					case 3: demux_row_onehot<3>(tile, rs, a.table, diff, first, last); break;
					case 4: demux_row_onehot<4>(tile, rs, a.table, diff, first, last); break;
					case 5: demux_row_onehot<5>(tile, rs, a.table, diff, first, last); break;
					case 6: demux_row_onehot<6>(tile, rs, a.table, diff, first, last); break;
					case 7: demux_row_onehot<7>(tile, rs, a.table, diff, first, last); break;
					default: demux_row_onehot<8>(tile, rs, a.table, diff, first, last); break;
					}
				} else {
					demux_row_bytes(tile, rs, a.table, diff, first, last);
				}
			}
			// D3: src/fasta_demultiplex.rs:168-194
			int code = kAssignNone;
			if (S > 0 && diff <= a.table.max_diff) code = (first == last) ? first : kAssignAmbiguous;
			if (active) {
				const int64_t r = row0 + lane;
				a.assign[r] = code;
				if (a.lowest_diff) a.lowest_diff[r] = (uint8_t)(diff > 255 ? 255 : diff);
				if (a.first_idx) a.first_idx[r] = (int16_t)first;
				if (a.last_idx) a.last_idx[r] = (int16_t)last;
				if (code >= 0) {
					if (use_lds_hist) atomicAdd(&hist[code], 1u);
					else atomicAdd(&a.counts[code], 1ull);
				}
			}
			n_total += (u32)__popcll(__ballot(active));
			n_ident += (u32)__popcll(__ballot(active && code >= 0));
			n_ambig += (u32)__popcll(__ballot(active && code == kAssignAmbiguous));
			wave_lds_fence();
		}
	}

	if (do_demux) {
		if (use_lds_hist) {
			wave_lds_fence();
			for (int i = lane; i < S; i += kWave) {
				u32 c = hist[i];
				if (c) atomicAdd(&a.counts[i], (unsigned long long)c);
			}
		}
		if (lane == 0) {
			if (n_total) atomicAdd(&a.counts[S], (unsigned long long)n_total);
			if (n_ident) atomicAdd(&a.counts[S + 1], (unsigned long long)n_ident);
			if (n_ambig) atomicAdd(&a.counts[S + 2], (unsigned long long)n_ambig);
		}
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

hipError_t launch_tile_pass(const TileArgs &a, int n_cu, hipStream_t st)
{
	if (a.n <= 0) return hipSuccess;
	TileArgs b = a;
	// rows that do not fit an LDS tile: trim falls back to the row-per-thread kernel, mask to the flat one
	bool any_mate = false;
	for (int mi = 0; mi < b.n_mates; mi++) any_mate = any_mate || b.mate[mi].out_seq || b.mate[mi].lowest_k;
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
		if (!b.bc) return hipSuccess;
	}
	int row_bytes = 0;
	for (int mi = 0; mi < b.n_mates; mi++)
		if (b.mate[mi].lowest_k || b.mate[mi].out_seq) row_bytes = b.stride;
	if (b.bc && b.bc_stride > row_bytes) row_bytes = b.bc_stride;
	int tile_bytes = (kTileRows * row_bytes + 15) & ~15;
	int hist_off = kLdsPad + tile_bytes + kLdsPad;
	int use_lds_hist = (b.bc && b.table.S + 3 <= kMaxLdsHist) ? 1 : 0;
	int lds = hist_off + (use_lds_hist ? (b.table.S + 3) * 4 : 0);
	lds = (lds + 15) & ~15;
	int per_cu = (160 * 1024) / lds;
	if (per_cu > 16) per_cu = 16;
	if (per_cu < 1) per_cu = 1;
	int64_t ntiles = (b.n + kTileRows - 1) / kTileRows;
	int64_t cap = (int64_t)n_cu * per_cu;
	int grid = (int)(ntiles < cap ? ntiles : cap);
	tile_pass_kernel<<<grid, kWave, lds, st>>>(b, hist_off, use_lds_hist);
	return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// S1 + H1: BAM flag counters and |TLEN| histogram.
// src/sam_statistics.rs:63-69; src/sam_fragment_lengths.rs:29-43.
// out = u64[3 counters][1 hist_total][max_frag+1 bins]
// ---------------------------------------------------------------------------------------------------
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

	u32 c_total = 0, c_aligned = 0, c_dup = 0, c_hist = 0;
	const int64_t step = (int64_t)gridDim.x * blockDim.x;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
		const u32 f = flag[i];
		if (want_counters) {
			const bool primary = (f & (0x100u | 0x800u)) == 0u;
			const bool mapped = primary && !(f & 0x4u);
			c_total += primary ? 1u : 0u;
			c_aligned += mapped ? 1u : 0u;
			c_dup += (mapped && (f & 0x400u)) ? 1u : 0u;
		}
		if (want_hist) {
			// paired, first, both mapped, not dup/secondary/supplementary, same reference
			const bool flags_ok = (f & (0x1u | 0x40u | 0x4u | 0x8u | 0x400u | 0x100u | 0x800u)) == (0x1u | 0x40u);
			if (flags_ok && tid[i] == mtid[i]) {
				const int32_t tl = tlen[i];
				// |tlen| in 64-bit like insert_size().abs(): INT32_MIN maps to 2^31 > any max_frag
				const u32 af = tl < 0 ? (u32)0 - (u32)tl : (u32)tl;
				if (af <= (u32)max_frag) {
					c_hist += 1u;
					if ((int)af < lds_bins) atomicAdd(&lh[af], 1u);
					else atomicAdd(&out[4 + af], 1ull);
				}
			}
		}
	}
	if (c_total) atomicAdd(&wg_cnt[0], c_total);
	if (c_aligned) atomicAdd(&wg_cnt[1], c_aligned);
	if (c_dup) atomicAdd(&wg_cnt[2], c_dup);
	if (c_hist) atomicAdd(&wg_cnt[3], c_hist);
	__syncthreads();
	if (threadIdx.x < 4 && wg_cnt[threadIdx.x]) atomicAdd(&out[threadIdx.x], (unsigned long long)wg_cnt[threadIdx.x]);
	for (int i = threadIdx.x; i < lds_bins; i += blockDim.x) {
		u32 c = lh[i];
		if (c) atomicAdd(&out[4 + i], (unsigned long long)c);
	}
}

hipError_t launch_bam_flag_tlen(const uint16_t *flag, const int32_t *tid, const int32_t *mtid, const int32_t *tlen,
                                int64_t n, int32_t max_frag, unsigned long long *out, int want_counters, int want_hist,
                                int n_cu, hipStream_t st)
{
	if (n <= 0) return hipSuccess;
	int64_t bins = want_hist ? (int64_t)max_frag + 1 : 0;
	int lds_bins = (int)(bins < 15360 ? bins : 15360);     // 60 KiB of LDS at most
	int64_t want = (n + 255) / 256;
	int grid = (int)(want < (int64_t)n_cu * 4 ? want : (int64_t)n_cu * 4);
	bam_flag_tlen_kernel<<<grid, 256, lds_bins * 4, st>>>(flag, tid, mtid, tlen, n, max_frag, out, want_counters, want_hist, lds_bins);
	return hipGetLastError();
}

}  // namespace sk
