// sk_deflate.hip — F2 on the device: the per-sample gzip writers' DEFLATE, a BGZF block per wave.
//
// Reference behaviour served: src/common.rs:49-81 (GzipWriter: every output file is a pipe into a `gzip` / `pigz` child) behind
// src/fasta_demultiplex.rs:76-87,205-237.  What the reference's tests could pin is the DECOMPRESSED stream (the compressed bytes
// depend on the gzip build at hand): any valid DEFLATE of the same bytes is the same output.  Through round 5 the host deflated
// on CPU threads — `fasta demultiplex` of 8 M reads into 96 files was 27 CPU-seconds of deflate for 0.1 s of kernels
// (profiles/r04_cli_e2e.txt).  A BGZF block (SAMv1 §4.1) is a gzip member of at most 64 KiB that depends on nothing outside
// it: the writers' output is thousands of independent compression jobs.
//
// bgzf_deflate_kernel — ONE WAVE PER BLOCK of at most 0xff00 input bytes, three phases:
//   A  tokens.  64 positions at a time, a lane each: the four bytes at the position are hashed (12 bits), the table of last
//      positions (LDS) gives a candidate, the lane measures the match (4 bytes per step, up to 258); the table then takes
//      the chunk's own positions.  The greedy parse — a match of three or more is taken and skips what it covers — is a
//      scalar loop over the chunk's 64 lengths (v_readlane, a few instructions a position); the positions where a token
//      begins leave as 4-byte tokens (a ballot's prefix count places them) and are counted in the two histograms (LDS
//      atomics).
//   B  codes.  A real Huffman code per block for literals/lengths and for distances: the used symbols ranked by frequency
//      (every lane counts how many precede its symbols), the two-queue merge (a scalar loop, one step a node), depths from
//      the root down, the textbook retry with halved frequencies should a code come out longer than 15 bits, canonical codes,
//      bit-reversed for the stream.  The code-length alphabet takes a FIXED complete code (thirteen 4-bit and six 5-bit
//      codes, no run-length symbols used): the header is 190 bytes a block instead of 60, and a third tree is not built.
//   C  bits.  64 tokens at a time: a lane turns its token into at most 48 bits, a wave prefix sum gives its bit offset, the
//      lanes OR their bits into a staging area in LDS, whole dwords leave for the block's slot.
// A block that does not shrink is marked: the caller frames its bytes as a stored block.  bgzf_crc_out_kernel gives the
// CRC-32 of every block's input (sk_inflate.hip's slicing-by-four).  The caller frames the members (18-byte BGZF header,
// payload, CRC32, ISIZE): sk_bgzf_deflate in sk_capi.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "sk_internal.h"

namespace sk {

typedef uint32_t u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

constexpr int kDefWaves = 4;
constexpr int kDefHashBits = 12;
constexpr u32 kDefMaxIn = 0xff00u;
constexpr int kNLL = 286, kND = 30;

struct DefLds {                           // one wave's
	uint16_t head[1 << kDefHashBits];      // hash -> last position + 1 (0 = none)
	u32 freq[kNLL + kND];                  // histograms: literal/length symbols, then distance symbols
	uint16_t code[kNLL + kND];             // the codes, bit-reversed
	uint8_t len[kNLL + kND + 2];
	// Huffman construction (one alphabet at a time)
	u32 nfreq[2 * kNLL];                   // node weights: leaves in rank order, then internal nodes in creation order
	uint16_t parent[2 * kNLL];
	uint16_t order[kNLL];                  // rank -> symbol
	uint8_t depth[2 * kNLL];
	u32 stage[128];                        // phase C: the bits of 64 tokens (+ the carry)
};

// the fixed code of the code-length alphabet (symbols 0..18): lengths and bit-reversed canonical codes.
// Thirteen symbols of 4 bits (0 and 4..15) and six of 5 (1, 2, 3, 16, 17, 18): 13/16 + 6/32 = 1, a complete code.
__device__ const uint8_t kPreLen[19] = {4, 5, 5, 5, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 5, 5, 5};
__device__ const uint8_t kPreOrderD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ __forceinline__ void def_fence()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ u32 def_uniform(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u32 def_rev(u32 code, u32 len) { return __builtin_bitreverse32(code) >> (32u - len); }

// four bytes at any alignment (the buffer is readable to the next dword behind its last byte)
__device__ __forceinline__ u32 def_load4(const uint8_t *p)
{
	const uintptr_t a = (uintptr_t)p;
	const u32 *q = reinterpret_cast<const u32 *>(a & ~(uintptr_t)3);
	const u32 sh = (u32)(a & 3u);
	const u32 lo = q[0];
	return sh == 0u ? lo : __builtin_amdgcn_alignbyte(q[1], lo, sh);
}

// length 3..258 -> symbol 257..285, extra bits, extra value;  distance 1..32768 -> symbol 0..29, extra bits, extra value
__device__ __forceinline__ void def_len_code(u32 len, u32 &sym, u32 &xb, u32 &xv)
{
	const u32 l = len - 3u;
	if (l < 8u) { sym = 257u + l; xb = 0u; xv = 0u; return; }
	if (len == 258u) { sym = 285u; xb = 0u; xv = 0u; return; }
	const u32 nb = 31u - (u32)__builtin_clz(l);
	sym = 257u + 4u * (nb - 1u) + ((l >> (nb - 2u)) & 3u);
	xb = nb - 2u;
	xv = l & ((1u << xb) - 1u);
}
__device__ __forceinline__ void def_dist_code(u32 dist, u32 &sym, u32 &xb, u32 &xv)
{
	const u32 d = dist - 1u;
	if (d < 4u) { sym = d; xb = 0u; xv = 0u; return; }
	const u32 nb = 31u - (u32)__builtin_clz(d);
	sym = 2u * nb + ((d >> (nb - 1u)) & 1u);
	xb = nb - 1u;
	xv = d & ((1u << xb) - 1u);
}

// A Huffman code of at most 15 bits for freq[0 .. n): len[] (0 for unused symbols) and the bit-reversed canonical codes.
// Out of line (as sk_inflate.hip's symbol loop: a function of its own keeps its scalars in scalar registers).
__device__ __attribute__((noinline)) void def_huffman(int wave, int base, int n, int lane)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t def_smem[];
	DefLds &L = reinterpret_cast<DefLds *>(def_smem)[def_uniform((u32)wave)];
	base = (int)def_uniform((u32)base); n = (int)def_uniform((u32)n);
	u32 *const freq = L.freq + base;
	for (int shift = 0;; shift++) {
		// the used symbols in the order of their (scaled) frequencies: a lane counts, for each of its symbols, the symbols before it
		u32 used = 0u;
		for (int s0 = 0; s0 < n; s0 += 64) {
			const int s = s0 + lane;
			used += (u32)__popcll(__ballot(s < n && freq[s] != 0u));
		}
		for (int s = lane; s < n; s += 64) L.len[base + s] = 0;
		if (used == 0u) {                                              // nothing to code (no match in the block): one code of one bit, so that the alphabet is not empty
			if (lane == 0) { L.len[base] = 1; L.code[base] = 0; }
			def_fence();
			return;
		}
		if (used == 1u) {
			for (int s = lane; s < n; s += 64) if (freq[s] != 0u) { L.len[base + s] = 1; L.code[base + s] = 0; }
			def_fence();
			return;
		}
		for (int s = lane; s < n; s += 64) {
			const u32 f = freq[s];
			if (f == 0u) continue;
			const u32 fs = max(1u, f >> shift);
			u32 r = 0u;
			for (int t = 0; t < n; t++) {
				const u32 g = freq[t];
				if (g == 0u) continue;
				const u32 gs = max(1u, g >> shift);
				r += (gs < fs || (gs == fs && t < s)) ? 1u : 0u;
			}
			L.order[r] = (uint16_t)s;
			L.nfreq[r] = fs;
		}
		def_fence();
		// the two-queue merge: leaves 0 .. used-1 (ascending), internal nodes used .. 2 used - 2 in the order they are made (ascending too)
		{
			u32 li = 0u, ii = used, made = used;                           // next leaf, next internal node not yet merged, next node to make
			const u32 total = 2u * used - 1u;
			while (made < total) {
				u32 pick[2];
#pragma unroll
				for (int k = 0; k < 2; k++) {
					const bool leaf_ok = li < used, int_ok = ii < made;
					bool take_leaf = leaf_ok;
					if (leaf_ok && int_ok) take_leaf = def_uniform(L.nfreq[li]) <= def_uniform(L.nfreq[ii]);
					pick[k] = take_leaf ? li++ : ii++;
				}
				if (lane == 0) {
					L.nfreq[made] = L.nfreq[pick[0]] + L.nfreq[pick[1]];
					L.parent[pick[0]] = (uint16_t)made;
					L.parent[pick[1]] = (uint16_t)made;
				}
				def_fence();
				made++;
			}
			// depths from the root down (a node's parent was made after it)
			if (lane == 0) L.depth[total - 1u] = 0;
			def_fence();
			for (int i = (int)total - 2; i >= 0; i--) {
				if (lane == 0) L.depth[i] = (uint8_t)(L.depth[L.parent[i]] + 1);
				def_fence();
			}
		}
		u32 maxd = 0u;
		for (u32 r0 = 0; r0 < used; r0 += 64) {
			const u32 r = r0 + (u32)lane;
			u32 d = r < used ? L.depth[r] : 0u;
			for (int o = 32; o > 0; o >>= 1) d = max(d, (u32)__shfl_xor((int)d, o));
			maxd = max(maxd, d);
		}
		if (maxd > 15u) continue;                                          // halve the frequencies and build again (a Fibonacci-shaped histogram: rare)
		for (u32 r = (u32)lane; r < used; r += 64u) L.len[base + L.order[r]] = L.depth[r];
		def_fence();
		break;
	}
	// canonical codes: symbols of a length in symbol order, lengths ascending
	u32 cnt[16];
#pragma unroll
	for (int l = 0; l < 16; l++) cnt[l] = 0u;
	for (int s0 = 0; s0 < n; s0 += 64) {
		const int s = s0 + lane;
		const u32 ls = s < n ? L.len[base + s] : 0u;
#pragma unroll
		for (int l = 1; l < 16; l++) cnt[l] += (u32)__popcll(__ballot(ls == (u32)l));
	}
	u32 first[16];
	{
		u32 c = 0u;
		first[0] = 0u;
#pragma unroll
		for (int l = 1; l < 16; l++) { c = (c + cnt[l - 1]) << 1; first[l] = c; }
	}
	u32 run[16];
#pragma unroll
	for (int l = 0; l < 16; l++) run[l] = 0u;
	for (int s0 = 0; s0 < n; s0 += 64) {
		const int s = s0 + lane;
		const u32 ls = s < n ? L.len[base + s] : 0u;
#pragma unroll
		for (int l = 1; l < 16; l++) {
			const u64 m = __ballot(ls == (u32)l);
			if (ls == (u32)l) L.code[base + s] = (uint16_t)def_rev(first[l] + run[l] + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)), (u32)l);
			run[l] += (u32)__popcll(m);
		}
	}
	def_fence();
}

struct DefBlock { u64 in_off; u32 in_len, pad; };      // == sk_deflate_block
static_assert(sizeof(DefBlock) == 16, "== sk_deflate_block");

// the wave's bit writer (phase C): bits are ORed into L.stage, whole dwords leave for `dst`
struct DefOut { u32 *dst; u32 words; u32 carry_bits; };     // dwords written; bits waiting in stage[0]

__global__ __launch_bounds__(kDefWaves * 64) void bgzf_deflate_kernel(const uint8_t *in, const DefBlock *blocks, int64_t n_blocks, uint8_t *out, u32 out_stride,
                                                                      u32 *tokens, u32 *result)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t def_smem[];
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	DefLds &L = reinterpret_cast<DefLds *>(def_smem)[wave];
	for (int64_t bi = (int64_t)blockIdx.x * kDefWaves + wave; bi < n_blocks; bi += (int64_t)gridDim.x * kDefWaves) {
		const u64 in_off = (u64)def_uniform((u32)blocks[bi].in_off) | ((u64)def_uniform((u32)(blocks[bi].in_off >> 32)) << 32);
		const u32 n = def_uniform(blocks[bi].in_len);
		const uint8_t *src = in + in_off;
		u32 *const tok = tokens + (size_t)bi * kDefMaxIn;
		u32 *const dst = reinterpret_cast<u32 *>(out + (size_t)bi * out_stride);
		// ---- A: tokens
		for (int i = lane; i < (1 << kDefHashBits); i += 64) L.head[i] = 0;
		for (int i = lane; i < kNLL + kND; i += 64) L.freq[i] = 0u;
		def_fence();
		u32 ntok = 0u, skip = 0u;
		for (u32 c0 = 0; c0 < n; c0 += 64u) {
			const u32 p = c0 + (u32)lane;
			const u32 cnt = min(64u, n - c0);
			u32 mlen = 0u, mdist = 0u, v = 0u, h = 0u;
			const bool can = p + 4u <= n;                                  // (the last three positions of a block are literals)
			if (can) {
				v = def_load4(src + p);
				h = (v * 2654435761u) >> (32 - kDefHashBits);
				const u32 cand1 = L.head[h];
				if (cand1 != 0u) {
					const u32 cand = cand1 - 1u;
					const u32 d = p - cand;
					if (d <= 32768u && def_load4(src + cand) == v) {
						u32 l = 4u;
						const u32 maxl = min(258u, n - p);
						while (l + 4u <= maxl) {
							const u32 x = def_load4(src + cand + l) ^ def_load4(src + p + l);
							if (x != 0u) { l += (u32)__builtin_ctz(x) >> 3; break; }
							l += 4u;
						}
						if (l + 4u > maxl) { while (l < maxl && src[cand + l] == src[p + l]) l++; }
						mlen = l; mdist = d;
					}
				}
			}
			def_fence();
			if (can) L.head[h] = (uint16_t)(p + 1u);                       // (positions are below 0xff00)
			// the greedy parse of this chunk: a scalar walk over the 64 lengths
			u64 starts = 0ull;
			u32 q = skip;
			while (q < cnt) {
				const u32 l = (u32)__builtin_amdgcn_readlane((int)mlen, (int)q);
				starts |= 1ull << q;
				q += l >= 3u ? l : 1u;
			}
			skip = q - cnt;                                                // positions of the next chunk(s) a match of this one covers (a chunk wholly covered emits nothing)
			const bool mine = (starts >> lane) & 1ull;
			const bool is_match = mine && mlen >= 3u;
			if (mine) {
				const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(starts >> 32), __builtin_amdgcn_mbcnt_lo((u32)starts, 0u));
				u32 t;
				if (is_match) {
					t = 0x80000000u | ((mlen - 3u) << 16) | (mdist - 1u);
					u32 sym, xb, xv;
					def_len_code(mlen, sym, xb, xv);
					atomicAdd(&L.freq[sym], 1u);
					def_dist_code(mdist, sym, xb, xv);
					atomicAdd(&L.freq[kNLL + sym], 1u);
				} else {
					const u32 b = can ? (v & 0xffu) : (u32)src[p];
					t = b;
					atomicAdd(&L.freq[b], 1u);
				}
				tok[ntok + rank] = t;
			}
			ntok += (u32)__popcll(starts);
		}
		if (lane == 0) atomicAdd(&L.freq[256], 1u);                        // end of block
		def_fence();
		// ---- B: codes
		def_huffman(wave, 0, kNLL, lane);
		def_huffman(wave, kNLL, kND, lane);
		// ---- C: bits
		for (int i = lane; i < 128; i += 64) L.stage[i] = 0u;
		def_fence();
		u32 words = 0u, carry = 0u;                                        // dwords written to dst; bits waiting in stage[0]
		// one round of the writer: every lane has `nb` bits `bv` (nb <= 48 here, 0 = nothing)
		auto put = [&](u64 bv, u32 nb) {
			u32 off = nb;                                                    // inclusive prefix sum of the bit counts
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) {
				const u32 t = (u32)__shfl_up((int)off, o);
				if (lane >= o) off += t;
			}
			const u32 total = (u32)__builtin_amdgcn_readlane((int)off, 63);
			const u32 at = carry + off - nb;                                 // this lane's first bit in the staging area
			if (nb != 0u) {
				const u32 w = at >> 5, sh = at & 31u;
				const u64 lo = bv << sh;
				atomicOr(&L.stage[w], (u32)lo);
				if (sh + nb > 32u) atomicOr(&L.stage[w + 1u], (u32)(lo >> 32));
				if (sh + nb > 64u) atomicOr(&L.stage[w + 2u], (u32)(bv >> (64u - sh)));
			}
			def_fence();
			const u32 bits = carry + total;
			const u32 full = bits >> 5;                                      // whole dwords to write
			u32 keep = 0u;
			if ((u32)lane < full) dst[words + (u32)lane] = L.stage[lane];
			if ((u32)lane + 64u < full) dst[words + 64u + (u32)lane] = L.stage[64 + lane];
			keep = def_uniform(L.stage[full]);                               // the partial dword (stage has a spare one behind the last)
			def_fence();
			for (int i = lane; i < 128; i += 64) L.stage[i] = 0u;
			def_fence();
			if (lane == 0) L.stage[0] = keep;
			def_fence();
			words += full;
			carry = bits & 31u;
		};
		// the header: BFINAL = 1, BTYPE = 10, HLIT = 29 (286 codes), HDIST = 29 (30 codes), HCLEN = 15 (19 code-length codes), their 19 lengths
		{
			u64 bv = 0ull;
			u32 nb = 0u;
			if (lane == 0) { bv = 1ull | (2ull << 1) | (29ull << 3) | (29ull << 8) | (15ull << 13); nb = 17u; }
			else if (lane <= 19) { bv = kPreLen[kPreOrderD[lane - 1]]; nb = 3u; }
			put(bv, nb);
		}
		// the code lengths of the two alphabets, each with the fixed code-length code (canonical codes of kPreLen, bit-reversed)
		{
			// canonical code of symbol s of the code-length alphabet: the 4-bit symbols come first
			auto pre_code = [&](u32 s, u32 &cl) -> u32 {
				cl = kPreLen[s];
				u32 before = 0u;                                             // symbols of the same length before s
				for (u32 t = 0; t < s; t++) before += kPreLen[t] == cl ? 1u : 0u;
				const u32 c = cl == 4u ? before : (13u << 1) + before;        // first 5-bit code = (0 + 13) << 1
				return def_rev(c, cl);
			};
			for (int s0 = 0; s0 < kNLL + kND; s0 += 64) {
				const int s = s0 + lane;
				u64 bv = 0ull;
				u32 nb = 0u;
				if (s < kNLL + kND) { u32 cl; bv = pre_code(L.len[s], cl); nb = cl; }
				put(bv, nb);
			}
		}
		for (u32 t0 = 0; t0 < ntok + 1u; t0 += 64u) {
			const u32 ti = t0 + (u32)lane;
			u64 bv = 0ull;
			u32 nb = 0u;
			if (ti < ntok) {
				const u32 t = tok[ti];
				if (t & 0x80000000u) {
					const u32 len = ((t >> 16) & 0xffu) + 3u, dist = (t & 0x7fffu) + 1u;
					u32 sym, xb, xv;
					def_len_code(len, sym, xb, xv);
					bv = L.code[sym]; nb = L.len[sym];
					bv |= (u64)xv << nb; nb += xb;
					def_dist_code(dist, sym, xb, xv);
					bv |= (u64)L.code[kNLL + sym] << nb; nb += L.len[kNLL + sym];
					bv |= (u64)xv << nb; nb += xb;
				} else {
					bv = L.code[t]; nb = L.len[t];
				}
			} else if (ti == ntok) {
				bv = L.code[256]; nb = L.len[256];
			}
			put(bv, nb);
		}
		// the last bits
		if (carry != 0u) {
			if (lane == 0) dst[words] = L.stage[0];
			words++;
		}
		if (lane == 0) {
			const u32 nbytes = (words - (carry != 0u ? 1u : 0u)) * 4u + ((carry + 7u) >> 3);
			result[2 * bi] = nbytes;                                         // the payload's bytes (>= n + 5: the caller stores the block instead)
			result[2 * bi + 1] = ntok;
		}
		def_fence();
	}
}

// CRC-32 of every block's input (the polynomial and the tables of sk_inflate.hip's bgzf_crc_kernel)
__device__ __forceinline__ u32 dcrc_mul(u32 a, u32 b)
{
	u32 r = 0u;
	for (int i = 0; i < 32; i++) {
		if (a & 0x80000000u) r ^= b;
		a <<= 1;
		b = (b >> 1) ^ ((b & 1u) ? 0xEDB88320u : 0u);
	}
	return r;
}
__global__ __launch_bounds__(256) void bgzf_crc_out_kernel(const uint8_t *in, const DefBlock *blocks, int64_t n_blocks, u32 *crc_out)
{
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
			u32 p = 0x00800000u;
			for (int k = 0; k < 20; k++) { pw[k] = p; p = dcrc_mul(p, p); }
		}
	}
	__syncthreads();
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (int64_t bi = (int64_t)blockIdx.x * 4 + wave; bi < n_blocks; bi += (int64_t)gridDim.x * 4) {
		const u32 n = blocks[bi].in_len;
		const uint8_t *p = in + blocks[bi].in_off;
		const u32 mis = (u32)((16u - ((uintptr_t)p & 15u)) & 15u);
		const u32 headn = mis < n ? mis : n;
		const u32 body = n - headn;
		const u32 piece = (((body + 63u) >> 6) + 15u) & ~15u;
		const u32 lo = headn + min(body, piece * (u32)lane), hi = headn + min(body, piece * (u32)lane + piece);
		u32 c = lane == 0 ? 0xFFFFFFFFu : 0u;
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
		for (int o = 1; o < 64; o <<= 1) {
			const u32 cr = __shfl_down(c, o), lr = __shfl_down(len, o);
			if ((lane & (2 * o - 1)) == 0 && lane + o < 64) {
				u32 sh = 0x80000000u;
				u32 nb = lr;
				for (int k = 0; nb != 0u; k++, nb >>= 1) if (nb & 1u) sh = dcrc_mul(sh, pw[k]);
				c = dcrc_mul(c, sh) ^ cr;
				len += lr;
			}
		}
		if (lane == 0) crc_out[bi] = c ^ 0xFFFFFFFFu;
	}
}

hipError_t launch_bgzf_deflate(const uint8_t *in, const void *blocks, int64_t n_blocks, uint8_t *out, uint32_t out_stride, uint32_t *tokens, uint32_t *result,
                               uint32_t *crc, int n_cu, hipStream_t st)
{
	if (n_blocks <= 0) return hipSuccess;
	static bool attr_set = false;
	const size_t lds = sizeof(DefLds) * kDefWaves;
	if (!attr_set) {
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(bgzf_deflate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) return e;
		attr_set = true;
	}
	const int per_cu = std::max(1, (int)((160 * 1024) / lds));
	int64_t grid = (n_blocks + kDefWaves - 1) / kDefWaves;
	const int64_t cap = (int64_t)n_cu * per_cu * 4;
	if (grid > cap) grid = cap;
	bgzf_deflate_kernel<<<dim3((unsigned)grid), dim3(kDefWaves * 64), lds, st>>>(in, reinterpret_cast<const DefBlock *>(blocks), n_blocks, out, out_stride, tokens, result);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return e;
	int64_t cgrid = (n_blocks + 3) / 4;
	if (cgrid > (int64_t)n_cu * 16) cgrid = (int64_t)n_cu * 16;
	bgzf_crc_out_kernel<<<dim3((unsigned)cgrid), dim3(256), 0, st>>>(in, reinterpret_cast<const DefBlock *>(blocks), n_blocks, crc);
	return hipGetLastError();
}

size_t deflate_tokens_per_block() { return kDefMaxIn; }

}  // namespace sk
