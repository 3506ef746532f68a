// sk_census.hip — barcode census on gfx950: how often does each distinct barcode occur?
//
// Reference behaviour served (row f3 of SURVEY.md §8):
//   src/fasta_demultiplex.rs:190-194   `*extra_barcodes.entry(barcode).or_insert(0) += 1`  (dry run, unmatched reads)
//   src/fasta_statistics.rs:23-27      `*sample_barcodes.entry(sample_barcode).or_insert(0) += 1`
// Both are a HashMap<String, u64> fed one barcode per read.  Here the map is an open-addressing table in HBM
// keyed by the barcode packed at 4 bits per character, in front of which every workgroup keeps a small LDS
// table: the barcodes that dominate a run (the sample sheet's, plus their one-error neighbours) are counted
// with LDS atomics and reach HBM once per workgroup instead of once per read.
//
// Key: the alphabet is what the reference's regexes admit, " BC:[ACGTNacgtn+]+" — 11 symbols, code 1..11, code 0
// ends the barcode.  Nibble 0 of the low word is the marker 0xF (so a key is never 0, the empty-slot value);
// characters 0..14 follow in the low word, 15..30 in the high word: 31 characters at most.  The high word is
// stored inverted so that 0 means "claimed but not published yet" on a table that is cleared with memset.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sk_internal.h"

namespace sk {

typedef unsigned long long u64;
typedef uint32_t u32;

struct CensusSlot {          // 32 bytes, one HBM sector pair
	u64 klo;                 // 0 = empty
	u64 khi_inv;             // ~khi; 0 = owner has not published the high word yet
	u64 count;
	u64 first_inv;           // ~(lowest row index seen): atomicMax keeps the first occurrence
};

struct Census {
	CensusSlot *tab = nullptr;
	u64 slots = 0;           // power of two
	u64 *stats = nullptr;    // device u64[kCensusStats]
	u64 distinct = 0;        // host mirror of stats[0], valid after sync_stats()
	u64 *scratch = nullptr;  // device: entry compaction output / histogram
	size_t scratch_bytes = 0;
};

constexpr int kCensusStats = 4;          // [0] distinct keys, [1] rows counted, [2] rows rejected, [3] probe overflows
constexpr u64 kInitialSlots = 1ull << 26;   // 2 GiB of the 288: one launch may then take 32 M rows (SK_CENSUS_SLOTS_LOG2 overrides; tests use it)
constexpr int64_t kCensusChunk = 1 << 25;   // most rows per launch; the table is grown between launches so that it is never
constexpr int64_t kCensusMinChunk = 1 << 22;   // more than half full even if every row of the next launch is a new key
constexpr int kLdsSlots = 2048;
constexpr int kLdsProbes = 4;
constexpr u32 kMaxProbes = 1u << 16;

// The workgroup's front table, one array per field: slot i of a u64 array lies in bank pair i mod 32, so the 64 probes of a
// wave spread over all banks.  (As 32-byte records every slot began in one of 8 bank groups: SQ_LDS_BANK_CONFLICT was
// 1.5 x SQ_ACTIVE_INST_LDS, profiles/r02_census_pmc_summary.txt.)
struct LdsTable {
	u64 klo[kLdsSlots];
	u64 khi_inv[kLdsSlots];
	u64 first_inv[kLdsSlots];
	u32 count[kLdsSlots];
};

// 32-bit mix of the four key words; the HBM table uses the low bits, the LDS table the high bits
__device__ __forceinline__ u32 census_hash(u64 klo, u64 khi)
{
	const u32 w0 = (u32)klo, w1 = (u32)(klo >> 32), w2 = (u32)khi, w3 = (u32)(khi >> 32);
	u32 h = w0 * 0x9E3779B1u ^ __builtin_rotateleft32(w1, 13) * 0x85EBCA77u ^ w2 * 0xC2B2AE3Du ^ __builtin_rotateleft32(w3, 7) * 0x27D4EB2Fu;
	h ^= h >> 15;
	h *= 0x2C1B3C6Du;
	h ^= h >> 12;
	h *= 0x297A2D39u;
	h ^= h >> 15;
	return h;
}

// byte -> 4-bit code; 0 for NUL (end of barcode), 15 for a byte outside the alphabet.  The kernel reads this
// from a 256-byte table in LDS.
__host__ __device__ inline u32 census_code(u32 b)
{
	switch (b) {
	case 0: return 0;
	case 'A': return 1; case 'C': return 2; case 'G': return 3; case 'T': return 4; case 'N': return 5;
	case 'a': return 6; case 'c': return 7; case 'g': return 8; case 't': return 9; case 'n': return 10;
	case '+': return 11;
	default: return 15;
	}
}

struct SlotView { u64 k, v; };          // klo and ~khi of a slot as loaded at some earlier time

// one 16-byte load for the two key words (agent scope, like the atomic loads it replaces: sc1).  What bounds the HBM leg
// is the number of scattered memory operations (about 60 G/s chip-wide), not their bytes: a probe is this one load (three
// 8-byte loads before: every row a new key 7.2 -> 10 G rows/s), and the slot's first row is read only when the key matched.
__device__ __forceinline__ SlotView census_peek(const CensusSlot *s)
{
	typedef u32 u32x4_t __attribute__((ext_vector_type(4)));
	u32x4_t w;
	asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(w) : "v"(s) : "memory");
	__builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the compiler does not know of the load above
	asm volatile("" : "+v"(w));
	SlotView sv;
	sv.k = (u64)w[0] | ((u64)w[1] << 32);
	sv.v = (u64)w[2] | ((u64)w[3] << 32);
	return sv;
}

// add (cnt, first) for one key to the HBM table, starting at slot idx whose contents were fetched before (sv);
// returns false when the probe budget ran out.  sv may be stale: a slot's key never changes once it is published, and an
// empty or unpublished view is checked again (CAS / reload).
__device__ __forceinline__ bool census_insert_at(CensusSlot *tab, u64 mask, u64 idx, SlotView sv, u64 klo, u64 khi, u64 cnt, u64 first_inv, u32 &claimed)
{
	const u64 want = ~khi;
	u32 probes = 0;
	u64 k = sv.k, v = sv.v;
	while (probes < kMaxProbes) {
		CensusSlot *s = tab + idx;
		if (k == 0) {
			k = atomicCAS(&s->klo, 0ull, klo);
			if (k == 0) {
				// The slot is ours until its high word is published: everybody else who finds klo here waits for that
				// (v == 0 below).  So the first count and the first row go in as plain stores, and the store of the high
				// word hands the slot over — ONE atomic for a new key instead of three.
				// (the stores are write-through at agent scope and the wait is for their acknowledgement — an
				// agent-scope release would write the whole L2 back for every key)
				__hip_atomic_store(&s->count, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(&s->first_inv, first_inv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				__builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the workgroup fence alone does not wait for the two stores
				__hip_atomic_store(&s->khi_inv, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				claimed++;
				return true;
			}
			v = __hip_atomic_load(&s->khi_inv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if (k == klo) {
			if (v == 0) {                                              // owner is between its CAS and its publishing store: look again
				__builtin_amdgcn_s_sleep(1);
				v = __hip_atomic_load(&s->khi_inv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				continue;
			}
			if (v == want) {
				atomicAdd(&s->count, cnt);
				// (an unconditional atomicMax was measured: one more operation on an address every workgroup adds to — the
				// duplicate-heavy shapes lost 8 %; the first row is read, on a hit only, and folded in when it is earlier)
				if (__hip_atomic_load(&s->first_inv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < first_inv) atomicMax(&s->first_inv, first_inv);
				return true;
			}
		}
		idx = (idx + 1) & mask;
		probes++;
		const SlotView nx = census_peek(tab + idx);
		k = nx.k; v = nx.v;
	}
	return false;
}

__device__ __forceinline__ bool census_insert(CensusSlot *tab, u64 mask, u64 klo, u64 khi, u64 cnt, u64 first_inv, u32 &claimed)
{
	const u64 idx = (u64)census_hash(klo, khi) & mask;
	return census_insert_at(tab, mask, idx, census_peek(tab + idx), klo, khi, cnt, first_inv, claimed);
}

extern __shared__ __attribute__((aligned(16))) uint8_t census_smem[];

struct CensusArgs {
	const uint8_t *bc;
	int bc_stride;
	int L;
	int64_t n;
	const int32_t *assign;    // nullable: count row r only when assign[r] == SK_ASSIGN_NONE
	int64_t row_base;
	CensusSlot *tab;
	u64 mask;
	u64 *stats;
};

constexpr int kCensusWaves = 16;          // waves per workgroup (one per CU): they share the LDS table
constexpr int kCensusMaxStride = 64;      // 64 rows x 64 B = 4 KiB per wave tile = 4 x 16 B per lane

__device__ __forceinline__ void census_wave_fence()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kCensusMaxSub = 4;          // 64-row tiles per wave and step
constexpr int kCensusStepBytes = 5120;    // most bytes per wave and step: 5 x 16 B per lane in flight
constexpr int kCensusQueue = 128;         // flush queue entries (16 B key + 4 B row): fewer than 64 left over + one tile's 64

struct CensusTileRegs { uint4 v[5]; int32_t code[kCensusMaxSub]; };

// The R x 64 rows of step t are one contiguous, 16-byte aligned byte range: 16 bytes per lane and load, as unconditional
// raw-buffer loads whose descriptor ends at the range's last valid dword (the hardware drops what lies beyond it, so
// like the tile pass this may read up to 3 bytes past the end of the matrix).  What a wave has in flight is what
// bounds this kernel when everything is counted in LDS (one 1 KiB tile per wave: 2 TB/s), hence R tiles per step.
__device__ __forceinline__ void census_load_tile(const CensusArgs &a, int64_t t, int R, int lane, CensusTileRegs &rg)
{
	const int step_bytes = R * 64 * a.bc_stride;
	const int64_t total = a.n * (int64_t)a.bc_stride;
	const int64_t base = t * (int64_t)step_bytes;
	const int64_t rem = total - base;
	const int bytes = rem <= 0 ? 0 : (int)(rem < step_bytes ? rem : step_bytes);      // a step past the end loads nothing
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.bc) + (rem <= 0 ? 0 : base), 0, (bytes + 3) & ~3, 0x00020000);
	// No branches around the loads — a load inside a conditional block is waited for at the end of that block, which
	// serialises the five of them and the counting behind them; what lies beyond the step is clipped by the descriptor.
#pragma unroll
	for (int k = 0; k < 5; k++) {
		const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + k * 1024, 0, 0);
		memcpy(&rg.v[k], &v, 16);
	}
	const int64_t r0 = t * R * 64;
	const int64_t left = a.n - r0;
	const int rows = a.assign == nullptr || left <= 0 ? 0 : (int)(left < R * 64 ? left : R * 64);
	const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(a.assign) + (rows ? r0 : 0), 0, rows * 4, 0x00020000);
#pragma unroll
	for (int j = 0; j < kCensusMaxSub; j++)
		rg.code[j] = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(ra, j < R ? (j * 64 + lane) * 4 : 0x7ffffff0, 0, 0);
}

// One row per lane, R 64-row tiles per wave and step.  A wave keeps the next step's bytes in registers while it
// works on the current one (its private LDS tile), so the only workgroup barriers are the two around the loop.
// Keys are counted in the workgroup's LDS table; a key that finds no room there within kLdsProbes slots goes
// straight to HBM.  The LDS table is merged into HBM when the workgroup is done.
template <int R> __global__ __launch_bounds__(kCensusWaves * 64, 1) void census_kernel(const CensusArgs a, const int tile_slot)
{
	LdsTable *lt = reinterpret_cast<LdsTable *>(census_smem);
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int nwave = blockDim.x >> 6;
	uint8_t *lut = census_smem + sizeof(LdsTable);
	uint8_t *tile = lut + 256 + (size_t)wave * tile_slot;
	for (int i = tid; i < (int)(sizeof(LdsTable) / 8); i += blockDim.x) reinterpret_cast<u64 *>(lt)[i] = 0ull;
	if (tid < 256) lut[tid] = (uint8_t)census_code((u32)tid);
	__syncthreads();

	const int stride = a.bc_stride;
	const int step_bytes = R * 64 * stride;
	const int64_t nsteps = (a.n + (int64_t)R * 64 - 1) / ((int64_t)R * 64);
	const int64_t step = (int64_t)gridDim.x * nwave;
	u32 claimed = 0, counted = 0, rejected = 0, overflow = 0;
	CensusTileRegs rg;
	uint4 *qkey = reinterpret_cast<uint4 *>(tile);                 // the flush queue reuses the wave's tile
	u32 *qrel = reinterpret_cast<u32 *>(tile + kCensusQueue * 16);
	int64_t t = (int64_t)blockIdx.x * nwave + wave;
	census_load_tile(a, t, R, lane, rg);
	for (; t < nsteps; t += step) {
#pragma unroll
		for (int k = 0; k < 5; k++) {
			const int off = lane * 16 + k * 1024;
			if (off < step_bytes) *reinterpret_cast<uint4 *>(tile + off) = rg.v[k];
		}
		// which of the step's rows are counted, worked out BEFORE the next step's loads are issued: nothing below may
		// wait for a register that a load of this or an earlier step wrote, or it waits for the new loads as well
		u32 take = 0xFu;
		if (a.assign != nullptr) {
			take = 0u;
#pragma unroll
			for (int j = 0; j < kCensusMaxSub; j++) take |= (rg.code[j] == kAssignNone ? 1u : 0u) << j;
		}
		census_wave_fence();
		census_load_tile(a, t + step, R, lane, rg);       // in flight while this step is counted
		// Counting touches LDS only.  Keys the LDS table had no room for are parked in registers (one per lane and tile)
		// and go to HBM after the step's last tile: any memory operation in between would make the compiler wait for
		// the loads just issued (vmcnt is one in-order counter), and a loop header does the same, hence the unrolling.
		u64 pklo[R], pkhi[R];
		u32 parked = 0u;
#pragma unroll
		for (int j = 0; j < R; j++) {
		const int64_t r = (t * R + j) * 64 + lane;
		pklo[j] = 0ull;
		pkhi[j] = 0ull;
		if (r < a.n && ((take >> j) & 1u)) {
			// the row as dwords: aligned LDS reads funnel-shifted to the row's first byte
			const int rs = (j * 64 + lane) * stride;
			const u32 *t32 = reinterpret_cast<const u32 *>(tile) + (rs >> 2);
			const u32 sh = (u32)rs & 3u;
			u32 kw[4] = {0u, 0u, 0u, 0u};
			u32 live = 0xFu;
			u32 lo = t32[0];
#pragma unroll
			for (int q = 0; q < 8; q++) {
				if (4 * q < a.L) {                                   // wave-uniform
					const u32 hi = t32[q + 1];
					const u32 x = __builtin_amdgcn_alignbyte(hi, lo, sh);
					lo = hi;
#pragma unroll
					for (int b = 0; b < 4; b++) {
						const int jj = 4 * q + b;
						if (jj < kMaxCensusLen) {
							const u32 c = lut[(x >> (8 * b)) & 0xFFu] & live;
							if (c == 0u) live = 0u;                      // bytes after the first NUL are padding
							kw[(jj + 1) >> 3] |= c << (4 * ((jj + 1) & 7));
						}
					}
				}
			}
			// characters L.. of the last dword are not part of the barcode: clear them, then add the marker nibble
			u32 any15 = 0u;
#pragma unroll
			for (int w = 0; w < 4; w++) {
				const int keep = a.L + 1 - 8 * w;                      // nibbles of this word that hold characters (wave-uniform)
				kw[w] &= keep >= 8 ? 0xFFFFFFFFu : (keep <= 0 ? 0u : ((1u << (4 * keep)) - 1u));
				any15 |= kw[w] & (kw[w] >> 1) & (kw[w] >> 2) & (kw[w] >> 3) & 0x11111111u;
			}
			kw[0] |= 0xFu;
			const u64 klo = (u64)kw[0] | ((u64)kw[1] << 32), khi = (u64)kw[2] | ((u64)kw[3] << 32);
			const bool bad = any15 != 0u;                                // some nibble is 15: a byte outside the alphabet
			if (bad) rejected++;
			else {
				counted++;
				const u64 first_inv = ~(u64)(a.row_base + r);
				const u64 want = ~khi;
				u32 idx = (census_hash(klo, khi) >> 16) & (kLdsSlots - 1);
				bool done = false;
				for (int p = 0; p < kLdsProbes && !done;) {
					u64 k = __hip_atomic_load(&lt->klo[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					u64 v = __hip_atomic_load(&lt->khi_inv[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					u64 f = __hip_atomic_load(&lt->first_inv[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					if (k == 0) {
						k = atomicCAS(&lt->klo[idx], 0ull, klo);
						if (k == 0) {
							__hip_atomic_store(&lt->khi_inv[idx], want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
							k = klo;
							v = want;
						} else {
							v = __hip_atomic_load(&lt->khi_inv[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						}
						f = 0;
					}
					if (k == klo && v == 0) continue;                    // claimed, high word not published yet: look again
					if (k == klo && v == want) {
						atomicAdd(&lt->count[idx], 1u);
						if (f < first_inv) atomicMax(&lt->first_inv[idx], first_inv);
						done = true;
					} else {
						idx = (idx + 1) & (kLdsSlots - 1);
						p++;
					}
				}
				if (!done) { pklo[j] = klo; pkhi[j] = khi; parked |= 1u << j; }
			}
		}
		}
		census_wave_fence();
		// The parked keys, packed densely through the (now dead) tile so that one insert serves up to 64 of them: its
		// dependent round trips are paid per call, not per key.  (Fetching the slots a step ahead of the insert was tried:
		// no gain — what bounds this leg is the rate of scattered atomics, not their latency.)
		if (__any(parked != 0u)) {
			int qn = 0;
#pragma unroll
			for (int j = 0; j < R; j++) {
				const bool has = ((parked >> j) & 1u) != 0u;
				const u64 bal = __ballot(has);
				if (has) {
					const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
					qkey[pos] = make_uint4((u32)pklo[j], (u32)(pklo[j] >> 32), (u32)pkhi[j], (u32)(pkhi[j] >> 32));
					qrel[pos] = (u32)(j * 64 + lane);
				}
				qn += __popcll(bal);
				while (qn >= 64 || (j == R - 1 && qn > 0)) {
					census_wave_fence();
					const int cnt = qn < 64 ? qn : 64;
					if (lane < cnt) {
						const uint4 kq = qkey[qn - cnt + lane];
						const u32 rel = qrel[qn - cnt + lane];
						const u64 klo = (u64)kq.x | ((u64)kq.y << 32), khi = (u64)kq.z | ((u64)kq.w << 32);
						const u64 first_inv = ~(u64)(a.row_base + t * R * 64 + rel);
						if (!census_insert(a.tab, a.mask, klo, khi, 1ull, first_inv, claimed)) overflow++;
					}
					qn -= cnt;
					census_wave_fence();
				}
			}
		}
		census_wave_fence();
	}
	__syncthreads();
	// merge the workgroup's table into HBM; every workgroup starts somewhere else, so that the keys all of them
	// hold (the frequent ones) are not hit by all of them at the same moment
	for (int i0 = tid; i0 < kLdsSlots; i0 += blockDim.x) {
		const int i = (i0 + (int)blockIdx.x * 67) & (kLdsSlots - 1);
		const u64 sk = lt->klo[i];
		const u32 sc = lt->count[i];
		if (sk != 0 && !census_insert(a.tab, a.mask, sk, ~lt->khi_inv[i], (u64)sc, lt->first_inv[i], claimed)) overflow += sc;
	}
	// one atomic per wave and statistic
	for (int o = 32; o > 0; o >>= 1) {
		claimed += __shfl_xor(claimed, o);
		counted += __shfl_xor(counted, o);
		rejected += __shfl_xor(rejected, o);
		overflow += __shfl_xor(overflow, o);
	}
	if (lane == 0) {
		if (claimed) atomicAdd(&a.stats[0], (u64)claimed);
		if (counted) atomicAdd(&a.stats[1], (u64)counted);
		if (rejected) atomicAdd(&a.stats[2], (u64)rejected);
		if (overflow) atomicAdd(&a.stats[3], (u64)overflow);
	}
}

// grow: re-insert every slot of the old table into the new one
__global__ __launch_bounds__(256) void census_rehash_kernel(const CensusSlot *old_tab, u64 old_slots, CensusSlot *tab, u64 mask, u64 *stats)
{
	u32 claimed = 0, overflow = 0;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < old_slots; i += (u64)gridDim.x * blockDim.x) {
		const CensusSlot s = old_tab[i];
		if (s.klo != 0 && !census_insert(tab, mask, s.klo, ~s.khi_inv, s.count, s.first_inv, claimed)) overflow++;
	}
	if (overflow) atomicAdd(&stats[3], (u64)overflow);
}

// hist[b] += 1 for every key whose count has floor(log2(count)) == b
__global__ __launch_bounds__(256) void census_hist_kernel(const CensusSlot *tab, u64 slots, u64 *hist)
{
	__shared__ u32 lh[64];
	if (threadIdx.x < 64) lh[threadIdx.x] = 0;
	__syncthreads();
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (u64)gridDim.x * blockDim.x) {
		const u64 klo = tab[i].klo, c = tab[i].count;
		if (klo != 0 && c != 0) atomicAdd(&lh[63 - __clzll(c)], 1u);
	}
	__syncthreads();
	if (threadIdx.x < 64 && lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (u64)lh[threadIdx.x]);
}

// out[0] = number of entries written; entries (4 x u64 each, the slot as stored) follow from out + 4
__global__ __launch_bounds__(256) void census_compact_kernel(const CensusSlot *tab, u64 slots, u64 min_count, u64 *out, u64 cap)
{
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (u64)gridDim.x * blockDim.x) {
		const CensusSlot s = tab[i];
		if (s.klo != 0 && s.count >= min_count && s.count != 0) {
			const u64 at = atomicAdd(&out[0], 1ull);
			if (at < cap) {
				u64 *e = out + 4 + at * 4;
				e[0] = s.klo; e[1] = s.khi_inv; e[2] = s.count; e[3] = s.first_inv;
			}
		}
	}
}

// ---- host side -----------------------------------------------------------------------------------------------

static hipError_t census_alloc_table(Census *cs, u64 slots, hipStream_t st)
{
	hipError_t e = hipMalloc((void **)&cs->tab, slots * sizeof(CensusSlot));
	if (e != hipSuccess) { cs->tab = nullptr; return e; }
	cs->slots = slots;
	return hipMemsetAsync(cs->tab, 0, slots * sizeof(CensusSlot), st);
}

hipError_t census_create(Census **out, hipStream_t st)
{
	Census *cs = new Census();
	hipError_t e = hipMalloc((void **)&cs->stats, kCensusStats * sizeof(u64));
	if (e == hipSuccess) e = hipMemsetAsync(cs->stats, 0, kCensusStats * sizeof(u64), st);
	u64 init_slots = kInitialSlots;
	if (const char *ev = getenv("SK_CENSUS_SLOTS_LOG2")) {
		const int lg = atoi(ev);
		if (lg >= 10 && lg <= 32) init_slots = 1ull << lg;
	}
	if (e == hipSuccess) e = census_alloc_table(cs, init_slots, st);
	if (e == hipSuccess) e = hipFuncSetAttribute((const void *)census_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	if (e == hipSuccess) e = hipFuncSetAttribute((const void *)census_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	if (e == hipSuccess) e = hipFuncSetAttribute((const void *)census_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	if (e == hipSuccess) e = hipFuncSetAttribute((const void *)census_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	if (e != hipSuccess) { census_destroy(cs); return e; }
	*out = cs;
	return hipSuccess;
}

void census_destroy(Census *cs)
{
	if (!cs) return;
	if (cs->tab) (void)hipFree(cs->tab);
	if (cs->stats) (void)hipFree(cs->stats);
	if (cs->scratch) (void)hipFree(cs->scratch);
	delete cs;
}

hipError_t census_reset(Census *cs, hipStream_t st)
{
	hipError_t e = hipMemsetAsync(cs->stats, 0, kCensusStats * sizeof(u64), st);
	if (e == hipSuccess) e = hipMemsetAsync(cs->tab, 0, cs->slots * sizeof(CensusSlot), st);
	cs->distinct = 0;
	return e;
}

hipError_t census_stats(Census *cs, uint64_t out[4], hipStream_t st)
{
	u64 h[kCensusStats];
	hipError_t e = hipMemcpyAsync(h, cs->stats, sizeof h, hipMemcpyDeviceToHost, st);
	if (e == hipSuccess) e = hipStreamSynchronize(st);
	if (e != hipSuccess) return e;
	cs->distinct = h[0];
	for (int i = 0; i < kCensusStats; i++) out[i] = h[i];
	return hipSuccess;
}

uint64_t census_slots(const Census *cs) { return cs->slots; }

// make room for `incoming` more keys at a load factor of at most 1/2
static hipError_t census_reserve(Census *cs, u64 incoming, int n_cu, hipStream_t st)
{
	const u64 need = 2 * (cs->distinct + incoming);
	if (need <= cs->slots) return hipSuccess;
	u64 slots = cs->slots;
	while (slots < need) slots <<= 1;
	if (slots > (1ull << 32)) return hipErrorOutOfMemory;      // the hash is 32 bits wide (and this would be a 128 GiB table)
	CensusSlot *old_tab = cs->tab;
	const u64 old_slots = cs->slots;
	hipError_t e = census_alloc_table(cs, slots, st);
	if (e != hipSuccess) { (void)hipGetLastError(); cs->tab = old_tab; cs->slots = old_slots; return e; }
	census_rehash_kernel<<<n_cu * 8, 256, 0, st>>>(old_tab, old_slots, cs->tab, slots - 1, cs->stats);
	e = hipGetLastError();
	if (e == hipSuccess) e = hipStreamSynchronize(st);
	(void)hipFree(old_tab);
	return e;
}

hipError_t census_add(Census *cs, const uint8_t *bc, int bc_stride, int L, int64_t n, const int32_t *assign, int64_t row_base,
                      int n_cu, hipStream_t st)
{
	if (bc_stride > kCensusMaxStride) return hipErrorInvalidValue;
	int R = kCensusStepBytes / (64 * bc_stride);
	R = R < 1 ? 1 : (R > kCensusMaxSub ? kCensusMaxSub : R);
	if (const char *ev = getenv("SK_CENSUS_TILES")) { const int v = atoi(ev); if (v >= 1 && v <= R) R = v; }      // experiments
	int tile_slot = (R * 64 * bc_stride + 15) & ~15;
	if (tile_slot < kCensusQueue * 20) tile_slot = kCensusQueue * 20;
	const size_t lds = sizeof(LdsTable) + 256 + (size_t)kCensusWaves * tile_slot + 64;      // + slack: a row is read as 9 dwords
	// (SK_CENSUS_CHUNK_LOG2 / SK_CENSUS_MIN_CHUNK_LOG2: tests shrink the launches to walk the grow / smaller-bite decisions)
	int64_t chunk = kCensusChunk, min_chunk = kCensusMinChunk;
	if (const char *ev = getenv("SK_CENSUS_CHUNK_LOG2")) { const int lg = atoi(ev); if (lg >= 6 && lg <= 30) chunk = (int64_t)1 << lg; }
	if (const char *ev = getenv("SK_CENSUS_MIN_CHUNK_LOG2")) { const int lg = atoi(ev); if (lg >= 6 && lg <= 30) min_chunk = (int64_t)1 << lg; }
	if (min_chunk > chunk) min_chunk = chunk;
	int64_t nr = 0;
	for (int64_t o = 0; o < n; o += nr) {
		// The launch must not be able to fill the table beyond one half even if every row is a new key.  When
		// the bound on the key count says it could, fetch the exact count; then either take a smaller bite
		// (at least kCensusMinChunk rows: launches have a fixed cost) or grow the table.
		nr = (n - o) < chunk ? (n - o) : chunk;
		hipError_t e = hipSuccess;
		if (2 * (cs->distinct + (u64)nr) > cs->slots) {
			uint64_t s[4];
			e = census_stats(cs, s, st);
			if (e != hipSuccess) return e;
			const int64_t room = (int64_t)(cs->slots / 2) - (int64_t)cs->distinct;
			const int64_t least = nr < min_chunk ? nr : min_chunk;
			if (room >= least) nr = nr < room ? nr : room;        // a smaller bite fits the table as it is
			else e = census_reserve(cs, (u64)nr, n_cu, st);       // no room worth a launch: grow for the whole bite
			if (e != hipSuccess) return e;
			if (nr < n - o && nr >= 64) nr &= ~(int64_t)63;       // later launches start on a 16-byte boundary of the matrix
		}
		cs->distinct += (u64)nr;                              // upper bound until the next census_stats()
		CensusArgs a;
		a.bc = bc + o * (int64_t)bc_stride;
		a.bc_stride = bc_stride;
		a.L = L;
		a.n = nr;
		a.assign = assign ? assign + o : nullptr;
		a.row_base = row_base + o;
		a.tab = cs->tab;
		a.mask = cs->slots - 1;
		a.stats = cs->stats;
		const int64_t groups = (nr + (int64_t)64 * R * kCensusWaves - 1) / ((int64_t)64 * R * kCensusWaves);
		int grid = n_cu;
		if (grid > groups) grid = (int)groups;
		switch (R) {
		case 1: census_kernel<1><<<grid, kCensusWaves * 64, lds, st>>>(a, tile_slot); break;
		case 2: census_kernel<2><<<grid, kCensusWaves * 64, lds, st>>>(a, tile_slot); break;
		case 3: census_kernel<3><<<grid, kCensusWaves * 64, lds, st>>>(a, tile_slot); break;
		default: census_kernel<4><<<grid, kCensusWaves * 64, lds, st>>>(a, tile_slot); break;
		}
		e = hipGetLastError();
		if (e != hipSuccess) return e;
	}
	return hipSuccess;
}

static hipError_t census_scratch(Census *cs, size_t bytes)
{
	if (bytes <= cs->scratch_bytes) return hipSuccess;
	if (cs->scratch) { (void)hipFree(cs->scratch); cs->scratch = nullptr; cs->scratch_bytes = 0; }
	hipError_t e = hipMalloc((void **)&cs->scratch, bytes);
	if (e == hipSuccess) cs->scratch_bytes = bytes;
	return e;
}

hipError_t census_count_hist(Census *cs, uint64_t hist[64], int n_cu, hipStream_t st)
{
	hipError_t e = census_scratch(cs, 64 * sizeof(u64));
	if (e == hipSuccess) e = hipMemsetAsync(cs->scratch, 0, 64 * sizeof(u64), st);
	if (e != hipSuccess) return e;
	census_hist_kernel<<<n_cu * 8, 256, 0, st>>>(cs->tab, cs->slots, cs->scratch);
	e = hipGetLastError();
	if (e == hipSuccess) e = hipMemcpyAsync(hist, cs->scratch, 64 * sizeof(u64), hipMemcpyDeviceToHost, st);
	if (e == hipSuccess) e = hipStreamSynchronize(st);
	return e;
}

// entries with count >= min_count, at most cap of them, ordered by first occurrence; *total = how many qualify
hipError_t census_entries(Census *cs, uint64_t min_count, CensusEntry *out, uint64_t cap, uint64_t *total, int n_cu, hipStream_t st)
{
	uint64_t s[4];
	hipError_t e = census_stats(cs, s, st);
	if (e != hipSuccess) return e;
	const u64 room = std::min<u64>(cap, s[0]);
	e = census_scratch(cs, (4 + room * 4) * sizeof(u64));
	if (e == hipSuccess) e = hipMemsetAsync(cs->scratch, 0, 4 * sizeof(u64), st);
	if (e != hipSuccess) return e;
	census_compact_kernel<<<n_cu * 8, 256, 0, st>>>(cs->tab, cs->slots, min_count ? min_count : 1, cs->scratch, room);
	e = hipGetLastError();
	if (e != hipSuccess) return e;
	u64 found = 0;
	e = hipMemcpyAsync(&found, cs->scratch, sizeof found, hipMemcpyDeviceToHost, st);
	if (e == hipSuccess) e = hipStreamSynchronize(st);
	if (e != hipSuccess) return e;
	*total = found;
	const u64 got = std::min<u64>(found, room);
	std::vector<u64> raw(got * 4);
	if (got) {
		e = hipMemcpy(raw.data(), cs->scratch + 4, got * 4 * sizeof(u64), hipMemcpyDeviceToHost);
		if (e != hipSuccess) return e;
	}
	std::vector<u64> order(got);
	for (u64 i = 0; i < got; i++) order[i] = i;
	std::sort(order.begin(), order.end(), [&](u64 x, u64 y) { return raw[x * 4 + 3] > raw[y * 4 + 3]; });   // first_inv descending = first ascending
	static const char kAlphabet[] = "\0ACGTNacgtn+????";
	for (u64 i = 0; i < got; i++) {
		const u64 *r = &raw[order[i] * 4];
		const u64 klo = r[0], khi = ~r[1];
		CensusEntry &en = out[i];
		memset(en.barcode, 0, sizeof en.barcode);
		for (int j = 0; j < 31; j++) {
			const u32 c = (u32)((j < 15 ? klo >> (4 * (j + 1)) : khi >> (4 * (j - 15))) & 15u);
			if (c == 0) break;
			en.barcode[j] = kAlphabet[c];
		}
		en.count = r[2];
		en.first_row = (int64_t)~r[3];
	}
	return hipSuccess;
}

}  // namespace sk
