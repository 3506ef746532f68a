// sk_census.hip — barcode census on gfx950: how often does each distinct barcode occur?
//
// Reference behaviour served (row f3 of SURVEY.md §8):
//   src/fasta_demultiplex.rs:190-194   `*extra_barcodes.entry(barcode).or_insert(0) += 1`  (dry run, unmatched reads)
//   src/fasta_statistics.rs:23-27      `*sample_barcodes.entry(sample_barcode).or_insert(0) += 1`
// Both are a HashMap<String, u64> fed one barcode per read.  Here the map is an open-addressing table in HBM
// keyed by the barcode packed at 4 bits per character, in front of which every workgroup keeps a table in LDS keyed by
// the rows' raw bytes: the barcodes that dominate a run (the sample sheet's, plus their one-error neighbours) are
// counted with LDS atomics — no key is built for them — and reach HBM once per launch (large launches: the workgroups'
// tables and the rows they had no room for are written out as records, partitioned by table region and combined) or
// once per workgroup (small launches) instead of once per read.
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

// The partition path (large launches): keys the front tables had no room for leave the front kernel as 16-byte records,
// partitioned by the region of the HBM table their slot lies in (kSpillBuckets regions) INSIDE that kernel: a record draws
// its place in the (workgroup, bucket) region of the record array from the workgroup's cursor of that bucket (an LDS add with
// return) and goes there with one 16-byte store.  A wave's 64 records go to 64 different lines; the lines of the
// workgroups' open regions (256 x 128 x 128 B = 4 MiB) stay in L2 until they are full.  The records are combined per bucket in
// LDS before they touch the table (census_add below).  (Through round 4 the records were written as they came, then
// counted, scanned and moved to 1 024 buckets by two more kernels: every record crossed HBM four times — 62 + 7 of a noisy
// launch's 314 us of kernels and two of its five kernel boundaries.  A first form of this round staged the records per
// bucket in each wave's LDS and flushed whole pieces: the stages took the front table's room and the flushes its time.)
constexpr int kSpillBucketsLog2 = 8;
constexpr int kSpillBuckets = 1 << kSpillBucketsLog2;
// a key's bucket: the top bits of its hash — the 1/256 of LEVEL 1 its home slot lies in, whatever the tables' sizes
__device__ __forceinline__ uint32_t census_bucket(uint32_t h) { return h >> (32 - kSpillBucketsLog2); }
constexpr int kSpillMaxLen = 24;         // 24 characters = 96 bits; the key's fourth dword is then the marker alone
constexpr int kSpillMaxGrid = 1024;      // workgroups of the front kernel the combine pass can address

struct CensusSpill {
	uint4 *key = nullptr;        // [grid][kSpillBuckets][cap]: a record is the key's first three dwords (barcodes of at most
	                             // kSpillMaxLen characters) and the row index within the launch | the weight
	u32 *hist = nullptr;         // [kSpillBuckets][grid]: records per bucket and workgroup (what lies in the region)
	u32 *btot = nullptr;         // [kSpillBuckets]: records per bucket (added to by the front kernel's workgroups; census_direct_kernel,
	                             // the launch's last kernel, leaves it zero for the next launch)
	u32 *wg_count = nullptr;     // [grid]: records per workgroup
	u32 *wg_stats = nullptr;     // [grid][kCensusStats]: the front kernel's statistics per workgroup (summed by census_combine_kernel)
	u32 *work = nullptr;         // census_combine_kernel's item counter (left zero by census_direct_kernel)
	u32 cap = 0;                 // records per (workgroup, bucket) region
	u32 direct_above = 0;        // more records than this in the launch: they are inserted as they lie (census_direct_kernel), not combined
	u32 grid = 0;                // workgroups of the front kernel
	u32 merge_copies = 0;        // != 0: the front kernel's tables leave their workgroup as records, not as inserts
};

// A record's fourth dword: the row index within the launch (a launch is at most 2^25 rows: kCensusChunk) and how many rows
// the record stands for — digit d = 1..31 at level v = 0..3, d << (5 v) rows: a row the front tables had no room for is
// (d, v) = (1, 0); a front-table key counted c times leaves the workgroup as the non-zero base-32 digits of c.
constexpr int kRecRowBits = 25;
__device__ __forceinline__ u32 census_rec_row(u32 w) { return w & ((1u << kRecRowBits) - 1u); }
__device__ __forceinline__ u32 census_rec_count(u32 w) { return (((w >> 27) & 31u) + 1u) << (5u * ((w >> kRecRowBits) & 3u)); }

struct Census {
	CensusSpill sp;
	size_t sp_records = 0;       // capacity of key / skey
	int sp_grid = 0;             // capacity of hist / wg_count in workgroups
	CensusSlot *tab = nullptr;
	u64 slots = 0;           // power of two
	u32 *log = nullptr;      // the claim log ("the HBM table and its CLAIM LOG"): kLogRegions regions of log_cap slot indices ...
	u32 *log_count = nullptr;    // ... and their counters, a 128-byte line each
	u32 log_cap = 0;
	u64 *stats = nullptr;    // device u64[kCensusStats]
	u64 distinct = 0;        // host mirror of stats[0], valid after sync_stats()
	u64 *scratch = nullptr;  // device: entry compaction output / histogram
	size_t scratch_bytes = 0;
};

constexpr int kCensusStats = 4;          // [0] distinct keys, [1] rows counted, [2] rows rejected, [3] probe overflows
constexpr u64 kInitialSlots = 1ull << 26;   // 2 GiB of the 288: one launch may then take 32 M rows (SK_CENSUS_SLOTS_LOG2 overrides; tests use it)
constexpr u32 kLogCap = 1u << 16;           // claim-log entries per region: 256 x 65 536 slots = 64 MiB (SK_CENSUS_LOG_CAP_LOG2 overrides; tests shrink it)
constexpr int64_t kCensusChunk = 1 << 25;   // most rows per launch; the table is grown between launches so that it is never
constexpr int64_t kCensusMinChunk = 1 << 22;   // more than half full even if every row of the next launch is a new key
constexpr int kLdsSlots = 2048;
constexpr int kLdsProbes = 8;             // (3 through round 4, when an item met a hundred distinct keys: with 256 buckets a noisy item meets 430 in 2 048
                                          // slots, and a key that finds no place sends every one of its records to HBM by itself)
constexpr u32 kMaxProbes = 1u << 16;

// The combine pass's table (keyed by the packed key), one array per field: slot i of a u64 array lies in bank pair i mod 32, so the 64 probes of a
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

// Four characters of a dword -> their 4-bit codes, each in the low nibble of its byte; `diff` gets a non-zero byte where
// the character is outside the alphabet "ACGTNacgtn+" (NUL, the end of a barcode, is code 0 and not outside).
//   lower-case letters are folded onto upper case (bit 5 cleared where bit 6 is set: 0x20 itself and '+' stay as they are),
//   bits 0..2 of y ^ (y >> 5) tell NUL + C A N G T apart (0 2 1 3 4 5 6: found by search), the byte that class stands
//   for comes from an eight-byte table (v_perm_b32) and must equal the folded byte, and the case is the code's bit 3.
constexpr u32 kClassLo = 0x412B4300u;      // classes 0..3: NUL C + A
constexpr u32 kClassHi = 0xFF54474Eu;      // classes 4..7: N G T (none)
__device__ __forceinline__ u32 census_classify(u32 x, u32 &diff)
{
	const u32 lower = x & (x >> 1) & 0x20202020u;
	const u32 y = x ^ lower;
	const u32 sel = (y ^ (y >> 5)) & 0x07070707u;
	diff = y ^ __builtin_amdgcn_perm(kClassHi, kClassLo, sel);
	return sel | (lower >> 2);
}
static const char kCensusAlphabet[17] = "\0C+ANGT??c?angt?";      // by code

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

// ---- the HBM table and its CLAIM LOG (round 6) ---------------------------------------------------------------------
// The table is sized for the worst launch — 2^26 slots, 2 GiB, never more than half full even if every one of a launch's 32 M rows
// is a new key — and no run of the reference's commands comes near that: a noisy 32 M-row run has 1.6 M distinct barcodes.  Through
// round 5 every reset cleared the 2 GiB (0.32 ms: as long as the census of 32 M rows itself).  Now whoever claims a slot writes its
// index into a log — the lanes of a wave that win their CAS in the same step append together: ONE atomic on the counter of their
// workgroup's region of the log, so that the log costs a claim a 4-byte store and the counters are never contended — and a reset
// clears the logged slots (two 16-byte stores each) and the counters.  A region that overflows (the log holds 16 M slots: a
// launch of mostly new keys) makes the next reset clear the whole table as before, decided on the device.  (A first form of this
// round put a small table of 2^22 slots in front of the big one: a key whose 8-slot group was full went on to the big table.  The
// noisy run put 0.5 % of its keys there — the big table then had to be cleared after all — and a launch of 32 M new keys walked
// eight slots of the saturated small table before every insert: 18.9 ms where this table takes 2.8.)
// A key's home slot is the TOP bits of its hash: its bucket (census_bucket, the top 8) is then the 1/256 of the table it lies in
// whatever the table's size, and the combine pass's LDS table takes the low bits.
constexpr int kLogRegions = 256;           // regions of the claim log, one per workgroup modulo
struct CensusTab {
	CensusSlot *tab;
	u64 mask;                // slots - 1
	u32 sh;                  // home slot = hash >> sh
	u32 *log;                // [kLogRegions][log_cap]: slot indices claimed since the last reset
	u32 *log_count;          // [kLogRegions * 32]: entries of region r at [32 r] (a 128-byte line each); beyond log_cap: the region overflowed
	u32 log_cap;
};

// the lanes that are active here have each claimed slot s: they append to their workgroup's region of the log with one atomic
__device__ __forceinline__ void census_log_claim(const CensusTab &t, const CensusSlot *s)
{
	if (t.log_cap == 0u) return;                                        // (SK_CENSUS_LOG_CAP_LOG2=-1: no log, every reset clears the whole table — A/B)
	const u64 m = __ballot(1);
	const int leader = __ffsll((unsigned long long)m) - 1;
	const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
	const u32 region = blockIdx.x & (kLogRegions - 1);
	u32 base = 0u;
	if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(&t.log_count[region * 32u], (u32)__popcll(m));
	base = (u32)__builtin_amdgcn_readlane((int)base, leader);
	if (base + rank < t.log_cap) t.log[(size_t)region * t.log_cap + base + rank] = (u32)(s - t.tab);
}

// One slot of a probe walk, its two key words (k, v) as fetched at some earlier time (a slot's key never changes once it is
// published, and an empty or unpublished view is checked again: CAS / reload).  true = the key is in, claimed here or added
// to; false = the slot is another key's.
__device__ __forceinline__ bool census_try_slot(const CensusTab &t, CensusSlot *s, u64 k, u64 v, u64 klo, u64 want, u64 cnt, u64 first_inv, u32 &claimed)
{
	bool done = false;
	if (k == 0) {
		k = atomicCAS(&s->klo, 0ull, klo);
		const bool won = k == 0;
		if (won) {
			// The slot is ours until its high word is published: everybody else who finds klo here waits for that
			// (v == 0 below).  So the first count and the first row go in as plain stores, and the store of the high
			// word hands the slot over — ONE atomic for a new key instead of three.
			// (the stores are write-through at agent scope and the wait is for their acknowledgement — an
			// agent-scope release would write the whole L2 back for every key)
			{	// count and first row are the slot's second half: ONE 16-byte store (scattered stores, like atomics, cost per operation)
				typedef u32 u32x4_t __attribute__((ext_vector_type(4)));
				const u32x4_t w = {(u32)cnt, (u32)(cnt >> 32), (u32)first_inv, (u32)(first_inv >> 32)};
				asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(&s->count), "v"(w) : "memory");
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			__builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the workgroup fence alone does not wait for the store
			__hip_atomic_store(&s->khi_inv, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			claimed++;
			census_log_claim(t, s);
		}
		// The winner's lanes and the loser's meet again HERE, before anybody waits for a publishing store: a lane that lost the slot
		// to another lane OF ITS OWN WAVE for the same key spins below until that lane has published — which it has, by now.  (The
		// barrier is a convergent no-op: it keeps the compiler from threading the winner's path past this point to the exit and
		// ordering the loser's spin loop in front of the winner's stores — a wave would then wait for itself.)
		__builtin_amdgcn_wave_barrier();
		done = won;
		if (!won) v = __hip_atomic_load(&s->khi_inv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	if (!done && k == klo) {
		while (v == 0) {                                               // owner is between its CAS and its publishing store: look again
			__builtin_amdgcn_s_sleep(1);
			v = __hip_atomic_load(&s->khi_inv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if (v == want) {
			atomicAdd(&s->count, cnt);
			// (an unconditional atomicMax was measured: one more operation on an address every workgroup adds to — the
			// duplicate-heavy shapes lost 8 %; the first row is read, on a hit only, and folded in when it is earlier)
			if (__hip_atomic_load(&s->first_inv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < first_inv) atomicMax(&s->first_inv, first_inv);
			done = true;
		}
	}
	return done;
}

// add (cnt, first) for one key, starting at slot idx whose contents were fetched before (sv); returns false when the probe
// budget ran out
__device__ __forceinline__ bool census_insert_at(const CensusTab &t, u64 idx, SlotView sv, u64 klo, u64 khi, u64 cnt, u64 first_inv, u32 &claimed)
{
	const u64 want = ~khi;
	for (u32 probes = 0; probes < kMaxProbes; probes++) {
		if (census_try_slot(t, t.tab + idx, sv.k, sv.v, klo, want, cnt, first_inv, claimed)) return true;
		idx = (idx + 1) & t.mask;
		sv = census_peek(t.tab + idx);
	}
	return false;
}

__device__ __forceinline__ u64 census_home_idx(const CensusTab &t, u32 h) { return (u64)(h >> t.sh) & t.mask; }
__device__ __forceinline__ CensusSlot *census_home(const CensusTab &t, u32 h) { return t.tab + census_home_idx(t, h); }

// the key's walk from its home slot on, whose contents were fetched before (home)
__device__ __forceinline__ bool census_insert_from(const CensusTab &t, u32 h, SlotView home, u64 klo, u64 khi, u64 cnt, u64 first_inv, u32 &claimed)
{
	return census_insert_at(t, census_home_idx(t, h), home, klo, khi, cnt, first_inv, claimed);
}

__device__ __forceinline__ bool census_insert(const CensusTab &t, u64 klo, u64 khi, u64 cnt, u64 first_inv, u32 &claimed)
{
	const u32 h = census_hash(klo, khi);
	return census_insert_from(t, h, census_peek(census_home(t, h)), klo, khi, cnt, first_inv, claimed);
}

// N keys of a thread at once: what bounds an insert is its chain of dependent memory round trips (the slot, the CAS, the
// acknowledged store, the publishing store), and a loop over keys pays the chain per key — with lanes that have nothing to
// insert waiting beside those that have.  Here the N home slots are fetched together, the CAS of those found empty are in
// flight together, the winners' stores are acknowledged by ONE wait; a key already there takes its add (and the first-row
// check) likewise.  Whatever is left — a slot taken by another key, a lost race — goes the long way, census_insert_from.
template <int N>
__device__ __forceinline__ void census_insert_many(const CensusTab &t, const bool (&want)[N], const u64 (&klo)[N], const u64 (&khi)[N], const u32 (&cnt)[N],
                                                   const u64 (&first_inv)[N], u32 &claimed, u32 &overflow)
{
	typedef u32 u32x4_t __attribute__((ext_vector_type(4)));
	CensusSlot *hs[N];
	u32 hh[N];
	u32x4_t w[N];
#pragma unroll
	for (int q = 0; q < N; q++) {
		hh[q] = want[q] ? census_hash(klo[q], khi[q]) : 0u;
		hs[q] = census_home(t, hh[q]);
		asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(w[q]) : "v"(hs[q]) : "memory");
	}
	__builtin_amdgcn_s_waitcnt(0x0F70);                                // vmcnt(0): the compiler does not know of the loads above
	u64 k[N], v[N], old[N];
	bool tryc[N], won[N], hit[N];
#pragma unroll
	for (int q = 0; q < N; q++) {
		asm volatile("" : "+v"(w[q]));
		k[q] = (u64)w[q][0] | ((u64)w[q][1] << 32);
		v[q] = (u64)w[q][2] | ((u64)w[q][3] << 32);
		tryc[q] = want[q] && k[q] == 0ull;
		old[q] = 1ull;
		if (tryc[q]) old[q] = atomicCAS(&hs[q]->klo, 0ull, klo[q]);
	}
	bool any_won = false;
#pragma unroll
	for (int q = 0; q < N; q++) {
		won[q] = tryc[q] && old[q] == 0ull;
		any_won = any_won || won[q];
		if (won[q]) {                                                  // ours until the high word is published (census_try_slot)
			const u32x4_t cw = {cnt[q], 0u, (u32)first_inv[q], (u32)(first_inv[q] >> 32)};
			asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(&hs[q]->count), "v"(cw) : "memory");
		}
	}
	if (any_won) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0): the stores are acknowledged
	}
	u64 f[N];
#pragma unroll
	for (int q = 0; q < N; q++) {
		if (won[q]) {
			__hip_atomic_store(&hs[q]->khi_inv, ~khi[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			claimed++;
			census_log_claim(t, hs[q]);
		}
		hit[q] = want[q] && !tryc[q] && k[q] == klo[q] && v[q] == ~khi[q];
		f[q] = ~0ull;
		if (hit[q]) {
			atomicAdd(&hs[q]->count, (u64)cnt[q]);
			f[q] = __hip_atomic_load(&hs[q]->first_inv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
#pragma unroll
	for (int q = 0; q < N; q++) {
		if (hit[q] && f[q] < first_inv[q]) atomicMax(&hs[q]->first_inv, first_inv[q]);
		if (want[q] && !won[q] && !hit[q]) {
			SlotView sv;
			sv.k = tryc[q] ? old[q] : k[q];
			sv.v = tryc[q] ? __hip_atomic_load(&hs[q]->khi_inv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : v[q];
			if (!census_insert_from(t, hh[q], sv, klo[q], khi[q], (u64)cnt[q], first_inv[q], claimed)) overflow += cnt[q];
		}
	}
}

// N records of a thread counted in the combine pass's table together.  A wave walks a probe sequence as long as its longest lane
// does, and a thread's records one after the other summed those walks.  Here every turn of the loop serves all the thread's records that are still looking —
// their slots' three words are fetched together — so the loop is as long as the longest walk among 64 N records, not the sum
// of N longest walks among 64.  fail[q]: no place within kLdsProbes slots (the key then goes to HBM by itself).
template <int N>
__device__ __forceinline__ void lds_count_many(LdsTable *lt, const bool (&have)[N], const u32 (&h)[N], const u64 (&klo)[N], const u64 (&khi)[N], const u32 (&cnt)[N],
                                               const u64 (&first_inv)[N], bool (&fail)[N])
{
	u32 idx[N];
	int p[N];
	bool todo[N];
#pragma unroll
	for (int q = 0; q < N; q++) { idx[q] = h[q] & (kLdsSlots - 1); p[q] = 0; todo[q] = have[q]; fail[q] = false; }
	for (;;) {
		bool any = false;
#pragma unroll
		for (int q = 0; q < N; q++) any = any || todo[q];
		if (!__any(any)) break;                                            // (uniform)
		u64 k[N], v[N], f[N];
#pragma unroll
		for (int q = 0; q < N; q++) {
			k[q] = __hip_atomic_load(&lt->klo[idx[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			v[q] = __hip_atomic_load(&lt->khi_inv[idx[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			f[q] = __hip_atomic_load(&lt->first_inv[idx[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
#pragma unroll
		for (int q = 0; q < N; q++) {
			if (!todo[q]) continue;
			const u64 want = ~khi[q];
			bool mine = k[q] == klo[q] && v[q] == want;
			if (!mine && k[q] == 0ull) {
				if (atomicCAS(&lt->klo[idx[q]], 0ull, klo[q]) == 0ull) {
					__hip_atomic_store(&lt->khi_inv[idx[q]], want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					mine = true;
					f[q] = 0ull;
				}
				// (lost to another lane: the slot is looked at again in the next turn — it may have become this key's)
			} else if (!mine && !(k[q] == klo[q] && v[q] == 0ull)) {         // another key's (a claimed slot whose high word is not there yet is looked at again)
				idx[q] = (idx[q] + 1u) & (kLdsSlots - 1);
				if (++p[q] >= kLdsProbes) { fail[q] = true; todo[q] = false; }
			}
			if (mine) {
				atomicAdd(&lt->count[idx[q]], cnt[q]);
				if (f[q] < first_inv[q]) atomicMax(&lt->first_inv[idx[q]], first_inv[q]);
				todo[q] = false;
			}
		}
	}
}

// ---- the workgroup's front table: keyed by a row's BYTES ---------------------------------------------------------------
// Building a row's key — classify every character, pack, hash — is most of what the census does per row, and most rows of a
// run repeat a few thousand byte strings (the sheet's barcodes and their neighbours).  So the table a workgroup keeps in LDS
// in front of the HBM table is keyed by the row's L RAW bytes (its masked dwords as they lie in memory): a row whose
// string is there is one hash of its dwords, one entry read, a compare and the LDS add — no key is built for it at all.  A
// string enters the table on the long way (below), after its characters were checked; what the long way checks — the
// alphabet; a NUL before L ends the barcode, and the bytes behind it are zeroed before the string is looked up or stored —
// then holds for every later row that matches the entry byte for byte.  Keys are built once per ENTRY, when the table
// leaves the workgroup.  (Rounds 2-3 kept a table keyed by the packed key with this one as an "alias" in front of it: two
// lookups, two installs and two sets of probes per new string.)
//
// The table is three or four arrays of `entries` elements, so that neighbouring entries lie in neighbouring banks (as one
// 32-byte record per entry, the counts of ALL entries lay in four of the 32 banks — every add of a wave queued up behind
// the others: SQ_LDS_BANK_CONFLICT was 61 % of SQ_LDS_IDX_ACTIVE): chunk B {string dword 4, 0, first row, state}, chunk A
// {string dwords 0..3}, for strings of more than 20 bytes chunk C {dwords 4..7} (B's first dword is then unused), and the
// counts (u32).  The first row is the index within the launch.  Two candidate places per string.
// Protocol: a writer claims an EMPTY entry with a CAS on the state (-> BUSY), writes chunk A (and C), then chunk B — its
// row, READY — as ONE 16-byte store; its count goes in with an add like everybody's.  A reader loads chunk B FIRST: the LDS
// executes a wave's operations in order and a lane's 16 bytes in one cycle, so a READY seen there means the string read after
// it is complete (the loads are volatile asm for that: nothing may reorder them).
constexpr u32 kFrontEmpty = 0u, kFrontBusy = 1u, kFrontReady = 2u;
template <int NW> struct FrontShape { static constexpr int CH = NW <= 5 ? 2 : 3; static constexpr int kEntryBytes = CH * 16 + 4; };
typedef u32 u32x4_t __attribute__((ext_vector_type(4)));
template <int NW> struct FrontWords {
	u32x4_t b, a, c;                                                   // (c: only for NW > 5)
	__device__ __forceinline__ u32 word(int q) const { return q < 4 ? a[q & 3] : (NW <= 5 ? b[q & 3] : c[q & 3]); }
	__device__ __forceinline__ u32 first() const { return b[2]; }
	__device__ __forceinline__ u32 state() const { return b[3]; }
};
// where the table's arrays begin (LDS byte addresses) and how many entries they have
struct FrontTable {
	u32 b, a, c, cnt, entries;
};

template <int NW> __device__ __forceinline__ u32 front_hash(const u32 (&xs)[NW])
{
	u32 h = xs[0];
#pragma unroll
	for (int q = 1; q < NW; q++) h ^= __builtin_rotateleft32(xs[q], (7 * q + 4) & 31);
	h *= 0x9E3779B1u;
	return h ^ (h >> 15);
}
// the string's two places (entry indices)
__device__ __forceinline__ u32 front_place(u32 h, int which, u32 entries)
{
	const u32 g = which == 0 ? h : h * 0x85EBCA77u + 0x165667B1u;
	return __umulhi(g, entries);
}
// issue the loads of entry e (the chunk with the state first); front_wait() before the words are used
template <int NW> __device__ __forceinline__ void front_issue(const FrontTable &ft, u32 e, FrontWords<NW> &w)
{
	asm volatile("ds_read_b128 %0, %1" : "=&v"(w.b) : "v"(ft.b + e * 16u) : "memory");
	asm volatile("ds_read_b128 %0, %1" : "=&v"(w.a) : "v"(ft.a + e * 16u) : "memory");
	if (NW > 5) asm volatile("ds_read_b128 %0, %1" : "=&v"(w.c) : "v"(ft.c + e * 16u) : "memory");
}
template <int NW> __device__ __forceinline__ void front_wait(FrontWords<NW> &w)
{
	__builtin_amdgcn_s_waitcnt(0xC07F);                                // lgkmcnt(0): the compiler does not know of the loads above
	asm volatile("" : "+v"(w.b));
	asm volatile("" : "+v"(w.a));
	if (NW > 5) asm volatile("" : "+v"(w.c));
}
// a claimed entry's string, then its row and READY
template <int NW> __device__ __forceinline__ void front_publish(const FrontTable &ft, u32 e, const u32 (&xs)[NW], u32 row)
{
	u32 x[8];
#pragma unroll
	for (int q = 0; q < 8; q++) x[q] = q < NW ? xs[q < NW ? q : 0] : 0u;
	const u32x4_t wa = {x[0], x[1], x[2], x[3]};
	const u32x4_t wc = {x[4], x[5], x[6], x[7]};
	const u32x4_t wb = {NW <= 5 ? x[4] : 0u, 0u, row, kFrontReady};
	asm volatile("ds_write_b128 %0, %1" :: "v"(ft.a + e * 16u), "v"(wa) : "memory");
	if (NW > 5) asm volatile("ds_write_b128 %0, %1" :: "v"(ft.c + e * 16u), "v"(wc) : "memory");
	asm volatile("ds_write_b128 %0, %1" :: "v"(ft.b + e * 16u), "v"(wb) : "memory");
}
template <int NW> __device__ __forceinline__ bool front_match(const FrontWords<NW> &w, const u32 (&xs)[NW])
{
	u32 df = 0u;
#pragma unroll
	for (int q = 0; q < NW; q++) df |= w.word(q) ^ xs[q];
	return df == 0u && w.state() == kFrontReady;
}
// Four characters of a dword of a (canonical) string -> their codes in the key's layout; `bad` collects the bytes outside the alphabet
template <int NW> __device__ __forceinline__ void census_key_of(const u32 (&xs)[NW], u64 &klo, u64 &khi, u32 &bad)
{
	u32 c[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
	bad = 0u;
#pragma unroll
	for (int q = 0; q < NW; q++) {
		u32 d;
		c[q] = census_classify(xs[q], d);
		bad |= d;
	}
	const u32 kw0 = c[0] | (c[1] << 4), kw1 = c[2] | (c[3] << 4), kw2 = c[4] | (c[5] << 4), kw3 = c[6] | (c[7] << 4) | 0xF0000000u;
	klo = (u64)kw3 | ((u64)kw0 << 32);
	khi = (u64)kw1 | ((u64)kw2 << 32);
}
// zero the bytes from the string's first NUL on (kms: the bytes before L); returns false when there was none
template <int NW> __device__ __forceinline__ bool census_canonical(u32 (&xs)[NW], const u32 (&kms)[NW])
{
	u32 alive = 0xFFFFFFFFu;
	bool any = false;
#pragma unroll
	for (int q = 0; q < NW; q++) {
		const u32 x = xs[q];
		const u32 zf = ~((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x)) & 0x80808080u & kms[q];      // bit 7 of the bytes that are NUL
		const u32 m = (zf != 0u ? (1u << (__builtin_ctz(zf) & 31)) - 1u : 0xFFFFFFFFu) & alive;
		xs[q] = x & m;
		if (zf != 0u) { alive = 0u; any = true; }
	}
	return any;
}

extern __shared__ __attribute__((aligned(16))) uint8_t census_smem[];

// -DSK_CENSUS_STAMPS (diagnostic builds only, tools/census_stamps.sh): every wave of the front kernel adds up the shader
// cycles (s_memtime) it spent in each phase of its steps; sk_debug_census_stamps() reads the sums back.  No stamp executes in
// the product build.
#ifdef SK_CENSUS_STAMPS
constexpr int kStampSlots = 16;
__device__ u64 g_census_stamps[8192 * kStampSlots];
__device__ u64 g_combine_stamps[8192 * kStampSlots];
#define SK_CSTAMP(i) do { const u64 now_ = __builtin_amdgcn_s_memtime(); cst_acc[i] += now_ - cst_last; cst_last = now_; } while (0)
#else
#define SK_CSTAMP(i) do { } while (0)
#endif
#ifdef SK_CENSUS_STAMPS
#define SK_STAMP(i) do { const u64 now_ = __builtin_amdgcn_s_memtime(); st_acc[i] += now_ - st_last; st_last = now_; } while (0)
#else
#define SK_STAMP(i) do { } while (0)
#endif

struct CensusArgs {
	const uint8_t *bc;
	int bc_stride;
	int L;
	int64_t n;
	const int32_t *assign;    // nullable: count row r only when assign[r] == SK_ASSIGN_NONE
	int64_t row_base;
	CensusTab t;
	u64 *stats;
	CensusSpill sp;
	int long_way_only;        // SK_CENSUS_LONG_WAY_ONLY=1: no row is counted by the fast look at the front table (its ordered, compiler-
	                          // invisible reads of an entry another wave may be publishing): every row queues up for the long way, whose
	                          // reads are waited for one entry at a time.  A fallback should a compiler or a part ever break the
	                          // protocol the fast look leans on (tests/test_census_isa.py guards the compiler side); ~3 x slower
};

constexpr int kCensusWaves = 16;          // waves per workgroup (one per CU): they share the front table
constexpr int kCensusMaxStride = 64;      // a row's span is at most 9 dwords (census_load_rows)

__device__ __forceinline__ void census_wave_fence()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kCensusMaxSub = 2;          // 64-row tiles per wave and step (three and four timed the same and needed scratch for the longest strings)
constexpr int kCensusStepBytes = 5120;    // most bytes per wave and step before the cap on R (census_add)
constexpr int kCensusQueue = 128;         // flush queue entries (16 B key + 4 B row): fewer than 64 left over + one tile's 64

template <int R, int NW> struct CensusRowRegs { u32 g[R][NW + 1]; int32_t code[kCensusMaxSub] = {0, 0}; };

// The input as raw-buffer descriptors over the whole launch (the rows, and the assignment codes when there are any): what
// lies beyond the matrix is clipped by the descriptor (a load returns zeros there, dword by dword).
struct CensusStreams { __amdgpu_buffer_rsrc_t bc, assign; };
__device__ __forceinline__ CensusStreams census_streams(const CensusArgs &a)
{
	CensusStreams cs;
	const u32 total = (u32)(a.n * (int64_t)a.bc_stride);
	cs.bc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.bc), 0, (int)((total + 3u) & ~3u), 0x00020000);
	cs.assign = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(a.assign), 0, a.assign ? (int)(a.n * 4) : 0, 0x00020000);
	return cs;
}
// A lane fetches the dwords ITS rows of the step lie in (row j * 64 + lane for j < R), at any alignment: the NW dwords from
// the one that holds the row's first byte, and the one behind them (which a row that begins late in its first dword reaches).
// No image of the step in LDS: written once and read back at the rows' pitch (every read of a 17-byte-pitch walk a conflicted
// one) it kept the LDS pipe, which the table probes need, busier than anything else in this kernel.
// No branches around the loads — a load inside a conditional block is waited for at the end of that block, which serialises
// them and the counting behind them.  (The assignment codes, when there are any, are asked for BEFORE the rows: the one
// conditional block with loads in it then lies in front of the unconditional ones, and the wait for the rows counts the
// same loads on both paths.)
template <int R, int NW> __device__ __forceinline__ void census_load_rows(const CensusStreams &cs, bool has_assign, int t, int step_bytes, const u32 (&off)[R], int lane,
                                                                          CensusRowRegs<R, NW> &rg)
{
	typedef u32 u32x2_t __attribute__((ext_vector_type(2)));
	typedef u32 u32x3_t __attribute__((ext_vector_type(3)));
	if (has_assign) {
		const int r0 = (t * R * 64 + lane) * 4;
#pragma unroll
		for (int j = 0; j < kCensusMaxSub; j++)
			rg.code[j] = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(cs.assign, j < R ? r0 + j * 256 : 0x7ffffff0, 0, 0);
	}
	const u32 base = (u32)t * (u32)step_bytes;
#pragma unroll
	for (int j = 0; j < R; j++) {
		const int o = (int)(base + off[j]);
		if constexpr (NW == 2) {
			const u32x3_t v = __builtin_amdgcn_raw_buffer_load_b96(cs.bc, o, 0, 0);
			rg.g[j][0] = v[0]; rg.g[j][1] = v[1]; rg.g[j][2] = v[2];
		} else if constexpr (NW == 5) {
			const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(cs.bc, o, 0, 0);
			const u32x2_t w = __builtin_amdgcn_raw_buffer_load_b64(cs.bc, o + 16, 0, 0);
			rg.g[j][0] = v[0]; rg.g[j][1] = v[1]; rg.g[j][2] = v[2]; rg.g[j][3] = v[3]; rg.g[j][4] = w[0]; rg.g[j][5] = w[1];
		} else {
			static_assert(NW == 8, "strings of at most 8, 20 or 32 bytes");
			const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(cs.bc, o, 0, 0), w = __builtin_amdgcn_raw_buffer_load_b128(cs.bc, o + 16, 0, 0);
			rg.g[j][0] = v[0]; rg.g[j][1] = v[1]; rg.g[j][2] = v[2]; rg.g[j][3] = v[3];
			rg.g[j][4] = w[0]; rg.g[j][5] = w[1]; rg.g[j][6] = w[2]; rg.g[j][7] = w[3];
			rg.g[j][8] = __builtin_amdgcn_raw_buffer_load_b32(cs.bc, o + 32, 0, 0);
		}
	}
}

// One row per lane, R 64-row tiles per wave and step.  A wave keeps the next step's rows in registers (census_load_rows)
// while it counts the current one; what it shares with nobody — its queue of rows that take the long way — is its own
// slot of LDS, so the only workgroup barriers are the two around the loop.
// Rows are counted in the workgroup's front table (above); a string that finds neither of its two places free goes
// straight to HBM as a packed key.  The table is merged into HBM when the workgroup is done.
// SPILL: such keys are not inserted but appended to the workgroup's region of the spill arrays (unconditional, clipped
// stores: nothing waits for them), counted per bucket, and so is the table at the end; census_scatter_kernel /
// census_combine_kernel take them from there.
template <int R, int NW, bool SPILL> __global__ __launch_bounds__(kCensusWaves * 64, 1) void census_kernel(const CensusArgs a, const int tile_slot, const int front_entries)
{
	constexpr int CH = FrontShape<NW>::CH;
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int nwave = blockDim.x >> 6;
	const u32 front_bytes = ((u32)front_entries * (u32)FrontShape<NW>::kEntryBytes + 15u) & ~15u;
	u32 *front = reinterpret_cast<u32 *>(census_smem);
	FrontTable ft;                                                     // LDS addresses, for the asm loads and stores
	ft.entries = (u32)front_entries;
	ft.b = (u32)(uintptr_t)census_smem;
	ft.a = ft.b + ft.entries * 16u;
	ft.c = ft.a + (CH > 2 ? ft.entries * 16u : 0u);
	ft.cnt = ft.b + ft.entries * (CH * 16u);
	u32 *const fb = front;                                             // the same arrays as pointers: chunk B ...
	u32 *const fa = front + front_entries * 4;                        // ... A ...
	u32 *const fc = fa + (CH > 2 ? front_entries * 4 : 0);            // ... C ...
	u32 *const fcnt = front + front_entries * (CH * 4);               // ... and the counts
	// The wave's slot: its QUEUE of rows that take the long way — a row that misses leaves its NW masked dwords and its index
	// there (qx: 4 NW bytes per entry, qr: the row's index within the launch).  Rows that miss wait until 64 of them are
	// together: a pass of the long way costs its ~300 instructions whether 6 lanes or 64 have a row (a clean run missed 6-15
	// rows per step and paid a pass for them in every step: 23-27 % of a wave's lifetime,
	// profiles/r04_census_stamps_before.txt).  What is left over at the end of a step — fewer than 64 entries — moves to the
	// front of the queue and goes first in the next step's passes.  Capacity: 64 left over + the step's R x 64 rows.
	uint8_t *slot = census_smem + front_bytes + (size_t)wave * tile_slot;
	u32 *lh = reinterpret_cast<u32 *>(census_smem + front_bytes + (size_t)nwave * tile_slot + 64);      // SPILL: the workgroup's records per bucket (its cursors into the bucket regions)
	constexpr int kQueueCap = 64 + R * 64;
	u32 *const qr = reinterpret_cast<u32 *>(slot);
	u32 *const qx = qr + kQueueCap;
	u32 n_carry = 0u;                                                  // wave-uniform: entries left over from the step before
	// Once the table is full a string that is not in it finds both its places taken, every time: after kFullAfter rows in a
	// row (of this wave) for which that was so, the long way no longer looks — it builds the key and parks the row.  (A string
	// that another lane put into the table a moment ago is then parked instead of counted there: the sums are the same.)
	constexpr u32 kFullAfter = 256u;
	u32 fails_in_a_row = 0u;                                           // wave-uniform
	for (int i = tid; i < (int)(front_bytes / 16u); i += blockDim.x) reinterpret_cast<uint4 *>(front)[i] = make_uint4(0u, 0u, 0u, 0u);
	if (SPILL) for (int i = tid; i <= kSpillBuckets; i += blockDim.x) lh[i] = 0u;
	// The workgroup's steps are dealt to its waves as they come free (an LDS counter): the waves of a SIMD do not advance at
	// the same rate, and with a fixed deal the workgroup waited 9-16 % of its lifetime at the barrier behind the loop for its
	// slowest wave (profiles/r04_census_stamps_before.txt).  Step c of workgroup b is tile-step ((c / nwave) * grid + b) * nwave +
	// c % nwave — the same set of steps as the fixed deal, so what the chip reads at any moment is still one advancing window.
	u32 *step_ctr = lh + kSpillBuckets + 1;
	if (tid == 0) *step_ctr = (u32)nwave;                              // the first nwave steps are the waves' own
#ifdef SK_CENSUS_STAMPS
	u64 st_acc[kStampSlots] = {};
	u64 st_last = __builtin_amdgcn_s_memtime();
#endif
	__syncthreads();
	SK_STAMP(0);                                                       // tables cleared
	__amdgpu_buffer_rsrc_t sp_key;                                     // the workgroup's kSpillBuckets regions of a.sp.cap records each
	if (SPILL) sp_key = __builtin_amdgcn_make_buffer_rsrc(a.sp.key + (size_t)blockIdx.x * kSpillBuckets * a.sp.cap, 0, (int)(kSpillBuckets * a.sp.cap * 16u), 0x00020000);

	const int stride = a.bc_stride;
	const int step_bytes = R * 64 * stride;
	u32 kms[NW];                                                       // the characters of a row's dword q that lie before L
#pragma unroll
	for (int q = 0; q < NW; q++) {
		const int keep = a.L - 4 * q;
		kms[q] = keep >= 4 ? 0xFFFFFFFFu : (keep <= 0 ? 0u : (1u << (8 * keep)) - 1u);
	}
	const int nsteps = (int)((a.n + (int64_t)R * 64 - 1) / ((int64_t)R * 64));      // (a launch is fewer than 2^31 bytes: census_add)
	const int n32 = (int)a.n;
	const int nwave_lg = 31 - __builtin_clz((u32)nwave);              // (the launch is kCensusWaves = 16 waves: a power of two)
	auto step_of = [&](u32 c) -> int { return (int)((((c >> nwave_lg) * gridDim.x + blockIdx.x) << nwave_lg) + (c & ((u32)nwave - 1u))); };
	const CensusStreams streams = census_streams(a);
	u32 off[R], shf[R];                                                // where the lane's row j begins within a step: its dword, and the byte in it
#pragma unroll
	for (int j = 0; j < R; j++) {
		const u32 o = (u32)((j * 64 + lane) * stride);
		off[j] = o & ~3u;
		shf[j] = o & 3u;
	}
	u32 claimed = 0, counted = 0, rejected = 0, overflow = 0;
	CensusRowRegs<R, NW> rg;
	// (the flush queue of the launches that insert by themselves lies behind the first 64 queue entries: dead once the passes ran)
	uint4 *qkey = reinterpret_cast<uint4 *>(slot + ((kQueueCap * 4 + 64 * 4 * NW + 15) & ~15));
	u32 *qrel = reinterpret_cast<u32 *>(reinterpret_cast<uint8_t *>(qkey) + kCensusQueue * 16);
	// one more row for the entry at LDS address ea, whose first row was `first` when it was read
	auto front_count = [&](u32 e, u32 first, u32 r) {
		__hip_atomic_fetch_add(&fcnt[e], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		if (r < first) __hip_atomic_fetch_min(&fb[e * 4u + 2u], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (rows come in roughly ascending order: rare)
	};
	// ---- SPILL: a record on its way out: its place in the workgroup's region of bucket b from the bucket's cursor, one store.  A
	// region is twice a workgroup's share of a launch in which every row is spilled: a record that finds its region full
	// (thousands of rows of ONE key that came too late for the front table) is inserted from here.
	auto record_put = [&](bool has, u32 b, const u32x4_t &rec) {
		u32 at = 0xffffffffu;
		if (has) at = atomicAdd(&lh[b], 1u);
		const bool fits = has && at < a.sp.cap;
		__builtin_amdgcn_raw_buffer_store_b128(rec, sp_key, fits ? (int)((b * a.sp.cap + at) * 16u) : 0x7ffffff0, 0, 0);      // (beyond the regions: dropped by the descriptor)
		if (has && !fits) {
			const u64 klo = 0xF0000000ull | ((u64)rec[0] << 32), khi = (u64)rec[1] | ((u64)rec[2] << 32);
			const u32 cnt = census_rec_count(rec[3]);
			if (!census_insert(a.t, klo, khi, (u64)cnt, ~(u64)(a.row_base + census_rec_row(rec[3])), claimed)) overflow += cnt;
		}
	};
	int t = (int)blockIdx.x * nwave + wave;
	census_load_rows<R, NW>(streams, a.assign != nullptr, t, step_bytes, off, lane, rg);
	for (;;) {
		const bool drain = t >= nsteps;                                // no step left: one more turn for the rows still waiting in the queue
		if (drain && n_carry == 0u) break;
		u32 next_c = 0u;                                               // asked for before the rows are taken out of their registers: the answer is there when they are
		if (lane == 0) next_c = atomicAdd(step_ctr, 1u);
		u32 xs[R][NW];                                                 // the step's rows: their first L bytes as dwords (a drain turn's are zeros nobody looks at)
#pragma unroll
		for (int j = 0; j < R; j++)
#pragma unroll
			for (int q = 0; q < NW; q++) xs[j][q] = __builtin_amdgcn_alignbyte(rg.g[j][q + 1], rg.g[j][q], shf[j]) & kms[q];
		SK_STAMP(1);                                                   // the step's loads waited for, rows aligned
		// which of the step's rows are counted, worked out BEFORE the next step's loads are issued: nothing below may
		// wait for a register that a load of this or an earlier step wrote, or it waits for the new loads as well
		u32 take = (1u << kCensusMaxSub) - 1u;
		if (a.assign != nullptr) {
			take = 0u;
#pragma unroll
			for (int j = 0; j < kCensusMaxSub; j++) take |= (rg.code[j] == kAssignNone ? 1u : 0u) << j;
		}
		census_wave_fence();
#ifdef SK_CENSUS_STATIC_STEPS                                              // (A/B builds: the fixed deal)
		const int tn = t + (int)gridDim.x * nwave;
#else
		const int tn = step_of((u32)__builtin_amdgcn_readfirstlane((int)next_c));
#endif
		census_load_rows<R, NW>(streams, a.assign != nullptr, tn, step_bytes, off, lane, rg);       // in flight while this step is counted
		SK_STAMP(2);                                                   // fence, next step's loads issued
		// Counting touches LDS only.  Keys the table had no room for are parked in registers (one per lane and pass) and go to
		// HBM after the step's last pass: any global memory operation in between would make the compiler wait for the loads
		// just issued (vmcnt is one in-order counter).
		u64 pklo[R], pkhi[R];
		u32 ph[R], prid[R];
		u32 parked = 0u;
		// ---- rows whose bytes the table knows are counted there; the others queue up for the long way ------------------------
		u32 qn = 0u;
		if (!drain) {
			// all of the step's row reads first, then all of its table reads, then the counting: three LDS round trips per
			// step instead of three per 64 rows
			u32 hs[R], en[R], first[R];
			bool hit[R];
			{
				FrontWords<NW> fw[R];
#pragma unroll
				for (int j = 0; j < R; j++) {
					hs[j] = front_hash<NW>(xs[j]);
					en[j] = front_place(hs[j], 0, ft.entries);
				}
#pragma unroll
				for (int j = 0; j < R; j++) front_issue<NW>(ft, en[j], fw[j]);
#pragma unroll
				for (int j = 0; j < R; j++) front_wait<NW>(fw[j]);
#pragma unroll
				for (int j = 0; j < R; j++) {
					hit[j] = !a.long_way_only && front_match<NW>(fw[j], xs[j]);
					first[j] = fw[j].first();
				}
			}
			{	// the strings' other places, for the lanes that need them (one more round trip for the whole step)
				FrontWords<NW> f2[R];
				u32 e2[R];
#pragma unroll
				for (int j = 0; j < R; j++) {
					e2[j] = front_place(hs[j], 1, ft.entries);
					front_issue<NW>(ft, hit[j] ? 0u : e2[j], f2[j]);              // (no branch around asm loads; lanes that need nothing read ONE address: a broadcast)
				}
#pragma unroll
				for (int j = 0; j < R; j++) front_wait<NW>(f2[j]);
#pragma unroll
				for (int j = 0; j < R; j++)
					if (!hit[j] && !a.long_way_only && front_match<NW>(f2[j], xs[j])) { hit[j] = true; en[j] = e2[j]; first[j] = f2[j].first(); }
			}
#pragma unroll
			for (int j = 0; j < R; j++) {
				const int r = (t * R + j) * 64 + lane;
				const bool want = r < n32 && ((take >> j) & 1u);
				if (want && hit[j]) {
					front_count(en[j], first[j], (u32)r);
					counted++;
				}
				const bool miss = want && !hit[j];
				const u64 bal = __ballot(miss);
				if (miss) {
					const u32 at = n_carry + qn + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
					qr[at] = (u32)r;
#pragma unroll
					for (int q = 0; q < NW; q++) qx[at * NW + q] = xs[j][q];
				}
				qn += (u32)__popcll(bal);
			}
			census_wave_fence();
		}
		SK_STAMP(3);                                                   // rows the table knows
		// ---- the long way, 64 queued rows at a time: check the string, look for it again (another lane may have put it there
		// meanwhile), else claim one of its places, else build its key and park it -----------------------------------------------
		// (the lane index as a value the compiler cannot see through: what the long way computes from it — queue
		// addresses — is then worked out here, per step, instead of being kept in registers across the whole loop; the kernel sits
		// at its 128 registers, and one more live value was a scratch slot whose reload waited for the step's prefetch)
		u32 lane_l = (u32)lane;
		asm volatile("" : "+v"(lane_l));
		const u32 q_total = n_carry + qn;                              // the carried rows first, then the step's
		const u32 n_pass = drain ? (q_total + 63u) >> 6 : q_total >> 6;    // whole passes only, but for the last turn
#pragma unroll
		for (int j = 0; j < R; j++) {
			ph[j] = 0u;
			prid[j] = 0u;
			pklo[j] = pkhi[j] = 0ull;                                 // (only read where `parked` says so: these cost nothing)
			if ((u32)j < n_pass) {
				const u32 qi = (u32)(j * 64) + lane_l;
				const bool have = qi < q_total;
				const u32 qe = have ? qi : 0u;                             // (lanes without a row read entry 0: nothing of theirs is used)
				const u32 r = qr[qe];                                      // the row's index within the launch
				prid[j] = r;
				u32 xs[NW];
#pragma unroll
				for (int q = 0; q < NW; q++) xs[q] = qx[qe * NW + q];
				{	// a NUL before L ends the barcode and what follows it is padding: zeroed, only when some row of the pass has one
					u32 z = 0u;
#pragma unroll
					for (int q = 0; q < NW; q++) z |= ~((((xs[q] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | xs[q])) & 0x80808080u & kms[q];
					if (__any(z != 0u)) (void)census_canonical<NW>(xs, kms);
				}
				u32 bad;
				census_key_of<NW>(xs, pklo[j], pkhi[j], bad);
				if (have) {
					if (bad != 0u) rejected++;                                 // a byte outside the alphabet before the barcode's end
					else {
						counted++;
						const u32 h = front_hash<NW>(xs);
						bool done = false;
#pragma unroll 1
						for (int pl = 0; pl < 2 && !done && fails_in_a_row < kFullAfter; pl++) {
							const u32 e = front_place(h, pl, ft.entries);
#pragma unroll 1
							for (int spin = 0; spin < 256; spin++) {               // (a BUSY entry is published by its writer a few LDS operations later)
								FrontWords<NW> f;
								front_issue<NW>(ft, e, f);
								front_wait<NW>(f);
								const u32 st = f.state();
								if (st == kFrontReady) {
									if (front_match<NW>(f, xs)) { front_count(e, f.first(), r); done = true; }
									break;                                             // another string's: on to the other place
								}
								if (st == kFrontEmpty && atomicCAS(&fb[e * 4u + 3u], kFrontEmpty, kFrontBusy) == kFrontEmpty) {
									front_publish<NW>(ft, e, xs, r);
									__hip_atomic_fetch_add(&fcnt[e], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
									done = true;
									break;
								}
							}
						}
						if (!done) { ph[j] = census_hash(pklo[j], pkhi[j]); parked |= 1u << j; }
					}
				}
				if (fails_in_a_row < kFullAfter) {                             // (uniform) how did the pass's rows fare?
					const u32 np = (u32)__popcll(__ballot(((parked >> j) & 1u) != 0u));
					const u32 nr = (u32)__popcll(__ballot(have && bad == 0u));
					fails_in_a_row = np == nr ? fails_in_a_row + np : 0u;      // a pass in which any row found or claimed a place starts the count again
				}
			}
		}
		{	// what is left — fewer than 64 entries — moves to the front of the queue (read by every lane that has one, then written)
			const u32 done = n_pass << 6;
			const u32 left = q_total > done ? q_total - done : 0u;
			if (done != 0u && left != 0u) {
				census_wave_fence();
				const bool mine = lane_l < left;
				const u32 from = mine ? done + lane_l : 0u;
				const u32 r = qr[from];
				u32 ys[NW];
#pragma unroll
				for (int q = 0; q < NW; q++) ys[q] = qx[from * NW + q];
				census_wave_fence();
				if (mine) {
					qr[lane_l] = r;
#pragma unroll
					for (int q = 0; q < NW; q++) qx[lane_l * NW + q] = ys[q];
				}
			}
			n_carry = left;
		}
		census_wave_fence();
		SK_STAMP(4);                                                   // the long way
		// The parked keys, packed densely through the dead part of the queue so that one insert serves up to 64 of them: its
		// dependent round trips are paid per call, not per key.  (Fetching the slots a step ahead of the insert was tried:
		// no gain — what bounds this leg is the rate of scattered atomics, not their latency.)
		if (SPILL) {
			if (__any(parked != 0u)) {                                     // (stores inside a branch are fine: nothing waits for them)
#pragma unroll
				for (int j = 0; j < R; j++) {
					const bool has = ((parked >> j) & 1u) != 0u;
					const u32x4_t kq = {(u32)(pklo[j] >> 32), (u32)pkhi[j], (u32)(pkhi[j] >> 32), prid[j]};
					record_put(has, census_bucket(ph[j]), kq);
				}
			}
		} else if (__any(parked != 0u)) {
			int qn = 0;
#pragma unroll
			for (int j = 0; j < R; j++) {
				const bool has = ((parked >> j) & 1u) != 0u;
				const u64 bal = __ballot(has);
				if (has) {
					const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
					qkey[pos] = make_uint4((u32)pklo[j], (u32)(pklo[j] >> 32), (u32)pkhi[j], (u32)(pkhi[j] >> 32));
					qrel[pos] = prid[j];
				}
				qn += __popcll(bal);
				while (qn >= 64 || (j == R - 1 && qn > 0)) {
					census_wave_fence();
					const int cnt = qn < 64 ? qn : 64;
					if (lane < cnt) {
						const uint4 kq = qkey[qn - cnt + lane];
						const u32 rel = qrel[qn - cnt + lane];
						const u64 klo = (u64)kq.x | ((u64)kq.y << 32), khi = (u64)kq.z | ((u64)kq.w << 32);
						const u64 first_inv = ~(u64)(a.row_base + rel);
						if (!census_insert(a.t, klo, khi, 1ull, first_inv, claimed)) overflow++;
					}
					qn -= cnt;
					census_wave_fence();
				}
			}
		}
		census_wave_fence();
		SK_STAMP(5);                                                   // parked keys: spill records / inserts
		t = tn;
	}
	SK_STAMP(6);                                                       // (loop exit)
	__syncthreads();
	SK_STAMP(7);                                                       // waiting for the workgroup's other waves
	// ---- the table leaves the workgroup: a key is built per ENTRY ----------------------------------------------------------
	// Into HBM directly (small launches), every workgroup starting somewhere else so that the strings all of them hold are not
	// hit by all of them at the same moment.  With the partition path behind it (SPILL) the table leaves as RECORDS instead:
	// a noisy run's table is full of strings this workgroup saw a few times — and 255 other workgroups hold the same ones:
	// inserting them from here was half a million CAS / adds at the end of the launch with nothing to hide behind, and the
	// sheet's own barcodes were added to by every workgroup at once (21-25 % of a wave's lifetime,
	// profiles/r04_census_stamps_before.txt).  An entry counted c times is the non-zero base-32 digits of c — at most four
	// 16-byte stores, one for most — each with the entry's first row; the combine pass adds them up with the other workgroups'
	// before anything touches HBM.  Such records stand for rows that were NOT spilled, so the workgroup's region (one record
	// per row) has room for them.
	const bool as_records = SPILL && a.sp.merge_copies != 0u;
	for (int i0 = tid; i0 < ((front_entries + 63) & ~63); i0 += blockDim.x) {
		int i = i0 + (as_records ? 0 : (int)blockIdx.x * 67);
		if (i >= front_entries) i -= front_entries * (i / front_entries);
		const bool in = i0 < front_entries;
		const u32 ei = in ? (u32)i : 0u;
		u32 xs[NW];
#pragma unroll
		for (int q = 0; q < NW; q++) xs[q] = q < 4 ? fa[ei * 4u + (q & 3)] : (CH > 2 ? fc[ei * 4u + (q & 3)] : fb[ei * 4u + (q & 3)]);
		const u32 sc = fcnt[ei], first = fb[ei * 4u + 2u];
		const bool live = in && fb[ei * 4u + 3u] == kFrontReady;
		u64 sk, khi;
		u32 bad;
		census_key_of<NW>(xs, sk, khi, bad);
		const u64 first_inv = ~(u64)(a.row_base + first);
		bool rec = false;
		if (SPILL) {
			rec = live && as_records && sc < (1u << 20);                       // (a count beyond four digits is inserted below: never in a launch of 2^25 rows over 256 workgroups)
			u32 digit[4];
#pragma unroll
			for (int v = 0; v < 4; v++) digit[v] = rec ? (sc >> (5 * v)) & 31u : 0u;
			const u32 bkt = census_bucket(census_hash(sk, khi));
#pragma unroll
			for (int v = 0; v < 4; v++) {
				const u32x4_t kq = {(u32)(sk >> 32), (u32)khi, (u32)(khi >> 32), first | ((u32)v << kRecRowBits) | ((digit[v] - 1u) << 27)};
				record_put(digit[v] != 0u, bkt, kq);
			}
		}
		if (live && !rec && !census_insert(a.t, sk, khi, (u64)sc, first_inv, claimed)) overflow += sc;
	}
	if (SPILL) {
		__syncthreads();
		if (tid < 64) {                                                    // what lies in the workgroup's regions (a cursor beyond the region: the records behind it were inserted)
			u32 sum = 0u;
			for (int b = tid; b < kSpillBuckets; b += 64) {
				const u32 v = lh[b] < a.sp.cap ? lh[b] : a.sp.cap;
				a.sp.hist[(size_t)b * a.sp.grid + blockIdx.x] = v;
				if (v != 0u) atomicAdd(&a.sp.btot[b], v);
				sum += v;
			}
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
			if (tid == 0) a.sp.wg_count[blockIdx.x] = sum;
		}
	}
	SK_STAMP(8);                                                       // front table merged into HBM
	// one atomic per WORKGROUP and statistic: the waves of a launch end together, and 4 096 additions to one address are
	// served one after the other (34 us of an 84 us launch of 1 M rows when every wave sent its own)
	for (int o = 32; o > 0; o >>= 1) {
		claimed += __shfl_xor(claimed, o);
		counted += __shfl_xor(counted, o);
		rejected += __shfl_xor(rejected, o);
		overflow += __shfl_xor(overflow, o);
	}
	__syncthreads();                                                   // the front table is dead: its first words take the sums
	u32 *wg_stats = front;
	if (tid < kCensusStats) wg_stats[tid] = 0u;
	__syncthreads();
	if (lane == 0) {
		if (claimed) atomicAdd(&wg_stats[0], claimed);
		if (counted) atomicAdd(&wg_stats[1], counted);
		if (rejected) atomicAdd(&wg_stats[2], rejected);
		if (overflow) atomicAdd(&wg_stats[3], overflow);
	}
	__syncthreads();
	if (SPILL) { if (tid < kCensusStats) a.sp.wg_stats[blockIdx.x * kCensusStats + tid] = wg_stats[tid]; }      // census_scan_kernel adds them up
	else if (tid < kCensusStats && wg_stats[tid] != 0u) atomicAdd(&a.stats[tid], (u64)wg_stats[tid]);
#ifdef SK_CENSUS_STAMPS
	SK_STAMP(9);
	if (lane == 0 && blockIdx.x * nwave + wave < 8192)
		for (int i = 0; i < kStampSlots; i++) g_census_stamps[(blockIdx.x * nwave + wave) * kStampSlots + i] = st_acc[i];
#endif
}

// ---- the partition path -------------------------------------------------------------------------------------------
// exclusive prefix sum of one value per thread over the workgroup (at most 16 waves): shuffles inside a wave, the wave sums
// through LDS (ws: 17 words) — three barriers where the textbook loop over LDS has two per doubling
__device__ __forceinline__ u32 census_block_exclusive(u32 mine, u32 *ws, u32 *total)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
	u32 v = mine;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const u32 t = __shfl_up(v, o);
		if (lane >= o) v += t;
	}
	__syncthreads();                                                   // ws may still be read from an earlier call
	if (lane == 63) ws[wave] = v;
	__syncthreads();
	if (wave == 0) {
		const u32 w = lane < nw ? ws[lane] : 0u;
		u32 x = w;
#pragma unroll
		for (int o = 1; o < 16; o <<= 1) {
			const u32 t = __shfl_up(x, o);
			if (lane >= o) x += t;
		}
		if (lane < nw) ws[lane] = x - w;
		if (lane == nw - 1) ws[16] = x;
	}
	__syncthreads();
	if (total) *total = ws[16];
	return v - mine + ws[wave];
}

// how many records the front kernel's workgroups wrote (every thread of the workgroup gets the sum)
__device__ __forceinline__ u32 census_spill_total(const CensusSpill &sp, u32 *red)
{
	u32 v = 0u;
	for (u32 g = threadIdx.x; g < sp.grid; g += blockDim.x) v += sp.wg_count[g];
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
	__syncthreads();
	u32 tot = 0u;
	for (u32 w = 0; w < (blockDim.x + 63) / 64; w++) tot += red[w];
	__syncthreads();
	return tot;
}

// claimed / overflow of a workgroup's threads -> stats[0] / stats[3], one atomic per workgroup and statistic (red: two words
// of LDS; every thread of the workgroup calls this)
__device__ __forceinline__ void census_add_stats(u64 *stats, u32 claimed, u32 overflow, u32 *red)
{
	for (int o = 32; o > 0; o >>= 1) {
		claimed += __shfl_xor(claimed, o);
		overflow += __shfl_xor(overflow, o);
	}
	__syncthreads();
	if (threadIdx.x < 2) red[threadIdx.x] = 0u;
	__syncthreads();
	if ((threadIdx.x & 63) == 0) {
		if (claimed) atomicAdd(&red[0], claimed);
		if (overflow) atomicAdd(&red[1], overflow);
	}
	__syncthreads();
	if (threadIdx.x == 0 && red[0] != 0u) atomicAdd(&stats[0], (u64)red[0]);
	if (threadIdx.x == 1 && red[1] != 0u) atomicAdd(&stats[3], (u64)red[1]);
}

// Combine the buckets in LDS tables and insert every distinct key once.  All keys of a bucket hash into ONE region of the
// HBM table (1/128 of it); its records lie in the front kernel's workgroups' regions of that bucket, one behind the other in
// workgroup order.  A work item is at most kCombineChunk consecutive records of that sequence — a bucket is split among
// workgroups, since what they add is additive — and the workgroups take items from a counter until none is left.
constexpr int kCombineThreads = 512;
#ifndef SK_COMBINE_CHUNK
#define SK_COMBINE_CHUNK 16384
#endif
constexpr int kCombineChunk = SK_COMBINE_CHUNK;
__global__ __launch_bounds__(kCombineThreads, 4) void census_combine_kernel(const CensusArgs a)
{
	__shared__ LdsTable lt_s;
	__shared__ u32 cpre[kSpillBuckets + 1];                            // items before bucket b
	__shared__ u32 gpre[kSpillMaxGrid + 1];                            // records of the item's bucket before workgroup g's
	__shared__ u32 red[kCombineThreads / 64];
	__shared__ u32 ws[17];
	__shared__ u32 item_s, nlist;
	__shared__ uint16_t list[kLdsSlots];
	LdsTable *lt = &lt_s;
	const int tid = threadIdx.x;
	const u32 grid = a.sp.grid;
#ifdef SK_CENSUS_STAMPS
	u64 cst_acc[kStampSlots] = {};
	u64 cst_last = __builtin_amdgcn_s_memtime();
#endif
	if (blockIdx.x == 0) {                                             // the front kernel's statistics: 4 atomics per launch instead of 4 per workgroup
		__shared__ u32 ssum[kCensusStats];
		if (tid < kCensusStats) ssum[tid] = 0u;
		__syncthreads();
		u32 v = 0u;
		for (u32 i = tid; i < grid * kCensusStats; i += blockDim.x) v += a.sp.wg_stats[i];      // (blockDim is a multiple of kCensusStats: thread x sums statistic x % 4)
		for (int o = 32; o >= kCensusStats; o >>= 1) v += __shfl_xor(v, o);
		if ((tid & 63) < kCensusStats && v != 0u) atomicAdd(&ssum[tid & 3], v);
		__syncthreads();
		if (tid < kCensusStats && ssum[tid] != 0u) atomicAdd(&a.stats[tid], (u64)ssum[tid]);
	}
	if (census_spill_total(a.sp, red) > a.sp.direct_above) return;
	{	// items per bucket, scanned (a bucket per thread)
		static_assert(kSpillBuckets <= kCombineThreads, "one bucket per thread");
		const u32 c = tid < kSpillBuckets ? (a.sp.btot[tid] + kCombineChunk - 1) / kCombineChunk : 0u;
		u32 total;
		const u32 before = census_block_exclusive(c, ws, &total);
		if (tid < kSpillBuckets) cpre[tid] = before;
		if (tid == 0) cpre[kSpillBuckets] = total;
		__syncthreads();
	}
	const u32 items = cpre[kSpillBuckets];
	SK_CSTAMP(0);                                                      // statistics, records counted, items scanned
	u32 claimed = 0, overflow = 0;
	int b_have = -1;                                                   // the bucket gpre was made for
	for (;;) {
		__syncthreads();
		// (items dealt round-robin instead — no returning atomic, two round trips less per workgroup — were measured: the clean
		// launch 0.24 -> 0.23 ms, the noisy one 0.31 -> 0.34)
		if (tid == 0) item_s = atomicAdd(a.sp.work, 1u);
		for (int i = tid; i < (int)(sizeof(LdsTable) / 8); i += blockDim.x) reinterpret_cast<u64 *>(lt)[i] = 0ull;
		__syncthreads();
		const u32 item = item_s;
		SK_CSTAMP(1);                                                  // an item taken, the table cleared
		if (item >= items) break;
		int b = 0;                                                     // the last bucket with cpre[b] <= item
		for (int o = kSpillBuckets / 2; o > 0; o >>= 1) if (cpre[b + o] <= item) b += o;
		if (b != b_have) {                                             // (a workgroup's items come in ascending order: mostly the bucket it has)
			u32 carry = 0u;
			for (u32 g0 = 0; g0 < grid; g0 += blockDim.x) {
				const u32 g = g0 + tid;
				const u32 v = g < grid ? a.sp.hist[(size_t)b * grid + g] : 0u;          // (bucket-major: a bucket's region sizes are one contiguous piece)
				u32 total;
				const u32 before = census_block_exclusive(v, ws, &total);
				if (g < grid) gpre[g] = carry + before;
				carry += total;
			}
			if (tid == 0) gpre[grid] = carry;
			__syncthreads();
			b_have = b;
		}
		const u32 lo = (item - cpre[b]) * kCombineChunk;
		const u32 hi = min(gpre[grid], lo + kCombineChunk);
		// record i of the bucket's sequence lies in the region of the workgroup g with gpre[g] <= i < gpre[g + 1]: found once by
		// bisection, then g only moves forward (a thread's records are blockDim apart)
		u32 gcur = 0u;
		{
			const u32 i = lo + tid;
			u32 l = 0u, r = grid;                                      // the last g with gpre[g] <= i
			while (r - l > 1u) { const u32 m = (l + r) >> 1; if (gpre[m] <= i) l = m; else r = m; }
			gcur = l;
		}
		SK_CSTAMP(2);                                                  // the bucket's regions scanned, the thread's first record found
		const uint4 *const base = a.sp.key + (size_t)b * a.sp.cap;
		const size_t gpitch = (size_t)kSpillBuckets * a.sp.cap;
		// (four records of a thread are counted while its next four are on their way: with one load per trip the loop was a
		// chain of memory latencies — 32 of them for a full item —, with four loads per trip still one per trip)
		constexpr int kAhead = 4;
		uint4 kk[kAhead], kn[kAhead];
		auto fetch = [&](u32 i0, uint4 (&dst)[kAhead]) {
#pragma unroll
			for (int q = 0; q < kAhead; q++) {
				const u32 i = i0 + (u32)q * blockDim.x;
				if (i < hi) {
					while (gpre[gcur + 1] <= i) gcur++;
					dst[q] = base[(size_t)gcur * gpitch + (i - gpre[gcur])];
				} else dst[q] = make_uint4(0u, 0u, 0u, 0u);
			}
		};
		fetch(lo + tid, kk);
		for (u32 i0 = lo + tid; i0 < hi; i0 += kAhead * blockDim.x) {
			fetch(i0 + kAhead * blockDim.x, kn);
			// (a key the table has no room for goes to HBM with the others of the thread's four records)
			bool dir[kAhead], have[kAhead];
			u64 klo[kAhead], khi[kAhead], first_inv[kAhead];
			u32 cnt[kAhead], hh[kAhead];
#pragma unroll
			for (int q = 0; q < kAhead; q++) {
				const uint4 k = kk[q];
				have[q] = i0 + (u32)q * blockDim.x < hi;
				klo[q] = 0xF0000000ull | ((u64)k.x << 32);
				khi[q] = (u64)k.y | ((u64)k.z << 32);
				first_inv[q] = ~(u64)(a.row_base + census_rec_row(k.w));
				cnt[q] = census_rec_count(k.w);
				hh[q] = census_hash(klo[q], khi[q]);                     // (the table takes the hash's LOW bits: its high bits are the bucket's, the same for every key here)
			}
			SK_CSTAMP(3);                                              // next records asked for, keys and hashes
			lds_count_many<kAhead>(lt, have, hh, klo, khi, cnt, first_inv, dir);
			SK_CSTAMP(4);                                              // counted in the table
			bool any_dir = false;
#pragma unroll
			for (int q = 0; q < kAhead; q++) any_dir = any_dir || dir[q];
			if (__any(any_dir)) census_insert_many<kAhead>(a.t, dir, klo, khi, cnt, first_inv, claimed, overflow);      // (rare: its wait is for the next records' loads too)
			SK_CSTAMP(5);                                              // keys without a place: to HBM
#pragma unroll
			for (int q = 0; q < kAhead; q++) kk[q] = kn[q];
		}
		__syncthreads();
		SK_CSTAMP(6);                                                  // barrier behind the records
		// the occupied slots go to a list, and the list to HBM four keys of a thread at a time
		if (tid == 0) nlist = 0u;
		__syncthreads();
		for (int i = tid; i < kLdsSlots; i += blockDim.x)
			if (lt->klo[i] != 0) list[atomicAdd(&nlist, 1u)] = (uint16_t)i;
		__syncthreads();
		for (u32 j0 = tid; j0 < nlist; j0 += kAhead * blockDim.x) {
			bool want[kAhead];
			u64 klo[kAhead], khi[kAhead], first_inv[kAhead];
			u32 cnt[kAhead];
#pragma unroll
			for (int q = 0; q < kAhead; q++) {
				const u32 j = j0 + (u32)q * blockDim.x;
				want[q] = j < nlist;
				const int i = list[want[q] ? j : 0u];
				klo[q] = lt->klo[i]; khi[q] = ~lt->khi_inv[i]; first_inv[q] = lt->first_inv[i]; cnt[q] = lt->count[i];
			}
			census_insert_many<kAhead>(a.t, want, klo, khi, cnt, first_inv, claimed, overflow);
		}
		SK_CSTAMP(7);                                                  // occupied slots listed and inserted
	}
	census_add_stats(a.stats, claimed, overflow, red);
#ifdef SK_CENSUS_STAMPS
	SK_CSTAMP(8);
	if ((tid & 63) == 0 && blockIdx.x * (kCombineThreads / 64) + (tid >> 6) < 8192)
		for (int i = 0; i < kStampSlots; i++) g_combine_stamps[(blockIdx.x * (kCombineThreads / 64) + (tid >> 6)) * kStampSlots + i] = cst_acc[i];
#endif
}

// When most rows of a launch were spilled there is little to combine (the keys hardly repeat): workgroup g inserts the
// records of its region as they lie.
// (an insert is a chain of four memory round trips, so what this kernel needs is inserts in flight: `split` workgroups of 512
// threads share a region, as many resident per CU as the registers allow)
constexpr int kDirectThreads = 512;
__global__ __launch_bounds__(kDirectThreads) void census_direct_kernel(const CensusArgs a, const int split)
{
	__shared__ u32 red[16];
	const bool mine = census_spill_total(a.sp, red) > a.sp.direct_above;
	// the launch's last kernel leaves the bucket totals and the combine pass's counter zero for the next launch (every kernel
	// that reads them has finished: this one is behind them on the stream)
	if (blockIdx.x == 0) {
		if (threadIdx.x < (u32)kSpillBuckets) a.sp.btot[threadIdx.x] = 0u;
		if (threadIdx.x == 0) *a.sp.work = 0u;
	}
	if (!mine) return;
	const u32 g = blockIdx.x / (u32)split, part = blockIdx.x % (u32)split;
	u32 claimed = 0, overflow = 0;
	// the records of workgroup g's regions as one sequence, bucket after bucket; a thread takes four of them at a time
	__shared__ u32 bpre[kSpillBuckets + 1];
	__shared__ u32 ws[17];
	{
		static_assert(kSpillBuckets <= kDirectThreads, "one bucket per thread");
		const u32 v = threadIdx.x < (u32)kSpillBuckets ? a.sp.hist[(size_t)threadIdx.x * a.sp.grid + g] : 0u;
		u32 total;
		const u32 before = census_block_exclusive(v, ws, &total);
		if (threadIdx.x < (u32)kSpillBuckets) bpre[threadIdx.x] = before;
		if (threadIdx.x == 0) bpre[kSpillBuckets] = total;
		__syncthreads();
	}
	const u32 n = bpre[kSpillBuckets];
	const u32 rot = n != 0u ? (u32)(((u64)g * 0x9E3779B1ull) % n) : 0u;
	const uint4 *const key = a.sp.key + (size_t)g * kSpillBuckets * a.sp.cap;
	// (four keys of a thread at a time, census_insert_many, was measured here: 3.0 ms against 2.5 for 32 M new keys — what bounds
	// this kernel is the memory side's rate of scattered atomics and stores, not the chain's latency, and the plain loop asks less of it)
	for (u32 jj = part * blockDim.x + threadIdx.x; jj < n; jj += (u32)split * blockDim.x) {
		// (every workgroup begins somewhere else in its sequence: with all of them in the same bucket at the same time the
		// whole chip inserts into 1/256 of the table — the same few DRAM rows)
		u32 j = jj + rot;
		j = j >= n ? j - n : j;
		u32 l = 0u, r = (u32)kSpillBuckets;                               // the last bucket with bpre[b] <= j
		while (r - l > 1u) { const u32 m = (l + r) >> 1; if (bpre[m] <= j) l = m; else r = m; }
		const uint4 k = key[(size_t)l * a.sp.cap + (j - bpre[l])];
		const u64 klo = 0xF0000000ull | ((u64)k.x << 32), khi = (u64)k.y | ((u64)k.z << 32);
		const u32 c = census_rec_count(k.w);
		if (!census_insert(a.t, klo, khi, (u64)c, ~(u64)(a.row_base + census_rec_row(k.w)), claimed)) overflow += c;
	}
	census_add_stats(a.stats, claimed, overflow, red);
}

// grow: re-insert every slot of the old table into the new one (the claim log is started again for the new table)
__global__ __launch_bounds__(256) void census_rehash_kernel(const CensusSlot *old_tab, u64 old_slots, const CensusTab t, u64 *stats)
{
	u32 claimed = 0, overflow = 0;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < old_slots; i += (u64)gridDim.x * blockDim.x) {
		const CensusSlot s = old_tab[i];
		if (s.klo == 0) continue;
		if (!census_insert(t, s.klo, ~s.khi_inv, s.count, s.first_inv, claimed)) overflow++;
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

// the instantiations: R tiles per step x dwords per row (2, 5, 8: barcodes of at most 8, 20, 31 characters) x spill
constexpr int kCensusVariants = kCensusMaxSub * 3 * 2;
static const void *census_variant(int v)
{
	static const void *const tab[kCensusVariants] = {
		(const void *)census_kernel<1, 2, false>, (const void *)census_kernel<1, 2, true>, (const void *)census_kernel<1, 5, false>, (const void *)census_kernel<1, 5, true>,
		(const void *)census_kernel<1, 8, false>, (const void *)census_kernel<1, 8, true>,
		(const void *)census_kernel<2, 2, false>, (const void *)census_kernel<2, 2, true>, (const void *)census_kernel<2, 5, false>, (const void *)census_kernel<2, 5, true>,
		(const void *)census_kernel<2, 8, false>, (const void *)census_kernel<2, 8, true>,
	};
	return tab[v];
}

// ---- host side -----------------------------------------------------------------------------------------------

static void census_spill_free(Census *cs);

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
	cs->log_cap = kLogCap;
	if (const char *ev = getenv("SK_CENSUS_LOG_CAP_LOG2")) { const int lg = atoi(ev); if (lg >= 0 && lg <= 20) cs->log_cap = 1u << lg; else if (lg == -1) cs->log_cap = 0u; }
	if (e == hipSuccess) e = hipMalloc((void **)&cs->log, (size_t)kLogRegions * std::max(cs->log_cap, 1u) * sizeof(u32));
	if (e == hipSuccess) e = hipMalloc((void **)&cs->log_count, (size_t)kLogRegions * 32 * sizeof(u32));
	if (e == hipSuccess) e = hipMemsetAsync(cs->log_count, 0, (size_t)kLogRegions * 32 * sizeof(u32), st);
	for (int v = 0; v < kCensusVariants; v++)
		if (e == hipSuccess) e = hipFuncSetAttribute(census_variant(v), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	if (e != hipSuccess) { census_destroy(cs); return e; }
	*out = cs;
	return hipSuccess;
}

void census_destroy(Census *cs)
{
	if (!cs) return;
	if (cs->tab) (void)hipFree(cs->tab);
	if (cs->log) (void)hipFree(cs->log);
	if (cs->log_count) (void)hipFree(cs->log_count);
	if (cs->stats) (void)hipFree(cs->stats);
	if (cs->scratch) (void)hipFree(cs->scratch);
	census_spill_free(cs);
	delete cs;
}

static CensusTab census_tab_of(const Census *cs)
{
	CensusTab t;
	t.tab = cs->tab;
	t.mask = cs->slots - 1;
	int lg = 0;
	while ((1ull << lg) < cs->slots) lg++;
	t.sh = (u32)(32 - lg);
	t.log = cs->log;
	t.log_count = cs->log_count;
	t.log_cap = cs->log_cap;
	return t;
}

// Empty the table: the slots the claim log names — or, when a region of the log overflowed, all of them (decided here, on the
// device: every workgroup reads the 256 counters) — then the counters (by the kernel behind this one on the stream).
__global__ __launch_bounds__(256) void census_clear_kernel(const CensusTab t)
{
	__shared__ u32 cnt[kLogRegions];
	__shared__ u32 over;
	if (threadIdx.x == 0) over = 0u;
	__syncthreads();
	for (int r = threadIdx.x; r < kLogRegions; r += blockDim.x) {
		const u32 c = t.log_count[r * 32];
		cnt[r] = c < t.log_cap ? c : t.log_cap;
		if (c > t.log_cap || t.log_cap == 0u) over = 1u;
	}
	__syncthreads();
	uint4 *const p = reinterpret_cast<uint4 *>(t.tab);
	if (over) {
		const u64 n = (t.mask + 1) * (sizeof(CensusSlot) / 16);
		for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) p[i] = make_uint4(0u, 0u, 0u, 0u);
		return;
	}
	// region r's entries are cleared by the workgroups r, r + 256, ... : a thread per entry
	const int r = blockIdx.x & (kLogRegions - 1);
	const u32 per = gridDim.x / kLogRegions ? gridDim.x / kLogRegions : 1u, part = blockIdx.x / kLogRegions;
	for (u32 i = part * blockDim.x + threadIdx.x; i < cnt[r]; i += per * blockDim.x) {
		const u64 slot = t.log[(size_t)r * t.log_cap + i];
		p[2 * slot] = make_uint4(0u, 0u, 0u, 0u);
		p[2 * slot + 1] = make_uint4(0u, 0u, 0u, 0u);
	}
}

hipError_t census_reset(Census *cs, int n_cu, hipStream_t st)
{
	hipError_t e = hipMemsetAsync(cs->stats, 0, kCensusStats * sizeof(u64), st);
	if (e == hipSuccess) {
		census_clear_kernel<<<kLogRegions * 8, 256, 0, st>>>(census_tab_of(cs));
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemsetAsync(cs->log_count, 0, (size_t)kLogRegions * 32 * sizeof(u32), st);
	(void)n_cu;
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
	e = hipMemsetAsync(cs->log_count, 0, (size_t)kLogRegions * 32 * sizeof(u32), st);      // the log names slots of the table that goes
	if (e != hipSuccess) return e;
	census_rehash_kernel<<<n_cu * 8, 256, 0, st>>>(old_tab, old_slots, census_tab_of(cs), cs->stats);
	e = hipGetLastError();
	if (e == hipSuccess) e = hipStreamSynchronize(st);
	(void)hipFree(old_tab);
	return e;
}

static void census_spill_free(Census *cs)
{
	void *ptrs[] = {cs->sp.key, cs->sp.hist, cs->sp.btot, cs->sp.wg_count, cs->sp.wg_stats, cs->sp.work};
	for (void *q : ptrs) if (q) (void)hipFree(q);
	cs->sp = CensusSpill();
	cs->sp_records = 0;
	cs->sp_grid = 0;
}

// room for `grid` workgroups x kSpillBuckets regions of `cap` records each
static hipError_t census_spill_reserve(Census *cs, int grid, u32 cap, hipStream_t st)
{
	const size_t records = (size_t)grid * kSpillBuckets * cap;
	if (records <= cs->sp_records && grid <= cs->sp_grid) return hipSuccess;
	const size_t want_records = std::max(records, cs->sp_records);
	const int want_grid = std::max(grid, cs->sp_grid);
	hipError_t e = hipStreamSynchronize(st);                           // (an earlier launch may still be reading the arrays that go)
	if (e != hipSuccess) return e;
	census_spill_free(cs);
	e = hipMalloc((void **)&cs->sp.key, want_records * sizeof(uint4));
	if (e == hipSuccess) e = hipMalloc((void **)&cs->sp.hist, (size_t)want_grid * kSpillBuckets * sizeof(u32));
	if (e == hipSuccess) e = hipMalloc((void **)&cs->sp.btot, kSpillBuckets * sizeof(u32));
	if (e == hipSuccess) e = hipMalloc((void **)&cs->sp.wg_count, (size_t)want_grid * sizeof(u32));
	if (e == hipSuccess) e = hipMalloc((void **)&cs->sp.wg_stats, (size_t)want_grid * kCensusStats * sizeof(u32));
	if (e == hipSuccess) e = hipMalloc((void **)&cs->sp.work, sizeof(u32));
	if (e == hipSuccess) e = hipMemsetAsync(cs->sp.btot, 0, kSpillBuckets * sizeof(u32), st);      // (later launches find them zero: census_direct_kernel)
	if (e == hipSuccess) e = hipMemsetAsync(cs->sp.work, 0, sizeof(u32), st);
	if (e != hipSuccess) { (void)hipGetLastError(); census_spill_free(cs); return e; }
	cs->sp_records = want_records;
	cs->sp_grid = want_grid;
	return hipSuccess;
}

constexpr int64_t kSpillMinRows = 1 << 19;      // smaller launches insert directly: the partition path is four more kernels.  Measured (tools/census_small.py, round 4,
                                                // insert-in-place / partition, noisy run): 250 k rows 58 / 68 us, 1 M 100 / 80, 4 M 187 / 110, 8 M 308 / 149; rows that
                                                // are all sheet barcodes: 250 k 40 / 47, 1 M 65 / 55, 4 M 84 / 68 (round 3's crossover was at 8 M rows)

hipError_t census_add(Census *cs, const uint8_t *bc, int bc_stride, int L, int64_t n, const int32_t *assign, int64_t row_base,
                      int n_cu, hipStream_t st)
{
	if (bc_stride > kCensusMaxStride) return hipErrorInvalidValue;
	int R = kCensusStepBytes / (64 * bc_stride);
	R = R < 1 ? 1 : (R > kCensusMaxSub ? kCensusMaxSub : R);
	if (const char *ev = getenv("SK_CENSUS_TILES")) { const int v = atoi(ev); if (v >= 1 && v <= R) R = v; }      // experiments
	const int nw_dwords = L <= 8 ? 2 : (L <= 20 ? 5 : 8);
	// (two 64-row tiles per step at most: more bought nothing — R = 1 ... 4 timed the same within the boxes' noise in round 4 — and
	// a tile's 64 queue entries are 43 front-table entries of a 17-byte sheet)
	// a wave's slot: its queue of rows that take the long way, 64 left over + the step's R x 64, each its NW dwords and its
	// row's index; behind it the flush queue of the launches that insert by themselves (over the queue's dead part)
	auto slot_bytes = [&](int r, bool spill_layout) {
		const int queue_cap = 64 + r * 64;
		int bytes = queue_cap * (4 * nw_dwords + 4);
		if (!spill_layout) {
			const int flush_end = ((queue_cap * 4 + 64 * 4 * nw_dwords + 15) & ~15) + kCensusQueue * 20;
			if (bytes < flush_end) bytes = flush_end;
		}
		return (bytes + 15) & ~15;
	};
	// the front table takes what is left of the CU's 160 KiB (SK_CENSUS_FRONT_ENTRIES: tests shrink it so that small inputs
	// walk "no room in either place")
	const int nw_class = L <= 8 ? 0 : (L <= 20 ? 1 : 2);
	const int front_entry_bytes = nw_class == 2 ? 52 : 36;            // FrontShape<NW>::kEntryBytes
	auto plan_lds = [&](int r, bool spill_layout, int &tile_slot, int &front_entries) {
		tile_slot = slot_bytes(r, spill_layout);
		size_t lds = (size_t)kCensusWaves * tile_slot + 64 + (kSpillBuckets + 4) * sizeof(u32);      // + the workgroup's cursors and its step counter
		front_entries = (int)((160 * 1024 - 16 - lds) / front_entry_bytes) & ~63;
		if (const char *ev = getenv("SK_CENSUS_FRONT_ENTRIES")) { const int v = atoi(ev) & ~63; if (v >= 64 && v <= front_entries) front_entries = v; }
		return lds + (((size_t)front_entries * front_entry_bytes + 15) & ~(size_t)15);
	};
	int wgs_per_cu = 1;
	if (const char *ev = getenv("SK_CENSUS_WGS")) { const int v = atoi(ev); if (v >= 1 && v <= 4) wgs_per_cu = v; }      // experiments
	int direct_pct = 50;                                    // SK_CENSUS_SPILL_MAX_PCT: more spilled rows than this share of a launch are inserted directly
	if (const char *ev = getenv("SK_CENSUS_SPILL_MAX_PCT")) { const int v = atoi(ev); if (v >= 0 && v <= 100) direct_pct = v; }
	int64_t spill_min_rows = kSpillMinRows;                 // SK_CENSUS_SPILL_MIN_ROWS_LOG2: tests lower it
	if (const char *ev = getenv("SK_CENSUS_SPILL_MIN_ROWS_LOG2")) { const int lg = atoi(ev); if (lg >= 6 && lg <= 30) spill_min_rows = (int64_t)1 << lg; }
	int direct_split = 8;                                   // SK_CENSUS_DIRECT_SPLIT: workgroups per region of census_direct_kernel (experiments)
	if (const char *ev = getenv("SK_CENSUS_DIRECT_SPLIT")) { const int v = atoi(ev); if (v >= 1 && v <= 64) direct_split = v; }
	int merge_copies = 1;                                   // SK_CENSUS_MERGE_RECORDS=0: the front kernel inserts its tables itself, as small launches do (A/B, tests)
	if (const char *ev = getenv("SK_CENSUS_MERGE_RECORDS")) merge_copies = atoi(ev) != 0;
	int spill_mode = -1;                                    // SK_CENSUS_SPILL=0/1: never / always (tests, experiments)
	if (const char *ev = getenv("SK_CENSUS_SPILL")) spill_mode = atoi(ev) != 0;
	// (SK_CENSUS_CHUNK_LOG2 / SK_CENSUS_MIN_CHUNK_LOG2: tests shrink the launches to walk the grow / smaller-bite decisions)
	int64_t chunk = kCensusChunk, min_chunk = kCensusMinChunk;
	if (const char *ev = getenv("SK_CENSUS_CHUNK_LOG2")) { const int lg = atoi(ev); if (lg >= 6 && lg <= kRecRowBits) chunk = (int64_t)1 << lg; }      // (a record's row index has kRecRowBits bits)
	if (const char *ev = getenv("SK_CENSUS_MIN_CHUNK_LOG2")) { const int lg = atoi(ev); if (lg >= 6 && lg <= 30) min_chunk = (int64_t)1 << lg; }
	{	// a launch's matrix stays below 2 GiB: the front kernel addresses it through one descriptor with 32-bit offsets
		const int64_t max_rows = ((((int64_t)1 << 31) - (1 << 24)) / bc_stride) & ~(int64_t)63;
		if (chunk > max_rows) chunk = max_rows;
	}
	if (min_chunk > chunk) min_chunk = chunk;
	int64_t nr = 0;
	for (int64_t o = 0; o < n; o += nr) {
		// The launch must not be able to fill the table beyond one half even if every row is a new key.  When
		// the bound on the key count says it could, fetch the exact count; then either take a smaller bite
		// (at least kCensusMinChunk rows: launches have a fixed cost) or grow the table.
		nr = (n - o) < chunk ? (n - o) : chunk;
		hipError_t e = hipSuccess;
		if (2 * (cs->distinct + (u64)nr) > cs->slots && !getenv("SK_CENSUS_TRUST_SLOTS")) {      // (the env: experiments with a table that is known to be large enough)
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
		a.t = census_tab_of(cs);
		a.stats = cs->stats;
		a.long_way_only = getenv("SK_CENSUS_LONG_WAY_ONLY") && atoi(getenv("SK_CENSUS_LONG_WAY_ONLY")) != 0;
		bool spill = L <= kSpillMaxLen && (spill_mode < 0 ? nr >= spill_min_rows : spill_mode != 0);
		// One tile per step for the launches that write records: the step's queue is a third smaller, and what it gives back goes
		// to the front table (2 270 -> 2 950 entries for 17-byte rows) — the table's size is worth more than the second tile (32 M
		// noisy rows 0.320 -> 0.312 ms, clean 0.245 -> 0.231; with 1 024 entries 0.34-0.41, with 512 0.53; tools/ab/census_env_ab.py)
		const int Rk = (spill && nw_class >= 1 && !getenv("SK_CENSUS_TILES")) ? 1 : R;
		auto grid_of = [&](int r) {
			const int64_t groups = (nr + (int64_t)64 * r * kCensusWaves - 1) / ((int64_t)64 * r * kCensusWaves);
			const int g = n_cu * wgs_per_cu;
			return g > groups ? (int)groups : g;
		};
		int grid = grid_of(Rk);
		a.sp = CensusSpill();
		if (spill) {
			// a (workgroup, bucket) region: twice the workgroup's share of a launch in which EVERY row is spilled and spreads evenly, and
			// room for its front table's records; a piece that finds its region full is inserted by the front kernel itself
			const int64_t nsteps = (nr + (int64_t)64 * Rk - 1) / ((int64_t)64 * Rk);
			const int64_t per_wave = (nsteps + (int64_t)grid * kCensusWaves - 1) / ((int64_t)grid * kCensusWaves);
			const int64_t wg_rows = per_wave * kCensusWaves * Rk * 64;
			const int64_t cap = (((wg_rows + kSpillBuckets - 1) / kSpillBuckets) * 2 + 256 + 7) & ~(int64_t)7;
			e = (grid <= kSpillMaxGrid && cap * kSpillBuckets < ((int64_t)1 << 27)) ? census_spill_reserve(cs, grid, (u32)cap, st) : hipErrorInvalidValue;
			if (e != hipSuccess) { (void)hipGetLastError(); spill = false; grid = grid_of(Rk); }      // no room for the records: insert directly
			else {
				a.sp = cs->sp;
				a.sp.cap = (u32)cap;
				a.sp.grid = (u32)grid;
				a.sp.direct_above = (u32)((uint64_t)nr * (uint64_t)direct_pct / 100);
				a.sp.merge_copies = (u32)merge_copies;
			}
		}
		int tile_slot = 0, front_entries = 0;
		const size_t lds = plan_lds(Rk, spill, tile_slot, front_entries);
		hipLaunchKernelGGL(reinterpret_cast<void (*)(const CensusArgs, const int, const int)>(const_cast<void *>(census_variant(((Rk - 1) * 3 + nw_class) * 2 + (spill ? 1 : 0)))),
		                   dim3(grid), dim3(kCensusWaves * 64), lds, st, a, tile_slot, front_entries);
		if (spill) {
			census_combine_kernel<<<2 * n_cu, kCombineThreads, 0, st>>>(a);
			census_direct_kernel<<<grid * direct_split, kDirectThreads, 0, st>>>(a, direct_split);
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
	for (u64 i = 0; i < got; i++) {
		const u64 *r = &raw[order[i] * 4];
		const u64 klo = r[0], khi = ~r[1];
		const u32 kw[4] = {(u32)(klo >> 32), (u32)khi, (u32)(khi >> 32), (u32)klo};
		CensusEntry &en = out[i];
		memset(en.barcode, 0, sizeof en.barcode);
		for (int j = 0; j < 31; j++) {
			const int q = j >> 2, b = j & 3;
			const u32 c = (kw[q >> 1] >> (8 * b + 4 * (q & 1))) & 15u;
			if (c == 0) break;
			en.barcode[j] = kCensusAlphabet[c];
		}
		en.count = r[2];
		en.first_row = (int64_t)~r[3];
	}
	return hipSuccess;
}

}  // namespace sk

#ifdef SK_CENSUS_STAMPS
extern "C" int sk_debug_census_stamps(unsigned long long *out, int waves)
{
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sk::g_census_stamps), (size_t)waves * sk::kStampSlots * sizeof(unsigned long long));
}
extern "C" int sk_debug_combine_stamps(unsigned long long *out, int waves)
{
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sk::g_combine_stamps), (size_t)waves * sk::kStampSlots * sizeof(unsigned long long));
}
#endif
