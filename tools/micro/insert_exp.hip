// What does putting a NEW key into the census table cost, by protocol and by access order?  2^26 slots of 32 bytes (2 GiB),
// 32 M inserts, every thread one slot per iteration.
//   protocol 0  today's: peek (16 B, sc1) -> 64-bit CAS on the key word -> 16-byte store of count/first (sc1) -> wait -> 8-byte store of ~khi
//   protocol 1  owned region: peek -> two plain 16-byte stores (the whole slot); no atomic
//   protocol 2  two plain 16-byte stores, no peek
//   protocol 3  peek only
//   protocol 4  CAS only
//   protocol 5  the slot's 32 bytes as ONE store instruction: lanes 2k and 2k+1 write its two halves (each lane pair one insert)
//   protocol 6  peek (both lanes of the pair load their half) + the pair store
//   protocol 7  as 5 with the nt hint; 8: as 5 with sc1
// order 0: slots random over the whole table; order 1: workgroup g works inside region g mod 1024 (1/1024 of the table: the
// partition path's buckets), random within it
// hipcc --offload-arch=gfx950 -O3 -o tools/ab/insert_exp tools/micro/insert_exp.hip && tools/ab/insert_exp
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef unsigned long long u64;
typedef uint32_t u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ u32 mix(u32 h) { h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16; return h; }

template <int PROTO, int ORDER> __global__ __launch_bounds__(512) void k(u64 *tab, int lg, int per_thread, u64 *out)
{
	const u64 slots = 1ull << lg;
	const u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
	u64 acc = 0;
	for (int i = 0; i < per_thread; i++) {
		const u32 h = mix(gid * 977u + (u32)i * 0x9E3779B1u);
		u64 idx;
		if (ORDER == 0) idx = (u64)h & (slots - 1);
		else idx = ((u64)(blockIdx.x & 1023) << (lg - 10)) | ((u64)h & ((slots >> 10) - 1));
		u64 *s = tab + idx * 4;
		const u64 key = ((u64)gid << 20) | (u64)i | 1ull;
		u32x4 v = {0, 0, 0, 0};
		if (PROTO == 0 || PROTO == 1 || PROTO == 3) {
			asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(s) : "memory");
			acc += v.x;
		}
		if (PROTO == 0 || PROTO == 4) {
			u64 expect = (u64)v.x | ((u64)v.y << 32);
			if (PROTO == 4) expect = 0;
			const u64 old = atomicCAS(s, expect, key);
			acc += old;
		}
		if (PROTO == 0) {
			const u32x4 w = {1u, 0u, gid, (u32)i};
			asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(s + 2), "v"(w) : "memory");
			__builtin_amdgcn_s_waitcnt(0x0F70);
			__hip_atomic_store(s + 1, ~key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if (PROTO == 1 || PROTO == 2) {
			const u32x4 w = {1u, 0u, gid, (u32)i};
			const u32x4 kk = {(u32)key, (u32)(key >> 32), ~(u32)key, ~(u32)(key >> 32)};
			*(u32x4 *)(s + 2) = w;
			*(u32x4 *)s = kk;
		}
	}
	if (acc == 0x1234567) out[0] = acc;
}

template <int PROTO, int ORDER> __global__ __launch_bounds__(512) void kp(u64 *tab, int lg, int per_thread, u64 *out)
{
	const u64 slots = 1ull << lg;
	const u32 gid = (blockIdx.x * blockDim.x + threadIdx.x) >> 1;      // the pair's insert
	const u32 half = threadIdx.x & 1;
	u64 acc = 0;
	for (int i = 0; i < 2 * per_thread; i++) {                         // twice the iterations: a pair does one insert per iteration
		const u32 h = mix(gid * 977u + (u32)i * 0x9E3779B1u);
		u64 idx;
		if (ORDER == 0) idx = (u64)h & (slots - 1);
		else idx = ((u64)(blockIdx.x & 1023) << (lg - 10)) | ((u64)h & ((slots >> 10) - 1));
		u64 *s = tab + idx * 4 + half * 2;
		const u64 key = ((u64)gid << 20) | (u64)i | 1ull;
		if (PROTO == 6) {
			u32x4 v;
			asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(s) : "memory");
			acc += v.x;
		}
		const u32x4 w = half ? u32x4{1u, 0u, gid, (u32)i} : u32x4{(u32)key, (u32)(key >> 32), ~(u32)key, ~(u32)(key >> 32)};
		if (PROTO == 5 || PROTO == 6) *(u32x4 *)s = w;
		if (PROTO == 7) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(s), "v"(w) : "memory");
		if (PROTO == 8) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(s), "v"(w) : "memory");
	}
	if (acc == 0x1234567) out[0] = acc;
}

int main()
{
	const int lg = 26;
	const int grid = 2048, block = 512, per_thread = 32;      // 32 M inserts
	const u64 ops = (u64)grid * block * per_thread;
	u64 *tab, *out;
	CK(hipMalloc(&tab, (32ull << lg)));
	CK(hipMalloc(&out, 64));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	static const char *pn[] = {"peek + CAS + store16 + wait + store8 (today)", "peek + 2 x store16 (owned)", "2 x store16", "peek", "CAS",
	                           "pair store 32 B", "pair peek + pair store", "pair store nt", "pair store sc1"};
	for (int order = 0; order < 2; order++)
		for (int proto = 0; proto < 9; proto++) {
			float best = 1e9f;
			for (int rep = 0; rep < 3; rep++) {
				CK(hipMemset(tab, 0, 32ull << lg));
				CK(hipDeviceSynchronize());
				CK(hipEventRecord(e0));
#define L(P, O) k<P, O><<<grid, block>>>(tab, lg, per_thread, out)
#define LP(P, O) kp<P, O><<<grid, block>>>(tab, lg, per_thread, out)
				switch (order * 9 + proto) {
				case 0: L(0, 0); break; case 1: L(1, 0); break; case 2: L(2, 0); break; case 3: L(3, 0); break; case 4: L(4, 0); break;
				case 5: LP(5, 0); break; case 6: LP(6, 0); break; case 7: LP(7, 0); break; case 8: LP(8, 0); break;
				case 9: L(0, 1); break; case 10: L(1, 1); break; case 11: L(2, 1); break; case 12: L(3, 1); break; case 13: L(4, 1); break;
				case 14: LP(5, 1); break; case 15: LP(6, 1); break; case 16: LP(7, 1); break; case 17: LP(8, 1); break;
				}
				CK(hipEventRecord(e1));
				CK(hipEventSynchronize(e1));
				float ms;
				CK(hipEventElapsedTime(&ms, e0, e1));
				if (ms < best) best = ms;
			}
			printf("order %d (%s)  %-48s %8.3f ms  %6.2f G inserts/s\n", order, order ? "workgroup inside one of 1024 regions" : "random over the table", pn[proto], best, ops / best / 1e6);
			fflush(stdout);
		}
	return 0;
}
