// Where do atomics of the two scopes run, and how fast?  (census second level: DESIGN.md §3.7)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_scope_exp tools/atomic_scope_exp.hip && /tmp/atomic_scope_exp
// Every thread adds 1 to a pseudo-random slot (32-byte slots) of a table; ops: agent-scope add, workgroup-scope add on a
// table of the workgroup's own XCD (8 tables, chosen by HW_REG_XCC_ID), and the same with the returning form.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned long long u64;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16; return h; }

template <int MODE> __global__ __launch_bounds__(512) void k(u64 *tab, u64 slots_per_tab, int per_thread, u64 *xcd_rows)
{
	uint32_t xcc;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	xcc &= 0xf;
	u64 *t = tab;
	if (MODE >= 2) t = tab + (u64)xcc * slots_per_tab * 4;
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	u64 acc = 0;
	for (int i = 0; i < per_thread; i++) {
		const u64 idx = (u64)mix(gid * 977u + (uint32_t)i * 0x9E3779B1u) & (slots_per_tab - 1);
		u64 *p = t + idx * 4 + 2;
		if (MODE == 0) __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (MODE == 1) acc += __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (MODE == 2) __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		if (MODE == 3) acc += __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		if (MODE == 4) {      // the census hit path at workgroup scope: CAS on the key word + add on the count
			const u64 key = idx * 2 + 1;
			const u64 old = __hip_atomic_compare_exchange_strong(t + idx * 4, (u64 *)&acc, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			(void)old;
			acc = 0;
			__hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		if (MODE == 5) acc += __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (MODE == 6) {
			acc += __hip_atomic_load(p - 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			acc += __hip_atomic_load(p - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			acc += __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if (MODE == 7 || MODE == 10) {
			u32x4 v;
			asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p - 2) : "memory");
			acc += v.x + v.z;
			if (MODE == 10) __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if (MODE == 8) acc += *(volatile u64 *)p;
		if (MODE == 9) {
			u32x4 v = {gid, (uint32_t)i, 1u, 0u};
			asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
		}
		if (MODE == 11) {
			acc += __hip_atomic_load(p - 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			acc += __hip_atomic_load(p - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			acc += __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (acc != 0x7777) __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if (MODE == 12) {
			u32x4 v = {gid, (uint32_t)i, 1u, 0u};
			*(u32x4 *)p = v;
		}
	}
	if (acc == 0x1234567) tab[0] = acc;
	if (threadIdx.x == 0) atomicAdd(&xcd_rows[xcc], (u64)blockDim.x * per_thread);
}

__global__ void sum_k(const u64 *tab, u64 slots, u64 *out)
{
	u64 s = 0;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (u64)gridDim.x * blockDim.x) s += tab[i * 4 + 2];
	atomicAdd(out, s);
}

int main()
{
	const int grid = 2048, block = 512, per_thread = 64;
	const u64 ops = (u64)grid * block * per_thread;
	u64 *d_rows, *d_sum;
	CK(hipMalloc(&d_rows, 16 * 8));
	CK(hipMalloc(&d_sum, 8));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int lg : {13, 16, 18, 21, 26}) {
		const u64 slots = 1ull << lg;
		u64 *tab;
		CK(hipMalloc(&tab, slots * 32 * 8));
		for (int mode = 0; mode < 13; mode++) {
			float best = 1e9f;
			u64 total = 0;
			for (int rep = 0; rep < 3; rep++) {
				CK(hipMemset(tab, 0, slots * 32 * 8));
				CK(hipMemset(d_rows, 0, 16 * 8));
				CK(hipMemset(d_sum, 0, 8));
				CK(hipDeviceSynchronize());
				CK(hipEventRecord(e0));
				switch (mode) {
				case 0: k<0><<<grid, block>>>(tab, slots, per_thread, d_rows); break;
				case 1: k<1><<<grid, block>>>(tab, slots, per_thread, d_rows); break;
				case 2: k<2><<<grid, block>>>(tab, slots, per_thread, d_rows); break;
				case 3: k<3><<<grid, block>>>(tab, slots, per_thread, d_rows); break;
				case 4: k<4><<<grid, block>>>(tab, slots, per_thread, d_rows); break;
				case 5: k<5><<<grid, block>>>(tab, slots, per_thread, d_rows); break;
				case 6: k<6><<<grid, block>>>(tab, slots, per_thread, d_rows); break;
				case 7: k<7><<<grid, block>>>(tab, slots, per_thread, d_rows); break;
				case 8: k<8><<<grid, block>>>(tab, slots, per_thread, d_rows); break;
				case 9: k<9><<<grid, block>>>(tab, slots, per_thread, d_rows); break;
				case 10: k<10><<<grid, block>>>(tab, slots, per_thread, d_rows); break;
				case 11: k<11><<<grid, block>>>(tab, slots, per_thread, d_rows); break;
				case 12: k<12><<<grid, block>>>(tab, slots, per_thread, d_rows); break;
				}
				CK(hipEventRecord(e1));
				CK(hipEventSynchronize(e1));
				float ms;
				CK(hipEventElapsedTime(&ms, e0, e1));
				if (ms < best) best = ms;
				sum_k<<<1024, 256>>>(tab, slots * 8, d_sum);
				CK(hipMemcpy(&total, d_sum, 8, hipMemcpyDeviceToHost));
			}
			u64 rows[16];
			CK(hipMemcpy(rows, d_rows, sizeof rows, hipMemcpyDeviceToHost));
			static const char *names[] = {"agent add", "agent add, returning", "workgroup add, table per XCD", "workgroup add returning, per XCD", "workgroup CAS + add, per XCD", "agent load 8 B", "3 agent loads, one slot", "load 16 B sc1", "plain load 8 B", "store 16 B sc1", "load 16 B sc1 + agent add", "3 agent loads + agent add", "plain store 16 B"};
			printf("slots/table 2^%d (%6.0f MiB)  %-34s %8.3f ms %7.2f G ops/s  sum %s (%llu of %llu)  xcd rows:", lg, slots * 32 / 1048576.0, names[mode], best, ops / best / 1e6,
			       total == ops ? "exact" : "-", total, ops);
			for (int x = 0; x < 8; x++) printf(" %llu", rows[x] / 1000000);
			printf("\n");
			fflush(stdout);
		}
		CK(hipFree(tab));
	}
	return 0;
}
