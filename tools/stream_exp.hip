// What does plain streaming reach on this GPU?  The ceiling the byte-stream kernels of this repository are up against.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_exp tools/stream_exp.hip && /tmp/stream_exp
// Kernels: read only (xor-reduce) of one array and of three at once, copy, two reads + one write (the mask pass's
// traffic shape), hipMemcpyAsync D2D and
// hipMemsetAsync; 16 bytes per lane and access, persistent workgroups, with and without the nt cache hint; each on a
// few launch shapes, best of 5 launches of 9.4 GB per array.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int KIND, bool NT> __global__ __launch_bounds__(256) void k(const u32x4 *a, const u32x4 *b, u32x4 *o, int64_t n16, uint32_t *sink)
{
	u32x4 acc = {0, 0, 0, 0};
	const int64_t stride = (int64_t)gridDim.x * 256 * 4;
	for (int64_t i = (int64_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n16; i += stride) {
		u32x4 x[4], y[4];
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const int64_t j = i + u * 256;
			if (j < n16) {
				x[u] = NT ? __builtin_nontemporal_load(a + j) : a[j];
				if (KIND == 2 || KIND == 3) y[u] = NT ? __builtin_nontemporal_load(b + j) : b[j];
				if (KIND == 3) { const u32x4 z = NT ? __builtin_nontemporal_load(o + j) : o[j]; y[u] ^= z; }
			}
		}
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const int64_t j = i + u * 256;
			if (j < n16) {
				if (KIND == 0) acc ^= x[u];
				if (KIND == 3) acc ^= x[u] ^ y[u];
				if (KIND == 1) { if (NT) __builtin_nontemporal_store(x[u], o + j); else o[j] = x[u]; }
				if (KIND == 2) { const u32x4 r = x[u] & y[u]; if (NT) __builtin_nontemporal_store(r, o + j); else o[j] = r; }
			}
		}
	}
	if ((KIND == 0 || KIND == 3) && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = 1;
}

int main()
{
	const int64_t bytes = 9375000000ll / 16 * 16, n16 = bytes / 16;
	u32x4 *a, *b, *o;
	uint32_t *sink;
	CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, bytes)); CK(hipMalloc(&sink, 4));
	CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes)); CK(hipMemset(o, 0, bytes));
	hipDeviceProp_t p;
	CK(hipGetDeviceProperties(&p, 0));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	static const char *names[] = {"read", "copy", "2 reads + 1 write", "3 arrays read"};
	static const int traffic[] = {1, 2, 3, 3};
	for (int kind = 0; kind < 4; kind++)
		for (int nt = 0; nt < 2; nt++)
			for (int wgs = 1; wgs <= 8; wgs *= 2) {
				const int grid = p.multiProcessorCount * wgs;
				float best = 1e9f;
				for (int rep = 0; rep < 5; rep++) {
					CK(hipEventRecord(e0));
					if (kind == 0 && !nt) k<0, false><<<grid, 256>>>(a, b, o, n16, sink);
					if (kind == 0 && nt) k<0, true><<<grid, 256>>>(a, b, o, n16, sink);
					if (kind == 1 && !nt) k<1, false><<<grid, 256>>>(a, b, o, n16, sink);
					if (kind == 1 && nt) k<1, true><<<grid, 256>>>(a, b, o, n16, sink);
					if (kind == 2 && !nt) k<2, false><<<grid, 256>>>(a, b, o, n16, sink);
					if (kind == 2 && nt) k<2, true><<<grid, 256>>>(a, b, o, n16, sink);
					if (kind == 3 && !nt) k<3, false><<<grid, 256>>>(a, b, o, n16, sink);
					if (kind == 3 && nt) k<3, true><<<grid, 256>>>(a, b, o, n16, sink);
					CK(hipEventRecord(e1));
					CK(hipEventSynchronize(e1));
					float ms;
					CK(hipEventElapsedTime(&ms, e0, e1));
					if (ms < best) best = ms;
				}
				printf("%-18s %-3s %d workgroups/CU  %7.3f ms  %6.2f TB/s (%4.1f %% of 8 TB/s)\n", names[kind], nt ? "nt" : "", wgs, best,
				       traffic[kind] * bytes / best / 1e9, traffic[kind] * bytes / best / 1e9 / 8 * 100);
				fflush(stdout);
			}
	for (int what = 0; what < 2; what++) {
		float best = 1e9f;
		for (int rep = 0; rep < 5; rep++) {
			CK(hipEventRecord(e0));
			if (what == 0) CK(hipMemcpyAsync(o, a, bytes, hipMemcpyDeviceToDevice, 0));
			else CK(hipMemsetAsync(o, 3, bytes, 0));
			CK(hipEventRecord(e1));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			if (ms < best) best = ms;
		}
		const int tr = what == 0 ? 2 : 1;
		printf("%-37s  %7.3f ms  %6.2f TB/s (%4.1f %% of 8 TB/s)\n", what == 0 ? "hipMemcpyAsync device to device" : "hipMemsetAsync", best, tr * bytes / best / 1e9, tr * bytes / best / 1e9 / 8 * 100);
	}
	return 0;
}
