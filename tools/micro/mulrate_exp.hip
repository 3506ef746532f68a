// Issue rate of v_mul_lo_u32 and v_mul_u32_u24 against plain ops in dependent chains: 4.5 / 4.4 / 4.65 cycles per wave instruction on MI355X (full rate).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mulrate_exp tools/micro/mulrate_exp.hip && /tmp/mulrate_exp
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND> __global__ __launch_bounds__(256) void k(unsigned *out, unsigned b, unsigned c, int iters)
{
	unsigned a0 = threadIdx.x * 2654435761u + 1, a1 = a0 ^ 0x1234567u, a2 = a0 + 77u, a3 = a0 * 3u + 5u;
	for (int i = 0; i < iters; i++) {
		if (KIND == 0) { a0 = (a0 ^ b) + c; a1 = (a1 ^ b) + c; a2 = (a2 ^ b) + c; a3 = (a3 ^ b) + c; }               // 2 plain ops per chain step
		if (KIND == 1) { a0 = (a0 * b) ^ c; a1 = (a1 * b) ^ c; a2 = (a2 * b) ^ c; a3 = (a3 * b) ^ c; }               // mul_lo + xor
		if (KIND == 2) { a0 = __umul24(a0, b) ^ c; a1 = __umul24(a1, b) ^ c; a2 = __umul24(a2, b) ^ c; a3 = __umul24(a3, b) ^ c; }   // mul_u32_u24 + xor
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}
int main()
{
	hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
	const int grid = p.multiProcessorCount * 8, iters = 20000;
	unsigned *out; hipMalloc(&out, grid * 256 * 4);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const char *names[] = {"xor + add (2 ops)", "v_mul_lo_u32 + xor", "v_mul_u32_u24 + xor"};
	for (int kind = 0; kind < 3; kind++) {
		float best = 1e9;
		for (int rep = 0; rep < 3; rep++) {
			hipEventRecord(e0);
			if (kind == 0) k<0><<<grid, 256>>>(out, 0x9E3779B1u, 12345u, iters);
			if (kind == 1) k<1><<<grid, 256>>>(out, 0x9E3779B1u, 12345u, iters);
			if (kind == 2) k<2><<<grid, 256>>>(out, 0x9E3779B1u, 12345u, iters);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
		}
		// per SIMD: 8 waves x iters x 4 chains x 2 ops
		const double ops = 8.0 * iters * 4 * 2;
		printf("%-24s %.3f ms  %.2f cycles per instruction pair-op (2.4 GHz)\n", names[kind], best, best * 1e-3 * 2.4e9 / ops);
	}
	return 0;
}
