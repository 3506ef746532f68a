// Do scalar instructions of one wave issue beside vector instructions of another?  k<NV, NS>: NV v_xor + NS s_xor per group (the compiler puts an s_nop after most inline-asm VALU ops).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_exp tools/micro/issue_exp.hip && /tmp/issue_exp
// MI355X: 4 waves/SIMD: 4V 0.990 ms, 4V+2S 1.160, 4V+4S 1.335, 4S alone 0.992; 8 waves/SIMD: 1.846 / 1.990 / 2.255 / 1.920 -- a scalar instruction costs 0.2-0.35 of a vector slot.
#include <hip/hip_runtime.h>
#include <cstdio>
// per iteration: NV VALU ops (dependent chain per wave) and NS SALU ops (dependent chain, wave-uniform)
template <int NV, int NS> __global__ __launch_bounds__(256) void k(unsigned *out, unsigned b, int iters)
{
	unsigned v = threadIdx.x * 2654435761u + 1;
	unsigned s = __builtin_amdgcn_readfirstlane(blockIdx.x * 77u + b);
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int q = 0; q < 8; q++) {
#pragma unroll
			for (int j = 0; j < NV; j++) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v) : "s"(b));
#pragma unroll
			for (int j = 0; j < NS; j++) asm volatile("s_xor_b32 %0, %0, %1" : "+s"(s) : "s"(b) : "scc");
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = v ^ s;
}
template <int NV, int NS> float run(unsigned *out, int grid, int iters)
{
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	float best = 1e9;
	for (int rep = 0; rep < 3; rep++) {
		fprintf(stderr, "launch %d %d\n", NV, NS);
		hipEventRecord(e0);
		k<NV, NS><<<grid, 256>>>(out, 0x9E3779B1u, iters);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
	}
	return best;
}
int main()
{
	hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
	fprintf(stderr, "cus %d\n", p.multiProcessorCount);
	unsigned *out; hipMalloc(&out, (size_t)p.multiProcessorCount * 16 * 256 * 4);
	const int iters = 4000;
	for (int wpc : {4, 8}) {                 // workgroups of 4 waves per CU -> wpc waves per SIMD... (1 WG = 1 wave per SIMD)
		const int grid = p.multiProcessorCount * wpc;
		const double per_simd_instr_slots = (double)wpc * iters * 8;      // x (NV + NS) instructions per SIMD
		fprintf(stderr, "wpc %d\n", wpc);
		float a = run<4, 0>(out, grid, iters), b2 = run<4, 2>(out, grid, iters), c = run<4, 4>(out, grid, iters), d = run<0, 4>(out, grid, iters), e = run<2, 4>(out, grid, iters);
		auto cyc = [&](float ms, int n) { return ms * 1e-3 * 2.4e9 / (per_simd_instr_slots * n); };
		printf("%d waves/SIMD: 4V+0S %.3f ms (%.2f cyc/instr) | 4V+2S %.3f (%.2f) | 4V+4S %.3f (%.2f) | 0V+4S %.3f (%.2f) | 2V+4S %.3f (%.2f)\n", wpc,
		       a, cyc(a, 4), b2, cyc(b2, 6), c, cyc(c, 8), d, cyc(d, 4), e, cyc(e, 6));
	}
	return 0;
}
