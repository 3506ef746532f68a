// What does one wave instruction of each kind in the trim scan's step cost?  Four waves per SIMD, every wave runs 8 independent
// chains of the instruction under test; cycles per instruction per SIMD at an assumed 2.4 GHz.
// hipcc --offload-arch=gfx950 -O3 -o tools/ab/oprate tools/micro/oprate_exp.hip && tools/ab/oprate
#include <hip/hip_runtime.h>
#include <cstdio>
#define OPK(NAME, ASM) \
__global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned b, int iters) \
{ \
	unsigned v[8]; \
	for (int q = 0; q < 8; q++) v[q] = threadIdx.x * 2654435761u + q; \
	unsigned w = threadIdx.x ^ b, x = threadIdx.x * 3u; \
	for (int i = 0; i < iters; i++) { \
		_Pragma("unroll") for (int r = 0; r < 4; r++) { \
			_Pragma("unroll") for (int q = 0; q < 8; q++) asm volatile(ASM : "+v"(v[q]) : "s"(b), "v"(w), "v"(x)); \
		} \
	} \
	unsigned s = 0; \
	for (int q = 0; q < 8; q++) s ^= v[q]; \
	out[blockIdx.x * blockDim.x + threadIdx.x] = s; \
}
OPK(k_add, "v_add_u32 %0, %0, %2")
OPK(k_add_s, "v_add_u32 %0, %1, %0")
OPK(k_dot4_s, "v_dot4_u32_u8 %0, %2, %1, %0")
OPK(k_dot4_v, "v_dot4_u32_u8 %0, %2, %3, %0")
OPK(k_sad, "v_sad_u8 %0, %2, %1, %0")
OPK(k_lshl_add, "v_lshl_add_u32 %0, %0, 11, %1")
OPK(k_max3, "v_max3_i32 %0, %0, %2, %3")
OPK(k_min3, "v_min3_i32 %0, %0, %2, %3")
OPK(k_align, "v_alignbyte_b32 %0, %0, %2, %3")
OPK(k_perm, "v_perm_b32 %0, %0, %2, %3")
OPK(k_mad24, "v_mad_u32_u24 %0, %0, %2, %3")
OPK(k_add3, "v_add3_u32 %0, %0, %2, %3")
OPK(k_max, "v_max_i32 %0, %0, %2")
OPK(k_pkadd, "v_pk_add_u16 %0, %0, %2")
OPK(k_pkmin, "v_pk_min_i16 %0, %0, %2")
OPK(k_pkmax, "v_pk_max_i16 %0, %0, %2")
OPK(k_pkmad, "v_pk_mad_u16 %0, %0, %2, %3")
OPK(k_mullo, "v_mul_lo_u32 %0, %0, %2")
OPK(k_mqsad, "v_sad_u16 %0, %2, %3, %0")
typedef void (*kfn)(unsigned *, unsigned, int);
int main()
{
	hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
	unsigned *out; hipMalloc(&out, (size_t)p.multiProcessorCount * 16 * 256 * 4);
	const int iters = 2000;
	struct { const char *n; kfn f; } ks[] = {{"v_add_u32 vv", k_add}, {"v_add_u32 sv", k_add_s}, {"v_dot4_u32_u8 (sgpr weights)", k_dot4_s}, {"v_dot4_u32_u8 (vgpr weights)", k_dot4_v},
		{"v_sad_u8", k_sad}, {"v_lshl_add_u32", k_lshl_add}, {"v_max3_i32", k_max3}, {"v_min3_i32", k_min3}, {"v_alignbyte_b32", k_align}, {"v_perm_b32", k_perm},
		{"v_mad_u32_u24", k_mad24}, {"v_add3_u32", k_add3}, {"v_max_i32", k_max}, {"v_pk_add_u16", k_pkadd}, {"v_pk_min_i16", k_pkmin}, {"v_pk_max_i16", k_pkmax},
		{"v_pk_mad_u16", k_pkmad}, {"v_mul_lo_u32", k_mullo}, {"v_sad_u16", k_mqsad}};
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int wpc : {4, 8}) {
		const int grid = p.multiProcessorCount * wpc;
		for (auto &k : ks) {
			float best = 1e9;
			for (int rep = 0; rep < 3; rep++) {
				hipEventRecord(e0);
				k.f<<<grid, 256>>>(out, 0x01010101u, iters);
				hipEventRecord(e1); hipEventSynchronize(e1);
				float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
			}
			const double instr_per_simd = (double)wpc * iters * 32;
			printf("%d waves/SIMD  %-32s %.3f ms  %.2f cycles per wave instruction\n", wpc, k.n, best, best * 1e-3 * 2.4e9 / instr_per_simd);
		}
	}
	return 0;
}
