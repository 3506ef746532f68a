// Issue rate of a few VALU instructions on gfx950 (what bounds the trim scan?): every wave runs a long chain-free loop of
// one instruction kind; reported: instructions per cycle and SIMD, from the kernel's duration at the measured clock.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate_exp tools/valu_rate_exp.hip && /tmp/valu_rate_exp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32;

template <int KIND> __global__ __launch_bounds__(256) void k(u32 *out, int iters, u32 seed)
{
	u32 a[8];
	for (int i = 0; i < 8; i++) a[i] = seed * (threadIdx.x + 1 + i);
	u32 b = seed ^ 0x01010101u;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 4; r++) {
#pragma unroll
			for (int i = 0; i < 8; i++) {
				if (KIND == 0) a[i] = a[i] + b;                                                     // v_add_u32
				if (KIND == 1) a[i] = __builtin_amdgcn_udot4(a[i], 0x01010100u, b, false);          // v_dot4_u32_u8
				if (KIND == 2) a[i] = (a[i] << 11) + b;                                             // v_lshl_add_u32
				if (KIND == 3) a[i] = min(min(a[i], b), a[(i + 1) & 7]);                            // v_min3_u32
				if (KIND == 4) a[i] = __builtin_amdgcn_alignbyte(a[i], b, 1);                       // v_alignbyte_b32
				if (KIND == 5) a[i] = __builtin_amdgcn_perm(a[i], b, 0x03010200u);                  // v_perm_b32
				if (KIND == 6) a[i] = a[i] * b;                                                     // v_mul_lo_u32
				if (KIND == 7) a[i] = __builtin_amdgcn_sad_u8(a[i], b, a[i]);                       // v_sad_u8
				if (KIND == 8) a[i] = (a[i] > b) ? a[i] : a[(i + 1) & 7];                           // v_cmp + v_cndmask
			}
		}
	}
	u32 s = 0;
	for (int i = 0; i < 8; i++) s ^= a[i];
	if (s == 0x12345678u) out[0] = s;
}

int main()
{
	u32 *d;
	CK(hipMalloc(&d, 64));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	hipDeviceProp_t p;
	CK(hipGetDeviceProperties(&p, 0));
	const double ghz = p.clockRate / 1e6;
	const int grid = p.multiProcessorCount * 8, iters = 20000;      // 8 workgroups of 4 waves per CU: 8 waves per SIMD
	static const char *names[] = {"v_add_u32", "v_dot4_u32_u8", "v_lshl_add_u32", "v_min3_u32", "v_alignbyte_b32", "v_perm_b32", "v_mul_lo_u32", "v_sad_u8", "v_cmp + v_cndmask"};
	for (int kind = 0; kind < 9; kind++) {
		float best = 1e9f;
		for (int rep = 0; rep < 3; rep++) {
			CK(hipEventRecord(e0));
			switch (kind) {
			case 0: k<0><<<grid, 256>>>(d, iters, 3u + rep); break;
			case 1: k<1><<<grid, 256>>>(d, iters, 3u + rep); break;
			case 2: k<2><<<grid, 256>>>(d, iters, 3u + rep); break;
			case 3: k<3><<<grid, 256>>>(d, iters, 3u + rep); break;
			case 4: k<4><<<grid, 256>>>(d, iters, 3u + rep); break;
			case 5: k<5><<<grid, 256>>>(d, iters, 3u + rep); break;
			case 6: k<6><<<grid, 256>>>(d, iters, 3u + rep); break;
			case 7: k<7><<<grid, 256>>>(d, iters, 3u + rep); break;
			case 8: k<8><<<grid, 256>>>(d, iters, 3u + rep); break;
			}
			CK(hipEventRecord(e1));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			if (ms < best) best = ms;
		}
		// wave instructions per SIMD: 8 waves x iters x 32 (two for the compare + select kind)
		const double instr = 8.0 * iters * 32 * (kind == 8 ? 2 : 1);
		const double cycles = best * 1e-3 * ghz * 1e9;
		printf("%-20s %8.3f ms  %.3f wave instructions per cycle and SIMD (%.2f cycles each) at %.2f GHz\n", names[kind], best, instr / cycles, cycles / instr, ghz);
	}
	return 0;
}
