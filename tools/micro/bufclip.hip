// micro-test: range-check granularity of raw buffer loads/stores on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ void k(const unsigned char* in, unsigned char* out, int nrec_ld, int nrec_st, int nrec_b)
{
	__amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, nrec_ld, 0x00020000);
	__amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, nrec_st, 0x00020000);
	__amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, nrec_b, 0x00020000);
	int off = threadIdx.x * 16;
	u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rl, off, 0, 2);
	__builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 2);
	// byte stores of the three bytes after the last whole dword
	int b0 = nrec_st;
	for (int i = 0; i < 3; i++)
		__builtin_amdgcn_raw_buffer_store_b8((unsigned char)(0xA0 + i), rb, b0 + i, 0, 0);
}
int main()
{
	const int N = 64 * 16;
	std::vector<unsigned char> h(N), o(N);
	for (int i = 0; i < N; i++) h[i] = (unsigned char)(i * 7 + 1);
	unsigned char *din, *dout;
	hipMalloc(&din, N); hipMalloc(&dout, N);
	hipMemcpy(din, h.data(), N, hipMemcpyHostToDevice);
	int cases[][3] = {{N, N, N}, {100, 100, 100}, {104, 100, 103}, {37, 36, 38}, {0, 0, 0}};
	for (auto &c : cases) {
		hipMemset(dout, 0xEE, N);
		k<<<1, 64>>>(din, dout, c[0], c[1], c[2]);
		hipMemcpy(o.data(), dout, N, hipMemcpyDeviceToHost);
		int last_copied = -1, first_bad = -1;
		for (int i = 0; i < N; i++) { if (o[i] == h[i]) last_copied = i; else if (first_bad < 0) first_bad = i; }
		printf("ld_rec=%d st_rec=%d byte_rec=%d : first non-copied byte %d, last copied %d; bytes[st_rec..+3]= %02x %02x %02x %02x\n",
		       c[0], c[1], c[2], first_bad, last_copied, o[c[1] < N ? c[1] : N - 1], o[c[1] + 1 < N ? c[1] + 1 : N - 1], o[c[1] + 2 < N ? c[1] + 2 : N - 1], o[c[1] + 3 < N ? c[1] + 3 : N - 1]);
	}
	return 0;
}
