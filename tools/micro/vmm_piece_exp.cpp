// What does mapping memory into a reserved address range cost, by the size of a piece?  (sk_bamfile.cpp maps its output range piece by piece.)
// build: hipcc -O2 -o /tmp/vmm_piece_exp tools/micro/vmm_piece_exp.cpp ; run on the GPU box
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	hipSetDevice(0);
	void *warm = nullptr; hipMalloc(&warm, 1 << 20); hipFree(warm);
	hipMemAllocationProp prop{};
	prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
	size_t gran = 0;
	hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
	printf("granularity %zu\n", gran);
	const size_t total = (size_t)6 << 30;
	for (size_t piece : {(size_t)64 << 20, (size_t)256 << 20, (size_t)512 << 20, (size_t)1 << 30, (size_t)2 << 30, (size_t)6 << 30}) {
		for (int rep = 0; rep < 2; rep++) {
			void *va = nullptr;
			if (hipMemAddressReserve(&va, total, 0, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); return 1; }
			std::vector<hipMemGenericAllocationHandle_t> hs;
			double t_create = 0, t_map = 0, t_acc = 0;
			const double t0 = now_ms();
			for (size_t off = 0; off < total; off += piece) {
				hipMemGenericAllocationHandle_t h;
				double a = now_ms();
				if (hipMemCreate(&h, piece, &prop, 0) != hipSuccess) { printf("create failed\n"); return 1; }
				double b = now_ms();
				hipMemMap((char *)va + off, piece, 0, h, 0);
				double c = now_ms();
				hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
				hipMemSetAccess((char *)va + off, piece, &acc, 1);
				double d = now_ms();
				t_create += b - a; t_map += c - b; t_acc += d - c;
				hs.push_back(h);
			}
			const double t1 = now_ms();
			hipMemset(va, 1, total); hipDeviceSynchronize();
			const double t2 = now_ms();
			for (size_t i = 0; i < hs.size(); i++) { hipMemUnmap((char *)va + i * piece, piece); hipMemRelease(hs[i]); }
			hipMemAddressFree(va, total);
			const double t3 = now_ms();
			printf("piece %5zu MiB: 6 GiB mapped in %7.1f ms (create %.1f, map %.1f, set access %.1f); first memset %.1f ms; unmap + release %.1f ms\n", piece >> 20, t1 - t0, t_create, t_map, t_acc, t2 - t1, t3 - t2);
		}
	}
	double t0 = now_ms();
	void *p = nullptr; hipMalloc(&p, total);
	double t1 = now_ms();
	hipMemset(p, 1, total); hipDeviceSynchronize();
	double t2 = now_ms();
	hipFree(p);
	printf("hipMalloc of 6 GiB: %.1f ms; first memset %.1f ms; hipFree %.1f ms\n", t1 - t0, t2 - t1, now_ms() - t2);
	return 0;
}
