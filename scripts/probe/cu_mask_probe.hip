// Dev probe: which compute units (XCC, SE, CU) the workgroups of a CU-masked stream land on, and how many workgroups each XCC is handed.
// usage: cu_mask_probe FIRST COUNT [STRIDE_BLOCK PER_BLOCK]  — mask bits [FIRST, FIRST+COUNT), or PER_BLOCK bits at the start of every STRIDE_BLOCK bits
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <set>
#include <algorithm>
__global__ void where(unsigned *out, int spin)
{
	unsigned xcc, hw;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
	long long t0 = clock64();
	while (clock64() - t0 < spin) { }
	if (threadIdx.x == 0)
		out[blockIdx.x] = ((xcc & 15u) << 16) | (hw & 0xFFFFu);
}
__global__ void persistent(unsigned long long *out, int spin)
{
	extern __shared__ char lds[];
	unsigned xcc, hw;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
	const unsigned long long t0 = wall_clock64();
	lds[threadIdx.x] = 1;
	long long c0 = clock64();
	while (clock64() - c0 < spin) { }
	if (threadIdx.x == 0) { out[2 * blockIdx.x] = t0; out[2 * blockIdx.x + 1] = ((xcc & 15u) << 16) | (hw & 0xFFFFu); }
}
int main(int argc, char **argv)
{
	const int first = atoi(argv[1]), count = atoi(argv[2]);
	const int stride = (argc > 4) ? atoi(argv[3]) : 0, per = (argc > 4) ? atoi(argv[4]) : 0;
	std::vector<uint32_t> mask(8, 0u);
	if (stride > 0)
	{ for (int b = 0; b < 256; b += stride) for (int i = 0; i < per; i++) mask[(b + first + i) / 32] |= 1u << ((b + first + i) % 32); }
	else
		for (int c = first; c < first + count; c++) mask[c / 32] |= 1u << (c % 32);
	hipStream_t s;
	if (hipExtStreamCreateWithCUMask(&s, 8, mask.data()) != hipSuccess) { printf("mask stream failed\n"); return 1; }
	const int blocks = 8192;
	unsigned *d; hipMalloc(&d, blocks * 4);
	hipLaunchKernelGGL(where, dim3(blocks), dim3(64), 0, s, d, 20000);
	hipStreamSynchronize(s);
	std::vector<unsigned> h(blocks); hipMemcpy(h.data(), d, blocks * 4, hipMemcpyDeviceToHost);
	std::map<int, int> per_xcc; std::map<int, std::set<int>> cus;
	for (unsigned v : h) { const int x = v >> 16; per_xcc[x]++; cus[x].insert((v >> 8) & 0xFF); } // cu_id, sh_id, se_id
	printf("mask"); for (auto w : mask) printf(" %08x", w); printf("\n");
	{ // a persistent grid: one 150-KB-LDS workgroup per enabled compute unit; a workgroup that has to wait for another's unit starts late
		int enabled = 0; for (auto w : mask) enabled += __builtin_popcount(w);
		unsigned long long *t; hipMalloc(&t, enabled * 2 * 8);
		hipFuncSetAttribute((const void*) persistent, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
		hipLaunchKernelGGL(persistent, dim3(enabled), dim3(512), 150 * 1024, s, t, 2000000);
		hipStreamSynchronize(s);
		std::vector<unsigned long long> ht(enabled * 2); hipMemcpy(ht.data(), t, enabled * 16, hipMemcpyDeviceToHost);
		unsigned long long first = ~0ull; for (int i = 0; i < enabled; i++) first = std::min(first, ht[2 * i]);
		int late = 0; std::map<int, int> se_count; std::map<int, int> late_xcc;
		for (int i = 0; i < enabled; i++) { const unsigned v = (unsigned) ht[2 * i + 1]; const bool l = (ht[2 * i] - first) > 50000; late += l; if (l) late_xcc[v >> 16]++; se_count[((v >> 16) << 4) | ((v >> 13) & 7)]++; }
		printf("  persistent grid of %d workgroups: %d started late;", enabled, late);
		for (auto &kv : late_xcc) printf(" xcc%d:%d", kv.first, kv.second);
		printf("\n  workgroups per (xcc, se):"); for (auto &kv : se_count) printf(" %d.%d=%d", kv.first >> 4, kv.first & 15, kv.second); printf("\n");
	}
	for (auto &kv : per_xcc) printf("  xcc %d: %d workgroups on %zu distinct (se, sh, cu)\n", kv.first, kv.second, cus[kv.first].size());
	return 0;
}
