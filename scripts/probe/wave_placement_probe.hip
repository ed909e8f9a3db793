// Dev probe: where the one-wave workgroups of a persistent launch land, in dispatch order (k_search_spec's shape: 1024 x 64 threads, 10 240 B of LDS, 128 registers worth of
// occupancy, on a 64-unit CU mask) — do the first 256 to start (the ones that take the games off the select cursor) spread over the compute units or fill them one by one?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>
__global__ __launch_bounds__(64, 4) void land(unsigned long long *out, int *cursor, int spin)
{
	extern __shared__ char lds[];
	unsigned xcc, hw;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
	int order = 0;
	if (threadIdx.x == 0)
		order = atomicAdd(cursor, 1); // the order in which the waves reach the cursor
	lds[threadIdx.x] = 1;
	long long c0 = clock64();
	while (clock64() - c0 < spin) { }
	if (threadIdx.x == 0)
		out[blockIdx.x] = (static_cast<unsigned long long>(order) << 32) | ((xcc & 15u) << 16) | (hw & 0xFFFFu);
}
int main(int argc, char **argv)
{
	const int first = (argc > 1) ? atoi(argv[1]) : 0, count = (argc > 2) ? atoi(argv[2]) : 64, waves = (argc > 3) ? atoi(argv[3]) : 1024;
	std::vector<uint32_t> mask(8, 0u);
	for (int c = first; c < first + count; c++) mask[c / 32] |= 1u << (c % 32);
	hipStream_t s;
	if (hipExtStreamCreateWithCUMask(&s, 8, mask.data()) != hipSuccess) { printf("mask stream failed\n"); return 1; }
	unsigned long long *d; int *cur;
	(void) hipMalloc(&d, waves * 8); (void) hipMalloc(&cur, 4); (void) hipMemset(cur, 0, 4);
	hipLaunchKernelGGL(land, dim3(waves), dim3(64), 10240, s, d, cur, 400000);
	(void) hipStreamSynchronize(s);
	std::vector<unsigned long long> h(waves); (void) hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost);
	// per compute unit (xcc, se, cu): how many of the first quarter of the waves (by cursor order) it holds
	std::map<int, int> firsts, all;
	int same_block = 0;
	for (int b = 0; b < waves; b++)
	{
		const int order = static_cast<int>(h[b] >> 32);
		const unsigned v = static_cast<unsigned>(h[b]);
		const int cu = ((v >> 16) << 8) | ((v >> 8) & 0xFF);
		all[cu]++;
		if (order < waves / 4) { firsts[cu]++; if (b < waves / 4) same_block++; }
	}
	std::map<int, int> hist;
	for (auto &kv : all) hist[firsts.count(kv.first) ? firsts[kv.first] : 0]++;
	printf("%d waves on %zu compute units; of the first %d to reach the cursor %d are blocks 0..%d; per unit they number:", waves, all.size(), waves / 4, same_block, waves / 4 - 1);
	for (auto &kv : hist) printf("  %d on %d units", kv.first, kv.second);
	printf("\n");
	return 0;
}
