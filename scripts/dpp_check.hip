#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include "../alphagomoku_amd/csrc/agx_internal.hpp"
#include "../alphagomoku_amd/csrc/engine_types.hpp"
#include "../alphagomoku_amd/csrc/dev_mcts.hpp"
using namespace agx; using namespace agx::dev;
__global__ void k(const uint32_t *in, const float *fin, uint32_t *out, float *fout, int *iout)
{
	const int lane = threadIdx.x;
	const uint32_t v = in[blockIdx.x * 64 + lane];
	out[(blockIdx.x * 5 + 0) * 64 + lane] = wave_reduce_umax(v);
	out[(blockIdx.x * 5 + 1) * 64 + lane] = wave_reduce_add(v & 0xFFFF);
	const u64 x = wave_reduce_xor64((static_cast<u64>(v) << 32) | (v * 2654435761u));
	out[(blockIdx.x * 5 + 2) * 64 + lane] = static_cast<uint32_t>(x);
	out[(blockIdx.x * 5 + 3) * 64 + lane] = static_cast<uint32_t>(x >> 32);
	out[(blockIdx.x * 5 + 4) * 64 + lane] = wave_scan32_add(v & 31);
	float val = fin[blockIdx.x * 64 + lane];
	int idx = (blockIdx.x % 3 == 0 && lane > 40) ? 0x7FFFFFFF : lane * 3 + 1;
	if (idx == 0x7FFFFFFF) val = -3.402823466e+38f;
	wave_argmax(val, idx);
	fout[blockIdx.x * 64 + lane] = val;
	iout[blockIdx.x * 64 + lane] = idx;
}
int main()
{
	const int B = 200;
	std::mt19937 rng(1);
	std::vector<uint32_t> in(B * 64), out(B * 5 * 64);
	std::vector<float> fin(B * 64), fout(B * 64);
	std::vector<int> iout(B * 64);
	for (auto &x : in) x = rng();
	for (int i = 0; i < B * 64; i++) fin[i] = (i / 64 % 2) ? float(int(rng() % 7) - 3) : (float(rng() % 100000) / 1000.0f - 50.0f);
	uint32_t *din, *dout; float *dfin, *dfout; int *diout;
	hipMalloc(&din, in.size() * 4); hipMalloc(&dout, out.size() * 4); hipMalloc(&dfin, fin.size() * 4); hipMalloc(&dfout, fout.size() * 4); hipMalloc(&diout, iout.size() * 4);
	hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dfin, fin.data(), fin.size() * 4, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(B), dim3(64), 0, 0, din, dfin, dout, dfout, diout);
	hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(fout.data(), dfout, fout.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(iout.data(), diout, iout.size() * 4, hipMemcpyDeviceToHost);
	int bad[6] = {0,0,0,0,0,0};
	for (int b = 0; b < B; b++)
	{
		uint32_t mx = 0, sum = 0; uint64_t xr = 0;
		for (int l = 0; l < 64; l++) { const uint32_t v = in[b * 64 + l]; mx = std::max(mx, v); sum += v & 0xFFFF; xr ^= (uint64_t(v) << 32) | uint32_t(v * 2654435761u); }
		float bv = -3.402823466e+38f; int bi = 0x7FFFFFFF;
		for (int l = 0; l < 64; l++)
		{
			int idx = (b % 3 == 0 && l > 40) ? 0x7FFFFFFF : l * 3 + 1; float val = (idx == 0x7FFFFFFF) ? -3.402823466e+38f : fin[b * 64 + l];
			if (val > bv || (val == bv && idx < bi)) { bv = val; bi = idx; }
		}
		uint32_t run = 0;
		for (int l = 0; l < 64; l++)
		{
			if (out[(b * 5 + 0) * 64 + l] != mx) bad[0]++;
			if (out[(b * 5 + 1) * 64 + l] != sum) bad[1]++;
			if (out[(b * 5 + 2) * 64 + l] != uint32_t(xr) || out[(b * 5 + 3) * 64 + l] != uint32_t(xr >> 32))
			{
				if (bad[2] < 3 && l == 0)
					printf("block %d: want %08x %08x got %08x %08x\n", b, uint32_t(xr), uint32_t(xr >> 32), out[(b * 5 + 2) * 64 + l], out[(b * 5 + 3) * 64 + l]);
				bad[2]++;
			}
			if (l < 32) { run += in[b * 64 + l] & 31; if (out[(b * 5 + 4) * 64 + l] != run) bad[3]++; }
			if (fout[b * 64 + l] != bv) bad[4]++;
			if (iout[b * 64 + l] != bi) bad[5]++;
		}
	}
	printf("bad: umax %d add %d xor %d scan %d argmax value %d index %d\n", bad[0], bad[1], bad[2], bad[3], bad[4], bad[5]);
	return 0;
}
