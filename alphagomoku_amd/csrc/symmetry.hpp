/*
 * symmetry.hpp — the 8 board symmetries of the network input, for host and device code.
 * utils/augmentations.hpp:31-53,62-216 (square boards): source cell of destination (r, c) under symmetry s, the inverse map;
 * NNInputFeatures::augment (src/networks/NNInputFeatures.cpp:33-50,114-154): the per-direction feature bits follow the board symmetry.
 */
#ifndef AGX_SYMMETRY_HPP_
#define AGX_SYMMETRY_HPP_

#include <cstdint>

#if defined(__HIPCC__)
#define AGX_SYM_HD __host__ __device__ __forceinline__
#else
#define AGX_SYM_HD inline
#endif

namespace agx
{
	AGX_SYM_HD void symmetry_source(int s, int n, int r, int c, int &sr, int &sc)
	{
		const int last = n - 1;
		switch (s)
		{
			default: sr = r; sc = c; break;              // IDENTITY
			case 1: sr = last - r; sc = c; break;        // FLIP_VERTICALLY
			case 2: sr = r; sc = last - c; break;        // FLIP_HORIZONTALLY
			case 3: sr = last - r; sc = last - c; break; // ROTATE_180
			case 4: sr = c; sc = r; break;               // FLIP_DIAGONALLY
			case 5: sr = last - c; sc = last - r; break; // FLIP_ANTIDIAGONALLY
			case 6: sr = c; sc = last - r; break;        // ROTATE_90
			case 7: sr = last - c; sc = r; break;        // ROTATE_270
		}
	}
	AGX_SYM_HD int inverse_symmetry(int s) { return (s == 6) ? 7 : ((s == 7) ? 6 : s); }
	AGX_SYM_HD uint32_t shuffle_feature_directions(uint32_t data, int s)
	{
		int d0, d1, d2, d3;
		switch (s)
		{
			case 1: case 2: d0 = 0; d1 = 1; d2 = 3; d3 = 2; break;
			case 4: case 5: d0 = 1; d1 = 0; d2 = 2; d3 = 3; break;
			case 6: case 7: d0 = 1; d1 = 0; d2 = 3; d3 = 2; break;
			default: return data;
		}
		const uint32_t mask = (1u << 8) | (1u << 12) | (1u << 20) | (1u << 24);
		uint32_t result = data & 0xF00F00FFu;
		result |= ((data >> d0) & mask) << 0;
		result |= ((data >> d1) & mask) << 1;
		result |= ((data >> d2) & mask) << 2;
		result |= ((data >> d3) & mask) << 3;
		return result;
	}
}

#endif
