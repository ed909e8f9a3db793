/*
 * renju_static.hpp — the stand-alone renju foul test used when a game outcome is decided (no pattern calculator around).
 *
 * Restates isForbidden (src/game/rules.cpp:134-173): the move of the cross (black) player is a foul when it makes an overline,
 * a 4x4 fork, or a 3x3 fork in which at least two open threes are "real" — an open three is real when one of its promotion
 * cells gives a straight four and is not itself a foul (the recursion of the reference, run here on an explicit stack over one
 * board that is restored before returning).  Shared by the device (k_advance, one thread) and the host (agx_make_opening).
 */
#ifndef AGX_RENJU_STATIC_HPP_
#define AGX_RENJU_STATIC_HPP_

#include <cstdint>

#if defined(__HIPCC__)
#define AGX_HD __host__ __device__
#else
#define AGX_HD
#endif

namespace agx
{
	struct RenjuFrame
	{
			int cell;
			uint32_t raw[4];
			uint8_t pt[4];
			int8_t d, i;
			uint8_t saved, real;
	};
	constexpr int RENJU_STATIC_DEPTH = 12;

	AGX_HD inline int rs_row_step(int d) { return d == 0 ? 0 : 1; }
	AGX_HD inline int rs_col_step(int d) { return d == 0 ? 1 : (d == 1 ? 0 : (d == 2 ? 1 : -1)); }
	AGX_HD inline uint32_t rs_line(const uint8_t *board, int n, int r, int c, int d, uint32_t centre_or, bool clear_centre)
	{ // RawPatternCalculator.hpp:114-141: 11 cells, 2 bits each, 3 outside the board
		uint32_t result = 0;
		for (int k = -5, shf = 0; k <= 5; k++, shf += 2)
		{
			const int rr = r + k * rs_row_step(d), cc = c + k * rs_col_step(d);
			uint32_t v = (rr >= 0 && rr < n && cc >= 0 && cc < n) ? board[rr * n + cc] : 3u;
			if (k == 0)
				v = clear_centre ? 0u : (v | centre_or);
			result |= v << shf;
		}
		return result;
	}
	AGX_HD inline uint32_t rs_promotion_moves(uint32_t pattern)
	{ // getOpenThreePromotionMoves (DefensiveMoveTable.cpp:329-377): first matching shape wins
		const uint32_t patterns[12] = { 320u, 4352u, 20480u, 80u, 16640u, 69632u, 272u, 4160u, 81920u, 320u, 4352u, 20480u };
		const uint32_t masks[12] = { 65520u, 262080u, 1048320u, 16380u, 262080u, 1048320u, 16380u, 65520u, 1048320u, 16380u, 65520u, 262080u };
		const uint32_t results[12] = { 196u, 392u, 784u, 82u, 328u, 656u, 74u, 148u, 592u, 70u, 140u, 280u };
		for (int k = 0; k < 12; k++)
			if ((pattern & masks[k]) == patterns[k])
				return results[k];
		return 0;
	}
	AGX_HD inline int rs_threat(const uint8_t *t_threat, const uint8_t *pt)
	{
		return t_threat[2 * (pt[0] | (pt[1] << 3) | (pt[2] << 6) | (pt[3] << 9))];
	}
	/* t_pattern / t_threat: the RENJU tables (HostTables::pattern / ::threat layout).  `board` is restored before returning. */
	AGX_HD inline bool renju_static_foul(const uint8_t *t_pattern, const uint8_t *t_threat, uint8_t *board, int n, int cell0)
	{
		RenjuFrame st[RENJU_STATIC_DEPTH];
		int sp = 0;
		bool ret = false;
		bool entering = true;
		st[0].cell = cell0;
		while (true)
		{
			RenjuFrame &fr = st[sp];
			const int r = fr.cell / n, c = fr.cell % n;
			if (entering)
			{
				entering = false;
				fr.saved = board[fr.cell];
				for (int d = 0; d < 4; d++)
				{
					fr.raw[d] = rs_line(board, n, r, c, d, 0u, true);
					fr.pt[d] = t_pattern[(fr.raw[d] & 1023u) | ((fr.raw[d] & 4190208u) >> 2)] & 15;
				}
				const int tt = rs_threat(t_threat, fr.pt);
				if (tt != 3)
				{
					ret = (tt == 9 || tt == 6);
					if (sp == 0)
						return ret;
					sp--;
					st[sp].real = ret ? st[sp].real : 1;
					st[sp].i++;
					continue;
				}
				board[fr.cell] = 0;
				fr.d = 0;
				fr.i = -6;
			}
			bool descended = false;
			while (fr.d < 4 && !descended)
			{
				if (fr.pt[fr.d] == 2)
				{
					if (fr.i == -6)
					{
						board[fr.cell] = 1;
						fr.i = -5;
						fr.real = 0;
					}
					const uint32_t promotion = rs_promotion_moves(fr.raw[fr.d]);
					while (fr.i <= 5 && !fr.real)
					{
						const int i = fr.i;
						if (i != 0 && ((promotion >> (5 + i)) & 1))
						{
							const int rr = r + i * rs_row_step(fr.d), cc = c + i * rs_col_step(fr.d);
							if (board[rr * n + cc] == 0)
							{
								uint32_t line = rs_line(board, n, rr, cc, fr.d, 1u, false);
								bool straight = false;
								for (int k = 0; k < 7; k++, line >>= 2)
									if ((line & 255u) == 85u)
										straight = true;
								if (straight)
								{
									if (sp + 1 >= RENJU_STATIC_DEPTH)
									{ // deeper than any real position goes: treat the promotion cell as a foul
										fr.i++;
										continue;
									}
									st[sp + 1].cell = rr * n + cc;
									sp++;
									entering = true;
									descended = true;
									break;
								}
							}
						}
						fr.i++;
					}
					if (descended)
						break;
					board[fr.cell] = 0;
					if (!fr.real)
						fr.pt[fr.d] = 0;
				}
				fr.d++;
				fr.i = -6;
			}
			if (descended)
				continue;
			const int tt = rs_threat(t_threat, fr.pt);
			ret = (tt == 9 || tt == 6 || tt == 3);
			board[fr.cell] = fr.saved;
			if (sp == 0)
				return ret;
			sp--;
			st[sp].real = ret ? st[sp].real : 1;
			st[sp].i++;
		}
	}
}

#endif
