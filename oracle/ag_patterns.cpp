/*
 * oracle/ag_patterns.cpp — TEST INFRASTRUCTURE ONLY.  Incremental pattern/threat state, rules, NN input features.
 */
#include "agoracle.hpp"

namespace ago
{
	Calc::Calc(GameConfig c) : cfg(c), tab(&Tables::get(c.rules))
	{
		assert(c.rows <= MAXN && c.cols <= MAXN);
		std::memset(board, 0, sizeof(board));
		std::memset(ptype, 0, sizeof(ptype));
		std::memset(threat, 0, sizeof(threat));
		std::memset(legal, 0, sizeof(legal));
	}
	uint32_t Calc::raw_pattern(int r, int c, Direction d, int pad) const
	{ // RawPatternCalculator.hpp:62-78,196-210: window of 2*pad+1 cells, 2 bits each, off-board = 3, lowest bits = most negative offset
		uint32_t result = 0;
		const int dr = row_step(d), dc = col_step(d);
		for (int i = -pad, sh = 0; i <= pad; i++, sh += 2)
		{
			const int rr = r + i * dr, cc = c + i * dc;
			const uint32_t v = inside(rr, cc) ? board[idx(rr, cc)] : 3u;
			result |= v << sh;
		}
		return result;
	}
	void Calc::set_board(const Sign *b, Sign to_move)
	{ // PatternCalculator.cpp:40-66 (+ classify_feature_types :245-260, prepare_threat_lists :261-277)
		sign_to_move = to_move;
		depth = 0;
		for (int r = 0; r < cfg.rows; r++)
		{
			legal[r] = 0;
			for (int c = 0; c < cfg.cols; c++)
			{
				board[idx(r, c)] = b[idx(r, c)];
				if (b[idx(r, c)] == NONE)
					legal[r] |= (1u << c);
				else
					depth++;
			}
		}
		for (int s = 0; s < 2; s++)
			for (int t = 0; t < 10; t++)
				hist[s][t].v.clear();
		for (int r = 0; r < cfg.rows; r++)
			for (int c = 0; c < cfg.cols; c++)
			{
				const int i = idx(r, c);
				if (board[i] == NONE)
				{
					for (Direction d = 0; d < 4; d++)
					{
						const uint32_t p = normal_pattern(r, c, d);
						ptype[i][0][d] = tab->pattern_type(p, CROSS);
						ptype[i][1][d] = tab->pattern_type(p, CIRCLE);
					}
					threat[i][0] = tab->threat(ptype[i][0], CROSS);
					threat[i][1] = tab->threat(ptype[i][1], CIRCLE);
				}
				else
				{
					std::memset(ptype[i], 0, sizeof(ptype[i]));
					threat[i][0] = threat[i][1] = T_NONE;
				}
			}
		for (int r = 0; r < cfg.rows; r++)
			for (int c = 0; c < cfg.cols; c++)
				if (board[idx(r, c)] == NONE)
				{ // ThreatHistogram::add ignores NONE (ThreatHistogram.hpp:101-111)
					if (threat[idx(r, c)][0] != T_NONE)
						hist[0][threat[idx(r, c)][0]].add(Loc(r, c));
					if (threat[idx(r, c)][1] != T_NONE)
						hist[1][threat[idx(r, c)][1]].add(Loc(r, c));
				}
	}
	void Calc::add_move(Move m)
	{ // PatternCalculator.cpp:68-86
		board[idx(m.row, m.col)] = m.sign;
		legal[m.row] &= ~(1u << m.col);
		update_around(m.row, m.col, true, Loc());
		sign_to_move = invert_sign(sign_to_move);
		depth++;
	}
	void Calc::undo_move(Move m)
	{ // PatternCalculator.cpp:87-105
		board[idx(m.row, m.col)] = NONE;
		legal[m.row] |= (1u << m.col);
		update_around(m.row, m.col, false, Loc());
		sign_to_move = invert_sign(sign_to_move);
		depth--;
	}
	void Calc::update_around(int r, int c, bool added, Loc)
	{ // PatternCalculator.cpp:278-329.  The reference consults a per-pattern update mask (PatternTable.cpp:166-276) that
	  // only says which of the +-5 cells in each direction CAN change; recomputing every empty cell in range is equivalent
	  // because the threat lists are touched only when a cell's threat type really changes (:346-364).
		const int i = idx(r, c);
		if (added)
		{
			const uint8_t old0 = threat[i][0], old1 = threat[i][1];
			if (old0 != T_NONE)
				hist[0][old0].remove(Loc(r, c));
			if (old1 != T_NONE)
				hist[1][old1].remove(Loc(r, c));
			std::memset(ptype[i], 0, sizeof(ptype[i]));
			threat[i][0] = threat[i][1] = T_NONE;
		}
		else
		{
			for (Direction d = 0; d < 4; d++)
			{
				const uint32_t p = normal_pattern(r, c, d);
				ptype[i][0][d] = tab->pattern_type(p, CROSS);
				ptype[i][1][d] = tab->pattern_type(p, CIRCLE);
			}
			threat[i][0] = tab->threat(ptype[i][0], CROSS);
			threat[i][1] = tab->threat(ptype[i][1], CIRCLE);
			if (threat[i][0] != T_NONE)
				hist[0][threat[i][0]].add(Loc(r, c));
			if (threat[i][1] != T_NONE)
				hist[1][threat[i][1]].add(Loc(r, c));
		}
		for (int k = -5; k <= 5; k++)
			if (k != 0)
				for (Direction d = 0; d < 4; d++)
				{ // order inside one k: horizontal, vertical, diagonal, antidiagonal (:319-327)
					const int rr = r + k * row_step(d), cc = c + k * col_step(d);
					if (inside(rr, cc) && board[idx(rr, cc)] == NONE)
						update_cell(rr, cc, d);
				}
	}
	void Calc::update_cell(int r, int c, Direction d)
	{ // PatternCalculator.cpp:330-367
		const int i = idx(r, c);
		const uint8_t old0 = threat[i][0], old1 = threat[i][1];
		const uint32_t p = normal_pattern(r, c, d);
		ptype[i][0][d] = tab->pattern_type(p, CROSS);
		const uint8_t new0 = tab->threat(ptype[i][0], CROSS);
		threat[i][0] = new0;
		if (old0 != new0)
		{
			if (old0 != T_NONE)
				hist[0][old0].remove(Loc(r, c));
			if (new0 != T_NONE)
				hist[0][new0].add(Loc(r, c));
		}
		ptype[i][1][d] = tab->pattern_type(p, CIRCLE);
		const uint8_t new1 = tab->threat(ptype[i][1], CIRCLE);
		threat[i][1] = new1;
		if (old1 != new1)
		{
			if (old1 != T_NONE)
				hist[1][old1].remove(Loc(r, c));
			if (new1 != T_NONE)
				hist[1][new1].add(Loc(r, c));
		}
	}
	bool Calc::has_any_four(Sign s) const
	{ // ThreatHistogram.hpp:130-134
		const LocList *h = hist[s == CROSS ? 0 : 1];
		return h[T_HALF_OPEN_4].size() > 0 || h[T_FORK_4x3].size() > 0 || h[T_FORK_4x4].size() > 0 || h[T_OPEN_4].size() > 0;
	}
	int Calc::defensive_moves(Sign defender, int r, int c, Direction d, Loc out[6]) const
	{ // PatternCalculator.hpp:150-160
		const uint32_t ext = raw_pattern(r, c, d, 6);
		const PatternType to_defend = static_cast<PatternType>(patterns(invert_sign(defender), r, c)[d]);
		const uint16_t mask = tab->defensive_moves(ext, defender, to_defend);
		int n = 0;
		for (int i = -6; i <= 6; i++)
			if ((mask >> (6 + i)) & 1)
				out[n++] = shift(d, i, Loc(r, c));
		return n;
	}
	bool Calc::is_forbidden(Sign s, int r, int c)
	{ // PatternCalculator.hpp:161-177
		if (cfg.rules == RENJU && s == CROSS)
		{
			if (at(r, c) != NONE)
				return false;
			const ThreatType t = threat_at(CROSS, r, c);
			if (t == T_OVERLINE || t == T_FORK_4x4)
				return true;
			if (t == T_FORK_3x3)
				return is_3x3_forbidden(s, r, c);
		}
		return false;
	}
	bool is_straight_four_at(const Calc &pc, int r, int c, Direction d)
	{ // RawPatternCalculator.hpp:142-178: window +-5 with a cross stone put at the centre contains XXXX
		uint32_t result = 0;
		for (int i = -5, sh = 0; i <= 5; i++, sh += 2)
		{
			const int rr = r + i * row_step(d), cc = c + i * col_step(d);
			uint32_t v = pc.inside(rr, cc) ? pc.board[pc.idx(rr, cc)] : 3u;
			if (i == 0)
				v |= CROSS;
			result |= v << sh;
		}
		for (int i = 0; i < 11 - 4; i++, result >>= 2)
			if ((result & 255u) == 85u)
				return true;
		return false;
	}
	bool Calc::is_3x3_forbidden(Sign s, int r, int c)
	{ // PatternCalculator.cpp:213-244
		int open3_count = 0;
		for (Direction d = 0; d < 4; d++)
			if (patterns(CROSS, r, c)[d] == P_OPEN_3)
			{
				const uint16_t promotion = open_three_promotion_moves(normal_pattern(r, c, d));
				board[idx(r, c)] = CROSS; // Board::putMove only (the calculator state is NOT updated here)
				for (int i = -5; i <= 5; i++)
					if ((promotion >> (5 + i)) & 1)
					{
						const Loc l = shift(d, i, Loc(r, c));
						if (at(l.row, l.col) == NONE && is_straight_four_at(*this, l.row, l.col, d))
						{
							board[idx(r, c)] = NONE;
							add_move(Move(CROSS, r, c));
							const bool forb = is_forbidden(s, l.row, l.col);
							undo_move(Move(CROSS, r, c));
							board[idx(r, c)] = CROSS;
							if (!forb)
							{
								open3_count++;
								break;
							}
						}
					}
				board[idx(r, c)] = NONE;
			}
		return open3_count >= 2;
	}

	/* ---- rules.cpp ---- */
	static uint32_t pattern_on(const Sign *board, int rows, int cols, int r, int c, Direction d, bool clear_center)
	{ // RawPatternCalculator.hpp:114-141
		uint32_t result = 0;
		for (int i = -5, sh = 0; i <= 5; i++, sh += 2)
		{
			const int rr = r + i * row_step(d), cc = c + i * col_step(d);
			uint32_t v = (rr >= 0 && rr < rows && cc >= 0 && cc < cols) ? board[rr * cols + cc] : 3u;
			if (i == 0 && clear_center)
				v = 0;
			result |= v << sh;
		}
		return result;
	}
	bool is_forbidden_static(const Sign *board, int rows, int cols, Move m)
	{ // game/rules.cpp:134-173
		if (m.sign == CIRCLE)
			return false;
		const Tables &tab = Tables::get(RENJU);
		uint32_t raw[4];
		uint8_t pt[4];
		for (Direction d = 0; d < 4; d++)
		{
			raw[d] = pattern_on(board, rows, cols, m.row, m.col, d, true);
			pt[d] = tab.pattern_type(raw[d], CROSS);
		}
		ThreatType tt = tab.threat(pt, CROSS);
		if (tt == T_FORK_3x3)
		{
			std::vector<Sign> tmp(board, board + rows * cols);
			tmp[m.row * cols + m.col] = NONE;
			for (Direction d = 0; d < 4; d++)
				if (pt[d] == P_OPEN_3)
				{
					tmp[m.row * cols + m.col] = m.sign;
					const uint16_t promotion = open_three_promotion_moves(raw[d]);
					bool real = false;
					for (int i = -5; i <= 5 && !real; i++)
						if (i != 0 && ((promotion >> (5 + i)) & 1))
						{
							const Loc l = shift(d, i, m.loc());
							if (tmp[l.row * cols + l.col] == NONE)
							{
								// isStraightFourAt on tmp
								uint32_t line = 0;
								for (int k = -5, sh = 0; k <= 5; k++, sh += 2)
								{
									const int rr = l.row + k * row_step(d), cc = l.col + k * col_step(d);
									uint32_t v = (rr >= 0 && rr < rows && cc >= 0 && cc < cols) ? tmp[rr * cols + cc] : 3u;
									if (k == 0)
										v |= CROSS;
									line |= v << sh;
								}
								bool straight = false;
								for (int k = 0; k < 7; k++, line >>= 2)
									if ((line & 255u) == 85u)
										straight = true;
								if (straight && !is_forbidden_static(tmp.data(), rows, cols, Move(CROSS, l)))
									real = true;
							}
						}
					tmp[m.row * cols + m.col] = NONE;
					if (!real)
						pt[d] = P_NONE;
				}
			tt = tab.threat(pt, CROSS);
		}
		return tt == T_OVERLINE || tt == T_FORK_4x4 || tt == T_FORK_3x3;
	}
	Outcome get_outcome(Rules rules, const Sign *board, int rows, int cols, Move last, int draw_after)
	{ // game/rules.cpp:110-133
		if (!(last.row >= 0 && last.row < rows && last.col >= 0 && last.col < cols))
			return O_UNKNOWN;
		const Tables &tab = Tables::get(rules);
		bool win = false;
		for (Direction d = 0; d < 4; d++)
			if (tab.pattern_type(pattern_on(board, rows, cols, last.row, last.col, d, true), last.sign) == P_FIVE)
				win = true;
		if (win)
			return (last.sign == CROSS) ? O_CROSS_WIN : O_CIRCLE_WIN;
		if (rules == RENJU && is_forbidden_static(board, rows, cols, last))
			return O_CIRCLE_WIN;
		int stones = 0;
		for (int i = 0; i < rows * cols; i++)
			stones += (board[i] != NONE);
		const bool is_draw = (draw_after > 0) ? (stones >= draw_after) : (stones == rows * cols);
		return is_draw ? O_DRAW : O_UNKNOWN;
	}

	/* ---- NNInputFeatures.cpp:15-32,59-113 ---- */
	void encode_features(Calc &calc, uint32_t *out)
	{
		static const uint32_t directional[8] = { 0u, 0u, 1u, (1u << 4), 0u, 0u, 0u, 0u };
		static const uint32_t isotropic[8] = { 0u, 0u, 0u, 0u, (1u << 8), (1u << 9), (1u << 10), (1u << 11) };
		const Sign own = calc.sign_to_move;
		const uint32_t stone_bits[4] = { 1u, (own == CROSS) ? 2u : 4u, (own == CROSS) ? 4u : 2u, 0u };
		const uint32_t base = (1u << 3) | ((own == CROSS) ? (1u << 4) : (1u << 5));
		for (int r = 0; r < calc.cfg.rows; r++)
			for (int c = 0; c < calc.cfg.cols; c++)
			{
				const int i = calc.idx(r, c);
				uint32_t r1 = 0, r2 = 0;
				for (uint32_t d = 0; d < 4; d++)
				{
					const int p1 = calc.ptype[i][0][d], p2 = calc.ptype[i][1][d];
					r1 |= (directional[p1] << d) | isotropic[p1];
					r2 |= (directional[p2] << d) | isotropic[p2];
				}
				const uint32_t pat = (own == CROSS) ? ((r1 << 8) | (r2 << 20)) : ((r1 << 20) | (r2 << 8));
				out[i] = base | stone_bits[calc.board[i]] | pat;
			}
		if (calc.cfg.rules == RENJU && own == CROSS)
			for (int r = 0; r < calc.cfg.rows; r++)
				for (int c = 0; c < calc.cfg.cols; c++)
					if (calc.is_forbidden(CROSS, r, c))
						out[calc.idx(r, c)] |= (1u << 6);
	}
}
