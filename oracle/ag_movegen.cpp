/*
 * oracle/ag_movegen.cpp — TEST INFRASTRUCTURE ONLY.  Staged threat-based move generator.
 * Follows src/search/alpha_beta/MoveGenerator.cpp; pinned by the 37 cases of the reference's
 * test/search/alpha_beta/test_move_generator.cpp restated as data in tests/golden/movegen_cases.json.
 */
#include "agoracle.hpp"

namespace ago
{
	void ActionList::add(Move m, Score s, int num)
	{ // ActionList.hpp:405-410 — the slot is written even when num == 0
		if (stack->data.size() <= base + size + 1)
			stack->data.resize(2 * (base + size + 1) + 64);
		stack->data[base + size].move = m;
		stack->data[base + size].score = s;
		size += num;
		stack->offset += num;
		stack->max_offset = std::max(stack->max_offset, stack->offset);
	}
	void ActionList::release()
	{ // ActionList.hpp:344-347
		stack->offset -= size;
	}
}

namespace
{
	using namespace ago;

	struct LocSet
	{ // StackVector<Location, N> (patterns/common.hpp:154-245): add = push back, remove(i) = move last into slot i
			Loc v[32];
			int n = 0;
			bool contains(Loc l) const
			{
				for (int i = 0; i < n; i++)
					if (v[i] == l)
						return true;
				return false;
			}
			void add(Loc l) { v[n++] = l; }
			void remove_at(int i) { v[i] = v[--n]; }
			void remove(Loc l)
			{
				for (int i = 0; i < n; i++)
					if (v[i] == l)
					{
						v[i] = v[--n];
						return;
					}
			}
	};
	void intersect(LocSet &lhs, const LocSet &rhs)
	{ // MoveGenerator.cpp:46-56
		int i = 0;
		while (i < lhs.n)
		{
			if (rhs.contains(lhs.v[i]))
				i++;
			else
				lhs.remove_at(i);
		}
	}
	void unite(LocSet &lhs, const LocSet &rhs)
	{ // :60-66
		for (int i = 0; i < rhs.n; i++)
			if (!lhs.contains(rhs.v[i]))
				lhs.add(rhs.v[i]);
	}
	struct DefensiveMoves
	{ // :94-121
			LocSet list;
			bool not_initialized = true;
			void intersect_with(const LocSet &other)
			{
				if (not_initialized)
				{
					for (int i = 0; i < other.n; i++)
						list.add(other.v[i]);
					not_initialized = false;
				}
				else
					intersect(list, other);
			}
			bool empty() const { return list.n == 0; }
	};
	int find_direction_of(const uint8_t *g, uint8_t v)
	{ // common.hpp:134-147
		for (int d = 0; d < 4; d++)
			if (g[d] == v)
				return d;
		return -1;
	}
	int count_of(const uint8_t *g, uint8_t v) { return (g[0] == v) + (g[1] == v) + (g[2] == v) + (g[3] == v); }
	bool is_a_four(uint8_t pt) { return pt == P_HALF_OPEN_4 || pt == P_OPEN_4 || pt == P_DOUBLE_4; }
	LocSet to_set(const Loc *p, int n)
	{
		LocSet s;
		for (int i = 0; i < n; i++)
			s.add(p[i]);
		return s;
	}
}

namespace ago
{
	bool MoveGen::is_forbidden(Sign s, Loc l)
	{ // MoveGenerator.cpp:1159-1173
		if (!anything_forbidden_for(s))
			return false;
		for (auto &e : forbidden_cache)
			if (e.first == l)
				return e.second;
		const bool r = pc.is_forbidden(s, l.row, l.col);
		forbidden_cache.push_back(std::make_pair(l, r));
		return r;
	}
	void MoveGen::add_move(Loc l, Score s, bool override_duplicate)
	{ // :228-249
		if ((added[l.row] >> l.col) & 1)
		{
			if (override_duplicate)
				for (int i = 0; i < act->size; i++)
					if ((*act)[i].move.loc() == l)
					{
						(*act)[i].score = s;
						return;
					}
		}
		else
		{
			act->add(Move(own(), l), s);
			added[l.row] |= (1u << l.col);
		}
	}
	void MoveGen::add_moves(const std::vector<Loc> &ls, Score s, bool override_duplicate)
	{
		for (size_t i = 0; i < ls.size(); i++)
			add_move(ls[i], s, override_duplicate);
	}
	int MoveGen::get_defensive_moves(Loc move, Direction d, Loc out[8])
	{ // :263-308
		Loc tmp[6];
		int n = pc.defensive_moves(own(), move.row, move.col, d, tmp);
		LocSet result = to_set(tmp, n);
		if (anything_forbidden_for(own()))
		{
			int i = 0;
			while (i < result.n)
			{
				if (is_forbidden(own(), result.v[i]))
				{
					add_move(result.v[i], Score::loss_in(1), true);
					result.remove_at(i);
				}
				else
					i++;
			}
		}
		else if (anything_forbidden_for(opp()))
		{
			const uint8_t pt = pc.patterns(opp(), move.row, move.col)[d];
			if (pt == P_OPEN_4)
			{
				const uint32_t raw = pc.raw_pattern(move.row, move.col, d, 6);
				int type = 0;
				if ((raw & 65520u) == 1344u)
					type = -1;
				if ((raw & 4193280u) == 344064u)
					type = +1;
				if (type != 0)
				{
					const Loc l = shift(d, 4 * type, move);
					if (is_forbidden(opp(), l))
						result.add(shift(d, -1 * type, move));
				}
			}
		}
		for (int i = 0; i < result.n; i++)
			out[i] = result.v[i];
		return result.n;
	}

	Score MoveGen::generate(ActionList &actions, GenMode mode)
	{ // MoveGenerator.cpp:159-223
		const int distance_to_draw = cfg.draw_after - pc.depth;
		if (distance_to_draw <= 0)
			return Score(PV_DRAW, 0);
		act = &actions;
		std::memset(added, 0, sizeof(added));
		forbidden_cache.clear();

		Result result;
		if (result.must_continue && distance_to_draw >= 1)
			result = try_win_in_1();
		if (result.must_continue && distance_to_draw == 1)
			result = try_draw_in_1();
		if (mode == G_THREATS || mode == G_OPTIMAL)
		{
			if (result.must_continue && distance_to_draw >= 2)
				result = defend_loss_in_2();
			if (result.must_continue && distance_to_draw >= 3)
				result = try_win_in_3();
			if (result.must_continue && distance_to_draw >= 4)
				result = defend_loss_in_4();
			if (result.must_continue && distance_to_draw >= 5)
				result = try_win_in_5();
			if (result.must_continue && distance_to_draw >= 6)
				result = defend_loss_in_6();
			if (result.must_continue && distance_to_draw >= 3)
				add_own_half_open_fours();
		}
		if (result.must_continue && mode >= G_OPTIMAL)
		{
			if (mode == G_OPTIMAL)
			{
				if (distance_to_draw >= 6)
				{
					add_moves(pc.threats(opp(), T_FORK_3x3).v, Score(3), false);
					add_moves(pc.threats(opp(), T_OPEN_3).v, Score(2), false);
				}
				if (distance_to_draw >= 5)
				{
					add_moves(pc.threats(own(), T_FORK_3x3).v, Score(13), false);
					add_moves(pc.threats(own(), T_OPEN_3).v, Score(1), false);
				}
				if (distance_to_draw >= 3)
					add_moves(pc.threats(opp(), T_HALF_OPEN_4).v, Score(4), false);
			}
			uint32_t mask[Calc::MAXN];
			if (mode <= G_REDUCED)
				mark_neighborhood(mask);
			else
				for (int r = 0; r < cfg.rows; r++)
					mask[r] = pc.legal[r];
			create_remaining_moves(mask, Score());
		}
		if (anything_forbidden_for(own()))
			mark_forbidden_moves();
		actions.is_fully_expanded = actions.must_defend || mode >= G_OPTIMAL;
		act = nullptr;
		return result.score;
	}

	MoveGen::Result MoveGen::try_draw_in_1()
	{ // :309-354
		act->baseline_score = Score::draw_in(1);
		if (anything_forbidden_for(own()))
		{
			bool found = false;
			for (int r = 0; r < cfg.rows; r++)
				for (int c = 0; c < cfg.cols; c++)
					if (pc.at(r, c) == NONE)
					{
						const Loc l(r, c);
						switch (pc.threat_at(own(), r, c))
						{
							default:
								add_move(l, Score::draw_in(1), false);
								found = true;
								break;
							case T_FORK_3x3:
								if (is_forbidden(own(), l))
									add_move(l, Score::loss_in(1), false);
								else
								{
									add_move(l, Score::draw_in(1), false);
									found = true;
								}
								break;
							case T_FORK_4x4:
							case T_OVERLINE:
								add_move(l, Score::loss_in(1), false);
								break;
						}
					}
			return Result { false, found ? Score::draw_in(1) : Score::loss_in(1) };
		}
		create_remaining_moves(pc.legal, Score::draw_in(1));
		return Result { false, Score::draw_in(1) };
	}
	MoveGen::Result MoveGen::try_win_in_1()
	{ // :355-371
		const std::vector<Loc> &fives = pc.threats(own(), T_FIVE).v;
		if (!fives.empty())
		{
			act->has_initiative = true;
			add_moves(fives, Score::win_in(1), false);
			return Result { false, Score::win_in(1) };
		}
		return Result();
	}
	MoveGen::Result MoveGen::defend_loss_in_2()
	{ // :372-463
		if (pc.threats(opp(), T_FIVE).size() == 0)
			return Result();
		const std::vector<Loc> &opp_fives = pc.threats(opp(), T_FIVE).v;
		act->must_defend = true;
		act->baseline_score = Score::loss_in(2);

		DefensiveMoves dm;
		for (size_t k = 0; k < opp_fives.size(); k++)
		{
			const Loc mv = opp_fives[k];
			const Direction dir = find_direction_of(pc.patterns(opp(), mv.row, mv.col), P_FIVE);
			Loc tmp[8];
			const int n = get_defensive_moves(mv, dir, tmp);
			dm.intersect_with(to_set(tmp, n));
			if (dm.empty())
			{
				add_moves(opp_fives, Score::loss_in(2), false);
				return Result { false, Score::loss_in(2) };
			}
		}
		Score best = Score::minus_inf();
		for (int k = 0; k < dm.list.n; k++)
		{
			const Loc mv = dm.list.v[k];
			Score response;
			switch (pc.threat_at(own(), mv.row, mv.col))
			{
				case T_FORK_3x3:
					if (anything_forbidden_for(own()))
					{
						if (count_of(pc.patterns(own(), mv.row, mv.col), P_OPEN_4) > 0)
							response = Score::win_in(3);
					}
					else if (!pc.has_any_four(opp()))
						response = Score::win_in(5);
					break;
				case T_FORK_4x3:
				{
					const Score solution = try_solve_own_fork_4x3(mv);
					response = solution.is_proven() ? solution : Score(15);
					break;
				}
				case T_FORK_4x4:
				case T_OPEN_4:
					response = Score::win_in(3);
					break;
				default:
					if (count_of(pc.patterns(own(), mv.row, mv.col), P_HALF_OPEN_4) > 0)
					{
						act->has_initiative = true;
						response = Score(14);
					}
					break;
			}
			if (response.is_win())
				act->has_initiative = true;
			add_move(mv, response, false);
			best = std::max(best, response);
		}
		return Result { false, best };
	}
	MoveGen::Result MoveGen::try_win_in_3()
	{ // :464-555
		int threat_count = 0;
		if (anything_forbidden_for(own()))
		{
			const std::vector<Loc> copy = pc.threats(own(), T_FORK_3x3).v;
			for (const Loc &mv : copy)
				if (count_of(pc.patterns(own(), mv.row, mv.col), P_OPEN_4) > 0 && !is_forbidden(own(), mv))
				{
					threat_count++;
					add_move(mv, Score::win_in(3), false);
				}
		}
		add_moves(pc.threats(own(), T_OPEN_4).v, Score::win_in(3), false);
		threat_count += pc.threats(own(), T_OPEN_4).size();

		if (pc.threats(own(), T_FORK_4x4).size() > 0 && !anything_forbidden_for(own()))
		{
			threat_count += pc.threats(own(), T_FORK_4x4).size();
			add_moves(pc.threats(own(), T_FORK_4x4).v, Score::win_in(3), false);
		}
		if (anything_forbidden_for(opp()))
		{
			const std::vector<Loc> copy = pc.threats(own(), T_HALF_OPEN_4).v;
			for (const Loc &mv : copy)
			{
				const Direction dir = find_direction_of(pc.patterns(own(), mv.row, mv.col), P_HALF_OPEN_4);
				bool winning = false;
				switch (pc.threat_at(opp(), mv.row, mv.col))
				{
					default:
						break;
					case T_FORK_3x3:
						if (pc.patterns(opp(), mv.row, mv.col)[dir] != P_OPEN_3 && is_forbidden(opp(), mv))
							winning = true;
						break;
					case T_FORK_4x4:
					case T_OVERLINE:
						winning = true;
						break;
				}
				if (winning)
				{
					Loc tmp[6];
					pc.defensive_moves(opp(), mv.row, mv.col, dir, tmp);
					const Loc original = (tmp[0] == mv) ? tmp[1] : tmp[0];
					add_move(original, Score::win_in(3), false);
					threat_count++;
					return Result { false, Score::win_in(3) };
				}
			}
		}
		if (threat_count > 0)
		{
			act->has_initiative = true;
			return Result { false, Score::win_in(3) };
		}
		return Result();
	}
	MoveGen::Result MoveGen::defend_loss_in_4()
	{ // :556-689
		const bool has_any_four = pc.has_any_four(own());
		act->baseline_score = Score::loss_in(4);
		Loc tmp[8];
		if (cfg.rules != RENJU)
		{
			DefensiveMoves dm;
			const std::vector<Loc> &opp_open_four = pc.threats(opp(), T_OPEN_4).v;
			for (const Loc &mv : opp_open_four)
			{
				act->must_defend = true;
				const Direction dir = find_direction_of(pc.patterns(opp(), mv.row, mv.col), P_OPEN_4);
				const int n = get_defensive_moves(mv, dir, tmp);
				dm.intersect_with(to_set(tmp, n));
				if (dm.empty() && !has_any_four)
				{
					add_moves(opp_open_four, Score::loss_in(4), false);
					return Result { false, Score::loss_in(4) };
				}
			}
			const std::vector<Loc> &opp_fork_4x4 = pc.threats(opp(), T_FORK_4x4).v;
			LocSet storage;
			for (const Loc &mv : opp_fork_4x4)
			{
				act->must_defend = true;
				const uint8_t *group = pc.patterns(opp(), mv.row, mv.col);
				for (Direction d = 0; d < 4; d++)
					if (group[d] == P_OPEN_4 || group[d] == P_DOUBLE_4)
					{
						const int n = get_defensive_moves(mv, d, tmp);
						dm.intersect_with(to_set(tmp, n));
					}
				if (count_of(group, P_HALF_OPEN_4) > 0)
				{
					storage.n = 0;
					for (Direction d = 0; d < 4; d++)
						if (group[d] == P_HALF_OPEN_4)
						{
							const int n = get_defensive_moves(mv, d, tmp);
							unite(storage, to_set(tmp, n));
						}
					dm.intersect_with(storage);
				}
				if (dm.empty() && !has_any_four)
				{
					add_moves(opp_fork_4x4, Score::loss_in(4), false);
					return Result { false, Score::loss_in(4) };
				}
			}
			for (int i = 0; i < dm.list.n; i++)
				add_move(dm.list.v[i], Score(), false);
		}
		else
		{
			{
				const std::vector<Loc> copy = pc.threats(opp(), T_OPEN_4).v;
				for (const Loc &mv : copy)
				{
					act->must_defend = true;
					const Direction dir = find_direction_of(pc.patterns(opp(), mv.row, mv.col), P_OPEN_4);
					const int n = get_defensive_moves(mv, dir, tmp);
					for (int i = 0; i < n; i++)
						add_move(tmp[i], Score(), false);
				}
			}
			if (anything_forbidden_for(opp()))
			{
				const std::vector<Loc> copy = pc.threats(opp(), T_FORK_3x3).v;
				for (const Loc &mv : copy)
				{
					const uint8_t *group = pc.patterns(opp(), mv.row, mv.col);
					if (count_of(group, P_OPEN_4) > 0 && !is_forbidden(opp(), mv))
					{
						act->must_defend = true;
						const Direction dir = find_direction_of(group, P_OPEN_4);
						const int n = get_defensive_moves(mv, dir, tmp);
						for (int i = 0; i < n; i++)
							add_move(tmp[i], Score(), false);
					}
				}
			}
			if (!anything_forbidden_for(opp()))
			{
				const std::vector<Loc> copy = pc.threats(opp(), T_FORK_4x4).v;
				for (const Loc &mv : copy)
				{
					act->must_defend = true;
					const uint8_t *group = pc.patterns(opp(), mv.row, mv.col);
					for (Direction d = 0; d < 4; d++)
						if (is_a_four(group[d]))
						{
							const int n = get_defensive_moves(mv, d, tmp);
							for (int i = 0; i < n; i++)
								add_move(tmp[i], Score(), false);
						}
				}
			}
		}
		if (act->must_defend)
		{
			act->has_initiative = has_any_four;
			const Score best = add_own_4x3_forks();
			add_own_half_open_fours();
			if (best.is_win())
				return Result { false, best };
			return Result { false, Score() };
		}
		act->baseline_score = Score();
		return Result();
	}
	MoveGen::Result MoveGen::try_win_in_5()
	{ // :690-720
		Score best = add_own_4x3_forks();
		if (!anything_forbidden_for(own()))
			if (number_of_available_fours_for(opp()) == 0)
			{
				const std::vector<Loc> &own_fork_3x3 = pc.threats(own(), T_FORK_3x3).v;
				if (!own_fork_3x3.empty())
				{
					add_moves(own_fork_3x3, Score::win_in(5), false);
					best = std::max(best, Score::win_in(5));
				}
			}
		if (best.is_win())
		{
			act->has_initiative = true;
			return Result { false, best };
		}
		return Result();
	}
	MoveGen::Result MoveGen::defend_loss_in_6()
	{ // :721-816
		if (number_of_available_fours_for(own()) > 0)
			return Result();
		const int fork_4x3_count = pc.threats(opp(), T_FORK_4x3).size();
		const int fork_3x3_count = pc.threats(opp(), T_FORK_3x3).size();
		if (fork_4x3_count > 0 || fork_3x3_count > 0)
		{
			act->must_defend = true;
			act->baseline_score = Score::loss_in(6);
		}
		Loc tmp[8];
		if (fork_4x3_count > 0)
		{
			const std::vector<Loc> &list = pc.threats(opp(), T_FORK_4x3).v;
			for (size_t k = 0; k < list.size(); k++)
			{
				const Loc mv = list[k];
				const uint8_t *group = pc.patterns(opp(), mv.row, mv.col);
				for (Direction d = 0; d < 4; d++)
					if (group[d] == P_OPEN_3)
					{
						const int n = get_defensive_moves(mv, d, tmp);
						for (int i = 0; i < n; i++)
							add_move(tmp[i], Score(0), false);
					}
				const Direction dir = find_direction_of(group, P_HALF_OPEN_4);
				Loc half4[8];
				const int nh = get_defensive_moves(mv, dir, half4);
				for (int i = 0; i < nh; i++)
					add_move(half4[i], Score(0), false);
				for (int i = 0; i < nh; i++)
					for (Direction d = 0; d < 4; d++)
					{
						const uint32_t reduced = pc.raw_pattern(half4[i].row, half4[i].col, d, 4);
						for (int j = -4; j <= 4; j++)
							if (((reduced >> (2 * (j + 4))) & 3u) == 0u)
							{
								const Loc l = shift(d, j, half4[i]);
								const uint8_t pt = pc.patterns(own(), l.row, l.col)[d];
								if (pt > P_NONE || pc.tab->is_half_open_three(pc.normal_pattern(l.row, l.col, d), own()))
									add_move(l, Score(), false);
							}
					}
			}
		}
		if (fork_3x3_count > 0)
		{
			const std::vector<Loc> &list = pc.threats(opp(), T_FORK_3x3).v;
			for (size_t k = 0; k < list.size(); k++)
			{
				const Loc mv = list[k];
				const uint8_t *group = pc.patterns(opp(), mv.row, mv.col);
				for (Direction d = 0; d < 4; d++)
					if (group[d] == P_OPEN_3)
					{
						const int n = get_defensive_moves(mv, d, tmp);
						for (int i = 0; i < n; i++)
							add_move(tmp[i], Score(0), false);
					}
				add_moves(pc.threats(own(), T_FORK_3x3).v, Score(13), false);
				add_moves(pc.threats(own(), T_OPEN_3).v, Score(1), false);
				uint32_t mask[Calc::MAXN];
				mark_star_like_pattern_for(own(), mask);
				for (int r = 0; r < cfg.rows; r++)
				{
					uint32_t bits = mask[r] & (~added[r]);
					for (int c = 0; c < cfg.cols; c++, bits >>= 1)
						if (bits & 1)
							for (Direction d = 0; d < 4; d++)
								if (pc.tab->is_half_open_three(pc.normal_pattern(r, c, d), own()))
								{
									add_move(Loc(r, c), Score(1), false);
									break;
								}
				}
			}
		}
		if (act->must_defend)
		{
			add_own_half_open_fours();
			return Result { false, Score() };
		}
		return Result();
	}
	Score MoveGen::add_own_4x3_forks()
	{ // :881-893
		Score result;
		const std::vector<Loc> &list = pc.threats(own(), T_FORK_4x3).v;
		for (size_t k = 0; k < list.size(); k++)
		{
			const Score solution = try_solve_own_fork_4x3(list[k]);
			add_move(list[k], solution, true);
			if (solution.is_proven())
				result = std::max(result, solution);
		}
		return result;
	}
	void MoveGen::add_own_half_open_fours()
	{ // :894-946
		const Score prior(14);
		int hidden = 0;
		if (anything_forbidden_for(own()))
		{
			const std::vector<Loc> copy = pc.threats(own(), T_FORK_3x3).v;
			for (const Loc &mv : copy)
				if (count_of(pc.patterns(own(), mv.row, mv.col), P_HALF_OPEN_4) > 0 && !is_forbidden(own(), mv))
				{
					add_move(mv, prior, false);
					hidden++;
				}
		}
		add_moves(pc.threats(own(), T_HALF_OPEN_4).v, prior, false);
		if (hidden + pc.threats(own(), T_HALF_OPEN_4).size() > 0)
			act->has_initiative = true;
	}
	Score MoveGen::try_solve_own_fork_4x3(Loc move)
	{ // :947-992
		const Score prior(15);
		if (anything_forbidden_for(own()))
			return prior;
		const Direction dir = find_direction_of(pc.patterns(own(), move.row, move.col), P_HALF_OPEN_4);
		Loc tmp[6];
		const int n = pc.defensive_moves(opp(), move.row, move.col, dir, tmp);
		LocSet dm = to_set(tmp, n);
		dm.remove(move);
		ThreatType best = T_NONE;
		for (int i = 0; i < dm.n; i++)
		{
			const ThreatType tt = pc.threat_at(opp(), dm.v[i].row, dm.v[i].col);
			if ((tt != T_FORK_4x4 && tt != T_OVERLINE) || !anything_forbidden_for(opp()))
				best = std::max(best, tt);
		}
		switch (best)
		{
			default:
			case T_NONE:
			case T_HALF_OPEN_3:
			case T_OPEN_3:
			case T_FORK_3x3:
				return Score::win_in(5);
			case T_HALF_OPEN_4:
			case T_FORK_4x3:
				return prior;
			case T_FORK_4x4:
			case T_OPEN_4:
				return Score::loss_in(4);
			case T_FIVE:
			case T_OVERLINE:
				return Score::loss_in(2);
		}
	}
	void MoveGen::mark_forbidden_moves()
	{ // :993-1010
		add_moves(pc.threats(own(), T_OVERLINE).v, Score::loss_in(1), true);
		add_moves(pc.threats(own(), T_FORK_4x4).v, Score::loss_in(1), true);
		const std::vector<Loc> copy = pc.threats(own(), T_FORK_3x3).v;
		for (const Loc &mv : copy)
			if (is_forbidden(CROSS, mv))
				add_move(mv, Score::loss_in(1), true);
	}
	static void stamp(uint32_t *rows, int nrows, int r, int c, const uint32_t pattern[7])
	{ // the 7x7 stencil of MoveGenerator.cpp:1011-1071 centred on (r, c); bit 6 of a stencil row is the leftmost column
		for (int i = 0; i < 7; i++)
		{
			const int rr = r - 3 + i;
			if (rr < 0 || rr >= nrows)
				continue;
			for (int j = 0; j < 7; j++)
				if ((pattern[i] >> (6 - j)) & 1)
				{
					const int cc = c - 3 + j;
					if (cc >= 0 && cc < 32)
						rows[rr] |= (1u << cc);
				}
		}
	}
	void MoveGen::mark_neighborhood(uint32_t out[Calc::MAXN])
	{ // :1011-1071
		static const uint32_t pattern[7] = { 73u, 62u, 62u, 119u, 62u, 62u, 73u };
		uint32_t tmp[Calc::MAXN];
		std::memset(tmp, 0, sizeof(tmp));
		for (int r = 0; r < cfg.rows; r++)
			for (int c = 0; c < cfg.cols; c++)
				if (((pc.legal[r] >> c) & 1) == 0)
					stamp(tmp, cfg.rows, r, c, pattern);
		if (pc.depth == 0)
			tmp[cfg.rows / 2] |= (1u << (cfg.cols / 2));
		for (int r = 0; r < cfg.rows; r++)
			out[r] = tmp[r] & pc.legal[r];
	}
	void MoveGen::mark_star_like_pattern_for(Sign s, uint32_t out[Calc::MAXN])
	{ // :1072-1126
		static const uint32_t pattern[7] = { 73u, 42u, 28u, 119u, 28u, 42u, 73u };
		uint32_t tmp[Calc::MAXN];
		std::memset(tmp, 0, sizeof(tmp));
		for (int r = 0; r < cfg.rows; r++)
			for (int c = 0; c < cfg.cols; c++)
				if (pc.at(r, c) == s)
					stamp(tmp, cfg.rows, r, c, pattern);
		for (int r = 0; r < cfg.rows; r++)
			out[r] = tmp[r] & pc.legal[r];
	}
	void MoveGen::create_remaining_moves(const uint32_t mask[Calc::MAXN], Score s)
	{ // :1127-1137
		for (int r = 0; r < cfg.rows; r++)
		{
			uint32_t bits = mask[r] & (~added[r]);
			for (int c = 0; c < cfg.cols; c++, bits >>= 1)
				act->add(Move(own(), r, c), s, bits & 1);
			added[r] |= mask[r];
		}
	}
	int MoveGen::number_of_available_fours_for(Sign s) const
	{ // :1198-1207
		const int open4 = pc.threats(s, T_OPEN_4).size();
		const int f44 = anything_forbidden_for(s) ? 0 : pc.threats(s, T_FORK_4x4).size();
		return open4 + f44 + pc.threats(s, T_FORK_4x3).size() + pc.threats(s, T_HALF_OPEN_4).size();
	}
}
