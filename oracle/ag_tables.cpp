/*
 * oracle/ag_tables.cpp — TEST INFRASTRUCTURE ONLY.  Pattern / threat / defensive-move tables and the score algebra.
 * Pinned entry-by-entry against oracle/_ref/libagref.so (the reference's own PatternTable.cpp, PatternClassifier.cpp,
 * ThreatTable.cpp, DefensiveMoveTable.cpp, Score.hpp compiled unmodified) by tests/test_oracle_tables.py.
 */
#include "agoracle.hpp"

#include <array>
#include <mutex>
#include <memory>

namespace ago
{
	Score negate(Score s)
	{ // Score.hpp:213-228
		switch (s.pv())
		{
			case PV_LOSS: return s.is_finite() ? Score(PV_WIN, -s.eval()) : Score::plus_inf();
			case PV_DRAW: return Score(PV_DRAW, s.eval());
			case PV_WIN: return s.is_finite() ? Score(PV_LOSS, -s.eval()) : Score::minus_inf();
			default: return Score(PV_UNKNOWN, -s.eval());
		}
	}
	Score invert_up(Score s)
	{ // Score.hpp:285-300
		switch (s.pv())
		{
			case PV_LOSS: return s.is_finite() ? Score::win_in(s.distance() + 1) : negate(s);
			case PV_DRAW: return Score::draw_in(s.distance() + 1);
			case PV_WIN: return s.is_finite() ? Score::loss_in(s.distance() + 1) : negate(s);
			default: return negate(s);
		}
	}
	Score invert_down(Score s)
	{ // Score.hpp:303-318
		switch (s.pv())
		{
			case PV_LOSS: return s.is_finite() ? Score::win_in(s.distance() - 1) : negate(s);
			case PV_DRAW: return Score::draw_in(s.distance() - 1);
			case PV_WIN: return s.is_finite() ? Score::loss_in(s.distance() - 1) : negate(s);
			default: return negate(s);
		}
	}
}

namespace
{
	using namespace ago;

	/* A matching rule = one allowed-sign bitset per cell (PatternClassifier.cpp:15-69: '_' 'X' 'O' '|', [not s], [any], [_|]). */
	typedef std::vector<uint8_t> MatchRule;
	constexpr uint8_t BIT_NONE = 1, BIT_X = 2, BIT_O = 4, BIT_WALL = 8, BIT_ANY = 15;

	MatchRule from_text(const char *txt, Sign own)
	{ // txt uses 'S' for an own stone and '_' for an empty spot
		MatchRule r;
		for (const char *p = txt; *p; ++p)
			r.push_back(*p == '_' ? BIT_NONE : (own == CROSS ? BIT_X : BIT_O));
		return r;
	}
	MatchRule wrap(uint8_t pre, const MatchRule &core, uint8_t post)
	{
		MatchRule r;
		r.push_back(pre);
		r.insert(r.end(), core.begin(), core.end());
		r.push_back(post);
		return r;
	}
	bool matches_anywhere(const MatchRule &rule, const uint8_t *line, int len)
	{ // PatternClassifier.cpp:70-83 (the window slides over the whole line)
		const int n = static_cast<int>(rule.size());
		for (int i = 0; i + n <= len; i++)
		{
			bool ok = true;
			for (int j = 0; j < n && ok; j++)
				ok = (rule[j] >> line[i + j]) & 1;
			if (ok)
				return true;
		}
		return false;
	}

	struct Classifier
	{ // PatternTable.cpp:31-70 ThreatClassifier + PatternClassifier.cpp:182-326 rule sets
			std::vector<MatchRule> overline, five, open4, double4, half4, open3, half3;
			static void modify(std::vector<MatchRule> &rules, Rules r, Sign own, bool closed_kind)
			{
				const uint8_t own_bit = (own == CROSS) ? BIT_X : BIT_O;
				const uint8_t opp_bit = (own == CROSS) ? BIT_O : BIT_X;
				const uint8_t not_own = BIT_ANY & ~own_bit;
				const uint8_t not_opp = BIT_ANY & ~opp_bit;
				const uint8_t free_or_wall = BIT_NONE | BIT_WALL;
				std::vector<MatchRule> out;
				const bool exact_five = (r == STANDARD) || (r == RENJU && own == CROSS);
				for (const MatchRule &m : rules)
				{
					if (exact_five)
						out.push_back(wrap(not_own, m, not_own));                 // modifyPatternsAND("[not X]", "[not X]")
					else if (r == CARO5)
					{
						if (closed_kind)
						{ // modifyPatternsOR("[_|]", "[not X]", "[_|]") — PatternClassifier.cpp:169-179
							out.push_back(wrap(free_or_wall, m, not_own));
							out.push_back(wrap(not_own, m, free_or_wall));
						}
						else
							out.push_back(wrap(free_or_wall, m, free_or_wall));     // modifyPatternsAND("[_|]", "[_|]")
					}
					else if (r == CARO6)
					{
						if (closed_kind)
						{ // modifyPatternsOR("[not O]", "[any]", "[not O")
							out.push_back(wrap(not_opp, m, BIT_ANY));
							out.push_back(wrap(BIT_ANY, m, not_opp));
						}
						else
							out.push_back(wrap(not_opp, m, not_opp));
					}
					else
						out.push_back(m);
				}
				rules = out;
			}
			Classifier(Rules r, Sign own)
			{
				const char *t_over[] = { "SSSSSS" };
				const char *t_five[] = { "SSSSS" };
				const char *t_open4[] = { "_SSSS_" };
				const char *t_double4[] = { "S_SSS_S", "SS_SS_SS", "SSS_S_SSS" };
				const char *t_half4[] = { "_SSSS", "S_SSS", "SS_SS", "SSS_S", "SSSS_" };
				const char *t_open3[] = { "_SSS__", "_SS_S_", "_S_SS_", "__SSS_" };
				const char *t_half3[] = { "__SSS", "_S_SS", "_SS_S", "_SSS_", "S__SS", "S_S_S", "S_SS_", "SS__S", "SS_S_", "SSS__" };
				for (const char *t : t_over) overline.push_back(from_text(t, own));
				for (const char *t : t_five) five.push_back(from_text(t, own));
				for (const char *t : t_open4) open4.push_back(from_text(t, own));
				for (const char *t : t_double4) double4.push_back(from_text(t, own));
				for (const char *t : t_half4) half4.push_back(from_text(t, own));
				for (const char *t : t_open3) open3.push_back(from_text(t, own));
				for (const char *t : t_half3) half3.push_back(from_text(t, own));
				modify(five, r, own, true);
				modify(open4, r, own, false);
				modify(double4, r, own, false);
				modify(half4, r, own, true);
				modify(open3, r, own, false);
				modify(half3, r, own, true);
			}
			static bool any(const std::vector<MatchRule> &rules, const uint8_t *line)
			{
				for (const MatchRule &m : rules)
					if (matches_anywhere(m, line, 11))
						return true;
				return false;
			}
			PatternType classify(const uint8_t *line) const
			{ // PatternTable.cpp:51-69 — order matters
				if (any(five, line)) return P_FIVE;
				if (any(overline, line)) return P_OVERLINE;
				if (any(open4, line)) return P_OPEN_4;
				if (any(double4, line)) return P_DOUBLE_4;
				if (any(half4, line)) return P_HALF_OPEN_4;
				if (any(open3, line)) return P_OPEN_3;
				if (any(half3, line)) return P_HALF_OPEN_3;
				return P_NONE;
			}
	};

	bool line_is_valid(const uint8_t *line)
	{ // Pattern.hpp:52-63 (off-board cells must be contiguous from the outside; centre empty)
		if (line[5] != NONE)
			return false;
		for (int i = 0; i < 5; i++)
			if (line[i] != ILLEGAL && line[i + 1] == ILLEGAL)
				return false;
		for (int i = 6; i < 11; i++)
			if (line[i - 1] == ILLEGAL && line[i] != ILLEGAL)
				return false;
		return true;
	}

	void build_pattern_table(Tables &t)
	{ // PatternTable.cpp:120-165
		const Classifier for_cross(t.rules, CROSS), for_circle(t.rules, CIRCLE);
		t.pattern_types.assign(1u << 20, 0);
		t.half_open_3.assign(1u << 20, 0);
		uint8_t line[11];
		for (uint32_t i = 0; i < (1u << 20); i++)
		{
			const uint32_t expanded = (i & 1023u) | ((i & 1047552u) << 2u); // PatternTable.hpp:138-141
			for (int k = 0; k < 11; k++)
				line[k] = (expanded >> (2 * k)) & 3;
			if (!line_is_valid(line))
				continue;
			line[5] = CROSS;
			PatternType cross = for_cross.classify(line);
			line[5] = CIRCLE;
			PatternType circle = for_circle.classify(line);
			uint8_t h = 0;
			if (cross == P_HALF_OPEN_3)
			{
				h |= 1;
				cross = P_NONE;
			}
			if (circle == P_HALF_OPEN_3)
			{
				h |= 2;
				circle = P_NONE;
			}
			t.pattern_types[i] = static_cast<uint8_t>(cross | (circle << 4));
			t.half_open_3[i] = h;
		}
	}

	/* ---- ThreatTable.cpp:32-96 ---- */
	int count_of(const int g[4], int v) { return (g[0] == v) + (g[1] == v) + (g[2] == v) + (g[3] == v); }
	void threat_of(const int g[4], Rules rules, uint8_t out[2])
	{
		auto both = [&](ThreatType x) { out[0] = x; out[1] = x; };
		auto pair = [&](ThreatType x, ThreatType o) { out[0] = x; out[1] = o; };
		const bool five = count_of(g, P_FIVE) > 0;
		const bool overline = count_of(g, P_OVERLINE) > 0;
		const int sum3 = count_of(g, P_OPEN_3);
		const int sum4 = count_of(g, P_OPEN_4) + count_of(g, P_HALF_OPEN_4);
		const bool fork33 = sum3 >= 2;
		const bool fork43 = sum3 >= 1 && sum4 >= 1;
		const bool fork44 = count_of(g, P_DOUBLE_4) > 0 || sum4 >= 2;
		const bool open4 = count_of(g, P_OPEN_4) > 0;
		if (five) return both(T_FIVE);
		if (rules == RENJU)
		{
			if (overline) return pair(T_OVERLINE, T_FIVE);
			if (fork44) return both(T_FORK_4x4);
			if (open4) return fork33 ? pair(T_FORK_3x3, T_OPEN_4) : both(T_OPEN_4);
			if (fork43) return fork33 ? pair(T_FORK_3x3, T_FORK_4x3) : both(T_FORK_4x3);
		}
		else
		{
			if (fork44) return both(T_FORK_4x4);
			if (open4) return both(T_OPEN_4);
			if (fork43) return both(T_FORK_4x3);
		}
		if (fork33) return both(T_FORK_3x3);
		if (count_of(g, P_HALF_OPEN_4) > 0) return both(T_HALF_OPEN_4);
		if (count_of(g, P_OPEN_3) > 0) return both(T_OPEN_3);
		if (count_of(g, P_HALF_OPEN_3) > 0) return both(T_HALF_OPEN_3);
		return both(T_NONE);
	}

	/* ---- DefensiveMoveTable.cpp ---- */
	bool overline_allowed(Rules r, Sign attacker) { return r == FREESTYLE || (r == RENJU && attacker == CIRCLE) || r == CARO6; } // :21-24
	bool blocked_allowed(Rules r) { return r != CARO5 && r != CARO6; }                                                            // :25-28

	struct DefendFive
	{ // DefensiveMoveTable.cpp:118-217
			Sign attacker, defender;
			bool allow_overline, allow_blocked;
			DefendFive(Rules r, Sign def) : attacker(invert_sign(def)), defender(def), allow_overline(overline_allowed(r, invert_sign(def))), allow_blocked(blocked_allowed(r)) {}
			bool is_five(const uint8_t *line, int len) const
			{ // :200-216
				for (int i = 1; i < len - 5; i++)
				{
					bool five = true;
					for (int k = 0; k < 5 && five; k++)
						five = (line[i + k] == attacker);
					if (five)
					{
						const Sign first = line[i - 1], last = line[i + 5];
						const bool win_overline = allow_overline ? true : (first != attacker && last != attacker);
						const bool win_blocked = allow_blocked ? true : !(first == defender && last == defender);
						if (win_overline && win_blocked)
							return true;
					}
				}
				return false;
			}
			int search(uint8_t *line, int len, Sign sign, int depth_remaining) const
			{ // :175-199
				int outcome = -1;
				for (int i = 0; i < len; i++)
					if (line[i] == NONE)
					{
						line[i] = sign;
						int tmp = 0;
						if (is_five(line, len))
						{
							line[i] = NONE; // (the reference returns with its by-value copy still modified)
							return 1;
						}
						else
							tmp = (depth_remaining > 1) ? -search(line, len, invert_sign(sign), depth_remaining - 1) : 0;
						line[i] = NONE;
						outcome = std::max(outcome, tmp);
					}
				return outcome;
			}
			uint16_t operator()(uint32_t encoded, int len, int offset, int depth) const
			{ // :135-155
				uint8_t line[16];
				for (int i = 0; i < len; i++)
					line[i] = (encoded >> (2 * i)) & 3;
				if (is_five(line, len))
					return 0;
				if (search(line, len, attacker, depth) == 0)
					return 0;
				uint16_t result = 0;
				for (int i = 0; i < len; i++)
					if (line[i] == NONE)
					{
						line[i] = defender;
						if (search(line, len, attacker, depth) != 1)
							result |= static_cast<uint16_t>(1u << (offset + i));
						line[i] = NONE;
					}
				return result;
			}
	};

	// attacker masks (DefensiveMoveTable.cpp:168-173, 234-239, 259-264, 284-296, 316-323); circle = 2 x cross
	const uint32_t FIVE_MASKS[5] = { 85u, 277u, 325u, 337u, 340u };
	const uint32_t OPEN4_MASKS[4] = { 84u, 276u, 324u, 336u };
	const uint32_t DOUBLE4_MASKS[6] = { 4177u, 4369u, 4417u, 20549u, 20741u, 86037u };
	const int DOUBLE4_LEN[6] = { 7, 7, 7, 8, 8, 9 };
	const int DOUBLE4_OFF[6] = { 2, 3, 4, 2, 3, 2 };
	const uint32_t HALF4_MASKS[20] = { 21u, 69u, 81u, 84u, 21u, 261u, 273u, 276u, 69u, 261u, 321u, 324u, 81u, 273u, 321u, 336u, 84u, 276u, 324u, 336u };
	const int HALF4_OFF[20] = { 3, 4, 5, 6, 2, 4, 5, 6, 2, 3, 5, 6, 2, 3, 4, 6, 2, 3, 4, 5 };
	const uint32_t OPEN3_MASKS[12] = { 20u, 68u, 80u, 20u, 260u, 272u, 68u, 260u, 320u, 80u, 272u, 320u };
	const int OPEN3_OFF[12] = { 3, 4, 5, 2, 4, 5, 2, 3, 5, 2, 3, 4 };

	uint32_t mask_for(uint32_t cross_mask, Sign attacker) { return (attacker == CROSS) ? cross_mask : 2u * cross_mask; }
	uint32_t sub_pattern(uint32_t line, int start, int length) { return (line >> (2u * start)) & ((1u << (2u * length)) - 1u); }
	uint32_t side_index(uint32_t line, int left, int right) { return ((line >> (2 * (left - 2))) & 15u) | (((line >> (2 * right)) & 15u) << 4); } // :54-59

	void build_defensive_tables(Tables &t)
	{ // DefensiveMoveTable.cpp:503-577
		const DefendFive by_cross(t.rules, CROSS), by_circle(t.rules, CIRCLE);
		auto fill = [&](int rows, const uint32_t *masks, const int *lens, int fixed_len, const int *offs, int depth, uint16_t (*dst)[256][2])
		{
			for (int i = 0; i < rows; i++)
			{
				const int length = lens ? lens[i] : fixed_len;
				const int offset = (offs ? offs[i] : (2 + i)) - 2;
				for (uint32_t j = 0; j < 256; j++)
				{
					const uint32_t left = j & 15u;
					const uint32_t right = (j & 0xF0u) << (2 * length);
					const uint32_t ext_cross = left | (mask_for(masks[i], CIRCLE) << 4) | right;  // pattern to defend by cross
					const uint32_t ext_circle = left | (mask_for(masks[i], CROSS) << 4) | right;  // pattern to defend by circle
					dst[i][j][0] = by_cross(ext_cross, length + 4, offset, depth);
					dst[i][j][1] = by_circle(ext_circle, length + 4, offset, depth);
				}
			}
		};
		fill(5, FIVE_MASKS, nullptr, 5, nullptr, 1, t.five_defense);
		fill(4, OPEN4_MASKS, nullptr, 6, nullptr, 3, t.open_four_defense);
		fill(6, DOUBLE4_MASKS, DOUBLE4_LEN, 0, DOUBLE4_OFF, 3, t.double_four_defense);
	}
}

namespace ago
{
	const Tables& Tables::get(Rules r)
	{
		static std::mutex mtx;
		static std::unique_ptr<Tables> cache[5];
		std::lock_guard<std::mutex> lock(mtx);
		if (!cache[r])
		{
			std::unique_ptr<Tables> t(new Tables());
			t->rules = r;
			build_pattern_table(*t);
			for (int i = 0; i < 4096; i++)
			{
				const int g[4] = { i & 7, (i >> 3) & 7, (i >> 6) & 7, (i >> 9) & 7 };
				threat_of(g, r, t->threats[i]);
			}
			build_defensive_tables(*t);
			cache[r] = std::move(t);
		}
		return *cache[r];
	}

	uint16_t Tables::defensive_moves(uint32_t pattern, Sign defender, PatternType threat_to_defend) const
	{ // DefensiveMoveTable.cpp:380-461
		const Sign attacker = invert_sign(defender);
		const int d = (defender == CROSS) ? 0 : 1;
		const int center = 6;
		switch (threat_to_defend)
		{
			case P_FIVE:
				for (int i = 0; i < 5; i++)
				{
					const int begin = 2 + i;
					if (sub_pattern(pattern, begin, 5) == mask_for(FIVE_MASKS[i], attacker))
						return five_defense[i][side_index(pattern, begin, begin + 5)][d];
				}
				return 0;
			case P_OPEN_4:
				for (int i = 0; i < 4; i++)
				{
					const int begin = 2 + i;
					if (sub_pattern(pattern, begin, 6) == mask_for(OPEN4_MASKS[i], attacker))
						return open_four_defense[i][side_index(pattern, begin, begin + 6)][d];
				}
				return 0;
			case P_DOUBLE_4:
				for (int i = 0; i < 6; i++)
				{
					const int length = DOUBLE4_LEN[i], begin = DOUBLE4_OFF[i];
					if (sub_pattern(pattern, begin, length) == mask_for(DOUBLE4_MASKS[i], attacker))
						return double_four_defense[i][side_index(pattern, begin, begin + length)][d];
				}
				return 0;
			case P_HALF_OPEN_4:
			{
				const bool allow_overline = overline_allowed(rules, attacker), allow_blocked = blocked_allowed(rules);
				uint16_t result = static_cast<uint16_t>(1u << center);
				for (int i = 0; i < 20; i++)
				{
					const int begin = HALF4_OFF[i];
					if (sub_pattern(pattern, begin, 5) != mask_for(HALF4_MASKS[i], attacker))
						continue;
					// CheckSides (:93-116)
					const Sign first = (pattern >> (2 * (begin - 1))) & 3, last = (pattern >> (2 * (begin + 5))) & 3;
					if (!allow_overline && (first == attacker || last == attacker))
						continue;
					if (!allow_blocked && (first == defender && last == defender))
						continue;
					uint16_t tmp = five_defense[i / 4][side_index(pattern, begin, begin + 5)][d];
					const int sh = begin - (2 + i / 4);
					tmp = (sh >= 0) ? static_cast<uint16_t>(tmp << sh) : static_cast<uint16_t>(tmp >> (-sh));
					result |= tmp;
					if (rules != CARO5 && rules != CARO6)
						return result;
				}
				return result;
			}
			case P_OPEN_3:
				for (int i = 0; i < 12; i++)
				{
					const int begin = OPEN3_OFF[i];
					if (sub_pattern(pattern, begin, 6) == mask_for(OPEN3_MASKS[i], attacker))
					{
						uint16_t result = open_four_defense[i / 3][side_index(pattern, begin, begin + 6)][d];
						const int sh = begin - (2 + i / 3);
						result = (sh >= 0) ? static_cast<uint16_t>(result << sh) : static_cast<uint16_t>(result >> (-sh));
						result |= static_cast<uint16_t>(1u << center);
						return result;
					}
				}
				return 0;
			default:
				return 0;
		}
	}

	uint16_t open_three_promotion_moves(uint32_t pattern)
	{ // DefensiveMoveTable.cpp:329-377 (scalar branch): first match wins
		static const uint32_t patterns[12] = { 320u, 4352u, 20480u, 80u, 16640u, 69632u, 272u, 4160u, 81920u, 320u, 4352u, 20480u };
		static const uint32_t masks[12] = { 65520u, 262080u, 1048320u, 16380u, 262080u, 1048320u, 16380u, 65520u, 1048320u, 16380u, 65520u, 262080u };
		static const uint16_t results[12] = { 196u, 392u, 784u, 82u, 328u, 656u, 74u, 148u, 592u, 70u, 140u, 280u };
		for (int i = 0; i < 12; i++)
			if ((pattern & masks[i]) == patterns[i])
				return results[i];
		return 0;
	}
}
