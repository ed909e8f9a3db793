/*
 * tables_host.cpp — host-side construction of the lookup tables the device search kernels gather from.
 *
 * What they replace in the reference (all built once per rule set, then read-only):
 *   - PatternTable   (src/patterns/PatternTable.cpp:120-165, rules from PatternClassifier.cpp:182-326):
 *       4^10 entries, one byte: low nibble = line pattern made by a CROSS stone at the centre, high nibble = by CIRCLE;
 *       plus 2 bits "is a half-open three" per entry.
 *   - ThreatTable    (src/patterns/ThreatTable.cpp:52-96,179-189): 8^4 entries x 2 signs.
 *   - DefensiveMoveTable (src/patterns/DefensiveMoveTable.cpp:118-217,503-577): (5+4+6) x 256 x 2 16-bit masks.
 *
 * The formulation here is bit-parallel (one 11-bit occupancy mask per cell value, a rule = four "allowed" masks) instead
 * of the reference's per-cell string matcher; results are verified table-wide against the compiled reference
 * (tests/test_tables_product.py).
 */
#include "tables_host.hpp"

#include <cstring>
#include <algorithm>

namespace agx
{
	namespace
	{
		enum { V_EMPTY = 0, V_CROSS = 1, V_CIRCLE = 2, V_WALL = 3 };

		struct BitRule
		{
				int length;
				uint32_t allowed[4]; // allowed[v] bit j set <=> value v may stand at position j of the rule
		};
		BitRule make_rule(const char *core, int own, uint8_t pre, uint8_t post, bool wrapped)
		{
			// core: 'S' own stone, '_' empty; pre/post: 4-bit sets of allowed values for the flanking cells
			BitRule r;
			r.length = 0;
			for (int v = 0; v < 4; v++)
				r.allowed[v] = 0;
			auto push = [&](uint8_t set)
			{
				for (int v = 0; v < 4; v++)
					if ((set >> v) & 1)
						r.allowed[v] |= (1u << r.length);
				r.length++;
			};
			if (wrapped)
				push(pre);
			for (const char *p = core; *p; ++p)
				push(*p == '_' ? (1u << V_EMPTY) : (1u << own));
			if (wrapped)
				push(post);
			return r;
		}
		struct RuleSet
		{
				std::vector<BitRule> rules;
				bool matches(const uint32_t occupancy[4], int line_length) const
				{
					for (const BitRule &r : rules)
					{
						const uint32_t window = (1u << r.length) - 1u;
						for (int i = 0; i + r.length <= line_length; i++)
						{
							uint32_t bad = 0;
							for (int v = 0; v < 4; v++)
								bad |= (occupancy[v] >> i) & window & ~r.allowed[v];
							if (bad == 0)
								return true;
						}
					}
					return false;
				}
		};
		enum Kind { K_OVERLINE, K_FIVE, K_OPEN4, K_DOUBLE4, K_HALF4, K_OPEN3, K_HALF3, K_COUNT };

		void build_rule_sets(int rules, int own, RuleSet out[K_COUNT])
		{
			static const char *cores[K_COUNT][10] = { { "SSSSSS" }, { "SSSSS" }, { "_SSSS_" }, { "S_SSS_S", "SS_SS_SS", "SSS_S_SSS" }, { "_SSSS", "S_SSS",
					"SS_SS", "SSS_S", "SSSS_" }, { "_SSS__", "_SS_S_", "_S_SS_", "__SSS_" }, { "__SSS", "_S_SS", "_SS_S", "_SSS_", "S__SS", "S_S_S", "S_SS_",
					"SS__S", "SS_S_", "SSS__" } };
			const int opp = (own == V_CROSS) ? V_CIRCLE : V_CROSS;
			const uint8_t any = 15, not_own = any & ~(1u << own), not_opp = any & ~(1u << opp), free_or_wall = (1u << V_EMPTY) | (1u << V_WALL);
			const bool exact = (rules == AGX_STANDARD) || (rules == AGX_RENJU && own == V_CROSS);
			for (int k = 0; k < K_COUNT; k++)
			{
				const bool closed = (k == K_FIVE || k == K_HALF4 || k == K_HALF3); // kinds that use the OR form in caro
				for (int c = 0; c < 10 && cores[k][c] != nullptr; c++)
				{
					const char *core = cores[k][c];
					if (k == K_OVERLINE)
						out[k].rules.push_back(make_rule(core, own, 0, 0, false));
					else if (exact)
						out[k].rules.push_back(make_rule(core, own, not_own, not_own, true));
					else if (rules == AGX_CARO5)
					{
						if (closed)
						{
							out[k].rules.push_back(make_rule(core, own, free_or_wall, not_own, true));
							out[k].rules.push_back(make_rule(core, own, not_own, free_or_wall, true));
						}
						else
							out[k].rules.push_back(make_rule(core, own, free_or_wall, free_or_wall, true));
					}
					else if (rules == AGX_CARO6)
					{
						if (closed)
						{
							out[k].rules.push_back(make_rule(core, own, not_opp, any, true));
							out[k].rules.push_back(make_rule(core, own, any, not_opp, true));
						}
						else
							out[k].rules.push_back(make_rule(core, own, not_opp, not_opp, true));
					}
					else
						out[k].rules.push_back(make_rule(core, own, 0, 0, false));
				}
			}
		}
		int classify(const RuleSet sets[K_COUNT], const uint32_t occ[4])
		{ // priority order of PatternTable.cpp:51-69
			if (sets[K_FIVE].matches(occ, 11)) return 6;
			if (sets[K_OVERLINE].matches(occ, 11)) return 7;
			if (sets[K_OPEN4].matches(occ, 11)) return 4;
			if (sets[K_DOUBLE4].matches(occ, 11)) return 5;
			if (sets[K_HALF4].matches(occ, 11)) return 3;
			if (sets[K_OPEN3].matches(occ, 11)) return 2;
			if (sets[K_HALF3].matches(occ, 11)) return 1;
			return 0;
		}

		/* defensive moves: small game-tree search on a short line (depth 1 for fives, 3 for open / double fours) */
		struct LineGame
		{
				int attacker, defender;
				bool allow_overline, allow_blocked;
				bool has_five(const uint8_t *cells, int n) const
				{
					for (int i = 1; i + 5 < n; i++)
					{
						int run = 0;
						while (run < 5 && cells[i + run] == attacker)
							run++;
						if (run == 5)
						{
							const int a = cells[i - 1], b = cells[i + 5];
							if ((allow_overline || (a != attacker && b != attacker)) && (allow_blocked || !(a == defender && b == defender)))
								return true;
						}
					}
					return false;
				}
				int best_for(uint8_t *cells, int n, int side, int depth) const
				{ // +1: `side` can complete a five within the horizon, 0: nothing decided, -1: no empty cell
					int outcome = -1;
					for (int i = 0; i < n; i++)
						if (cells[i] == V_EMPTY)
						{
							cells[i] = static_cast<uint8_t>(side);
							const bool five = has_five_for(cells, n, side);
							int v = 0;
							if (!five && depth > 1)
								v = -best_for(cells, n, 3 - side, depth - 1);
							cells[i] = V_EMPTY;
							if (five)
								return 1;
							outcome = std::max(outcome, v);
						}
					return outcome;
				}
				bool has_five_for(const uint8_t *cells, int n, int) const
				{ // the reference only ever looks for the ATTACKER's five, whoever is to move (DefensiveMoveTable.cpp:175-216)
					return has_five(cells, n);
				}
				uint16_t refutations(uint32_t encoded, int n, int bit_offset, int depth) const
				{
					uint8_t cells[16];
					for (int i = 0; i < n; i++)
						cells[i] = (encoded >> (2 * i)) & 3;
					if (has_five(cells, n) || best_for(cells, n, attacker, depth) == 0)
						return 0;
					uint16_t mask = 0;
					for (int i = 0; i < n; i++)
						if (cells[i] == V_EMPTY)
						{
							cells[i] = static_cast<uint8_t>(defender);
							if (best_for(cells, n, attacker, depth) != 1)
								mask |= static_cast<uint16_t>(1u << (bit_offset + i));
							cells[i] = V_EMPTY;
						}
					return mask;
				}
		};
		const uint32_t FIVE_SHAPES[5] = { 85u, 277u, 325u, 337u, 340u };          // cross stones, 5 cells with one gap
		const uint32_t OPEN4_SHAPES[4] = { 84u, 276u, 324u, 336u };              // 6 cells
		const uint32_t DOUBLE4_SHAPES[6] = { 4177u, 4369u, 4417u, 20549u, 20741u, 86037u };
		const int DOUBLE4_LENGTH[6] = { 7, 7, 7, 8, 8, 9 };
		const int DOUBLE4_BEGIN[6] = { 2, 3, 4, 2, 3, 2 };
	}

	void build_host_tables(int rules, HostTables &t)
	{
		t.rules = rules;
		// ---- line patterns ----
		RuleSet cross_sets[K_COUNT], circle_sets[K_COUNT];
		build_rule_sets(rules, V_CROSS, cross_sets);
		build_rule_sets(rules, V_CIRCLE, circle_sets);
		t.pattern.assign(1u << 20, 0);
		t.half_open_three.assign(1u << 20, 0);
		for (uint32_t idx = 0; idx < (1u << 20); idx++)
		{
			const uint32_t line = (idx & 1023u) | ((idx & 1047552u) << 2u); // re-insert the (empty) centre cell
			uint32_t occ[4] = { 0, 0, 0, 0 };
			for (int j = 0; j < 11; j++)
				occ[(line >> (2 * j)) & 3] |= (1u << j);
			// off-board cells must form a prefix / suffix of the line (Pattern.hpp:52-63)
			const uint32_t wall = occ[V_WALL];
			const uint32_t left = wall & 31u, right = (wall >> 6) & 31u;
			if ((left & (left + 1u)) != 0u)
				continue; // left walls must be bits 0..k-1
			bool right_ok = true;
			for (int j = 0; j < 4; j++)
				if (((right >> j) & 1u) && !((right >> (j + 1)) & 1u))
					right_ok = false; // right walls must be bits k..4
			if (!right_ok)
				continue;
			occ[V_EMPTY] &= ~(1u << 5);
			uint32_t oc[4] = { occ[0], occ[1] | (1u << 5), occ[2], occ[3] };
			int cross = classify(cross_sets, oc);
			uint32_t oo[4] = { occ[0], occ[1], occ[2] | (1u << 5), occ[3] };
			int circle = classify(circle_sets, oo);
			uint8_t h = 0;
			if (cross == 1)
			{
				h |= 1;
				cross = 0;
			}
			if (circle == 1)
			{
				h |= 2;
				circle = 0;
			}
			t.pattern[idx] = static_cast<uint8_t>(cross | (circle << 4));
			t.half_open_three[idx] = h;
		}
		// ---- threats ----
		t.threat.assign(4096 * 2, 0);
		for (int i = 0; i < 4096; i++)
		{
			int n[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
			n[i & 7]++;
			n[(i >> 3) & 7]++;
			n[(i >> 6) & 7]++;
			n[(i >> 9) & 7]++;
			const int fours = n[4] + n[3];
			const bool fork44 = n[5] > 0 || fours >= 2, fork43 = n[2] >= 1 && fours >= 1, fork33 = n[2] >= 2;
			uint8_t x, o; // threat for cross / circle; ThreatType numbering of ThreatTable.hpp:22-34
			auto both = [&](int v) { x = o = static_cast<uint8_t>(v); };
			if (n[6] > 0) both(8);
			else if (rules == AGX_RENJU && n[7] > 0) { x = 9; o = 8; }
			else if (fork44) both(6);
			else if (n[4] > 0) { both(7); if (rules == AGX_RENJU && fork33) x = 3; }
			else if (fork43) { both(5); if (rules == AGX_RENJU && fork33) x = 3; }
			else if (fork33) both(3);
			else if (n[3] > 0) both(4);
			else if (n[2] > 0) both(2);
			else if (n[1] > 0) both(1);
			else both(0);
			t.threat[2 * i] = x;
			t.threat[2 * i + 1] = o;
		}
		// ---- defensive moves ----
		t.defense.assign(15 * 256 * 2, 0);
		for (int defender = 1; defender <= 2; defender++)
		{
			LineGame lg;
			lg.defender = defender;
			lg.attacker = 3 - defender;
			lg.allow_overline = (rules == AGX_FREESTYLE) || (rules == AGX_RENJU && lg.attacker == V_CIRCLE) || (rules == AGX_CARO6);
			lg.allow_blocked = (rules != AGX_CARO5 && rules != AGX_CARO6);
			auto fill = [&](int table_row, uint32_t cross_shape, int length, int begin, int depth)
			{
				const uint32_t shape = (lg.attacker == V_CROSS) ? cross_shape : 2u * cross_shape;
				for (uint32_t sides = 0; sides < 256; sides++)
				{
					const uint32_t ext = (sides & 15u) | (shape << 4) | ((sides & 0xF0u) << (2 * length));
					t.defense[(table_row * 256 + sides) * 2 + (defender - 1)] = lg.refutations(ext, length + 4, begin - 2, depth);
				}
			};
			for (int i = 0; i < 5; i++)
				fill(i, FIVE_SHAPES[i], 5, 2 + i, 1);
			for (int i = 0; i < 4; i++)
				fill(5 + i, OPEN4_SHAPES[i], 6, 2 + i, 3);
			for (int i = 0; i < 6; i++)
				fill(9 + i, DOUBLE4_SHAPES[i], DOUBLE4_LENGTH[i], DOUBLE4_BEGIN[i], 3);
		}
	}
}
