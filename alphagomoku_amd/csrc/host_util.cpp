/*
 * host_util.cpp — host-side helpers of the C ABI that need no GPU: synthetic openings.
 *
 * agx_make_opening restates the distribution of the reference's prepareOpening (src/utils/misc.cpp:142-170, with
 * generateOpeningMap :108-141 and randomizeMove :84-103): max(1, U[0,6)+U[0,6)+U[0,6)) stones (1 in 1000: none), the first one
 * drawn proportionally to 1.5^-(distance to the board centre - 1), the following ones proportionally to
 * sum over stones of t^-(distance - 1) with t ~ U[2,3) (plus 1e-6 on every empty cell); an opening whose last stone already
 * ends the game is rejected and redrawn.  The random stream is std::mt19937(seed) — the reference uses time-seeded thread-local
 * generators (src/utils/random.cpp:17-23), so streams cannot be matched, only the distribution.
 */
#include "agx_internal.hpp"
#include "tables_host.hpp"
#include "renju_static.hpp"

#include <cmath>
#include <cstring>
#include <mutex>
#include <memory>
#include <random>
#include <vector>

namespace
{
	const agx::HostTables& tables_for(int rules)
	{
		static std::mutex mtx;
		static std::unique_ptr<agx::HostTables> cache[5];
		std::lock_guard<std::mutex> lock(mtx);
		if (!cache[rules])
		{
			cache[rules].reset(new agx::HostTables());
			agx::build_host_tables(rules, *cache[rules]);
		}
		return *cache[rules];
	}
	bool last_move_wins(const agx::HostTables &t, const std::vector<uint8_t> &board, int n, int r, int c, int sign)
	{ // getOutcome's win test (src/game/rules.cpp:110-122): a FIVE for the mover in any direction through the last stone
		const int dr[4] = { 0, 1, 1, 1 }, dc[4] = { 1, 0, 1, -1 };
		for (int d = 0; d < 4; d++)
		{
			uint32_t pattern = 0;
			for (int k = -5, sh = 0; k <= 5; k++, sh += 2)
			{
				const int rr = r + k * dr[d], cc = c + k * dc[d];
				uint32_t v = (rr >= 0 && rr < n && cc >= 0 && cc < n) ? board[rr * n + cc] : 3u;
				if (k == 0)
					v = 0;
				pattern |= v << sh;
			}
			const uint32_t idx = (pattern & 1023u) | ((pattern & 4190208u) >> 2);
			const uint8_t e = t.pattern[idx];
			if (((sign == 1) ? (e & 15) : (e >> 4)) == 6)
				return true;
		}
		return false;
	}
}

extern "C" int agx_make_opening(int rules, int board_size, uint32_t seed, uint16_t *h_opening)
{
	AGX_REQUIRE(h_opening != nullptr, AGX_ERR_INVALID, "agx_make_opening: null output");
	AGX_REQUIRE(rules >= 0 && rules <= AGX_CARO6, AGX_ERR_UNSUPPORTED, "agx_make_opening: rules %d not supported", rules);
	AGX_REQUIRE(board_size >= 5 && board_size <= 20, AGX_ERR_INVALID, "agx_make_opening: board size %d", board_size);
	const agx::HostTables &tables = tables_for(rules);
	const int n = board_size, hw = n * n;
	std::mt19937 rng(seed);
	auto rand_int = [&](int m) { return static_cast<int>(rng() % static_cast<uint32_t>(m)); };
	auto rand_float = [&]() { return static_cast<float>(rng() >> 8) * (1.0f / 16777216.0f); };
	std::vector<float> dist(hw, 0.0f);
	std::vector<uint8_t> board(hw);
	while (true)
	{
		std::fill(board.begin(), board.end(), 0);
		std::vector<uint16_t> moves;
		int sign = 1, last_r = 0, last_c = 0;
		int count = std::max(1, rand_int(6) + rand_int(6) + rand_int(6));
		if (rand_int(1000) == 0)
			count = 0;
		for (int k = 0; k < count; k++)
		{
			if (moves.empty())
			{ // generateOpeningMap on an empty board (misc.cpp:111-120) ACCUMULATES into the map, and prepareOpening does not clear the
			  // map between attempts (:144): after a rejected attempt the first stone is drawn from the previous map plus the centre map
				for (int i = 0; i < n; i++)
					for (int j = 0; j < n; j++)
					{
						const float d = static_cast<float>(std::hypot(0.5 + i - 0.5 * n, 0.5 + j - 0.5 * n) - 1);
						dist[i * n + j] += static_cast<float>(std::pow(1.5f, -d));
					}
			}
			else
			{
				for (int i = 0; i < hw; i++)
					dist[i] = (board[i] != 0) ? 0.0f : 1.0e-6f;
				const float t = 2.0f + rand_float();
				for (int p = 0; p < n; p++)
					for (int q = 0; q < n; q++)
						if (board[p * n + q] != 0)
							for (int i = 0; i < n; i++)
								for (int j = 0; j < n; j++)
									if (board[i * n + j] == 0)
									{
										const float d = static_cast<float>(std::hypot(static_cast<double>(i - p), static_cast<double>(j - q)) - 1);
										dist[i * n + j] += static_cast<float>(std::pow(t, -d));
									}
			}
			// randomizeMove (misc.cpp:84-103)
			float total = 0.0f;
			for (int i = 0; i < hw; i++)
				total += dist[i];
			int cell;
			if (total == 0.0f)
				cell = rand_int(hw);
			else
			{
				const float pick = total * rand_float();
				float acc = 0.0f;
				cell = 0;
				for (; cell < hw; cell++)
				{
					acc += dist[cell];
					if (pick < acc)
						break;
				}
				if (cell >= hw)
					cell = hw - 1;
				while (board[cell] != 0 && cell > 0)
					cell--; // unreachable in exact arithmetic; guards the rounding tail
			}
			board[cell] = static_cast<uint8_t>(sign);
			last_r = cell / n;
			last_c = cell % n;
			moves.push_back(static_cast<uint16_t>(sign | (last_r << 2) | (last_c << 9)));
			sign = 3 - sign;
		}
		bool undecided = moves.empty() || !last_move_wins(tables, board, n, last_r, last_c, 3 - sign);
		if (undecided && !moves.empty() && rules == AGX_RENJU && 3 - sign == 1)
			undecided = !agx::renju_static_foul(tables.pattern.data(), tables.threat.data(), board.data(), n, last_r * n + last_c);
		if (undecided)
		{
			std::memset(h_opening, 0, AGX_OPENING_CAP * sizeof(uint16_t));
			h_opening[0] = static_cast<uint16_t>(moves.size());
			for (size_t i = 0; i < moves.size(); i++)
				h_opening[1 + i] = moves[i];
			return AGX_OK;
		}
	}
}

extern "C" int agx_get_outcome(int rules, int board_size, const uint8_t *h_board, int sign, int row, int col, int draw_after, int *outcome)
{
	AGX_REQUIRE(h_board != nullptr && outcome != nullptr, AGX_ERR_INVALID, "agx_get_outcome: null argument");
	AGX_REQUIRE(rules >= 0 && rules <= AGX_CARO6, AGX_ERR_INVALID, "agx_get_outcome: unknown rules %d", rules);
	AGX_REQUIRE(board_size >= 5 && board_size <= 20, AGX_ERR_INVALID, "agx_get_outcome: board size %d", board_size);
	AGX_REQUIRE(sign == AGX_CROSS || sign == AGX_CIRCLE, AGX_ERR_INVALID, "agx_get_outcome: sign %d", sign);
	const int n = board_size;
	*outcome = 0;
	if (!(row >= 0 && row < n && col >= 0 && col < n))
		return AGX_OK; // rules.cpp:112-113: a move outside the board decides nothing
	const agx::HostTables &tables = tables_for(rules);
	std::vector<uint8_t> board(h_board, h_board + n * n);
	if (last_move_wins(tables, board, n, row, col, sign))
		*outcome = (sign == AGX_CROSS) ? 2 : 3;
	else if (rules == AGX_RENJU && sign == AGX_CROSS && agx::renju_static_foul(tables.pattern.data(), tables.threat.data(), board.data(), n, row * n + col))
		*outcome = 3;
	else
	{
		int stones = 0;
		for (int i = 0; i < n * n; i++)
			stones += (board[i] != 0);
		if ((draw_after > 0) ? (stones >= draw_after) : (stones == n * n))
			*outcome = 1;
	}
	return AGX_OK;
}
