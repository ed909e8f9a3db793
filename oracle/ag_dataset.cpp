/*
 * oracle/ag_dataset.cpp — TEST INFRASTRUCTURE ONLY.  See ag_dataset.hpp for the reference lines each function follows.
 */
#include "ag_dataset.hpp"

#include <algorithm>

namespace ago
{
	namespace
	{
		template<typename T>
		void put(std::vector<uint8_t> &out, T v)
		{ // SerializedObject::save<T>: raw bytes of a POD appended
			const uint8_t *p = reinterpret_cast<const uint8_t*>(&v);
			out.insert(out.end(), p, p + sizeof(T));
		}
		template<typename T>
		T get(const uint8_t *data, size_t &offset)
		{
			T v;
			std::memcpy(&v, data + offset, sizeof(T));
			offset += sizeof(T);
			return v;
		}
		Value get_valid_value(float winrate, float drawrate)
		{ // SearchDataStorage.cpp:52-61
			const float tmp = winrate + drawrate;
			if (tmp > 1.0f)
			{
				winrate /= tmp;
				drawrate /= tmp;
			}
			return Value(winrate, drawrate);
		}
	}

	uint8_t score_to_int8(Score s)
	{ // SearchDataStorage.cpp:24-31
		const uint32_t pv = static_cast<uint32_t>(s.pv()) << 6;
		if (s.is_proven())
			return static_cast<uint8_t>(pv | static_cast<uint32_t>(std::max(0, std::min(63, s.distance()))));
		else
			return static_cast<uint8_t>(pv | score_format::to_lowp(s.eval() / 1000.0f));
	}
	Score int8_to_score(uint8_t x)
	{ // SearchDataStorage.cpp:32-50
		const int pv = x >> 6;
		const uint32_t eval = x & 63u;
		switch (pv)
		{
			case PV_LOSS:
				return Score::loss_in(eval);
			case PV_DRAW:
				return Score::draw_in(eval);
			case PV_UNKNOWN:
				return Score(static_cast<int>(1000.0f * score_format::to_fp32(eval) + 0.5f));
			case PV_WIN:
				return Score::win_in(eval);
			default:
				return Score();
		}
	}

	void SearchDataStorage_v201::load_from(const SearchDataPack &pack)
	{ // SearchDataStorage.cpp:326-374
		move_number = 0;
		size_t entries_count = 0;
		policy_scale = 0.0f;
		value_scale = 0.0f;
		visit_scale = 1.0f;
		int last_idx = 0;
		for (int i = 0; i < pack.size(); i++)
		{
			if (pack.visit_count[i] > 0 || pack.action_scores[i].is_proven() || (i - last_idx) >= 255)
			{
				entries_count++;
				last_idx = i;
			}
			move_number += static_cast<int>(pack.board[i] != NONE);
			policy_scale = std::max(policy_scale, pack.policy_prior[i]);
			value_scale = std::max(value_scale, std::max(pack.action_values[i].win, pack.action_values[i].draw));
			visit_scale = std::max(visit_scale, static_cast<float>(pack.visit_count[i]));
		}
		storage.assign(entries_count, entry());

		policy_scale = (policy_scale == 0.0f) ? 1.0f : (policy_scale / policy_format::max());
		value_scale = (value_scale == 0.0f) ? 1.0f : (value_scale / policy_format::max());
		visit_scale /= visit_format::max();

		minimax_score = pack.minimax_score;
		entries_count = 0;

		last_idx = 0;
		for (int i = 0; i < pack.size(); i++)
			if (pack.visit_count[i] > 0 || pack.action_scores[i].is_proven() || (i - last_idx) >= 255)
			{
				const int visits = pack.visit_count[i];
				const Value value = pack.action_values[i];
				const Score score = pack.action_scores[i];

				storage[entries_count].location_delta = static_cast<uint8_t>(i - last_idx);
				storage[entries_count].visit_count = static_cast<uint8_t>(visit_format::to_lowp(visits / visit_scale));
				storage[entries_count].policy_prior = static_cast<uint8_t>(policy_format::to_lowp(pack.policy_prior[i] / policy_scale));
				storage[entries_count].score = score_to_int8(score);
				storage[entries_count].win_rate = static_cast<uint8_t>(value_format::to_lowp(value.win / value_scale));
				storage[entries_count].draw_rate = static_cast<uint8_t>(value_format::to_lowp(value.draw / value_scale));
				entries_count++;
				last_idx = i;
			}
		flags = pack.flags;
	}
	void SearchDataStorage_v201::store_to(SearchDataPack &pack) const
	{ // SearchDataStorage.cpp:375-409
		int current_idx = 0;
		float win_rate = 0.0f, draw_rate = 0.0f;
		int sum_visits = 0;
		for (size_t i = 0; i < storage.size(); i++)
		{
			const int loc_delta = storage[i].location_delta;
			current_idx += loc_delta;
			const float visits = visit_format::to_fp32(storage[i].visit_count) * visit_scale + 0.5f;
			pack.visit_count[current_idx] = static_cast<int>(visits);

			const Value q = get_valid_value(value_format::to_fp32(storage[i].win_rate) * value_scale, value_format::to_fp32(storage[i].draw_rate) * value_scale);
			pack.action_values[current_idx] = q;
			pack.action_scores[current_idx] = int8_to_score(storage[i].score);
			pack.policy_prior[current_idx] = policy_format::to_fp32(storage[i].policy_prior) * policy_scale;

			sum_visits += visits; // int += float, as in the reference
			win_rate += q.win * visits;
			draw_rate += q.draw * visits;
		}

		pack.minimax_score = minimax_score;
		if (sum_visits == 0)
			pack.minimax_value = minimax_score.to_value();
		else
			pack.minimax_value = get_valid_value(win_rate / sum_visits, draw_rate / sum_visits);

		pack.flags = flags;
	}
	void SearchDataStorage_v201::serialize(std::vector<uint8_t> &out) const
	{ // SearchDataStorage.cpp:410-419; serializeVector (file_util.hpp:26-30)
		put<uint16_t>(out, static_cast<uint16_t>(fp16_format::to_lowp(value_scale)));
		put<uint16_t>(out, static_cast<uint16_t>(fp16_format::to_lowp(policy_scale)));
		put<uint16_t>(out, static_cast<uint16_t>(fp16_format::to_lowp(visit_scale)));
		put<uint16_t>(out, minimax_score.d);
		put<uint16_t>(out, move_number);
		put<uint16_t>(out, flags);
		put<uint32_t>(out, static_cast<uint32_t>(storage.size()));
		for (const entry &e : storage)
		{
			out.push_back(e.location_delta);
			out.push_back(e.visit_count);
			out.push_back(e.policy_prior);
			out.push_back(e.score);
			out.push_back(e.win_rate);
			out.push_back(e.draw_rate);
		}
	}
	size_t SearchDataStorage_v201::parse(const uint8_t *data, size_t offset)
	{ // SearchDataStorage_v201(const SerializedObject&, size_t&) (SearchDataStorage.cpp:300-320); unserializeVector (file_util.hpp:31-41)
		value_scale = fp16_format::to_fp32(get<uint16_t>(data, offset));
		policy_scale = fp16_format::to_fp32(get<uint16_t>(data, offset));
		visit_scale = fp16_format::to_fp32(get<uint16_t>(data, offset));
		minimax_score = Score::raw(get<uint16_t>(data, offset));
		move_number = get<uint16_t>(data, offset);
		flags = get<uint16_t>(data, offset);
		const uint32_t size = get<uint32_t>(data, offset);
		storage.assign(size, entry());
		for (uint32_t i = 0; i < size; i++)
		{
			storage[i].location_delta = data[offset++];
			storage[i].visit_count = data[offset++];
			storage[i].policy_prior = data[offset++];
			storage[i].score = data[offset++];
			storage[i].win_rate = data[offset++];
			storage[i].draw_rate = data[offset++];
		}
		return offset;
	}

	void serialize_game_v201(const std::vector<SearchDataStorage_v201> &samples, const std::vector<uint16_t> &played_moves, int outcome, int rows, int cols,
			std::vector<uint8_t> &out)
	{ // GameDataStorage::serialize, case 201 (GameDataStorage.cpp:217-250)
		put<uint32_t>(out, static_cast<uint32_t>(samples.size()));
		for (const SearchDataStorage_v201 &s : samples)
			s.serialize(out);
		put<uint32_t>(out, static_cast<uint32_t>(played_moves.size()));
		for (uint16_t m : played_moves)
			put<uint16_t>(out, m);
		put<int>(out, outcome);
		put<int>(out, rows);
		put<int>(out, cols);
	}
}
