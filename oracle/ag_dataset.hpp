/*
 * oracle/ag_dataset.hpp — TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product).
 *
 * Literal restatement of the self-play record sink of the reference:
 *   LowFP<S,E,M,B>                       include/alphagomoku/utils/low_precision.hpp:20-170
 *   score_to_int8 / int8_to_score        src/dataset/SearchDataStorage.cpp:24-50
 *   get_valid_value                      src/dataset/SearchDataStorage.cpp:52-61
 *   SearchDataPack(const Node&, board)   src/dataset/data_packs.cpp:24-43
 *   SearchDataStorage_v201::loadFrom     src/dataset/SearchDataStorage.cpp:326-374
 *   SearchDataStorage_v201::storeTo      src/dataset/SearchDataStorage.cpp:375-409
 *   SearchDataStorage_v201::serialize    src/dataset/SearchDataStorage.cpp:410-419  (+ the parsing constructor :300-320)
 *   GameDataStorage::serialize (201)     src/dataset/GameDataStorage.cpp:217-250    (+ the parsing constructor :27-69)
 * SerializedObject::save<T> is a raw POD append (the loaders walk `offset += sizeof(T)`), serializeVector is a u32 count + the raw
 * array (include/alphagomoku/utils/file_util.hpp:26-41).
 *
 * Pinned by: LowFP against the reference header compiled into oracle/_ref (tests/test_oracle_dataset.py: every code of the four
 * formats, 200 k floats per format).  score_to_int8, loadFrom/storeTo and the framing live in a TU that includes
 * <minml/utils/serialization.hpp> (absent) — restated, parity unpinned beyond LowFP and the Score algebra.
 */
#ifndef AG_DATASET_HPP_
#define AG_DATASET_HPP_

#include "agoracle.hpp"

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace ago
{
	template<int S, int E, int M, int B>
	struct LowFP
	{ // low_precision.hpp:20-170
			static constexpr uint32_t ones(uint32_t n) { return (n == 32) ? 0xFFFFFFFFu : ((n == 0) ? 0u : ((1u << n) - 1u)); }
			static constexpr int max_exponent = (1 << E) - 1 + B;
			static constexpr int min_exponent = B;
			static constexpr uint32_t max_mantissa = (1u << M) - 1u;
			static constexpr uint32_t sign_mask = (S == 1) ? (1u << (E + M)) : 0x00u;
			static constexpr uint32_t exponent_mask = ones(E) << M;
			static constexpr uint32_t mantissa_mask = ones(M);
			static constexpr int bitsize() { return S + E + M; }
			static float get_scale(int e) { return std::ldexp(1.0f, -e); }
			static uint32_t to_lowp(float x)
			{ // :120-129
				uint32_t bits;
				std::memcpy(&bits, &x, 4);
				const uint32_t sign = (S == 1) ? ((bits & 0x80000000u) >> (32u - bitsize())) : 0u;
				const int exponent = std::max(min_exponent, std::min(max_exponent, static_cast<int>((bits & 0x7F800000u) >> 23u) - 127));
				const int is_subnormal = (exponent == min_exponent) ? 1 : 0;
				const float base = ((sign == 0) ? x : (-x)) * get_scale(exponent + is_subnormal) + is_subnormal - 1;
				const uint32_t mantissa = std::min(max_mantissa, static_cast<uint32_t>(base * (1 << M) + 0.5f));
				return sign | (static_cast<uint32_t>(exponent - B) << M) | mantissa;
			}
			static float to_fp32(uint32_t x)
			{ // convert_to_fp32 :151-158 (the table of :130-139 holds the same values)
				const uint32_t sign = (S == 1) ? (x & sign_mask) : 0u;
				const int exponent = ((static_cast<uint32_t>(x) & exponent_mask) >> M) + B;
				const float base = static_cast<float>(x & mantissa_mask) / (1 << M);
				const int is_subnormal = (exponent == min_exponent) ? 1 : 0;
				return ((sign == 0) ? 1.0f : -1.0f) * (1 - is_subnormal + base) / get_scale(exponent + is_subnormal);
			}
			static float max() { return to_fp32((S == 0) ? ones(bitsize()) : ones(bitsize() - 1)); } // :102-105
	};
	using score_format = LowFP<1, 3, 2, -8>;   // SearchDataStorage.cpp:22
	using visit_format = LowFP<0, 3, 5, -8>;   // :161
	using policy_format = LowFP<0, 4, 4, -16>; // :162
	using value_format = LowFP<0, 4, 4, -16>;  // :163
	using fp16_format = LowFP<0, 5, 11, -16>;  // :164

	uint8_t score_to_int8(Score s);
	Score int8_to_score(uint8_t x);

	struct SearchDataPack
	{ // include/alphagomoku/dataset/data_packs.hpp, filled by data_packs.cpp:24-43
			int rows = 0, cols = 0;
			std::vector<Sign> board;
			std::vector<float> policy_prior;
			std::vector<int> visit_count;
			std::vector<Value> action_values;
			std::vector<Score> action_scores;
			Value minimax_value;
			Score minimax_score;
			uint16_t flags = 0;
			SearchDataPack(int r, int c) :
					rows(r), cols(c), board(r * c, NONE), policy_prior(r * c, 0.0f), visit_count(r * c, 0), action_values(r * c), action_scores(r * c)
			{
			}
			int size() const { return rows * cols; }
	};

	struct SearchDataStorage_v201
	{ // SearchDataStorage.hpp:83-121
			struct entry
			{
					uint8_t location_delta = 0, visit_count = 0, policy_prior = 0, score = 0, win_rate = 0, draw_rate = 0;
			};
			std::vector<entry> storage;
			float value_scale = 0.0f, policy_scale = 0.0f, visit_scale = 1.0f;
			Score minimax_score;
			uint16_t move_number = 0;
			uint16_t flags = 0;
			void load_from(const SearchDataPack &pack);
			void store_to(SearchDataPack &pack) const;
			void serialize(std::vector<uint8_t> &out) const;
			size_t parse(const uint8_t *data, size_t offset); // returns the new offset
	};

	/* GameDataStorage::serialize, format 201 */
	void serialize_game_v201(const std::vector<SearchDataStorage_v201> &samples, const std::vector<uint16_t> &played_moves, int outcome, int rows, int cols,
			std::vector<uint8_t> &out);
}

#endif
