/*
 * sample_v201.hpp — the byte format of one self-play sample, "format 201" of the reference's dataset
 * (SearchDataStorage_v201, src/dataset/SearchDataStorage.cpp:300-419), for host and device code.
 *
 * Layout written by SearchDataStorage_v201::serialize (:410-419) — SerializedObject::save<T> appends the raw bytes of T:
 *   u16 value_scale   u16 policy_scale   u16 visit_scale      (the three float scales as fp16_format codes)
 *   u16 minimax_score (Score raw bits)   u16 move_number (stones on the board)   u16 flags (bit 0 statically solved, 1 recursively
 *   solved, 2 must defend)   u32 entry count   then 6 bytes per entry:
 *   u8 location_delta (cell index minus the previous entry's, the first one counts from 0)  u8 visits  u8 prior  u8 score  u8 win  u8 draw
 * An entry exists for every cell (row-major) whose edge was visited or carries a proven score, and for any cell that is 255 or more
 * cells past the previous entry (loadFrom, :326-374).
 *
 * The small floating-point formats are the reference's LowFP<S, E, M, B> (include/alphagomoku/utils/low_precision.hpp:20-170):
 * S sign bits, E exponent bits with bias B (smallest exponent = B, the subnormal range), M mantissa bits.  encode() rounds to
 * nearest by adding one half before truncation and saturates the mantissa, exactly like LowFP::to_lowp (:120-129); the power of
 * two is built from its bit pattern instead of std::ldexp (same value).
 */
#ifndef AGX_SAMPLE_V201_HPP_
#define AGX_SAMPLE_V201_HPP_

#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define AGX_V201_HD __host__ __device__ __forceinline__
#else
#define AGX_V201_HD inline
#endif

namespace agx
{
	namespace v201
	{
		AGX_V201_HD uint32_t float_bits(float x)
		{
			uint32_t b;
			memcpy(&b, &x, 4);
			return b;
		}
		AGX_V201_HD float bits_float(uint32_t b)
		{
			float x;
			memcpy(&x, &b, 4);
			return x;
		}
		AGX_V201_HD float pow2(int e) { return bits_float(static_cast<uint32_t>(127 + e) << 23); } // -126 <= e <= 127

		template<int S, int E, int M, int B>
		struct SmallFloat
		{
				static constexpr int bits = S + E + M;
				static constexpr int exp_min = B, exp_max = (1 << E) - 1 + B;
				AGX_V201_HD static uint32_t encode(float x)
				{
					const uint32_t raw = float_bits(x);
					const uint32_t sign = (S == 1) ? ((raw & 0x80000000u) >> (32 - bits)) : 0u;
					int e = static_cast<int>((raw >> 23) & 255u) - 127;
					e = (e < exp_min) ? exp_min : ((e > exp_max) ? exp_max : e);
					const int sub = (e == exp_min) ? 1 : 0;
					const float magnitude = (sign == 0) ? x : -x;
					const float frac = magnitude * pow2(-(e + sub)) + static_cast<float>(sub) - 1.0f;
					uint32_t m = static_cast<uint32_t>(frac * static_cast<float>(1 << M) + 0.5f);
					if (m > (1u << M) - 1u)
						m = (1u << M) - 1u;
					return sign | (static_cast<uint32_t>(e - B) << M) | m;
				}
				AGX_V201_HD static float decode(uint32_t code)
				{ // LowFP::convert_to_fp32 (:151-158)
					const bool negative = (S == 1) && ((code >> (E + M)) & 1u);
					const int e = static_cast<int>((code >> M) & ((1u << E) - 1u)) + B;
					const float frac = static_cast<float>(code & ((1u << M) - 1u)) / static_cast<float>(1 << M);
					const int sub = (e == exp_min) ? 1 : 0;
					return (negative ? -1.0f : 1.0f) * (static_cast<float>(1 - sub) + frac) / pow2(-(e + sub));
				}
				AGX_V201_HD static float largest() { return decode((S == 0) ? ((1u << bits) - 1u) : ((1u << (bits - 1)) - 1u)); }
		};
		typedef SmallFloat<1, 3, 2, -8> ScoreFormat;   // SearchDataStorage.cpp:22
		typedef SmallFloat<0, 3, 5, -8> VisitFormat;   // :161
		typedef SmallFloat<0, 4, 4, -16> PriorFormat;  // :162 (policy_format; value_format :163 is the same format)
		typedef SmallFloat<0, 5, 11, -16> ScaleFormat; // :164 (fp16_format: 16 bits, NOT IEEE half)

		constexpr int HEADER_BYTES = 16, ENTRY_BYTES = 6;

		/* score_to_int8 (SearchDataStorage.cpp:24-31) on Score raw bits */
		AGX_V201_HD uint32_t score_code(uint32_t score)
		{
			const uint32_t pv = (score >> 13) & 3u;
			const int eval = static_cast<int>(score & 8191u) - 4000;
			const bool proven = (pv != 2u) && score != 0u && score != 0xFFFFu; // Score::isProven (Score.hpp)
			if (proven)
			{
				int distance = (pv == 3u) ? -eval : eval; // Score::getDistance
				distance = (distance < 0) ? 0 : ((distance > 63) ? 63 : distance);
				return (pv << 6) | static_cast<uint32_t>(distance);
			}
			return ((pv << 6) | ScoreFormat::encode(static_cast<float>(eval) / 1000.0f)) & 255u;
		}
		/* int8_to_score (:32-50) -> Score raw bits */
		AGX_V201_HD uint32_t score_from_code(uint32_t code)
		{
			const uint32_t pv = (code >> 6) & 3u, low = code & 63u;
			switch (pv)
			{
				case 0: return (0u << 13) | (4000u + low);                 // Score::loss_in(n)
				case 1: return (1u << 13) | (4000u + low);                 // Score::draw_in(n)
				case 3: return (3u << 13) | (4000u - low);                 // Score::win_in(n)
				default: return (2u << 13) | static_cast<uint32_t>(4000 + static_cast<int>(1000.0f * ScoreFormat::decode(low) + 0.5f));
			}
		}
		/* the three scales of loadFrom (:340-342) from the maxima over the board */
		AGX_V201_HD float prior_scale(float max_prior) { return (max_prior == 0.0f) ? 1.0f : (max_prior / PriorFormat::largest()); }
		AGX_V201_HD float value_scale(float max_value) { return (max_value == 0.0f) ? 1.0f : (max_value / PriorFormat::largest()); }
		AGX_V201_HD float visit_scale(float max_visits) { return max_visits / VisitFormat::largest(); } // max_visits >= 1
	}
}

#endif
