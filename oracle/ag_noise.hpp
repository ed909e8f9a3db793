/*
 * oracle/ag_noise.hpp — TEST INFRASTRUCTURE ONLY.  CPU restatement of the PUCT selector's root-noise generators
 * (createCustomNoise / createDirichletNoise / createGumbelNoise, src/utils/random.cpp:89-124, applied by EdgeSelector.cpp:602-623).
 * The reference draws from a time-seeded mt19937 and uses libm, so its streams cannot be replayed: the distributions are restated
 * over a counter-based stream and fixed-series log / exp in double precision (only + - * /), which is what the device computes too.
 * PARITY UNPINNED against the reference (no fixture can exist for a time-seeded generator); device-vs-oracle parity is bit-exact.
 */
#ifndef AG_NOISE_HPP_
#define AG_NOISE_HPP_

#include <cstdint>
#include <cstring>

namespace ago
{
	inline uint64_t noise_mix(uint64_t z)
	{ // splitmix64 finaliser
		z += 0x9E3779B97F4A7C15ull;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		return z ^ (z >> 31);
	}
	struct NoiseStream
	{
			uint64_t base;
			uint32_t k;
			NoiseStream(uint64_t seed, int serial, int move_number) :
					base(seed ^ (static_cast<uint64_t>(static_cast<uint32_t>(serial)) << 32) ^ (static_cast<uint64_t>(static_cast<uint32_t>(move_number)) << 20)),
					k(0)
			{
			}
			uint64_t next()
			{
				return noise_mix(base ^ k++);
			}
			float uniform_float()
			{ // [0, 1), 24 bits, like std::uniform_real_distribution<float>
				return static_cast<float>(next() >> 40) * (1.0f / 16777216.0f);
			}
			double uniform_open()
			{ // (0, 1), 53 bits
				return (static_cast<double>(next() >> 11) + 0.5) * (1.0 / 9007199254740992.0);
			}
	};

	inline double det_log(double x)
	{ // natural logarithm of a positive normal double: exponent split + atanh series (|t| <= 0.172, 10 terms, ~1e-16)
		uint64_t bits;
		memcpy(&bits, &x, 8);
		int e = static_cast<int>((bits >> 52) & 2047u) - 1023;
		bits = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
		double m;
		memcpy(&m, &bits, 8);
		if (m > 1.4142135623730951)
		{
			m *= 0.5;
			e += 1;
		}
		const double t = (m - 1.0) / (m + 1.0), t2 = t * t;
		double s = 1.0 / 19.0;
		s = s * t2 + 1.0 / 17.0;
		s = s * t2 + 1.0 / 15.0;
		s = s * t2 + 1.0 / 13.0;
		s = s * t2 + 1.0 / 11.0;
		s = s * t2 + 1.0 / 9.0;
		s = s * t2 + 1.0 / 7.0;
		s = s * t2 + 1.0 / 5.0;
		s = s * t2 + 1.0 / 3.0;
		s = s * t2 + 1.0;
		return 2.0 * t * s + static_cast<double>(e) * 0.6931471805599453;
	}
	inline double det_exp(double x)
	{ // exp by range reduction to |r| <= ln2 / 2 and a degree-13 Taylor polynomial; results below 2^-1000 flush to 0
		if (x < -690.0)
			return 0.0;
		if (x > 700.0)
			x = 700.0;
		const double y = x * 1.4426950408889634 + 0.5;
		int k = static_cast<int>(y);
		if (static_cast<double>(k) > y)
			k--;
		const double r = (x - static_cast<double>(k) * 0.693147180369123816490) - static_cast<double>(k) * 1.90821492927058770002e-10; // ln 2 in two parts
		double p = 1.0 / 6227020800.0;
		p = p * r + 1.0 / 479001600.0;
		p = p * r + 1.0 / 39916800.0;
		p = p * r + 1.0 / 3628800.0;
		p = p * r + 1.0 / 362880.0;
		p = p * r + 1.0 / 40320.0;
		p = p * r + 1.0 / 5040.0;
		p = p * r + 1.0 / 720.0;
		p = p * r + 1.0 / 120.0;
		p = p * r + 1.0 / 24.0;
		p = p * r + 1.0 / 6.0;
		p = p * r + 0.5;
		p = p * r + 1.0;
		p = p * r + 1.0;
		const uint64_t bits = static_cast<uint64_t>(static_cast<int64_t>(k + 1023)) << 52;
		double scale;
		memcpy(&scale, &bits, 8);
		return p * scale;
	}
	inline double det_normal(NoiseStream &rng)
	{ // Marsaglia polar method; sqrt(y) as exp(log(y) / 2) to stay inside the deterministic subset
		while (true)
		{
			const double a = 2.0 * rng.uniform_open() - 1.0, b = 2.0 * rng.uniform_open() - 1.0;
			const double s = a * a + b * b;
			if (s >= 1.0 || s < 1.0e-300)
				continue;
			return a * det_exp(0.5 * det_log(-2.0 * det_log(s) / s));
		}
	}
	inline double det_gamma(NoiseStream &rng, double alpha)
	{ // Gamma(alpha, 1), 0 < alpha < 1: Gamma(alpha + 1) by Marsaglia-Tsang, times U^(1 / alpha)
		const double d = (alpha + 1.0) - 1.0 / 3.0;
		const double c = 1.0 / det_exp(0.5 * det_log(9.0 * d));
		double g;
		while (true)
		{
			const double x = det_normal(rng);
			double v = 1.0 + c * x;
			if (v <= 0.0)
				continue;
			v = v * v * v;
			const double u = rng.uniform_open();
			if (det_log(u) < 0.5 * x * x + d - d * v + d * det_log(v))
			{
				g = d * v;
				break;
			}
		}
		return g * det_exp(det_log(rng.uniform_open()) / alpha);
	}

	/* out[i] = noisy prior of edge i.  type: 1 custom, 2 dirichlet, 3 gumbel.  `priors` is read through `prior_of(i)`. */
	template<typename PriorFn>
	inline void make_root_noise(int type, float weight, uint64_t seed, int serial, int move_number, int n, PriorFn prior_of, float *out)
	{
		NoiseStream rng(seed, serial, move_number);
		if (type == 1)
		{ // createCustomNoise (random.cpp:89-100) + applyCustomNoise (EdgeSelector.cpp:602-608)
			float sum = 0.0f;
			for (int i = 0; i < n; i++)
			{
				double p = static_cast<double>(rng.uniform_float());
				p = p * p;
				p = p * p;
				out[i] = static_cast<float>(p * static_cast<double>(1.0f - sum));
				sum += out[i];
			}
			for (int i = n - 1; i > 0; i--)
			{ // std::shuffle's role: a plain Fisher-Yates
				const int j = static_cast<int>(static_cast<uint32_t>(rng.next() >> 32) % static_cast<uint32_t>(i + 1));
				const float t = out[i];
				out[i] = out[j];
				out[j] = t;
			}
			for (int i = 0; i < n; i++)
				out[i] = (1.0f - weight) * prior_of(i) + weight * out[i];
		}
		else if (type == 2)
		{ // createDirichletNoise(n, 0.05) (random.cpp:101-116) + applyDirichletNoise (EdgeSelector.cpp:609-615)
			float sum = 0.0f;
			for (int i = 0; i < n; i++)
			{
				out[i] = static_cast<float>(det_gamma(rng, 0.05));
				sum += out[i];
			}
			if (sum > 0.0f)
			{
				sum = 1.0f / sum;
				for (int i = 0; i < n; i++)
					out[i] *= sum;
			}
			else
			{ // every draw underflowed (possible for one or two edges): the reference would divide by zero; use a flat vector
				for (int i = 0; i < n; i++)
					out[i] = 1.0f / static_cast<float>(n);
			}
			for (int i = 0; i < n; i++)
				out[i] = (1.0f - weight) * prior_of(i) + weight * out[i];
		}
		else
		{ // createGumbelNoise (random.cpp:117-123) + applyGumbelNoise (EdgeSelector.cpp:616-623): softmax(log_eps(prior) + w * g)
			const float eps = 1.1920929e-07f; // std::numeric_limits<float>::epsilon()
			float max_value = -3.402823466e+38f;
			for (int i = 0; i < n; i++)
			{
				const float inner = static_cast<float>(det_log(static_cast<double>(eps + rng.uniform_float())));
				const float g = -static_cast<float>(det_log(static_cast<double>(eps + (-inner))));
				out[i] = static_cast<float>(det_log(static_cast<double>(eps + prior_of(i)))) + weight * g;
				max_value = (out[i] > max_value) ? out[i] : max_value;
			}
			float sum = 0.0f;
			for (int i = 0; i < n; i++)
			{
				out[i] = static_cast<float>(det_exp(static_cast<double>(out[i] - max_value)));
				sum += out[i];
			}
			sum = 1.0f / sum;
			for (int i = 0; i < n; i++)
				out[i] *= sum;
		}
	}
}

#endif
