/*
 * selfplay_main.cpp — native self-play driver: the C++ host loop over the C ABI, no Python involved.
 * Stands where GeneratorThread::run stands in the reference (src/selfplay/GeneratorManager.cpp:124-141).
 *
 *   agx_selfplay [--games 1024] [--steps 200] [--warmup 20] [--sims 400] [--batch 8] [--blocks 6] [--filters 128] [--rules 0]
 * prints one JSON line with simulations/s.  Weights are synthetic (He-normal, fixed seed) — there are no checkpoints offline.
 */
#include "../../include/agx.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

int main(int argc, char **argv)
{
	int games = 1024, steps = 200, warmup = 20, sims = 400, batch = 8, blocks = 6, filters = 128, rules = 0, device = 0;
	for (int i = 1; i + 1 < argc; i += 2)
	{
		const std::string k = argv[i];
		const int v = std::atoi(argv[i + 1]);
		if (k == "--games") games = v;
		else if (k == "--steps") steps = v;
		else if (k == "--warmup") warmup = v;
		else if (k == "--sims") sims = v;
		else if (k == "--batch") batch = v;
		else if (k == "--blocks") blocks = v;
		else if (k == "--filters") filters = v;
		else if (k == "--rules") rules = v;
		else if (k == "--device") device = v;
		else
		{
			std::fprintf(stderr, "unknown option %s\n", argv[i]);
			return 2;
		}
	}
	try
	{
		agx::check(agx_set_device(device));
		agx::GameConfig game;
		game.rules = rules;
		agx::AGNetwork network(game, blocks, filters);
		std::vector<float> blob(network.numberOfWeights());
		std::mt19937 rng(1234);
		std::normal_distribution<float> normal(0.0f, 1.0f);
		for (float &w : blob)
			w = 0.05f * normal(rng); // plain synthetic weights: this driver measures throughput, tests use the documented He-init blob
		network.loadWeights(blob);

		agx::SelfplayConfig selfplay;
		selfplay.games_per_thread = games;
		selfplay.max_simulations = sims;
		selfplay.search_config.max_batch_size = batch;
		agx::GeneratorPool pool(game, selfplay);
		std::vector<uint16_t> openings(static_cast<size_t>(3 * games) * AGX_OPENING_CAP);
		for (int i = 0; i < 3 * games; i++)
			agx::check(agx_make_opening(rules, game.rows, static_cast<uint32_t>(i), openings.data() + static_cast<size_t>(i) * AGX_OPENING_CAP));
		pool.begin(openings);
		for (int i = 0; i < warmup; i++)
			pool.generate(network);
		agx::check(agx_device_synchronize());
		const AgxEngineStats s0 = pool.getStats();
		const auto t0 = std::chrono::steady_clock::now();
		for (int i = 0; i < steps; i++)
			pool.generate(network);
		agx::check(agx_device_synchronize());
		const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		const AgxEngineStats s1 = pool.getStats();
		if (s1.first_error != 0)
			throw std::runtime_error("engine stopped with error " + std::to_string(s1.first_error));
		std::printf("{\"simulations_per_sec\": %.1f, \"ms_per_step\": %.3f, \"moves_per_sec\": %.1f, \"games_finished\": %d, \"network_evaluations\": %llu}\n",
				(s1.evaluated_nodes - s0.evaluated_nodes) / seconds, 1e3 * seconds / steps, (s1.moves_played - s0.moves_played) / seconds,
				s1.games_finished - s0.games_finished, s1.network_evaluations - s0.network_evaluations);
	}
	catch (const std::exception &e)
	{
		std::fprintf(stderr, "agx_selfplay: %s\n", e.what());
		return 1;
	}
	return 0;
}
