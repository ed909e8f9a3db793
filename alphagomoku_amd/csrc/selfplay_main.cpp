/*
 * selfplay_main.cpp — native self-play driver: the C++ host loop over the C ABI, no Python involved.
 * Stands where GeneratorThread::run stands in the reference (src/selfplay/GeneratorManager.cpp:124-141).
 *
 *   agx_selfplay [--games 1024] [--steps 200] [--warmup 20] [--sims 400] [--batch 8] [--blocks 6] [--filters 128] [--rules 0]
 *                [--balanced-openings 0|1] [--drain-every N] [--pvq 0|1] [--symmetries 0|1] [--match 0|1]
 * prints one JSON line with simulations/s.  With --balanced-openings the openings come from the device OpeningGenerator
 * (agx_engine_generate_openings); every --drain-every steps the samples are handed over (drainRecords) and, when the pool runs
 * low, more openings are appended (addOpenings) — the loop a GeneratorThread runs for hours.  With --match 1 the pool plays
 * evaluation matches instead (EvaluatorThread::run, evaluation/EvaluationGame.cpp): --games pairs of players, two networks of
 * different weights, every opening twice with the colours swapped; the line then also carries the first player's score.  Weights are synthetic (He-normal, fixed seed) — there are no checkpoints offline.
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
	int balanced = 0, drain_every = 0, pvq = 0, symmetries = 0, match = 0;
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
		else if (k == "--balanced-openings") balanced = v;
		else if (k == "--drain-every") drain_every = v;
		else if (k == "--pvq") pvq = v;
		else if (k == "--symmetries") symmetries = v;
		else if (k == "--match") match = v;
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
		agx::AGNetwork network(game, blocks, filters, pvq ? "ResnetPVQ" : "ResnetPV");
		std::vector<float> blob(network.numberOfWeights());
		std::mt19937 rng(1234);
		std::normal_distribution<float> normal(0.0f, 1.0f);
		for (float &w : blob)
			w = 0.05f * normal(rng); // plain synthetic weights: this driver measures throughput, tests use the documented He-init blob
		network.loadWeights(blob);
		agx::AGNetwork second_network(game, blocks, filters, pvq ? "ResnetPVQ" : "ResnetPV"); // the second player of a match
		if (match)
		{
			for (float &w : blob)
				w = 0.05f * normal(rng);
			second_network.loadWeights(blob);
		}

		agx::SelfplayConfig selfplay;
		selfplay.games_per_thread = games;
		selfplay.max_simulations = sims;
		selfplay.search_config.max_batch_size = batch;
		selfplay.use_symmetries = (symmetries != 0);
		selfplay.network_outputs = network.getOutputConfig();
		agx::GeneratorPool pool(game, selfplay, match != 0);
		auto one_step = [&]() { if (match) pool.generate(network, second_network); else pool.generate(network); };
		uint32_t next_seed = 0;
		auto make_openings = [&](int count)
		{
			if (balanced && next_seed == 0)
			{ // the generator borrows the pool's task slots, so it can only run before begin(): the first batch is balanced
				next_seed += 1000000u;
				return pool.generateOpenings(network, count, 12345u);
			}
			std::vector<uint16_t> out(static_cast<size_t>(count) * AGX_OPENING_CAP);
			for (int i = 0; i < count; i++)
				agx::check(agx_make_opening(rules, game.rows, next_seed++, out.data() + static_cast<size_t>(i) * AGX_OPENING_CAP));
			return out;
		};
		int n_openings = (drain_every > 0) ? games + games / 2 : 3 * games;
		pool.begin(make_openings(n_openings));
		for (int i = 0; i < warmup; i++)
			one_step();
		agx::check(agx_device_synchronize());
		const AgxEngineStats s0 = pool.getStats();
		const auto t0 = std::chrono::steady_clock::now();
		unsigned long long samples = 0, refills = 0;
		std::vector<AgxMoveRecord> records;
		std::vector<AgxEdgeView> record_edges;
		for (int i = 0; i < steps; i++)
		{
			one_step();
			if (drain_every > 0 && (i + 1) % drain_every == 0)
			{ // hand the finished samples over and keep the opening list ahead of the games (GeneratorManager.cpp:160-164)
				pool.drainRecords(records, record_edges);
				samples += records.size();
				const AgxEngineStats st = pool.getStats();
				if (st.openings_taken + games / 2 > n_openings)
				{
					pool.addOpenings(make_openings(games));
					n_openings += games;
					refills++;
				}
			}
		}
		agx::check(agx_device_synchronize());
		const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		const AgxEngineStats s1 = pool.getStats();
		if (s1.first_error != 0)
			throw std::runtime_error("engine stopped with error " + std::to_string(s1.first_error));
		int first_score[3] = { 0, 0, 0 };
		if (match)
		{
			const std::vector<int> results = pool.getMatchResults();
			for (int p = 0; p < games; p++)
				for (int k = 0; k < 3; k++)
					first_score[k] += results[4 * p + k];
		}
		std::printf("{\"simulations_per_sec\": %.1f, \"ms_per_step\": %.3f, \"moves_per_sec\": %.1f, \"games_finished\": %d, \"network_evaluations\": %llu, "
				"\"samples_drained\": %llu, \"opening_refills\": %llu, \"openings_taken\": %d, \"first_player_won_drawn_lost\": [%d, %d, %d]}\n",
				(s1.evaluated_nodes - s0.evaluated_nodes) / seconds, 1e3 * seconds / steps, (s1.moves_played - s0.moves_played) / seconds,
				s1.games_finished - s0.games_finished, s1.network_evaluations - s0.network_evaluations, samples, refills, s1.openings_taken, first_score[0],
				first_score[1], first_score[2]);
	}
	catch (const std::exception &e)
	{
		std::fprintf(stderr, "agx_selfplay: %s\n", e.what());
		return 1;
	}
	return 0;
}
