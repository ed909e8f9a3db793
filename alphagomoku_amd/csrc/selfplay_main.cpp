/*
 * selfplay_main.cpp — native self-play driver: the C++ host loop over the C ABI, no Python involved.
 * Stands where GeneratorThread::run stands in the reference (src/selfplay/GeneratorManager.cpp:124-141).
 *
 *   agx_selfplay [--games 1024] [--steps 200] [--warmup 20] [--sims 400] [--batch 8] [--blocks 6] [--filters 128] [--rules 0]
 *                [--balanced-openings 0|1] [--drain-every N] [--pvq 0|1] [--symmetries 0|1] [--match 0|1] [--slices N (default 4: the pool
 *                 as N slices of the chip on CU-masked streams, 1 = one lock-step pool)]
 *                [--devices 0,1,...] [--save-buffer file]
 * --devices: one generator thread per listed device, each with its own pool of --games games (GeneratorManager.cpp:146-152, 39-53); the
 * threads share one game buffer (--save-buffer: finished games in dataset format 201, samples quantised on the device).
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
#include <algorithm>
#include <functional>
#include <string>
#include <thread>

struct Options
{
		int games = 1024, steps = 200, warmup = 20, sims = 400, batch = 8, blocks = 6, filters = 128, rules = 0;
		int balanced = 0, drain_every = 0, pvq = 0, symmetries = 0, match = 0, slices = 4;
		int host_steps_ahead = 2; // agx::HostPacer: steps the launch loop may run ahead of a slice's stream (0 = unbounded: it then spins on a full launch queue)
};
struct DeviceResult
{
		double seconds = 0.0;
		unsigned long long simulations = 0, moves = 0, evaluations = 0, samples = 0, refills = 0;
		int games_finished = 0, openings_taken = 0, slices = 1;
		int first_score[3] = { 0, 0, 0 };
		std::string error;
};

/* what one GeneratorThread does (src/selfplay/GeneratorManager.cpp:124-141) on its device: own network copy, own game pool, finished games
 * handed to the SHARED buffer (GeneratorManager::addToBuffer, :160-164 — the buffer locks its own mutex) */
static void run_device(int device, int thread_index, const Options &o, AgxGameBuffer *buffer, DeviceResult &result)
{
	try
	{
		agx::check(agx_set_device(device));
		agx::GameConfig game;
		game.rules = o.rules;
		agx::AGNetwork network(game, o.blocks, o.filters, o.pvq ? "ResnetPVQ" : "ResnetPV");
		std::vector<float> blob(network.numberOfWeights());
		std::mt19937 rng(1234);
		std::normal_distribution<float> normal(0.0f, 1.0f);
		for (float &w : blob)
			w = 0.05f * normal(rng); // plain synthetic weights: this driver measures throughput, tests use the documented He-init blob
		network.loadWeights(blob);
		agx::AGNetwork second_network(game, o.blocks, o.filters, o.pvq ? "ResnetPVQ" : "ResnetPV"); // the second player of a match
		if (o.match)
		{
			for (float &w : blob)
				w = 0.05f * normal(rng);
			second_network.loadWeights(blob);
		}

		agx::SelfplayConfig selfplay;
		selfplay.games_per_thread = o.games;
		selfplay.max_simulations = o.sims;
		selfplay.search_config.max_batch_size = o.batch;
		selfplay.use_symmetries = (o.symmetries != 0);
		selfplay.network_outputs = network.getOutputConfig();
		selfplay.record_format = (buffer != nullptr) ? 2 : 1; // format-201 samples when the games go to a buffer
		agx::GeneratorPool pool(game, selfplay, o.match != 0);
		pool.setHostStepsAhead(o.host_steps_ahead);
		result.slices = o.match ? 1 : pool.useChipSlices(network, o.slices, thread_index); // the pool as slices of the chip (agx.hpp); masked streams of its own per device thread
		auto one_step = [&]() { if (o.match) pool.generate(network, second_network); else pool.generate(network); };
		uint32_t next_seed = static_cast<uint32_t>(thread_index) * 1000003u; // disjoint openings per device thread
		bool first_batch = true;
		auto make_openings = [&](int count)
		{
			if (o.balanced && first_batch)
			{ // the generator borrows the pool's task slots, so it can only run before begin(): the first batch is balanced
				first_batch = false;
				return pool.generateOpenings(network, count, 12345u + next_seed);
			}
			first_batch = false;
			std::vector<uint16_t> out(static_cast<size_t>(count) * AGX_OPENING_CAP);
			for (int i = 0; i < count; i++)
				agx::check(agx_make_opening(o.rules, game.rows, next_seed++, out.data() + static_cast<size_t>(i) * AGX_OPENING_CAP));
			return out;
		};
		int n_openings = (o.drain_every > 0) ? o.games + o.games / 2 : 3 * o.games;
		pool.begin(make_openings(n_openings));
		for (int i = 0; i < o.warmup; i++)
			one_step();
		agx::check(agx_device_synchronize());
		const AgxEngineStats s0 = pool.getStats();
		const auto t0 = std::chrono::steady_clock::now();
		std::vector<AgxMoveRecord> records;
		std::vector<AgxEdgeView> record_edges;
		for (int i = 0; i < o.steps; i++)
		{
			one_step();
			if (o.drain_every > 0 && (i + 1) % o.drain_every == 0)
			{ // hand the finished samples over and keep the opening list ahead of the games (GeneratorManager.cpp:160-164)
				if (buffer != nullptr)
					agx::check(agx_game_buffer_collect(buffer, pool.handle(), nullptr));
				else
				{
					pool.drainRecords(records, record_edges);
					result.samples += records.size();
				}
				const AgxEngineStats st = pool.getStats();
				if (st.openings_taken + o.games > n_openings) // (slot s plays openings s, s + games, ...: a whole round ahead of the furthest slot)
				{
					pool.addOpenings(make_openings(o.games));
					n_openings += o.games;
					result.refills++;
				}
			}
		}
		agx::check(agx_device_synchronize());
		result.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		const AgxEngineStats s1 = pool.getStats();
		if (s1.first_error != 0)
			throw std::runtime_error("engine stopped with error " + std::to_string(s1.first_error));
		if (buffer != nullptr)
			agx::check(agx_game_buffer_collect(buffer, pool.handle(), nullptr));
		if (o.match)
		{
			const std::vector<int> results = pool.getMatchResults();
			for (int p = 0; p < o.games; p++)
				for (int k = 0; k < 3; k++)
					result.first_score[k] += results[4 * p + k];
		}
		result.simulations = s1.evaluated_nodes - s0.evaluated_nodes;
		result.moves = s1.moves_played - s0.moves_played;
		result.evaluations = s1.network_evaluations - s0.network_evaluations;
		result.games_finished = s1.games_finished - s0.games_finished;
		result.openings_taken = s1.openings_taken;
	}
	catch (const std::exception &e)
	{
		result.error = e.what();
	}
}

int main(int argc, char **argv)
{
	Options o;
	std::vector<int> devices = { 0 };
	std::string buffer_path;
	for (int i = 1; i + 1 < argc; i += 2)
	{
		const std::string k = argv[i];
		const int v = std::atoi(argv[i + 1]);
		if (k == "--games") o.games = v;
		else if (k == "--steps") o.steps = v;
		else if (k == "--warmup") o.warmup = v;
		else if (k == "--sims") o.sims = v;
		else if (k == "--batch") o.batch = v;
		else if (k == "--blocks") o.blocks = v;
		else if (k == "--filters") o.filters = v;
		else if (k == "--rules") o.rules = v;
		else if (k == "--device") devices = { v };
		else if (k == "--devices")
		{ // one generator thread per listed device (a device may be listed twice: two threads share it)
			devices.clear();
			std::string list = argv[i + 1];
			for (size_t pos = 0; pos <= list.size();)
			{
				const size_t comma = std::min(list.find(',', pos), list.size());
				devices.push_back(std::atoi(list.substr(pos, comma - pos).c_str()));
				pos = comma + 1;
			}
		}
		else if (k == "--balanced-openings") o.balanced = v;
		else if (k == "--drain-every") o.drain_every = v;
		else if (k == "--slices") o.slices = v;
		else if (k == "--host-steps-ahead") o.host_steps_ahead = v;
		else if (k == "--pvq") o.pvq = v;
		else if (k == "--symmetries") o.symmetries = v;
		else if (k == "--match") o.match = v;
		else if (k == "--save-buffer") buffer_path = argv[i + 1];
		else
		{
			std::fprintf(stderr, "unknown option %s\n", argv[i]);
			return 2;
		}
	}
	AgxGameBuffer *buffer = nullptr;
	if (!buffer_path.empty() && !o.match)
	{
		if (agx_game_buffer_create(o.rules, 15, 15, 225, &buffer) != AGX_OK)
		{
			std::fprintf(stderr, "agx_selfplay: %s\n", agx_last_error());
			return 1;
		}
		if (o.drain_every <= 0)
			o.drain_every = 128;
	}
	// GeneratorManager::generate (GeneratorManager.cpp:182-196): one thread per device, joined when all are done
	std::vector<DeviceResult> results(devices.size());
	std::vector<std::thread> threads;
	for (size_t i = 0; i < devices.size(); i++)
		threads.emplace_back(run_device, devices[i], static_cast<int>(i), std::cref(o), buffer, std::ref(results[i]));
	for (std::thread &t : threads)
		t.join();
	DeviceResult total;
	for (const DeviceResult &r : results)
	{
		if (!r.error.empty())
		{
			std::fprintf(stderr, "agx_selfplay: %s\n", r.error.c_str());
			return 1;
		}
		total.seconds = std::max(total.seconds, r.seconds);
		total.simulations += r.simulations;
		total.moves += r.moves;
		total.evaluations += r.evaluations;
		total.samples += r.samples;
		total.refills += r.refills;
		total.games_finished += r.games_finished;
		total.openings_taken += r.openings_taken;
		for (int k = 0; k < 3; k++)
			total.first_score[k] += r.first_score[k];
	}
	AgxGameBufferStats bs { };
	if (buffer != nullptr)
	{
		agx_game_buffer_stats(buffer, &bs);
		if (agx_game_buffer_save(buffer, buffer_path.c_str(), 1) != AGX_OK)
		{
			std::fprintf(stderr, "agx_selfplay: %s\n", agx_last_error());
			return 1;
		}
		total.samples = static_cast<unsigned long long>(bs.samples);
		agx_game_buffer_destroy(buffer);
	}
	std::printf("{\"devices\": %zu, \"simulations_per_sec\": %.1f, \"ms_per_step\": %.3f, \"moves_per_sec\": %.1f, \"games_finished\": %d, \"network_evaluations\": %llu, "
			"\"samples_drained\": %llu, \"opening_refills\": %llu, \"openings_taken\": %d, \"first_player_won_drawn_lost\": [%d, %d, %d], \"buffer_games\": %d, "
			"\"slices\": %d}\n",
			devices.size(), total.simulations / total.seconds, 1e3 * total.seconds / o.steps, total.moves / total.seconds, total.games_finished, total.evaluations,
			total.samples, total.refills, total.openings_taken, total.first_score[0], total.first_score[1], total.first_score[2], bs.games, results.empty() ? 1 : results[0].slices);
	return 0;
}
