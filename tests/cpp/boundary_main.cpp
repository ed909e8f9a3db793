/*
 * boundary_main.cpp — drives the reference-named C++ boundary (include/alphagomoku_agx/) the way the reference's own callers do.
 *
 *   generate  : the call chain of training_launcher/launcher.cpp:63-72 -> TrainingManager::generateGames (src/selfplay/TrainingManager.cpp:194-214):
 *               GeneratorManager(gameConfig, selfplayConfig) -> setWorkingDirectory -> loadState -> generate(NetworkLoader, games) ->
 *               saveState -> getGameBuffer().save(...), with one GeneratorThread per entry of selfplayConfig.device_config
 *   evaluator : NNEvaluator with host-side SearchTasks (addToQueue(task, symmetry), evaluateGraph, asyncEvaluateGraphLaunch / Join)
 *   errors    : the exceptions the reference throws at this boundary
 * Prints one JSON line per mode; tests/test_boundary_gpu.py runs it on the GPU box and checks the results.
 */
#include "../../include/alphagomoku_agx/selfplay.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>

using namespace ag;

static std::map<std::string, std::string> parse(int argc, char **argv)
{
	std::map<std::string, std::string> out;
	for (int i = 2; i + 1 < argc; i += 2)
		out[argv[i]] = argv[i + 1];
	return out;
}
static int geti(const std::map<std::string, std::string> &m, const char *k, int d)
{
	auto it = m.find(k);
	return (it == m.end()) ? d : std::atoi(it->second.c_str());
}

static int mode_generate(const std::map<std::string, std::string> &a)
{
	const int n = geti(a, "--board", 15);
	GameConfig game_config(static_cast<GameRules>(geti(a, "--rules", 0)), n);
	SelfplayConfig selfplay_config;
	selfplay_config.use_opening = geti(a, "--use-opening", 1) != 0;
	selfplay_config.use_symmetries = geti(a, "--symmetries", 1) != 0;
	selfplay_config.games_per_thread = geti(a, "--games-per-thread", 64);
	selfplay_config.constraints = Constraints::simulations(geti(a, "--sims", 100));
	selfplay_config.final_selector.policy = "best";
	selfplay_config.search_config.max_batch_size = geti(a, "--batch", 8);
	selfplay_config.search_config.tss_config.hash_table_size = geti(a, "--table-entries", 1 << 16);
	selfplay_config.search_config.tree_config.node_bucket_size = 4096;
	selfplay_config.search_config.tree_config.edge_bucket_size = 131072;
	selfplay_config.device_config.clear();
	std::stringstream devices(a.count("--devices") ? a.at("--devices") : std::string("0"));
	std::string item;
	while (std::getline(devices, item, ','))
	{
		DeviceConfig d;
		d.device = Device::hip(std::atoi(item.c_str()));
		d.batch_size = geti(a, "--nn-batch", 1 << 20);
		selfplay_config.device_config.push_back(d);
	}
	const std::string out = a.count("--out") ? a.at("--out") : std::string(".");

	GeneratorManager manager(game_config, selfplay_config);
	manager.setWorkingDirectory(out);
	manager.loadState();
	manager.generate(NetworkLoader(a.at("--network")), geti(a, "--games", 32));
	manager.saveState(true);
	manager.getGameBuffer().save(out + "/buffer_0.bin");
	manager.printStats();

	const GameDataBuffer &buffer = manager.getGameBuffer();
	const GameDataBufferStats st = buffer.getStats();
	std::ofstream raw(out + "/games.raw", std::ofstream::binary);
	for (int i = 0; i < buffer.numberOfGames(); i++)
	{
		const std::vector<uint8_t> g = buffer.getGameData(i);
		const uint32_t size = static_cast<uint32_t>(g.size());
		raw.write(reinterpret_cast<const char*>(&size), 4);
		raw.write(reinterpret_cast<const char*>(g.data()), g.size());
	}
	std::printf("{\"mode\": \"generate\", \"threads\": %zu, \"games\": %d, \"samples\": %d, \"cross_win\": %d, \"draws\": %d, \"circle_win\": %d, \"game_length\": %d}\n",
			selfplay_config.device_config.size(), st.games, st.samples, st.cross_win, st.draws, st.circle_win, st.game_length);
	return 0;
}

static int mode_evaluator(const std::map<std::string, std::string> &a)
{
	DeviceConfig device;
	device.device = Device::hip(0);
	device.batch_size = 4; // smaller than the queue: evaluateGraph has to run several launches
	NNEvaluator evaluator(device);
	evaluator.loadGraph(NetworkLoader(a.at("--network")));
	const GameConfig cfg = evaluator.get_network().getGameConfig();
	const int hw = cfg.rows * cfg.cols;
	std::ifstream in(a.at("--features"), std::ifstream::binary);
	std::vector<uint32_t> features(hw);
	in.read(reinterpret_cast<char*>(features.data()), sizeof(uint32_t) * hw);
	std::vector<SearchTask> tasks(16, SearchTask(cfg.rows, cfg.cols));
	for (int s = 0; s < 16; s++)
	{
		tasks[s].getFeatures() = features;
		evaluator.addToQueue(tasks[s], s % 8);
	}
	const bool full = evaluator.isQueueFull();
	const int queued = evaluator.getQueueSize();
	for (int s = 0; s < 8; s++) // the synchronous path ...
		;
	evaluator.evaluateGraph();
	// ... and the launch / join pair on a second set
	std::vector<SearchTask> later(3, SearchTask(cfg.rows, cfg.cols));
	for (int s = 0; s < 3; s++)
	{
		later[s].getFeatures() = features;
		evaluator.addToQueue(later[s], 5);
	}
	evaluator.asyncEvaluateGraphLaunch();
	bool second_launch_refused = false;
	try
	{
		evaluator.asyncEvaluateGraphLaunch();
	} catch (std::logic_error&)
	{
		second_launch_refused = true; // "some tasks are already being processed" (NNEvaluator.cpp:186-187)
	}
	evaluator.asyncEvaluateGraphJoin();
	std::ofstream out(a.at("--out"), std::ofstream::binary);
	auto dump = [&](const SearchTask &t)
	{
		out.write(reinterpret_cast<const char*>(t.getPolicy().data()), sizeof(float) * hw);
		const Value v = t.getValue();
		out.write(reinterpret_cast<const char*>(&v.win_rate), 4);
		out.write(reinterpret_cast<const char*>(&v.draw_rate), 4);
	};
	for (const SearchTask &t : tasks)
		dump(t);
	for (const SearchTask &t : later)
		dump(t);
	bool processed = true;
	for (const SearchTask &t : tasks)
		processed &= t.wasProcessedByNetwork();
	std::printf("{\"mode\": \"evaluator\", \"queue_full\": %d, \"queued\": %d, \"processed\": %d, \"second_launch_refused\": %d, \"samples\": %llu, \"outputs\": \"%s\"}\n",
			full ? 1 : 0, queued, processed ? 1 : 0, second_launch_refused ? 1 : 0, static_cast<unsigned long long>(evaluator.getStats().batch_sizes),
			evaluator.get_network().getOutputConfig().c_str());
	return 0;
}

static int mode_errors(const std::map<std::string, std::string> &a)
{
	int caught = 0;
	try
	{ // the device engine has no CPU path
		DeviceConfig d;
		d.device = Device::cpu();
		NNEvaluator e(d);
	} catch (std::logic_error&)
	{
		caught |= 1;
	}
	try
	{ // "NNEvaluator::get_network() : network has not been initialized" (NNEvaluator.cpp:233-236)
		NNEvaluator e( (DeviceConfig()));
		e.evaluateGraph();
	} catch (std::logic_error&)
	{
		caught |= 2;
	}
	try
	{ // FileLoader: "File ... does not exist" (file_util.cpp:58-60)
		NetworkLoader("/nonexistent/network.bin").get();
	} catch (std::runtime_error&)
	{
		caught |= 4;
	}
	try
	{ // EdgeSelector::create: unknown final selector (EdgeSelector.cpp:680-711)
		GameConfig g(GameRules::FREESTYLE, 15);
		SearchConfig s;
		EdgeSelectorConfig f;
		f.policy = "no_such_policy";
		GamePool pool(g, s, f, 4, 100, false, "pv");
	} catch (std::logic_error&)
	{
		caught |= 8;
	}
	try
	{ // expanding before the evaluator has joined the slice's launch
		GameConfig g(GameRules::FREESTYLE, 15);
		SearchConfig s;
		s.tss_config.hash_table_size = 4096;
		s.tree_config.node_bucket_size = 256;
		s.tree_config.edge_bucket_size = 8192;
		EdgeSelectorConfig f;
		f.policy = "best";
		GamePool pool(g, s, f, 4, 100, false, "pv");
		pool.begin(std::vector<uint16_t>(4 * AGX_OPENING_CAP, 0));
		Tree tree(pool, 0, 1, nullptr);
		Search search(pool, 0, 1, nullptr);
		NNEvaluator evaluator( (DeviceConfig()));
		search.select(tree);
		search.solve();
		search.scheduleToNN(evaluator);
		if (search.areTasksReady())
			throw std::runtime_error("tasks cannot be ready before the evaluator ran");
		search.expand(tree);
	} catch (std::logic_error&)
	{
		caught |= 16;
	}
	(void) a;
	std::printf("{\"mode\": \"errors\", \"caught\": %d}\n", caught);
	return 0;
}

int main(int argc, char **argv)
{
	if (argc < 2)
	{
		std::fprintf(stderr, "usage: agx_boundary_test generate|evaluator|errors [--key value ...]\n");
		return 2;
	}
	try
	{
		const std::map<std::string, std::string> args = parse(argc, argv);
		const std::string mode = argv[1];
		if (mode == "generate")
			return mode_generate(args);
		if (mode == "evaluator")
			return mode_evaluator(args);
		if (mode == "errors")
			return mode_errors(args);
		std::fprintf(stderr, "unknown mode %s\n", mode.c_str());
		return 2;
	} catch (const std::exception &e)
	{
		std::fprintf(stderr, "agx_boundary_test: %s\n", e.what());
		return 1;
	}
}
