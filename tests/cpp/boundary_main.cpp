/*
 * boundary_main.cpp — drives the reference-named C++ boundary (include/alphagomoku_agx/) the way the reference's own callers do.
 *
 *   generate  : the call chain of training_launcher/launcher.cpp:63-72 -> TrainingManager::generateGames (src/selfplay/TrainingManager.cpp:194-214):
 *               GeneratorManager(gameConfig, selfplayConfig) -> setWorkingDirectory -> loadState -> generate(NetworkLoader, games) ->
 *               saveState -> getGameBuffer().save(...), with one GeneratorThread per entry of selfplayConfig.device_config
 *   evaluator : NNEvaluator with host-side SearchTasks (addToQueue(task, symmetry), evaluateGraph, asyncEvaluateGraphLaunch / Join)
 *   player    : two evaluation Players (evaluation/Player.cpp:64-129,205-212), each with a Tree(const TreeConfig&) and a
 *               Search(const GameConfig&, const SearchConfig&) of its own, played against each other the way EvaluationGame::generate
 *               (evaluation/EvaluationGame.cpp:77-143) does it; prints the moves
 *   generator : GameGenerator's own four-argument constructor (selfplay/GameGenerator.hpp:54): one game, its own tree and search
 *   errors    : the exceptions the reference throws at this boundary
 * Prints one JSON line per mode; tests/test_boundary_gpu.py runs it on the GPU box and checks the results.
 */
#include "../../include/alphagomoku_agx/selfplay.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <cmath>
#include <map>
#include <mutex>
#include <sstream>
#include <thread>
#include <filesystem>

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

	// TrainingManager installs the custom SIGINT handler before anything runs (TrainingManager.cpp: setupSignalHandler(SignalType::INT, ...));
	// generate() then stops its threads on a captured signal and the caller saves the state
	const bool interruptible = geti(a, "--interruptible", 0) != 0;
	if (interruptible)
		setupSignalHandler(SignalType::INT, SignalHandlerMode::CUSTOM_HANDLER);
	GeneratorManager manager(game_config, selfplay_config);
	manager.setStatsPeriod(geti(a, "--stats-period", 60));
	manager.setWorkingDirectory(out);
	manager.loadState();
	if (interruptible)
		std::cout << "generating" << std::endl; // (the test sends its SIGINT some seconds after this line)
	// one generate() per training iteration in the reference (TrainingManager::runIterationRL): every call sets its generator threads up again,
	// and must get the SAME CU-masked streams back (they cannot be destroyed: agx.h)
	const int iterations = geti(a, "--iterations", 1);
	for (int it = 0; it < iterations; it++)
		manager.generate(NetworkLoader(a.at("--network")), geti(a, "--games", 32) * (it + 1) / iterations);
	const bool was_interrupted = hasCapturedSignal(SignalType::INT);
	manager.saveState(interruptible ? was_interrupted : true); // TrainingManager.cpp:205-207: saveState(was_interrupted)
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
	std::printf("{\"mode\": \"generate\", \"threads\": %zu, \"games\": %d, \"samples\": %d, \"cross_win\": %d, \"draws\": %d, \"circle_win\": %d, \"game_length\": %d, "
			"\"iterations\": %d, \"masked_streams\": %d, \"interrupted\": %s}\n",
			selfplay_config.device_config.size(), st.games, st.samples, st.cross_win, st.draws, st.circle_win, st.game_length, iterations, agx_stream_masked_count(),
			was_interrupted ? "true" : "false");
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

/* evaluation/Player.cpp:64-129,152-160,205-212 call for call (simulation-count constraint; the time-controlled variant reads wall clocks) */
class Player
{
		NNEvaluator &nn_evaluator;
		GameConfig game_config;
		EdgeSelectorConfig final_move_selection_config;
		Tree tree;
		Search search;
		Constraints constraints;
		Sign sign = Sign::NONE;
	public:
		Player(const GameConfig &gameOptions, const SelfplayConfig &options, NNEvaluator &evaluator) :
				nn_evaluator(evaluator), game_config(gameOptions), final_move_selection_config(options.final_selector), tree(options.search_config.tree_config),
				search(gameOptions, options.search_config), constraints(options.constraints)
		{
			search.setBatchSize(options.search_config.max_batch_size);
		}
		void setSign(Sign s) noexcept { sign = s; }
		Sign getSign() const noexcept { return sign; }
		AlphaBetaSearch& getSolver() noexcept { return search.getSolver(); }
		void setBoard(const matrix<Sign> &board, Sign signToMove)
		{ // Player.cpp:98-110
			search.cleanup(tree);
			tree.setBoard(board, signToMove);
			search.setBoard(board, signToMove);

			const MCTSConfig &mcts_config = search.getConfig().mcts_config;
			std::unique_ptr<EdgeSelector> tmp = EdgeSelector::create(mcts_config.edge_selector_config);
			tree.setEdgeSelector(*tmp);
			tree.setEdgeGenerator(UnifiedGenerator(mcts_config.max_children, mcts_config.policy_expansion_threshold, mcts_config.policy_temperature));
		}
		void selectSolveEvaluate()
		{ // Player.cpp:111-122
			search.select(tree, constraints.max_simulations);
			search.solve();
			search.scheduleToNN(nn_evaluator);
		}
		void expandBackup()
		{ // Player.cpp:123-131
			search.generateEdges(tree);
			search.expand(tree);
			search.backup(tree);
		}
		bool isSearchOver()
		{ // Player.cpp:152-160 + get_simulations_for_move (utils/misc.cpp:171-179)
			if (tree.isRootProven())
				return true;
			const Value root_eval = tree.getInfo( { }).getValue();
			const float reduction = std::max(0.0f, std::min(1.0f, (root_eval.draw_rate - 0.75f) / (1.0f - 0.75f)));
			const int sims = static_cast<int>(constraints.max_simulations - reduction * (constraints.max_simulations - 50));
			return tree.getSimulationCount() > sims;
		}
		Move getMove()
		{ // Player.cpp:205-212
			search.cleanup(tree);
			std::unique_ptr<EdgeSelector> selector = EdgeSelector::create(final_move_selection_config);
			const Node root_node = tree.getInfo( { });
			return selector->select(&root_node)->getMove();
		}
};

static int mode_player(const std::map<std::string, std::string> &a)
{
	const int n = geti(a, "--board", 15);
	GameConfig game_config(static_cast<GameRules>(geti(a, "--rules", 0)), n);
	SelfplayConfig options;
	options.constraints = Constraints::simulations(geti(a, "--sims", 100));
	options.final_selector.policy = "best";
	options.search_config.max_batch_size = geti(a, "--batch", 8);
	options.search_config.tss_config.hash_table_size = geti(a, "--table-entries", 1 << 16);
	options.search_config.tree_config.node_bucket_size = 4096;
	options.search_config.tree_config.edge_bucket_size = 65536;
	DeviceConfig device;
	device.batch_size = 64;
	NNEvaluator first_evaluator(device), second_evaluator(device);
	first_evaluator.loadGraph(NetworkLoader(a.at("--network")));
	second_evaluator.loadGraph(NetworkLoader(a.count("--network2") ? a.at("--network2") : a.at("--network")));
	first_evaluator.useSymmetries(false);
	second_evaluator.useSymmetries(false);
	Player first(game_config, options, first_evaluator), second(game_config, options, second_evaluator);
	first.setSign(Sign::CROSS);
	second.setSign(Sign::CIRCLE);

	// EvaluationGame::generate (EvaluationGame.cpp:77-143) for one game: opening, then the player to move searches until its search is over
	matrix<Sign> board(n, n);
	board.fill(Sign::NONE);
	std::vector<uint16_t> opening(AGX_OPENING_CAP, 0);
	if (agx_make_opening(static_cast<int>(game_config.rules), n, static_cast<uint32_t>(geti(a, "--opening-seed", 1)), opening.data()) != AGX_OK)
		throw std::runtime_error(agx_last_error());
	Sign sign_to_move = Sign::CROSS;
	std::vector<uint8_t> cells(static_cast<size_t>(n) * n, 0);
	for (int i = 0; i < opening[0]; i++)
	{
		const Move m(opening[1 + i]);
		board.at(m.row, m.col) = m.sign;
		cells[m.row * n + m.col] = static_cast<uint8_t>(m.sign);
		sign_to_move = (m.sign == Sign::CROSS) ? Sign::CIRCLE : Sign::CROSS;
	}
	first.getSolver().clear();
	second.getSolver().clear();
	std::vector<Move> played;
	int outcome = 0, steps = 0;
	auto current = [&]() -> Player& { return (sign_to_move == first.getSign()) ? first : second; };
	auto evaluator_of = [&](Player &p) -> NNEvaluator& { return (&p == &first) ? first_evaluator : second_evaluator; };
	current().setBoard(board, sign_to_move);
	const int max_plies = geti(a, "--plies", n * n);
	while (outcome == 0 && static_cast<int>(played.size()) < max_plies)
	{
		Player &p = current();
		p.selectSolveEvaluate();
		evaluator_of(p).evaluateGraph();
		p.expandBackup();
		steps++;
		if (p.isSearchOver())
		{
			const Move m = p.getMove();
			played.push_back(m);
			board.at(m.row, m.col) = m.sign;
			cells[m.row * n + m.col] = static_cast<uint8_t>(m.sign);
			sign_to_move = (m.sign == Sign::CROSS) ? Sign::CIRCLE : Sign::CROSS;
			if (agx_get_outcome(static_cast<int>(game_config.rules), n, cells.data(), static_cast<int>(m.sign), m.row, m.col, game_config.draw_after, &outcome) != AGX_OK)
				throw std::runtime_error(agx_last_error());
			if (outcome == 0)
				current().setBoard(board, sign_to_move);
		}
	}
	std::printf("{\"mode\": \"player\", \"opening_stones\": %d, \"outcome\": %d, \"steps\": %d, \"moves\": [", static_cast<int>(opening[0]), outcome, steps);
	for (size_t i = 0; i < played.size(); i++)
		std::printf("%s%d", i ? ", " : "", static_cast<int>(played[i].toShort()));
	std::printf("]}\n");
	return 0;
}

/* player/SearchThread.cpp:84-199 as written: the lock scopes on the tree's PriorityMutex (Tree::low_priority_lock), get_batch_size (the square
 * root of the simulation count, capped) into Search::setBatchSize, serial_run, asynchronous_run with useBuffer / switchBuffer,
 * asyncEvaluateGraphLaunch / Join and the launch's estimated end time as the deadline of Search::solve, is_running under search_mutex, the stop
 * condition on the simulation and node counts.  Two switches exist for the replayable parity run only (clock-independent results):
 * --fixed-batch 1 and --solve-deadline 0. */
class SearchThread
{
		Tree &tree;
		Search search;
		int max_simulations, max_nodes;
		int iterations = 0;
		mutable std::mutex search_mutex; // SearchThread.hpp: guards is_running
		bool is_running = true;
	public:
		bool fixed_batch = false;    // --fixed-batch 1: max_batch_size in every iteration (the replayable procedure of the parity test)
		bool solve_deadline = true;  // --solve-deadline 0: Search::solve without the end time of the network launch
		SearchThread(const GameConfig &gameOptions, const SearchConfig &searchOptions, Tree &tree, int maxSimulations, int maxNodes) :
				tree(tree), search(gameOptions, searchOptions), max_simulations(maxSimulations), max_nodes(maxNodes)
		{
		}
		Search& getSearch() noexcept { return search; }
		int getIterations() const noexcept { return iterations; }
		void setPosition(const matrix<Sign> &board, Sign signToMove)
		{ // SearchEngine::setPosition + SearchThread::setPosition: cleanup, Tree::setBoard, Search::setBoard, a fresh selector and generator
			search.cleanup(tree);
			tree.setBoard(board, signToMove);
			search.setBoard(board, signToMove);
			const MCTSConfig &mcts_config = search.getConfig().mcts_config;
			std::unique_ptr<EdgeSelector> tmp = EdgeSelector::create(mcts_config.edge_selector_config);
			tree.setEdgeSelector(*tmp);
			tree.setEdgeGenerator(UnifiedGenerator(mcts_config.max_children, mcts_config.policy_expansion_threshold, mcts_config.policy_temperature));
		}
		void run(NNEvaluator &evaluator, bool asynchronous)
		{ // SearchThread.cpp:87-112
			search.clearStats();
			iterations = 0;
			{ /* artificial scope for lock */
				LowPriorityLock lock = tree.low_priority_lock();
				if (isStopConditionFulfilled())
					return;
			}
			if (asynchronous)
				asynchronous_run(evaluator);
			else
				serial_run(evaluator);
			LowPriorityLock lock = tree.low_priority_lock();
			search.cleanup(tree);
		}
		void stop() noexcept
		{
			std::lock_guard<std::mutex> lock(search_mutex);
			is_running = false;
		}
	private:
		static int get_batch_size(int simulation_count, int max_batch_size) noexcept
		{ // SearchThread.cpp:23-27: doubling the batch size for every 4x increase of the simulation count
			const int tmp = static_cast<int>(std::sqrt(simulation_count));
			return std::max(1, std::min(max_batch_size, tmp));
		}
		void serial_run(NNEvaluator &evaluator)
		{ // SearchThread.cpp:121-146
			while (true)
			{
				{ /* artificial scope for lock */
					LowPriorityLock lock = tree.low_priority_lock();
					const int batch_size = fixed_batch ? search.getConfig().max_batch_size : get_batch_size(tree.getSimulationCount(), search.getConfig().max_batch_size);
					search.setBatchSize(batch_size);
					search.select(tree, max_simulations);
				}
				search.solve();
				search.scheduleToNN(evaluator);
				evaluator.evaluateGraph();

				search.generateEdges(tree); // this step doesn't require locking the tree
				{ /* artificial scope for lock */
					LowPriorityLock lock = tree.low_priority_lock();
					search.expand(tree);
					search.backup(tree);
					iterations++;
					if (isStopConditionFulfilled())
						break;
				}
				std::lock_guard<std::mutex> lock(search_mutex);
				if (is_running == false)
					break;
			}
		}
		void asynchronous_run(NNEvaluator &evaluator)
		{ // SearchThread.cpp:148-180
			search.useBuffer(0);
			double end_time = solve_deadline ? now() + 0.1 : -1.0;
			while (true)
			{
				search.generateEdges(tree); // this step doesn't require locking the tree

				{ /* artificial scope for lock */
					LowPriorityLock lock = tree.low_priority_lock();
					search.expand(tree);
					search.backup(tree);
					iterations++;

					if (isStopConditionFulfilled())
						break;

					const int batch_size = fixed_batch ? search.getConfig().max_batch_size : get_batch_size(tree.getSimulationCount(), search.getConfig().max_batch_size);
					search.setBatchSize(batch_size);
					search.select(tree, max_simulations);
				}
				search.solve(end_time);
				search.scheduleToNN(evaluator);
				evaluator.asyncEvaluateGraphJoin();

				end_time = evaluator.asyncEvaluateGraphLaunch();
				if (!solve_deadline)
					end_time = -1.0; // --solve-deadline 0: the solver keeps its ordinary node budget, so that the run can be replayed move for move
				search.switchBuffer();

				std::lock_guard<std::mutex> lock(search_mutex);
				if (is_running == false)
					break;
			}
			evaluator.asyncEvaluateGraphJoin();
		}
		static double now()
		{ // getTime() (utils/misc.hpp:52-55)
			return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
		}
		bool isStopConditionFulfilled() const
		{ // SearchThread.cpp:181-199 (the memory limit is the node limit here: flat arenas)
			if (tree.getNodeCount() == 0)
				return false;
			if (tree.getSimulationCount() >= max_simulations)
				return true;
			if (tree.getNodeCount() >= max_nodes)
				return true;
			return tree.isRootProven();
		}
};

static int mode_thread(const std::map<std::string, std::string> &a)
{ // a game played by ONE tournament-style engine against itself: per move a search of --sims simulations (serial_run or asynchronous_run),
  // the move by the "best" selector on the root (SearchEngine's choice), Tree::setBoard on the new position
	const int n = geti(a, "--board", 15);
	GameConfig game_config(static_cast<GameRules>(geti(a, "--rules", 0)), n);
	SearchConfig search_config;
	search_config.max_batch_size = geti(a, "--batch", 8);
	search_config.tss_config.hash_table_size = geti(a, "--table-entries", 1 << 16);
	search_config.tree_config.node_bucket_size = geti(a, "--nodes", 4096);
	search_config.tree_config.edge_bucket_size = geti(a, "--edges", 65536);
	const int sims = geti(a, "--sims", 400);
	const bool asynchronous = geti(a, "--async", 1) != 0;
	DeviceConfig device;
	device.batch_size = 64;
	NNEvaluator evaluator(device);
	evaluator.loadGraph(NetworkLoader(a.at("--network")));
	evaluator.useSymmetries(false);
	Tree tree(search_config.tree_config);
	SearchThread thread(game_config, search_config, tree, sims, 1 << 30);
	thread.fixed_batch = geti(a, "--fixed-batch", 0) != 0;
	thread.solve_deadline = geti(a, "--solve-deadline", 1) != 0;

	matrix<Sign> board(n, n);
	board.fill(Sign::NONE);
	std::vector<uint16_t> opening(AGX_OPENING_CAP, 0);
	if (agx_make_opening(static_cast<int>(game_config.rules), n, static_cast<uint32_t>(geti(a, "--opening-seed", 1)), opening.data()) != AGX_OK)
		throw std::runtime_error(agx_last_error());
	Sign sign_to_move = Sign::CROSS;
	std::vector<uint8_t> cells(static_cast<size_t>(n) * n, 0);
	for (int i = 0; i < opening[0]; i++)
	{
		const Move m(opening[1 + i]);
		board.at(m.row, m.col) = m.sign;
		cells[m.row * n + m.col] = static_cast<uint8_t>(m.sign);
		sign_to_move = (m.sign == Sign::CROSS) ? Sign::CIRCLE : Sign::CROSS;
	}
	thread.getSearch().getSolver().clear();
	EdgeSelectorConfig final_selector;
	final_selector.policy = "best";
	std::vector<Move> played;
	std::vector<int> visits;
	int outcome = 0, iterations = 0;
	const int max_plies = geti(a, "--plies", n * n);
	const auto t0 = std::chrono::steady_clock::now();
	while (outcome == 0 && static_cast<int>(played.size()) < max_plies)
	{
		thread.setPosition(board, sign_to_move);
		thread.run(evaluator, asynchronous);
		iterations += thread.getIterations();
		std::unique_ptr<EdgeSelector> selector = EdgeSelector::create(final_selector);
		const Node root_node = tree.getInfo( { });
		const Move m = selector->select(&root_node)->getMove();
		played.push_back(m);
		visits.push_back(root_node.getVisits());
		board.at(m.row, m.col) = m.sign;
		cells[m.row * n + m.col] = static_cast<uint8_t>(m.sign);
		sign_to_move = (m.sign == Sign::CROSS) ? Sign::CIRCLE : Sign::CROSS;
		if (agx_get_outcome(static_cast<int>(game_config.rules), n, cells.data(), static_cast<int>(m.sign), m.row, m.col, game_config.draw_after, &outcome) != AGX_OK)
			throw std::runtime_error(agx_last_error());
	}
	const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	const SearchStats stats = thread.getSearch().getStats();
	std::printf("{\"mode\": \"thread\", \"asynchronous\": %d, \"opening_stones\": %d, \"outcome\": %d, \"iterations\": %d, \"seconds\": %.6f, \"simulations\": %llu, "
			"\"network_evaluations\": %llu, \"moves\": [", asynchronous ? 1 : 0, static_cast<int>(opening[0]), outcome, iterations, seconds,
			static_cast<unsigned long long>(stats.nb_node_count), static_cast<unsigned long long>(stats.nb_network_evaluations));
	for (size_t i = 0; i < played.size(); i++)
		std::printf("%s%d", i ? ", " : "", static_cast<int>(played[i].toShort()));
	std::printf("], \"root_visits\": [");
	for (size_t i = 0; i < visits.size(); i++)
		std::printf("%s%d", i ? ", " : "", visits[i]);
	std::printf("]}\n");
	return 0;
}

static int mode_tree(const std::map<std::string, std::string> &a)
{ // the rest of Tree's public surface (Tree.hpp:70,79-87,100-103) on a stand-alone Tree / Search pair, read after every search of a short game
	const int n = geti(a, "--board", 15);
	GameConfig game_config(static_cast<GameRules>(geti(a, "--rules", 0)), n);
	SearchConfig search_config;
	search_config.max_batch_size = geti(a, "--batch", 4);
	search_config.tss_config.hash_table_size = 1 << 14;
	search_config.tree_config.node_bucket_size = 2048;
	search_config.tree_config.edge_bucket_size = 32768;
	const int sims = geti(a, "--sims", 100);
	DeviceConfig device;
	device.batch_size = 64;
	NNEvaluator evaluator(device);
	evaluator.loadGraph(NetworkLoader(a.at("--network")));
	evaluator.useSymmetries(false);
	Tree tree(search_config.tree_config);
	SearchThread thread(game_config, search_config, tree, sims, 1 << 30);
	thread.fixed_batch = true;
	thread.solve_deadline = false;
	matrix<Sign> board(n, n);
	board.fill(Sign::NONE);
	std::vector<uint16_t> opening(AGX_OPENING_CAP, 0);
	if (agx_make_opening(static_cast<int>(game_config.rules), n, static_cast<uint32_t>(geti(a, "--opening-seed", 1)), opening.data()) != AGX_OK)
		throw std::runtime_error(agx_last_error());
	Sign sign_to_move = Sign::CROSS;
	for (int i = 0; i < opening[0]; i++)
	{
		const Move m(opening[1 + i]);
		board.at(m.row, m.col) = m.sign;
		sign_to_move = (m.sign == Sign::CROSS) ? Sign::CIRCLE : Sign::CROSS;
	}
	EdgeSelectorConfig final_selector;
	final_selector.policy = "best";
	std::vector<int> depths;
	std::vector<float> moves_left;
	bool single_ok = true, proven_ok = true, non_losing_ok = true;
	int depth_after_set_board = -1;
	const int plies = geti(a, "--plies", 6);
	for (int ply = 0; ply < plies; ply++)
	{
		thread.setPosition(board, sign_to_move);
		if (ply == 1)
			depth_after_set_board = tree.getMaximumDepth(); // Tree::setBoard resets it (Tree.cpp:150)
		thread.run(evaluator, false);
		const Node root = tree.getInfo( { });
		depths.push_back(tree.getMaximumDepth());
		moves_left.push_back(tree.getMovesLeft());
		int proven = 0, non_losing = 0;
		for (const Edge *e = root.begin(); e < root.end(); e++)
		{
			proven += e->getScore().isProven() ? 1 : 0;
			non_losing += (e->getScore().getProvenValue() == ProvenValue::LOSS && e->getScore().isFinite()) ? 0 : 1;
		}
		single_ok = single_ok && (tree.hasSingleMove() == (root.numberOfEdges() == 1));
		proven_ok = proven_ok && (tree.hasAllMovesProven() == (proven == root.numberOfEdges()));
		non_losing_ok = non_losing_ok && (tree.hasSingleNonLosingMove() == (non_losing == 1));
		std::unique_ptr<EdgeSelector> selector = EdgeSelector::create(final_selector);
		const Move m = selector->select(&root)->getMove();
		board.at(m.row, m.col) = m.sign;
		sign_to_move = (m.sign == Sign::CROSS) ? Sign::CIRCLE : Sign::CROSS;
	}
	tree.clearNodeCacheStats();
	const NodeCacheStats after_clear_stats = tree.getNodeCacheStats();
	// the PriorityMutex: while this thread holds the tree, a low-priority and then a high-priority locker queue up; released, the
	// high-priority one must get the tree first (utils/PriorityMutex.hpp:15-41)
	std::vector<int> order;
	std::mutex order_mutex;
	{
		std::unique_ptr<HighPriorityLock> held = std::make_unique<HighPriorityLock>(tree.high_priority_lock());
		std::thread low([&]() { LowPriorityLock l = tree.low_priority_lock(); std::lock_guard<std::mutex> g(order_mutex); order.push_back(0); });
		std::this_thread::sleep_for(std::chrono::milliseconds(100));
		std::thread high([&]() { HighPriorityLock l = tree.high_priority_lock(); std::lock_guard<std::mutex> g(order_mutex); order.push_back(1); });
		std::thread low2([&]() { std::this_thread::sleep_for(std::chrono::milliseconds(50)); LowPriorityLock l = tree.low_priority_lock(); std::lock_guard<std::mutex> g(order_mutex); order.push_back(2); });
		std::this_thread::sleep_for(std::chrono::milliseconds(200));
		held.reset();
		low.join();
		high.join();
		low2.join();
	}
	// (the first low-priority locker already stands at the gate when the high-priority one arrives: it may go first; the SECOND low-priority
	//  locker waits in the queue behind it and must come after the high-priority one)
	const bool high_before_second_low = std::find(order.begin(), order.end(), 1) < std::find(order.begin(), order.end(), 2);
	tree.clear();
	const int nodes_after_clear = tree.getNodeCount();
	std::printf("{\"mode\": \"tree\", \"searches\": %d, \"max_depth\": [", plies);
	for (size_t i = 0; i < depths.size(); i++)
		std::printf("%s%d", i ? ", " : "", depths[i]);
	std::printf("], \"moves_left\": [");
	for (size_t i = 0; i < moves_left.size(); i++)
		std::printf("%s%.4f", i ? ", " : "", moves_left[i]);
	std::printf("], \"single_move_matches_edges\": %d, \"all_proven_matches_edges\": %d, \"non_losing_matches_edges\": %d, \"nodes_after_clear\": %d, "
			"\"depth_after_set_board\": %d, \"high_priority_passed_low\": %d, \"stats_after_clear\": %llu}\n", single_ok ? 1 : 0, proven_ok ? 1 : 0, non_losing_ok ? 1 : 0,
			nodes_after_clear, depth_after_set_board, high_before_second_low ? 1 : 0, static_cast<unsigned long long>(after_clear_stats.stored_nodes));
	return 0;
}

static int mode_generator(const std::map<std::string, std::string> &a)
{ // GameGenerator(gameOptions, selfplayOptions, manager, evaluator) as GeneratorThread constructs its generators in the reference
  // (GeneratorManager.cpp:107-110): one game per generator, each with its own tree and search
	const int n = geti(a, "--board", 15);
	GameConfig game_config(static_cast<GameRules>(geti(a, "--rules", 0)), n);
	SelfplayConfig options;
	options.games_per_iteration = geti(a, "--games", 2);
	options.constraints = Constraints::simulations(geti(a, "--sims", 40));
	options.final_selector.policy = "best";
	options.use_symmetries = false;
	options.search_config.max_batch_size = geti(a, "--batch", 8);
	options.search_config.tss_config.hash_table_size = 1 << 14;
	options.search_config.tree_config.node_bucket_size = 2048;
	options.search_config.tree_config.edge_bucket_size = 32768;
	GeneratorManager manager(game_config, options);
	DeviceConfig device;
	device.batch_size = 64;
	NNEvaluator evaluator(device);
	evaluator.loadGraph(NetworkLoader(a.at("--network")));
	std::vector<std::unique_ptr<GameGenerator>> generators;
	for (int i = 0; i < geti(a, "--generators", 3); i++)
		generators.push_back(std::make_unique<GameGenerator>(game_config, options, manager, evaluator));
	int iterations = 0;
	while (manager.getGameBuffer().numberOfGames() < options.games_per_iteration && iterations < geti(a, "--max-iterations", 20000))
	{ // GeneratorThread::run's loop (GeneratorManager.cpp:124-141)
		for (size_t i = 0; i < generators.size(); i++)
		{
			const GameGenerator::Status status = generators[i]->generate();
			if (evaluator.isQueueFull() or status == GameGenerator::TASKS_NOT_READY)
			{
				evaluator.asyncEvaluateGraphJoin();
				evaluator.asyncEvaluateGraphLaunch();
			}
		}
		iterations++;
	}
	evaluator.asyncEvaluateGraphJoin();
	const GameDataBufferStats st = manager.getGameBuffer().getStats();
	std::printf("{\"mode\": \"generator\", \"generators\": %zu, \"iterations\": %d, \"games\": %d, \"samples\": %d}\n", generators.size(), iterations, st.games, st.samples);
	return 0;
}

static int mode_solver(const std::map<std::string, std::string> &a)
{ // AlphaBetaSearch(const GameConfig&) used the way OpeningGenerator.cpp:58-66 and the solver tools use it: one solver object, one table, a
  // sequence of positions (file: one line per position, "<sign to move> <cells as digits 0/1/2>"), the table aged every fourth position
	const int n = geti(a, "--board", 15);
	GameConfig game_config(static_cast<GameRules>(geti(a, "--rules", 0)), n);
	AlphaBetaSearch solver(game_config);
	solver.setNodeLimit(geti(a, "--nodes", 1000));
	std::ifstream in(a.at("--positions"));
	if (!in)
		throw std::runtime_error("cannot open " + a.at("--positions"));
	SearchTask task(game_config);
	std::string cells;
	int sign = 0, index = 0;
	while (in >> sign >> cells)
	{
		if (static_cast<int>(cells.size()) != n * n)
			throw std::runtime_error("a position must have rows x cols cells");
		matrix<Sign> board(n, n);
		for (int i = 0; i < n * n; i++)
			board[i] = static_cast<Sign>(cells[i] - '0');
		if (index % 4 == 3)
			solver.increaseGeneration();
		task.set(board, static_cast<Sign>(sign));
		const int nodes = solver.solve(task);
		unsigned long long feature_sum = 0;
		for (size_t i = 0; i < task.getFeatures().size(); i++)
			feature_sum += static_cast<unsigned long long>(task.getFeatures()[i]) * (i + 1);
		std::printf("{\"index\": %d, \"nodes\": %d, \"score\": %d, \"must_defend\": %d, \"statically_solved\": %d, \"recursively_solved\": %d, \"processed\": %d, "
				"\"value\": [%.6f, %.6f], \"moves_left\": %.1f, \"feature_sum\": %llu, \"edges\": [", index, nodes, static_cast<int>(Score::to_short(task.getScore())),
				task.mustDefend() ? 1 : 0, task.wasStaticallySolved() ? 1 : 0, task.wasRecursivelySolved() ? 1 : 0, task.wasProcessedBySolver() ? 1 : 0,
				task.getValue().win_rate, task.getValue().draw_rate, task.getMovesLeft(), feature_sum);
		for (size_t i = 0; i < task.getEdges().size(); i++)
			std::printf("%s[%d, %d]", i ? ", " : "", static_cast<int>(task.getEdges()[i].getMove().toShort()), static_cast<int>(Score::to_short(task.getEdges()[i].getScore())));
		std::printf("]}\n");
		index++;
	}
	std::fprintf(stderr, "solver memory %lld bytes\n", static_cast<long long>(solver.getMemory()));
	solver.print_stats();
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
	{ // a checkpoint taken before any generator thread ever ran (saveState ahead of the first generate): valid files of zero generators, which the
	  // next loadState reads back — not 0-byte files that make it throw until saved_state/ is removed by hand
		const std::string dir = a.count("--out") ? a.at("--out") : std::string("/tmp/agx_boundary_empty_state");
		std::filesystem::remove_all(dir + "/saved_state");
		GameConfig g(GameRules::FREESTYLE, 15);
		SelfplayConfig sp;
		sp.device_config.assign(2, DeviceConfig());
		sp.device_config[0].device = sp.device_config[1].device = Device::hip(0);
		GeneratorManager manager(g, sp);
		manager.setWorkingDirectory(dir);
		manager.saveState(false);
		const auto size = std::filesystem::file_size(dir + "/saved_state/thread_1.bin");
		GeneratorManager again(g, sp);
		again.setWorkingDirectory(dir);
		again.loadState(); // (would throw "is not a saved generator state")
		if (size >= 12)
			caught |= 32;
	}
	std::printf("{\"mode\": \"errors\", \"caught\": %d}\n", caught);
	return 0;
}

int main(int argc, char **argv)
{
	if (argc < 2)
	{
		std::fprintf(stderr, "usage: agx_boundary_test generate|evaluator|player|thread|tree|generator|solver|errors [--key value ...]\n");
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
		if (mode == "player")
			return mode_player(args);
		if (mode == "thread")
			return mode_thread(args);
		if (mode == "tree")
			return mode_tree(args);
		if (mode == "generator")
			return mode_generator(args);
		if (mode == "solver")
			return mode_solver(args);
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
