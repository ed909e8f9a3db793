/*
 * alphagomoku_agx/search.hpp — ag::NNEvaluator, ag::Search, ag::Tree (and the small value types they exchange) over the device engine.
 *
 * In the reference one GameGenerator owns one Tree and one Search, and a GeneratorThread loops over its GameGenerators
 * (src/selfplay/GameGenerator.cpp:46-121, src/selfplay/GeneratorManager.cpp:124-141).  On the device the unit of work is a SLICE OF THE
 * POOL: every call below acts on all games of the slice in one kernel launch, so ONE Tree / Search pair stands for
 * `games` reference objects.  Method names, argument meaning, stage order and error behaviour are the reference's
 * (include/alphagomoku/search/monte_carlo/Search.hpp:70-90, Tree.hpp:68-104, NNEvaluator.hpp:61-77); everything is enqueued on the
 * slice's HIP stream and returns immediately — the host never waits for the device inside the loop.
 */
#ifndef ALPHAGOMOKU_AGX_SEARCH_HPP_
#define ALPHAGOMOKU_AGX_SEARCH_HPP_

#include "configs.hpp"
#include "networks.hpp"
#include "../agx.h"

#include <chrono>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace ag
{
	enum class Sign : int16_t
	{ // game/Move.hpp:17-23
		NONE, CROSS, CIRCLE, ILLEGAL
	};
	template<typename T>
	class matrix
	{ // utils/matrix.hpp: row-major rows x cols (the part of the interface the path's callers use)
			std::vector<T> m_data;
			int m_rows = 0, m_cols = 0;
		public:
			matrix() = default;
			matrix(int rows, int cols) :
					m_data(static_cast<size_t>(rows) * cols), m_rows(rows), m_cols(cols)
			{
			}
			int rows() const noexcept { return m_rows; }
			int cols() const noexcept { return m_cols; }
			int size() const noexcept { return m_rows * m_cols; }
			T* data() noexcept { return m_data.data(); }
			const T* data() const noexcept { return m_data.data(); }
			T& at(int r, int c) { return m_data.at(static_cast<size_t>(r) * m_cols + c); }
			const T& at(int r, int c) const { return m_data.at(static_cast<size_t>(r) * m_cols + c); }
			T& operator[](int i) noexcept { return m_data[i]; }
			const T& operator[](int i) const noexcept { return m_data[i]; }
			void fill(T value) { m_data.assign(m_data.size(), value); }
	};
	struct Move
	{ // game/Move.hpp:92-174
			Sign sign = Sign::NONE;
			int8_t row = 0, col = 0;
			Move() = default;
			Move(int r, int c, Sign s = Sign::NONE) :
					sign(s), row(static_cast<int8_t>(r)), col(static_cast<int8_t>(c))
			{
			}
			explicit Move(uint16_t s) :
					sign(static_cast<Sign>(s & 3)), row(static_cast<int8_t>((s >> 2) & 127)), col(static_cast<int8_t>((s >> 9) & 127))
			{
			}
			uint16_t toShort() const noexcept
			{ // :144-147
				return static_cast<uint16_t>(sign) | (static_cast<uint16_t>(row) << 2) | (static_cast<uint16_t>(col) << 9);
			}
	};
	enum class ProvenValue
	{ // search/Score.hpp:26-32
		LOSS, DRAW, UNKNOWN, WIN
	};
	class Score
	{ // search/Score.hpp:47-320 (the accessors the path's callers use)
			uint16_t m_data = static_cast<uint16_t>((2u << 13) | 4000u);
		public:
			Score() = default;
			static Score from_short(uint16_t raw) noexcept
			{
				Score s;
				s.m_data = raw;
				return s;
			}
			static uint16_t to_short(Score s) noexcept
			{
				return s.m_data;
			}
			int getEval() const noexcept
			{
				return (m_data & 8191) - 4000;
			}
			ProvenValue getProvenValue() const noexcept
			{
				return static_cast<ProvenValue>((m_data >> 13) & 3);
			}
			bool isFinite() const noexcept
			{
				return m_data != 0x0000 && m_data != 0xFFFF;
			}
			bool isProven() const noexcept
			{
				return getProvenValue() != ProvenValue::UNKNOWN && isFinite();
			}
			bool isUnproven() const noexcept
			{
				return getProvenValue() == ProvenValue::UNKNOWN;
			}
			Value convertToValue() const noexcept
			{ // Score.hpp:266-281
				switch (getProvenValue())
				{
					case ProvenValue::LOSS:
						return Value(0.0f, 0.0f);
					case ProvenValue::DRAW:
						return Value(0.0f, 1.0f);
					case ProvenValue::UNKNOWN:
						return Value((1000 + getEval()) / 2000.0f, 0.0f);
					default:
						return isFinite() ? Value(1.0f, 0.0f) : Value();
				}
			}
			int getDistance() const noexcept
			{
				switch (getProvenValue())
				{
					case ProvenValue::LOSS:
					case ProvenValue::DRAW:
						return getEval();
					case ProvenValue::WIN:
						return -getEval();
					default:
						return 0;
				}
			}
	};
	class Edge
	{ // monte_carlo/Edge.hpp:23-154 (read-only copy)
			AgxEdgeView m { };
		public:
			Edge() = default;
			explicit Edge(const AgxEdgeView &view) :
					m(view)
			{
			}
			float getPolicyPrior() const noexcept { return m.prior; }
			Value getValue() const noexcept { return Value(m.win, m.draw); }
			int getVisits() const noexcept { return m.visits; }
			Move getMove() const noexcept { return Move(m.move); }
			Score getScore() const noexcept { return Score::from_short(m.score); }
			int getVirtualLoss() const noexcept { return m.flag_and_virtual_loss & 0x7FFF; }
			bool isBeingExpanded() const noexcept { return (m.flag_and_virtual_loss & 0x8000) != 0; }
	};
	class Node
	{ // monte_carlo/Node.hpp:24-347: what Tree::getInfo returns — an OWNING copy of a node and its edges (Node.cpp:64-69)
			std::vector<Edge> edges;
			Value value;
			Score score;
			int visits = 0;
			Sign sign_to_move = Sign::NONE;
		public:
			Node() = default;
			Node(std::vector<Edge> e, Value v, Score s, int n, Sign sign) :
					edges(std::move(e)), value(v), score(s), visits(n), sign_to_move(sign)
			{
			}
			const Edge* begin() const noexcept { return edges.data(); }
			const Edge* end() const noexcept { return edges.data() + edges.size(); }
			int numberOfEdges() const noexcept { return static_cast<int>(edges.size()); }
			const Edge& getEdge(int i) const { return edges.at(i); }
			Value getValue() const noexcept { return value; }
			Score getScore() const noexcept { return score; }
			int getVisits() const noexcept { return visits; }
			Sign getSignToMove() const noexcept { return sign_to_move; }
			bool isProven() const noexcept { return score.isProven(); }
	};

	class TimedStat
	{ // utils/statistics.hpp:18-99
			using time_point = std::chrono::time_point<std::chrono::steady_clock, std::chrono::nanoseconds>;
			std::string m_name;
			time_point m_timer_start;
			int64_t m_total_time = 0;
			int64_t m_total_count = 0;
		public:
			TimedStat() = default;
			TimedStat(const std::string &name) :
					m_name(name)
			{
			}
			std::string getName() const { return m_name; }
			double getTotalTime() const noexcept { return m_total_time * 1.0e-9; }
			uint64_t getTotalCount() const noexcept { return m_total_count; }
			void reset() noexcept { m_total_time = m_total_count = 0; }
			void startTimer() noexcept { m_timer_start = std::chrono::steady_clock::now(); }
			void stopTimer(int count = 1) noexcept
			{
				m_total_time += std::chrono::duration<int64_t, std::nano>(std::chrono::steady_clock::now() - m_timer_start).count();
				m_total_count += count;
			}
			void add(double seconds, int64_t count) noexcept
			{
				m_total_time += static_cast<int64_t>(seconds * 1.0e9);
				m_total_count += count;
			}
			std::string toString() const;
			TimedStat& operator+=(const TimedStat &other) noexcept
			{
				m_total_time += other.m_total_time;
				m_total_count += other.m_total_count;
				return *this;
			}
	};

	/* A position evaluated OUTSIDE the pool (openings, a player's root): the part of SearchTask the evaluator touches
	 * (monte_carlo/SearchTask.hpp:35-329).  Pool slices hand their leaves to the evaluator on the device instead (scheduleToNN). */
	class SearchTask
	{
			int rows = 0, cols = 0;
			std::vector<uint32_t> features;    // NNInputFeatures: one word per cell
			std::vector<float> policy;
			std::vector<Value> action_values;
			Value value;
			float moves_left = 0.0f;
			Score score;
			bool processed_by_network = false;
			// the solver's side of a task (SearchTask.hpp:72, 111-117, 155-177, 187-217, 229-273): the position, and what AlphaBetaSearch::solve leaves
			matrix<Sign> board;
			Sign sign_to_move = Sign::NONE;
			std::vector<Edge> edges;
			matrix<Score> action_scores;
			bool processed_by_solver = false, must_defend = false, statically_solved = false, recursively_solved = false;
		public:
			SearchTask() = default;
			SearchTask(int rows, int cols) :
					rows(rows), cols(cols), features(rows * cols), policy(rows * cols), action_values(rows * cols), board(rows, cols), action_scores(rows, cols)
			{
			}
			explicit SearchTask(const GameConfig &config) :
					SearchTask(config.rows, config.cols)
			{
			}
			void set(const matrix<Sign> &base, Sign signToMove); // SearchTask.cpp:32-50
			const matrix<Sign>& getBoard() const noexcept { return board; }
			Sign getSignToMove() const noexcept { return sign_to_move; }
			const std::vector<Edge>& getEdges() const noexcept { return edges; }
			const matrix<Score>& getActionScores() const noexcept { return action_scores; }
			matrix<Score>& getActionScores() noexcept { return action_scores; }
			void addEdge(Move move);                               // SearchTask.cpp:62-71: the edge takes the cell's action score
			bool wasProcessedBySolver() const noexcept { return processed_by_solver; }
			bool mustDefend() const noexcept { return must_defend; }
			bool wasStaticallySolved() const noexcept { return statically_solved; }
			bool wasRecursivelySolved() const noexcept { return recursively_solved; }
			bool isReady() const noexcept { return score.isProven() || processed_by_network; }
			void markAsProcessedBySolver() noexcept { processed_by_solver = true; }
			void markAsDefensive() noexcept { must_defend = true; }
			void markAsStaticallySolved() noexcept { statically_solved = true; }
			void maskAsRecursivelySolved() noexcept { recursively_solved = true; }
			int getRows() const noexcept { return rows; }
			int getCols() const noexcept { return cols; }
			std::vector<uint32_t>& getFeatures() noexcept { return features; }
			const std::vector<uint32_t>& getFeatures() const noexcept { return features; }
			std::vector<float>& getPolicy() noexcept { return policy; }
			const std::vector<float>& getPolicy() const noexcept { return policy; }
			std::vector<Value>& getActionValues() noexcept { return action_values; }
			const std::vector<Value>& getActionValues() const noexcept { return action_values; }
			Value getValue() const noexcept { return value; }
			void setValue(Value v) noexcept { value = v; }
			float getMovesLeft() const noexcept { return moves_left; }
			void setMovesLeft(float m) noexcept { moves_left = m; }
			Score getScore() const noexcept { return score; }
			void setScore(Score s) noexcept { score = s; }
			void markAsProcessedByNetwork() noexcept { processed_by_network = true; }
			bool wasProcessedByNetwork() const noexcept { return processed_by_network; }
	};

	struct NNEvaluatorStats
	{ // NNEvaluator.hpp:29-40
			uint64_t batch_sizes = 0;
			TimedStat pack;
			TimedStat compute;
			TimedStat unpack;
			NNEvaluatorStats();
			std::string toString() const;
			NNEvaluatorStats& operator+=(const NNEvaluatorStats &other) noexcept;
			NNEvaluatorStats& operator/=(int i) noexcept;
	};

	/* EdgeSelector (monte_carlo/EdgeSelector.hpp:31-45).  The SEARCH selector ('puct') runs inside the device's select stage — an object of it
	 * carries its configuration to Tree::setEdgeSelector, which accepts what the engine implements; the FINAL-move selectors ('best',
	 * 'max_visit', 'min_visit', 'max_value', 'max_policy') run here, on the owning copy of the root Tree::getInfo returns (Player.cpp:205-212,
	 * GameGenerator.cpp:161-163). */
	class EdgeSelector
	{
		public:
			EdgeSelector() noexcept = default;
			EdgeSelector(const EdgeSelector &other) = delete;
			EdgeSelector& operator=(const EdgeSelector &other) = delete;
			virtual ~EdgeSelector() = default;
			virtual std::unique_ptr<EdgeSelector> clone() const = 0;
			virtual const Edge* select(const Node *node) noexcept = 0;
			virtual const EdgeSelectorConfig& getConfig() const noexcept = 0;
			static std::unique_ptr<EdgeSelector> create(const EdgeSelectorConfig &config);
	};
	/* EdgeGenerator / UnifiedGenerator (monte_carlo/EdgeGenerator.hpp:51-62): pruning and prior temperature of new nodes' edges; the work is
	 * the first pass of the device's expand stage, the object carries the parameters to Tree::setEdgeGenerator */
	class EdgeGenerator
	{
		public:
			virtual ~EdgeGenerator() = default;
			virtual std::unique_ptr<EdgeGenerator> clone() const = 0;
	};
	class UnifiedGenerator: public EdgeGenerator
	{
			int max_edges;
			float expansion_threshold, temperature;
			bool force_expand_root;
		public:
			UnifiedGenerator(int maxEdges, float expansionThreshold, float temperature, bool forceExpandRoot = false) :
					max_edges(maxEdges), expansion_threshold(expansionThreshold), temperature(temperature), force_expand_root(forceExpandRoot)
			{
			}
			std::unique_ptr<EdgeGenerator> clone() const { return std::make_unique<UnifiedGenerator>(max_edges, expansion_threshold, temperature, force_expand_root); }
			int maxEdges() const noexcept { return max_edges; }
			float expansionThreshold() const noexcept { return expansion_threshold; }
			float policyTemperature() const noexcept { return temperature; }
			bool forceExpandRoot() const noexcept { return force_expand_root; }
	};

	class Tree;
	class NNEvaluator
	{ // NNEvaluator.hpp:42-83
			struct TaskData
			{
					SearchTask *ptr = nullptr;
					int symmetry = 0;
			};
			struct SliceData
			{ // a pool slice whose leaves wait in its device-side queue (Search::scheduleToNN)
					AgxEngine *engine = nullptr;
					int group = 0, n_groups = 1, positions = 0;
					void *stream = nullptr;
					bool *ready_flag = nullptr;
					bool overlap = false; // the network launch goes onto the evaluator's own stream, ordered against `stream` by events
					int event = 0;        // which of the evaluator's two event pairs (= the task buffer)
			};
			void *own_stream = nullptr;            // asyncEvaluateGraphLaunch's stream for double-buffered searches
			void *scheduled_event[2] = { nullptr, nullptr }, *done_event[2] = { nullptr, nullptr };
			AgxTimer *network_timer[2] = { nullptr, nullptr }; // event pairs around the overlapped network launches (one per task buffer)
			bool network_timer_used[2] = { false, false };
			double network_seconds = 0.0;                  // smoothed device time of such a launch: the estimate asyncEvaluateGraphLaunch returns
			std::vector<TaskData> waiting_queue;
			std::vector<TaskData> in_progress_queue;
			std::vector<SliceData> waiting_slices;
			std::vector<SliceData> in_progress_slices;
			std::unique_ptr<AGNetwork> network;
			NNEvaluatorStats stats;
			bool use_symmetries = false;
			DeviceConfig config;
			uint64_t symmetry_counter = 0;
		public:
			NNEvaluator(const DeviceConfig &cfg);

			bool isOnGPU() const noexcept;
			void clearStats() noexcept;
			NNEvaluatorStats getStats() const noexcept;
			bool isQueueFull() const noexcept;
			int getQueueSize() const noexcept;
			void clearQueue() noexcept;
			void useSymmetries(bool b) noexcept;
			bool usesSymmetries() const noexcept
			{
				return use_symmetries;
			}

			void loadGraph(const NetworkLoader &loader);
			void unloadGraph();
			void addToQueue(SearchTask &task);
			void addToQueue(SearchTask &task, int symmetry);
			/* the device-side queue of a pool slice (what Search::scheduleToNN adds): *ready becomes true when the launch that evaluates
			 * it has been joined */
			void addToQueue(AgxEngine *engine, int group, int n_groups, int max_positions, void *stream, bool *ready);
			/* ... of one task buffer of a double-buffered Search (player/SearchThread.cpp:148-180): asyncEvaluateGraphLaunch puts the network
			 * on the evaluator's own stream behind the search stream's work so far; the asyncEvaluateGraphJoin that follows the NEXT launch
			 * makes the search stream wait for it (device-side, the host does not block) — the tower runs beside the other buffer's tree work */
			void addToQueueOverlapped(AgxEngine *engine, int buffer, int max_positions, void *stream, bool *ready);
			~NNEvaluator();
			double evaluateGraph();
			double asyncEvaluateGraphLaunch();
			void asyncEvaluateGraphJoin();
			AGNetwork& get_network();
			const AGNetwork& get_network() const;
			const DeviceConfig& getConfig() const noexcept
			{
				return config;
			}
		private:
			void pack_to_network();
			void unpack_from_network();
	};

	struct SearchStats
	{ // Search.hpp:33-54
			TimedStat select;
			TimedStat solve;
			TimedStat schedule;
			TimedStat generate;
			TimedStat expand;
			TimedStat backup;
			uint64_t nb_duplicate_nodes = 0;
			uint64_t nb_information_leaks = 0;
			uint64_t nb_wasted_expansions = 0;
			uint64_t nb_proven_states = 0;
			uint64_t nb_network_evaluations = 0;
			uint64_t nb_node_count = 0;
			SearchStats();
			std::string toString() const;
			SearchStats& operator+=(const SearchStats &other) noexcept;
			SearchStats& operator/=(int i) noexcept;
			double getTotalTime() const noexcept;
	};
	struct NodeCacheStats
	{ // monte_carlo/NodeCache.hpp (the counters the device keeps)
			uint64_t stored_nodes = 0;   // largest number of nodes a game's cache held
			uint64_t stored_edges = 0;
			std::string toString() const;
			NodeCacheStats& operator+=(const NodeCacheStats &other) noexcept;
			NodeCacheStats& operator/=(int i) noexcept;
	};

	/* The engine of one generator thread, shared by the slices (Tree / Search pairs) it is stepped in. */
	class GamePool
	{
			AgxEngine *engine = nullptr;
			GameConfig game_config;
			SearchConfig search_config;
			int games = 0, batch = 0;
		public:
			/* searchBuffers = 2: ONE tree with two task buffers (games must be 1; AgxEngineConfig.search_buffers) — a stand-alone Search's engine */
			GamePool(const GameConfig &gameOptions, const SearchConfig &searchOptions, const EdgeSelectorConfig &finalSelector, int games, int maxSimulations,
					bool useSymmetries, const std::string &networkOutputs, bool forceExpandRoot = true, int searchBuffers = 1);
			GamePool(const GamePool&) = delete;
			GamePool& operator=(const GamePool&) = delete;
			~GamePool();
			AgxEngine* handle() const noexcept { return engine; }
			const GameConfig& getGameConfig() const noexcept { return game_config; }
			int numberOfGames() const noexcept { return games; }
			int getBatchSize() const noexcept { return batch; }
			const SearchConfig& getSearchConfig() const noexcept { return search_config; }
			void begin(const std::vector<uint16_t> &openings);
			void addOpenings(const std::vector<uint16_t> &openings);
			AgxEngineStats getStats() const;
	};

	/* Tree (Tree.hpp:68-104).  Two ways to get one, as in the header comment:
	 *  - Tree(const TreeConfig&) — the reference's constructor: ONE game, driven from outside (evaluation/Player.cpp:64-75).  The tree's
	 *    storage lives in the one-game engine of the Search it is used with; the first Search method that receives the tree binds them
	 *    (Player::setBoard starts with search.cleanup(tree)).
	 *  - Tree(GamePool&, group, n_groups, stream) — the trees of a slice of a generator thread's pool; the per-game accessors take the game's
	 *    index within the slice. */
	/* utils/PriorityMutex.hpp:15-70: the tree lock of the tournament engine.  Search threads take it with low priority, the thread that
	 * wants the tree for itself (SearchEngine reading the result, setting a position) with high priority: it passes every queued
	 * low-priority waiter.  Three plain mutexes — the gate (`next`) that a high-priority locker only holds while it takes `data`, and the
	 * queue (`low`) that lets only one low-priority locker at a time compete at the gate. */
	class PriorityMutex
	{
			std::mutex data, next, low;
		public:
			void lockHigh()
			{
				std::lock_guard<std::mutex> gate(next);
				data.lock();
			}
			void unlockHigh() { data.unlock(); }
			void lockLow()
			{
				low.lock();
				std::lock_guard<std::mutex> gate(next);
				data.lock();
			}
			void unlockLow()
			{
				data.unlock();
				low.unlock();
			}
	};
	class HighPriorityLock
	{
			PriorityMutex *m;
		public:
			explicit HighPriorityLock(PriorityMutex &pm) : m(&pm) { m->lockHigh(); }
			HighPriorityLock(HighPriorityLock &&other) noexcept : m(other.m) { other.m = nullptr; }
			HighPriorityLock(const HighPriorityLock&) = delete;
			HighPriorityLock& operator=(const HighPriorityLock&) = delete;
			~HighPriorityLock() { if (m != nullptr) m->unlockHigh(); }
	};
	class LowPriorityLock
	{
			PriorityMutex *m;
		public:
			explicit LowPriorityLock(PriorityMutex &pm) : m(&pm) { m->lockLow(); }
			LowPriorityLock(LowPriorityLock &&other) noexcept : m(other.m) { other.m = nullptr; }
			LowPriorityLock(const LowPriorityLock&) = delete;
			LowPriorityLock& operator=(const LowPriorityLock&) = delete;
			~LowPriorityLock() { if (m != nullptr) m->unlockLow(); }
	};

	class Tree
	{
		GamePool *pool = nullptr;
			int group = 0, n_groups = 1;
			void *stream = nullptr;
			int first_game = 0, game_count = 1;
			TreeConfig config;
			bool standalone = false;
			std::unique_ptr<EdgeSelector> edge_selector;
			std::unique_ptr<EdgeGenerator> edge_generator;
			mutable matrix<Sign> board_copy;
			/* stand-alone: root visits / root proven / node count / error, read behind the tree's stream and waiting for that stream only (a
			 * network launch on the evaluator's stream runs on: the double-buffered loop, player/SearchThread.cpp:148-199) */
			mutable int summary[4] = { 0, 0, 0, 0 };
			mutable bool summary_valid = false;
			mutable PriorityMutex tree_mutex; // Tree.hpp:52 (host threads that share this Tree object; the device serialises its own work per stream)
			const int* root_summary() const;
			friend class Search;
			GamePool& bound() const;
		public:
			Tree(const TreeConfig &treeConfig);
			Tree(GamePool &pool, int group, int n_groups, void *stream);
			int64_t getMemory() const noexcept;
			int numberOfGames() const noexcept { return game_count; }
			int firstGame() const noexcept { return first_game; }

			/* Tree.hpp:70-74 (a stand-alone tree).  clear() (Tree.cpp:124-127: node_cache.clear()) empties the tree; the position stays. */
			void clear();
			void setBoard(const matrix<Sign> &newBoard, Sign signToMove, bool forceRemoveRootNode = false);
			void setEdgeSelector(const EdgeSelector &selector);
			void setEdgeGenerator(const EdgeGenerator &generator);

			/* Tree.hpp:76-104: the game of a stand-alone tree, or game `game` of the slice.  These read the game's state back and therefore wait
			 * for the device */
			int getSimulationCount(int game = 0) const;
			bool isRootProven(int game = 0) const;
			int getNodeCount(int game = 0) const;
			int getMoveNumber(int game = 0) const;
			Value getEvaluation(int game = 0) const;
			float getExpectation(int game = 0) const;
			Sign getSignToMove(int game = 0) const;
			const matrix<Sign>& getBoard() const;
			matrix<Sign> getBoard(int game) const;
			/* Tree.hpp:79,83,85-87 — getMovesLeft reads the root's running mean (the reference returns the value the last backup left, Tree.cpp:350:
			 * the same number between a backup and the next setBoard) */
			float getMovesLeft(int game = 0) const;
			int getMaximumDepth(int game = 0) const;
			bool hasAllMovesProven(int game = 0) const;
			bool hasSingleMove(int game = 0) const;
			bool hasSingleNonLosingMove(int game = 0) const;
			Node getInfo(const std::vector<Move> &moves) const; // Tree::getInfo({}) (Tree.cpp:403-424): an owning copy of the root
			Node getInfo(int game, const std::vector<Move> &moves = { }) const;
			void clearNodeCacheStats() noexcept; // Tree.cpp:425-428: the device keeps peaks per game since agx_engine_begin; this forgets what was reported so far
			NodeCacheStats getNodeCacheStats() const noexcept;
			/* Tree.hpp:102-103: guards for HOST threads sharing the Tree object (player/SearchThread.cpp:94,107,126,137,155) */
			LowPriorityLock low_priority_lock() const;
			HighPriorityLock high_priority_lock() const;
			private:
			NodeCacheStats stats_baseline;
			};

	/* AlphaBetaSearch (search/alpha_beta/AlphaBetaSearch.hpp:28-72).  Two ways to get one:
	 *  - Search::getSolver() — the handle of the solver that lives inside a Search's solve stage; what callers do with it is clear() it at
	 *    the start of a game (EvaluationGame.cpp:81-82);
	 *  - AlphaBetaSearch(const GameConfig&) — the reference's constructor: a solver of its own (a one-position engine with the reference's
	 *    4 Mi-entry table, created on first use), whose solve(SearchTask&) runs the device solver on the task's position and leaves what
	 *    AlphaBetaSearch.cpp:77-156 leaves: the feature words, the generated actions as edges with their scores, the position's score (and
	 *    value / moves left when it is proven) and the defensive / statically / recursively-solved marks.  Returns the nodes visited. */
	class AlphaBetaSearch
	{
			friend class Search;
			bool clear_requested = false;
			bool standalone = false;
			GameConfig game_config;
			int max_nodes = 1000;          // AlphaBetaSearch.hpp:36
			int64_t table_entries = 4 * 1024 * 1024;
			AgxEngine *engine = nullptr;   // standalone only
			size_t total_positions = 0, total_calls = 0;
			void require_engine();
		public:
			AlphaBetaSearch() = default;   // (the handle inside a Search)
			explicit AlphaBetaSearch(const GameConfig &gameConfig);
			AlphaBetaSearch(const AlphaBetaSearch&) = delete;
			AlphaBetaSearch& operator=(const AlphaBetaSearch&) = delete;
			~AlphaBetaSearch();
			void clear() noexcept;
			void increaseGeneration();
			int solve(SearchTask &task);
			void print_stats() const;
			int64_t getMemory() const noexcept;
			void setDepthLimit(int depth) noexcept { (void) depth; } // (the iterative deepening's depth cap is the reference's default 100 on the device)
			void setNodeLimit(int nodes);
			void setTimeLimit(double time) noexcept { (void) time; }   // (time shares are Search::solve(endTime)'s business)
	};

	/* Search (Search.hpp:56-101): Search(const GameConfig&, const SearchConfig&) is the reference's constructor — it owns a one-game engine
	 * (the AlphaBetaSearch with its table, the task buffers) and works on a Tree(const TreeConfig&); Search(GamePool&, ...) steps a slice of a
	 * generator thread's pool.  The solver (AlphaBetaSearch) lives inside the solve stage. */
	class Search
	{
			std::unique_ptr<GamePool> own_pool; // stand-alone: the engine of this Search / Tree pair
			GamePool &pool;
			int group, n_groups;         // stand-alone: group = the current task buffer of 2
			void *stream;
			void *own_stream = nullptr;  // stand-alone: created with the engine
			int batch_size;
			bool scheduled = false;
			bool ready_flags[2] = { true, true }; // per task buffer (an empty buffer is ready)
			bool select_pending = false; // select() asked for, enqueued together with solve()
			int current_task_buffer = 0;
			SearchStats stats;
			AlphaBetaSearch ab_search;
			void flush_select();
			void bind(Tree &tree);
		public:
			static constexpr int maximum_number_of_simulations = 16777216;
			Search(const GameConfig &gameOptions, const SearchConfig &searchOptions);
			Search(GamePool &pool, int group, int n_groups, void *stream);
			~Search();
			Search(const Search&) = delete;
			Search& operator=(const Search&) = delete;

			int64_t getMemory() const noexcept;
			const SearchConfig& getConfig() const noexcept;
			AlphaBetaSearch& getSolver() noexcept;
			void clearStats() noexcept;
			SearchStats getStats() const noexcept;  // counters of the whole pool (the device keeps them per game, not per slice)

			void setBoard(const matrix<Sign> &board, Sign signToMove);
			void select(Tree &tree, int maxSimulations = maximum_number_of_simulations);
			void solve(double endTime = -1.0);
			void scheduleToNN(NNEvaluator &evaluator);
			bool areTasksReady() const noexcept;
			void generateEdges(const Tree &tree);
			void expand(Tree &tree);
			void backup(Tree &tree);
			void cleanup(Tree &tree);

			/* Search.hpp:88-89: the two task buffers of the tournament engine's double-buffered loop (player/SearchThread.cpp:148-180).  A
			 * stand-alone Search's engine has both (AgxEngineConfig.search_buffers = 2): every stage works on the current one, the leaves of the
			 * other keep their virtual losses meanwhile, cleanup() drops both.  A pool slice has one buffer (a device step completes its batch). */
			void useBuffer(int index);
			void switchBuffer() noexcept;
			void setBatchSize(int batchSize);
			int getBatchSize() const noexcept;
	};
} /* namespace ag */

#endif
