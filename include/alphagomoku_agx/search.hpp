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
#include <string>
#include <vector>

namespace ag
{
	enum class Sign : int16_t
	{ // game/Move.hpp:17-23
		NONE, CROSS, CIRCLE, ILLEGAL
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
		public:
			SearchTask() = default;
			SearchTask(int rows, int cols) :
					rows(rows), cols(cols), features(rows * cols), policy(rows * cols), action_values(rows * cols)
			{
			}
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
			};
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
			int games = 0, batch = 0;
		public:
			GamePool(const GameConfig &gameOptions, const SearchConfig &searchOptions, const EdgeSelectorConfig &finalSelector, int games, int maxSimulations,
					bool useSymmetries, const std::string &networkOutputs);
			GamePool(const GamePool&) = delete;
			GamePool& operator=(const GamePool&) = delete;
			~GamePool();
			AgxEngine* handle() const noexcept { return engine; }
			const GameConfig& getGameConfig() const noexcept { return game_config; }
			int numberOfGames() const noexcept { return games; }
			int getBatchSize() const noexcept { return batch; }
			void begin(const std::vector<uint16_t> &openings);
			void addOpenings(const std::vector<uint16_t> &openings);
			AgxEngineStats getStats() const;
	};

	class Tree
	{ // Tree.hpp:68-104 for the games of one slice
			GamePool &pool;
			int group, n_groups;
			void *stream;
			int first_game, game_count;
			friend class Search;
		public:
			Tree(GamePool &pool, int group, int n_groups, void *stream);
			int64_t getMemory() const noexcept;
			int numberOfGames() const noexcept { return game_count; }
			int firstGame() const noexcept { return first_game; }
			/* per game (index within the slice); these read the game's state back and therefore wait for the device */
			int getSimulationCount(int game) const;
			bool isRootProven(int game) const;
			int getNodeCount(int game) const;
			int getMoveNumber(int game) const;
			Value getEvaluation(int game) const;
			Sign getSignToMove(int game) const;
			std::vector<Sign> getBoard(int game) const;
			Node getInfo(int game, const std::vector<Move> &moves = { }) const; // Tree::getInfo({}) (Tree.cpp:403-424): the root
			NodeCacheStats getNodeCacheStats() const noexcept;
	};

	class Search
	{ // Search.hpp:56-101 for the games of one slice; the solver (AlphaBetaSearch) lives inside the solve stage
			GamePool &pool;
			int group, n_groups;
			void *stream;
			int batch_size;
			bool scheduled = false;
			bool tasks_ready = true;
			bool select_pending = false; // select() asked for, enqueued together with solve()
			SearchStats stats;
			void flush_select();
		public:
			static constexpr int maximum_number_of_simulations = 16777216;
			Search(GamePool &pool, int group, int n_groups, void *stream);

			void clearStats() noexcept;
			SearchStats getStats() const noexcept;  // counters of the whole pool (the device keeps them per game, not per slice)

			void select(Tree &tree, int maxSimulations = maximum_number_of_simulations);
			void solve(double endTime = -1.0);
			void scheduleToNN(NNEvaluator &evaluator);
			bool areTasksReady() const noexcept;
			void generateEdges(const Tree &tree);
			void expand(Tree &tree);
			void backup(Tree &tree);
			void cleanup(Tree &tree);

			void setBatchSize(int batchSize);
			int getBatchSize() const noexcept;
	};
} /* namespace ag */

#endif
