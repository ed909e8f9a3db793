/*
 * alphagomoku_agx/selfplay.hpp — ag::GameDataBuffer, ag::GameGenerator, ag::GeneratorThread, ag::GeneratorManager: the objects
 * training_launcher reaches through TrainingManager::generateGames (src/selfplay/TrainingManager.cpp:194-214):
 *
 *     GeneratorManager manager(gameConfig, selfplayConfig);        GeneratorManager.hpp:96-121
 *     manager.setWorkingDirectory(path);  manager.loadState();
 *     manager.generate(NetworkLoader(path_to_network), games);     one GeneratorThread per selfplayConfig.device_config[i]
 *     manager.saveState(...);   manager.getGameBuffer().save(...)
 *
 * Same names, arguments, state machine and threading as the reference (src/selfplay/GeneratorManager.cpp:28-218,
 * src/selfplay/GameGenerator.cpp:46-185): one host thread (std::async) per device; everything reachable from it is single-threaded; the
 * threads meet only in GeneratorManager::addToBuffer / hasEnoughGames under buffer_mutex.  What differs is the granularity: one
 * GameGenerator stands for a SLICE of the thread's game pool (games_per_thread games split into DeviceConfig::batch_size-sized network
 * launches), so `generators` holds a few slices instead of games_per_thread single games, and GeneratorThread::run's loop — generate();
 * if the evaluator's queue is full or tasks are not ready: Join, Launch — pipelines the slices on their streams exactly as the
 * reference pipelines host search against the network (NNEvaluator.cpp:182-228).
 */
#ifndef ALPHAGOMOKU_AGX_SELFPLAY_HPP_
#define ALPHAGOMOKU_AGX_SELFPLAY_HPP_

#include "configs.hpp"
#include "networks.hpp"
#include "search.hpp"
#include "../agx.h"

#include <atomic>
#include <future>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace ag
{
	struct GameDataBufferStats
	{ // dataset/GameDataBuffer.hpp
			int games = 0, samples = 0, cross_win = 0, draws = 0, circle_win = 0, game_length = 0;
			std::string toString() const;
	};
	/* GameDataBuffer (src/dataset/GameDataBuffer.cpp) for dataset format 201, producing side: games arrive as the bytes of
	 * GameDataStorage::serialize, their samples quantised on the device. */
	class GameDataBuffer
	{
			AgxGameBuffer *buffer = nullptr;
			GameConfig game_config;
		public:
			GameDataBuffer(GameConfig cfg);
			GameDataBuffer(const GameDataBuffer&) = delete;
			GameDataBuffer& operator=(const GameDataBuffer&) = delete;
			~GameDataBuffer();
			const GameConfig& getConfig() const noexcept;
			void clear() noexcept;
			int numberOfGames() const noexcept;
			int numberOfSamples() const noexcept;
			std::vector<uint8_t> getGameData(int index) const; // GameDataStorage::serialize bytes
			void save(const std::string &path) const;
			void load(const std::string &path); // GameDataBuffer.cpp:115-131: appends the games of a file written by save()
			GameDataBufferStats getStats() const noexcept;
			AgxGameBuffer* handle() const noexcept
			{
				return buffer;
			}
	};

	class GeneratorManager;

	/* utils/os_utils.hpp:46-63: the process-wide signal flags.  TrainingManager installs the custom SIGINT handler once and GeneratorManager::generate
	 * polls hasCapturedSignal(SignalType::INT) to stop its threads and return, so that the caller reaches saveState (TrainingManager.cpp:88-113,
	 * 205-209).  The handler only sets a flag (async-signal-safe); a captured signal stays captured, as in the reference. */
	enum class SignalType
	{
		INT, ILL, ABRT, FPE, SEGV, TERM
	};
	enum class SignalHandlerMode
	{
		DEFAULT_HANDLER, IGNORE_SIGNAL, CUSTOM_HANDLER
	};
	void setupSignalHandler(SignalType type, SignalHandlerMode mode);
	bool hasCapturedSignal(SignalType type) noexcept;

	class GameGenerator
	{ // selfplay/GameGenerator.hpp:25-66
		private:
			enum GameState
			{
				GAME_NOT_STARTED, PREPARE_OPENING, GAMEPLAY_SELECT_SOLVE_EVALUATE, GAMEPLAY_EXPAND_AND_BACKUP
			};
			GeneratorManager &manager;
			NNEvaluator &nn_evaluator;
			GameConfig game_config;
			std::unique_ptr<GamePool> own_pool; // the 4-argument form: this generator's own one-game pool (created by the first generate())
			GamePool *pool = nullptr;
			std::unique_ptr<Tree> tree;
			std::unique_ptr<Search> search;
			GameState state = GAME_NOT_STARTED;
			SelfplayConfig selfplay_config;
			int group = 0, n_groups = 1;
			void *stream = nullptr;
			uint64_t steps = 0;
			std::vector<void*> pace_events;     // host pacing: blocking events behind the last steps (agx.h: agx_event_create_blocking)
			int own_openings = 0;
			uint32_t opening_seed = 0;
			void start_own_pool();
			void serve_own_pool();
		public:
			enum Status
			{
				OK, TASKS_NOT_READY
			};
			/* the reference's constructor (GameGenerator.hpp:54): ONE game with its own Tree and Search.  A generator thread that wants the
			 * device filled gives its generators slices of one pool instead (below). */
			GameGenerator(const GameConfig &gameOptions, const SelfplayConfig &selfplayOptions, GeneratorManager &manager, NNEvaluator &evaluator);
			/* slice `group` of `n_groups` of the thread's pool, driven on `stream` */
			GameGenerator(const GameConfig &gameOptions, const SelfplayConfig &selfplayOptions, GeneratorManager &manager, NNEvaluator &evaluator, GamePool &pool,
					int group, int n_groups, void *stream);

			~GameGenerator();
			void clearStats();
			NodeCacheStats getCacheStats() const noexcept;
			SearchStats getSearchStats() const noexcept;

			Status generate();
			/* GameGenerator.cpp:122-141: the games of this generator that are in flight — their moves so far and the samples collected so far; search
			 * trees are not saved, load() gives every restored game a fresh one (prepare_search).  The reference writes Json + SerializedObject
			 * (MinML); here both travel in one byte vector: u32 count, then per game an AgxSavedGame, u64 n, n bytes of samples.  load() returns the
			 * offset behind what it read; games whose slot lies outside this generator's slice are skipped. */
			void save(std::vector<uint8_t> &binary_data);
			size_t load(const std::vector<uint8_t> &binary_data, size_t offset = 0);
			private:
			void make_move();
			void prepare_search();
	};

	class GeneratorThread
	{ // selfplay/GeneratorManager.hpp:53-79
		private:
			std::string working_directory;
			std::future<void> generator_future;
			std::atomic<bool> is_running;

			GeneratorManager &manager;
			NNEvaluator nn_evaluator;
			GameConfig game_config;
			SelfplayConfig selfplay_config;
			int index;
			std::unique_ptr<GamePool> pool;
			std::vector<void*> streams;
			std::vector<std::unique_ptr<GameGenerator>> generators;
			mutable std::mutex stats_mutex;
			SearchStats last_search_stats;
			NodeCacheStats last_cache_stats;
			std::vector<uint8_t> saved_games; // the games in flight when run() ended (GameGenerator::save of every slice), or what loadGames read
		public:
			GeneratorThread(GeneratorManager &manager, const GameConfig &gameOptions, const SelfplayConfig &selfplayOptions, int index);
			~GeneratorThread();
			void start();
			void stop();
			bool isFinished() const noexcept;
			void clearStats() noexcept;
			void setWorkingDirectory(const std::string &path);
			NNEvaluatorStats getEvaluatorStats() const noexcept;
			NodeCacheStats getCacheStats() const noexcept;
			SearchStats getSearchStats() const noexcept;
			/* hands the samples and finished games of this thread's pool to the manager's buffer (GeneratorManager::addToBuffer) */
			void collectGames();
			/* GeneratorManager.cpp:98-122: the games that were in flight when the thread stopped, to / from `path`; the next start() continues them */
			void saveGames(const std::string &path) const;
			void loadGames(const std::string &path);
		private:
			void run();
			void setup();
			void teardown();
			void park_games_in_flight();
	};

	class GeneratorManager
	{ // selfplay/GeneratorManager.hpp:81-121
		private:
			mutable std::mutex buffer_mutex;
			std::vector<std::unique_ptr<GeneratorThread>> generators;
			GameDataBuffer game_buffer;

			int games_to_generate = 0;
			std::string working_directory;
			NetworkLoader network_loader;
			int stats_period_seconds = 60; // GeneratorManager.cpp:198-199
		public:
			GeneratorManager(const GameConfig &gameOptions, const SelfplayConfig &selfplayOptions);
			/* how often generate() prints the statistics (the reference: every 60 s, not configurable; tests shorten it) */
			void setStatsPeriod(int seconds) noexcept { stats_period_seconds = (seconds > 0) ? seconds : 60; }

			void setWorkingDirectory(const std::string &path);
			/* the device-resident counterpart of addToBuffer(const GameDataStorage&): drains `engine`'s record pools into the buffer under
			 * buffer_mutex; returns the number of games added */
			int addToBuffer(AgxEngine *engine);

			const GameDataBuffer& getGameBuffer() const noexcept;
			GameDataBuffer& getGameBuffer() noexcept;
			const NetworkLoader& getNetworkLoader() const noexcept;

			void generate(const NetworkLoader &loader, int numberOfGames);
			bool hasEnoughGames() const noexcept;

			void printStats();

			void saveState(bool saveBuffer);
			void loadState();
	};
} /* namespace ag */

#endif
