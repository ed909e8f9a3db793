/*
 * alphagomoku_agx/configs.hpp — the configuration structs the self-play path reads, with the reference's names, fields and defaults
 * (include/alphagomoku/utils/configs.hpp:22-262, include/alphagomoku/game/rules.hpp:18-35).  The reference builds them from MinML
 * Json objects; MinML is not part of the reference tree, so these are plain structs (a maintainer keeps the Json constructors and
 * copies the fields over).
 */
#ifndef ALPHAGOMOKU_AGX_CONFIGS_HPP_
#define ALPHAGOMOKU_AGX_CONFIGS_HPP_

#include <limits>
#include <string>
#include <vector>

namespace ag
{
	enum class GameRules
	{ // game/rules.hpp:18-25
		FREESTYLE, STANDARD, RENJU, CARO5, CARO6
	};
	enum class GameOutcome
	{ // game/rules.hpp:29-35
		UNKNOWN, DRAW, CROSS_WIN, CIRCLE_WIN
	};
	std::string toString(GameRules rules);
	std::string toString(GameOutcome outcome);

	struct GameConfig
	{ // configs.hpp:22-44
			GameRules rules = GameRules::FREESTYLE;
			int rows = 0;
			int cols = 0;
			int draw_after = 0;
			GameConfig() = default;
			GameConfig(GameRules rules, int rows, int cols) :
					rules(rules), rows(rows), cols(cols), draw_after(rows * cols)
			{
			}
			GameConfig(GameRules rules, int size) :
					GameConfig(rules, size, size)
			{
			}
	};
	struct TreeConfig
	{ // configs.hpp:46-66.  The device engine keeps per-game arenas: node_bucket_size / edge_bucket_size size them (records per game)
			float information_leak_threshold = 0.01f;
			int initial_node_cache_size = 65536;
			int edge_bucket_size = 200000;
			int node_bucket_size = 10000;
	};
	struct EdgeSelectorConfig
	{ // configs.hpp:68-88
			std::string policy = "puct"; // search: 'puct'; final selector: 'max_value', 'max_policy', 'max_visit', 'min_visit', 'best', 'lcb'
			std::string init_to = "q_head"; // 'parent', 'loss', 'draw', 'q_head'
			std::string noise_type = "none"; // 'none', 'custom', 'dirichlet', 'gumbel'
			float noise_weight = 0.0f;
			float exploration_constant = 1.25f;
			float exploration_scaling = 0.0f;
	};
	struct MCTSConfig
	{ // configs.hpp:90-109
			EdgeSelectorConfig edge_selector_config;
			int max_children = std::numeric_limits<int>::max();
			float policy_expansion_threshold = 1.0e-4f;
			float policy_temperature = 1.0f;
	};
	struct TSSConfig
	{ // configs.hpp:111-128.  NB AlphaBetaSearch sizes its SharedHashTable itself (4 Mi entries, AlphaBetaSearch.cpp:59): hash_table_size here
	  // is what the device engine uses per game
			int mode = 0;
			int max_positions = 100;
			int hash_table_size = 4 * 1048576;
	};
	struct SearchConfig
	{ // configs.hpp:130-151
			int max_batch_size = 1;
			double early_stopping = 0.99;
			double time_fraction_15x15 = 0.9;
			double time_fraction_20x20 = 0.9;
			TreeConfig tree_config;
			MCTSConfig mcts_config;
			TSSConfig tss_config;
	};
	/* ml::Device of MinML reduced to what the path needs: which GPU */
	class Device
	{
			int m_index = 0;
			bool m_cpu = false;
		public:
			static Device cpu() noexcept
			{
				Device d;
				d.m_cpu = true;
				return d;
			}
			static Device hip(int index) noexcept
			{
				Device d;
				d.m_index = index;
				return d;
			}
			bool isCPU() const noexcept
			{
				return m_cpu;
			}
			int index() const noexcept
			{
				return m_index;
			}
			std::string toString() const
			{
				return m_cpu ? std::string("CPU") : ("HIP:" + std::to_string(m_index));
			}
	};
	struct DeviceConfig
	{ // configs.hpp:153-167.  batch_size: positions per network launch; a pool whose games x max_batch_size exceeds it is stepped as
	  // several slices ("groups") on separate streams, each with its own launch (NNEvaluator::asyncEvaluateGraphLaunch per slice)
			Device device = Device::hip(0);
			int batch_size = 1;
	};
	struct Constraints
	{ // configs.hpp:192-214 (the self-play path uses the simulation budget only)
			enum Type
			{
				SIMULATIONS, TIME
			};
			double time_for_match = 0.0;
			double time_for_turn = 0.0;
			double time_increment = 0.0;
			int max_simulations = 0;
			Type type = Type::SIMULATIONS;
			static Constraints simulations(int max_sim) noexcept
			{
				Constraints c;
				c.max_simulations = max_sim;
				return c;
			}
	};
	struct SelfplayConfig
	{ // configs.hpp:216-232
			bool use_opening = true;
			bool use_symmetries = true;
			bool keep_loaded = false;
			int games_per_iteration = 100;
			int games_per_thread = 8;
			Constraints constraints;
			EdgeSelectorConfig final_selector;
			std::vector<DeviceConfig> device_config = { DeviceConfig() };
			SearchConfig search_config;
	};
} /* namespace ag */

#endif
