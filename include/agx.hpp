/*
 * agx.hpp — C++ host facade over the C ABI (agx.h), mirroring the reference's operator interface for the self-play path.
 *
 * Same names, argument meaning and error behaviour as the reference classes it stands in for:
 *   agx::AGNetwork      <- ag::AGNetwork   (include/alphagomoku/networks/AGNetwork.hpp:60-98): loadWeights / forward / setBatchSize-free
 *   agx::GeneratorPool  <- one ag::GeneratorThread with its GameGenerators (src/selfplay/GeneratorManager.cpp:124-141,
 *                          src/selfplay/GameGenerator.cpp:46-121): generate() = one select/solve/evaluate/expand/backup/move step
 * Errors surface as std::runtime_error / std::logic_error exactly where the reference throws (NNEvaluator.cpp:149,185-187).
 * Header-only convenience layer (used by csrc/selfplay_main.cpp); link with -lagx.  The compiled, reference-NAMED classes — ag::NNEvaluator,
 * ag::Search, ag::Tree, ag::GameGenerator, ag::GeneratorThread, ag::GeneratorManager, ag::GameDataBuffer — live in
 * include/alphagomoku_agx/ (libagx_ag.so).
 */
#ifndef AGX_HPP_
#define AGX_HPP_

#include "agx.h"

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <atomic>
#include <vector>

namespace agx
{
	inline void check(int status)
	{
		if (status == AGX_OK)
			return;
		const std::string msg = agx_last_error();
		if (status == AGX_ERR_INVALID || status == AGX_ERR_STATE)
			throw std::logic_error(msg);
		throw std::runtime_error(msg);
	}

	/* Keeps a launch loop at most `ahead` steps in front of a stream: call step(stream) behind every step's launches.  Without it the host
	 * enqueues until the launch queue is full and spins there — a whole CPU per device thread; with it the thread sleeps on a blocking event
	 * (agx.h: agx_event_create_blocking).  Pacing only: the device never waits for the host while `ahead` >= 1 step is queued. */
	class HostPacer
	{
			std::vector<void*> m_events;
			uint64_t m_count = 0;
			int m_ahead;
		public:
			explicit HostPacer(int ahead = 2) :
					m_events(static_cast<size_t>(ahead > 0 ? ahead + 1 : 0), nullptr), m_ahead(ahead)
			{
				for (void *&e : m_events)
					check(agx_event_create_blocking(&e));
			}
			HostPacer(const HostPacer&) = delete;
			HostPacer& operator=(const HostPacer&) = delete;
			HostPacer(HostPacer &&other) noexcept :
					m_events(std::move(other.m_events)), m_count(other.m_count), m_ahead(other.m_ahead)
			{
				other.m_events.clear();
			}
			~HostPacer()
			{
				for (void *e : m_events)
					agx_event_destroy(e);
			}
			void step(void *stream)
			{
				if (m_ahead <= 0)
					return;
				const size_t ring = m_events.size();
				check(agx_event_record(m_events[m_count % ring], stream));
				if (m_count >= static_cast<uint64_t>(m_ahead))
					check(agx_event_synchronize(m_events[(m_count - m_ahead) % ring]));
				m_count++;
			}
	};

	struct GameConfig
	{ // ag::GameConfig (utils/configs.hpp:23-44)
			int rules = AGX_FREESTYLE;
			int rows = 15, cols = 15;
			int draw_after = 225;
	};

	class AGNetwork
	{
			AgxNet *m_net = nullptr;
			AgxNetDesc m_desc { };
		public:
			/* architecture "ResnetPV" (networks.cpp:71-93) or "ResnetPVQ" (:143-168, adds the action-values head) */
			AGNetwork(const GameConfig &cfg, int blocks, int filters, const std::string &architecture = "ResnetPV")
			{
				if (architecture != "ResnetPV" && architecture != "ResnetPVQ")
					throw std::logic_error("AGNetwork: unknown architecture '" + architecture + "'");
				m_desc.action_values = (architecture == "ResnetPVQ") ? 1 : 0;
				m_desc.rows = cfg.rows;
				m_desc.cols = cfg.cols;
				m_desc.blocks = blocks;
				m_desc.filters = filters;
				m_desc.in_channels = 32;
				m_desc.value_hidden = (2 * filters < 256) ? 2 * filters : 256;
				check(agx_net_create(&m_desc, &m_net));
			}
			AGNetwork(const AGNetwork&) = delete;
			AGNetwork& operator=(const AGNetwork&) = delete;
			~AGNetwork()
			{
				agx_net_destroy(m_net);
			}
			size_t numberOfWeights() const
			{
				return agx_net_blob_floats(&m_desc);
			}
			void loadWeights(const std::vector<float> &blob)
			{
				check(agx_net_load_weights(m_net, blob.data(), blob.size()));
			}
			/* device buffers: features uint32[batch][rows*cols] -> policy float[batch][rows*cols], value float[batch][3] */
			void forward(const uint32_t *d_features, int batch, float *d_policy, float *d_value, void *stream = nullptr)
			{
				check(agx_nn_forward(m_net, d_features, batch, d_policy, d_value, stream));
			}
			/* 'pvq' networks: additionally action values float[batch][rows*cols][2] = (win, draw) */
			void forward(const uint32_t *d_features, int batch, float *d_policy, float *d_value, float *d_action_values, void *stream)
			{
				check(agx_nn_forward_pvq(m_net, d_features, batch, d_policy, d_value, d_action_values, stream));
			}
			std::string getOutputConfig() const
			{ // AGNetwork::getOutputConfig
				return m_desc.action_values ? "pvq" : "pv";
			}
			AgxNet* handle() const noexcept
			{
				return m_net;
			}
			const AgxNetDesc& description() const noexcept
			{
				return m_desc;
			}
	};

	struct SearchConfig
	{ // the fields of ag::SearchConfig / MCTSConfig / EdgeSelectorConfig / TreeConfig / TSSConfig that the path reads
			int max_batch_size = 8;
			float exploration_constant = 1.25f;
			float exploration_scaling = 0.0f;
			std::string init_to = "q_head";
			std::string noise_type = "none";      // EdgeSelectorConfig::noise_type: "none", "custom", "dirichlet", "gumbel"
			float noise_weight = 0.0f;
			float policy_expansion_threshold = 1.0e-4f;
			float policy_temperature = 1.0f; // MCTSConfig::policy_temperature
			int max_children = 0;                 // MCTSConfig::max_children, 0 = unlimited
			float information_leak_threshold = 0.01f;
			int tss_max_positions = 100;
			uint64_t tss_table_entries = 4ull * 1024ull * 1024ull;
	};
	inline int final_selector_id(const std::string &policy)
	{ // EdgeSelector::create (EdgeSelector.cpp:680-711): the selectors GameGenerator::make_move can be configured with
		if (policy == "best") return 0;
		if (policy == "max_visit") return 1;
		if (policy == "min_visit") return 2;
		if (policy == "max_value") return 3;
		if (policy == "max_policy") return 4;
		if (policy == "lcb") return 5;
		throw std::logic_error("Unknown selection policy '" + policy + "'");
	}

	struct SelfplayConfig
	{ // ag::SelfplayConfig subset (utils/configs.hpp:205-255)
			int games_per_thread = 1024;
			int max_simulations = 400;
			bool use_symmetries = false;         // SelfplayConfig::use_symmetries -> NNEvaluator::useSymmetries
			std::string network_outputs = "pv";  // AGNetwork::getOutputConfig of the network that will evaluate: "pv" or "pvq"
			std::string final_selector = "best"; // SelfplayConfig::final_selector.policy
			int record_format = 1;               // AgxEngineConfig::record_format: 1 root-edge snapshots, 2 format-201 samples, 3 both
			SearchConfig search_config;
	};

	class GeneratorPool
	{
			AgxEngine *m_engine = nullptr;
			AgxEngineBuffers m_buffers { };
			std::vector<void*> m_slice_streams; // useChipSlices: one CU-masked stream per slice of the pool
			std::vector<HostPacer> m_pacers;    // one per slice (or one for the un-sliced pool)
			int m_steps_ahead = 2;
			bool m_phase_started = false, m_skip_first_search_of_slice0 = false;
			int m_games = 0;
			bool m_match = false;
		public:
			/* evaluationMatches: the pool plays EvaluationGames instead (evaluation/EvaluationGame.cpp): games_per_thread pairs of Players,
			 * one tree per Player, every opening twice with the colours swapped; drive it with generate(first, second) */
			GeneratorPool(const GameConfig &game, const SelfplayConfig &selfplay, bool evaluationMatches = false)
			{
				AgxEngineConfig c;
				check(agx_engine_default_config(&c));
				c.rules = game.rules;
				c.board_size = game.rows;
				c.draw_after = game.draw_after;
				c.n_games = evaluationMatches ? 2 * selfplay.games_per_thread : selfplay.games_per_thread;
				c.match_mode = evaluationMatches ? 1 : 0;
				c.max_batch_size = selfplay.search_config.max_batch_size;
				c.max_simulations = selfplay.max_simulations;
				c.exploration_constant = selfplay.search_config.exploration_constant;
				c.exploration_scaling = selfplay.search_config.exploration_scaling;
				const std::string &init = selfplay.search_config.init_to;
				c.init_to = (init == "q_head") ? 0 : (init == "parent") ? 1 : (init == "draw") ? 2 : 3; // EdgeSelector.cpp:1140-1165
				c.policy_expansion_threshold = selfplay.search_config.policy_expansion_threshold;
				c.max_children = selfplay.search_config.max_children;
				c.policy_temperature = selfplay.search_config.policy_temperature;
				c.information_leak_threshold = selfplay.search_config.information_leak_threshold;
				c.tss_max_positions = selfplay.search_config.tss_max_positions;
				c.tss_table_entries = selfplay.search_config.tss_table_entries;
				const std::string &noise = selfplay.search_config.noise_type;
				if (noise != "none" && noise != "custom" && noise != "dirichlet" && noise != "gumbel")
					throw std::logic_error("GeneratorPool: unknown noise_type '" + noise + "'");
				c.noise_type = (noise == "custom") ? 1 : ((noise == "dirichlet") ? 2 : ((noise == "gumbel") ? 3 : 0));
				c.noise_weight = selfplay.search_config.noise_weight;
				c.final_selector = final_selector_id(selfplay.final_selector);
				c.use_symmetries = selfplay.use_symmetries ? 1 : 0;
				c.action_values = (selfplay.network_outputs == "pvq") ? 1 : 0;
				c.record_format = selfplay.record_format;
				// pacing only, per-game results do not depend on either (agx.h): the leaves of a batch solved in parallel, a launch's stragglers
				// put off to the next launch
				c.speculative_solver = 1;
				c.solver_yield_fraction = (c.n_games >= 64) ? 0.5f : 0.0f;
				if (game.rows != game.cols)
					throw std::logic_error("GeneratorPool: only square boards are supported");
				check(agx_engine_create(&c, &m_engine));
				check(agx_engine_buffers(m_engine, &m_buffers));
				m_games = c.n_games;
				m_match = evaluationMatches;
			}
			GeneratorPool(const GeneratorPool&) = delete;
			GeneratorPool& operator=(const GeneratorPool&) = delete;
			~GeneratorPool()
			{
				agx_engine_destroy(m_engine);
			}
			/* openings: n x AGX_OPENING_CAP uint16 ([0] = stone count, then Move::toShort words), cf. prepareOpening (utils/misc.cpp:142-170) */
			void begin(const std::vector<uint16_t> &openings, void *stream = nullptr)
			{
				if (openings.empty() || openings.size() % AGX_OPENING_CAP != 0)
					throw std::logic_error("GeneratorPool::begin: openings must hold a positive multiple of AGX_OPENING_CAP words");
				check(agx_engine_begin(m_engine, openings.data(), static_cast<int>(openings.size() / AGX_OPENING_CAP), stream));
				m_phase_started = m_skip_first_search_of_slice0 = false; // new games: every slice's cycle starts at its select stage again (generate())
			}
			/* OpeningGenerator::generate (OpeningGenerator.cpp:21-78): solver-unproven, network-balanced openings; before begin() */
			std::vector<uint16_t> generateOpenings(AGNetwork &network, int count, uint32_t seed = 0)
			{
				std::vector<uint16_t> out(static_cast<size_t>(count) * AGX_OPENING_CAP);
				check(agx_engine_generate_openings(m_engine, network.handle(), count, seed, out.data(), nullptr));
				return out;
			}
			/* Steps the pool as `slices` groups of games from now on, each on a stream that owns 1 / slices of the device's compute units
			 * (agx_stream_create_with_cu_mask), with the network's persistent grid narrowed to a slice: the slices run out of phase, the
			 * power-limited network launches never cover the whole chip at once (+10 % simulations/s on MI355X with 4 slices).  Returns the
			 * slice count in use (1 if the pool cannot be divided or the device offers no CU masks).  Self-play pools only. */
			int useChipSlices(AGNetwork &network, int slices, int instance = -1)
			{ // instance: which set of masked streams (they are cached per device, mask and instance): pools that run at the same time on one device
			  // must not share a queue — by default every call takes a set of its own
				static std::atomic<int> next_instance { 0 };
				if (instance < 0)
					instance = next_instance.fetch_add(1);
				m_slice_streams.clear();
				int cus = 0;
				if (m_match || slices <= 1 || slices > 16 || m_games % slices != 0 || agx_device_cu_count(&cus) != AGX_OK || cus < slices)
					return 1;
				const int per = cus / slices;
				for (int g = 0; g < slices; g++)
				{
					std::vector<uint32_t> mask((cus + 31) / 32, 0u);
					for (int c = g * per; c < (g + 1) * per; c++)
						mask[c / 32] |= 1u << (c % 32);
					void *s = nullptr;
					if (agx_stream_create_with_cu_mask_instance(&s, mask.data(), static_cast<int>(mask.size()), instance) != AGX_OK)
					{
						m_slice_streams.clear();
						return 1;
					}
					m_slice_streams.push_back(s);
				}
				check(agx_net_set_launch_width(network.handle(), per));
				return slices;
			}
		private:
			void run_stage(AGNetwork &network, int g, int which)
			{
				const int n = static_cast<int>(m_slice_streams.size());
				if (which == 0)
					check(agx_engine_select_solve_group(m_engine, g, n, m_slice_streams[g]));
				else if (which == 1)
					check(agx_engine_evaluate_group(m_engine, network.handle(), g, n, m_slice_streams[g]));
				else
					check(agx_engine_expand_backup_group(m_engine, g, n, m_slice_streams[g]));
			}
			static int slice_phase(int g) noexcept
			{ // stages of its NEXT cycle a slice has queued between two generate() calls
				static const int PHASES[4] = { 0, 1, 2, 1 };
				return PHASES[g % 4];
			}
			struct EventGuard
			{ // (an event must not outlive a stage that throws)
					void *event = nullptr;
					~EventGuard()
					{
						if (event != nullptr)
							agx_event_destroy(event);
					}
			};
		public:
			/* GameGenerator::generate for every game of the pool: one select -> solve -> evaluate -> expand/backup -> move step.
			 *
			 * A SLICED pool (setSlices) is left ROTATED between calls: slice g has the first slice_phase(g) = 0 / 1 / 2 / 1 stages of its next cycle
			 * queued already (slices 1 and 3 their search launch, slice 2 search + network), so a call that drains records, reads statistics or
			 * game_info, saves the games, or passes ANOTHER network sees a step boundary for slice 0 only — the network given to this call is not
			 * the one that evaluates slice 2's already-queued batch.  Callers that need step-aligned state (a network swap between training
			 * iterations, save_games, comparisons against a lock-step pool) call align() first; the next generate() re-enters the rotation. */
			void generate(AGNetwork &network, void *stream = nullptr)
			{
				if (m_slice_streams.empty())
				{
					check(agx_engine_step(m_engine, network.handle(), stream));
					if (m_pacers.empty())
						m_pacers.emplace_back(m_steps_ahead);
					m_pacers[0].step(stream);
				}
				else
				{ // The slices run a stage apart: slice g's cycle starts `phase` stages early (once, at the first call), after that every call runs one
				  // whole cycle per slice, rotated.  Slices that start together stay together while the host feeds them in step — four towers at
				  // once, the power-limited case the slicing exists to avoid (bench.py --stagger: 780 k -> 827 k simulations/s in a short window).
					const int n = static_cast<int>(m_slice_streams.size());
					if (!m_phase_started)
					{
						m_phase_started = true;
						// ... and on the DEVICE the odd slices begin when slice 0's first search launch is over (streams run their queues
						// independently: without this every slice starts its first search at the same moment, whatever the host's order)
						EventGuard first_search_done;
						for (int g = 0; g < n; g++)
						{
							if (g % 2 == 1 && first_search_done.event != nullptr)
								check(agx_stream_wait_event(m_slice_streams[g], first_search_done.event));
							for (int k = 0; k < slice_phase(g); k++)
								run_stage(network, g, k);
							if (g == 0 && n > 1)
							{
								check(agx_engine_select_solve_group(m_engine, 0, n, m_slice_streams[0])); // (slice 0's first cycle starts here; the loop below skips that stage once)
								check(agx_event_create(&first_search_done.event));
								check(agx_event_record(first_search_done.event, m_slice_streams[0]));
								m_skip_first_search_of_slice0 = true;
							}
						}
					}
					for (int g = 0; g < n; g++)
					{
						for (int k = 0; k < 3; k++)
						{
							const int which = (slice_phase(g) + k) % 3;
							if (g == 0 && which == 0 && m_skip_first_search_of_slice0)
							{
								m_skip_first_search_of_slice0 = false;
								continue;
							}
							run_stage(network, g, which);
						}
						while (static_cast<int>(m_pacers.size()) <= g)
							m_pacers.emplace_back(m_steps_ahead);
						m_pacers[g].step(m_slice_streams[g]); // the host stays two steps ahead of every slice and sleeps otherwise (HostPacer)
					}
				}
			}
			/* Brings every slice of a sliced pool to a step boundary: the stages of the cycles that generate() left begun are completed with
			 * `network` (no new cycle is started), so that records, statistics, save_games and a network swap see whole steps of every slice.  The
			 * next generate() starts the rotation again.  Nothing to do for an unsliced pool or before the first generate(). */
			void align(AGNetwork &network)
			{
				if (m_slice_streams.empty() || !m_phase_started)
					return;
				const int n = static_cast<int>(m_slice_streams.size());
				for (int g = 0; g < n; g++)
				{
					int begun = slice_phase(g);
					if (g == 0 && m_skip_first_search_of_slice0)
						begun = 1; // (only between the two halves of a first generate() that threw)
					for (int k = begun; begun > 0 && k < 3; k++)
						run_stage(network, g, k);
				}
				m_phase_started = false;
				m_skip_first_search_of_slice0 = false;
			}
			/* steps the host may run ahead of the device (0: unbounded — the launch loop then spins on a full queue); before the first generate() */
			void setHostStepsAhead(int steps) { m_steps_ahead = steps; }
			/* EvaluationGame::generate for every pair: the first players' trees with their network, then the second players' */
			void generate(AGNetwork &first, AGNetwork &second, void *stream = nullptr)
			{
				check(agx_engine_step_match(m_engine, first.handle(), second.handle(), stream));
			}
			/* per pair: games won / drawn / lost by the first player, games finished */
			std::vector<int> getMatchResults() const
			{
				std::vector<int> out(static_cast<size_t>(m_buffers.slots) * 4);
				check(agx_engine_match_results(m_engine, out.data(), static_cast<int>(out.size() / 4)));
				return out;
			}
			/* the three stages separately (Search::select+solve+scheduleToNN / NNEvaluator::evaluateGraph / generateEdges+expand+backup) */
			void selectSolveSchedule(void *stream = nullptr)
			{
				check(agx_engine_select_solve(m_engine, stream));
			}
			void evaluateGraph(AGNetwork &network, void *stream = nullptr)
			{
				check(agx_engine_evaluate(m_engine, network.handle(), stream));
			}
			void expandBackup(void *stream = nullptr)
			{
				check(agx_engine_expand_backup(m_engine, stream));
			}
			AgxEngineStats getStats() const
			{
				AgxEngineStats s;
				check(agx_engine_stats(m_engine, &s));
				return s;
			}
			/* Tree::getInfo({}) of one game */
			AgxGameInfo getInfo(int game, std::vector<AgxEdgeView> *root_edges = nullptr, std::vector<uint8_t> *board = nullptr) const
			{
				AgxGameInfo info;
				std::vector<AgxEdgeView> edges(400);
				std::vector<uint8_t> b(m_buffers.cells);
				check(agx_engine_game_info(m_engine, game, &info, b.data(), edges.data(), static_cast<int>(edges.size())));
				if (root_edges != nullptr)
					root_edges->assign(edges.begin(), edges.begin() + info.root_edges);
				if (board != nullptr)
					*board = b;
				return info;
			}
			/* samples produced so far (one per played move): what GameGenerator::make_move hands to GameDataStorage */
			void getRecords(std::vector<AgxMoveRecord> &records, std::vector<AgxEdgeView> &edges) const
			{
				int nr = 0, ne = 0;
				check(agx_engine_records(m_engine, nullptr, 0, nullptr, 0, &nr, &ne));
				records.resize(nr > 0 ? nr : 1);
				edges.resize(ne > 0 ? ne : 1);
				check(agx_engine_records(m_engine, records.data(), nr, edges.data(), ne, &nr, &ne));
				records.resize(nr);
				edges.resize(ne);
			}
			/* hand-over to the game buffer: returns the samples and empties the device-side pools */
			void drainRecords(std::vector<AgxMoveRecord> &records, std::vector<AgxEdgeView> &edges)
			{
				int nr = 0, ne = 0;
				check(agx_engine_records(m_engine, nullptr, 0, nullptr, 0, &nr, &ne));
				records.resize(nr > 0 ? nr : 1);
				edges.resize(ne > 0 ? ne : 1);
				check(agx_engine_drain_records(m_engine, records.data(), static_cast<int>(records.size()), edges.data(), static_cast<int>(edges.size()), &nr, &ne));
				records.resize(nr);
				edges.resize(ne);
			}
			void addOpenings(const std::vector<uint16_t> &openings)
			{
				if (openings.empty() || openings.size() % AGX_OPENING_CAP != 0)
					throw std::logic_error("GeneratorPool::addOpenings: openings must hold a positive multiple of AGX_OPENING_CAP words");
				check(agx_engine_add_openings(m_engine, openings.data(), static_cast<int>(openings.size() / AGX_OPENING_CAP)));
			}
			const AgxEngineBuffers& buffers() const noexcept
			{
				return m_buffers;
			}
			AgxEngine* handle() const noexcept
			{
				return m_engine;
			}
	};
}

#endif /* AGX_HPP_ */
