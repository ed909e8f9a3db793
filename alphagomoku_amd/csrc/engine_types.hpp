/*
 * engine_types.hpp — HBM-resident records of the device self-play engine (shared by host and device code).
 *
 * Layout philosophy: one flat arena per game, addressed by (game, index); no pointers inside records, so a whole pool of
 * thousands of games is a handful of hipMalloc'ed arrays and a game's tree can be compacted by a prefix sum + copy.
 *
 * Records mirror the reference's field semantics:
 *   DEdge  <- ag::Edge  (include/alphagomoku/search/monte_carlo/Edge.hpp:23-32, 24 bytes there and here)
 *   DNode  <- ag::Node + NodeCache::Entry{hash_key, CompressedBoard} (Node.hpp:24-42, NodeCache.hpp:51-71)
 *   DTask  <- ag::SearchTask (SearchTask.hpp:35-329)
 */
#ifndef AGX_ENGINE_TYPES_HPP_
#define AGX_ENGINE_TYPES_HPP_

#include <cstdint>

namespace agx
{
	constexpr int MAXN = 20;
	constexpr int MAXHW = MAXN * MAXN;
	constexpr int BWORDS = 13;      // 2 bits per cell, 32 cells per 64-bit word (NodeCache.hpp:56)
	constexpr int PATH_CAP = 256;
	constexpr int SPEC_PARK_POOL = 128;   // park buffers of a pool (a launch parks a handful of solves; none free = the solve simply runs on)
	constexpr int SPEC_PARK_WORDS = 2048 + 8; // 64-bit words per park buffer: the largest solver state (16 064 bytes at 20x20) + the solve's loop variables
	constexpr int MAX_FRAMES = 104; // alpha-beta recursion depth is bounded by the iterative-deepening limit (100 plies) + root
	constexpr int OPENING_CAP = 32;

	struct alignas(16) FrameBytes { uint32_t w[8]; }; // storage of one dev::Frame (dev_solver.hpp)

	struct DEdge
	{
			float prior;
			float win;
			float draw;
			int32_t visits;
			uint16_t move;    // Move::toShort (Move.hpp:144-147): sign | row << 2 | col << 9
			uint16_t score;   // Score raw bits
			uint16_t flag_vl; // bit 15 being expanded, bits 0-14 virtual loss
			uint16_t pad;
	};
	static_assert(sizeof(DEdge) == 24, "edge record must stay 24 bytes");

	struct DNode
	{
			int32_t edge_begin;
			float win;
			float draw;
			float moves_left;
			int32_t visits;
			uint16_t score;
			int16_t n_edges;
			int16_t depth;
			int16_t vl;
			uint8_t sign_to_move;
			uint8_t pad8;
			uint16_t flags; // root 2, fully expanded 4, statically solved 8, recursively solved 16, must defend 32 (Node.hpp:26-31)
			uint64_t hash;
			uint64_t cboard[BWORDS];
	};
	static_assert(sizeof(DNode) == 144, "node record layout");

	enum TaskFlags : uint32_t
	{
		TF_MUST_DEFEND = 1, TF_BY_NETWORK = 2, TF_BY_SOLVER = 4, TF_SKIP_EDGE_GENERATION = 8, TF_STATICALLY_SOLVED = 16, TF_RECURSIVELY_SOLVED = 32
	};

	struct DTask
	{
			int32_t path_len;
			int32_t final_node;
			int32_t n_edges;
			uint32_t flags;
			int32_t sign_to_move;
			uint32_t score;
			float win;
			float draw;
			float moves_left;
			int32_t needs_nn;
			int32_t symmetry; // input symmetry the position was handed to the network with (0 = identity)
			int32_t pad;
			uint64_t hash;
			uint64_t cboard[BWORDS];
			int32_t path_node[PATH_CAP];
			int32_t path_edge[PATH_CAP];
			uint8_t board[MAXHW];
			uint16_t emove[MAXHW];
			uint16_t escore[MAXHW];
	};

	struct GameState
	{
			int32_t active;
			int32_t sign_to_move;
			int32_t n_moves;
			int32_t outcome;
			int32_t root;
			int32_t n_nodes;
			int32_t n_edges;
			int32_t arena;
			int32_t n_tasks;
			int32_t need_move;
			int32_t generation;
			int32_t error;
			int32_t opening_id;
			int32_t games_done;
			int32_t solve_pos;     // first task of the batch that still has to be solved (solver launches may yield between tasks)
			int32_t solve_pending; // 1 while the batch is only partly solved: the game sits out select / network / expand until it is done
			int32_t nn_queued;     // positions handed to the network so far in this game (index of the symmetry hash)
			int32_t noise_ready;   // 1 once the root noise of the current move has been drawn (a fresh selector per move in the reference)
			int32_t my_sign;       // match mode: the colour this tree's player has in the current game (evaluation/EvaluationGame.cpp:59-71)
			int32_t restart_id;    // 0: playing; -1: the game is over and waits for an opening; k > 0: it starts again from opening k - 1 in k_restart
			int32_t match_score[4]; // match mode, first players' trees: games won / drawn / lost by the first player of this pair (+ pad)
			// the game's tree arenas live in pool-wide heaps (ArenaHeap): two node regions, two edge regions (search / compaction target)
			// and the node-cache table, all of size class `arena_class` (capacities = class-0 capacities << class)
			int32_t arena_class;
			int32_t grow_pending;  // 0 none; 1 k_expand found the arenas too small for this step's batch (nothing modified); 3 a larger bundle is
			                       // reserved, copy pending; 2 grown: the batch waits for expand (select / solve sit this step out); 4 the heap
			                       // had no bundle left: expand proceeds in the old arenas
			int32_t node_cap, edge_cap, ht_cap, grow_count;
			int32_t grow_owner, max_depth; // max_depth: Tree::max_depth (Tree.cpp:150,249: longest path of a select that reached a leaf since the last setBoard); grow_owner: double-buffered search: first record of the buffer whose expansion asked for the larger arenas (it goes first afterwards)
			uint64_t node_off[2], edge_off[2], ht_off;          // element offsets into EngineDev::nodes / edges / ht
			uint64_t new_node_off[2], new_edge_off[2], new_ht_off; // the bundle reserved by k_arena_service (grow_pending == 3)
			uint64_t root_hash;
			uint64_t cboard[BWORDS];
			unsigned long long prof[8];   // optional cycle counters (AGX_SOLVER_PROFILE builds only)
			unsigned long long dprof[24]; // finer stage stamps of the same builds (shader cycles)
			unsigned long long spec_stats[4]; // speculative solver: leaves solved against the pre-batch table, of which re-run serially, batches deferred, solves parked
			unsigned long long stats[12]; // nodes, nn, leaks, proven, wasted, solver nodes, select levels, select edges, moves, duplicates, max nodes, max edges
			uint8_t board[MAXHW];
			uint16_t moves[MAXHW];
	};

	/* One record per played move (what SearchDataPack(const Node&, board) keeps, dataset/data_packs.cpp:24-43). */
	struct MoveRecordHeader
	{
			int32_t game_serial; // opening id of the game this move belongs to
			int32_t move_number;
			uint16_t move;
			uint16_t root_score;
			int32_t root_visits;
			float root_win;
			float root_draw;
			int32_t n_edges;
			int32_t edge_offset; // into the record edge pool
			int32_t root_flags;  // bit 0 statically solved, 1 recursively solved, 2 must defend (SearchDataPack::flags, data_packs.cpp:40-42)
			int32_t game_slot;   // the pool slot (tree) that produced the sample
			int32_t game_index;  // how many games that slot had finished before this one: (slot, index) identifies a game
			int32_t sample_offset; // the sample in dataset format 201 (sample_v201.hpp) inside the sample byte pool, -1 = not recorded
			int32_t sample_bytes;
			int32_t outcome;     // GameOutcome after this move: 0 = the game goes on, else this was its last move
	};
	/* One record per finished game: what GameGenerator::generate hands over when Game::isOver (GameGenerator.cpp:104-114): the outcome
	 * and ALL moves of the game, opening included (Game::getMoves). */
	struct GameEndRecord
	{
			int32_t game_serial, game_slot, game_index;
			int32_t outcome;
			int32_t n_moves;
			uint16_t moves[MAXHW];
	};

	/* speculative solver (k_search_spec): what a task's solving wave leaves for the wave that commits the game's batch */
	constexpr int SPEC_OV_CAP = 256;    // overlay slots per task (== dev::OV_CAP)
	constexpr int SPEC_COUNTER0 = 64;   // counters[SPEC_COUNTER0 + 4 * group + {0 select cursor, 1 item cursor, 2 items queued, 3 games selected}]
	constexpr int SPEC_QUEUE_SLACK = 8192; // extra queue slots per group: item cursors run past the end by one per wave
	constexpr int N_COUNTERS = 192;
	struct SpecTask
	{
			int32_t solved;     // 1: solved speculatively in this launch (cleared by the commit)
			int32_t count;      // overlay slots in use
			int32_t overflow;   // the overlay was full: the result is void
			int32_t nodes;      // solver positions of the speculative run
			uint32_t flags0;    // the task's flags before the solve (a discarded run must not leave its own behind)
			float moves_left0;  // ... and its moves-left value: SearchTask::set does not reset it (SearchTask.cpp:32-50), a leaf that nobody evaluates
			                    // (a proven edge) backs up what the slot's previous occupant left there
			uint32_t pad[2];
			uint32_t dirty[SPEC_OV_CAP / 32];
			uint32_t keys[SPEC_OV_CAP]; // table bucket of every slot
	};

	constexpr int ARENA_CLASSES = 6; // bundle capacities: class-0 capacity << class, class < ARENA_CLASSES

	/* Bump allocators over the node / edge / table heaps + one free list of bundles per size class (k_arena_service, engine.hip). */
	struct ArenaBundle
	{
			uint64_t node_off[2], edge_off[2], ht_off;
	};
	struct ArenaHeap
	{
			uint64_t node_cursor, edge_cursor, ht_cursor; // first free element of each heap
			uint64_t node_total, edge_total, ht_total;
			int32_t free_count[ARENA_CLASSES];
			int32_t free_capacity;                        // bundles per class list
			int32_t grows, releases, failures, lock;
	};

	enum EngineError : int32_t
	{
		ERR_NONE = 0, ERR_NODE_CAPACITY = 1, ERR_EDGE_CAPACITY = 2, ERR_PATH_CAPACITY = 3, ERR_ACTION_STACK = 4, ERR_HASH_TABLE = 5, ERR_RECORDS = 6, ERR_FRAMES = 7, ERR_SPEC_STATE = 8,
		ERR_OVERLAY = 100 // internal to speculative solves (overlay full): never reported, the task is solved again serially
	};

	struct EngineDev
	{
			// configuration
			int rules, n, hw, draw_after;
			int n_games, batch, max_sims;
			int batch_limit; // Search::setBatchSize: leaves a select stage takes per game (<= batch, which stays the stride of the task buffers)
			float c_puct, c_scale;
			int init_to;
			float leak_threshold, expansion_threshold;
			int max_children;
			float policy_temperature; // MCTSConfig::policy_temperature (initialize_edges, EdgeGenerator.cpp:88-127)
			int tss_max_nodes, tss_max_depth;
			unsigned long long solve_time_ticks; // 0 = no time limit; else Search::solve(endTime >= 0) (Search.cpp:159-183): the launch's budget in 100 MHz wall-clock ticks
			unsigned long long zobrist_seed;
			unsigned long long tt_bucket_mask; // buckets - 1 (4 entries of 16 bytes per bucket)
			int node_cap, edge_cap, ht_cap, act_cap;
			int record_cap, record_edge_cap;
			int record_format;   // bit 0: root-edge snapshots (24-byte edges), bit 1: samples in dataset format 201 (6 bytes per entry)
			int sample_cap;      // bytes of the format-201 sample pool
			int game_end_cap;    // finished-game records
			int n_openings;
			float yield_fraction; // 0 = never; else a game yields between two solves once this fraction of the launch's games is done
			int yield_counter;    // index into counters[] of the launch's "games done" count
			int final_selector, use_symmetries;
			unsigned long long symmetry_seed;
			int noise_type;
			float noise_weight;
			unsigned long long noise_seed;
			float *noise;          // [game][hw] noisy root priors of the current move
			int g0;          // first game handled by this launch (a launch covers games [g0, g0 + gridDim.x): one "group" of the pool)
			int nn_counter;  // index into counters[] of this group's scheduled-position count
			// state
			GameState *games;
			DNode *nodes;   // node heap: a game's two regions at GameState::node_off[0/1], node_cap << class records each
			DEdge *edges;   // edge heap (GameState::edge_off)
			int *ht;        // node-cache tables (GameState::ht_off): node index + 1, 0 = empty
			ArenaHeap *heap;
			ArenaBundle *free_bundles; // [ARENA_CLASSES][heap->free_capacity]
			DTask *tasks;   // [game][batch]
			uint32_t *act;  // [game][act_cap] alpha-beta action stack: move | score << 16
			uint16_t *list_spill;       // [game][2][10][hw] tails of the solver's threat lists beyond their LDS capacity (dev_solver.hpp)
			FrameBytes *frame_spill;    // [game][MAX_FRAMES] alpha-beta frames (dev::Frame, 32 bytes) beyond the LDS-resident ones
			uint64_t *snap_spill;       // [game][hw + 2][64] undo snapshots of the solver's pattern update (dev_solver.hpp: solver_update_around)
			uint64_t *tt;   // [game][buckets][4][2]
			// read-only tables
			const uint8_t *t_pattern;
			const uint8_t *t_ho3;
			const uint8_t *t_threat;
			const uint8_t *t_threat_packed; // [4096] cross | circle << 4
			const uint16_t *t_defense;
			const uint64_t *nc_keys; // node-cache Zobrist keys [3 + 3*hw]
			const uint64_t *zob;     // solver Zobrist keys [2*hw][2] (lo, hi)
			const uint16_t *openings; // [n_openings][OPENING_CAP] (count in slot 0)
			// evaluation exchange
			uint32_t *nn_features; // [game*batch][hw]
			float *nn_policy;      // [game*batch][hw]
			float *nn_value;       // [game*batch][3]
			float *nn_q;           // [game*batch][hw][2] action values (win, draw) per cell, 'pvq' networks only
			int has_q;
			int match_merged; // this launch covers both players' trees: network slot lists by half of the pool, not by launch
			int prune_root;  // the root is pruned like any node (UnifiedGenerator without forceExpandRoot: evaluation players)
			int match_mode;  // evaluation matches: tree g (first player) and tree g + n_games / 2 (second player) share one game
			int shared_tree; // tournament search: the n_games records are the search threads of ONE tree (game 0): own task buffer and solver each
			int search_buffers; // 2: double-buffered tournament search (Search::useBuffer / switchBuffer, Search.cpp:243-252): record b * threads + t is
			                    // buffer b of search thread t; a group (= one launch) is one buffer of every thread
			int tt_mod;      // solver table of record g = table g % tt_mod (n_games, or the thread count when a thread has two buffers)
			int grp_first, grp_count; // the records of this launch's group (kept when g0 is redirected to the tree's record)
			// speculative solver
			int spec_on;           // the pool has speculative state (spec_tasks / spec_overlay per task, park buffers): AgxEngineConfig.speculative_solver took effect
			int spec_group;        // index of this launch's group (its queue segment and counters)
			int *spec_watchdog;            // [16] what a wave that gave up waiting for a queue slot saw
			unsigned long long *spec_trace; // AGX_SPEC_PROFILE builds: [game][4] time stamps of the last launch
			unsigned long long *spec_prof; // AGX_SPEC_PROFILE builds: time sums of k_search_spec
			int spec_waves;        // waves of this launch (its spill areas start at area n_games + spec_group * spec_waves)
			int *spec_items;       // [n_games * batch + 16 * SPEC_QUEUE_SLACK] work queue: (game * 16 + task) + 1, 0 = empty
			int *spec_left;        // [game] tasks of the batch still being solved
			SpecTask *spec_tasks;  // [game * batch]
			uint64_t *spec_overlay; // [game * batch][SPEC_OV_CAP][16]
			// parked solves (engine.hip: "Parking"): a speculative solve that is still running when all but a few games of the launch are done is set
			// aside — its LDS state in park_lds, its HBM tails in a spill area of its own — and taken up again by the next launch
			int *park_slot;        // [game * batch] park buffer + 1 of a task whose solve is parked, 0 = none
			int *park_owner;       // [SPEC_PARK_POOL] task slot + 1 that holds the buffer, 0 = free
			int *parts_done;       // [2][n_games] workgroups of k_arena_copy / k_clear_tables that have finished their part of a game (the last one commits / restarts it)
			uint64_t *park_lds;    // [SPEC_PARK_POOL][SPEC_PARK_WORDS] the solver's LDS state, then 8 words of the solve's loop variables
			int park_area0;        // spill area of park buffer 0 (behind the games' and the waves' areas)
			float park_fraction;   // solves are parked once this fraction of the launch's games is done (0 = never)
			int *nn_list;          // compacted slots to evaluate
			int *counters;         // [0] (unused), [1] next opening, [2] finished games, [3] records used, [4] record edges used, [5] total moves, [6] sample bytes used, [7] game-end records used, [16 + group] positions scheduled for the network by that group, [32 + group] games of that group done with their solver batch
			// output records
			MoveRecordHeader *records;
			DEdge *record_edges;
			uint8_t *samples;      // format-201 bytes, one 4-byte aligned block per recorded move
			GameEndRecord *game_ends;
	};
}

#endif
