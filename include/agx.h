/*
 * agx.h — C ABI of the MI355X-native self-play MCTS + policy/value evaluation engine.
 *
 * This is the drop-in boundary for the hot path of AlphaGomoku's src/search + src/selfplay +
 * src/networks. The reference has no FFI layer on this path (its boundary is a set of C++ classes over
 * the MinML C++ API), so the entry points below are what a cgo/ctypes/C++ facade for that path binds.
 * The only extern "C" surface in the reference is the dataset reader
 * include/alphagomoku/dataset/torch_api.h:13-43; its conventions are kept: plain structs, caller-owned
 * flat buffers, sizes queried first.  Unlike torch_api.h every call returns an int status (0 = ok) and
 * agx_last_error() returns the message — the C++ facade turns a non-zero status into the same
 * std::logic_error / std::runtime_error the reference throws (NNEvaluator.cpp:149,185-187).
 *
 * No torch types, no C++ types, no HIP types appear in the signatures.  Pointers named d_* are device
 * (HBM) addresses, h_* are host addresses, `stream` is a hipStream_t passed as void* (NULL = default).
 *
 * One handle per GPU, each driven by exactly one host thread (mirrors GeneratorThread,
 * src/selfplay/GeneratorManager.cpp:124-141).
 */
#ifndef AGX_H_
#define AGX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGX_OK 0
#define AGX_ERR_INVALID 1
#define AGX_ERR_HIP 2
#define AGX_ERR_UNSUPPORTED 3
#define AGX_ERR_STATE 4

/* Game rules, same numbering as ag::GameRules (include/alphagomoku/game/rules.hpp:18-25). */
enum AgxRules { AGX_FREESTYLE = 0, AGX_STANDARD = 1, AGX_RENJU = 2, AGX_CARO5 = 3, AGX_CARO6 = 4 };

/* Signs, same numbering as ag::Sign (include/alphagomoku/game/Move.hpp:17-23). */
enum AgxSign { AGX_NONE = 0, AGX_CROSS = 1, AGX_CIRCLE = 2, AGX_ILLEGAL = 3 };

const char* agx_last_error(void);
int agx_version(void);
/* sha256[:16] of the sources this library was compiled from (csrc + this header; alphagomoku_amd/build.py:source_hash) */
const char* agx_build_hash(void);
/* Selects the HIP device for the calling thread (one engine per GPU). */
int agx_set_device(int device);
int agx_device_count(int* count);
int agx_device_cu_count(int* count); /* compute units of the current device */

/* ------------------------------------------------------------------------------------------------
 * Policy/value network (replaces AGNetwork::forward / asyncForwardLaunch+Join over ml::Graph,
 * src/networks/AGNetwork.cpp:61-88, for the ResnetPV architecture of src/networks/networks.cpp:71-93
 * built from src/networks/blocks.cpp:32-38,45-55,99-118 after optimize(2) has folded the BNs).
 * ---------------------------------------------------------------------------------------------- */
typedef struct AgxNetDesc
{
	int rows;           /* board rows (15 or 20) */
	int cols;           /* board cols */
	int blocks;         /* residual blocks */
	int filters;        /* conv filters F (64 or 128) */
	int in_channels;    /* 32 (bit-packed features, NNInputFeatures.cpp:59-113) or 8 (ResnetPVraw, networks.cpp:107-129: ml::unpackInput
	                       expands the 8 low bits of the feature word, AGNetwork.cpp:249-258; without the action-values head) */
	int value_hidden;   /* D = min(256, 2F) (blocks.cpp:113) */
	int action_values;  /* 0: ResnetPV (outputs "pv", networks.cpp:71-93); 1: ResnetPVQ (:143-168) with the action-values head 'q' */
} AgxNetDesc;

typedef struct AgxNet AgxNet; /* opaque */

/* Number of fp32 values in the canonical (BN-folded) weight blob, in this order:
 *   conv_in  W[5][5][Cin][F]  b[F]
 *   blocks x { W1[3][3][F][F] b1[F]  W2[3][3][F][F] b2[F] }
 *   policy   Wp1[3][3][F][F] bp1[F]  Wp2[F] bp2[1]
 *   value    Wv1[F][4] bv1[4]  Wv2[rows*cols*4][D] bv2[D]  Wv3[D][3] bv3[3]
 *   action values (only with desc.action_values; createActionValuesHead, blocks.cpp:119-127):
 *            Wq1[3][3][F][F] bq1[F] (tanh)  Wq2[F][3] bq2[3] (softmax over the 3 outputs of every cell)
 * Conv weights are [kh][kw][cin][cout] (cross-correlation, "same" zero padding); the value-head flatten
 * order is NHWC row-major: index = (row*cols + col)*4 + c.  */
size_t agx_net_blob_floats(const AgxNetDesc* desc);
int agx_net_create(const AgxNetDesc* desc, AgxNet** out);
int agx_net_load_weights(AgxNet* net, const float* h_blob, size_t n_floats);
/* d_features: uint32[batch][rows*cols] (one bit-packed word per cell);
 * d_policy: float[batch][rows*cols] (softmax over the board); d_value: float[batch][3] = (win, draw, loss). */
int agx_nn_forward(AgxNet* net, const uint32_t* d_features, int batch, float* d_policy, float* d_value, void* stream);
/* ResnetPVQ: additionally d_action_values float[batch][rows*cols][2] = (win, draw) of the per-cell softmax-3 'q' output, the part
 * NetworkDataPack::unpackActionValues keeps (NetworkDataPack.cpp:214-224).  Passing NULL skips the head. */
int agx_nn_forward_pvq(AgxNet* net, const uint32_t* d_features, int batch, float* d_policy, float* d_value, float* d_action_values, void* stream);
int agx_net_description(const AgxNet* net, AgxNetDesc* out);
/* Caps the persistent grid of the tower kernel at `workgroups` CUs (0 = all of them).  A tower workgroup fills its CU (LDS, registers), so
 * a narrowed launch leaves whole CUs to kernels running at the same time on other streams: the other slices of a pool stepped as
 * pipelined groups (agx_engine_*_group). */
int agx_net_set_launch_width(AgxNet* net, int workgroups);
int agx_net_destroy(AgxNet* net);

/* Same network, but the batch is a device-side list: position i is slot d_slot_list[i] of the slot-indexed buffers
 * (features uint32[slots][cells], policy float[slots][cells], value float[slots][3]); the batch size is read from
 * *d_count on the device, so no host synchronisation is needed between the search kernels and the network
 * (replaces NNEvaluator::pack_to_network / asyncEvaluateGraphLaunch, src/search/monte_carlo/NNEvaluator.cpp:182-262). */
int agx_nn_forward_indirect(AgxNet* net, const uint32_t* d_features, const int* d_slot_list, const int* d_count, int max_batch,
		float* d_policy, float* d_value, void* stream);
int agx_nn_forward_indirect_pvq(AgxNet* net, const uint32_t* d_features, const int* d_slot_list, const int* d_count, int max_batch,
		float* d_policy, float* d_value, float* d_action_values, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Device-resident self-play engine: a pool of independent games, each with its own search tree, solver transposition
 * table and task buffer in HBM.  Replaces, for one GeneratorThread (src/selfplay/GeneratorManager.cpp:124-141), the
 * per-game objects GameGenerator{Game, Tree, Search{AlphaBetaSearch}} and their calls
 *   Search::select / solve / scheduleToNN / generateEdges / expand / backup / cleanup   (Search.hpp:78-90)
 *   Tree::setBoard / select / expand / backup / getInfo                                   (Tree.hpp:68-104)
 *   GameGenerator::generate / make_move / prepare_search                                  (GameGenerator.cpp:46-185)
 * Configuration fields carry the reference's names (utils/configs.hpp: GameConfig, TreeConfig, EdgeSelectorConfig,
 * MCTSConfig, TSSConfig, SearchConfig, SelfplayConfig::constraints.max_simulations).
 * ---------------------------------------------------------------------------------------------- */
typedef struct AgxEngineConfig
{
	int rules;                        /* AgxRules (all five rule sets run on the device) */
	int board_size;                   /* square boards, <= 20 */
	int draw_after;                   /* GameConfig::draw_after, <= 0 means rows*cols */
	int n_games;                      /* games resident on this GPU (SelfplayConfig::games_per_thread) */
	int max_batch_size;               /* SearchConfig::max_batch_size: simulations selected per game per step */
	int max_simulations;              /* SelfplayConfig::constraints.max_simulations (playouts per move).  Keep it >= 50: with a drawish root the
	                                     move rule asks for max - clamp((draw - 0.75) / 0.25) * (max - 50) visits (utils/misc.cpp:171-179), which
	                                     EXCEEDS max below 50 while select stops at max — such a game never moves, here as in the reference. */
	float exploration_constant;       /* EdgeSelectorConfig */
	float exploration_scaling;
	int init_to;                      /* 0 "q_head", 1 "parent", 2 "draw", 3 "loss" */
	float information_leak_threshold; /* TreeConfig */
	float policy_expansion_threshold; /* MCTSConfig */
	int tss_max_positions;            /* TSSConfig::max_positions, <= 1000 (the search depth is capped at 100 plies either way) */
	uint64_t tss_table_entries;       /* AlphaBetaSearch's SharedHashTable size per game (reference: 4 Mi) */
	uint64_t zobrist_seed;            /* seed of the solver / node-cache Zobrist keys (the reference draws them from a time-seeded RNG) */
	int node_capacity;                /* per game, per arena, size class 0 (TreeConfig::node_bucket_size analogue; see arena_reserve) */
	int edge_capacity;                /* per game, per arena, size class 0 (TreeConfig::edge_bucket_size analogue).  The arenas grow on demand
	                                     like the reference's pools (arena_reserve below) */
	int record_capacity;              /* move records kept on the device, 0 = n_games * cells */
	int record_edge_capacity;         /* root-edge snapshots kept on the device, 0 = 64 per record */
	float solver_yield_fraction;      /* 0 = off.  A pool step lasts as long as its slowest game's solver batch; with f in (0,1] a game
	                                     whose batch is only partly solved when a fraction f of the launch's games are done sits out the
	                                     rest of this step (network / expand) and resumes in the next one.  Per-game results are
	                                     unchanged (every game executes the same sequence of operations), only the pacing differs. */
	int final_selector;               /* GameGenerator::make_move's EdgeSelector (SelfplayConfig::final_selector.policy): 0 "best",
	                                     1 "max_visit", 2 "min_visit", 3 "max_value", 4 "max_policy", 5 "lcb" with exploration_constant
	                                     (EdgeSelector.cpp:446-536) */
	int use_symmetries;               /* NNEvaluator::useSymmetries (NNEvaluator.cpp:134-141,244-286): every position handed to the network
	                                     is augmented by one of the 8 board symmetries and the policy is mapped back.  The reference draws
	                                     randInt(8) from a time-seeded generator; here the k-th position of game `serial` uses
	                                     splitmix64(symmetry_seed ^ serial << 32 ^ k) >> 61, so runs are reproducible. */
	uint64_t symmetry_seed;
	int max_children;                 /* MCTSConfig::max_children: non-root nodes keep at most this many edges (the best by proven score, then
	                                     prior) and of those only the ones with prior >= policy_expansion_threshold * (their prior sum)
	                                     (prune_weak_moves, EdgeGenerator.cpp:49-86).  0 = unlimited, the reference default. */
	int noise_type;                   /* EdgeSelectorConfig::noise_type at the root: 0 "none", 1 "custom", 2 "dirichlet" (alpha 0.05), 3 "gumbel"
	                                     (create*Noise, utils/random.cpp:89-124; mixed into the priors as EdgeSelector.cpp:602-623 does).
	                                     Drawn once per move by the first select that sees an expanded root, like the selector that
	                                     prepare_search creates per move.  Stream: counter-based hash of (noise_seed, game serial, move
	                                     number) — the reference uses a time-seeded mt19937; log / exp are fixed double-precision series
	                                     (csrc/root_noise.hpp), so runs are reproducible and identical on host and device. */
	float noise_weight;
	uint64_t noise_seed;
	int action_values;                /* 1: the network is a 'pvq' network (AGNetwork::getOutputConfig): agx_engine_evaluate runs its
	                                     action-values head and new edges start from those values (initialize_edges, EdgeGenerator.cpp:
	                                     119-124) — what the default init_to = "q_head" selector reads.  0: 'pv' network, edges start
	                                     at (0, 0) exactly as the reference's zero-filled 'q' tensor gives (NetworkDataPack.cpp:122-126). */
	int match_mode;                   /* 1: evaluation matches (evaluation/EvaluationGame.cpp, evaluation/Player.cpp).  n_games must be even:
	                                     tree g < n_games/2 belongs to the first player of match g, tree g + n_games/2 to the second player;
	                                     both see one game and a tree searches only on its own player's turns (Player::setBoard jumps two
	                                     plies; trees persist across games, the solver table is cleared per game; the root is pruned like
	                                     any other node, UnifiedGenerator's forceExpandRoot = false).  Every opening is played twice with
	                                     the colours swapped.  Drive it with agx_engine_step_match(e, first_net, second_net, s), or with two
	                                     groups on ONE stream: agx_engine_step_group(e, first_net, 0, 2, s); ..._group(e, second_net, 1, 2, s). */
	float policy_temperature;         /* MCTSConfig::policy_temperature (initialize_edges, EdgeGenerator.cpp:88-127): 1 = priors are the policy
	                                     (reference default); 0 = prior 1 for the cells that hold the policy maximum, 0 elsewhere; otherwise
	                                     policy^(1/T), evaluated as exp(log(p)/T) with the fixed double-precision series of csrc/root_noise.hpp
	                                     (the reference calls std::pow), so device and oracle agree bit for bit. */
	float arena_reserve;              /* the tree arenas are regions of pool-wide heaps sized n_games x (class-0 bundle) x (1 + arena_reserve).
	                                     node_capacity / edge_capacity are the CLASS-0 sizes every game starts with; a game whose batch would
	                                     not fit moves into a bundle of twice the size from the reserve (NodeCache::resize x2, NodeCache.cpp:
	                                     320-355; ObjectPool growth, utils/ObjectPool.hpp:74-289) — it sits out one expand stage, its results do
	                                     not change — and hands it back when its game is over.  Only with the reserve exhausted (or beyond 32 x
	                                     the class-0 size) does an overflowing tree stop its game (agx_engine_stats.first_error).  < 0 = default 1.0 */
	int search_threads;               /* > 1: tournament search (player/SearchThread.cpp:121-180): the pool is ONE game tree searched by
	                                     `search_threads` SearchThreads, each with its own Search — own task buffer of max_batch_size leaves, own
	                                     threat solver and table — and n_games must equal search_threads (record 0 owns the tree, the others are the
	                                     threads).  Per step the threads take the tree in thread order for select (virtual loss makes them spread),
	                                     solve their batches in parallel (one wave each), share one network launch, then take the tree in order
	                                     again for expand + backup: the lock-step interleaving of SearchThread::serial_run.  0 / 1 = self-play pool. */
	int record_format;                /* what k_advance keeps of every played move's root (SearchDataPack, dataset/data_packs.cpp:24-43):
	                                     bit 0 (value 1, the default): the root edges as 24-byte AgxEdgeView snapshots;
	                                     bit 1 (value 2): the sample quantised on the device to dataset format 201
	                                     (SearchDataStorage_v201::loadFrom + serialize, dataset/SearchDataStorage.cpp:326-374,410-419:
	                                     16-byte header + 6 bytes per visited / proven cell) — a quarter of the bytes to keep and copy;
	                                     3 = both.  Finished games are always reported (AgxGameEnd). */
	int record_sample_capacity;       /* bytes of the format-201 sample pool, 0 = room for an entry on every cell of every record (capped at 2 GiB) */
	int game_end_capacity;            /* finished-game records kept on the device, 0 = max(2 * n_games, record_capacity / 16) */
	int speculative_solver;           /* 1: agx_engine_select_solve* runs select + threat solver as ONE persistent launch (k_search_spec) in which the
	                                     leaves of a game's batch are solved in PARALLEL, each wave against the game's transposition table as it was
	                                     before the batch (every touched bucket copied into a per-task overlay), and committed in batch order; a task
	                                     that saw a bucket an earlier task of the batch changed is solved again serially — results are bit-identical
	                                     to the serial order of Search::solve (Search.cpp:159-183), the solver runs with 3-4 waves per SIMD instead of
	                                     one wave per game.  0 (default): one wave per game, tasks in order.  Tournament-search pools (search_threads > 1)
	                                     take the speculative launch too; ignored for solver budgets above 250 positions (the overlay
	                                     holds 256 buckets) and batches above 16. */
	int speculative_waves;            /* waves of that launch over the whole pool, 0 = as many as stay resident: 16 per compute unit of the device on 15x15
	                                     boards, 10 on 20x20 (agx_engine_speculative_waves reports the number in use) */
	int force_expand_root;            /* UnifiedGenerator's forceExpandRoot (EdgeGenerator.cpp:283-285): 1 (default, self-play: GameGenerator.cpp:183-184)
	                                     never prunes the root's edges; 0 prunes the root like any node (evaluation Player, Player.cpp:109; match_mode
	                                     engines always do) */
	int search_buffers;               /* 2: the double-buffered tournament search of SearchThread::asynchronous_run (player/SearchThread.cpp:148-180,
	                                     Search::useBuffer / switchBuffer, Search.cpp:243-252): every search thread has TWO task buffers, one solver
	                                     and table; n_games must equal 2 x max(search_threads, 1) and record b * threads + t is buffer b of thread t.
	                                     The pool is stepped buffer by buffer as group b of 2 (agx_engine_expand_backup_group, then
	                                     agx_engine_select_solve_group, then the network on that group): the leaves of buffer b stay in the network
	                                     — virtual losses applied — while buffer 1 - b is expanded, backed up, selected and solved, which is the
	                                     overlap of tower and tree work inside ONE game that the reference gets from asyncEvaluateGraphLaunch / Join.
	                                     When the move rule fires after a buffer's backup, the other buffer's leaves are dropped and their virtual
	                                     losses taken back (Search::cleanup).  Also valid with search_threads 0 / 1 (one thread, two buffers — the
	                                     reference's "1 thread per GPU" setting).  0 / 1 (default): one buffer. */
} AgxEngineConfig;

typedef struct AgxEngine AgxEngine; /* opaque */

typedef struct AgxEngineBuffers
{
	uint32_t* d_nn_features; /* [slots][cells] */
	float* d_nn_policy;      /* [slots][cells] */
	float* d_nn_value;       /* [slots][3] */
	float* d_nn_action_values; /* [slots][cells][2] = (win, draw) per cell; read only when AgxEngineConfig.action_values is set */
	int* d_nn_list;          /* slots scheduled for evaluation by the last agx_engine_select_solve */
	int* d_nn_count;         /* number of entries of d_nn_list */
	int slots;
	int cells;
} AgxEngineBuffers;

typedef struct AgxEngineStats
{ /* SearchStats counters (Search.hpp:33-54) summed over the pool + pool-level counters */
	unsigned long long evaluated_nodes;       /* nb_node_count: simulations backed up */
	unsigned long long network_evaluations;   /* nb_network_evaluations */
	unsigned long long information_leaks;     /* nb_information_leaks */
	unsigned long long proven_edge_visits;    /* nb_proven_states */
	unsigned long long wasted_expansions;     /* nb_wasted_expansions */
	unsigned long long duplicate_selections;  /* nb_duplicate_nodes */
	unsigned long long solver_nodes;          /* positions visited by the alpha-beta solver */
	unsigned long long select_levels;         /* tree levels descended by select (incl. dropped leak descents) */
	unsigned long long select_edge_reads;     /* edges scanned by PUCT argmax */
	unsigned long long moves_played;
	unsigned long long peak_nodes;            /* largest per-game node / edge arena use seen */
	unsigned long long peak_edges;
	int games_finished;
	int openings_taken;
	int active_games;
	int records_used;
	int record_edges_used;
	int first_error;                          /* 0 = none; EngineError of the first game that stopped */
	int arena_grows;                          /* bundles handed out by the arena heap to games that outgrew theirs */
	int arena_releases;                       /* grown bundles handed back by finished games */
	int arena_failures;                       /* growth requests the heap could not serve (reserve exhausted) */
	int arena_max_class;                      /* largest size class in use: capacities = class-0 capacities << class */
	float arena_heap_used;                    /* high-water mark of the edge heap, fraction of its size */
	int speculative_parks;                    /* speculative solves set aside at the end of a launch and taken up again by the next one (engine: "Parking") */
	unsigned long long speculative_solves;    /* speculative_solver: leaves solved against the pre-batch table ... */
	unsigned long long speculative_reruns;    /* ... and how many of them had to be solved again serially (conflict or full overlay) */
	unsigned long long speculative_deferrals; /* batches whose commit was put off to the next launch (solver_yield_fraction) */
} AgxEngineStats;

typedef struct AgxEdgeView
{ /* ag::Edge (Edge.hpp:23-32) */
	float prior, win, draw;
	int32_t visits;
	uint16_t move;   /* Move::toShort: sign | row << 2 | col << 9 */
	uint16_t score;  /* Score raw bits */
	uint16_t flag_and_virtual_loss;
	uint16_t reserved;
} AgxEdgeView;

typedef struct AgxGameInfo
{
	int active, sign_to_move, n_moves, outcome, error, opening_id, games_done, n_nodes, n_edges;
	int root_visits;
	float root_win, root_draw;
	int root_score;
	int root_edges;
	int grow_pending; /* non-zero: the game's last batch waits for its expansion in larger arenas (one step later than a lock-step pool) */
	int arena_class;  /* its tree arenas hold class-0 capacity << arena_class records */
	float root_moves_left; /* Tree::getMovesLeft (Tree.cpp:173-176, :350): the root's running mean of the moves-left estimates backed up through it */
	int max_depth;         /* Tree::getMaximumDepth (Tree.cpp:188-191): longest select path that reached a leaf since the last setBoard */
} AgxGameInfo;

typedef struct AgxMoveRecord
{ /* one self-play sample: SearchDataPack(const Node&, board) (dataset/data_packs.cpp:24-43) */
	int game_serial, move_number;
	uint16_t move, root_score;
	int root_visits;
	float root_win, root_draw;
	int n_edges, edge_offset;
	int root_flags; /* SearchDataPack::flags: bit 0 root statically solved, 1 recursively solved, 2 must defend (data_packs.cpp:40-42) */
	int game_slot;      /* the pool slot (tree) that produced the sample */
	int game_index;     /* games that slot had finished before this one: (game_slot, game_index) identifies a game; game_serial is the
	                       opening id, which the two colour-swapped games of a match share */
	int sample_offset;  /* record_format bit 1: the format-201 bytes of this sample inside the sample pool (-1 = not recorded) */
	int sample_bytes;
	int outcome;        /* GameOutcome after this move (0 unknown / 1 draw / 2 cross win / 3 circle win): non-zero on a game's last move */
} AgxMoveRecord;

typedef struct AgxGameEnd
{ /* one finished game: what GameGenerator::generate passes on when Game::isOver (GameGenerator.cpp:104-114) */
	int game_serial, game_slot, game_index;
	int outcome;         /* GameOutcome (game/rules.hpp:29-35) */
	int n_moves;         /* Game::getMoves(): opening stones included */
	uint16_t moves[400]; /* Move::toShort */
} AgxGameEnd;

typedef struct AgxSavedGame
{ /* one game in flight, as GameGenerator::save keeps it (selfplay/GameGenerator.cpp:122-130: Game::serialize + the samples so far — the samples
     stay with the AgxGameBuffer, see agx_game_buffer_take_pending): search trees are not checkpointed */
	int game_slot, game_index;   /* the pool slot and the number of games it had finished (the key of the game's pending samples) */
	int opening_id, sign_to_move, nn_queued, n_moves;
	uint16_t moves[400];         /* Move::toShort, opening stones included */
} AgxSavedGame;

typedef struct AgxRecordCounts
{
	int records, edges, sample_bytes, game_ends;
} AgxRecordCounts;

#define AGX_OPENING_CAP 32 /* uint16 per opening: [0] = number of stones, [1..] = Move::toShort */

/* Host-only: one synthetic random opening with the distribution of the reference's prepareOpening (utils/misc.cpp:142-170);
 * h_opening receives AGX_OPENING_CAP words ([0] = stones, then Move::toShort, cross first). */
int agx_make_opening(int rules, int board_size, uint32_t seed, uint16_t* h_opening);
/* Host-only: getOutcome (src/game/rules.cpp:110-133) after the stone `sign` was put on (row, col) of h_board (board_size^2
 * bytes, 0 empty / 1 cross / 2 circle, the stone already on it): 0 unknown, 1 draw, 2 cross win, 3 circle win.  Under renju a
 * foul of cross (isForbidden, rules.cpp:134-173) is a circle win.  draw_after <= 0 means "full board".  The same test runs on
 * the device after every played move; this entry point is what openings and host-side callers use. */
int agx_get_outcome(int rules, int board_size, const uint8_t* h_board, int sign, int row, int col, int draw_after, int* outcome);

int agx_engine_default_config(AgxEngineConfig* cfg);
int agx_engine_create(const AgxEngineConfig* cfg, AgxEngine** out);
int agx_engine_destroy(AgxEngine* engine);
/* OpeningGenerator::generate (selfplay/OpeningGenerator.cpp:21-78) for `count` openings, batched on the device: candidates from
 * agx_make_opening(seed, seed + 1, ...) that the threat solver cannot prove within 1000 nodes are evaluated by `net` and accepted
 * when |expectation - 0.5| < 0.1 + 0.01 * trials.  h_openings receives count x AGX_OPENING_CAP words (the layout agx_engine_begin
 * takes); h_stats (optional, 4 ints): candidates drawn, proven by the solver, rejected as unbalanced, network evaluations.
 * Borrows the pool's task slots: only valid before agx_engine_begin. */
int agx_engine_generate_openings(AgxEngine* engine, AgxNet* net, int count, uint32_t seed, uint16_t* h_openings, int* h_stats);
/* Starts every game of the pool from h_openings[n_openings][AGX_OPENING_CAP]; finished games take the next unused opening. */
int agx_engine_begin(AgxEngine* engine, const uint16_t* h_openings, int n_openings, void* stream);
/* One pool step = select_solve -> evaluate -> expand_backup.  The three stages are exposed separately so that a caller
 * (or a test) can run any evaluator over AgxEngineBuffers between them. */
int agx_engine_select_solve(AgxEngine* engine, void* stream);
int agx_engine_evaluate(AgxEngine* engine, AgxNet* net, void* stream);
int agx_engine_expand_backup(AgxEngine* engine, void* stream);
int agx_engine_step(AgxEngine* engine, AgxNet* net, void* stream);
/* The same stages restricted to group `group` of `n_groups` equal slices of the pool.  Games are independent, so slices can be
 * driven from different streams and drift apart: while one slice waits for its slowest solver wave, the others use the CUs
 * (at most 16 groups).  Results per game do not depend on the grouping.  A network may be shared by the slices: the single-plane
 * kernels keep their scratch per stream. */
int agx_engine_select_solve_group(AgxEngine* engine, int group, int n_groups, void* stream);
/* the two halves of select_solve separately: Search::select, then Search::solve + scheduleToNN (Search.hpp:78-82) */
int agx_engine_select_group(AgxEngine* engine, int group, int n_groups, void* stream);
int agx_engine_solve_group(AgxEngine* engine, int group, int n_groups, void* stream);
int agx_engine_evaluate_group(AgxEngine* engine, AgxNet* net, int group, int n_groups, void* stream);
int agx_engine_expand_backup_group(AgxEngine* engine, int group, int n_groups, void* stream);
/* the two halves of expand_backup separately: Search::generateEdges + expand + backup (Search.hpp:84-86), then GameGenerator::make_move +
 * prepare_search for the games whose search is complete and the next openings for the games that ended (GameGenerator.cpp:97-118,145-185) */
int agx_engine_expand_group(AgxEngine* engine, int group, int n_groups, void* stream);
int agx_engine_advance_group(AgxEngine* engine, int group, int n_groups, void* stream);
int agx_engine_step_group(AgxEngine* engine, AgxNet* net, int group, int n_groups, void* stream);
int agx_stream_create(void** out_stream);
/* A stream whose kernels run only on the compute units set in the mask (bit i = compute unit i).  A pool stepped as N slices
 * (agx_engine_*_group) on N such streams with disjoint masks — 4 x 64 CUs on MI355X — runs its slices out of phase, each on its own part
 * of the chip: the power-limited network launches never cover the whole chip at once and hold a higher clock, and no slice waits for
 * another's stragglers (+10 % simulations/s).  Narrow the network's persistent grid to the slice's CU count
 * (agx_net_set_launch_width).  Do not destroy such a stream while the process lives (hipStreamDestroy of a CU-masked stream hangs on
 * ROCm 7.2): agx_stream_destroy leaves them to process exit. */
int agx_stream_create_with_cu_mask(void** out_stream, const uint32_t* cu_mask, int n_words);
/* the same, for callers that need several streams on one mask (the slices of a pool sharing the search partition of the chip): streams are
 * cached by (device, mask, instance); instance 0 is what agx_stream_create_with_cu_mask returns */
int agx_stream_create_with_cu_mask_instance(void** out_stream, const uint32_t* cu_mask, int n_words, int instance);
/* events order launches across streams on the device (hipStreamWaitEvent): record on one stream, make another wait for it */
int agx_event_create(void** out_event);
/* host pacing: a launch loop that runs ahead of the device without bound ends up spinning on a full launch queue (a whole CPU per rank).  Record a
 * BLOCKING event behind every step and wait for the one N steps back: the host thread sleeps, the device stays N steps fed
 * (measured: 1.98 -> 0.12 CPUs per rank, simulations/s +1 %; agx.hpp: HostPacer).  agx_event_synchronize SLEEPS: it polls the event between
 * 100 us naps (hipEventSynchronize spins in user space on ROCm 7.2 even for a hipEventBlockingSync event). */
int agx_event_create_blocking(void** out_event);
int agx_event_synchronize(void* event);
int agx_event_record(void* event, void* stream);
int agx_stream_wait_event(void* stream, void* event);
int agx_event_destroy(void* event);
/* streams agx_stream_create_with_cu_mask has created in this process: they are cached by (device, mask) and handed out again, never destroyed */
int agx_stream_masked_count(void);
int agx_stream_destroy(void* stream);
int agx_stream_synchronize(void* stream);
/* One game driven from OUTSIDE, the way evaluation/Player.cpp:100-129 drives a Tree / Search pair: agx_engine_set_board is
 * Search::cleanup + Tree::setBoard(board, signToMove) + Search::setBoard (Tree.cpp:128-151: the cached states still reachable from the new
 * position are kept, the root is the cached node of the position if there is one; the solver table ages by a generation); the caller then
 * steps select_solve / evaluate / expand (agx_engine_expand_group: no move is played by the engine) until ITS stopping rule says so, reads the
 * root (agx_engine_game_info) and decides the move.  h_board: one byte per cell, 0 empty / 1 cross / 2 circle.  forceRemoveRootNode is not
 * provided.  agx_engine_set_max_simulations: the budget Search::select(tree, maxSimulations) takes per call. */
int agx_engine_set_board(AgxEngine* engine, int game, const uint8_t* h_board, int sign_to_move, void* stream);
int agx_engine_set_max_simulations(AgxEngine* engine, int max_simulations);
/* Search::setBatchSize (search/monte_carlo/Search.cpp:252-255; SearchThread.cpp:125-126 grows it with sqrt(simulations)): the number of leaves the
 * select stage takes per game from the next launch on, 1 .. max_batch_size (the capacity the engine was created with) */
int agx_engine_set_batch_size(AgxEngine* engine, int batch_size);
/* Search::cleanup (search/monte_carlo/Search.cpp:233-242): the leaves that were selected but not expanded — in every task buffer, of every
 * tree of the engine — are dropped and their virtual losses taken back (Tree::cancelVirtualLoss, Tree.cpp:377-384).  agx_engine_set_board does
 * this for its game by itself; a caller that stops a double-buffered search (SearchThread.cpp:108-109) calls it before reading the tree. */
int agx_engine_cancel_pending(AgxEngine* engine, void* stream);
/* Tree::getSimulationCount / isRootProven / getNodeCount (Tree.hpp:86-92) as a search loop reads them between two steps (the stop condition of
 * player/SearchThread.cpp:181-199): out4 = { root visits, root proven (0 / 1), nodes in the tree, the game's error code }, read behind the work
 * `stream` holds and waiting for THAT stream only — agx_engine_game_info synchronises the whole device, which would also wait for a network
 * launch running beside the search on another stream. */
int agx_engine_root_summary(AgxEngine* engine, int game, void* stream, int* out4);
/* GeneratorThread::saveGames / loadGames (selfplay/GeneratorManager.cpp:98-122, GameGenerator::save / load, GameGenerator.cpp:122-141) for a
 * self-play pool: agx_engine_save_games lists the games in flight (h_out NULL: only counts them); agx_engine_restore_game, after
 * agx_engine_begin, makes pool slot game_slot continue from the saved position with an empty tree and solver table (GameGenerator::load calls
 * prepare_search on a fresh Tree) — the slot's game index starts again at 0 in the new pool. */
int agx_engine_save_games(AgxEngine* engine, AgxSavedGame* h_out, int capacity, int* count);
int agx_engine_restore_game(AgxEngine* engine, const AgxSavedGame* game, void* stream);
/* Search::solve(endTime >= 0) (search/monte_carlo/Search.cpp:159-183, called by SearchThread::asynchronous_run, player/SearchThread.cpp:150,168): the
 * solver's node limit becomes `max_nodes` (the reference: 10 000) and the leaves of the group's batches share `seconds` of wall-clock time from
 * the start of the launch — leaf i of a batch gets (time left) / (batch size - i), a leaf whose time is over stops after the node it is in
 * (AlphaBetaSearch.cpp:110-111,277).  seconds <= 0: every leaf gets its first node only.  Runs the serial solver (one wave per game) behind a
 * select stage enqueued separately (agx_engine_select_group); results depend on the clock, like the reference's. */
int agx_engine_solve_timed_group(AgxEngine* engine, int group, int n_groups, int max_nodes, double seconds, void* stream);
int agx_engine_set_force_expand_root(AgxEngine* engine, int force_expand_root); /* AgxEngineConfig.force_expand_root, for the launches that follow */
int agx_engine_buffers(AgxEngine* engine, AgxEngineBuffers* out);
int agx_engine_stats(AgxEngine* engine, AgxEngineStats* out);
/* Device memory this engine holds (every hipMalloc of agx_engine_create: tree heaps, the solver tables — 16 bytes x tss_table_entries per
 * game —, task / exchange buffers, the solver's spill areas and undo snapshots, record pools).  What one rank of a multi-GPU job needs of its
 * GPU's HBM next to the network's weights (GeneratorManager.cpp:146-152: one generator thread, i.e. one such pool, per device). */
int agx_engine_device_bytes(AgxEngine* engine, unsigned long long* bytes);
/* The same number for an engine that has not been created: what agx_engine_create(cfg) would allocate on a device with `compute_units` compute units
 * (256 on MI355X).  Touches no device — a launcher can size its ranks (one pool per device, GeneratorManager.cpp:146-152) before it starts them. */
int agx_engine_estimate_device_bytes(const AgxEngineConfig* cfg, int compute_units, unsigned long long* bytes);
/* Waves of the speculative search launch over the whole pool (AgxEngineConfig.speculative_waves, or the default resolved for this device, rules and
 * board); 0 when the pool runs the serial solver.  A group's launch takes waves / n_groups of them. */
int agx_engine_speculative_waves(AgxEngine* engine, int* waves);
/* Per-kernel timing of the engine's own launches, by HIP events recorded on the launch stream around every kernel (the role of
 * SearchStats' TimedStat members, search/monte_carlo/Search.hpp, for the device kernels).  Synchronises the device, returns the
 * time and launch count accumulated since the previous call in ms_out[4] / launches_out[4] (0 k_select, 1 k_solve, 2 k_expand,
 * 3 k_advance; either pointer may be null), then switches recording on or off. */
int agx_engine_kernel_timing(AgxEngine* engine, int enable, double* ms_out, long long* launches_out);
/* Tree::getInfo({}) of one game (Tree.cpp:403-424): root snapshot + board. */
int agx_engine_game_info(AgxEngine* engine, int game, AgxGameInfo* info, uint8_t* h_board, AgxEdgeView* h_root_edges, int edge_capacity);
int agx_engine_records(AgxEngine* engine, AgxMoveRecord* h_records, int record_capacity, AgxEdgeView* h_edges, int edge_capacity, int* n_records, int* n_edges);
/* Same, then empties the device-side record pools (what GeneratorManager::addToBuffer's hand-over does, GeneratorManager.cpp:
 * 160-164): a long-running loop calls this every few hundred steps so that record_capacity is never exhausted. */
int agx_engine_drain_records(AgxEngine* engine, AgxMoveRecord* h_records, int record_capacity, AgxEdgeView* h_edges, int edge_capacity, int* n_records, int* n_edges);
/* Everything the record pools hold: move records, root-edge snapshots (record_format bit 0), format-201 sample bytes (bit 1) and
 * finished games.  Any buffer may be NULL (with capacity 0) to skip that part; `counts` receives what the device holds.  drain != 0
 * empties all four pools afterwards (GeneratorManager::addToBuffer's hand-over, GeneratorManager.cpp:160-164). */
int agx_engine_fetch_records(AgxEngine* engine, AgxMoveRecord* h_records, int record_capacity, AgxEdgeView* h_edges, int edge_capacity,
		uint8_t* h_samples, int sample_capacity, AgxGameEnd* h_game_ends, int game_end_capacity, AgxRecordCounts* counts, int drain);
/* Match mode, the efficient way to step: every stage is ONE launch over both players' trees (about half of them search at any time),
 * only the network stage runs per player: select/solve for all, first_net on the first players' leaves, second_net on the second
 * players', expand/backup/move for all.  A tree that gets the move in this step starts searching in the next one; per-tree
 * results are the same as with two agx_engine_step_group calls (each tree performs the same sequence of operations). */
int agx_engine_step_match(AgxEngine* engine, AgxNet* first_net, AgxNet* second_net, void* stream);
int agx_engine_select_solve_match(AgxEngine* engine, void* stream);  /* the stages separately: this, then                        */
int agx_engine_expand_backup_match(AgxEngine* engine, void* stream); /* agx_engine_evaluate_group(e, net, 0 / 1, 2, s), then this */
/* Match mode: h_results int[n_games / 2][4] = games won / drawn / lost by the FIRST player of each pair and the games the pair has
 * finished (what EvaluationManager sums per player, evaluation/EvaluationManager.cpp); the moves of every game are in the records. */
int agx_engine_match_results(AgxEngine* engine, int* h_results, int pair_capacity);
/* Appends openings to the pool's list.  Finished games take the next unused opening in game order after every step; games that
 * found none left wait and pick one up here (the reference's generators produce openings on demand, GameGenerator.cpp:54-77). */
int agx_engine_add_openings(AgxEngine* engine, const uint16_t* h_openings, int n_openings);
/* FastZobristHashing keys used by the solver tables: uint64[2 * cells][2] = (lo, hi) per (cell, colour). */
int agx_engine_zobrist(AgxEngine* engine, uint64_t* h_keys, size_t n_words);

/* Test hooks: run single stages on caller-supplied positions (boards uint8[count][cells], 0 empty 1 cross 2 circle). */
int agx_debug_solve(AgxEngine* engine, const uint8_t* h_boards, const int* h_signs, int count, uint32_t* h_features, uint16_t* h_moves,
		uint16_t* h_scores, int* h_counts, uint32_t* h_flags, uint16_t* h_result_scores);
int agx_debug_solve_nodes(AgxEngine* engine, int count, unsigned long long* h_nodes);   /* solver nodes visited per position by the last agx_debug_solve (AlphaBetaSearch::solve's return value) */
int agx_debug_new_generation(AgxEngine* engine);                                       /* AlphaBetaSearch::increaseGeneration for the positions agx_debug_solve solves */
int agx_debug_pattern_state(AgxEngine* engine, const uint8_t* h_boards, const int* h_signs, const uint16_t* h_moves, int count, int n_moves,
		uint8_t* h_ptypes, uint8_t* h_threats, int16_t* h_lists, int lists_stride);
/* Host-only: the lookup tables the engine uploads (pattern uint8[1<<20], half-open-three uint8[1<<20], threat uint8[4096][2],
 * defence uint16[15][256][2]) for verification against the reference tables. */
int agx_host_tables(int rules, uint8_t* h_pattern, uint8_t* h_half_open_three, uint8_t* h_threat, uint16_t* h_defense);

/* ------------------------------------------------------------------------------------------------
 * Self-play record sink: finished games in the reference's dataset format 201.
 * Replaces GameDataStorage / GameDataBuffer on the producing side (src/dataset/GameDataStorage.cpp:217-250,
 * src/dataset/GameDataBuffer.cpp:97-113) and GeneratorManager::addToBuffer (src/selfplay/GeneratorManager.cpp:160-164).
 * The samples themselves are quantised on the device (AgxEngineConfig.record_format bit 1).
 * ---------------------------------------------------------------------------------------------- */
typedef struct AgxGameBuffer AgxGameBuffer; /* opaque; every call locks the buffer's mutex, so one buffer may serve the generator threads of several GPUs */

typedef struct AgxGameBufferStats
{ /* GameDataBufferStats (dataset/GameDataBuffer.hpp) */
	int games, samples, cross_win, draws, circle_win, game_length;
} AgxGameBufferStats;

int agx_game_buffer_create(int rules, int rows, int cols, int draw_after, AgxGameBuffer** out);
int agx_game_buffer_destroy(AgxGameBuffer* buffer);
int agx_game_buffer_clear(AgxGameBuffer* buffer);
/* Drains the engine's record pools (synchronises its device) and appends every game that has finished: samples by move number, all
 * moves of the game, outcome, rows, cols = the bytes of GameDataStorage::serialize.  Samples of games still running are kept until
 * their game ends.  The engine must record format-201 samples.  games_added may be NULL. */
int agx_game_buffer_collect(AgxGameBuffer* buffer, AgxEngine* engine, int* games_added);
int agx_game_buffer_stats(const AgxGameBuffer* buffer, AgxGameBufferStats* out);
/* GameDataStorage::serialize bytes of game `index`; h_bytes == NULL only queries *size. */
int agx_game_buffer_game(const AgxGameBuffer* buffer, int index, uint8_t* h_bytes, size_t capacity, size_t* size);
/* GameDataBuffer::save: {"format": 201, "config": {...}, "offsets": [...]} + newline + the games' bytes; compress != 0 wraps the file in
 * a zlib stream (the reference compresses with MinML's ZipWrapper, whose format is not in the reference tree). */
int agx_game_buffer_save(const AgxGameBuffer* buffer, const char* path, int compress);
/* GameDataBuffer::load (dataset/GameDataBuffer.cpp:115-131; called by GeneratorManager::loadState, GeneratorManager.cpp:263-275) for files written
 * by agx_game_buffer_save, compressed or not: the games are appended to the buffer */
int agx_game_buffer_load(AgxGameBuffer* buffer, const char* path);
/* Checkpoints of games in flight (GameGenerator::save / load, selfplay/GameGenerator.cpp:122-141): the format-201 samples a game has collected
 * so far wait in the buffer, keyed by (engine, game_slot, game_index), until its AgxGameEnd arrives.  take_pending serialises and removes
 * them ({ i32 move number, u32 bytes, sample } records; h_bytes NULL: size only, nothing removed); restore_pending hands them to the game that
 * continues it in another engine (agx_engine_restore_game: game_index 0); forget_engine drops whatever an engine that is about to be destroyed
 * left pending. */
int agx_game_buffer_take_pending(AgxGameBuffer* buffer, const AgxEngine* engine, int game_slot, int game_index, uint8_t* h_bytes, size_t capacity, size_t* size);
int agx_game_buffer_restore_pending(AgxGameBuffer* buffer, const AgxEngine* engine, int game_slot, int game_index, const uint8_t* h_bytes, size_t size);
int agx_game_buffer_forget_engine(AgxGameBuffer* buffer, const AgxEngine* engine);
/* Host-only reader of one format-201 sample (SearchDataStorage_v201's parsing constructor + storeTo, dataset/SearchDataStorage.cpp:
 * 300-320,375-409): per cell visits int32[rows*cols], prior float[rows*cols], value float[rows*cols][2] = (win, draw), score
 * uint16[rows*cols]; header int[3] = (minimax score raw bits, move number, flags); minimax_value float[2].  consumed may be NULL. */
int agx_sample_v201_unpack(const uint8_t* h_bytes, size_t size, int rows, int cols, int32_t* visits, float* prior, float* value, uint16_t* score,
		int* header, float* minimax_value, size_t* consumed);

/* Raw device-memory helpers so that non-HIP hosts (ctypes, cgo) can stage buffers. */
int agx_malloc(void** d_ptr, size_t bytes);
int agx_free(void* d_ptr);
int agx_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes);
int agx_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes);
int agx_memset(void* d_ptr, int value, size_t bytes);
int agx_device_synchronize(void);

/* Timing of work enqueued on `stream` with HIP events recorded on that same stream. */
typedef struct AgxTimer AgxTimer;
int agx_timer_create(AgxTimer** out);
int agx_timer_start(AgxTimer* t, void* stream);
int agx_timer_stop(AgxTimer* t, void* stream);
int agx_timer_elapsed_ms(AgxTimer* t, float* ms); /* synchronises on the stop event */
/* the same without waiting: *ready = 0 (and *ms untouched) while the timed work has not finished.  What NNEvaluator::asyncEvaluateGraphLaunch's
 * end-time estimate reads (NNEvaluator.cpp:197-206): a launch loop must not block on its previous network launch for a statistic. */
int agx_timer_poll_ms(AgxTimer* t, float* ms, int* ready);
int agx_timer_destroy(AgxTimer* t);

#ifdef __cplusplus
}
#endif
#endif /* AGX_H_ */
