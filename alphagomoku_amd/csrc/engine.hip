/*
 * engine.hip — the device-resident self-play engine: kernels, host object and its C ABI (include/agx.h).
 *
 * One step of the pool = what GameGenerator::generate does for every game of a GeneratorThread
 * (src/selfplay/GameGenerator.cpp:79-118, src/selfplay/GeneratorManager.cpp:124-141), as four launches:
 *
 *   k_solve<.., FUSED>   one wave per game, two stages back to back (a game's stages depend on no other game):
 *              Search::select          (Search.cpp:117-158)   up to `batch` PUCT descents per game, virtual loss
 *              Search::solve           (Search.cpp:159-183)   threat solver on every new leaf + NN feature encode,
 *              Search::scheduleToNN    (Search.cpp:184-199)   compacts the positions that need the network
 *              (k_select + k_solve<.., false> are the same stages as two launches: agx_engine_select_group / _solve_group, tournament pools)
 *   (network)  NNEvaluator::evaluate                         agx_nn_forward over the compacted list (nn_forward.hip)
 *   k_expand   Search::generateEdges / expand / backup (Search.cpp:206-232) and the move rule (GameGenerator.cpp:97-103)
 *   k_advance  GameGenerator::make_move + prepare_search (:145-185): final selector, sample record, outcome test (incl. renju
 *              fouls), NodeCache::cleanup as a keep-test + prefix-sum compaction into the game's other arena
 *   k_assign_openings / k_restart   finished games take the next openings in game order (GAME_NOT_STARTED -> next game)
 *   k_arena_service / _copy   trees that outgrew their arenas move into larger ones (NodeCache::resize, ObjectPool growth)
 *
 * Everything stays in HBM between steps; the host only enqueues launches.  One wavefront per game for the sequential
 * tree work (games are independent, the batch inside a game is order-dependent through virtual loss); k_advance uses a
 * 1024-thread workgroup per game because compaction is data parallel.
 */
#include "agx_internal.hpp"
#include "engine_types.hpp"
#include "dev_mcts.hpp"
#include "tables_host.hpp"
#include "renju_static.hpp"
#include "root_noise.hpp"
#include "sample_v201.hpp"

#include <vector>
#include <thread>
#include <chrono>
#include <mutex>
#include <cstring>

using namespace agx;
using namespace agx::dev;

namespace
{
	/* ------------------------------------------------------------------------------------------------------------ */
	/* Search::select for one task buffer: `g` owns the buffer (its GameState `ls` carries the batch bookkeeping and statistics), `tg` owns
	 * the tree (its GameState `gs`).  Self-play: tg == g.  Tournament search (shared_tree): every search thread g descends tree 0. */
	__device__ __forceinline__ void select_batch(EngineDev &E, int tg, int g, int lane, uint8_t *sh_board, u64 *sh_cboard, const u64 *sh_keys)
	{
		GameState &gs = E.games[tg];
		GameState &ls = E.games[g];
		DNode *nodes = nodes_of(E, tg, gs.arena);
		DEdge *edges = edges_of(E, tg, gs.arena);
		const int *ht = ht_of(E, tg);
		const int n = E.n;

		int n_tasks = ls.n_tasks;
		int trials = 2 * E.batch_limit; // Search.cpp:119: twice the buffer's current size (Search::setBatchSize)
		unsigned long long st_levels = 0, st_edges = 0, st_leaks = 0, st_proven = 0, st_dup = 0;
		while (n_tasks < E.batch_limit)
		{
			const int root = gs.root;
			const int sims = (root < 0) ? 0 : nodes[root].visits;
			if (sims > E.max_sims)
				break;
			if (E.noise_type != 0 && root >= 0 && !gs.noise_ready && nodes[root].n_edges > 0)
			{ // PUCTSelector::select's lazy noise initialisation on the first visit of an expanded root (EdgeSelector.cpp:1127-1137)
				if (lane == 0)
				{
					const DNode rn = nodes[root];
					const DEdge *root_edges = edges + rn.edge_begin;
					make_root_noise(E.noise_type, E.noise_weight, E.noise_seed, gs.opening_id, gs.n_moves, rn.n_edges, [&](int i) { return root_edges[i].prior; },
							E.noise + static_cast<size_t>(tg) * E.hw);
					gs.noise_ready = 1;
				}
				__threadfence_block();
				__syncthreads();
			}
			DTask &t = E.tasks[static_cast<size_t>(g) * E.batch + n_tasks];
			n_tasks++;
			// SearchTask::set (SearchTask.cpp:32-50)
			for (int i = lane; i < E.hw; i += 64)
				sh_board[i] = gs.board[i];
			if (lane < BWORDS)
				sh_cboard[lane] = gs.cboard[lane];
			__syncthreads();
			int path_len = 0, final_node = -1, out = 0, sign = gs.sign_to_move;
			u64 hash = gs.root_hash;
			int last_edge = -1;
			int node = root;
			// One dependent chain per level: edges of the node -> the chosen edge -> table slot -> the child's record.  The node
			// record is read ONCE (by the seek that found it), the chosen edge once, and the hash keys come from LDS.
			DNode nd;
			if (node >= 0)
				node_head(nd, nodes[node]);
			while (node >= 0)
			{
				if (path_len >= PATH_CAP)
				{
					if (lane == 0)
						ls.error = ERR_PATH_CAPACITY;
					break;
				}
				const int e = select_edge(E, nd, edges, lane, st_edges, (node == root && gs.noise_ready) ? E.noise + static_cast<size_t>(tg) * E.hw : nullptr);
				st_levels++;
				const DEdge ee = edges[e]; // as it was before this visit's virtual loss (what the leak test compares, Tree.cpp:75-85)
				const uint32_t mv = ee.move;
				const int s = mv & 3, cell = ((mv >> 2) & 127) * n + ((mv >> 9) & 127);
				const uint16_t flag_vl = static_cast<uint16_t>((ee.flag_vl & 0x8000u) | (((ee.flag_vl & 0x7FFF) + 1) & 0x7FFF));
				if (lane == 0)
				{ // SearchTask::append (SearchTask.cpp:52-60) + virtual loss (Tree.cpp:235-236)
					sh_board[cell] = static_cast<uint8_t>(s);
					sh_cboard[cell >> 5] |= static_cast<u64>(s) << (2 * (cell & 31));
					t.path_node[path_len] = node;
					t.path_edge[path_len] = e;
					nodes[node].vl = static_cast<int16_t>(nd.vl + 1);
					edges[e].flag_vl = flag_vl;
				}
				hash ^= sh_keys[3 + 3 * cell] ^ sh_keys[3 + 3 * cell + s] ^ sh_keys[sign] ^ sh_keys[3 - s];
				sign = 3 - s;
				path_len++;
				last_edge = e;
				__syncthreads();
				if (s_proven(ee.score))
				{
					out = 2;
					break;
				}
				DNode child;
				node = cache_seek_head(E, nodes, ht, hash, sh_cboard, sign, lane, child);
				final_node = node;
				if (node < 0 && lane == 0)
					edges[e].flag_vl = static_cast<uint16_t>(flag_vl | 0x8000u);
				if (has_leak(E, ee, node >= 0, child.score, child.win, child.draw))
				{
					out = 1;
					break;
				}
				nd = child;
			}
			// publish the task
			for (int i = lane; i < E.hw; i += 64)
				t.board[i] = sh_board[i];
			if (lane < BWORDS)
				t.cboard[lane] = sh_cboard[lane];
			if (lane == 0)
			{
				t.path_len = path_len;
				t.final_node = final_node;
				t.n_edges = 0;
				t.flags = t.flags & (TF_STATICALLY_SOLVED | TF_RECURSIVELY_SOLVED); // these two survive SearchTask::set
				t.sign_to_move = sign;
				t.score = s_unknown(0);
				t.win = 0.0f;
				t.draw = 0.0f;
				// (t.moves_left stays: SearchTask::set (SearchTask.cpp:32-50) resets the value, not the moves-left estimate — a proven-edge leaf, which neither
				//  the solver nor the network touches, backs up what the previous occupant of this task slot left behind, Tree.cpp:304)
				t.needs_nn = 0;
				t.hash = hash;
				if (out == 0 && path_len > gs.max_depth)
					gs.max_depth = path_len; // Tree.cpp:249 (REACHED_LEAF only)
			}
			__syncthreads();
			if (path_len == 0)
				break; // the root itself has to be evaluated
			for (int i = 0; i < n_tasks - 1; i++)
			{ // Search::is_duplicate (statistics only)
				const DTask &o = E.tasks[static_cast<size_t>(g) * E.batch + i];
				if (o.path_len > 0 && o.path_edge[o.path_len - 1] == last_edge)
				{
					st_dup++;
					break;
				}
			}
			if (out == 1)
			{
				correct_information_leak(nodes, edges, t, path_len, final_node, lane);
				cancel_virtual_loss(nodes, edges, t, path_len, lane);
				st_leaks++;
				n_tasks--;
			}
			if (out == 2 && lane == 0)
			{ // Search.cpp:139-150
				const uint32_t sc = edges[last_edge].score;
				float w, d;
				s_to_value(sc, w, d);
				t.final_node = -1;
				t.sign_to_move = 3 - sign;
				t.score = sc;
				t.win = w;
				t.draw = d;
				t.flags |= TF_BY_SOLVER | TF_SKIP_EDGE_GENERATION;
			}
			if (out == 2)
				st_proven++;
			__syncthreads();
			if (--trials <= 0)
				break;
		}
		if (lane == 0)
		{
			ls.n_tasks = n_tasks;
			ls.stats[2] += st_leaks;
			ls.stats[3] += st_proven;
			ls.stats[6] += st_levels;
			ls.stats[7] += st_edges;
			ls.stats[9] += st_dup;
		}
	}
	__global__ __launch_bounds__(64) void k_select(EngineDev E)
	{
		__shared__ uint8_t sh_board[MAXHW];
		__shared__ u64 sh_cboard[BWORDS];
		__shared__ u64 sh_keys[3 * (1 + MAXHW)]; // FullZobristHashing keys of the node cache: four per level, from LDS instead of L2
		const int g = E.g0 + blockIdx.x, lane = threadIdx.x;
		const int tg = E.shared_tree ? 0 : g;
		GameState &gs = E.games[tg];
		if (!gs.active || gs.error != 0 || gs.outcome != 0 || gs.grow_pending)
			return; // (grow_pending: the previous batch still waits for its expansion in larger arenas)
		use_game_arenas(E, tg);
		for (int i = lane; i < 3 * (1 + E.hw); i += 64)
			sh_keys[i] = E.nc_keys[i];
		if (!E.shared_tree)
		{
			if (!gs.solve_pending)
				select_batch(E, g, g, lane, sh_board, sh_cboard, sh_keys);
			return;
		}
		// tournament search: the search threads take the tree one after the other (SearchThread.cpp:124-129), in thread order
		// (double-buffered search: this launch's buffer of every thread — the other buffer's leaves keep their virtual losses meanwhile)
		for (int t = E.grp_first; t < E.grp_first + E.grp_count; t++)
		{
			if (E.games[t].error == 0)
				select_batch(E, 0, t, lane, sh_board, sh_cboard, sh_keys);
			__syncthreads();
		}
	}

	/* ------------------------------------------------------------------------------------------------------------ */
	/* Where a solve keeps what does not fit into LDS: `area` indexes the per-game spill areas (the game itself, or for speculative solves —
	 * several tasks of one game at once — the wave's own area behind the games'); `overlay` != nullptr makes the solve speculative
	 * (dev_solver.hpp: the table is only read, every touched bucket lives in the overlay). */
	/*
	 * Parking (round 5).  A launch of k_search_spec ends when its last solve does, and a few solves per launch run several times the
	 * average (renju: positions whose move generation tests dozens of cells for fouls — 0.5 % of the leaves take 5-9 ms against 1.7 on average,
	 * profiles/r05_c5_search_launch_timeline.txt): the compute units of the slice idle for half the launch behind a handful of waves.  Once
	 * SPEC_PARK_FRACTION of the launch's games are done and the queue has run dry, a speculative solve that is still running is PARKED at its
	 * next command boundary: the solver's LDS state (everything the frame machine needs: it already yields to the caller for every stone) goes
	 * into a park buffer, the in-use parts of the wave's HBM tails (undo snapshots, action stack / frame / list tails) into the buffer's own
	 * spill area, the game sits this step out like a game whose commit was deferred (solve_pending), and the next launch — which queues the
	 * game's unsolved leaves as it does for a deferred game — finds the task's park buffer and takes the solve up where it stopped, in any
	 * wave.  The solve's result cannot depend on it: the machine continues from the same state against the same table (the game's table is not
	 * written while its batch is in flight).  A solve is parked at most once (it then has a whole launch in front of it), and only when a
	 * buffer is free.  Pacing only, like the yield rule.
	 */
	constexpr float SPEC_PARK_FRACTION = 0.90f; // (C5 on one box: 0.80 484 k, 0.86 511 k, 0.90 520 k, 0.94 469 k, 0.97 406 k simulations/s; without parking 408 k)
	struct ParkCtl
	{
			int count;     // games of the launch
			int threshold; // games done from which running solves are parked (0x7FFFFFFF: never)
	};
	/* what a solve keeps in spill area `from`, copied to area `to`: the parts in use (or a superset of them) */
	template<class SH>
	__device__ __forceinline__ void park_copy_tails(const SH &sh, const EngineDev &E, int from, int to, int lane)
	{
		lane = fresh_lane(lane); // (cold code: none of its per-lane addresses is to be computed at the top of the kernel and kept in registers or scratch)
		{ // undo snapshots: one 512-byte level per stone on the board
			const u64 *a = E.snap_spill + static_cast<size_t>(from) * (E.hw + 2) * 64;
			u64 *b = E.snap_spill + static_cast<size_t>(to) * (E.hw + 2) * 64;
			const int words = min(sh.depth, E.hw + 2) * 64;
			for (int i = lane; i < words; i += 64)
				b[i] = a[i];
		}
		{ // action stack beyond its LDS part
			const uint32_t *a = E.act + static_cast<size_t>(from) * E.act_cap;
			uint32_t *b = E.act + static_cast<size_t>(to) * E.act_cap;
			const int top = min(sh.stack_max, E.act_cap);
			for (int i = SH::ACT_LDS + lane; i < top; i += 64)
				b[i] = a[i];
		}
		{ // frames beyond the LDS-resident ones (32 bytes each)
			const u64 *a = reinterpret_cast<const u64*>(E.frame_spill) + static_cast<size_t>(from) * MAX_FRAMES * 4;
			u64 *b = reinterpret_cast<u64*>(E.frame_spill) + static_cast<size_t>(to) * MAX_FRAMES * 4;
			const int top = min(sh.level + 2, MAX_FRAMES) * 4;
			for (int i = SH::FRAMES * 4 + lane; i < top; i += 64)
				b[i] = a[i];
		}
		{ // threat-list entries beyond a list's LDS capacity
			const uint16_t *a = E.list_spill + static_cast<size_t>(from) * 20 * MAXHW;
			uint16_t *b = E.list_spill + static_cast<size_t>(to) * 20 * MAXHW;
			for (int st = 0; st < 20; st++)
			{
				const int side = st / 10, type = st % 10;
				if (type < 2)
					continue;
				const int cnt = sh.count[side][type];
				for (int i = SH::list_cap(type) + lane; i < cnt; i += 64)
					b[st * SH::HW + i] = a[st * SH::HW + i];
			}
		}
	}
	template<class SH>
	__device__ __forceinline__ void park_point_tails_at(SH &sh, const EngineDev &E, int area)
	{ // HBM tails of the LDS-resident threat lists and frames (dev_solver.hpp: list_get / frame_get), undo snapshots (lane 0)
		sh.spill_lists = E.list_spill + static_cast<size_t>(area) * 20 * MAXHW; // (stride of list_get / list_set: SolverSharedT::HW <= MAXHW whatever the kernel instantiation)
		sh.spill_frames = reinterpret_cast<Frame*>(E.frame_spill) + static_cast<size_t>(area) * MAX_FRAMES;
		sh.snap = E.snap_spill + static_cast<size_t>(area) * (E.hw + 2) * 64; // undo snapshots, one level per stone on the board (dev_solver.hpp)
	}
	/* every lane: is it time to park (wave-uniform result)?  One counter read while the launch is busy, three more once the threshold is reached. */
	__device__ __forceinline__ bool park_wanted(const EngineDev &E, const ParkCtl &p, int lane)
	{
		const int *c_done = E.counters + E.yield_counter;                 // games of the launch that are done with their batch
		const int *c_queue = E.counters + SPEC_COUNTER0 + 4 * E.spec_group; // [0] select cursor, [1] queue head, [2] queue tail, [3] games selected
		int yes = 0;
		if (lane == 0 && __hip_atomic_load(c_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= p.threshold)
			yes = (__hip_atomic_load(c_queue + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= p.count
					&& __hip_atomic_load(c_queue + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= __hip_atomic_load(c_queue + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ? 1 : 0;
		return __builtin_amdgcn_readfirstlane(yes) != 0;
	}
	/* claims a free park buffer for task `slot` (every lane gets the same answer: the buffer's index, or -1) */
	__device__ __forceinline__ int park_claim(const EngineDev &E, int slot, int lane)
	{
		int b = -1;
		if (lane == 0)
			for (int probe = 0; probe < 8 && b < 0; probe++)
			{
				const int i = (slot * 7 + probe * 13) & (SPEC_PARK_POOL - 1);
				if (atomicCAS(&E.park_owner[i], 0, slot + 1) == 0)
					b = i;
			}
		return __builtin_amdgcn_readfirstlane(b);
	}
	/* forgets what game g has parked (the game is restarted, its board set from outside, its pending batch cancelled): lane 0 / one thread.
	 * With a parked solve the game's OTHER leaves of that launch stay "solved speculatively" (SpecTask::solved, overlay uncommitted) across
	 * launches; whoever discards the batch discards those too, or a later batch's proven-edge task in the same slot — which the solver never
	 * sees — would be taken for one by spec_commit_game and have a stale overlay written into the table. */
	__device__ __forceinline__ void park_forget_game(const EngineDev &E, int g)
	{
		for (int k = 0; k < E.batch; k++)
		{
			const int b = E.park_slot[g * E.batch + k];
			if (b != 0)
			{
				E.park_slot[g * E.batch + k] = 0;
				__hip_atomic_store(&E.park_owner[b - 1], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
			}
			if (E.spec_on)
				E.spec_tasks[g * E.batch + k].solved = 0;
		}
	}
	/* A serial solve launch (k_solve: Search::solve with a time limit, or solve() behind flush_select()) on a pool whose last speculative launch
	 * left game g waiting (a deferred commit or a parked solve): the leaves of the batch that were solved speculatively but never committed go
	 * back to what the select stage made of them — as spec_commit_game's defer path does — and the parked solve is dropped, so that the serial
	 * loop solves every pending leaf on the table itself, in batch order.  lane 0 / one thread. */
	__device__ __forceinline__ void spec_discard_pending(const EngineDev &E, int g)
	{
		if (!E.spec_on)
			return;
		for (int k = 0; k < E.batch; k++)
		{
			SpecTask &h = E.spec_tasks[g * E.batch + k];
			if (h.solved != 0)
			{
				DTask &t = E.tasks[static_cast<size_t>(g) * E.batch + k];
				t.flags = h.flags0;
				t.win = 0.0f;
				t.draw = 0.0f;
				t.moves_left = h.moves_left0;
			}
		}
		park_forget_game(E, g); // (also clears SpecTask::solved)
	}
	/* returns true when the solve was parked (nothing of `t` has been written then), false when it is finished */
	template<bool RENJU, class SH>
	__device__ __forceinline__ bool solve_task(SH &sh, const EngineDev &E, int g, DTask &t, int slot, int generation, int lane, unsigned long long &solver_nodes,
			int area, u64 *overlay = nullptr, const ParkCtl *park = nullptr)
	{ // AlphaBetaSearch::solve (AlphaBetaSearch.cpp:77-156)
		uint32_t *act = E.act + static_cast<size_t>(area) * E.act_cap;
		u64 *tt = E.tt + static_cast<size_t>(g % E.tt_mod) * (E.tt_bucket_mask + 1ull) * 8ull; // (double-buffered search: both buffers of a thread use its one solver)
		const int parked_in = (park != nullptr) ? __builtin_amdgcn_readfirstlane(E.park_slot[slot]) : 0; // park buffer + 1 this task's solve waits in
		int depth_first = 0, stack_before_first = 0;
		if (parked_in != 0)
		{ // ---- take a parked solve up again: LDS state and loop variables out of the buffer, the HBM tails into this wave's area ----
			static_assert(sizeof(SH) % 8 == 0 && sizeof(SH) / 8 <= SPEC_PARK_WORDS - 8, "park buffer holds the solver's LDS state");
			const u64 *src = E.park_lds + static_cast<size_t>(parked_in - 1) * SPEC_PARK_WORDS;
			u64 *lds = reinterpret_cast<u64*>(&sh);
			for (int i = fresh_lane(lane); i < static_cast<int>(sizeof(SH) / 8); i += 64)
				lds[i] = global_view(src)[i];
			depth_first = static_cast<int>(src[SPEC_PARK_WORDS - 8]);
			stack_before_first = static_cast<int>(src[SPEC_PARK_WORDS - 7]);
			wave_sync();
			park_copy_tails(sh, E, E.park_area0 + parked_in - 1, area, lane);
			if (lane == 0)
			{
				park_point_tails_at(sh, E, area);
				sh.pf_valid = 0; // (the bucket prefetched for the next frame was in registers)
				E.park_slot[slot] = 0;
			}
			__threadfence(); // the copies are done before the buffer can be claimed again
			wave_sync();
			if (lane == 0)
				__hip_atomic_store(&E.park_owner[parked_in - 1], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
		}
		else
		{
		if (lane == 0)
		{
			park_point_tails_at(sh, E, area);
			sh.ov_on = (overlay != nullptr) ? 1 : 0;
			sh.ov_data = overlay;
			sh.ov_count = 0;
			sh.ov_overflow = 0;
			sh.pf_slot = -1;
		}
		if (lane < OV_CAP / 32)
			sh.ov_dirty[lane] = 0;
		}
#ifdef AGX_SOLVER_PROFILE
		unsigned long long c0 = wall_clock64(), c_run = 0, c_place = 0, n_place = 0;
		unsigned long long c1 = c0, t_forbidden = 0;
#endif
		if (parked_in == 0)
		{
		solver_set_board(sh, E, t.board, t.sign_to_move, lane);
		solver_encode_features(sh, E, E.nn_features + static_cast<size_t>(slot) * E.hw, lane);
		#ifdef AGX_SOLVER_PROFILE
		const unsigned long long c_enc = wall_clock64();
		#endif
		if (RENJU)
			solver_encode_forbidden(sh, E, E.nn_features + static_cast<size_t>(slot) * E.hw, lane);
		#ifdef AGX_SOLVER_PROFILE
		c1 = wall_clock64();
		t_forbidden = c1 - c_enc; // renju: the forbidden-move bits of the network input (probes place and remove stones)
		#endif
		u64 lo = 0, hi = 0;
		for (int i = lane; i < E.hw; i += 64)
		{
			const int v = sh.board[i];
			if (v == 1 || v == 2)
			{
				lo ^= zobrist_word(E.zobrist_seed, 2 * (2 * i + v - 1));
				hi ^= zobrist_word(E.zobrist_seed, 2 * (2 * i + v - 1) + 1);
			}
		}
		lo = wave_reduce_xor64(lo);
		hi = wave_reduce_xor64(hi);
		if (lane == 0)
		{
			sh.hash_lo = lo;
			sh.hash_hi = hi;
			sh.node_counter = 0;
			sh.stack_offset = 0;
			sh.stack_max = 0;
			sh.error = 0;
			sh.pf_valid = 0;
#ifdef AGX_SOLVER_PROFILE
			for (int i = 0; i < 8; i++)
				sh.prof[i] = 0;
			for (int i = 0; i < 24; i++)
				sh.dprof[i] = 0;
#endif
			Frame &f = sh.frames[0];
			f.base = 0;
			f.size = 0;
			f.baseline = static_cast<uint16_t>(s_unknown(0));
			f.must_defend = f.has_initiative = f.fully_expanded = 0;
		}
		wave_sync();
		}
		uint32_t result = s_unknown(0);
		u64 pf_word = 0; // lanes 0-7: the transposition-table bucket prefetched for the frame about to be entered
		uint8_t pf_pattern = 0; // pattern-table entries prefetched for the stone about to be placed / removed (dev_solver.hpp:pattern_prefetch)
		int pf_pattern_tag = -1; // move | add << 16 they belong to
		u64 pf_snap = 0; // an undo's snapshot word, requested by the frame machine when the node returns (tag: the move, add bit clear)
		// a solve may be parked once, and not in a launch in which nothing can be gained by it
		const bool may_park = (park != nullptr) && (parked_in == 0) && (park->threshold != 0x7FFFFFFF) && (overlay != nullptr);
		int turns = 0;
		for (int depth = depth_first; depth <= E.tss_max_depth; depth += 4)
		{
			int stack_before = 0;
			if (parked_in != 0 && depth == depth_first)
				stack_before = stack_before_first; // (the iteration the solve was parked in goes on)
			else if (lane == 0)
			{
				stack_before = sh.stack_max;
				Frame &f = sh.frames[0];
				f.depth_remaining = depth;
				f.alpha = 0x0000u;
				f.beta = 0xFFFFu;
				f.i = 0;
				sh.level = 0;
				sh.phase = 0;
				sh.pending_value = 0;
			}
			wave_sync();
			while (true)
			{
				if (may_park && ((++turns) & 15) == 0 && park_wanted(E, *park, lane))
				{ // ---- park: between two commands everything the machine needs is in LDS ----
					const int b = park_claim(E, slot, lane);
					if (b >= 0)
					{
						u64 *dst = E.park_lds + static_cast<size_t>(b) * SPEC_PARK_WORDS;
						const u64 *lds = reinterpret_cast<const u64*>(&sh);
						for (int i = fresh_lane(lane); i < static_cast<int>(sizeof(SH) / 8); i += 64)
							global_view(dst)[i] = lds[i];
						const int sb = __builtin_amdgcn_readfirstlane(stack_before); // (lane 0's)
						if (lane == 0)
						{
							dst[SPEC_PARK_WORDS - 8] = static_cast<u64>(depth);
							dst[SPEC_PARK_WORDS - 7] = static_cast<u64>(sb);
						}
						park_copy_tails(sh, E, area, E.park_area0 + b, lane);
						__threadfence();
						if (lane == 0)
						{
							GameState &pg = E.games[g];
							E.park_slot[slot] = b + 1;
							pg.solve_pending = 1; // (solve_pos stays: 0, or where a deferred game's batch went on)
							atomicAdd(&pg.spec_stats[3], 1ull);
						}
						__threadfence();
						wave_sync();
						return true;
					}
				}
#ifdef AGX_SOLVER_PROFILE
				const unsigned long long r0 = wall_clock64();
#endif
				pf_pattern_tag = -1;
				const int cmd_now = solver_run<RENJU>(sh, E, act, tt, generation, lane, pf_word, pf_pattern, pf_pattern_tag, pf_snap);
				if (lane == 0)
					sh.cmd = cmd_now;
				wave_sync();
#ifdef AGX_SOLVER_PROFILE
				const unsigned long long r1 = wall_clock64();
				c_run += r1 - r0;
#endif
				const int cmd = sh.cmd;
				if (cmd == CMD_ADD)
					solver_place(sh, E, static_cast<uint32_t>(sh.cmd_move), true, lane, pf_pattern_tag == (sh.cmd_move | 0x10000), pf_pattern);
				else if (cmd == CMD_UNDO)
					solver_place(sh, E, static_cast<uint32_t>(sh.cmd_move), false, lane, pf_pattern_tag == sh.cmd_move, pf_pattern, pf_pattern_tag == sh.cmd_move, pf_snap);
				else
					break;
#ifdef AGX_SOLVER_PROFILE
				c_place += wall_clock64() - r1;
				n_place++;
#endif
			}
			int stop = 0;
			if (lane == 0)
			{
				const uint32_t r = static_cast<uint32_t>(sh.result_score);
				stop = (sh.frames[0].size == 0 || s_proven(r) || sh.node_counter >= E.tss_max_nodes || sh.stack_max == stack_before || sh.error != 0) ? 1 : 0;
				if (E.solve_time_ticks != 0ull && wall_clock64() >= sh.time_deadline)
					stop = 1; // AlphaBetaSearch.cpp:111
			}
			stop = __builtin_amdgcn_readfirstlane(stop);
			result = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(sh.result_score));
			wave_sync();
			if (stop)
				break;
		}
		const int size = sh.frames[0].size;
		for (int i = lane; i < size; i += 64)
		{
			const uint32_t a = act_get(sh, act, i);
			t.emove[i] = static_cast<uint16_t>(a & 0xFFFFu);
			t.escore[i] = static_cast<uint16_t>(a >> 16);
		}
		if (lane == 0)
		{
			t.n_edges = size;
			t.score = result;
			if (s_proven(result))
			{
				float w, d;
				s_to_value(result, w, d);
				t.win = w;
				t.draw = d;
				t.moves_left = static_cast<float>(s_distance(result));
				t.flags |= TF_RECURSIVELY_SOLVED;
			}
			if (sh.frames[0].must_defend)
				t.flags |= TF_MUST_DEFEND;
			if ((sh.node_counter & NODE_COUNT_MASK) <= 1)
				t.flags |= TF_STATICALLY_SOLVED;
			t.flags |= TF_BY_SOLVER;
			if (sh.error != 0 && sh.error != ERR_OVERLAY)
				E.games[g].error = sh.error;
		}
		solver_nodes += static_cast<unsigned long long>(sh.node_counter & NODE_COUNT_MASK);
#ifdef AGX_SOLVER_PROFILE
		if (lane == 0)
		{
			GameState &pg = E.games[g];
			pg.prof[0] += c1 - c0;               // set_board + encode
			pg.prof[1] += c_run;                 // lane-0 frame machine (move generation, table probes, ordering)
			pg.prof[2] += c_place;               // place / remove stone
			pg.prof[3] += n_place;
			pg.prof[4] += wall_clock64() - c0;   // whole solve
			pg.prof[5] += 1;
			pg.prof[6] += sh.prof[0] | (sh.prof[1] << 32);
			pg.prof[7] += sh.prof[2] | (sh.prof[3] << 32);
			for (int i = 0; i < 24; i++)
				pg.dprof[i] += sh.dprof[i];
			pg.dprof[23] += t_forbidden;
		}
#endif
		wave_sync();
		return false;
	}

	/* Search::scheduleToNN (Search.cpp:184-199) for one game whose whole batch has been solved: input symmetries, the device-side queue of
	 * positions for the network.  Returns the number of positions scheduled. */
	template<class SH>
	__device__ __forceinline__ unsigned long long schedule_to_nn(SH &sh, const EngineDev &E, int g, GameState &gs, int n_tasks, int lane)
	{
		unsigned long long scheduled = 0;
		int queued = gs.nn_queued;
		for (int j = 0; j < n_tasks; j++)
		{
			DTask &t = E.tasks[static_cast<size_t>(g) * E.batch + j];
			const bool needs = (t.path_len == 0) || !s_proven(t.score);
			int symmetry = 0;
			if (needs && E.use_symmetries)
			{ // NNEvaluator::addToQueue + pack_to_network (NNEvaluator.cpp:134-141,244-262): augment the features in place
				symmetry = static_cast<int>(symmetry_mix(E.symmetry_seed ^ (static_cast<u64>(static_cast<uint32_t>(gs.opening_id)) << 32) ^ static_cast<uint32_t>(queued)) >> 61);
				if (symmetry != 0)
				{
					uint32_t *feat = E.nn_features + static_cast<size_t>(g * E.batch + j) * E.hw;
					__threadfence_block();
					for (int i = lane; i < E.hw; i += 64)
						sh.act[i] = feat[i]; // the action stack is idle between solves
					wave_sync();
					for (int i = lane; i < E.hw; i += 64)
					{
						int sr, sc;
						symmetry_source(symmetry, E.n, i / E.n, i % E.n, sr, sc);
						feat[i] = shuffle_feature_directions(sh.act[sr * E.n + sc], symmetry);
					}
					wave_sync();
				}
			}
			if (lane == 0)
			{
				t.needs_nn = needs ? 1 : 0;
				t.symmetry = symmetry;
				if (needs)
				{
					// each group owns the list segment of its games; a merged match launch fills the two players' lists (= groups 0 and 1 of 2)
					const int half = (E.match_merged && g >= E.n_games / 2) ? 1 : 0;
					const int first = E.match_merged ? half * (E.n_games / 2) : E.g0;
					const int idx = atomicAdd(&E.counters[E.nn_counter + half], 1);
					E.nn_list[static_cast<size_t>(first) * E.batch + idx] = g * E.batch + j;
					scheduled++;
				}
			}
			if (needs)
				queued++;
		}
		if (lane == 0)
			gs.nn_queued = queued;
		return scheduled;
	}

	/* NFIX: board size known at compile time (15, 20) or 0 for any size.  The solver divides and takes remainders by the board size
	 * all the time (cell <-> row, column); with a constant these are a multiply and a shift instead of ~40 instructions each. */
	/* FUSED: Search::select of the game runs in the same wave right before its solver (self-play and match pools; a tournament-search pool
	 * selects for all threads in one wave, k_select).  As two launches the select stage lasts as long as its slowest game (deep end-game
	 * paths, leak retries: measured 2.4 x the mean wave) while the other SIMDs idle; fused, a slow descent only delays its own game's solver
	 * and the launch ends with the yield rule as before.  The select stage's LDS (board, keys) aliases the threat lists, which the solver
	 * initialises afterwards. */
#ifndef AGX_SOLVE_WAVES
#define AGX_SOLVE_WAVES 1 /* waves per SIMD the register allocation of k_solve leaves room for: a pool of n games has n solver waves, one per SIMD at the
                             BASELINE pool size, and an uncapped allocation is the fastest single wave (k_search_spec is the kernel built for occupancy) */
#endif
	template<bool RENJU, int NFIX, bool FUSED>
	__global__ __launch_bounds__(64, AGX_SOLVE_WAVES) void k_solve(EngineDev E)
	{
		if (NFIX != 0)
		{
			E.n = NFIX;
			E.hw = NFIX * NFIX;
		}
		typedef SolverSharedT<(NFIX != 0) ? NFIX : MAXN> SH;
		__shared__ SH sh;
		const int g = E.g0 + blockIdx.x, lane = threadIdx.x;
		if (FUSED)
		{
			static_assert(offsetof(SH, act) % 8 == 0 && offsetof(SH, frames) == offsetof(SH, act) + sizeof(sh.act) && offsetof(SH, ptype) == offsetof(SH, frames) + sizeof(sh.frames)
				&& offsetof(SH, threat) == offsetof(SH, ptype) + sizeof(sh.ptype) && offsetof(SH, items) == offsetof(SH, threat) + sizeof(sh.threat),
				"select-stage keys: 64-bit words over act + frames + ptype + threat + items (contiguous) and on over the per-solve scratch up to board");
			static_assert(offsetof(SH, board) - offsetof(SH, act) >= SH::SELECT_KEY_BYTES && offsetof(SH, board) > offsetof(SH, items) && sizeof(sh.lines) >= BWORDS * sizeof(u64),
				"select-stage LDS must fit");
			const GameState &sg = E.games[g];
			if (sg.active && sg.error == 0 && sg.outcome == 0 && !sg.grow_pending)
			{
				use_game_arenas(E, g);
				u64 *sel_keys = reinterpret_cast<u64*>(&sh.act[0]);
				u64 *sel_cboard = &sh.lines[0];
				uint8_t *sel_board = &sh.board[0];
				for (int i = lane; i < 3 * (1 + E.hw); i += 64)
					sel_keys[i] = E.nc_keys[i];
				if (!sg.solve_pending)
					select_batch(E, g, g, lane, sel_board, sel_cboard, sel_keys);
			}
			__threadfence(); // the tasks written by the select stage are read back below (other lanes, vector L1)
			__syncthreads();
		}
		GameState &gs = E.games[g];
		const bool idle = (!gs.active || gs.error != 0 || gs.outcome != 0 || E.games[E.shared_tree ? 0 : g].grow_pending != 0);
		const int n_tasks = idle ? 0 : gs.n_tasks;
		if (E.spec_on && !idle && gs.solve_pending)
		{ // (a speculative pool stepped with a serial launch: nothing of the last speculative launch may outlive this one)
			if (lane == 0)
				spec_discard_pending(E, g);
			__threadfence();
			__syncthreads();
		}
		if (n_tasks > 0)
			solver_load_threat_table(sh, E, lane);
		unsigned long long solver_nodes = 0;
		/*
		 * The tasks of a game are solved strictly in order (they share the game's transposition table), so a launch lasts as long as
		 * its slowest game.  To keep the other CUs from idling behind stragglers a game may YIELD between two tasks once
		 * yield_fraction of the launch's games have finished: it keeps its position in the batch, sits out this step's network /
		 * expand stages and resumes in the next launch.  Each game still sees exactly the same sequence of operations.
		 */
		// (match mode: about half of a group's trees wait for their opponents and count as done at once)
		const float fraction = E.match_mode ? 0.5f + 0.5f * E.yield_fraction : E.yield_fraction;
		const int threshold = (E.yield_fraction > 0.0f) ? static_cast<int>(fraction * gridDim.x) : 0x7FFFFFFF;
		int k = idle ? 0 : gs.solve_pos;
		bool yielded = false;
		const unsigned long long t_launch = (E.solve_time_ticks != 0ull) ? wall_clock64() : 0ull;
		for (; k < n_tasks; k++)
		{
			DTask &t = E.tasks[static_cast<size_t>(g) * E.batch + k];
			const int slot = g * E.batch + k;
			if ((t.flags & TF_BY_SOLVER) == 0)
			{
				if (E.solve_time_ticks != 0ull && lane == 0)
				{ // ab_search.setTimeLimit((endTime - getTime()) / (getBatchSize() - i)) (Search.cpp:175-180): this task's share of what is left
					const unsigned long long now = wall_clock64(), end = t_launch + E.solve_time_ticks;
					sh.time_deadline = now + ((end > now) ? (end - now) / static_cast<unsigned long long>(max(1, E.batch_limit - k)) : 0ull); // (the buffer's SIZE, not its fill: a partly filled batch gets the reference's smaller shares)
				}
				if (threshold != 0x7FFFFFFF && k > gs.solve_pos)
				{ // at least one task per launch is always solved, so every game makes progress
					int done = 0;
					if (lane == 0)
						done = __hip_atomic_load(&E.counters[E.yield_counter], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					done = __builtin_amdgcn_readfirstlane(done);
					if (done >= threshold)
					{
						yielded = true;
						break;
					}
				}
				solve_task<RENJU>(sh, E, g, t, slot, gs.generation, lane, solver_nodes, g);
			}
		}
		if (yielded)
		{
			if (lane == 0)
			{
				gs.solve_pos = k;
				gs.solve_pending = 1;
				gs.stats[5] += solver_nodes;
			}
			return;
		}
		// Search::scheduleToNN (Search.cpp:184-199), once the whole batch has been solved
		const unsigned long long scheduled = idle ? 0ull : schedule_to_nn(sh, E, g, gs, n_tasks, lane);
		if (lane == 0)
		{
			if (!idle)
			{
				gs.solve_pos = 0;
				gs.solve_pending = 0;
				gs.stats[5] += solver_nodes;
				gs.stats[1] += scheduled;
			}
			atomicAdd(&E.counters[E.yield_counter], 1);
		}
	}


	/* ------------------------------------------------------------------------------------------------------------ */
	/*
	 * k_search_spec — Search::select + Search::solve + Search::scheduleToNN of a group of games as ONE launch of persistent waves
	 * (AgxEngineConfig.speculative_solver).
	 *
	 * Why: a game's leaves share its transposition table, so Search::solve (Search.cpp:159-183) handles them one after the other and
	 * k_solve gives a game one wave — 1024 waves on 1024 SIMDs, each a dependent chain that waits on LDS / L2 about half of the time with
	 * nobody to fill the gaps.  Measured on MI355X (profiles/r03_solver_occupancy.txt): the same solver code with three waves per SIMD
	 * delivers 2.6 x the solver nodes per second.  A pool of 1024 games only has that many waves if the ~6.6 leaves of a batch are solved at
	 * the same time, which is what this kernel does without changing any result:
	 *   1. every wave first takes games off a cursor and runs their select stage (select_batch, exactly as k_solve<.., FUSED>), then
	 *      queues one work item per leaf the solver has to look at;
	 *   2. the waves take items off the queue and solve them against the game's table AS IT WAS BEFORE THE BATCH: the table is only
	 *      read, every bucket a task touches is copied into the task's overlay (dev_solver.hpp) and modified there;
	 *   3. the wave that finishes a game's last leaf commits the batch in order: a task whose first-touch bucket copies still equal the
	 *      table has seen exactly what it would have seen in its turn, so its versions of the buckets it wrote are stored; otherwise (an
	 *      earlier leaf of the same batch changed a bucket it read: 3-5 % of the leaves) it is solved again right there, serially,
	 *      on the table itself.  Then the game's positions go to the network queue.
	 * No wave ever waits for another wave's result (the commit is done by whoever arrives last), the only spinning is on an empty queue
	 * while other waves are still selecting, and selects are themselves taken off a cursor by resident waves — so the launch cannot
	 * deadlock whatever the grid size.
	 */
	template<class SH>
	__device__ __forceinline__ void spec_flush(const SH &sh, SpecTask &h, int nodes, uint32_t flags0, float moves_left0, int lane)
	{
		const int cnt = sh.ov_count;
		for (int i = lane; i < cnt; i += 64)
			h.keys[i] = sh.ov_keys[i];
		if (lane < OV_CAP / 32)
			h.dirty[lane] = sh.ov_dirty[lane];
		if (lane == 0)
		{
			h.count = cnt;
			h.overflow = (sh.ov_overflow || sh.error == ERR_OVERLAY) ? 1 : 0;
			h.nodes = nodes;
			h.flags0 = flags0;
			h.moves_left0 = moves_left0;
			h.solved = 1;
		}
	}
	/* The commit of one game's batch, resumable: the wave that solved the game's last leaf walks the leaves in batch order from `k` on.
	 * Returns the index of a leaf that has to be solved again serially (the caller does that, straight on the table, and calls again with
	 * k + 1), -1 when the batch is committed and on the network queue, or -2 when the game was DEFERRED: a leaf needs its serial re-run while
	 * all but a few games of the launch are done (the yield rule of k_solve, AgxEngineConfig.solver_yield_fraction) — the leaves before it
	 * stay committed, the game sits out this step's network / expand stages and the next launch queues its remaining leaves again: the
	 * same operations one step later. */
#ifndef AGX_DEFER_ON_EMPTY_QUEUE
#define AGX_DEFER_ON_EMPTY_QUEUE 0 /* 1: a batch whose commit needs a serial re-run is also deferred once the launch's queue has run dry (spec_commit_game) */
#endif
	struct SpecCommit
	{
		int game, k, first; // first: where the game's batch began in THIS launch (0, or the solve_pos a deferred game came back with)
			bool table_changed;
			unsigned long long nodes, solved, reruns;
	};
	template<class SH>
	__device__ __forceinline__ int spec_commit_game(SH &sh, const EngineDev &E, SpecCommit &c, int lane, const int *c_done, int defer_threshold, const int *c_queue, int count)
	{ // c_queue: the launch's counters (select cursor, queue head, queue tail, games selected)
		const int g = c.game;
		GameState &gs = E.games[g];
		const int n_tasks = gs.n_tasks;
		u64 *tt = E.tt + static_cast<size_t>(g % E.tt_mod) * (E.tt_bucket_mask + 1ull) * 8ull; // (double-buffered search: both buffers of a thread use its one solver)
		for (int k = c.k; k < n_tasks; k++)
		{
			const int slot = g * E.batch + k;
			SpecTask &h = E.spec_tasks[slot];
			if (h.solved == 0)
				continue; // a proven edge: the solver never saw this task
			DTask &t = E.tasks[slot];
			const u64 *ov = E.spec_overlay + static_cast<size_t>(slot) * (SPEC_OV_CAP * 16);
			const int cnt = h.count;
			bool valid = (h.overflow == 0);
			{ // a header that does not describe an overlay (count or a bucket index out of range) can only mean that the solving wave's stores were
			  // not visible here: stop the game with an error instead of following a wild index
				bool bad = (cnt < 0 || cnt > SPEC_OV_CAP);
				for (int i = lane; !bad && i < cnt; i += 64)
					bad = static_cast<u64>(h.keys[i]) > E.tt_bucket_mask;
				if (__ballot(bad) != 0ull)
				{
					if (lane == 0)
					{
						gs.error = ERR_SPEC_STATE;
						h.solved = 0;
					}
					continue;
				}
			}
			if (valid && c.table_changed)
			{ // has any bucket this task looked at changed since the batch began?  Eight lanes per bucket.
				bool mismatch = false;
				for (int base = 0; base < cnt; base += 8)
				{
					const int p = base + (lane >> 3), w = lane & 7;
					if (p < cnt)
						mismatch = mismatch || (ov[p * 16 + w] != tt[8 * static_cast<u64>(h.keys[p]) + w]);
				}
				valid = (__ballot(mismatch) == 0ull);
			}
			if (!valid)
			{
				int done = 0;
				if (lane == 0)
					done = __hip_atomic_load(c_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				// (like k_solve's rule: the first leaf of the launch is always finished — a game whose first pending leaf needs the re-run every time,
				//  e.g. one whose overlay overflows, would otherwise be deferred launch after launch with its speculative work thrown away)
#if AGX_DEFER_ON_EMPTY_QUEUE
				// ... or the launch has nothing left to hide a serial re-run behind: every game has queued its leaves and every leaf has been taken, so
				// the waves are leaving and a 1.1 ms re-run now only lengthens the launch (in-kernel trace, profiles/r05_search_launch_timeline.txt: the
				// games that end a launch are the ones whose commit began when the queue ran dry and needed one re-run)
				int dry = 0;
				if (lane == 0 && defer_threshold != 0x7FFFFFFF)
					dry = (__hip_atomic_load(c_queue + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= count
							&& __hip_atomic_load(c_queue + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= __hip_atomic_load(c_queue + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ? 1 : 0;
				const bool defer = (__builtin_amdgcn_readfirstlane(done) >= defer_threshold || __builtin_amdgcn_readfirstlane(dry) != 0) && k > c.first;
#else
				const bool defer = __builtin_amdgcn_readfirstlane(done) >= defer_threshold && k > c.first;
#endif
				// forget the speculative result: of this leaf (it is solved again), or of this leaf and the ones behind it (deferred)
				for (int j = k; j < (defer ? n_tasks : k + 1); j++)
				{
					SpecTask &hj = E.spec_tasks[g * E.batch + j];
					if (hj.solved != 0 && lane == 0)
					{
						DTask &tj = E.tasks[g * E.batch + j];
						tj.flags = hj.flags0;
						tj.win = 0.0f;
						tj.draw = 0.0f;
						tj.moves_left = hj.moves_left0;
						hj.solved = 0;
					}
				}
				wave_sync();
				if (defer)
				{
					if (lane == 0)
					{
						gs.solve_pos = k;
						gs.solve_pending = 1;
						gs.stats[5] += c.nodes;
						gs.spec_stats[0] += c.solved;
						gs.spec_stats[1] += c.reruns;
						gs.spec_stats[2] += 1;
					}
					return -2;
				}
				// Search::solve's own order: this task again, on the table as the earlier tasks of the batch left it
				c.solved++;
				c.reruns++;
				c.table_changed = true;
				c.k = k + 1;
				return k;
			}
			bool any = false;
			for (int base = 0; base < cnt; base += 8)
			{
				const int p = base + (lane >> 3), w = lane & 7;
				if (p < cnt && ((h.dirty[p >> 5] >> (p & 31)) & 1u))
				{
					tt[8 * static_cast<u64>(h.keys[p]) + w] = ov[p * 16 + 8 + w];
					any = true;
				}
			}
			c.table_changed = c.table_changed || (__ballot(any) != 0ull);
			c.nodes += static_cast<unsigned long long>(h.nodes);
			c.solved++;
			if (lane == 0)
				h.solved = 0;
			wave_sync();
		}
		const unsigned long long scheduled = schedule_to_nn(sh, E, g, gs, n_tasks, lane);
		if (lane == 0)
		{
			gs.solve_pos = 0;
			gs.solve_pending = 0;
			gs.stats[5] += c.nodes;
			gs.stats[1] += scheduled;
			gs.spec_stats[0] += c.solved;
			gs.spec_stats[1] += c.reruns;
#ifdef AGX_SPEC_PROFILE
			E.spec_trace[4 * g + 3] = c.reruns * 16 + c.solved;
#endif
		}
		return -1;
	}

#ifdef AGX_SPEC_PROFILE /* developer builds: where the waves of k_search_spec spend their time (100 MHz wall clock ticks, summed over waves) */
#define SPEC_T(var) unsigned long long var = wall_clock64()
#define SPEC_ADD(k, v) do { if (lane == 0) atomicAdd(&E.spec_prof[k], static_cast<unsigned long long>(v)); } while (0)
#define SPEC_MAX(k, v) do { if (lane == 0) atomicMax(&E.spec_prof[k], static_cast<unsigned long long>(v)); } while (0)
/* per wave of the LAST launch: launch-relative ticks of its select phase's end and of its exit, leaves it solved, ticks it waited for items */
#define SPEC_WAVE_EXIT() do { if (lane == 0) { unsigned long long *w_ = E.spec_trace + 4 * (static_cast<size_t>(E.n_games) + E.spec_group * E.spec_waves + blockIdx.x); \
		w_[0] = t_selected; w_[1] = wall_clock64(); w_[2] = wave_solves; w_[3] = wave_waited; } } while (0)
#else
#define SPEC_T(var) do { } while (0)
#define SPEC_ADD(k, v) do { } while (0)
#define SPEC_MAX(k, v) do { } while (0)
#define SPEC_WAVE_EXIT() do { } while (0)
#endif
#ifndef AGX_SPEC_WAVES
#define AGX_SPEC_WAVES 3 /* waves per SIMD the register allocation of k_search_spec leaves room for on boards whose LDS state allows 9-12 waves per compute unit */
#endif
#ifndef AGX_SPEC_WAVES_15
#define AGX_SPEC_WAVES_15 4 /* ... and on 15x15 boards (10 240 bytes of LDS state: 16 waves per compute unit): 128 registers.  Measured (profiles/r05_search_variants_ab.txt,
                               boxes 11-12): the spills of the tighter allocation sit outside the solver's loops (69 scratch instructions in the whole kernel) and cost
                               nothing — at 12 waves per unit the 128-register kernel is as fast as the 168-register one —, and with 16 the launch is 8.6 % shorter */
#endif
	constexpr int spec_waves_per_simd(int nfix) { return (nfix == 15) ? AGX_SPEC_WAVES_15 : AGX_SPEC_WAVES; }
	template<bool RENJU, int NFIX>
	__global__ __launch_bounds__(64, spec_waves_per_simd(NFIX)) void k_search_spec(EngineDev E, int count)
	{
		if (NFIX != 0)
		{
			E.n = NFIX;
			E.hw = NFIX * NFIX;
		}
		typedef SolverSharedT<(NFIX != 0) ? NFIX : MAXN> SH;
		__shared__ SH sh;
		const int lane = threadIdx.x;
		int *const c_select = E.counters + SPEC_COUNTER0 + 4 * E.spec_group, *const c_head = c_select + 1, *const c_tail = c_select + 2, *const c_selected = c_select + 3;
		int *const c_done = E.counters + E.yield_counter; // games of this launch whose batch is on the network queue (or that had nothing to do)
		const float fraction = E.match_mode ? 0.5f + 0.5f * E.yield_fraction : E.yield_fraction; // (match mode: half of the trees wait for their opponents)
		const int defer_threshold = (E.yield_fraction > 0.0f) ? static_cast<int>(fraction * count) : 0x7FFFFFFF;
		// Only the renju kernels park: their launches end with a few solves of several times the average length; on the other rules a launch ends with
		// commits and their re-runs (the yield rule's business), nothing was ever parked at this threshold (C2: 0-1 solves in 300 steps), and the code's
		// registers cost the launch 4 % (3.25 -> 3.40 ms).  For them the pointer below is a compile-time null and all of it folds away.
		constexpr bool SPEC_PARKS = RENJU;
		const ParkCtl park_ctl = { count, (E.park_fraction > 0.0f && count >= 4) ? static_cast<int>(E.park_fraction * count) : 0x7FFFFFFF };
		int *const items = E.spec_items + static_cast<size_t>(E.g0) * E.batch + static_cast<size_t>(E.spec_group) * SPEC_QUEUE_SLACK;
		const int item_cap = count * E.batch + SPEC_QUEUE_SLACK;
		const int area = E.n_games + E.spec_group * E.spec_waves + blockIdx.x; // this wave's spill areas (action stack, list / frame tails)

		/* ---- 1. select: games off a cursor ---- */
		static_assert(offsetof(SH, act) % 8 == 0 && offsetof(SH, frames) == offsetof(SH, act) + sizeof(sh.act) && offsetof(SH, ptype) == offsetof(SH, frames) + sizeof(sh.frames)
				&& offsetof(SH, threat) == offsetof(SH, ptype) + sizeof(sh.ptype) && offsetof(SH, items) == offsetof(SH, threat) + sizeof(sh.threat),
				"select-stage keys: 64-bit words over act + frames + ptype + threat + items (contiguous) and on over the per-solve scratch up to board");
		static_assert(offsetof(SH, board) - offsetof(SH, act) >= SH::SELECT_KEY_BYTES && offsetof(SH, board) > offsetof(SH, items) && sizeof(sh.lines) >= BWORDS * sizeof(u64),
				"select-stage LDS must fit");
		u64 *sel_keys = reinterpret_cast<u64*>(&sh.act[0]);
		bool keys_loaded = false;
		SPEC_T(t_begin);
		__builtin_amdgcn_s_setprio(3); // selects (and below: commits) are the launch's critical path, the speculative solves fill the SIMDs around them
		while (true)
		{
			int s = 0;
			if (lane == 0)
				s = atomicAdd(c_select, 1);
			s = __builtin_amdgcn_readfirstlane(s);
			if (s >= count)
				break;
			const int g = E.g0 + s;
			GameState &gs = E.games[g];
			// (tournament search: record g is a task buffer of tree 0, already filled by k_select — the threads take the tree in turn there; here
			//  its leaves are solved in parallel like any game's)
			const bool idle = (!gs.active || gs.error != 0 || gs.outcome != 0 || E.games[E.shared_tree ? 0 : g].grow_pending != 0);
			if (!idle)
			{
				if (!keys_loaded && !E.shared_tree)
				{
					for (int i = lane; i < 3 * (1 + E.hw); i += 64)
						sel_keys[i] = E.nc_keys[i];
					keys_loaded = true;
				}
				use_game_arenas(E, E.shared_tree ? 0 : g);
				const int first = gs.solve_pending ? gs.solve_pos : 0; // a deferred game: no new descents, its batch goes on behind the committed leaves
				if (!gs.solve_pending && !E.shared_tree)
					select_batch(E, g, g, lane, &sh.board[0], &sh.lines[0], sel_keys);
				__threadfence(); // the tasks are read by other waves
				__syncthreads();
				const int n_tasks = gs.n_tasks;
				int n_items = 0;
				for (int k = first; k < n_tasks; k++)
					if ((E.tasks[static_cast<size_t>(g) * E.batch + k].flags & TF_BY_SOLVER) == 0)
						n_items++;
				if (n_items == 0)
				{ // nothing for the solver (proven edges only): the batch goes to the network queue at once
					const unsigned long long scheduled = schedule_to_nn(sh, E, g, gs, n_tasks, lane);
					if (lane == 0)
					{
						gs.stats[1] += scheduled;
						gs.solve_pos = 0;
						gs.solve_pending = 0;
						atomicAdd(c_done, 1);
					}
					keys_loaded = keys_loaded && !E.use_symmetries; // (symmetric features pass through sh.act, where the keys are)
				}
				else
				{
					if (lane == 0)
					{
						__hip_atomic_store(&E.spec_left[g], n_items, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						__threadfence();
						const int base = atomicAdd(c_tail, n_items);
						int j = 0;
						for (int k = first; k < n_tasks; k++)
							if ((E.tasks[static_cast<size_t>(g) * E.batch + k].flags & TF_BY_SOLVER) == 0)
								__hip_atomic_store(&items[base + j++], (g * 16 + k) + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
					}
				}
			}
			if (idle && lane == 0)
			{
				if (gs.error != 0 && gs.solve_pending)
				{ // a game stopped for good with a batch in flight: its park buffers go back to the pool
					gs.solve_pending = 0;
					park_forget_game(E, g);
				}
				atomicAdd(c_done, 1);
			}
#ifdef AGX_SPEC_PROFILE
			if (lane == 0)
			{
				E.spec_trace[4 * g + 0] = wall_clock64();
				E.spec_trace[4 * g + 3] = 0;
			}
#endif
			if (lane == 0)
			{
				__threadfence();
				atomicAdd(c_selected, 1);
			}
			__syncthreads();
		}
		__syncthreads();

		__builtin_amdgcn_s_setprio(0);
		SPEC_T(t_selected);
		SPEC_ADD(0, t_selected - t_begin); // select phase, all waves (incl. the ones that found the cursor exhausted at once)
		SPEC_MAX(1, t_selected - t_begin); // the slowest wave's select phase
		/* ---- 2. + 3. solve the queued leaves, commit a game's batch when its last leaf is done ----
		 * One call site of the solver for both kinds of solves (speculative: against the task's overlay; re-run during a commit: straight on
		 * the table) — the solver is ~25 k instructions, a second inlined copy costs registers and instruction cache. */
		solver_load_threat_table(sh, E, lane);
		SpecCommit commit;
		commit.game = -1;
		SPEC_T(t_commit0);
#ifdef AGX_SPEC_PROFILE
		unsigned long long wave_solves = 0, wave_waited = 0;
#endif
		while (true)
		{
			int g, k;
			bool speculative;
			SPEC_T(t_pop);
			if (commit.game >= 0)
			{ // this wave is committing a game: go on until a leaf needs its serial re-run
				const int r = spec_commit_game(sh, E, commit, lane, c_done, defer_threshold, c_select, count);
				if (r < 0)
				{
					if (r == -1 && lane == 0)
						atomicAdd(c_done, 1);
#ifdef AGX_SPEC_PROFILE
					SPEC_T(t_c1);
					if (lane == 0)
					{
						E.spec_trace[4 * commit.game + 1] = t_commit0;
						E.spec_trace[4 * commit.game + 2] = t_c1;
					}
					SPEC_ADD(4, t_c1 - t_commit0); // commits incl. serial re-runs
					SPEC_MAX(5, t_c1 - t_commit0);
#endif
					commit.game = -1;
					__builtin_amdgcn_s_setprio(0);
					continue;
				}
				g = commit.game;
				k = r;
				speculative = false;
			}
			else
			{
				int i = 0;
				if (lane == 0)
					i = atomicAdd(c_head, 1);
				i = __builtin_amdgcn_readfirstlane(i);
				if (i >= item_cap)
				{
					SPEC_WAVE_EXIT();
					return;
				}
				int v = 0;
				if (lane == 0)
				{ // wait for queue slot i to be filled.  The waiting waves must not disturb the ones that still select (they are the critical path):
				  // each polls only its OWN slot (no shared hot spot), sleeps in between, and looks at the shared counters once in 32 polls
					for (int polls = 0;; polls++)
					{
						v = __hip_atomic_load(&items[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						if (v != 0)
							break;
						if ((polls & 31) == 31 && __hip_atomic_load(c_selected, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= count)
						{ // every game has queued its leaves: the queue's length is final
							__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
							if (i >= __hip_atomic_load(c_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
							{
								v = -1;
								break;
							}
						}
						if (polls > (1 << 22))
						{ // watchdog (seconds): a queue slot that never fills would hang the launch — report instead
							E.spec_watchdog[0] = 1;
							E.spec_watchdog[1] = i;
							E.spec_watchdog[2] = __hip_atomic_load(c_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							E.spec_watchdog[3] = __hip_atomic_load(c_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							E.spec_watchdog[4] = __hip_atomic_load(c_selected, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							E.spec_watchdog[5] = count;
							E.spec_watchdog[6] = __hip_atomic_load(c_select, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							E.games[E.g0].error = ERR_SPEC_STATE;
							v = -1;
							break;
						}
						__builtin_amdgcn_s_sleep(64);
					}
					if (v > 0)
						__hip_atomic_store(&items[i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // empty again for the next launch
				}
				v = __builtin_amdgcn_readfirstlane(v);
				SPEC_T(t_got);
				SPEC_ADD(2, t_got - t_pop); // waiting for an item
#ifdef AGX_SPEC_PROFILE
				wave_waited += t_got - t_pop;
#endif
				if (v < 0)
				{
					SPEC_MAX(7, t_got - t_begin); // the last wave's exit = the launch
					SPEC_WAVE_EXIT();
					return;
				}
				__threadfence(); // (acquire: the task as its selecting wave wrote it)
				g = (v - 1) >> 4;
				k = (v - 1) & 15;
				speculative = true;
			}
			const int slot = g * E.batch + k;
			DTask &t = E.tasks[slot];
			const uint32_t flags0 = t.flags;
			const float moves_left0 = t.moves_left;
			unsigned long long n1 = 0;
			SPEC_T(t_s0);
			const bool parked = solve_task<RENJU>(sh, E, g, t, slot, E.games[g].generation, lane, n1, area,
					speculative ? E.spec_overlay + static_cast<size_t>(slot) * (SPEC_OV_CAP * 16) : nullptr, (SPEC_PARKS && speculative) ? &park_ctl : nullptr);
			if (!speculative)
			{ // a re-run inside a commit: its result stands
				commit.nodes += n1;
				continue;
			}
			if (parked)
			{ // the solve goes on in the next launch (the queue is dry: this wave's next turn finds nothing and leaves)
				SPEC_T(t_parked);
				SPEC_ADD(3, t_parked - t_s0);
				continue;
			}
			spec_flush(sh, E.spec_tasks[slot], static_cast<int>(n1), flags0, moves_left0, lane);
			SPEC_T(t_solved);
			SPEC_ADD(3, t_solved - t_s0); // speculative solves
			SPEC_ADD(6, 1);
#ifdef AGX_SPEC_PROFILE
			wave_solves++;
#endif
			__threadfence(); // release: task results, features, overlay and its header
			int left = 0;
			if (lane == 0)
				left = __hip_atomic_fetch_add(&E.spec_left[g], -1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
			left = __builtin_amdgcn_readfirstlane(left);
			if (left == 1)
			{ // the last leaf of this game's batch: this wave commits it, in batch order (the loop's next turns)
				__threadfence();
				commit.game = g;
				commit.k = 0;
				commit.first = E.games[g].solve_pending ? E.games[g].solve_pos : 0;
				commit.table_changed = false;
				commit.nodes = commit.solved = commit.reruns = 0;
				__builtin_amdgcn_s_setprio(3);
#ifdef AGX_SPEC_PROFILE
				t_commit0 = wall_clock64();
#endif
			}
		}
	}

	/* ------------------------------------------------------------------------------------------------------------ */
#ifndef AGX_WIDE_BACKUP
#define AGX_WIDE_BACKUP 1 /* Tree::backup with all levels of a path in flight at once (k_expand); 0: level by level */
#endif
	__global__ __launch_bounds__(64) void k_expand(EngineDev E)
	{
		__shared__ float e_prior[MAXHW], e_win[MAXHW], e_draw[MAXHW];
		__shared__ uint16_t e_move[MAXHW], e_score[MAXHW];
		__shared__ float sh_sum;
		__shared__ u64 sort_keys[512]; // prune_weak_moves with max_children: (score band, prior, original index) of every edge
		const int g0 = E.g0 + blockIdx.x, lane = threadIdx.x;
		const int tg = E.shared_tree ? 0 : g0; // the game that owns the tree; tournament search: every search thread works on tree 0
		GameState &gs = E.games[tg];
		if (!gs.active || gs.error != 0 || gs.outcome != 0 || (!E.shared_tree && gs.solve_pending) || gs.grow_pending == 1 || gs.grow_pending == 3)
			return;
		use_game_arenas(E, tg);
		DNode *nodes = nodes_of(E, tg, gs.arena);
		DEdge *edges = edges_of(E, tg, gs.arena);
		int *ht = ht_of(E, tg);
		const int n = E.n, hw = E.hw;
		const int first_lane = E.shared_tree ? E.grp_first : g0, last_lane = E.shared_tree ? E.grp_first + E.grp_count - 1 : g0;
		if (E.shared_tree && gs.grow_pending != 0 && gs.grow_owner != E.grp_first)
			return; // double-buffered search: the buffer whose expansion waits for larger arenas goes first, this one keeps its leaves one more turn
		{
			/*
			 * NodeCache::resize / ObjectPool growth (NodeCache.cpp:320-355, utils/ObjectPool.hpp:74-289) for flat arenas: BEFORE anything is
			 * modified, the batch's worst case (a node per task, every solver edge kept) is held against the game's arenas.  If it does not
			 * fit, the game asks for the next size class and sits this stage out with its batch untouched: k_arena_service / k_arena_copy
			 * move the tree into larger regions, select and solve skip the game in the next step, and this kernel then expands the SAME
			 * batch — the game performs exactly the same sequence of operations, one step later.
			 */
			int need_edges = gs.n_edges, need_nodes = gs.n_nodes;
			for (int lg = first_lane; lg <= last_lane; lg++)
			{
				const int lane_tasks = E.games[lg].n_tasks;
				need_nodes += lane_tasks;
				for (int k = 0; k < lane_tasks; k++)
					need_edges += E.tasks[static_cast<size_t>(lg) * E.batch + k].n_edges;
			}
			const bool fits = need_nodes <= E.node_cap && need_edges <= E.edge_cap && 2 * need_nodes <= E.ht_cap;
			if (!fits && gs.arena_class + 1 < ARENA_CLASSES && gs.grow_pending != 4)
			{ // (also straight after a growth that was not enough: one more class)
				if (lane == 0)
				{
					gs.grow_pending = 1;
					gs.grow_owner = E.grp_first;
				}
				return;
			}
			if (lane == 0)
				gs.grow_pending = 0; // 2 -> 0: grown; 4 -> 0: no heap space left (or the largest class reached): the exact per-node test below decides
		}
		const u64 lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
		// tournament search: the search threads take the tree in thread order, each expands its batch and backs it up (SearchThread.cpp:
		// 135-141); self-play: the one buffer of the game
		for (int g = first_lane; g <= last_lane; g++)
		{
		GameState &ls = E.games[g];
		const int n_tasks = ls.n_tasks;
		unsigned long long wasted = 0;

		for (int k = 0; k < n_tasks; k++)
		{
			DTask &t = E.tasks[static_cast<size_t>(g) * E.batch + k];
			const int slot = g * E.batch + k;
			uint32_t flags = t.flags;
			const uint32_t score = t.score;
			float win = t.win, draw = t.draw, moves_left = t.moves_left;
			const int path_len = t.path_len;
			const int sign = t.sign_to_move;
			if (t.needs_nn)
			{ // NNEvaluator::unpack_from_network (NNEvaluator.cpp:263-286), 'pv' network: the q / moves-left outputs are zero
				flags |= TF_BY_NETWORK;
				win = E.nn_value[3 * slot];
				draw = E.nn_value[3 * slot + 1];
				if (s_unproven(score))
					moves_left = 0.0f;
			}
			int n_e = t.n_edges;
			if ((flags & TF_SKIP_EDGE_GENERATION) == 0 && n_e > 0)
			{ // UnifiedGenerator::generate (EdgeGenerator.cpp:269-303)
				const bool by_network = (flags & TF_BY_NETWORK) != 0;
				const int inv_symmetry = inverse_symmetry(t.symmetry);
				for (int i = lane; i < n_e; i += 64)
				{ // initialize_edges (:88-127)
					const uint32_t mv = t.emove[i], sc = t.escore[i];
					int pr, pc; // the network saw the position under t.symmetry: task policy = apply_symmetry(output, inverse) (NNEvaluator.cpp:277-279)
					symmetry_source(inv_symmetry, n, (mv >> 2) & 127, (mv >> 9) & 127, pr, pc);
					const int cell = pr * n + pc;
					e_move[i] = static_cast<uint16_t>(mv);
					e_score[i] = static_cast<uint16_t>(sc);
					e_prior[i] = by_network ? E.nn_policy[static_cast<size_t>(slot) * hw + cell] : 0.0f;
					float w = 0.0f, d = 0.0f;
					if (!by_network && s_proven(sc))
						s_to_value(sc, w, d);
					if (by_network && E.has_q)
					{ // unpack_from_network overwrites every action value with the 'q' output (NNEvaluator.cpp:277-279)
						w = E.nn_q[(static_cast<size_t>(slot) * hw + cell) * 2];
						d = E.nn_q[(static_cast<size_t>(slot) * hw + cell) * 2 + 1];
					}
					e_win[i] = w;
					e_draw[i] = d;
				}
				if (E.policy_temperature != 1.0f)
				{ // initialize_edges with a temperature (:90-118)
					if (E.policy_temperature == 0.0f)
					{ // prior 1 where the policy holds its maximum over the WHOLE plane (maxValue(task.getPolicy())), 0 elsewhere
						uint32_t mx = 0; // non-negative floats order like their bit patterns
						if (by_network)
							for (int i = lane; i < hw; i += 64)
								mx = max(mx, __float_as_uint(E.nn_policy[static_cast<size_t>(slot) * hw + i]));
						mx = wave_max_u32(mx);
						for (int i = lane; i < n_e; i += 64)
							e_prior[i] = (__float_as_uint(e_prior[i]) == mx) ? 1.0f : 0.0f;
					}
					else
					{
						const float inv_t = 1.0f / E.policy_temperature;
						for (int i = lane; i < n_e; i += 64)
						{
							const float p = e_prior[i];
							e_prior[i] = (p > 0.0f) ? static_cast<float>(det_exp(det_log(static_cast<double>(p)) * static_cast<double>(inv_t))) : 0.0f;
						}
					}
				}
				__syncthreads();
				if ((path_len > 0 || E.prune_root) && s_proven(score))
				{ // the root is exempt in self-play only (forceExpandRoot: GameGenerator.cpp:183-184, Player.cpp:111); prune_weak_moves, proven branch (:55-68): keep the best-scored edges in their original order
					uint32_t best = s_loss_in(0);
					for (int i = lane; i < n_e; i += 64)
						best = max(best, static_cast<uint32_t>(e_score[i]));
					best = wave_max_u32(best);
					int kept = 0;
					for (int base = 0; base < n_e; base += 64)
					{
						const int i = base + lane;
						const bool keep = (i < n_e) && (e_score[i] == best);
						const uint16_t mv = (i < n_e) ? e_move[i] : 0, sc = (i < n_e) ? e_score[i] : 0;
						const float p = (i < n_e) ? e_prior[i] : 0.0f, w = (i < n_e) ? e_win[i] : 0.0f, d = (i < n_e) ? e_draw[i] : 0.0f;
						const u64 m = __ballot(keep);
						__syncthreads();
						if (keep)
						{
							const int dst = kept + __popcll(m & lower);
							e_move[dst] = mv;
							e_score[dst] = sc;
							e_prior[dst] = p;
							e_win[dst] = w;
							e_draw[dst] = d;
						}
						kept += __popcll(m);
						__syncthreads();
					}
					n_e = kept;
				}
				else if ((path_len > 0 || E.prune_root) && n_e > E.max_children && (flags & TF_MUST_DEFEND) == 0)
				{ // prune_weak_moves, unproven branch (:69-83): the max_children best edges by EdgeComparator<MaxPolicyPrior>
				  // (Edge.hpp:156-172: proven scores first, then prior), then those whose prior reaches threshold * (their prior sum).
				  // std::partial_sort leaves the order of equal keys unspecified; here (and in the oracle) equal keys keep edge order.
					for (int i = lane; i < 512; i += 64)
					{
						u64 key = 0;
						if (i < n_e)
						{
							const uint32_t sc = e_score[i];
							const u64 band = s_proven(sc) ? sc : 0x4000u; // every unproven score shares one band between DRAW and WIN
							key = (band << 48) | (static_cast<u64>(__float_as_uint(e_prior[i])) << 16) | static_cast<u64>(0xFFFF - i);
						}
						sort_keys[i] = key;
					}
					__syncthreads();
					for (int k = 2; k <= 512; k <<= 1)
						for (int j = k >> 1; j > 0; j >>= 1)
						{ // bitonic network, descending
							for (int t = lane; t < 256; t += 64)
							{
								const int lo = ((t / j) * 2 * j) + (t % j), hi = lo + j;
								const u64 a = sort_keys[lo], b = sort_keys[hi];
								const bool descending = ((lo & k) == 0);
								if ((a < b) == descending)
								{
									sort_keys[lo] = b;
									sort_keys[hi] = a;
								}
							}
							__syncthreads();
						}
					const int K = E.max_children;
					if (lane == 0)
					{
						float sum = 0.0f;
						for (int r = 0; r < K; r++)
							sum += e_prior[0xFFFF - static_cast<int>(sort_keys[r] & 0xFFFFu)];
						sh_sum = E.expansion_threshold * sum;
					}
					__syncthreads();
					const float threshold = sh_sum;
					// gather the survivors (a prefix of the sorted list) into registers, then overwrite the edge arrays
					uint16_t g_mv[7], g_sc[7];
					float g_p[7], g_w[7], g_d[7];
					int kept = 0;
#pragma unroll
					for (int c = 0; c < 7; c++)
					{
						const int r = c * 64 + lane;
						const int src = (r < K) ? (0xFFFF - static_cast<int>(sort_keys[r] & 0xFFFFu)) : 0;
						g_mv[c] = e_move[src];
						g_sc[c] = e_score[src];
						g_p[c] = e_prior[src];
						g_w[c] = e_win[src];
						g_d[c] = e_draw[src];
						kept += __popcll(__ballot(r < K && g_p[c] >= threshold));
					}
					__syncthreads();
#pragma unroll
					for (int c = 0; c < 7; c++)
					{
						const int r = c * 64 + lane;
						if (r < kept)
						{
							e_move[r] = g_mv[c];
							e_score[r] = g_sc[c];
							e_prior[r] = g_p[c];
							e_win[r] = g_w[c];
							e_draw[r] = g_d[c];
						}
					}
					__syncthreads();
					n_e = kept;
				}
				// renormalize_policy (:23-40): the sum runs in edge order in fp32, exactly like the reference.  (As a loop of one lane over LDS every
				// addend was a full LDS round trip; now 64 priors travel into registers at once and the additions take them lane by lane: the same
				// additions in the same order, one v_readlane apiece, on every lane alike.)
				float sum = 0.0f;
				for (int base = 0; base < n_e; base += 64)
				{
					const int mine_i = __float_as_int((base + lane < n_e) ? e_prior[base + lane] : 0.0f);
					const int m = min(64, n_e - base);
					int j = 0;
					for (; j + 4 <= m; j += 4)
					{
						sum += __int_as_float(__builtin_amdgcn_readlane(mine_i, j));
						sum += __int_as_float(__builtin_amdgcn_readlane(mine_i, j + 1));
						sum += __int_as_float(__builtin_amdgcn_readlane(mine_i, j + 2));
						sum += __int_as_float(__builtin_amdgcn_readlane(mine_i, j + 3));
					}
					for (; j < m; j++)
						sum += __int_as_float(__builtin_amdgcn_readlane(mine_i, j));
				}
				if (sum == 0.0f)
				{
					const float u = 1.0f / n_e;
					for (int i = lane; i < n_e; i += 64)
						e_prior[i] = u;
				}
				else
				{
					const float inv = 1.0f / sum;
					for (int i = lane; i < n_e; i += 64)
						e_prior[i] = e_prior[i] * inv;
				}
				__syncthreads();
			}
			else
				n_e = 0;

			// ---- Tree::expand (Tree.cpp:257-298) ----
			int final_node = t.final_node;
			if (n_e > 0)
			{
				const int found = cache_seek(E, nodes, ht, t.hash, t.cboard, sign, lane);
				if (found < 0)
				{
					const int nid = gs.n_nodes, ebeg = gs.n_edges;
					if (nid >= E.node_cap || ebeg + n_e > E.edge_cap || 2 * (nid + 1) > E.ht_cap)
					{
						if (lane == 0)
							gs.error = (nid >= E.node_cap) ? ERR_NODE_CAPACITY : ((ebeg + n_e > E.edge_cap) ? ERR_EDGE_CAPACITY : ERR_HASH_TABLE);
						__syncthreads();
						return;
					}
					uint32_t max_score = 0;
					for (int i = lane; i < n_e; i += 64)
					{
						DEdge e;
						e.prior = e_prior[i];
						e.win = e_win[i];
						e.draw = e_draw[i];
						e.visits = 0;
						e.move = e_move[i];
						e.score = e_score[i];
						e.flag_vl = 0;
						e.pad = 0;
						edges[ebeg + i] = e;
						max_score = max(max_score, static_cast<uint32_t>(e_score[i]));
					}
					max_score = wave_max_u32(max_score);
					if (lane < BWORDS)
						nodes[nid].cboard[lane] = t.cboard[lane];
					if (lane == 0)
					{
						DNode &nd = nodes[nid];
						nd.edge_begin = ebeg;
						nd.n_edges = static_cast<int16_t>(n_e);
						nd.depth = static_cast<int16_t>(gs.n_moves + path_len);
						nd.vl = 0;
						nd.sign_to_move = static_cast<uint8_t>(sign);
						nd.pad8 = 0;
						nd.hash = t.hash;
						// updateValue from the cleared state (Node.hpp:268-274): visits 0 -> 1
						nd.visits = 1;
						const float tmp = static_cast<float>(1.0 / 1);
						float w = 0.0f + (win - 0.0f) * tmp, d = 0.0f + (draw - 0.0f) * tmp;
						nd.win = fmaxf(0.0f, fminf(1.0f, w));
						nd.draw = fmaxf(0.0f, fminf(1.0f, d));
						nd.moves_left = 0.0f + (moves_left - 0.0f) / 1;
						uint16_t nf = 0;
						if ((flags & TF_MUST_DEFEND) || (n_e + nd.depth) == hw)
							nf |= 4;
						nf |= ((flags & TF_STATICALLY_SOLVED) ? 8 : 0) | ((flags & TF_RECURSIVELY_SOLVED) ? 16 : 0) | ((flags & TF_MUST_DEFEND) ? 32 : 0);
						if (path_len == 0)
							nf |= 2;
						nd.flags = nf;
						nd.score = static_cast<uint16_t>(s_unknown(0));
						if ((nf & 4) || s_win(max_score) || s_unproven(max_score))
							nd.score = static_cast<uint16_t>(max_score);
						cache_insert(E, ht, t.hash, nid);
						gs.n_nodes = nid + 1;
						gs.n_edges = ebeg + n_e;
						if (path_len == 0)
							gs.root = nid;
					}
					final_node = nid;
					__syncthreads();
				}
				else
				{
					final_node = found;
					wasted++;
					if (path_len > 0)
					{
						const DEdge le = edges[t.path_edge[path_len - 1]];
						if (has_leak(E, le, true, nodes[found].score, nodes[found].win, nodes[found].draw))
							correct_information_leak(nodes, edges, t, path_len, final_node, lane);
					}
				}
			}

			if (lane == 0)
			{ // keep what the backup pass needs (Search::expand runs over ALL tasks before Search::backup, Search.cpp:214-232)
				t.final_node = final_node;
				t.win = win;
				t.draw = draw;
				t.moves_left = moves_left;
				t.flags = flags;
			}
			__syncthreads();
		}

		for (int k = 0; k < n_tasks; k++)
		{
			DTask &t = E.tasks[static_cast<size_t>(g) * E.batch + k];
			const float win = t.win, draw = t.draw;
			const int path_len = t.path_len, sign = t.sign_to_move, final_node = t.final_node;
			// ---- Tree::backup (Tree.cpp:299-351) ----
			// Every lane carries the same arithmetic (lane 0 stores it): a level costs one round trip for its node head + edge and one
			// for the score scan; the child's score travels up in a register instead of being read back.
			float ml = t.moves_left;
			bool have_child = final_node >= 0;
			uint32_t child_score = have_child ? nodes[final_node].score : 0u;
#if AGX_WIDE_BACKUP
			if (path_len <= 64)
			{ /* All levels of the path at once.  Level by level (below) a leaf costs two DEPENDENT round trips to HBM per level — the node head + edge,
			   * then the scan of the node's edge scores — and a path is ~8 levels deep: the expand stage was that chain (75 % of its wave cycles parked
			   * at s_waitcnt).  Nothing in a level's running means depends on another level (a position occurs once on a path), only the proven SCORE
			   * travels upwards: lane i takes level i — head, edge and value update —, the score scans of up to eight levels are requested together,
			   * and the score chain then runs over registers.  The arithmetic of every level is the serial loop's, operation for operation. */
				const bool mine = lane < path_len;
				const int node = mine ? t.path_node[lane] : 0, e = mine ? t.path_edge[lane] : 0;
				DNode nd;
				node_head(nd, nodes[node]);
				DEdge ed = edges[e];
				float ml_i = ml; // the serial loop adds 1.0f per level on its way up: the same additions, not one multiple
				for (int k = path_len - 1; k > lane; k--)
					ml_i += 1.0f;
				float vw = win, vd = draw;
				if (nd.sign_to_move != sign)
				{
					vw = 1.0f - (win + draw);
					vd = draw;
				}
				nd.visits++;
				const float tn = static_cast<float>(1.0 / nd.visits);
				nd.win = fmaxf(0.0f, fminf(1.0f, nd.win + (vw - nd.win) * tn));
				nd.draw = fmaxf(0.0f, fminf(1.0f, nd.draw + (vd - nd.draw) * tn));
				ed.visits++;
				const float te = 1.0f / ed.visits;
				ed.win = fmaxf(0.0f, fminf(1.0f, ed.win + (vw - ed.win) * te));
				ed.draw = fmaxf(0.0f, fminf(1.0f, ed.draw + (vd - ed.draw) * te));
				nd.moves_left += (ml_i - nd.moves_left) / nd.visits;
				nd.vl--;
				ed.flag_vl = static_cast<uint16_t>(((ed.flag_vl & 0x7FFF) - 1) & 0x7FFF);
				// the largest score among the OTHER edges of every level's node
				uint32_t others = 0;
				constexpr int LEVELS = 8, LOADS = (MAXHW + 63) / 64;
				for (int base = 0; base < path_len; base += LEVELS)
				{
					uint32_t sc[LEVELS][LOADS];
#pragma unroll
					for (int u = 0; u < LEVELS; u++)
					{
						const int lv = min(base + u, path_len - 1);
						const int eb = __builtin_amdgcn_readlane(nd.edge_begin, __builtin_amdgcn_readfirstlane(lv));
						const int ne = __builtin_amdgcn_readlane(static_cast<int>(nd.n_edges), __builtin_amdgcn_readfirstlane(lv));
						const int ee = __builtin_amdgcn_readlane(e, __builtin_amdgcn_readfirstlane(lv));
#pragma unroll
						for (int jj = 0; jj < LOADS; jj++)
						{
							const int j = lane + 64 * jj;
							sc[u][jj] = (base + u < path_len && j < ne && eb + j != ee) ? static_cast<uint32_t>(edges[eb + j].score) : 0u;
						}
					}
#pragma unroll
					for (int u = 0; u < LEVELS; u++)
					{
						uint32_t m = 0;
#pragma unroll
						for (int jj = 0; jj < LOADS; jj++)
							m = max(m, sc[u][jj]);
						m = wave_max_u32(m);
						if (lane == base + u)
							others = m;
					}
				}
				// the score chain, leaf to root, on registers
				for (int i = path_len - 1; i >= 0; i--)
				{
					const int li = __builtin_amdgcn_readfirstlane(i);
					uint32_t new_score = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ed.score), li));
					if (have_child)
						new_score = s_invert_up(child_score);
					const uint32_t result = max(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(others), li)), new_score);
					const uint32_t level_flags = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(nd.flags), li));
					uint32_t node_score = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(nd.score), li));
					if (((level_flags & 4u) != 0u) || s_win(result) || s_unproven(result))
						node_score = result;
					if (lane == i)
					{
						ed.score = static_cast<uint16_t>(new_score);
						nd.score = static_cast<uint16_t>(node_score);
					}
					child_score = node_score;
					have_child = true;
				}
				if (mine)
				{
					DNode &dn = nodes[node];
					dn.visits = nd.visits;
					dn.win = nd.win;
					dn.draw = nd.draw;
					dn.moves_left = nd.moves_left;
					dn.vl = nd.vl;
					dn.score = nd.score;
					edges[e] = ed;
				}
				__syncthreads();
				continue;
			}
#endif
			for (int i = path_len - 1; i >= 0; i--)
			{
				const int node = t.path_node[i], e = t.path_edge[i];
				DNode nd;
				node_head(nd, nodes[node]);
				DEdge ed = edges[e];
				float vw = win, vd = draw;
				if (nd.sign_to_move != sign)
				{
					vw = 1.0f - (win + draw);
					vd = draw;
				}
				nd.visits++;
				const float tn = static_cast<float>(1.0 / nd.visits); // Node::updateValue: reciprocal in double, narrowed
				nd.win = fmaxf(0.0f, fminf(1.0f, nd.win + (vw - nd.win) * tn));
				nd.draw = fmaxf(0.0f, fminf(1.0f, nd.draw + (vd - nd.draw) * tn));
				ed.visits++;
				const float te = 1.0f / ed.visits; // Edge::updateValue: fp32 reciprocal
				ed.win = fmaxf(0.0f, fminf(1.0f, ed.win + (vw - ed.win) * te));
				ed.draw = fmaxf(0.0f, fminf(1.0f, ed.draw + (vd - ed.draw) * te));
				nd.moves_left += (ml - nd.moves_left) / nd.visits;
				uint32_t new_score = ed.score;
				if (have_child)
					new_score = s_invert_up(child_score);
				ed.score = static_cast<uint16_t>(new_score);
				nd.vl--;
				ed.flag_vl = static_cast<uint16_t>(((ed.flag_vl & 0x7FFF) - 1) & 0x7FFF);
				// update_score(Node*) (Tree.cpp:93-104) on the registers
				uint32_t result = 0;
				for (int j = lane; j < nd.n_edges; j += 64)
					result = max(result, (nd.edge_begin + j == e) ? new_score : static_cast<uint32_t>(edges[nd.edge_begin + j].score));
				result = wave_max_u32(result);
				if (((nd.flags & 4) != 0) || s_win(result) || s_unproven(result))
					nd.score = static_cast<uint16_t>(result);
				if (lane == 0)
				{
					DNode &dn = nodes[node];
					dn.visits = nd.visits;
					dn.win = nd.win;
					dn.draw = nd.draw;
					dn.moves_left = nd.moves_left;
					dn.vl = nd.vl;
					dn.score = nd.score;
					edges[e] = ed;
				}
				child_score = nd.score;
				have_child = true;
				ml += 1.0f;
			}
			__syncthreads();
		}

		if (lane == 0)
		{
			ls.stats[0] += n_tasks;
			ls.stats[4] += wasted;
			ls.n_tasks = 0;
		}
		__syncthreads();
		} // search threads
		if (lane == 0)
		{
			gs.stats[10] = max(gs.stats[10], static_cast<unsigned long long>(gs.n_nodes));
			gs.stats[11] = max(gs.stats[11], static_cast<unsigned long long>(gs.n_edges));
			// move rule (GameGenerator.cpp:97-103, utils/misc.cpp:171-179)
			if (gs.root >= 0)
			{
				const DNode &r = nodes[gs.root];
				const float reduction = fmaxf(0.0f, fminf(1.0f, (r.draw - 0.75f) / (1.0f - 0.75f)));
				const int simulations = static_cast<int>(E.max_sims - reduction * (E.max_sims - 50));
				gs.need_move = (r.visits > simulations || s_proven(r.score)) ? 1 : 0;
			}
		}
	}

	/* ------------------------------------------------------------------------------------------------------------ */
	template<int TPB = 256>
	__device__ void block_reduce_xor(u64 &v, u64 *scratch, int tid)
	{ // scratch: TPB / 64 words
		v = wave_reduce_xor64(v);
		__syncthreads();
		if ((tid & 63) == 0)
			scratch[tid >> 6] = v;
		__syncthreads();
		v = 0;
		for (int w = 0; w < TPB / 64; w++)
			v ^= scratch[w];
		__syncthreads();
	}
	__device__ void clear_solver_table(const EngineDev &E, int g, int tid)
	{ // AlphaBetaSearch::clear (SharedHashTable.hpp:137-140)
		ulonglong2 *tt = reinterpret_cast<ulonglong2*>(E.tt + static_cast<size_t>(g % E.tt_mod) * (E.tt_bucket_mask + 1ull) * 8ull);
		const size_t entries = (E.tt_bucket_mask + 1ull) * 4ull;
		ulonglong2 empty;
		empty.x = 0ull;
		empty.y = tt_pack(0, 0, s_unknown(0), 0);
		for (size_t i = tid; i < entries; i += 256)
			tt[i] = empty;
	}
	__device__ void clear_tree_and_table(const EngineDev &E, int g, int tid)
	{ // Tree::clear + AlphaBetaSearch::clear (GameGenerator.cpp:52-53)
		int *ht = ht_of(E, g);
		for (int i = tid; i < E.games[g].ht_cap; i += 256)
			ht[i] = 0;
		clear_solver_table(E, g, tid);
	}
	/* loads opening `id` into game g: Game::loadOpening + prepare_search on an empty tree (GameGenerator.cpp:48-77,174-185) */
	__device__ void begin_game(const EngineDev &E, int g, int id, int tid, u64 *scratch, bool tables_cleared = false, const uint16_t *saved_moves = nullptr, int saved_count = 0)
	{ // saved_moves: a game in flight restored from a checkpoint (GameGenerator::load, GameGenerator.cpp:131-141) — its moves so far instead of opening `id`
		GameState &gs = E.games[g];
		if (!tables_cleared)
			clear_tree_and_table(E, g, tid);
		for (int i = tid; i < E.hw; i += 256)
			gs.board[i] = 0;
		if (tid < BWORDS)
			gs.cboard[tid] = 0;
		__syncthreads();
		const uint16_t *op = (saved_moves != nullptr) ? saved_moves - 1 : E.openings + static_cast<size_t>(id) * OPENING_CAP;
		const int count = (saved_moves != nullptr) ? saved_count : op[0];
		if (tid == 0)
		{
			int sign = 1;
			for (int i = 0; i < count; i++)
			{
				const uint32_t mv = op[1 + i];
				const int cell = ((mv >> 2) & 127) * E.n + ((mv >> 9) & 127);
				gs.board[cell] = static_cast<uint8_t>(mv & 3);
				gs.cboard[cell >> 5] |= static_cast<u64>(mv & 3) << (2 * (cell & 31));
				gs.moves[i] = static_cast<uint16_t>(mv);
				sign = 3 - static_cast<int>(mv & 3);
			}
			gs.sign_to_move = sign;
			gs.n_moves = count;
			gs.outcome = 0;
			gs.root = -1;
			gs.n_nodes = 0;
			gs.n_edges = 0;
			gs.n_tasks = 0;
			gs.need_move = 0;
			gs.solve_pos = 0;
			gs.solve_pending = 0;
			park_forget_game(E, g);
			gs.max_depth = 0;
			gs.nn_queued = 0;
			gs.restart_id = 0;
			gs.noise_ready = 0;
			gs.generation = (gs.generation + 1) % 64; // prepare_search -> increaseGeneration
			gs.opening_id = id;
			gs.active = 1;
		}
		__syncthreads();
		u64 h = 0;
		for (int i = tid; i < E.hw; i += 256)
			h ^= E.nc_keys[3 + 3 * i + gs.board[i]];
		block_reduce_xor(h, scratch, tid);
		if (tid == 0)
			gs.root_hash = h ^ E.nc_keys[gs.sign_to_move];
		__syncthreads();
	}

	/*
	 * prepare_search / Player::setBoard for tree t on the position in its GameState (GameGenerator.cpp:174-185, Player.cpp:98-110):
	 * NodeCache::cleanup (NodeCache.cpp:221-249) as keep-test + prefix sum + copy to the other arena, Search::setBoard
	 * (increaseGeneration), Tree::setBoard (root = seek(new board), Tree.cpp:146-149).  Whole workgroup of TPB threads.
	 */
	template<int TPB>
	__device__ void rebase_tree(const EngineDev &E0, int t, int tid, u64 *scratch, int *scan_nodes, int *scan_edges)
	{
		EngineDev E = E0;
		use_game_arenas(E, t); // (in match mode t is the partner's tree, not the workgroup's own game)
		GameState &gs = E.games[t];
		const int lane = tid & 63, wave = tid >> 6;
		DNode *nodes = nodes_of(E, t, gs.arena);
		DEdge *edges = edges_of(E, t, gs.arena);
		DNode *dst_nodes = nodes_of(E, t, gs.arena ^ 1);
		DEdge *dst_edges = edges_of(E, t, gs.arena ^ 1);
		int *ht = ht_of(E, t);
		for (int i = tid; i < E.ht_cap; i += TPB)
			ht[i] = 0;
		const int total = gs.n_nodes;
		int node_base = 0, edge_base = 0;
		__syncthreads();
		for (int base = 0; base < total; base += TPB)
		{
			const int i = base + tid;
			int keep = 0, ne = 0;
			if (i < total)
			{
				keep = 1;
				for (int w = 0; w < BWORDS; w++)
				{ // isTransitionPossibleFrom (NodeCache.cpp:95-115): every stone of the new position must be present
					const u64 from = gs.cboard[w], to = nodes[i].cboard[w];
					if (((from ^ to) & from) != 0)
						keep = 0;
				}
				ne = keep ? nodes[i].n_edges : 0;
			}
			scan_nodes[tid] = keep;
			scan_edges[tid] = ne;
			__syncthreads();
			for (int o = 1; o < TPB; o <<= 1)
			{ // inclusive Hillis-Steele scan
				const int a = (tid >= o) ? scan_nodes[tid - o] : 0, b = (tid >= o) ? scan_edges[tid - o] : 0;
				__syncthreads();
				scan_nodes[tid] += a;
				scan_edges[tid] += b;
				__syncthreads();
			}
			int cp_src = 0, cp_dst = 0;
			if (keep)
			{
				const int ni = node_base + scan_nodes[tid] - 1, eb = edge_base + scan_edges[tid] - ne;
				DNode nd = nodes[i];
				cp_src = nd.edge_begin;
				cp_dst = eb;
				nd.edge_begin = eb;
				dst_nodes[ni] = nd;
				// re-insert (atomic linear probing; slot order is irrelevant to lookups)
				const int mask = E.ht_cap - 1;
				int slot = static_cast<int>(nd.hash & static_cast<u64>(mask));
				while (atomicCAS(&ht[slot], 0, ni + 1) != 0)
					slot = (slot + 1) & mask;
			}
			// the edges: one node per wave at a time, its records copied by 64 lanes as 8-byte words (coalesced; 24-byte records of
			// per-thread copies touch three times the cache lines).  The node's (source, destination, count) come from its thread by
			// v_readlane — node k of the chunk belongs to lane k % 64 of wave k / 64.
			{
				const u64 *src64 = reinterpret_cast<const u64*>(edges);
				u64 *dst64 = reinterpret_cast<u64*>(dst_edges);
				for (int k = 0; k < 64; k++)
				{
					const int words = 3 * __builtin_amdgcn_readlane(ne, k);
					if (words == 0)
						continue;
					const size_t so = 3 * static_cast<size_t>(__builtin_amdgcn_readlane(cp_src, k)), dof = 3 * static_cast<size_t>(__builtin_amdgcn_readlane(cp_dst, k));
					for (int d = lane; d < words; d += 256)
					{
						const u64 w0 = src64[so + d];
						const u64 w1 = (d + 64 < words) ? src64[so + d + 64] : 0ull;
						const u64 w2 = (d + 128 < words) ? src64[so + d + 128] : 0ull;
						const u64 w3 = (d + 192 < words) ? src64[so + d + 192] : 0ull;
						dst64[dof + d] = w0;
						if (d + 64 < words)
							dst64[dof + d + 64] = w1;
						if (d + 128 < words)
							dst64[dof + d + 128] = w2;
						if (d + 192 < words)
							dst64[dof + d + 192] = w3;
					}
				}
			}
			node_base += scan_nodes[TPB - 1];
			edge_base += scan_edges[TPB - 1];
			__syncthreads();
		}
		u64 h = 0;
		for (int i = tid; i < E.hw; i += TPB)
			h ^= E.nc_keys[3 + 3 * i + gs.board[i]];
		block_reduce_xor<TPB>(h, scratch, tid);
		if (tid == 0)
		{
			gs.root_hash = h ^ E.nc_keys[gs.sign_to_move];
			gs.n_nodes = node_base;
			gs.n_edges = edge_base;
			gs.arena ^= 1;
			gs.generation = (gs.generation + 1) % 64;
			gs.max_depth = 0; // Tree.cpp:150
		}
		__syncthreads();
		if (wave == 0)
		{ // Tree::setBoard: root = seek(new board) (Tree.cpp:146-149)
			const int found = cache_seek(E, dst_nodes, ht, gs.root_hash, gs.cboard, gs.sign_to_move, lane);
			if (lane == 0)
			{
				gs.root = found;
				if (found >= 0)
					dst_nodes[found].flags |= 2;
			}
		}
		__syncthreads();
	}

	__global__ __launch_bounds__(256) void k_begin(EngineDev E)
	{
		__shared__ u64 scratch[4];
		const int g = blockIdx.x, tid = threadIdx.x;
		GameState &gs = E.games[g];
		__shared__ int sh_id;
		if (tid == 0)
		{
			sh_id = E.shared_tree ? 0 : g; // the first wave of games takes openings 0..n_games-1 in order; counters[1] is preset to n_games
			                               // (tournament search: every search thread starts on opening 0, counters[1] is preset to 1)
			gs.generation = 0;
			gs.error = 0;
			gs.games_done = 0;
			gs.arena = 0;
			for (int i = 0; i < 12; i++)
				gs.stats[i] = 0;
		}
		__syncthreads();
		if (E.match_mode)
		{ // every pair starts through k_assign_openings / k_match_restart: empty trees, the first players' trees ask for openings
			clear_tree_and_table(E, g, tid);
			if (tid == 0)
			{
				gs.active = 0;
				gs.root = -1;
				gs.n_nodes = 0;
				gs.n_edges = 0;
				gs.n_tasks = 0;
				gs.need_move = 0;
				gs.solve_pos = 0;
				gs.solve_pending = 0;
				park_forget_game(E, g);
				gs.outcome = 0;
				gs.n_moves = 0;
				gs.restart_id = (g < E.n_games / 2) ? -1 : 0;
				gs.match_score[0] = gs.match_score[1] = gs.match_score[2] = gs.match_score[3] = 0;
			}
		}
		else if (sh_id < E.n_openings)
			begin_game(E, g, sh_id, tid, scratch);
		else if (tid == 0)
		{
			gs.active = 0;
			gs.restart_id = -1; // waits for an opening like a finished game
		}
	}

	/* ADV_THREADS threads per game that must move: the compaction of a large tree (2000 nodes, 250 k edges: 6 MB) is a chain of dependent
	 * load -> store rounds per wave, so its duration falls with the number of waves that copy (256 -> 1024 threads: k_advance 0.36 -> see
	 * DESIGN 6.1); the other games' workgroups leave at once. */
	constexpr int ADV_THREADS = 1024;
	__global__ __launch_bounds__(ADV_THREADS) void k_advance(EngineDev E)
	{
		constexpr int TPB = ADV_THREADS, WAVES = ADV_THREADS / 64;
		__shared__ u64 scratch[WAVES];
		__shared__ float red_v[WAVES];
		__shared__ int red_i[WAVES];
		__shared__ int sh_int[8];
		__shared__ int scan_nodes[TPB], scan_edges[TPB];
		__shared__ uint16_t cell_edge[MAXHW], entry_cell[MAXHW]; // format-201 sample: edge of a cell (+1, bit 15 = visited or proven), cells with an entry
		__shared__ uint32_t sh_max[3];
		const int g = E.g0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
		GameState &gs = E.games[g];
		if (!gs.active || gs.error != 0 || !gs.need_move || gs.solve_pending || gs.grow_pending)
			return;
		use_game_arenas(E, g);
		DNode *nodes = nodes_of(E, g, gs.arena);
		DEdge *edges = edges_of(E, g, gs.arena);
		const int n = E.n;
		if (E.shared_tree && E.search_buffers > 1)
		{ // Search::cleanup (Search.cpp:233-242) of every thread before the move: the OTHER buffer's leaves — selected, solved, in the
		  // network or back from it — are dropped and their virtual losses taken back (SearchThread.cpp:108-109 after asynchronous_run's break)
			if (wave == 0)
				for (int r = 0; r < E.n_games; r++)
				{
					if (r >= E.grp_first && r < E.grp_first + E.grp_count)
						continue; // this buffer was expanded and backed up by k_expand just before
					GameState &os = E.games[r];
					const int pending = os.n_tasks;
					for (int k = 0; k < pending; k++)
					{
						const DTask &t = E.tasks[static_cast<size_t>(r) * E.batch + k];
						cancel_virtual_loss(nodes, edges, t, t.path_len, lane);
						wave_sync();
					}
					if (lane == 0)
						os.n_tasks = 0;
				}
			__threadfence_block();
			__syncthreads();
		}
		const DNode root = nodes[gs.root];

		// ---- final selector: "best" (EdgeSelector.cpp:515-536) or max_visit / min_visit / max_value / max_policy (:476-514) ----
		float best_value = -3.402823466e+38f;
		int best = 0x7FFFFFFF;
		for (int i = tid; i < root.n_edges; i += TPB)
		{
			const DEdge e = edges[root.edge_begin + i];
			float value;
			switch (E.final_selector)
			{
				default:
					switch (s_pv(e.score))
					{
						case 0:
							value = -1.0e8f + s_distance(e.score);
							break;
						case 3:
							value = +1.0e8f - s_distance(e.score);
							break;
						default:
							value = e.visits + (e.win + 0.5f * e.draw) * root.visits + 0.001f * e.prior;
							break;
					}
					break;
				case 1:
					value = e.visits;
					break;
				case 2:
					value = -e.visits;
					break;
				case 3:
					switch (s_pv(e.score))
					{
						case 0:
							value = -1000.0f + s_distance(e.score);
							break;
						case 1:
							value = 0.5f; // Value::draw().getExpectation()
							break;
						case 3:
							value = +1000.0f - s_distance(e.score);
							break;
						default:
							value = e.win + 0.5f * e.draw;
							break;
					}
					break;
				case 4:
					value = e.prior;
					break;
				case 5:
				{ // LCB (EdgeSelector.cpp:446-475): lower confidence bound on unproven edges, proven ones by distance
					const int pv = s_pv(e.score);
					if (pv == 0)
						value = -1.0e6f + s_distance(e.score) + e.prior;
					else if (pv == 3)
						value = +1.0e6f - s_distance(e.score) + e.prior;
					else
					{
						const int vl = e.flag_vl & 0x7FFF;
						const float visits = 1.0e-8f + e.visits;
						const float vl_factor = visits / (visits + static_cast<float>(vl));
						const float Q = (e.visits > 0) ? (e.win + 0.5f * e.draw) : (root.win + 0.5f * root.draw);
						const float parent_log_visit = static_cast<float>(det_log(static_cast<double>(root.visits + root.vl)));
						const float U = E.c_puct * sqrtf(parent_log_visit / (1.0f + e.visits + vl));
						value = Q * vl_factor - U;
					}
					break;
				}
			}
			if (value > best_value)
			{
				best_value = value;
				best = i;
			}
		}
		wave_argmax(best_value, best);
		if (lane == 0)
		{
			red_v[wave] = best_value;
			red_i[wave] = best;
		}
		__syncthreads();
		if (tid == 0)
		{
			for (int w = 1; w < WAVES; w++)
				if (red_v[w] > red_v[0] || (red_v[w] == red_v[0] && red_i[w] < red_i[0]))
				{
					red_v[0] = red_v[w];
					red_i[0] = red_i[w];
				}
			sh_int[4] = red_i[0];
		}
		__syncthreads();
		const int best_edge = sh_int[4];
		const uint32_t mv = edges[root.edge_begin + best_edge].move;

		// ---- the sample in dataset format 201 (SearchDataPack of the root -> SearchDataStorage_v201::loadFrom + serialize,
		//      dataset/data_packs.cpp:24-43, dataset/SearchDataStorage.cpp:326-374,410-419), quantised here so that 6 bytes per visited
		//      cell leave the device instead of 24 per root edge ----
		int n_entries = 0;
		if (E.record_format & 2)
		{
			for (int i = tid; i < E.hw; i += TPB)
				cell_edge[i] = 0;
			if (tid < 3)
				sh_max[tid] = 0u;
			__syncthreads();
			for (int i = tid; i < root.n_edges; i += TPB)
			{ // scatter the edges over the board; the maxima that become the three scales (non-negative floats order like their bits)
				const DEdge e = edges[root.edge_begin + i];
				const int cell = ((e.move >> 2) & 127) * n + ((e.move >> 9) & 127);
				const bool natural = e.visits > 0 || s_proven(e.score);
				cell_edge[cell] = static_cast<uint16_t>((i + 1) | (natural ? 0x8000 : 0));
				const float wd = fmaxf(e.win, e.draw);
				atomicMax(&sh_max[0], __float_as_uint((e.prior > 0.0f) ? e.prior : 0.0f));
				atomicMax(&sh_max[1], __float_as_uint((wd > 0.0f) ? wd : 0.0f));
				atomicMax(&sh_max[2], static_cast<uint32_t>(max(e.visits, 0)));
			}
			__syncthreads();
			// which cells get an entry: visited or proven ones, and any cell 255 or more past the previous entry (:333-337, in cell order)
			if (E.hw <= 255)
			{ // no cell can be 255 past anything: the entries are the flagged cells, compacted in order by the first wave
				if (wave == 0)
				{
					int count = 0;
					for (int base = 0; base < E.hw; base += 64)
					{
						const int i = base + lane;
						const bool flagged = (i < E.hw) && (cell_edge[i] & 0x8000);
						const u64 m = __ballot(flagged);
						if (flagged)
							entry_cell[count + __popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))))] = static_cast<uint16_t>(i);
						count += __popcll(m);
					}
					if (lane == 0)
						sh_int[5] = count;
				}
			}
			else if (tid == 0)
			{
				int last = 0, count = 0;
				for (int i = 0; i < E.hw; i++)
					if ((cell_edge[i] & 0x8000) || (i - last) >= 255)
					{
						entry_cell[count++] = static_cast<uint16_t>(i);
						last = i;
					}
				sh_int[5] = count;
			}
			__syncthreads();
			n_entries = sh_int[5];
		}
		if (tid == 0)
		{
			// ---- sample record (GameGenerator.cpp:166-171): the header is written whenever its slot exists, so that the host never
			//      reads a slot that was reserved but not filled; a part that does not fit is left out and stops the game with ERR_RECORDS ----
			const int rec = atomicAdd(&E.counters[3], 1);
			int eoff = -1, soff = -1;
			const int sbytes = v201::HEADER_BYTES + v201::ENTRY_BYTES * n_entries;
			bool fits = rec < E.record_cap;
			if (E.record_format & 1)
			{
				eoff = atomicAdd(&E.counters[4], static_cast<int>(root.n_edges));
				if (eoff + root.n_edges > E.record_edge_cap)
				{
					eoff = -1;
					fits = false;
				}
			}
			if (E.record_format & 2)
			{
				soff = atomicAdd(&E.counters[6], (sbytes + 3) & ~3);
				if (soff + sbytes > E.sample_cap)
				{
					soff = -1;
					fits = false;
				}
			}
			sh_int[0] = (rec < E.record_cap) ? rec : -1;
			sh_int[1] = eoff;
			sh_int[6] = soff;
			sh_int[7] = fits ? 1 : 0;
		}
		__syncthreads();
		const int rec = sh_int[0], eoff = sh_int[1], soff = sh_int[6];
		if (eoff >= 0)
			for (int i = tid; i < root.n_edges; i += TPB)
				E.record_edges[eoff + i] = edges[root.edge_begin + i];
		if (soff >= 0)
		{
			const float prior_scale = v201::prior_scale(__uint_as_float(sh_max[0])), value_scale = v201::value_scale(__uint_as_float(sh_max[1]));
			const float visit_scale = v201::visit_scale(fmaxf(1.0f, static_cast<float>(sh_max[2])));
			uint8_t *out = E.samples + soff;
			if (tid == 0)
			{
				uint16_t *h16 = reinterpret_cast<uint16_t*>(out);
				h16[0] = static_cast<uint16_t>(v201::ScaleFormat::encode(value_scale));
				h16[1] = static_cast<uint16_t>(v201::ScaleFormat::encode(prior_scale));
				h16[2] = static_cast<uint16_t>(v201::ScaleFormat::encode(visit_scale));
				h16[3] = root.score;
				h16[4] = static_cast<uint16_t>(gs.n_moves); // stones on the board the root stands for
				h16[5] = static_cast<uint16_t>((root.flags >> 3) & 7);
				*reinterpret_cast<uint32_t*>(out + 12) = static_cast<uint32_t>(n_entries);
			}
			for (int k = tid; k < n_entries; k += TPB)
			{
				const int cell = entry_cell[k], previous = (k > 0) ? entry_cell[k - 1] : 0;
				const int ei = (cell_edge[cell] & 0x7FFF) - 1;
				int visits = 0;
				float prior = 0.0f, win = 0.0f, draw = 0.0f;
				uint32_t score = s_unknown(0); // a filler cell without an edge: the SearchDataPack defaults (Score(), Value())
				if (ei >= 0)
				{
					const DEdge e = edges[root.edge_begin + ei];
					visits = e.visits;
					prior = e.prior;
					win = e.win;
					draw = e.draw;
					score = e.score;
				}
				uint8_t *q = out + v201::HEADER_BYTES + v201::ENTRY_BYTES * k;
				q[0] = static_cast<uint8_t>(cell - previous);
				q[1] = static_cast<uint8_t>(v201::VisitFormat::encode(static_cast<float>(visits) / visit_scale));
				q[2] = static_cast<uint8_t>(v201::PriorFormat::encode(prior / prior_scale));
				q[3] = static_cast<uint8_t>(v201::score_code(score));
				q[4] = static_cast<uint8_t>(v201::PriorFormat::encode(win / value_scale));
				q[5] = static_cast<uint8_t>(v201::PriorFormat::encode(draw / value_scale));
			}
		}
		if (tid == 0 && !sh_int[7])
			gs.error = ERR_RECORDS;

		// ---- Game::makeMove + getOutcome (Game.cpp:104-122, rules.cpp:110-133) ----
		const int s = mv & 3, r = (mv >> 2) & 127, c = (mv >> 9) & 127, cell = r * n + c;
		if (tid == 0)
		{
			gs.board[cell] = static_cast<uint8_t>(s);
			gs.cboard[cell >> 5] |= static_cast<u64>(s) << (2 * (cell & 31));
			gs.moves[gs.n_moves] = static_cast<uint16_t>(mv);
			gs.n_moves++;
			gs.sign_to_move = 3 - s;
			gs.need_move = 0;
			gs.noise_ready = 0; // prepare_search creates a fresh selector for the next move (GameGenerator.cpp:181-183)
			gs.stats[8]++;
			atomicAdd(&E.counters[5], 1);
			bool win = false;
			for (int d = 0; d < 4; d++)
			{
				uint32_t pattern = 0;
				for (int k = -5, shf = 0; k <= 5; k++, shf += 2)
				{
					const int rr = r + k * row_step(d), cc = c + k * col_step(d);
					uint32_t v = (rr >= 0 && rr < n && cc >= 0 && cc < n) ? gs.board[rr * n + cc] : 3u;
					if (k == 0)
						v = 0;
					pattern |= v << shf;
				}
				const uint8_t e = E.t_pattern[narrow(pattern)];
				if (((s == 1) ? (e & 15) : (e >> 4)) == 6)
					win = true;
			}
			int outcome = 0;
			if (win)
				outcome = (s == 1) ? 2 : 3;
			else if (E.rules == AGX_RENJU && s == 1 && renju_static_foul(E.t_pattern, E.t_threat, gs.board, n, cell))
				outcome = 3; // a foul of the cross player loses the game (rules.cpp:127-128)
			else if (gs.n_moves >= E.draw_after)
				outcome = 1;
			gs.outcome = outcome;
			sh_int[2] = outcome;
			sh_int[3] = -1;
			if (rec >= 0)
			{
				MoveRecordHeader h;
				h.game_serial = gs.opening_id;
				h.move_number = gs.n_moves - 1;
				h.move = static_cast<uint16_t>(mv);
				h.root_score = root.score;
				h.root_visits = root.visits;
				h.root_win = root.win;
				h.root_draw = root.draw;
				h.n_edges = (eoff >= 0) ? root.n_edges : 0;
				h.edge_offset = eoff;
				h.root_flags = (root.flags >> 3) & 7; // DNode flags 8 / 16 / 32
				h.game_slot = g;
				h.game_index = gs.games_done;
				h.sample_offset = soff;
				h.sample_bytes = (soff >= 0) ? v201::HEADER_BYTES + v201::ENTRY_BYTES * n_entries : 0;
				h.outcome = outcome;
				E.records[rec] = h;
			}
			if (outcome != 0)
			{ // game over (GameGenerator.cpp:104-114 -> GAME_NOT_STARTED): k_assign_openings / k_restart give the slot its next opening
				const int ge = atomicAdd(&E.counters[7], 1);
				if (ge < E.game_end_cap)
				{ // setOutcome + addMoves(game.getMoves()) (:107-108)
					GameEndRecord &r = E.game_ends[ge];
					r.game_serial = gs.opening_id;
					r.game_slot = g;
					r.game_index = gs.games_done;
					r.outcome = outcome;
					r.n_moves = gs.n_moves;
					for (int i = 0; i < gs.n_moves; i++)
						r.moves[i] = gs.moves[i];
				}
				else
					gs.error = ERR_RECORDS;
				gs.games_done++;
				atomicAdd(&E.counters[2], 1);
				gs.active = 0;
				gs.restart_id = -1;
			}
		}
		__syncthreads();
		if (E.match_mode)
		{ // EvaluationGame.cpp:126-143: the move goes into the shared game, the OTHER player gets setBoard; this tree is left as it
		  // is until its player's next turn (two plies on)
			const int p = (g + E.n_games / 2) % E.n_games;
			GameState &ps = E.games[p];
			for (int i = tid; i < E.hw; i += TPB)
				ps.board[i] = gs.board[i];
			if (tid < BWORDS)
				ps.cboard[tid] = gs.cboard[tid];
			if (tid == 0)
			{
				ps.moves[gs.n_moves - 1] = static_cast<uint16_t>(mv);
				ps.n_moves = gs.n_moves;
				ps.sign_to_move = gs.sign_to_move;
				ps.outcome = sh_int[2];
				gs.active = 0;
				if (sh_int[2] != 0)
				{ // both trees wait; the first player's tree carries the restart request of the pair
					ps.games_done++;
					GameState &lead = E.games[min(g, p)];
					GameState &part = E.games[max(g, p)];
					part.restart_id = 0;
					// a match is two games on one opening with the colours swapped (EvaluationGame.cpp:44-71): after the first game the
					// pair starts again from the same opening, after the second it waits for a new one
					lead.restart_id = (lead.games_done % 2 == 1) ? lead.opening_id + 1 : -1;
					const int first_won = (sh_int[2] == 2 && lead.my_sign == 1) || (sh_int[2] == 3 && lead.my_sign == 2);
					lead.match_score[(sh_int[2] == 1) ? 1 : (first_won ? 0 : 2)]++;
				}
				else
				{
					ps.active = 1;
					ps.need_move = 0;
					ps.noise_ready = 0;
				}
			}
			__syncthreads();
			if (sh_int[2] == 0)
				rebase_tree<TPB>(E, p, tid, scratch, scan_nodes, scan_edges);
			return;
		}
		if (E.shared_tree)
		{ // every SearchThread's Search sees the move: Search::setBoard -> AlphaBetaSearch::increaseGeneration; a finished game stops them all
			for (int t = 1 + tid; t < E.n_games; t += TPB)
			{
				GameState &ls = E.games[t];
				ls.generation = (ls.generation + 1) % 64;
				ls.outcome = sh_int[2];
				ls.n_moves = gs.n_moves;
				if (sh_int[2] != 0)
					ls.active = 0;
			}
		}
		if (sh_int[2] != 0)
			return;
		rebase_tree<TPB>(E, g, tid, scratch, scan_nodes, scan_edges);
	}

	/*
	 * Finished games take the next openings of the pool in GAME ORDER (not in the order their workgroups happen to finish), so a
	 * run is reproducible: one workgroup scans the games of the launch's range, numbers the waiting ones and reserves that many
	 * openings; games for which none is left keep waiting (agx_engine_add_openings can supply more).
	 */
	/* A finished game takes its next opening.  In the reference every GameGenerator draws its own openings (OpeningGenerator per generator,
	 * GameGenerator.cpp:46-77); here slot s of S takes openings s, s + S, s + 2 S, ... — a function of the slot and of how many games it has
	 * played, NOT of which slot finished first, so a pool plays the same games however it is sliced into groups and however their launches
	 * interleave on their streams.  A slot whose next opening is not in the list yet waits (agx_engine_add_openings).  counters[1] keeps the
	 * high-water mark (openings_taken). */
	__device__ __forceinline__ void assign_openings(const EngineDev &E, int count)
	{ // (1024 threads of one workgroup)
		const int slots = E.shared_tree ? 1 : (E.match_mode ? E.n_games / 2 : E.n_games);
		for (int i = threadIdx.x; i < count; i += 1024)
		{
			GameState &gs = E.games[E.g0 + i];
			if (gs.restart_id != -1)
				continue;
			// (match mode: a pair plays every opening twice, colours swapped — games_done counts games, the opening changes every other one)
			const int round = E.match_mode ? gs.games_done / 2 : gs.games_done;
			const long long id = static_cast<long long>(E.g0 + i) + static_cast<long long>(slots) * round;
			if (id < E.n_openings)
			{
				gs.restart_id = static_cast<int>(id) + 1;
				atomicMax(&E.counters[1], static_cast<int>(id) + 1);
			}
		}
	}
	__global__ __launch_bounds__(1024) void k_assign_openings(EngineDev E, int count)
	{
		assign_openings(E, count);
	}
	/* the workgroup that finishes a game's LAST part (of k_arena_copy, k_clear_tables) does what used to be the launch behind it: every part
	 * makes its writes visible (agent-scope fence), takes a ticket, and the holder of the last ticket knows that all the others are through */
	__device__ __forceinline__ bool last_part_of(int *counter, int parts)
	{
		__shared__ int sh_last;
		__threadfence();
		__syncthreads();
		if (threadIdx.x == 0)
		{
			const int ticket = __hip_atomic_fetch_add(counter, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
			sh_last = (ticket == parts - 1) ? 1 : 0;
			if (sh_last)
				__hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (ready for the game's next time)
		}
		__syncthreads();
		const bool last = sh_last != 0;
		if (last)
			__threadfence();
		return last;
	}
	/*
	 * Tree::clear + AlphaBetaSearch::clear for the games about to restart, spread over `parts` workgroups per game: a 4 Mi-entry
	 * solver table is 64 MB, which one workgroup takes a millisecond to fill — longer than every other kernel of the step but two.
	 * Match mode: both trees of a pair whose first player's tree carries the restart request; the node tables stay (Player keeps
	 * its tree).
	 */
	__global__ __launch_bounds__(256) void k_clear_tables(EngineDev E, int parts, int restart)
	{ // restart != 0 (self-play pools with a tree per game): the workgroup that finishes a game's last part starts its next game (k_restart's work)
		__shared__ u64 scratch[4];
		const int g = E.g0 + blockIdx.x / parts, part = blockIdx.x % parts, tid = threadIdx.x;
		const int asks = E.shared_tree ? 0 : (E.match_mode ? g % (E.n_games / 2) : g);
		if (E.games[asks].restart_id <= 0)
			return;
		ulonglong2 *tt = reinterpret_cast<ulonglong2*>(E.tt + static_cast<size_t>(g % E.tt_mod) * (E.tt_bucket_mask + 1ull) * 8ull);
		const size_t entries = (E.tt_bucket_mask + 1ull) * 4ull;
		const size_t per = (entries + parts - 1) / parts;
		const size_t end = (per * (part + 1) < entries) ? per * (part + 1) : entries;
		ulonglong2 empty;
		empty.x = 0ull;
		empty.y = tt_pack(0, 0, s_unknown(0), 0);
		for (size_t i = per * part + tid; i < end; i += 256)
			tt[i] = empty;
		if (!E.match_mode)
		{
			int *ht = ht_of(E, g);
			for (int i = part * 256 + tid; i < E.games[g].ht_cap; i += parts * 256)
				ht[i] = 0;
		}
		if (restart != 0 && last_part_of(E.parts_done + E.n_games + g, parts))
			begin_game(E, g, E.games[g].restart_id - 1, tid, scratch, true); // (every table of the game is empty and visible: last_part_of)
	}
	__global__ __launch_bounds__(256) void k_restart(EngineDev E)
	{
		__shared__ u64 scratch[4];
		const int g = E.g0 + blockIdx.x, tid = threadIdx.x;
		const int id = E.games[E.shared_tree ? 0 : g].restart_id; // (tournament search: thread 0's request counts for every search thread,
		if (id <= 0)                                              //  and thread 0 is restarted by a later launch than the others)
			return;
		__syncthreads();
		begin_game(E, g, id - 1, tid, scratch, true); // k_clear_tables ran just before
	}

	/*
	 * Match mode: one workgroup per pair.  EvaluationGame.cpp:77-106: both solvers are cleared, the opening is loaded into the
	 * shared game, the first player takes cross in the first game of a match and circle in the second, and only the player to
	 * move gets setBoard — on a tree that still holds its previous game (Player keeps its Tree; what the new position cannot reach
	 * is dropped by the cleanup inside setBoard).
	 */
	__global__ __launch_bounds__(256) void k_match_restart(EngineDev E)
	{
		__shared__ u64 scratch[4];
		__shared__ int scan_nodes[256], scan_edges[256];
		const int lead = blockIdx.x, part = lead + E.n_games / 2, tid = threadIdx.x;
		const int id = E.games[lead].restart_id - 1;
		if (id < 0)
			return;
		__syncthreads();
		const uint16_t *op = E.openings + static_cast<size_t>(id) * OPENING_CAP;
		const int count = op[0];
		for (int k = 0; k < 2; k++)
		{
			const int t = k ? part : lead;
			GameState &gs = E.games[t]; // (both solver tables were emptied by k_clear_tables just before)
			for (int i = tid; i < E.hw; i += 256)
				gs.board[i] = 0;
			if (tid < BWORDS)
				gs.cboard[tid] = 0;
			__syncthreads();
			if (tid == 0)
			{
				int sign = 1;
				for (int i = 0; i < count; i++)
				{
					const uint32_t mv = op[1 + i];
					const int cell = ((mv >> 2) & 127) * E.n + ((mv >> 9) & 127);
					gs.board[cell] = static_cast<uint8_t>(mv & 3);
					gs.cboard[cell >> 5] |= static_cast<u64>(mv & 3) << (2 * (cell & 31));
					gs.moves[i] = static_cast<uint16_t>(mv);
					sign = 3 - static_cast<int>(mv & 3);
				}
				gs.sign_to_move = sign;
				gs.n_moves = count;
				gs.outcome = 0;
				gs.n_tasks = 0;
				gs.need_move = 0;
				gs.solve_pos = 0;
				gs.solve_pending = 0;
				park_forget_game(E, t);
				gs.nn_queued = 0;
				gs.restart_id = 0;
				gs.noise_ready = 0;
				gs.opening_id = id;
				gs.active = 0;
				const int lead_sign = (E.games[lead].games_done % 2 == 0) ? 1 : 2;
				gs.my_sign = k ? 3 - lead_sign : lead_sign;
			}
			__syncthreads();
		}
		const int mover = (E.games[lead].sign_to_move == E.games[lead].my_sign) ? lead : part;
		rebase_tree<256>(E, mover, tid, scratch, scan_nodes, scan_edges);
		if (tid == 0)
			E.games[mover].active = 1;
	}

	/*
	 * Arena heap service, once per step after k_advance: ONE workgroup; its thread 0 is the only allocator, so the per-class free lists
	 * need no lock.  (1) games that ended with a grown bundle give it back and return to class 0 (self-play only: a match player keeps its
	 * tree); (2) games whose expand asked for room get a bundle of the next class reserved (copied over by k_arena_copy).
	 */
	/* The heap and its free lists are shared by the whole pool, and slices of a pool may be stepped on different streams: every access goes
	 * through agent-scope atomics under one spin lock (taken by a single thread of a workgroup, for a handful of words). */
	template<typename T>
	__device__ __forceinline__ T heap_load(T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
	template<typename T>
	__device__ __forceinline__ void heap_store(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
	__device__ void heap_lock(ArenaHeap *h)
	{
		int expected = 0;
		while (!__hip_atomic_compare_exchange_strong(&h->lock, &expected, 1, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
		{
			expected = 0;
			__builtin_amdgcn_s_sleep(2);
		}
	}
	__device__ void heap_unlock(ArenaHeap *h) { __hip_atomic_store(&h->lock, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
	__device__ void bundle_store(ArenaBundle *dst, const ArenaBundle &b)
	{
		heap_store(&dst->node_off[0], b.node_off[0]);
		heap_store(&dst->node_off[1], b.node_off[1]);
		heap_store(&dst->edge_off[0], b.edge_off[0]);
		heap_store(&dst->edge_off[1], b.edge_off[1]);
		heap_store(&dst->ht_off, b.ht_off);
	}
	__device__ bool arena_alloc(const EngineDev &E, int cls, ArenaBundle &out)
	{ // one calling thread per workgroup
		ArenaHeap *h = E.heap;
		bool ok = true;
		heap_lock(h);
		const int free_count = heap_load(&h->free_count[cls]);
		if (free_count > 0)
		{
			ArenaBundle *src = E.free_bundles + static_cast<size_t>(cls) * h->free_capacity + (free_count - 1);
			out.node_off[0] = heap_load(&src->node_off[0]);
			out.node_off[1] = heap_load(&src->node_off[1]);
			out.edge_off[0] = heap_load(&src->edge_off[0]);
			out.edge_off[1] = heap_load(&src->edge_off[1]);
			out.ht_off = heap_load(&src->ht_off);
			heap_store(&h->free_count[cls], free_count - 1);
		}
		else
		{
			const u64 nodes = static_cast<u64>(E.node_cap) << cls, edges = static_cast<u64>(E.edge_cap) << cls, ht = static_cast<u64>(E.ht_cap) << cls;
			const u64 nc = heap_load(&h->node_cursor), ec = heap_load(&h->edge_cursor), hc = heap_load(&h->ht_cursor);
			if (nc + 2 * nodes > h->node_total || ec + 2 * edges > h->edge_total || hc + ht > h->ht_total)
				ok = false;
			else
			{
				out.node_off[0] = nc;
				out.node_off[1] = nc + nodes;
				out.edge_off[0] = ec;
				out.edge_off[1] = ec + edges;
				out.ht_off = hc;
				heap_store(&h->node_cursor, nc + 2 * nodes);
				heap_store(&h->edge_cursor, ec + 2 * edges);
				heap_store(&h->ht_cursor, hc + ht);
			}
		}
		heap_unlock(h);
		return ok;
	}
	__device__ void arena_free(const EngineDev &E, int cls, const GameState &gs)
	{ // one calling thread per workgroup
		ArenaHeap *h = E.heap;
		ArenaBundle b;
		b.node_off[0] = gs.node_off[0];
		b.node_off[1] = gs.node_off[1];
		b.edge_off[0] = gs.edge_off[0];
		b.edge_off[1] = gs.edge_off[1];
		b.ht_off = gs.ht_off;
		heap_lock(h);
		const int free_count = heap_load(&h->free_count[cls]);
		if (free_count < h->free_capacity)
		{ // (the lists hold one entry per game and class: always true; a bundle that found no room would be leaked, never handed out twice)
			bundle_store(E.free_bundles + static_cast<size_t>(cls) * h->free_capacity + free_count, b);
			heap_store(&h->free_count[cls], free_count + 1);
		}
		heap_unlock(h);
	}
	__global__ __launch_bounds__(1024) void k_arena_service(EngineDev E, int count, int assign_count)
	{ // assign_count > 0: the finished games of the first assign_count take their next openings here as well (k_assign_openings' work, one launch less per cycle)
		__shared__ int sh_list[1024];
		__shared__ int sh_n;
		const int tid = threadIdx.x;
		for (int start = 0; start < count; start += 1024)
		{
			if (tid == 0)
				sh_n = 0;
			__syncthreads();
			const int i = start + tid;
			if (i < count)
			{
				const GameState &gs = E.games[E.g0 + i];
				const bool release = !E.match_mode && gs.restart_id == -1 && gs.arena_class > 0 && gs.grow_pending == 0;
				if (gs.grow_pending == 1 || release)
					sh_list[atomicAdd(&sh_n, 1)] = i;
			}
			__syncthreads();
			if (tid == 0)
			{
				// requests in game order (the lanes appended them in any order): results do not depend on it, the heap layout stays reproducible
				for (int a = 1; a < sh_n; a++)
					for (int b = a; b > 0 && sh_list[b - 1] > sh_list[b]; b--)
					{
						const int tmp = sh_list[b];
						sh_list[b] = sh_list[b - 1];
						sh_list[b - 1] = tmp;
					}
				for (int k = 0; k < sh_n; k++)
				{
					GameState &gs = E.games[E.g0 + sh_list[k]];
					if (gs.grow_pending == 1)
					{
						ArenaBundle nb;
						if (arena_alloc(E, gs.arena_class + 1, nb))
						{
							gs.new_node_off[0] = nb.node_off[0];
							gs.new_node_off[1] = nb.node_off[1];
							gs.new_edge_off[0] = nb.edge_off[0];
							gs.new_edge_off[1] = nb.edge_off[1];
							gs.new_ht_off = nb.ht_off;
							gs.grow_pending = 3;
							atomicAdd(&E.heap->grows, 1);
						}
						else
						{ // heap exhausted: the game carries on in its arenas; if the batch really overflows them, k_expand reports it
							gs.grow_pending = 4;
							atomicAdd(&E.heap->failures, 1);
						}
					}
					else
					{ // a finished game hands its grown bundle back and waits for its next opening in class-0 arenas
						ArenaBundle nb;
						if (arena_alloc(E, 0, nb))
						{
							arena_free(E, gs.arena_class, gs);
							gs.node_off[0] = nb.node_off[0];
							gs.node_off[1] = nb.node_off[1];
							gs.edge_off[0] = nb.edge_off[0];
							gs.edge_off[1] = nb.edge_off[1];
							gs.ht_off = nb.ht_off;
							gs.arena_class = 0;
							gs.node_cap = E.node_cap;
							gs.edge_cap = E.edge_cap;
							gs.ht_cap = E.ht_cap;
							atomicAdd(&E.heap->releases, 1);
						}
					}
				}
			}
			__syncthreads();
		}
		if (assign_count > 0)
			assign_openings(E, assign_count); // (behind the releases: a finished game gives its grown bundle back before it is handed its next opening)
	}
	/* moves a game's tree into the bundle k_arena_service reserved for it: nodes and edges of the active arena copied as they are (indices
	 * stay valid), the node-cache table rebuilt at its new size.  `parts` workgroups per game: the edges (a few MB for a large tree) are
	 * split among parts 1.., part 0 moves the nodes and rebuilds the table; the workgroup that finishes the game's last part switches the game over (arena_commit). */
	/* the old bundle goes onto the free list, the game continues in the new one (ONE acting thread: the heap lock must never be contended by lanes of one wave) */
	__device__ void arena_commit(const EngineDev &E, GameState &gs);
	__global__ __launch_bounds__(256) void k_arena_copy(EngineDev E, int parts)
	{
		const int g = E.g0 + blockIdx.x / parts, part = blockIdx.x % parts, tid = threadIdx.x;
		GameState &gs = E.games[g];
		if (gs.grow_pending != 3)
			return;
		if (part != 0)
		{ // 24-byte edge records copied as 16-byte words (arena regions start at multiples of the even class-0 capacity: 16-byte aligned)
			const uint4 *src_edges = reinterpret_cast<const uint4*>(edges_of(E, g, gs.arena));
			uint4 *dst_edges = reinterpret_cast<uint4*>(E.edges + gs.new_edge_off[gs.arena]);
			const size_t halves = 3 * static_cast<size_t>(gs.n_edges), words = halves / 2;
			for (size_t i = static_cast<size_t>(part - 1) * 256 + tid; i < words; i += static_cast<size_t>(parts - 1) * 256)
				dst_edges[i] = src_edges[i];
			if ((halves & 1) != 0 && part == 1 && tid == 0)
				reinterpret_cast<u64*>(dst_edges)[halves - 1] = reinterpret_cast<const u64*>(src_edges)[halves - 1];
		}
		else
		{
			const int new_ht_cap = E.ht_cap << (gs.arena_class + 1);
			const DNode *src_nodes = nodes_of(E, g, gs.arena);
			DNode *dst_nodes = E.nodes + gs.new_node_off[gs.arena];
			int *ht = E.ht + gs.new_ht_off;
			for (int i = tid; i < new_ht_cap; i += 256)
				ht[i] = 0;
			__syncthreads();
			const int mask = new_ht_cap - 1;
			for (int i = tid; i < gs.n_nodes; i += 256)
			{
				const DNode nd = src_nodes[i];
				dst_nodes[i] = nd;
				int slot = static_cast<int>(nd.hash & static_cast<u64>(mask));
				while (atomicCAS(&ht[slot], 0, i + 1) != 0)
					slot = (slot + 1) & mask;
			}
		}
		// (once a launch of its own, k_arena_commit: the game is switched over by whichever workgroup finishes its last part — the others only READ the game's state,
		//  and none of them is still running when it changes)
		if (last_part_of(E.parts_done + g, parts) && tid == 0)
			arena_commit(E, gs);
	}
	__device__ void arena_commit(const EngineDev &E, GameState &gs)
	{
		const int cls = gs.arena_class + 1;
		arena_free(E, gs.arena_class, gs);
		gs.node_off[0] = gs.new_node_off[0];
		gs.node_off[1] = gs.new_node_off[1];
		gs.edge_off[0] = gs.new_edge_off[0];
		gs.edge_off[1] = gs.new_edge_off[1];
		gs.ht_off = gs.new_ht_off;
		gs.arena_class = cls;
		gs.node_cap = E.node_cap << cls;
		gs.edge_cap = E.edge_cap << cls;
		gs.ht_cap = E.ht_cap << cls;
		gs.grow_count++;
		gs.grow_pending = 2;
	}

	/* Tree::setBoard(board, signToMove) + Search::setBoard + Search::cleanup for ONE game driven from outside (evaluation/Player.cpp:100-110):
	 * the position becomes the tree's base board, every cached state that can still be reached from it is kept (NodeCache::cleanup:
	 * compaction into the other arena, table rebuilt), the root is whatever the cache holds for it, abandoned tasks are dropped and the
	 * solver table ages by one generation (Tree.cpp:128-151, Search.cpp:112-115,233-242). */
	/* Search::cleanup (Search.cpp:233-242) from outside: the leaves selected but not expanded — of every task buffer that works on the tree —
	 * are dropped and their virtual losses taken back (Tree::cancelVirtualLoss, Tree.cpp:377-384).  One wave per tree. */
	__global__ __launch_bounds__(64) void k_cancel_pending(EngineDev E)
	{
		const int tg = E.shared_tree ? 0 : E.g0 + blockIdx.x, lane = threadIdx.x;
		GameState &gs = E.games[tg];
		use_game_arenas(E, tg);
		DNode *nodes = nodes_of(E, tg, gs.arena);
		DEdge *edges = edges_of(E, tg, gs.arena);
		const int first = E.shared_tree ? 0 : tg, last = E.shared_tree ? E.n_games - 1 : tg;
		for (int r = first; r <= last; r++)
		{
			GameState &os = E.games[r];
			const int pending = os.n_tasks;
			for (int k = 0; k < pending; k++)
			{
				const DTask &t = E.tasks[static_cast<size_t>(r) * E.batch + k];
				cancel_virtual_loss(nodes, edges, t, t.path_len, lane);
				wave_sync();
			}
			if (lane == 0)
			{
				os.n_tasks = 0;
				os.solve_pos = 0;
				os.solve_pending = 0;
				park_forget_game(E, r);
			}
		}
	}
	__global__ __launch_bounds__(256) void k_set_board(EngineDev E, int g, const uint8_t *board, int sign_to_move)
	{
		__shared__ u64 scratch[4];
		__shared__ int scan_nodes[256], scan_edges[256];
		const int tid = threadIdx.x;
		GameState &gs = E.games[g];
		for (int i = tid; i < E.hw; i += 256)
			gs.board[i] = board[i];
		if (tid < BWORDS)
			gs.cboard[tid] = 0;
		__syncthreads();
		if (tid == 0)
		{
			int stones = 0;
			for (int cell = 0; cell < E.hw; cell++)
				if (board[cell] != 0)
				{
					gs.cboard[cell >> 5] |= static_cast<u64>(board[cell] & 3) << (2 * (cell & 31));
					stones++;
				}
			gs.sign_to_move = sign_to_move;
			gs.n_moves = stones;
			gs.outcome = 0;
			gs.n_tasks = 0;
			gs.need_move = 0;
			gs.solve_pos = 0;
			gs.solve_pending = 0;
			park_forget_game(E, g);
			gs.grow_pending = 0;
			gs.noise_ready = 0;
			gs.restart_id = 0;
			gs.active = 1;
		}
		__syncthreads();
		rebase_tree<256>(E, g, tid, scratch, scan_nodes, scan_edges);
		if (E.shared_tree)
		{ // every task buffer of the tree follows (its solver table ages with the tree's, Search::setBoard -> increaseGeneration)
			__syncthreads();
			for (int t = 1 + tid; t < E.n_games; t += 256)
			{
				GameState &ls = E.games[t];
				ls.generation = (ls.generation + 1) % 64;
				ls.outcome = 0;
				ls.n_moves = gs.n_moves;
				ls.n_tasks = 0;
				ls.active = 1;
			}
		}
	}

	/* GameGenerator::load (GameGenerator.cpp:131-141) for pool slot g: the saved game (its moves so far) with an EMPTY tree and solver table — the
	 * reference's checkpoint holds the Game and the samples collected so far, the search starts again with prepare_search */
	__global__ __launch_bounds__(256) void k_restore_game(EngineDev E, int g, const uint16_t *moves, int count, int opening_id, int nn_queued)
	{
		__shared__ u64 scratch[4];
		if (threadIdx.x == 0)
			E.games[g].generation = 0; // a loaded GameGenerator owns a new AlphaBetaSearch: its table's generation counts from 0 again (SharedHashTable.hpp:126,
			                           // :155 — prepare_search makes it 1), and the ageing of entries (depth - (generation - entry's), modulo 64) follows that count
		__syncthreads();
		begin_game(E, g, opening_id, threadIdx.x, scratch, false, moves, count);
		if (threadIdx.x == 0)
			E.games[g].nn_queued = nn_queued;
	}

	/* what a search loop asks the tree between two steps (Tree::getSimulationCount / isRootProven / getNodeCount under the tree lock,
	 * SearchThread.cpp:181-199) */
	__global__ void k_root_summary(EngineDev E, int g, int *out)
	{
		const GameState &gs = E.games[g];
		int visits = 0, proven = 0;
		if (gs.root >= 0)
		{
			const DNode &r = nodes_of(E, g, gs.arena)[gs.root];
			visits = r.visits;
			proven = s_proven(r.score) ? 1 : 0;
		}
		out[0] = visits;
		out[1] = proven;
		out[2] = gs.n_nodes;
		out[3] = gs.error;
	}
	__global__ void k_reset_counter(int *counter, int *second)
	{
		*counter = 0;
		if (second != nullptr)
			*second = 0;
	}

	/* debug / test kernels --------------------------------------------------------------------------------------- */
	__global__ __launch_bounds__(64) void k_debug_load_tasks(EngineDev E, const uint8_t *boards, const int *signs, int count)
	{ // makes every game look as if one fresh leaf had been selected: task 0 = given position
		const int g = blockIdx.x, lane = threadIdx.x;
		if (g >= count)
			return;
		GameState &gs = E.games[g];
		DTask &t = E.tasks[static_cast<size_t>(g) * E.batch];
		for (int i = lane; i < E.hw; i += 64)
			t.board[i] = boards[static_cast<size_t>(g) * E.hw + i];
		if (lane == 0)
		{
			t.path_len = 1;
			t.final_node = -1;
			t.n_edges = 0;
			t.flags = 0;
			t.sign_to_move = signs[g];
			t.score = s_unknown(0);
			t.needs_nn = 0;
			gs.n_tasks = 1;
			gs.active = 1;
			gs.outcome = 0;
			gs.error = 0;
			gs.solve_pos = 0; // (gs.generation: the idle template's — 0 until agx_debug_new_generation)
			gs.solve_pending = 0;
			park_forget_game(E, g);
			gs.nn_queued = 0;
		}
	}
	__global__ __launch_bounds__(64) void k_debug_pattern_state(EngineDev E, const uint8_t *boards, const int *signs, const uint16_t *moves, int n_moves,
			uint8_t *ptypes, uint8_t *threats, int16_t *lists, int lists_stride)
	{
		__shared__ SolverShared sh;
		const int g = blockIdx.x, lane = threadIdx.x;
		if (lane == 0)
		{
			sh.spill_lists = E.list_spill + static_cast<size_t>(g) * 20 * MAXHW;
			sh.spill_frames = reinterpret_cast<Frame*>(E.frame_spill) + static_cast<size_t>(g) * MAX_FRAMES;
			sh.snap = E.snap_spill + static_cast<size_t>(g) * (E.hw + 2) * 64;
		}
		solver_load_threat_table(sh, E, lane);
		solver_set_board(sh, E, boards + static_cast<size_t>(g) * E.hw, signs[g], lane);
		__shared__ uint16_t done[512];
		__shared__ int n_done;
		if (lane == 0)
			n_done = 0;
		__syncthreads();
		for (int i = 0; i < n_moves; i++)
		{
			const uint32_t m = moves[static_cast<size_t>(g) * n_moves + i];
			if (m == 0xFFFFu)
				continue;
			if ((m & 3) == 0)
			{
				const uint32_t u = done[n_done - 1];
				solver_place(sh, E, u, false, lane);
				if (lane == 0)
					n_done--;
			}
			else
			{
				solver_place(sh, E, m, true, lane);
				if (lane == 0)
					done[n_done++] = static_cast<uint16_t>(m);
			}
			__syncthreads();
		}
		for (int i = lane; i < E.hw; i += 64)
		{
			for (int k = 0; k < 8; k++)
				ptypes[(static_cast<size_t>(g) * E.hw + i) * 8 + k] = sh.ptype[i][k];
			threats[(static_cast<size_t>(g) * E.hw + i) * 2] = sh.threat[i][0];
			threats[(static_cast<size_t>(g) * E.hw + i) * 2 + 1] = sh.threat[i][1];
		}
		if (lane == 0)
		{
			int16_t *out = lists + static_cast<size_t>(g) * lists_stride;
			int pos = 0;
			for (int s = 0; s < 2; s++)
				for (int t = 0; t < 10; t++)
				{
					out[pos++] = static_cast<int16_t>(sh.count[s][t]);
					if (t == 1)
					{ // HALF_OPEN_3: only the size is kept on the device; the cells are listed in row-major order (the caller compares them as a set)
						for (int cell = 0; cell < E.hw; cell++)
							if (sh.threat[cell][s] == 1)
							{
								out[pos++] = static_cast<int16_t>(cell / E.n);
								out[pos++] = static_cast<int16_t>(cell % E.n);
							}
						continue;
					}
					for (int k = 0; k < sh.count[s][t]; k++)
					{
						const int cell = (t == 0) ? 0 : static_cast<int>(list_get(sh, s, t, k));
						out[pos++] = static_cast<int16_t>(cell / E.n);
						out[pos++] = static_cast<int16_t>(cell % E.n);
					}
				}
			out[lists_stride - 1] = static_cast<int16_t>(pos);
		}
	}
}

/* ================================================================================================================ */
struct AgxEngine
{
		AgxEngineConfig cfg;
		EngineDev dev;
		std::vector<void*> allocations;
		unsigned long long device_bytes = 0; // sum of the allocations' sizes (agx_engine_device_bytes)
		std::vector<GameState> idle_games; // the pool before agx_engine_begin: every game idle, in its class-0 arena bundle
		std::vector<unsigned long long> debug_solve_nodes; // solver nodes per position of the last agx_debug_solve
		ArenaHeap idle_heap;
		std::vector<uint64_t> zobrist; // [2*hw][2]
		bool begun = false;
		// Search::select and the threat solver of a game in one launch (k_solve<.., FUSED>); AGX_FUSE_SELECT=0 keeps them as two launches
		// (separate k_select / k_solve times in agx_engine_kernel_timing and in profiles)
		bool fuse_select = true;
		// AgxEngineConfig.speculative_solver: select + solver as one persistent launch with the leaves of a batch solved in parallel (k_search_spec)
		uint8_t *board_staging = nullptr; // agx_engine_set_board: the caller's board on its way to the device
		uint16_t *moves_staging = nullptr; // agx_engine_restore_game: the saved move list on its way to the device
		bool sizing_only = false; // agx_engine_estimate_device_bytes: dev_alloc only adds up, nothing touches a device
		int sizing_cus = 0;
		int *summary_dev = nullptr, *summary_host = nullptr; // agx_engine_root_summary: four words on their way back (device, pinned host)
		bool speculative = false;
		bool external_moves = false; // agx_engine_set_board has been called: the caller makes the moves, no advance stage services the arenas
		int spec_waves = 0; // waves of that launch over the whole pool
		// optional per-kernel timing (agx_engine_kernel_timing): HIP events on the launch stream around every kernel of a step
		bool timing = false;
		std::vector<hipEvent_t> events;   // groups of (before, after) per kernel launch
		std::vector<int> event_kernel;    // kernel id of each pair: 0 select, 1 solve, 2 expand, 3 advance
		std::vector<hipEvent_t> free_events;
};

namespace
{
	hipEvent_t timing_event(AgxEngine *e)
	{
		hipEvent_t ev = nullptr;
		if (!e->free_events.empty())
		{
			ev = e->free_events.back();
			e->free_events.pop_back();
		}
		else
			(void) hipEventCreate(&ev);
		return ev;
	}
	struct KernelTimer
	{ // records an event pair around one kernel launch when timing is enabled
			AgxEngine *e;
			hipStream_t s;
			hipEvent_t a = nullptr;
			KernelTimer(AgxEngine *engine, hipStream_t stream, int kernel) :
					e(engine), s(stream)
			{
				if (e->timing)
				{
					a = timing_event(e);
					(void) hipEventRecord(a, s);
					e->event_kernel.push_back(kernel);
				}
			}
			~KernelTimer()
			{
				if (e->timing)
				{
					hipEvent_t b = timing_event(e);
					(void) hipEventRecord(b, s);
					e->events.push_back(a);
					e->events.push_back(b);
				}
			}
	};
	template<typename T>
	int dev_alloc(AgxEngine *e, T **ptr, size_t count)
	{
		if (e->sizing_only)
		{
			e->device_bytes += std::max<size_t>(count * sizeof(T), 16);
			*ptr = nullptr;
			return AGX_OK;
		}
		void *p = nullptr;
		const hipError_t err = hipMalloc(&p, std::max<size_t>(count * sizeof(T), 16));
		if (err != hipSuccess)
		{
			(void) hipGetLastError(); // the failure must not stay behind as the "last error" of this thread's later, successful calls
			agx::set_error("hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(err));
			return AGX_ERR_HIP;
		}
		e->allocations.push_back(p);
		e->device_bytes += std::max<size_t>(count * sizeof(T), 16);
		*ptr = static_cast<T*>(p);
		return AGX_OK;
	}
	template<typename T>
	int dev_upload(AgxEngine *e, const T **ptr, const std::vector<T> &src)
	{
		T *p = nullptr;
		const int st = dev_alloc(e, &p, src.size());
		if (st != AGX_OK)
			return st;
		if (!e->sizing_only)
			AGX_HIP_CHECK(hipMemcpy(p, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
		*ptr = p;
		return AGX_OK;
	}
	uint64_t splitmix64(uint64_t &state)
	{
		uint64_t z = (state += 0x9E3779B97F4A7C15ull);
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		return z ^ (z >> 31);
	}
	size_t round_pow2(size_t x)
	{
		size_t r = 1;
		while (r < x)
			r <<= 1;
		return r;
	}
}

extern "C" {

int agx_engine_default_config(AgxEngineConfig *cfg)
{
	AGX_REQUIRE(cfg != nullptr, AGX_ERR_INVALID, "agx_engine_default_config: null argument");
	std::memset(cfg, 0, sizeof(*cfg));
	cfg->rules = AGX_FREESTYLE;
	cfg->board_size = 15;
	cfg->draw_after = 225;
	cfg->n_games = 1024;
	cfg->max_batch_size = 8;
	cfg->max_simulations = 400;
	cfg->exploration_constant = 1.25f;
	cfg->exploration_scaling = 0.0f;
	cfg->init_to = 0;
	cfg->information_leak_threshold = 0.01f;
	cfg->policy_expansion_threshold = 1.0e-4f;
	cfg->max_children = 0;
	cfg->tss_max_positions = 100;
	cfg->tss_table_entries = 4ull * 1024ull * 1024ull;
	cfg->zobrist_seed = 0x9E3779B97F4A7C15ull;
	cfg->node_capacity = 8192;
	cfg->edge_capacity = 524288; // whole freestyle games at 400 playouts peak at ~250 k edges per game with an untrained (flat) policy
	cfg->record_capacity = 0;
	cfg->record_edge_capacity = 0;
	cfg->solver_yield_fraction = 0.0f;
	cfg->final_selector = 0;
	cfg->use_symmetries = 0;
	cfg->symmetry_seed = 0x5DEECE66Dull;
	cfg->action_values = 0;
	cfg->match_mode = 0;
	cfg->policy_temperature = 1.0f;
	cfg->arena_reserve = 1.0f;
	cfg->search_threads = 0;
	cfg->record_format = 1;
	cfg->record_sample_capacity = 0;
	cfg->game_end_capacity = 0;
	cfg->speculative_solver = 0;
	cfg->speculative_waves = 0;
	cfg->force_expand_root = 1;
	cfg->search_buffers = 0;
	cfg->noise_type = 0;
	cfg->noise_weight = 0.0f;
	cfg->noise_seed = 0x2545F4914F6CDD1Dull;
	return AGX_OK;
}

#if defined(AGX_QUICK) && !defined(AGX_QUICK_RENJU)
#define AGX_QUICK_RENJU false /* developer builds (AGX_QUICK: only the 15x15 solver is instantiated); -DAGX_QUICK_RENJU=true: the renju solver instead of the other rules' */
#endif
/* the instantiation of k_search_spec a pool of these rules and board size launches */
typedef void (*SpecKernel)(EngineDev, int);
static SpecKernel spec_kernel(int rules, int n)
{
#ifdef AGX_QUICK
	(void) rules;
	(void) n;
	return k_search_spec<AGX_QUICK_RENJU, 15>;
#else
	if (rules == AGX_RENJU)
		return (n == 15) ? k_search_spec<true, 15> : k_search_spec<true, 0>;
	if (n == 15)
		return k_search_spec<false, 15>;
	return (n == 20) ? k_search_spec<false, 20> : k_search_spec<false, 0>;
#endif
}
/* how many of its one-wave workgroups a compute unit keeps resident (LDS state and registers of that instantiation; asked of the runtime, 12 if it will not say) */
static int spec_resident_waves_per_cu(int rules, int n)
{
	int blocks = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, reinterpret_cast<const void*>(spec_kernel(rules, n)), 64, 0) != hipSuccess || blocks <= 0)
	{
		(void) hipGetLastError();
		return 12;
	}
	return blocks;
}
/* sizing_cus > 0: nothing is allocated and no device is touched — the engine only adds up what it WOULD allocate on a device with that many
 * compute units (agx_engine_estimate_device_bytes); *out then is a host-only object for the caller to read device_bytes from and delete */
static int engine_create(const AgxEngineConfig *cfg, AgxEngine **out, int sizing_cus)
{
	AGX_REQUIRE(cfg != nullptr && out != nullptr, AGX_ERR_INVALID, "agx_engine_create: null argument");
	AGX_REQUIRE(cfg->rules >= 0 && cfg->rules <= AGX_CARO6, AGX_ERR_INVALID, "agx_engine_create: unknown rules %d", cfg->rules);
	AGX_REQUIRE(cfg->board_size >= 5 && cfg->board_size <= MAXN, AGX_ERR_UNSUPPORTED, "agx_engine_create: board size %d not in [5, %d]", cfg->board_size, MAXN);
	AGX_REQUIRE(cfg->n_games > 0 && cfg->max_batch_size > 0 && cfg->max_simulations > 0, AGX_ERR_INVALID, "agx_engine_create: non-positive sizes");
	AGX_REQUIRE(cfg->tss_max_positions >= 1 && cfg->tss_max_positions <= 1000, AGX_ERR_UNSUPPORTED, "agx_engine_create: tss_max_positions must be in [1, 1000]");
	AGX_REQUIRE(cfg->init_to >= 0 && cfg->init_to <= 3, AGX_ERR_INVALID, "agx_engine_create: init_to must be 0..3");
	AGX_REQUIRE(cfg->solver_yield_fraction >= 0.0f && cfg->solver_yield_fraction <= 1.0f, AGX_ERR_INVALID, "agx_engine_create: solver_yield_fraction must be in [0, 1]");
	AGX_REQUIRE(cfg->final_selector >= 0 && cfg->final_selector <= 5, AGX_ERR_INVALID, "agx_engine_create: final_selector must be 0..5");
	AGX_REQUIRE(cfg->noise_type >= 0 && cfg->noise_type <= 3, AGX_ERR_INVALID, "agx_engine_create: noise_type must be 0 (none), 1 (custom), 2 (dirichlet) or 3 (gumbel)");
	AGX_REQUIRE(cfg->policy_temperature >= 0.0f, AGX_ERR_INVALID, "agx_engine_create: policy_temperature must not be negative");
	AGX_REQUIRE(!cfg->match_mode || cfg->n_games % 2 == 0, AGX_ERR_INVALID, "agx_engine_create: match_mode pairs the trees, n_games must be even");
	AGX_REQUIRE(cfg->noise_weight >= 0.0f && cfg->noise_weight <= 1.0f, AGX_ERR_INVALID, "agx_engine_create: noise_weight must be in [0, 1]");
	AGX_REQUIRE(cfg->search_buffers >= 0 && cfg->search_buffers <= 2, AGX_ERR_INVALID, "agx_engine_create: search_buffers must be 0, 1 or 2");
	{
		const int threads = std::max(1, cfg->search_threads), buffers = (cfg->search_buffers == 2) ? 2 : 1;
		AGX_REQUIRE((threads == 1 && buffers == 1) || (threads * buffers == cfg->n_games && !cfg->match_mode && cfg->solver_yield_fraction == 0.0f), AGX_ERR_INVALID,
				"agx_engine_create: search_threads > 1 / search_buffers = 2 make the pool ONE tree searched by n_games task buffers: n_games must equal "
				"search_threads x buffers (%d x %d), no match_mode, no yielding", threads, buffers);
	}
	AGX_REQUIRE(cfg->record_format >= 0 && cfg->record_format <= 3, AGX_ERR_INVALID, "agx_engine_create: record_format must be 0..3 (bit 0 edge snapshots, bit 1 format-201 samples)");

	AgxEngine *e = new AgxEngine();
	e->cfg = *cfg;
	e->sizing_only = sizing_cus > 0;
	e->sizing_cus = sizing_cus;
	if (const char *fuse = std::getenv("AGX_FUSE_SELECT"))
		e->fuse_select = std::atoi(fuse) != 0;
	EngineDev &d = e->dev;
	std::memset(&d, 0, sizeof(d));
	d.rules = cfg->rules;
	d.n = cfg->board_size;
	d.hw = d.n * d.n;
	d.draw_after = (cfg->draw_after > 0) ? cfg->draw_after : d.hw;
	d.n_games = cfg->n_games;
	d.batch = cfg->max_batch_size;
	d.max_sims = cfg->max_simulations;
	d.c_puct = cfg->exploration_constant;
	d.c_scale = cfg->exploration_scaling;
	d.init_to = cfg->init_to;
	d.leak_threshold = cfg->information_leak_threshold;
	d.expansion_threshold = cfg->policy_expansion_threshold;
	d.max_children = (cfg->max_children > 0) ? cfg->max_children : 0x7FFFFFFF;
	d.tss_max_nodes = cfg->tss_max_positions;
	d.tss_max_depth = 100;
	d.solve_time_ticks = 0ull;
	d.batch_limit = d.batch;
	d.yield_fraction = cfg->solver_yield_fraction;
	d.final_selector = cfg->final_selector;
	d.use_symmetries = cfg->use_symmetries;
	d.symmetry_seed = cfg->symmetry_seed;
	d.noise_type = (cfg->noise_weight > 0.0f) ? cfg->noise_type : 0;
	d.noise_weight = cfg->noise_weight;
	d.noise_seed = cfg->noise_seed;
	const size_t buckets = round_pow2(std::max<size_t>(cfg->tss_table_entries, 4)) / 4;
	d.tt_bucket_mask = buckets - 1;
	d.zobrist_seed = cfg->zobrist_seed;
	d.node_cap = cfg->node_capacity > 0 ? cfg->node_capacity : 8192;
	d.edge_cap = cfg->edge_capacity > 0 ? cfg->edge_capacity : 524288;
	d.edge_cap += d.edge_cap & 1; // even: every arena region then starts on a 16-byte boundary (k_arena_copy moves edges as 16-byte words)
	d.ht_cap = static_cast<int>(round_pow2(4 * static_cast<size_t>(d.node_cap)));
	d.act_cap = d.hw * (d.hw + 1) / 2 + 64;
	d.record_cap = cfg->record_capacity > 0 ? cfg->record_capacity : d.n_games * d.hw;
	d.record_edge_cap = cfg->record_edge_capacity > 0 ? cfg->record_edge_capacity : d.record_cap * 64;
	d.record_format = cfg->record_format;
	d.sample_cap = cfg->record_sample_capacity > 0 ? cfg->record_sample_capacity
			: static_cast<int>(std::min<size_t>(static_cast<size_t>(d.record_cap) * (16 + 6 * static_cast<size_t>(d.hw) + 2), 0x7FFFFFF0u)); // every cell an entry
	d.game_end_cap = cfg->game_end_capacity > 0 ? cfg->game_end_capacity : std::max(2 * d.n_games, d.record_cap / 16); // the record pool fills first

	HostTables tables;
	build_host_tables(cfg->rules, tables);
	std::vector<uint64_t> nc_keys(3 + 3 * d.hw);
	uint64_t st = cfg->zobrist_seed ^ 0x5851F42D4C957F2Dull;
	for (auto &k : nc_keys)
		k = splitmix64(st);
	e->zobrist.resize(4 * d.hw);
	st = cfg->zobrist_seed;
	for (auto &k : e->zobrist)
		k = splitmix64(st);

	int status = AGX_OK;
	const size_t G = d.n_games;
#define AGX_TRY(expr) if (status == AGX_OK) status = (expr)
	AGX_TRY(dev_alloc(e, &d.games, G));
	// tree arenas: pool-wide heaps; every game starts with a class-0 bundle (2 x node_cap nodes, 2 x edge_cap edges, ht_cap table slots),
	// the reserve behind them feeds the games that outgrow theirs (k_arena_service)
	const double reserve = (cfg->arena_reserve >= 0.0f) ? cfg->arena_reserve : 1.0;
	const size_t node_total = static_cast<size_t>(static_cast<double>(G * 2 * d.node_cap) * (1.0 + reserve));
	const size_t edge_total = static_cast<size_t>(static_cast<double>(G * 2 * d.edge_cap) * (1.0 + reserve));
	const size_t ht_total = static_cast<size_t>(static_cast<double>(G * d.ht_cap) * (1.0 + reserve));
	AGX_TRY(dev_alloc(e, &d.nodes, node_total));
	AGX_TRY(dev_alloc(e, &d.edges, edge_total));
	AGX_TRY(dev_alloc(e, &d.ht, ht_total));
	AGX_TRY(dev_alloc(e, &d.heap, 1));
	AGX_TRY(dev_alloc(e, &d.free_bundles, static_cast<size_t>(ARENA_CLASSES) * G));
	AGX_TRY(dev_alloc(e, &d.tasks, G * d.batch));
	// speculative solver: its waves solve several leaves of one game at once, so each wave brings its own spill areas (behind the games')
	e->speculative = cfg->speculative_solver != 0 && cfg->tss_max_positions <= 250 && cfg->max_batch_size <= 16;
	if (e->speculative)
	{
		int cus = e->sizing_cus, device_of_engine = 0;
		if (!e->sizing_only)
		{
			(void) hipGetDevice(&device_of_engine);
			(void) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_of_engine);
		}
		// default: as many waves as the launch's kernel keeps resident (16 per compute unit on 15x15 boards, 10 on 20x20: its LDS state and registers), but
		// no more than one per leaf of a full batch (a one-tree engine has a few dozen leaves per launch)
		const int per_cu = spec_resident_waves_per_cu(cfg->rules, cfg->board_size);
		e->spec_waves = (cfg->speculative_waves > 0) ? cfg->speculative_waves : std::min<long long>(static_cast<long long>(per_cu) * std::max(1, cus), std::max<long long>(64, static_cast<long long>(G) * cfg->max_batch_size));
		e->spec_waves = std::min(e->spec_waves, SPEC_QUEUE_SLACK);
	}
	// (a group's waves take the areas n_games + group * waves + wave with waves = max(1, spec_waves / n_groups): at least one area per possible group)
	// ... and behind the waves' areas one per park buffer (a parked solve keeps its tails there until the next launch takes it up)
	const bool parking = e->speculative && cfg->rules == AGX_RENJU && cfg->solver_yield_fraction > 0.0f && !cfg->match_mode && std::max(1, cfg->search_threads) == 1;
	const size_t areas = G + static_cast<size_t>(std::max(e->spec_waves, 16)) + (parking ? SPEC_PARK_POOL : 0);
	d.park_area0 = static_cast<int>(G) + std::max(e->spec_waves, 16);
	d.spec_on = e->speculative ? 1 : 0;
	d.park_fraction = parking ? SPEC_PARK_FRACTION : 0.0f;
	if (const char *pf = std::getenv("AGX_PARK_FRACTION")) // (developer knob: the sweep behind SPEC_PARK_FRACTION, profiles/r05_search_variants_ab.txt)
		if (parking && std::atof(pf) > 0.0)
			d.park_fraction = std::min(1.0f, static_cast<float>(std::atof(pf))); // (a fraction of the launch's games: (0, 1])
	AGX_TRY(dev_alloc(e, &d.park_slot, G * d.batch));
	AGX_TRY(dev_alloc(e, &d.park_owner, SPEC_PARK_POOL));
	AGX_TRY(dev_alloc(e, &d.parts_done, 2 * G));
	if (!e->sizing_only)
		(void) hipMemset(d.parts_done, 0, 2 * G * sizeof(int));
	AGX_TRY(dev_alloc(e, &d.park_lds, parking ? static_cast<size_t>(SPEC_PARK_POOL) * SPEC_PARK_WORDS : 1));
	if (!e->sizing_only)
		(void) hipMemset(d.park_slot, 0, G * d.batch * sizeof(int));
	if (!e->sizing_only)
		(void) hipMemset(d.park_owner, 0, SPEC_PARK_POOL * sizeof(int));
	AGX_TRY(dev_alloc(e, &d.act, areas * d.act_cap));
	// per area 20 lists x MAXHW entries: list_get / list_set stride by the kernel's compile-time board (SolverSharedT::HW, = MAXHW in the any-size kernels)
	AGX_TRY(dev_alloc(e, &d.list_spill, areas * 20 * static_cast<size_t>(MAXHW)));
	AGX_TRY(dev_alloc(e, &d.frame_spill, areas * MAX_FRAMES));
	AGX_TRY(dev_alloc(e, &d.snap_spill, areas * static_cast<size_t>(d.hw + 2) * 64)); // undo snapshots: 512 bytes per stone on the board and area
	d.spec_group = 0;
	d.spec_waves = e->spec_waves;
	AGX_TRY(dev_alloc(e, &d.spec_watchdog, 16));
	if (!e->sizing_only)
		(void) hipMemset(d.spec_watchdog, 0, 16 * sizeof(int));
	AGX_TRY(dev_alloc(e, &d.spec_prof, 16));
	AGX_TRY(dev_alloc(e, &d.spec_trace, 4 * (G + static_cast<size_t>(std::max(e->spec_waves, 16))))); // per game, then per wave (profile builds)
	if (!e->sizing_only)
		(void) hipMemset(d.spec_trace, 0, 4 * (G + static_cast<size_t>(std::max(e->spec_waves, 16))) * sizeof(unsigned long long));
	if (!e->sizing_only)
		(void) hipMemset(d.spec_prof, 0, 16 * sizeof(unsigned long long));
	AGX_TRY(dev_alloc(e, &d.spec_items, e->speculative ? G * d.batch + 16 * static_cast<size_t>(SPEC_QUEUE_SLACK) : 1));
	AGX_TRY(dev_alloc(e, &d.spec_left, e->speculative ? G : 1));
	AGX_TRY(dev_alloc(e, &d.spec_tasks, e->speculative ? G * d.batch : 1));
	AGX_TRY(dev_alloc(e, &d.spec_overlay, e->speculative ? G * d.batch * (SPEC_OV_CAP * 16) : 1));
	AGX_TRY(dev_alloc(e, &d.tt, G * buckets * 8));
	AGX_TRY(dev_upload(e, &d.t_pattern, tables.pattern));
	AGX_TRY(dev_upload(e, &d.t_ho3, tables.half_open_three));
	AGX_TRY(dev_upload(e, &d.t_threat, tables.threat));
	{ // the solver's form of the same table: cross type | circle type << 4 in one byte (dev_solver.hpp:threat_lookup)
		std::vector<uint8_t> packed(4096);
		for (int i = 0; i < 4096; i++)
			packed[i] = static_cast<uint8_t>((tables.threat[2 * i] & 15u) | ((tables.threat[2 * i + 1] & 15u) << 4));
		AGX_TRY(dev_upload(e, &d.t_threat_packed, packed));
	}
	AGX_TRY(dev_upload(e, &d.t_defense, tables.defense));
	AGX_TRY(dev_upload(e, &d.nc_keys, nc_keys));
	AGX_TRY(dev_upload(e, &d.zob, e->zobrist));
	AGX_TRY(dev_alloc(e, &d.nn_features, G * d.batch * d.hw));
	AGX_TRY(dev_alloc(e, &d.nn_policy, G * d.batch * d.hw));
	AGX_TRY(dev_alloc(e, &d.nn_value, G * d.batch * 3));
	d.has_q = cfg->action_values ? 1 : 0;
	d.match_mode = cfg->match_mode ? 1 : 0;
	d.prune_root = (cfg->match_mode || !cfg->force_expand_root) ? 1 : 0; // UnifiedGenerator(.., forceExpandRoot): true in self-play (GameGenerator.cpp:183-184), false for a Player (Player.cpp:109)
	d.search_buffers = (cfg->search_buffers == 2) ? 2 : 1;
	d.shared_tree = (cfg->search_threads > 1 || d.search_buffers == 2) ? 1 : 0;
	d.tt_mod = (d.search_buffers == 2) ? std::max(1, cfg->search_threads) : static_cast<int>(G);
	d.grp_first = 0;
	d.grp_count = static_cast<int>(G);
	d.policy_temperature = cfg->policy_temperature;
	AGX_TRY(dev_alloc(e, &d.nn_q, d.has_q ? G * d.batch * d.hw * 2 : 1));
	AGX_TRY(dev_alloc(e, &d.noise, d.noise_type ? G * d.hw : 1));
	AGX_TRY(dev_alloc(e, &d.nn_list, G * d.batch));
	AGX_TRY(dev_alloc(e, &d.counters, N_COUNTERS));
	AGX_TRY(dev_alloc(e, &d.records, static_cast<size_t>(d.record_cap)));
	AGX_TRY(dev_alloc(e, &d.record_edges, (d.record_format & 1) ? static_cast<size_t>(d.record_edge_cap) : 1));
	AGX_TRY(dev_alloc(e, &d.samples, (d.record_format & 2) ? static_cast<size_t>(d.sample_cap) : 4));
	AGX_TRY(dev_alloc(e, &d.game_ends, static_cast<size_t>(d.game_end_cap)));
#undef AGX_TRY
	if (status == AGX_OK)
	{
		std::vector<GameState> games(G);
		std::memset(static_cast<void*>(games.data()), 0, G * sizeof(GameState));
		for (size_t g = 0; g < G; g++)
		{
			games[g].node_off[0] = (2 * g) * d.node_cap;
			games[g].node_off[1] = (2 * g + 1) * d.node_cap;
			games[g].edge_off[0] = (2 * g) * static_cast<uint64_t>(d.edge_cap);
			games[g].edge_off[1] = (2 * g + 1) * static_cast<uint64_t>(d.edge_cap);
			games[g].ht_off = g * static_cast<uint64_t>(d.ht_cap);
			games[g].node_cap = d.node_cap;
			games[g].edge_cap = d.edge_cap;
			games[g].ht_cap = d.ht_cap;
		}
		ArenaHeap heap;
		std::memset(&heap, 0, sizeof(heap));
		heap.node_cursor = G * 2 * d.node_cap;
		heap.edge_cursor = G * 2 * static_cast<uint64_t>(d.edge_cap);
		heap.ht_cursor = G * static_cast<uint64_t>(d.ht_cap);
		heap.node_total = node_total;
		heap.edge_total = edge_total;
		heap.ht_total = ht_total;
		heap.free_capacity = static_cast<int32_t>(G);
		e->idle_games = games;
		e->idle_heap = heap;
		hipError_t err = e->sizing_only ? hipSuccess : hipMemcpy(d.games, games.data(), G * sizeof(GameState), hipMemcpyHostToDevice);
		if (e->sizing_only)
			err = hipErrorUnknown; // (skips the chain below; reset behind it)
		if (err == hipSuccess)
			err = hipMemcpy(d.heap, &heap, sizeof(heap), hipMemcpyHostToDevice);
		if (err == hipSuccess)
			err = hipMemset(d.counters, 0, N_COUNTERS * sizeof(int));
		if (err == hipSuccess)
			err = hipMemset(d.tasks, 0, G * d.batch * sizeof(DTask));
		if (err == hipSuccess && e->speculative)
			err = hipMemset(d.spec_items, 0, (G * d.batch + 16 * static_cast<size_t>(SPEC_QUEUE_SLACK)) * sizeof(int));
		if (err == hipSuccess && e->speculative)
			err = hipMemset(d.spec_tasks, 0, G * d.batch * sizeof(SpecTask));
		if (err == hipSuccess && e->speculative)
			err = hipMemset(d.spec_left, 0, G * sizeof(int));
		if (err != hipSuccess && !e->sizing_only)
		{
			agx::set_error("hipMemset failed: %s", hipGetErrorString(err));
			status = AGX_ERR_HIP;
		}
	}
	if (status != AGX_OK)
	{
		for (void *p : e->allocations)
			(void) hipFree(p);
		delete e;
		return status;
	}
	*out = e;
	return AGX_OK;
}

int agx_engine_create(const AgxEngineConfig *cfg, AgxEngine **out)
{
	return engine_create(cfg, out, 0);
}

/* What agx_engine_create(cfg) allocates on a device with `compute_units` compute units, without touching one (a launcher sizing its ranks:
 * GeneratorManager.cpp:146-152 starts one generator thread per device; tests of the multi-GPU plan on a box without a GPU) */
int agx_engine_estimate_device_bytes(const AgxEngineConfig *cfg, int compute_units, unsigned long long *bytes)
{
	AGX_REQUIRE(bytes != nullptr && compute_units > 0, AGX_ERR_INVALID, "agx_engine_estimate_device_bytes: invalid argument");
	AgxEngine *e = nullptr;
	const int st = engine_create(cfg, &e, compute_units);
	if (st != AGX_OK)
		return st;
	*bytes = e->device_bytes;
	delete e;
	return AGX_OK;
}

int agx_engine_destroy(AgxEngine *e)
{
	if (e == nullptr)
		return AGX_OK;
	for (void *p : e->allocations)
		(void) hipFree(p);
	for (hipEvent_t ev : e->events)
		(void) hipEventDestroy(ev);
	for (hipEvent_t ev : e->free_events)
		(void) hipEventDestroy(ev);
	if (e->summary_host != nullptr)
		(void) hipHostFree(e->summary_host);
	delete e;
	return AGX_OK;
}

int agx_engine_begin(AgxEngine *e, const uint16_t *h_openings, int n_openings, void *stream)
{
	AGX_REQUIRE(e != nullptr && h_openings != nullptr, AGX_ERR_INVALID, "agx_engine_begin: null argument");
	AGX_REQUIRE(n_openings > 0, AGX_ERR_INVALID, "agx_engine_begin: need at least one opening");
	uint16_t *d_op = nullptr;
	AGX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_op), static_cast<size_t>(n_openings) * OPENING_CAP * sizeof(uint16_t)));
	e->allocations.push_back(d_op);
	AGX_HIP_CHECK(hipMemcpy(d_op, h_openings, static_cast<size_t>(n_openings) * OPENING_CAP * sizeof(uint16_t), hipMemcpyHostToDevice));
	e->dev.openings = d_op;
	e->dev.n_openings = n_openings;
	int counters[N_COUNTERS] = { 0 };
	counters[1] = e->dev.match_mode ? 0 : (e->dev.shared_tree ? 1 : e->dev.n_games);
	AGX_HIP_CHECK(hipMemcpy(e->dev.counters, counters, sizeof(counters), hipMemcpyHostToDevice));
	hipStream_t s = static_cast<hipStream_t>(stream);
	hipLaunchKernelGGL(k_begin, dim3(e->dev.n_games), dim3(256), 0, s, e->dev);
	if (e->dev.match_mode)
	{ // the pairs take openings 0 .. n_games/2 - 1 in order, exactly as they will after finished matches
		EngineDev d = e->dev;
		d.g0 = 0;
		hipLaunchKernelGGL(k_assign_openings, dim3(1), dim3(1024), 0, s, d, d.n_games / 2);
		hipLaunchKernelGGL(k_match_restart, dim3(d.n_games / 2), dim3(256), 0, s, d);
	}
	AGX_HIP_CHECK(hipGetLastError());
	e->begun = true;
	return AGX_OK;
}

int agx_engine_add_openings(AgxEngine *e, const uint16_t *h_openings, int n_openings)
{
	AGX_REQUIRE(e != nullptr && h_openings != nullptr, AGX_ERR_INVALID, "agx_engine_add_openings: null argument");
	AGX_REQUIRE(n_openings > 0, AGX_ERR_INVALID, "agx_engine_add_openings: need at least one opening");
	AGX_REQUIRE(e->begun, AGX_ERR_STATE, "agx_engine_add_openings: agx_engine_begin has not been called");
	AGX_HIP_CHECK(hipDeviceSynchronize());
	const size_t words_old = static_cast<size_t>(e->dev.n_openings) * OPENING_CAP, words_new = static_cast<size_t>(n_openings) * OPENING_CAP;
	uint16_t *d_op = nullptr;
	AGX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_op), (words_old + words_new) * sizeof(uint16_t)));
	AGX_HIP_CHECK(hipMemcpy(d_op, e->dev.openings, words_old * sizeof(uint16_t), hipMemcpyDeviceToDevice));
	AGX_HIP_CHECK(hipMemcpy(d_op + words_old, h_openings, words_new * sizeof(uint16_t), hipMemcpyHostToDevice));
	// the previous table is no longer referenced by any launch (device synchronised above): replace it in the allocation list
	for (void *&p : e->allocations)
		if (p == static_cast<const void*>(e->dev.openings))
		{
			(void) hipFree(p);
			p = d_op;
		}
	e->dev.openings = d_op;
	e->dev.n_openings += n_openings;
	return AGX_OK;
}

/* games [first, first + count) of group `group` out of `n_groups` equal parts of the pool */
static int group_range(const AgxEngine *e, int group, int n_groups, EngineDev &d, int &count)
{
	// counters[16 + g] / counters[32 + g] are group g's network count and yield count: at most 16 groups
	AGX_REQUIRE(n_groups >= 1 && n_groups <= 16 && group >= 0 && group < n_groups, AGX_ERR_INVALID, "group %d of %d is not valid (1..16 groups)", group, n_groups);
	const int per = (e->dev.n_games + n_groups - 1) / n_groups;
	d = e->dev;
	d.g0 = group * per;
	count = std::min(per, e->dev.n_games - d.g0);
	d.nn_counter = 16 + group;
	d.yield_counter = 32 + group;
	AGX_REQUIRE(count > 0, AGX_ERR_INVALID, "group %d of %d is empty for %d games", group, n_groups, e->dev.n_games);
	AGX_REQUIRE(!e->dev.shared_tree || n_groups == e->dev.search_buffers, AGX_ERR_INVALID,
			"a tournament-search pool (search_threads) is one tree: it is stepped as a whole, or buffer by buffer (2 groups) when search_buffers = 2 — not in %d groups", n_groups);
	d.grp_first = d.g0;
	d.grp_count = count;
	return AGX_OK;
}

static constexpr int CLEAR_PARTS = 16; // workgroups per restarting game in k_clear_tables

#ifdef AGX_QUICK /* developer builds: only the 15x15 non-renju solver is instantiated (a fifth of the compile time) */
#define AGX_LAUNCH_SOLVE(FUSED) hipLaunchKernelGGL((k_solve<AGX_QUICK_RENJU, 15, FUSED>), grid, block, 0, s, d)
#else
#define AGX_LAUNCH_SOLVE(FUSED) \
	do { \
		if (d.rules == AGX_RENJU) \
		{ \
			if (d.n == 15) \
				hipLaunchKernelGGL((k_solve<true, 15, FUSED>), grid, block, 0, s, d); \
			else \
				hipLaunchKernelGGL((k_solve<true, 0, FUSED>), grid, block, 0, s, d); \
		} \
		else if (d.n == 15) \
			hipLaunchKernelGGL((k_solve<false, 15, FUSED>), grid, block, 0, s, d); \
		else if (d.n == 20) \
			hipLaunchKernelGGL((k_solve<false, 20, FUSED>), grid, block, 0, s, d); \
		else \
			hipLaunchKernelGGL((k_solve<false, 0, FUSED>), grid, block, 0, s, d); \
	} while (0)
#endif
__global__ void k_reset_spec(int *nn_counter, int *nn_second, int *spec_counters, int *done_counter)
{
	*nn_counter = 0;
	*done_counter = 0;
	if (nn_second != nullptr)
		*nn_second = 0;
	for (int i = 0; i < 4; i++)
		spec_counters[i] = 0;
}
/* select + speculative solve + scheduleToNN of `count` games from d.g0 as one persistent launch of `waves` waves (k_search_spec) */
static void launch_search_spec(EngineDev d, int count, int group, int waves, hipStream_t s)
{
	d.spec_group = group;
	d.spec_waves = waves;
	const dim3 grid(waves), block(64);
	hipLaunchKernelGGL(spec_kernel(d.rules, d.n), grid, block, 0, s, d, count);
}
static void launch_solve(const EngineDev &d, int count, hipStream_t s, bool with_select = false)
{
	const dim3 grid(count), block(64);
	if (with_select)
		AGX_LAUNCH_SOLVE(true);
	else
		AGX_LAUNCH_SOLVE(false);
}
#undef AGX_LAUNCH_SOLVE

/* Search::select for every game of the group (Search.cpp:117-158) */
int agx_engine_select_group(AgxEngine *e, int group, int n_groups, void *stream)
{
	AGX_REQUIRE(e != nullptr, AGX_ERR_INVALID, "agx_engine_select: null engine");
	AGX_REQUIRE(e->begun, AGX_ERR_STATE, "agx_engine_select: agx_engine_begin has not been called");
	EngineDev d;
	int count = 0;
	const int st = group_range(e, group, n_groups, d, count);
	if (st != AGX_OK)
		return st;
	hipStream_t s = static_cast<hipStream_t>(stream);
	hipLaunchKernelGGL(k_reset_counter, dim3(1), dim3(1), 0, s, d.counters + d.nn_counter, d.counters + d.yield_counter);
	{
		KernelTimer t(e, s, 0);
		hipLaunchKernelGGL(k_select, dim3(d.shared_tree ? 1 : count), dim3(64), 0, s, d); // tournament search: one wave walks the threads in turn
	}
	AGX_HIP_CHECK(hipGetLastError());
	return AGX_OK;
}
/* Search::solve + Search::scheduleToNN (Search.cpp:159-199): the threat solver on the selected leaves, then the device-side queue */
int agx_engine_solve_group(AgxEngine *e, int group, int n_groups, void *stream)
{
	AGX_REQUIRE(e != nullptr, AGX_ERR_INVALID, "agx_engine_solve: null engine");
	AGX_REQUIRE(e->begun, AGX_ERR_STATE, "agx_engine_solve: agx_engine_begin has not been called");
	EngineDev d;
	int count = 0;
	const int st = group_range(e, group, n_groups, d, count);
	if (st != AGX_OK)
		return st;
	hipStream_t s = static_cast<hipStream_t>(stream);
	{
		KernelTimer t(e, s, 1);
		launch_solve(d, count, s);
	}
	AGX_HIP_CHECK(hipGetLastError());
	return AGX_OK;
}
/* Search::solve(endTime >= 0) (Search.cpp:159-183, the call of SearchThread::asynchronous_run): node limit `max_nodes` (the reference sets
 * 10 000) and a time budget of `seconds` from the moment the launch starts, shared out task by task like the reference does.  Always the
 * serial solver (one wave per game, straight on the table): overlays of speculative solves hold 256 buckets. */
int agx_engine_solve_timed_group(AgxEngine *e, int group, int n_groups, int max_nodes, double seconds, void *stream)
{
	AGX_REQUIRE(e != nullptr, AGX_ERR_INVALID, "agx_engine_solve_timed: null engine");
	AGX_REQUIRE(e->begun, AGX_ERR_STATE, "agx_engine_solve_timed: agx_engine_begin has not been called");
	AGX_REQUIRE(max_nodes >= 1 && max_nodes <= 100000, AGX_ERR_INVALID, "agx_engine_solve_timed: node limit %d outside [1, 100000]", max_nodes);
	EngineDev d;
	int count = 0;
	const int st = group_range(e, group, n_groups, d, count);
	if (st != AGX_OK)
		return st;
	d.tss_max_nodes = max_nodes;
	d.yield_fraction = 0.0f; // every task is solved in this launch: the time limit is what bounds it
	d.solve_time_ticks = (seconds <= 0.0) ? 1ull : static_cast<unsigned long long>(std::min(seconds, 3600.0) * 1.0e8) + 1ull;
	hipStream_t s = static_cast<hipStream_t>(stream);
	{
		KernelTimer t(e, s, 1);
		launch_solve(d, count, s);
	}
	AGX_HIP_CHECK(hipGetLastError());
	return AGX_OK;
}
int agx_engine_select_solve_group(AgxEngine *e, int group, int n_groups, void *stream)
{
	if (e != nullptr && e->begun && e->speculative)
	{ // one persistent launch: games select off a cursor, their leaves are solved in parallel against overlays, committed in batch order
		EngineDev d;
		int count = 0;
		const int st = group_range(e, group, n_groups, d, count);
		if (st != AGX_OK)
			return st;
		hipStream_t s = static_cast<hipStream_t>(stream);
		hipLaunchKernelGGL(k_reset_spec, dim3(1), dim3(1), 0, s, d.counters + d.nn_counter, static_cast<int*>(nullptr), d.counters + SPEC_COUNTER0 + 4 * group,
				d.counters + d.yield_counter);
		if (d.shared_tree)
		{ // tournament search: the threads descend the one tree in turn (one wave), then their leaves are solved in parallel
			KernelTimer t(e, s, 0);
			hipLaunchKernelGGL(k_select, dim3(1), dim3(64), 0, s, d);
		}
		{
			KernelTimer t(e, s, 1);
			launch_search_spec(d, count, group, std::max(1, e->spec_waves / n_groups), s);
		}
		AGX_HIP_CHECK(hipGetLastError());
		return AGX_OK;
	}
	if (e != nullptr && e->begun && e->fuse_select && !e->dev.shared_tree)
	{ // one launch: every game selects and solves in its own wave (k_solve<.., FUSED>); timed as the solve stage
		EngineDev d;
		int count = 0;
		const int st = group_range(e, group, n_groups, d, count);
		if (st != AGX_OK)
			return st;
		hipStream_t s = static_cast<hipStream_t>(stream);
		hipLaunchKernelGGL(k_reset_counter, dim3(1), dim3(1), 0, s, d.counters + d.nn_counter, d.counters + d.yield_counter);
		{
			KernelTimer t(e, s, 1);
			launch_solve(d, count, s, true);
		}
		AGX_HIP_CHECK(hipGetLastError());
		return AGX_OK;
	}
	const int st = agx_engine_select_group(e, group, n_groups, stream);
	return (st != AGX_OK) ? st : agx_engine_solve_group(e, group, n_groups, stream);
}

/* match mode, both players' trees in one launch per stage: about half of the trees search at any time, so one launch over all of
 * them keeps the GPU as busy as a self-play pool of n_games / 2; only the network stage is split (one slot list per player) */
static int match_launch(AgxEngine *e, EngineDev &d)
{
	AGX_REQUIRE(e != nullptr, AGX_ERR_INVALID, "agx_engine_*_match: null engine");
	AGX_REQUIRE(e->begun, AGX_ERR_STATE, "agx_engine_*_match: agx_engine_begin has not been called");
	AGX_REQUIRE(e->dev.match_mode, AGX_ERR_STATE, "agx_engine_*_match: the engine was not created with match_mode");
	d = e->dev;
	d.g0 = 0;
	d.nn_counter = 16;
	d.yield_counter = 32;
	d.match_merged = 1;
	return AGX_OK;
}
int agx_engine_select_solve_match(AgxEngine *e, void *stream)
{
	EngineDev d;
	const int st = match_launch(e, d);
	if (st != AGX_OK)
		return st;
	hipStream_t s = static_cast<hipStream_t>(stream);
	hipLaunchKernelGGL(k_reset_counter, dim3(1), dim3(1), 0, s, d.counters + 16, d.counters + 32);
	hipLaunchKernelGGL(k_reset_counter, dim3(1), dim3(1), 0, s, d.counters + 17, static_cast<int*>(nullptr));
	if (e->speculative)
	{
		hipLaunchKernelGGL(k_reset_spec, dim3(1), dim3(1), 0, s, d.counters + 16, d.counters + 17, d.counters + SPEC_COUNTER0, d.counters + 32);
		KernelTimer t(e, s, 1);
		launch_search_spec(d, d.n_games, 0, e->spec_waves, s);
	}
	else if (e->fuse_select)
	{
		KernelTimer t(e, s, 1);
		launch_solve(d, d.n_games, s, true);
	}
	else
	{
		{
			KernelTimer t(e, s, 0);
			hipLaunchKernelGGL(k_select, dim3(d.n_games), dim3(64), 0, s, d);
		}
		{
			KernelTimer t(e, s, 1);
			launch_solve(d, d.n_games, s);
		}
	}
	AGX_HIP_CHECK(hipGetLastError());
	return AGX_OK;
}
int agx_engine_expand_backup_match(AgxEngine *e, void *stream)
{
	EngineDev d;
	const int st = match_launch(e, d);
	if (st != AGX_OK)
		return st;
	hipStream_t s = static_cast<hipStream_t>(stream);
	{
		KernelTimer t(e, s, 2);
		hipLaunchKernelGGL(k_expand, dim3(d.n_games), dim3(64), 0, s, d);
	}
	{
		KernelTimer t(e, s, 3);
		// a moving tree's workgroup also rebases its partner's tree; the partner's own workgroup has nothing to do (it is not searching)
		hipLaunchKernelGGL(k_advance, dim3(d.n_games), dim3(ADV_THREADS), 0, s, d);
		hipLaunchKernelGGL(k_arena_service, dim3(1), dim3(1024), 0, s, d, d.n_games, 0);
		hipLaunchKernelGGL(k_arena_copy, dim3(d.n_games * CLEAR_PARTS), dim3(256), 0, s, d, CLEAR_PARTS); // (its last part per game commits)
		hipLaunchKernelGGL(k_assign_openings, dim3(1), dim3(1024), 0, s, d, d.n_games / 2);
		hipLaunchKernelGGL(k_clear_tables, dim3(d.n_games * CLEAR_PARTS), dim3(256), 0, s, d, CLEAR_PARTS, 0);
		hipLaunchKernelGGL(k_match_restart, dim3(d.n_games / 2), dim3(256), 0, s, d);
	}
	AGX_HIP_CHECK(hipGetLastError());
	return AGX_OK;
}
int agx_engine_step_match(AgxEngine *e, AgxNet *first_net, AgxNet *second_net, void *stream)
{
	int st = agx_engine_select_solve_match(e, stream);
	if (st == AGX_OK)
		st = agx_engine_evaluate_group(e, first_net, 0, 2, stream);
	if (st == AGX_OK)
		st = agx_engine_evaluate_group(e, second_net, 1, 2, stream);
	if (st == AGX_OK)
		st = agx_engine_expand_backup_match(e, stream);
	return st;
}

int agx_engine_evaluate_group(AgxEngine *e, AgxNet *net, int group, int n_groups, void *stream)
{
	AGX_REQUIRE(e != nullptr && net != nullptr, AGX_ERR_INVALID, "agx_engine_evaluate: null argument");
	EngineDev d;
	int count = 0;
	const int st = group_range(e, group, n_groups, d, count);
	if (st != AGX_OK)
		return st;
	AgxNetDesc desc;
	const int ds = agx_net_description(net, &desc);
	if (ds != AGX_OK)
		return ds;
	// the exchange buffers are laid out with the ENGINE's board (hw cells per slot): a network of another geometry would index them with
	// the wrong stride
	AGX_REQUIRE(desc.rows == d.n && desc.cols == d.n, AGX_ERR_INVALID, "agx_engine_evaluate: the network is built for %dx%d boards, the engine plays on %dx%d",
			desc.rows, desc.cols, d.n, d.n);
	if (d.has_q)
	{
		AGX_REQUIRE(desc.action_values, AGX_ERR_INVALID, "agx_engine_evaluate: the engine was configured for a 'pvq' network but this one has no action-values head");
		return agx_nn_forward_indirect_pvq(net, d.nn_features, d.nn_list + static_cast<size_t>(d.g0) * d.batch, d.counters + d.nn_counter, count * d.batch,
				d.nn_policy, d.nn_value, d.nn_q, stream);
	}
	return agx_nn_forward_indirect(net, d.nn_features, d.nn_list + static_cast<size_t>(d.g0) * d.batch, d.counters + d.nn_counter, count * d.batch, d.nn_policy,
			d.nn_value, stream);
}

/* Search::generateEdges + expand + backup for every game of the group (Search.cpp:206-232), incl. the move rule's decision */
static int expand_stage(AgxEngine *e, int group, int n_groups, void *stream, bool service_arenas)
{
	AGX_REQUIRE(e != nullptr, AGX_ERR_INVALID, "agx_engine_expand: null engine");
	AGX_REQUIRE(e->begun, AGX_ERR_STATE, "agx_engine_expand: agx_engine_begin has not been called");
	EngineDev d;
	int count = 0;
	const int st = group_range(e, group, n_groups, d, count);
	if (st != AGX_OK)
		return st;
	AGX_REQUIRE(!d.match_mode || n_groups == 2, AGX_ERR_INVALID, "agx_engine_expand: a match-mode engine is stepped as two groups (first players, second players)");
	hipStream_t s = static_cast<hipStream_t>(stream);
	{
		KernelTimer t(e, s, 2);
		hipLaunchKernelGGL(k_expand, dim3(d.shared_tree ? 1 : count), dim3(64), 0, s, d);
	}
	if (service_arenas && e->external_moves && !d.match_mode)
	{ // a caller that makes the moves itself (set_board) never runs the advance stage: the trees that asked for larger arenas get them here
	  // (a pool stepped through expand_group + advance_group — ag::Search::expand, then GameGenerator — has them serviced ONCE, by the advance stage)
		const int trees = d.shared_tree ? 1 : count;
		if (d.shared_tree)
			d.g0 = 0;
		hipLaunchKernelGGL(k_arena_service, dim3(1), dim3(1024), 0, s, d, trees, 0);
		hipLaunchKernelGGL(k_arena_copy, dim3(trees * CLEAR_PARTS), dim3(256), 0, s, d, CLEAR_PARTS); // (its last part per game commits)
	}
	AGX_HIP_CHECK(hipGetLastError());
	return AGX_OK;
}
int agx_engine_expand_group(AgxEngine *e, int group, int n_groups, void *stream)
{
	return expand_stage(e, group, n_groups, stream, true);
}
/* GameGenerator::make_move + prepare_search for the games whose search is complete (GameGenerator.cpp:145-185), then the next openings
 * for the games that ended */
int agx_engine_advance_group(AgxEngine *e, int group, int n_groups, void *stream)
{
	AGX_REQUIRE(e != nullptr, AGX_ERR_INVALID, "agx_engine_advance: null engine");
	AGX_REQUIRE(e->begun, AGX_ERR_STATE, "agx_engine_advance: agx_engine_begin has not been called");
	EngineDev d;
	int count = 0;
	const int st = group_range(e, group, n_groups, d, count);
	if (st != AGX_OK)
		return st;
	AGX_REQUIRE(!d.match_mode || n_groups == 2, AGX_ERR_INVALID, "agx_engine_advance: a match-mode engine is stepped as two groups (first players, second players)");
	hipStream_t s = static_cast<hipStream_t>(stream);
	{
		KernelTimer t(e, s, 3);
		const int trees = d.shared_tree ? 1 : count; // tournament search: one tree (game 0), the other records are its search threads
		if (d.shared_tree)
		{ // (whichever buffer this group is: the tree, its arenas and the restart are record 0's; grp_first / grp_count keep the buffer)
			d.g0 = 0;
			count = d.n_games;
		}
		hipLaunchKernelGGL(k_advance, dim3(trees), dim3(ADV_THREADS), 0, s, d);
		// four launches behind k_advance instead of seven (round 6): the openings are assigned by the service workgroup, a game's last copy part
		// commits its larger arenas, a game's last clear part restarts it
		hipLaunchKernelGGL(k_arena_service, dim3(1), dim3(1024), 0, s, d, trees, d.match_mode ? 0 : trees);
		hipLaunchKernelGGL(k_arena_copy, dim3(trees * CLEAR_PARTS), dim3(256), 0, s, d, CLEAR_PARTS);
		if (!d.match_mode)
		{
			hipLaunchKernelGGL(k_clear_tables, dim3(count * CLEAR_PARTS), dim3(256), 0, s, d, CLEAR_PARTS, d.shared_tree ? 0 : 1);
			if (d.shared_tree)
			{ // the search threads restart on thread 0's opening: they first (they read its request), thread 0 last (it clears the request)
				EngineDev others = d;
				others.g0 = 1;
				if (count > 1)
					hipLaunchKernelGGL(k_restart, dim3(count - 1), dim3(256), 0, s, others);
				hipLaunchKernelGGL(k_restart, dim3(1), dim3(256), 0, s, d);
			}

		}
		else if (group == 0)
		{ // restarts go by pair, requested through the first players' trees (= group 0)
			hipLaunchKernelGGL(k_assign_openings, dim3(1), dim3(1024), 0, s, d, count);
			{
				EngineDev all = d; // both players' trees
				all.g0 = 0;
				hipLaunchKernelGGL(k_clear_tables, dim3(all.n_games * CLEAR_PARTS), dim3(256), 0, s, all, CLEAR_PARTS, 0);
			}
			hipLaunchKernelGGL(k_match_restart, dim3(count), dim3(256), 0, s, d);
		}
	}
	AGX_HIP_CHECK(hipGetLastError());
	return AGX_OK;
}
int agx_engine_expand_backup_group(AgxEngine *e, int group, int n_groups, void *stream)
{
	const int st = expand_stage(e, group, n_groups, stream, false); // (the advance stage services the arenas)
	return (st != AGX_OK) ? st : agx_engine_advance_group(e, group, n_groups, stream);
}

int agx_engine_step_group(AgxEngine *e, AgxNet *net, int group, int n_groups, void *stream)
{
	int st = agx_engine_select_solve_group(e, group, n_groups, stream);
	if (st == AGX_OK)
		st = agx_engine_evaluate_group(e, net, group, n_groups, stream);
	if (st == AGX_OK)
		st = agx_engine_expand_backup_group(e, group, n_groups, stream);
	return st;
}

int agx_engine_select_solve(AgxEngine *e, void *stream) { return agx_engine_select_solve_group(e, 0, 1, stream); }
int agx_engine_evaluate(AgxEngine *e, AgxNet *net, void *stream) { return agx_engine_evaluate_group(e, net, 0, 1, stream); }
int agx_engine_expand_backup(AgxEngine *e, void *stream) { return agx_engine_expand_backup_group(e, 0, 1, stream); }
int agx_engine_step(AgxEngine *e, AgxNet *net, void *stream) { return agx_engine_step_group(e, net, 0, 1, stream); }

int agx_engine_buffers(AgxEngine *e, AgxEngineBuffers *out)
{
	AGX_REQUIRE(e != nullptr && out != nullptr, AGX_ERR_INVALID, "agx_engine_buffers: null argument");
	out->d_nn_features = e->dev.nn_features;
	out->d_nn_policy = e->dev.nn_policy;
	out->d_nn_value = e->dev.nn_value;
	out->d_nn_action_values = e->dev.has_q ? e->dev.nn_q : nullptr;
	out->d_nn_list = e->dev.nn_list;
	out->d_nn_count = e->dev.counters + 16; /* group 0 of 1 */
	out->slots = e->dev.n_games * e->dev.batch;
	out->cells = e->dev.hw;
	return AGX_OK;
}

int agx_engine_match_results(AgxEngine *e, int *h_results, int pair_capacity)
{
	AGX_REQUIRE(e != nullptr && h_results != nullptr, AGX_ERR_INVALID, "agx_engine_match_results: null argument");
	AGX_REQUIRE(e->dev.match_mode, AGX_ERR_STATE, "agx_engine_match_results: the engine was not created with match_mode");
	const int pairs = e->dev.n_games / 2;
	AGX_REQUIRE(pair_capacity >= pairs, AGX_ERR_INVALID, "agx_engine_match_results: %d pairs do not fit into %d", pairs, pair_capacity);
	AGX_HIP_CHECK(hipDeviceSynchronize());
	std::vector<GameState> games(pairs);
	AGX_HIP_CHECK(hipMemcpy(games.data(), e->dev.games, games.size() * sizeof(GameState), hipMemcpyDeviceToHost));
	for (int i = 0; i < pairs; i++)
	{
		h_results[4 * i + 0] = games[i].match_score[0];
		h_results[4 * i + 1] = games[i].match_score[1];
		h_results[4 * i + 2] = games[i].match_score[2];
		h_results[4 * i + 3] = games[i].games_done;
	}
	return AGX_OK;
}

int agx_engine_set_board(AgxEngine *e, int game, const uint8_t *h_board, int sign_to_move, void *stream)
{
	AGX_REQUIRE(e != nullptr && h_board != nullptr, AGX_ERR_INVALID, "agx_engine_set_board: null argument");
	AGX_REQUIRE(e->begun, AGX_ERR_STATE, "agx_engine_set_board: agx_engine_begin has not been called");
	AGX_REQUIRE(game >= 0 && game < e->dev.n_games, AGX_ERR_INVALID, "agx_engine_set_board: game %d of %d", game, e->dev.n_games);
	AGX_REQUIRE(sign_to_move == 1 || sign_to_move == 2, AGX_ERR_INVALID, "agx_engine_set_board: sign_to_move must be 1 (cross) or 2 (circle)");
	AGX_REQUIRE(!e->dev.match_mode, AGX_ERR_STATE, "agx_engine_set_board: a match-mode engine plays its own games");
	AGX_REQUIRE(!e->dev.shared_tree || game == 0, AGX_ERR_INVALID, "agx_engine_set_board: a tournament-search engine has one tree, game 0 (records 1.. are its task buffers)");
	for (int i = 0; i < e->dev.hw; i++)
		AGX_REQUIRE(h_board[i] <= 2, AGX_ERR_INVALID, "agx_engine_set_board: cell %d holds %d (0 empty, 1 cross, 2 circle)", i, h_board[i]);
	hipStream_t s = static_cast<hipStream_t>(stream);
	if (e->board_staging == nullptr)
	{
		AGX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e->board_staging), MAXHW));
		e->allocations.push_back(e->board_staging);
	}
	AGX_HIP_CHECK(hipStreamSynchronize(s)); // (the staging buffer of the previous call may still be read)
	AGX_HIP_CHECK(hipMemcpy(e->board_staging, h_board, e->dev.hw, hipMemcpyHostToDevice));
	{ // Player::setBoard begins with search.cleanup(tree) (Player.cpp:98-100): leaves still in flight give their virtual losses back first
		EngineDev one = e->dev;
		one.g0 = game;
		hipLaunchKernelGGL(k_cancel_pending, dim3(1), dim3(64), 0, s, one);
	}
	hipLaunchKernelGGL(k_set_board, dim3(1), dim3(256), 0, s, e->dev, game, e->board_staging, sign_to_move);
	e->external_moves = true;
	AGX_HIP_CHECK(hipGetLastError());
	return AGX_OK;
}
/* GeneratorThread::saveGames / loadGames (GeneratorManager.cpp:98-122) on the device pool: the games in flight as their move lists */
int agx_engine_save_games(AgxEngine *e, AgxSavedGame *h_out, int capacity, int *count)
{
	AGX_REQUIRE(e != nullptr && count != nullptr, AGX_ERR_INVALID, "agx_engine_save_games: null argument");
	AGX_REQUIRE(e->begun, AGX_ERR_STATE, "agx_engine_save_games: agx_engine_begin has not been called");
	AGX_REQUIRE(!e->dev.match_mode && !e->dev.shared_tree, AGX_ERR_UNSUPPORTED, "agx_engine_save_games: self-play pools only");
	AGX_HIP_CHECK(hipDeviceSynchronize());
	std::vector<GameState> games(e->dev.n_games);
	AGX_HIP_CHECK(hipMemcpy(games.data(), e->dev.games, games.size() * sizeof(GameState), hipMemcpyDeviceToHost));
	int n = 0;
	for (size_t g = 0; g < games.size(); g++)
	{
		const GameState &gs = games[g];
		if (!gs.active || gs.outcome != 0 || gs.error != 0)
			continue; // waiting for an opening, finished (its record is already in the pools) or stopped
		if (h_out != nullptr)
		{
			AGX_REQUIRE(n < capacity, AGX_ERR_INVALID, "agx_engine_save_games: more than %d games in flight", capacity);
			AgxSavedGame &o = h_out[n];
			std::memset(&o, 0, sizeof(o));
			o.game_slot = static_cast<int>(g);
			o.game_index = gs.games_done;
			o.opening_id = gs.opening_id;
			o.sign_to_move = gs.sign_to_move;
			o.nn_queued = gs.nn_queued;
			o.n_moves = gs.n_moves;
			for (int i = 0; i < gs.n_moves; i++)
				o.moves[i] = gs.moves[i];
		}
		n++;
	}
	*count = n;
	return AGX_OK;
}
int agx_engine_restore_game(AgxEngine *e, const AgxSavedGame *game, void *stream)
{
	AGX_REQUIRE(e != nullptr && game != nullptr, AGX_ERR_INVALID, "agx_engine_restore_game: null argument");
	AGX_REQUIRE(e->begun, AGX_ERR_STATE, "agx_engine_restore_game: agx_engine_begin has not been called");
	AGX_REQUIRE(!e->dev.match_mode && !e->dev.shared_tree, AGX_ERR_UNSUPPORTED, "agx_engine_restore_game: self-play pools only");
	AGX_REQUIRE(game->game_slot >= 0 && game->game_slot < e->dev.n_games, AGX_ERR_INVALID, "agx_engine_restore_game: slot %d of %d", game->game_slot, e->dev.n_games);
	AGX_REQUIRE(game->n_moves >= 0 && game->n_moves < e->dev.hw, AGX_ERR_INVALID, "agx_engine_restore_game: %d moves on a board of %d cells", game->n_moves, e->dev.hw);
	std::vector<uint8_t> seen(e->dev.hw, 0);
	for (int i = 0; i < game->n_moves; i++)
	{
		const uint32_t mv = game->moves[i];
		const int sg = mv & 3, r = (mv >> 2) & 127, c = (mv >> 9) & 127;
		AGX_REQUIRE((sg == 1 || sg == 2) && r < e->dev.n && c < e->dev.n && !seen[r * e->dev.n + c], AGX_ERR_INVALID,
				"agx_engine_restore_game: move %d (0x%x) is not a stone on an empty cell", i, mv);
		seen[r * e->dev.n + c] = 1;
	}
	hipStream_t s = static_cast<hipStream_t>(stream);
	if (e->moves_staging == nullptr)
	{
		AGX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e->moves_staging), MAXHW * sizeof(uint16_t)));
		e->allocations.push_back(e->moves_staging);
	}
	AGX_HIP_CHECK(hipStreamSynchronize(s)); // (the staging buffer of the previous call may still be read)
	AGX_HIP_CHECK(hipMemcpy(e->moves_staging, game->moves, sizeof(uint16_t) * std::max(1, game->n_moves), hipMemcpyHostToDevice));
	hipLaunchKernelGGL(k_restore_game, dim3(1), dim3(256), 0, s, e->dev, game->game_slot, e->moves_staging, game->n_moves, game->opening_id, game->nn_queued);
	AGX_HIP_CHECK(hipGetLastError());
	return AGX_OK;
}
int agx_engine_set_batch_size(AgxEngine *e, int batch_size)
{ // Search::setBatchSize (Search.cpp:252-255): the task buffers keep their capacity (max_batch_size), the select stage fills `batch_size` of it
	AGX_REQUIRE(e != nullptr, AGX_ERR_INVALID, "agx_engine_set_batch_size: null engine");
	AGX_REQUIRE(batch_size >= 1 && batch_size <= e->dev.batch, AGX_ERR_INVALID, "agx_engine_set_batch_size: %d outside [1, max_batch_size = %d]", batch_size, e->dev.batch);
	e->dev.batch_limit = batch_size;
	return AGX_OK;
}
int agx_engine_root_summary(AgxEngine *e, int game, void *stream, int *out4)
{
	AGX_REQUIRE(e != nullptr && out4 != nullptr, AGX_ERR_INVALID, "agx_engine_root_summary: null argument");
	AGX_REQUIRE(e->begun, AGX_ERR_STATE, "agx_engine_root_summary: agx_engine_begin has not been called");
	AGX_REQUIRE(game >= 0 && game < e->dev.n_games, AGX_ERR_INVALID, "agx_engine_root_summary: game %d of %d", game, e->dev.n_games);
	if (e->summary_dev == nullptr)
	{
		AGX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e->summary_dev), 4 * sizeof(int)));
		e->allocations.push_back(e->summary_dev);
		AGX_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&e->summary_host), 4 * sizeof(int), hipHostMallocDefault));
	}
	hipStream_t s = static_cast<hipStream_t>(stream);
	EngineDev d = e->dev;
	{ // (k_root_summary reads the game's arenas through its own record)
		hipLaunchKernelGGL(k_root_summary, dim3(1), dim3(1), 0, s, d, game, e->summary_dev);
	}
	AGX_HIP_CHECK(hipMemcpyAsync(e->summary_host, e->summary_dev, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
	AGX_HIP_CHECK(hipStreamSynchronize(s)); // THIS stream only: a network launch on another stream keeps running
	std::memcpy(out4, e->summary_host, 4 * sizeof(int));
	return AGX_OK;
}
int agx_engine_cancel_pending(AgxEngine *e, void *stream)
{
	AGX_REQUIRE(e != nullptr, AGX_ERR_INVALID, "agx_engine_cancel_pending: null engine");
	AGX_REQUIRE(e->begun, AGX_ERR_STATE, "agx_engine_cancel_pending: agx_engine_begin has not been called");
	AGX_REQUIRE(!e->dev.match_mode, AGX_ERR_STATE, "agx_engine_cancel_pending: a match-mode engine plays its own games");
	EngineDev d = e->dev;
	d.g0 = 0;
	hipLaunchKernelGGL(k_cancel_pending, dim3(d.shared_tree ? 1 : d.n_games), dim3(64), 0, static_cast<hipStream_t>(stream), d);
	AGX_HIP_CHECK(hipGetLastError());
	return AGX_OK;
}
int agx_engine_set_force_expand_root(AgxEngine *e, int force_expand_root)
{
	AGX_REQUIRE(e != nullptr, AGX_ERR_INVALID, "agx_engine_set_force_expand_root: null engine");
	AGX_REQUIRE(!e->dev.match_mode, AGX_ERR_STATE, "agx_engine_set_force_expand_root: a match-mode engine always prunes the root");
	e->dev.prune_root = force_expand_root ? 0 : 1;
	e->cfg.force_expand_root = force_expand_root ? 1 : 0;
	return AGX_OK;
}
int agx_engine_set_max_simulations(AgxEngine *e, int max_simulations)
{
	AGX_REQUIRE(e != nullptr && max_simulations > 0, AGX_ERR_INVALID, "agx_engine_set_max_simulations: invalid argument");
	e->dev.max_sims = max_simulations;
	e->cfg.max_simulations = max_simulations;
	return AGX_OK;
}
int agx_engine_speculative_waves(AgxEngine *e, int *waves)
{
	AGX_REQUIRE(e != nullptr && waves != nullptr, AGX_ERR_INVALID, "agx_engine_speculative_waves: null argument");
	*waves = e->speculative ? e->spec_waves : 0;
	return AGX_OK;
}
int agx_engine_device_bytes(AgxEngine *e, unsigned long long *bytes)
{
	AGX_REQUIRE(e != nullptr && bytes != nullptr, AGX_ERR_INVALID, "agx_engine_device_bytes: null argument");
	*bytes = e->device_bytes;
	return AGX_OK;
}
int agx_engine_stats(AgxEngine *e, AgxEngineStats *out)
{
	AGX_REQUIRE(e != nullptr && out != nullptr, AGX_ERR_INVALID, "agx_engine_stats: null argument");
	std::vector<GameState> games(e->dev.n_games);
	AGX_HIP_CHECK(hipMemcpy(games.data(), e->dev.games, games.size() * sizeof(GameState), hipMemcpyDeviceToHost));
	int counters[16];
	AGX_HIP_CHECK(hipMemcpy(counters, e->dev.counters, sizeof(counters), hipMemcpyDeviceToHost));
	std::memset(out, 0, sizeof(*out));
	for (const GameState &g : games)
	{
		out->evaluated_nodes += g.stats[0];
		out->network_evaluations += g.stats[1];
		out->information_leaks += g.stats[2];
		out->proven_edge_visits += g.stats[3];
		out->wasted_expansions += g.stats[4];
		out->solver_nodes += g.stats[5];
		out->select_levels += g.stats[6];
		out->select_edge_reads += g.stats[7];
		out->moves_played += g.stats[8];
		out->duplicate_selections += g.stats[9];
		out->speculative_solves += g.spec_stats[0];
		out->speculative_reruns += g.spec_stats[1];
		out->speculative_deferrals += g.spec_stats[2];
		out->speculative_parks += static_cast<int>(g.spec_stats[3]);
		out->peak_nodes = std::max<unsigned long long>(out->peak_nodes, g.stats[10]);
		out->peak_edges = std::max<unsigned long long>(out->peak_edges, g.stats[11]);
		out->active_games += g.active ? 1 : 0;
		if (g.error != 0 && out->first_error == 0)
			out->first_error = g.error;
	}
	if (e->speculative)
	{
		int w[16];
		AGX_HIP_CHECK(hipMemcpy(w, e->dev.spec_watchdog, sizeof(w), hipMemcpyDeviceToHost));
		if (w[0] != 0)
			fprintf(stderr, "[k_search_spec watchdog] a wave waited in vain for queue slot %d: head %d, tail %d, games selected %d of %d, select cursor %d\n", w[1], w[2], w[3], w[4],
					w[5], w[6]);
	}
#ifdef AGX_SPEC_PROFILE
	if (getenv("AGX_SPEC_TRACE"))
	{ // per game of the LAST launch: select done, commit begin, commit end (100 MHz ticks), re-runs * 16 + leaves
		std::vector<unsigned long long> tr(4 * games.size());
		AGX_HIP_CHECK(hipMemcpy(tr.data(), e->dev.spec_trace, tr.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
		FILE *f = fopen(getenv("AGX_SPEC_TRACE"), "w");
		if (f != nullptr)
		{
			for (size_t g = 0; g < games.size(); g++)
				fprintf(f, "%llu %llu %llu %llu\n", tr[4 * g], tr[4 * g + 1], tr[4 * g + 2], tr[4 * g + 3]);
			fclose(f);
		}
		// per wave of the last launches (one per group): select phase over, exit, leaves solved, ticks waited for items
		std::vector<unsigned long long> wv(4 * static_cast<size_t>(std::max(e->spec_waves, 16)));
		AGX_HIP_CHECK(hipMemcpy(wv.data(), e->dev.spec_trace + 4 * games.size(), wv.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
		f = fopen((std::string(getenv("AGX_SPEC_TRACE")) + ".waves").c_str(), "w");
		if (f != nullptr)
		{
			for (size_t w = 0; 4 * w < wv.size(); w++)
				fprintf(f, "%llu %llu %llu %llu\n", wv[4 * w], wv[4 * w + 1], wv[4 * w + 2], wv[4 * w + 3]);
			fclose(f);
		}
	}
	{
		unsigned long long p[16];
		AGX_HIP_CHECK(hipMemcpy(p, e->dev.spec_prof, sizeof(p), hipMemcpyDeviceToHost));
		fprintf(stderr, "[k_search_spec profile, wave-ms summed] select phase %.1f (slowest wave, max over launches %.3f ms), waiting for items %.1f, speculative solves %.1f (%llu, %.1f us each), "
				"commits + re-runs %.1f (longest %.3f ms), longest launch %.3f ms\n", p[0] / 1e5, p[1] / 1e5, p[2] / 1e5, p[3] / 1e5, p[6], p[3] / 1e2 / (p[6] ? p[6] : 1), p[4] / 1e5, p[5] / 1e5,
				p[7] / 1e5);
	}
#endif
#ifdef AGX_SOLVER_PROFILE
	{
		unsigned long long p[10] = { 0 };
		for (const GameState &g : games)
		{
			for (int i = 0; i < 6; i++)
				p[i] += g.prof[i];
			p[6] += g.prof[6] & 0xFFFFFFFFull;
			p[7] += g.prof[6] >> 32;
			p[8] += g.prof[7] & 0xFFFFFFFFull;
			p[9] += g.prof[7] >> 32;
		}
		fprintf(stderr, "[solver profile, 100 MHz ticks] solves %llu: set_board+encode %.1f us, frame machine %.1f us, place/remove %.1f us (%.1f per solve, %.2f us each), total %.1f us per solve\n",
				p[5], p[0] / 100.0 / p[5], p[1] / 100.0 / p[5], p[2] / 100.0 / p[5], (double) p[3] / p[5], p[2] / 100.0 / (p[3] ? p[3] : 1), p[4] / 100.0 / p[5]);
		fprintf(stderr, "[frame machine split, us per solve] table seek %.1f, move generation %.1f, ordering %.1f, evaluate+insert %.1f\n",
				p[6] / 100.0 / p[5], p[7] / 100.0 / p[5], p[8] / 100.0 / p[5], p[9] / 100.0 / p[5]);
		unsigned long long dp[24] = { 0 };
		for (const GameState &g : games)
			for (int i = 0; i < 24; i++)
				dp[i] += g.dprof[i];
		const double solves = static_cast<double>(p[5]);
		fprintf(stderr, "[generate(), shader cycles per solve; %.1f calls] win1 %.0f, loss2 %.0f, win3 %.0f, loss4 %.0f, win5 %.0f, loss6 %.0f, half4 %.0f, rest %.0f\n",
				dp[8] / solves, dp[0] / solves, dp[1] / solves, dp[2] / solves, dp[3] / solves, dp[4] / solves, dp[5] / solves, dp[6] / solves, dp[7] / solves);
		fprintf(stderr, "[frame machine, shader cycles per solve; %.1f loop turns, %.1f picks] resume %.0f, pick %.0f, descend %.0f, child-returned %.0f\n",
				dp[21] / solves, dp[22] / solves, dp[15] / solves, dp[16] / solves, dp[17] / solves, dp[20] / solves);
		fprintf(stderr, "[update_around, shader cycles per solve; %.1f calls, %.2f list changes per call] centre %.0f, gather %.0f, lists %.0f\n",
				dp[14] / solves, dp[13] / (dp[14] ? static_cast<double>(dp[14]) : 1.0), dp[10] / solves, dp[11] / solves, dp[12] / solves);
		fprintf(stderr, "[renju, per solve] forbidden bits of the network input %.1f us, mark_forbidden_moves %.0f shader cycles, generator foul tests %.0f shader cycles (%.2f tests)\n",
				dp[23] / 100.0 / solves, dp[9] / solves, dp[18] / solves, dp[19] / solves);
	}
#endif
	{
		ArenaHeap heap;
		AGX_HIP_CHECK(hipMemcpy(&heap, e->dev.heap, sizeof(heap), hipMemcpyDeviceToHost));
		out->arena_grows = heap.grows;
		out->arena_releases = heap.releases;
		out->arena_failures = heap.failures;
		for (const GameState &g : games)
			out->arena_max_class = std::max(out->arena_max_class, g.arena_class);
		out->arena_heap_used = static_cast<float>(static_cast<double>(heap.edge_cursor) / static_cast<double>(heap.edge_total));
	}
	out->games_finished = counters[2];
	out->openings_taken = counters[1];
	out->records_used = counters[3];
	out->record_edges_used = counters[4];
	return AGX_OK;
}

int agx_engine_game_info(AgxEngine *e, int game, AgxGameInfo *info, uint8_t *h_board, AgxEdgeView *h_root_edges, int edge_capacity)
{
	AGX_REQUIRE(e != nullptr && info != nullptr, AGX_ERR_INVALID, "agx_engine_game_info: null argument");
	AGX_REQUIRE(game >= 0 && game < e->dev.n_games, AGX_ERR_INVALID, "agx_engine_game_info: game %d out of range", game);
	AGX_HIP_CHECK(hipDeviceSynchronize());
	GameState gs;
	AGX_HIP_CHECK(hipMemcpy(&gs, e->dev.games + game, sizeof(GameState), hipMemcpyDeviceToHost));
	info->active = gs.active;
	info->sign_to_move = gs.sign_to_move;
	info->n_moves = gs.n_moves;
	info->outcome = gs.outcome;
	info->error = gs.error;
	info->opening_id = gs.opening_id;
	info->games_done = gs.games_done;
	info->n_nodes = gs.n_nodes;
	info->n_edges = gs.n_edges;
	info->grow_pending = gs.grow_pending;
	info->arena_class = gs.arena_class;
	info->root_visits = 0;
	info->root_win = info->root_draw = 0.0f;
	info->root_score = 0;
	info->root_edges = 0;
	info->root_moves_left = 0.0f;
	info->max_depth = gs.max_depth;
	if (h_board != nullptr)
		std::memcpy(h_board, gs.board, e->dev.hw);
	if (gs.root >= 0)
	{
		DNode root;
		const DNode *nodes = e->dev.nodes + gs.node_off[gs.arena];
		AGX_HIP_CHECK(hipMemcpy(&root, nodes + gs.root, sizeof(DNode), hipMemcpyDeviceToHost));
		info->root_visits = root.visits;
		info->root_win = root.win;
		info->root_draw = root.draw;
		info->root_score = root.score;
		info->root_edges = root.n_edges;
		info->root_moves_left = root.moves_left;
		if (h_root_edges != nullptr)
		{
			AGX_REQUIRE(root.n_edges <= edge_capacity, AGX_ERR_INVALID, "agx_engine_game_info: %d root edges do not fit into %d", root.n_edges, edge_capacity);
			std::vector<DEdge> edges(root.n_edges);
			const DEdge *pool = e->dev.edges + gs.edge_off[gs.arena];
			AGX_HIP_CHECK(hipMemcpy(edges.data(), pool + root.edge_begin, edges.size() * sizeof(DEdge), hipMemcpyDeviceToHost));
			for (int i = 0; i < root.n_edges; i++)
			{
				h_root_edges[i].prior = edges[i].prior;
				h_root_edges[i].win = edges[i].win;
				h_root_edges[i].draw = edges[i].draw;
				h_root_edges[i].visits = edges[i].visits;
				h_root_edges[i].move = edges[i].move;
				h_root_edges[i].score = edges[i].score;
				h_root_edges[i].flag_and_virtual_loss = edges[i].flag_vl;
			}
		}
	}
	return AGX_OK;
}

int agx_engine_fetch_records(AgxEngine *e, AgxMoveRecord *h_records, int record_capacity, AgxEdgeView *h_edges, int edge_capacity, uint8_t *h_samples,
		int sample_capacity, AgxGameEnd *h_game_ends, int game_end_capacity, AgxRecordCounts *counts, int drain)
{
	AGX_REQUIRE(e != nullptr && counts != nullptr, AGX_ERR_INVALID, "agx_engine_fetch_records: null argument");
	AGX_HIP_CHECK(hipDeviceSynchronize());
	int counters[16];
	AGX_HIP_CHECK(hipMemcpy(counters, e->dev.counters, sizeof(counters), hipMemcpyDeviceToHost));
	const int nr = std::min(counters[3], e->dev.record_cap), ne = (e->dev.record_format & 1) ? std::min(counters[4], e->dev.record_edge_cap) : 0;
	const int ns = (e->dev.record_format & 2) ? std::min(counters[6], e->dev.sample_cap) : 0, ng = std::min(counters[7], e->dev.game_end_cap);
	counts->records = nr;
	counts->edges = ne;
	counts->sample_bytes = ns;
	counts->game_ends = ng;
	if (h_records != nullptr)
	{
		AGX_REQUIRE(nr <= record_capacity, AGX_ERR_INVALID, "agx_engine_fetch_records: %d records do not fit into %d", nr, record_capacity);
		std::vector<MoveRecordHeader> headers(nr);
		AGX_HIP_CHECK(hipMemcpy(headers.data(), e->dev.records, headers.size() * sizeof(MoveRecordHeader), hipMemcpyDeviceToHost));
		for (int i = 0; i < nr; i++)
		{
			h_records[i].game_serial = headers[i].game_serial;
			h_records[i].move_number = headers[i].move_number;
			h_records[i].move = headers[i].move;
			h_records[i].root_score = headers[i].root_score;
			h_records[i].root_visits = headers[i].root_visits;
			h_records[i].root_win = headers[i].root_win;
			h_records[i].root_draw = headers[i].root_draw;
			h_records[i].root_flags = headers[i].root_flags;
			h_records[i].n_edges = headers[i].n_edges;
			h_records[i].edge_offset = headers[i].edge_offset;
			h_records[i].game_slot = headers[i].game_slot;
			h_records[i].game_index = headers[i].game_index;
			h_records[i].sample_offset = headers[i].sample_offset;
			h_records[i].sample_bytes = headers[i].sample_bytes;
			h_records[i].outcome = headers[i].outcome;
		}
	}
	if (h_edges != nullptr && ne > 0)
	{
		AGX_REQUIRE(ne <= edge_capacity, AGX_ERR_INVALID, "agx_engine_fetch_records: %d edges do not fit into %d", ne, edge_capacity);
		std::vector<DEdge> edges(ne);
		AGX_HIP_CHECK(hipMemcpy(edges.data(), e->dev.record_edges, edges.size() * sizeof(DEdge), hipMemcpyDeviceToHost));
		for (int i = 0; i < ne; i++)
		{
			h_edges[i].prior = edges[i].prior;
			h_edges[i].win = edges[i].win;
			h_edges[i].draw = edges[i].draw;
			h_edges[i].visits = edges[i].visits;
			h_edges[i].move = edges[i].move;
			h_edges[i].score = edges[i].score;
			h_edges[i].flag_and_virtual_loss = edges[i].flag_vl;
			h_edges[i].reserved = 0;
		}
	}
	if (h_samples != nullptr && ns > 0)
	{
		AGX_REQUIRE(ns <= sample_capacity, AGX_ERR_INVALID, "agx_engine_fetch_records: %d sample bytes do not fit into %d", ns, sample_capacity);
		AGX_HIP_CHECK(hipMemcpy(h_samples, e->dev.samples, static_cast<size_t>(ns), hipMemcpyDeviceToHost));
	}
	if (h_game_ends != nullptr && ng > 0)
	{
		AGX_REQUIRE(ng <= game_end_capacity, AGX_ERR_INVALID, "agx_engine_fetch_records: %d finished games do not fit into %d", ng, game_end_capacity);
		std::vector<GameEndRecord> ends(ng);
		AGX_HIP_CHECK(hipMemcpy(ends.data(), e->dev.game_ends, ends.size() * sizeof(GameEndRecord), hipMemcpyDeviceToHost));
		for (int i = 0; i < ng; i++)
		{
			h_game_ends[i].game_serial = ends[i].game_serial;
			h_game_ends[i].game_slot = ends[i].game_slot;
			h_game_ends[i].game_index = ends[i].game_index;
			h_game_ends[i].outcome = ends[i].outcome;
			h_game_ends[i].n_moves = ends[i].n_moves;
			std::memcpy(h_game_ends[i].moves, ends[i].moves, sizeof(ends[i].moves));
		}
	}
	if (drain)
	{ // the device is synchronised: the pools start empty again (game serials / indices keep counting)
		AGX_HIP_CHECK(hipMemset(e->dev.counters + 3, 0, 2 * sizeof(int)));
		AGX_HIP_CHECK(hipMemset(e->dev.counters + 6, 0, 2 * sizeof(int)));
	}
	return AGX_OK;
}

int agx_engine_records(AgxEngine *e, AgxMoveRecord *h_records, int record_capacity, AgxEdgeView *h_edges, int edge_capacity, int *n_records, int *n_edges)
{
	AGX_REQUIRE(e != nullptr && n_records != nullptr && n_edges != nullptr, AGX_ERR_INVALID, "agx_engine_records: null argument");
	AgxRecordCounts counts;
	const bool sizes_only = (h_records == nullptr || h_edges == nullptr);
	const int st = agx_engine_fetch_records(e, sizes_only ? nullptr : h_records, record_capacity, sizes_only ? nullptr : h_edges, edge_capacity, nullptr, 0, nullptr, 0,
			&counts, 0);
	if (st != AGX_OK)
		return st;
	*n_records = counts.records;
	*n_edges = counts.edges;
	return AGX_OK;
}

int agx_engine_drain_records(AgxEngine *e, AgxMoveRecord *h_records, int record_capacity, AgxEdgeView *h_edges, int edge_capacity, int *n_records, int *n_edges)
{
	AGX_REQUIRE(h_records != nullptr && h_edges != nullptr && n_records != nullptr && n_edges != nullptr, AGX_ERR_INVALID,
			"agx_engine_drain_records: null buffer (query the sizes with agx_engine_records)");
	AgxRecordCounts counts;
	const int st = agx_engine_fetch_records(e, h_records, record_capacity, h_edges, edge_capacity, nullptr, 0, nullptr, 0, &counts, 1);
	if (st != AGX_OK)
		return st;
	*n_records = counts.records;
	*n_edges = counts.edges;
	return AGX_OK;
}

int agx_engine_zobrist(AgxEngine *e, uint64_t *h_keys, size_t n_words)
{
	AGX_REQUIRE(e != nullptr && h_keys != nullptr, AGX_ERR_INVALID, "agx_engine_zobrist: null argument");
	AGX_REQUIRE(n_words == e->zobrist.size(), AGX_ERR_INVALID, "agx_engine_zobrist: expected %zu words", e->zobrist.size());
	std::memcpy(h_keys, e->zobrist.data(), n_words * sizeof(uint64_t));
	return AGX_OK;
}

int agx_stream_create(void **out)
{
	AGX_REQUIRE(out != nullptr, AGX_ERR_INVALID, "agx_stream_create: null argument");
	hipStream_t s = nullptr;
	AGX_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	*out = s;
	return AGX_OK;
}
/* a stream whose kernels run only on the compute units whose bits are set in `mask` (bit i of word i / 32 = CU i): slices of a pool
 * that own disjoint parts of the chip */
namespace
{
	/* hipStreamDestroy of a CU-masked stream hangs on ROCm 7.2, so such a stream lives until the process exits — and therefore must not be
	 * created twice: the streams are cached process-wide by (device, mask).  A generator thread that sets its slices up again for every
	 * training iteration (GeneratorManager::generate, once per iteration in the reference) gets the same streams back instead of leaking a
	 * hardware queue per slice and iteration. */
	struct MaskedStream
	{
			int device, instance;
			std::vector<uint32_t> mask;
			hipStream_t stream;
	};
	std::mutex g_masked_streams_mutex;
	std::vector<MaskedStream> g_masked_streams;
}
int agx_stream_create_with_cu_mask(void **out, const uint32_t *mask, int words)
{
	return agx_stream_create_with_cu_mask_instance(out, mask, words, 0);
}
int agx_stream_create_with_cu_mask_instance(void **out, const uint32_t *mask, int words, int instance)
{
	AGX_REQUIRE(out != nullptr && mask != nullptr && words > 0, AGX_ERR_INVALID, "agx_stream_create_with_cu_mask: invalid argument");
	bool any = false;
	for (int i = 0; i < words; i++)
		any = any || mask[i] != 0u;
	AGX_REQUIRE(any, AGX_ERR_INVALID, "agx_stream_create_with_cu_mask: the mask selects no compute unit");
	int device = 0;
	AGX_HIP_CHECK(hipGetDevice(&device));
	std::vector<uint32_t> key(mask, mask + words);
	while (key.size() > 1 && key.back() == 0u)
		key.pop_back(); // (trailing zero words select nothing: the same mask)
	std::lock_guard<std::mutex> lock(g_masked_streams_mutex);
	for (const MaskedStream &m : g_masked_streams)
		if (m.device == device && m.instance == instance && m.mask == key)
		{
			*out = m.stream;
			return AGX_OK;
		}
	hipStream_t s = nullptr;
	AGX_HIP_CHECK(hipExtStreamCreateWithCUMask(&s, static_cast<uint32_t>(words), mask));
	g_masked_streams.push_back(MaskedStream { device, instance, key, s });
	*out = s;
	return AGX_OK;
}
int agx_stream_masked_count(void)
{
	std::lock_guard<std::mutex> lock(g_masked_streams_mutex);
	return static_cast<int>(g_masked_streams.size());
}
int agx_stream_destroy(void *stream)
{
	if (stream == nullptr)
		return AGX_OK;
	hipStream_t s = static_cast<hipStream_t>(stream);
	bool masked = false;
	{
		std::lock_guard<std::mutex> lock(g_masked_streams_mutex);
		for (const MaskedStream &m : g_masked_streams)
			masked = masked || (m.stream == s);
	}
	if (masked)
	{ // drained (outside the lock), not destroyed: it stays in the cache for the next caller
		AGX_HIP_CHECK(hipStreamSynchronize(s));
		return AGX_OK;
	}
	AGX_HIP_CHECK(hipStreamDestroy(s));
	return AGX_OK;
}
/* events: ordering between streams without the host (a slice's tower launch on the network partition waits for its search launch on the
 * search partition and the other way round) */
int agx_event_create(void **out)
{
	AGX_REQUIRE(out != nullptr, AGX_ERR_INVALID, "agx_event_create: null argument");
	hipEvent_t ev = nullptr;
	AGX_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
	*out = ev;
	return AGX_OK;
}
int agx_event_create_blocking(void **out)
{ // for agx_event_synchronize: the waiting host thread sleeps (hipEventBlockingSync) instead of spinning
	AGX_REQUIRE(out != nullptr, AGX_ERR_INVALID, "agx_event_create_blocking: null argument");
	hipEvent_t ev = nullptr;
	AGX_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming));
	*out = ev;
	return AGX_OK;
}
int agx_event_synchronize(void *event)
{ // a SLEEPING wait: hipEventSynchronize spins in user space on this runtime even for a hipEventBlockingSync event (measured: the waiting
  // thread stays at 100 %), so the event is polled between 100 us naps — the callers wait for work that was queued steps ago
	AGX_REQUIRE(event != nullptr, AGX_ERR_INVALID, "agx_event_synchronize: null event");
	while (true)
	{
		const hipError_t st = hipEventQuery(static_cast<hipEvent_t>(event));
		if (st == hipSuccess)
			return AGX_OK;
		if (st != hipErrorNotReady)
			AGX_HIP_CHECK(st);
		std::this_thread::sleep_for(std::chrono::microseconds(100));
	}
}
int agx_event_record(void *event, void *stream)
{
	AGX_REQUIRE(event != nullptr, AGX_ERR_INVALID, "agx_event_record: null event");
	AGX_HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream)));
	return AGX_OK;
}
int agx_stream_wait_event(void *stream, void *event)
{
	AGX_REQUIRE(event != nullptr, AGX_ERR_INVALID, "agx_stream_wait_event: null event");
	AGX_HIP_CHECK(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(event), 0));
	return AGX_OK;
}
int agx_event_destroy(void *event)
{
	if (event != nullptr)
		AGX_HIP_CHECK(hipEventDestroy(static_cast<hipEvent_t>(event)));
	return AGX_OK;
}
int agx_stream_synchronize(void *stream)
{
	AGX_HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
	return AGX_OK;
}

/* ---- test hooks: single stages on caller-supplied positions ---- */
int agx_engine_kernel_timing(AgxEngine *e, int enable, double *ms_out, long long *launches_out)
{
	AGX_REQUIRE(e != nullptr, AGX_ERR_INVALID, "agx_engine_kernel_timing: null engine");
	AGX_HIP_CHECK(hipDeviceSynchronize());
	double ms[4] = { 0.0, 0.0, 0.0, 0.0 };
	long long launches[4] = { 0, 0, 0, 0 };
	for (size_t i = 0; i < e->event_kernel.size(); i++)
	{
		float t = 0.0f;
		AGX_HIP_CHECK(hipEventElapsedTime(&t, e->events[2 * i], e->events[2 * i + 1]));
		ms[e->event_kernel[i]] += t;
		launches[e->event_kernel[i]]++;
		e->free_events.push_back(e->events[2 * i]);
		e->free_events.push_back(e->events[2 * i + 1]);
	}
	e->events.clear();
	e->event_kernel.clear();
	for (int k = 0; k < 4; k++)
	{
		if (ms_out != nullptr)
			ms_out[k] = ms[k];
		if (launches_out != nullptr)
			launches_out[k] = launches[k];
	}
	e->timing = (enable != 0);
	return AGX_OK;
}

int agx_debug_solve(AgxEngine *e, const uint8_t *h_boards, const int *h_signs, int count, uint32_t *h_features, uint16_t *h_moves, uint16_t *h_scores,
		int *h_counts, uint32_t *h_flags, uint16_t *h_result_scores)
{
	AGX_REQUIRE(e != nullptr && h_boards != nullptr && h_signs != nullptr, AGX_ERR_INVALID, "agx_debug_solve: null argument");
	AGX_REQUIRE(count > 0 && count <= e->dev.n_games, AGX_ERR_INVALID, "agx_debug_solve: count must be in [1, n_games]");
	const EngineDev &d = e->dev;
	uint8_t *d_boards = nullptr;
	int *d_signs = nullptr;
	AGX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_boards), static_cast<size_t>(count) * d.hw));
	AGX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_signs), count * sizeof(int)));
	AGX_HIP_CHECK(hipMemcpy(d_boards, h_boards, static_cast<size_t>(count) * d.hw, hipMemcpyHostToDevice));
	AGX_HIP_CHECK(hipMemcpy(d_signs, h_signs, count * sizeof(int), hipMemcpyHostToDevice));
	hipLaunchKernelGGL(k_debug_load_tasks, dim3(count), dim3(64), 0, nullptr, d, d_boards, d_signs, count);
	EngineDev dd = d;
	dd.g0 = 0;
	dd.nn_counter = 16;
	dd.yield_fraction = 0.0f;
	hipLaunchKernelGGL(k_reset_counter, dim3(1), dim3(1), 0, nullptr, dd.counters + dd.nn_counter, static_cast<int*>(nullptr));
	launch_solve(dd, count, nullptr);
	AGX_HIP_CHECK(hipGetLastError());
	AGX_HIP_CHECK(hipDeviceSynchronize());
	std::vector<DTask> tasks(1);
	for (int g = 0; g < count; g++)
	{
		AGX_HIP_CHECK(hipMemcpy(tasks.data(), d.tasks + static_cast<size_t>(g) * d.batch, sizeof(DTask), hipMemcpyDeviceToHost));
		const DTask &t = tasks[0];
		h_counts[g] = t.n_edges;
		h_flags[g] = t.flags;
		h_result_scores[g] = static_cast<uint16_t>(t.score);
		for (int i = 0; i < t.n_edges; i++)
		{
			h_moves[static_cast<size_t>(g) * d.hw + i] = t.emove[i];
			h_scores[static_cast<size_t>(g) * d.hw + i] = t.escore[i];
		}
		if (h_features != nullptr)
			AGX_HIP_CHECK(hipMemcpy(h_features + static_cast<size_t>(g) * d.hw, d.nn_features + static_cast<size_t>(g) * d.batch * d.hw, d.hw * sizeof(uint32_t),
					hipMemcpyDeviceToHost));
	}
	{ // solver nodes of every position (GameState::stats[5]) for agx_debug_solve_nodes
		std::vector<GameState> after(count);
		AGX_HIP_CHECK(hipMemcpy(after.data(), d.games, static_cast<size_t>(count) * sizeof(GameState), hipMemcpyDeviceToHost));
		e->debug_solve_nodes.resize(count);
		for (int g = 0; g < count; g++)
			e->debug_solve_nodes[g] = after[g].stats[5];
	}
	// leave the pool idle again
	AGX_HIP_CHECK(hipMemcpy(d.games, e->idle_games.data(), e->idle_games.size() * sizeof(GameState), hipMemcpyHostToDevice)); // idle pool, arena descriptors intact
	(void) hipFree(d_boards);
	(void) hipFree(d_signs);
	return AGX_OK;
}

int agx_debug_solve_nodes(AgxEngine *e, int count, unsigned long long *h_nodes)
{
	AGX_REQUIRE(e != nullptr && h_nodes != nullptr, AGX_ERR_INVALID, "agx_debug_solve_nodes: null argument");
	AGX_REQUIRE(count >= 0 && static_cast<size_t>(count) <= e->debug_solve_nodes.size(), AGX_ERR_INVALID, "agx_debug_solve_nodes: the last agx_debug_solve solved %zu positions",
			e->debug_solve_nodes.size());
	for (int g = 0; g < count; g++)
		h_nodes[g] = e->debug_solve_nodes[g];
	return AGX_OK;
}

/*
 * OpeningGenerator::generate (src/selfplay/OpeningGenerator.cpp:21-78), batched: candidates from prepareOpening(config, 1), each
 * given to the threat solver with a 1000-node budget (:61-62) and redrawn (up to 100 times, :55) while the solver proves it; the
 * survivors are evaluated by the network and, in workspace order, accepted when |E - 0.5| < 0.1 + 0.01 * trials (:41-48).
 */
int agx_engine_generate_openings(AgxEngine *e, AgxNet *net, int count, uint32_t seed, uint16_t *h_openings, int *h_stats)
{
	AGX_REQUIRE(e != nullptr && net != nullptr && h_openings != nullptr, AGX_ERR_INVALID, "agx_engine_generate_openings: null argument");
	AGX_REQUIRE(count > 0, AGX_ERR_INVALID, "agx_engine_generate_openings: count must be positive");
	AGX_REQUIRE(!e->begun, AGX_ERR_STATE, "agx_engine_generate_openings: the pool is playing (the generator borrows its task slots); call it before agx_engine_begin");
	const EngineDev &d = e->dev;
	const int W = std::min(d.n_games, 64); // workspace entries evaluated together (the reference uses max_batch_size of them)
	struct Entry
	{
			uint16_t opening[AGX_OPENING_CAP];
			bool scheduled = false;
			float expectation = 0.0f;
	};
	std::vector<Entry> workspace(W);
	std::vector<uint8_t> boards(static_cast<size_t>(W) * d.hw);
	std::vector<int> signs(W), todo;
	std::vector<DTask> task(1);
	std::vector<float> values(static_cast<size_t>(W) * d.batch * 3);
	uint8_t *d_boards = nullptr;
	int *d_signs = nullptr;
	AGX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_boards), boards.size()));
	AGX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_signs), W * sizeof(int)));
	EngineDev dd = d;
	dd.g0 = 0;
	dd.nn_counter = 16;
	dd.yield_fraction = 0.0f;
	dd.use_symmetries = 0;
	dd.tss_max_nodes = 1000; // solver.setNodeLimit(1000)
	int completed = 0, trials = 0;
	long long stats[4] = { 0, 0, 0, 0 }; // candidates drawn, proven by the solver, rejected as unbalanced, network evaluations
	int status = AGX_OK;
	while (completed < count && status == AGX_OK)
	{
		for (int attempt = 0; attempt < 100; attempt++)
		{
			todo.clear();
			for (int i = 0; i < W; i++)
				if (!workspace[i].scheduled)
					todo.push_back(i);
			if (todo.empty())
				break;
			const int m = static_cast<int>(todo.size());
			for (int j = 0; j < m; j++)
			{
				Entry &en = workspace[todo[j]];
				if ((status = agx_make_opening(d.rules, d.n, seed++, en.opening)) != AGX_OK)
					break;
				stats[0]++;
				uint8_t *b = boards.data() + static_cast<size_t>(j) * d.hw;
				std::memset(b, 0, d.hw);
				for (int k = 0; k < en.opening[0]; k++)
				{
					const uint16_t mv = en.opening[1 + k];
					b[((mv >> 2) & 127) * d.n + ((mv >> 9) & 127)] = mv & 3;
				}
				signs[j] = (en.opening[0] == 0) ? AGX_CROSS : (3 - (en.opening[en.opening[0]] & 3));
			}
			if (status != AGX_OK)
				break;
			AGX_HIP_CHECK(hipMemcpy(d_boards, boards.data(), static_cast<size_t>(m) * d.hw, hipMemcpyHostToDevice));
			AGX_HIP_CHECK(hipMemcpy(d_signs, signs.data(), m * sizeof(int), hipMemcpyHostToDevice));
			hipLaunchKernelGGL(k_debug_load_tasks, dim3(m), dim3(64), 0, nullptr, dd, d_boards, d_signs, m);
			hipLaunchKernelGGL(k_reset_counter, dim3(1), dim3(1), 0, nullptr, dd.counters + dd.nn_counter, static_cast<int*>(nullptr));
			launch_solve(dd, m, nullptr);
			AGX_HIP_CHECK(hipGetLastError());
			// the unproven positions are in the pool's slot list exactly as after a search step: evaluate them
			if ((status = agx_nn_forward_indirect(net, dd.nn_features, dd.nn_list, dd.counters + dd.nn_counter, m * dd.batch, dd.nn_policy, dd.nn_value, nullptr))
					!= AGX_OK)
				break;
			AGX_HIP_CHECK(hipDeviceSynchronize());
			AGX_HIP_CHECK(hipMemcpy(values.data(), dd.nn_value, static_cast<size_t>(m) * dd.batch * 3 * sizeof(float), hipMemcpyDeviceToHost));
			for (int j = 0; j < m; j++)
			{
				AGX_HIP_CHECK(hipMemcpy(task.data(), dd.tasks + static_cast<size_t>(j) * dd.batch, offsetof(DTask, path_node), hipMemcpyDeviceToHost));
				Entry &en = workspace[todo[j]];
				if (task[0].needs_nn)
				{
					const float *v = values.data() + static_cast<size_t>(j) * dd.batch * 3;
					en.expectation = v[0] + 0.5f * v[1];
					en.scheduled = true;
					stats[3]++;
				}
				else
					stats[1]++;
			}
		}
		if (status != AGX_OK)
			break;
		for (int i = 0; i < W && completed < count; i++)
		{
			Entry &en = workspace[i];
			if (!en.scheduled)
				continue; // proven 100 times in a row: the reference leaves such an entry for the next call as well
			const float balance = std::fabs(en.expectation - 0.5f);
			if (balance < (0.1f + 0.01f * trials))
			{
				std::memcpy(h_openings + static_cast<size_t>(completed) * AGX_OPENING_CAP, en.opening, sizeof(en.opening));
				completed++;
				trials = 0;
			}
			else
			{
				trials++;
				stats[2]++;
			}
			en.scheduled = false;
		}
		for (Entry &en : workspace)
			en.scheduled = false; // entries not consumed because enough openings were found
	}
	(void) hipMemcpy(d.games, e->idle_games.data(), e->idle_games.size() * sizeof(GameState), hipMemcpyHostToDevice); // leave the pool idle again
	(void) hipFree(d_boards);
	(void) hipFree(d_signs);
	if (h_stats != nullptr)
		for (int i = 0; i < 4; i++)
			h_stats[i] = static_cast<int>(stats[i]);
	return status;
}

int agx_debug_new_generation(AgxEngine *e)
{ // AlphaBetaSearch::increaseGeneration (AlphaBetaSearch.cpp:63-66) for the positions agx_debug_solve solves: every game's solver table ages by one
	AGX_REQUIRE(e != nullptr, AGX_ERR_INVALID, "agx_debug_new_generation: null engine");
	// (agx_debug_solve leaves the pool idle after every call: the generation lives in the idle template it restores)
	for (GameState &g : e->idle_games)
		g.generation = (g.generation + 1) % 64;
	AGX_HIP_CHECK(hipMemcpy(e->dev.games, e->idle_games.data(), e->idle_games.size() * sizeof(GameState), hipMemcpyHostToDevice));
	return AGX_OK;
}

int agx_debug_pattern_state(AgxEngine *e, const uint8_t *h_boards, const int *h_signs, const uint16_t *h_moves, int count, int n_moves, uint8_t *h_ptypes,
		uint8_t *h_threats, int16_t *h_lists, int lists_stride)
{
	AGX_REQUIRE(e != nullptr, AGX_ERR_INVALID, "agx_debug_pattern_state: null engine");
	AGX_REQUIRE(count > 0 && count <= e->dev.n_games, AGX_ERR_INVALID, "agx_debug_pattern_state: count must be in [1, n_games] (a position uses its game's spill areas)");
	const EngineDev &d = e->dev;
	uint8_t *d_boards = nullptr, *d_pt = nullptr, *d_th = nullptr;
	int *d_signs = nullptr;
	uint16_t *d_moves = nullptr;
	int16_t *d_lists = nullptr;
	const size_t cells = static_cast<size_t>(count) * d.hw;
	AGX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_boards), cells));
	AGX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_signs), count * sizeof(int)));
	AGX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_moves), std::max<size_t>(static_cast<size_t>(count) * n_moves * 2, 16)));
	AGX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_pt), cells * 8));
	AGX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_th), cells * 2));
	AGX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_lists), static_cast<size_t>(count) * lists_stride * 2));
	AGX_HIP_CHECK(hipMemcpy(d_boards, h_boards, cells, hipMemcpyHostToDevice));
	AGX_HIP_CHECK(hipMemcpy(d_signs, h_signs, count * sizeof(int), hipMemcpyHostToDevice));
	if (n_moves > 0)
		AGX_HIP_CHECK(hipMemcpy(d_moves, h_moves, static_cast<size_t>(count) * n_moves * 2, hipMemcpyHostToDevice));
	hipLaunchKernelGGL(k_debug_pattern_state, dim3(count), dim3(64), 0, nullptr, d, d_boards, d_signs, d_moves, n_moves, d_pt, d_th, d_lists, lists_stride);
	AGX_HIP_CHECK(hipGetLastError());
	AGX_HIP_CHECK(hipDeviceSynchronize());
	AGX_HIP_CHECK(hipMemcpy(h_ptypes, d_pt, cells * 8, hipMemcpyDeviceToHost));
	AGX_HIP_CHECK(hipMemcpy(h_threats, d_th, cells * 2, hipMemcpyDeviceToHost));
	AGX_HIP_CHECK(hipMemcpy(h_lists, d_lists, static_cast<size_t>(count) * lists_stride * 2, hipMemcpyDeviceToHost));
	(void) hipFree(d_boards);
	(void) hipFree(d_signs);
	(void) hipFree(d_moves);
	(void) hipFree(d_pt);
	(void) hipFree(d_th);
	(void) hipFree(d_lists);
	return AGX_OK;
}

/* host-only: the lookup tables the engine uploads (for CPU-side verification against the reference) */
int agx_host_tables(int rules, uint8_t *h_pattern, uint8_t *h_half_open_three, uint8_t *h_threat, uint16_t *h_defense)
{
	AGX_REQUIRE(rules >= 0 && rules <= AGX_CARO6, AGX_ERR_INVALID, "agx_host_tables: unknown rules %d", rules);
	HostTables t;
	build_host_tables(rules, t);
	if (h_pattern != nullptr)
		std::memcpy(h_pattern, t.pattern.data(), t.pattern.size());
	if (h_half_open_three != nullptr)
		std::memcpy(h_half_open_three, t.half_open_three.data(), t.half_open_three.size());
	if (h_threat != nullptr)
		std::memcpy(h_threat, t.threat.data(), t.threat.size());
	if (h_defense != nullptr)
		std::memcpy(h_defense, t.defense.data(), t.defense.size() * sizeof(uint16_t));
	return AGX_OK;
}

} /* extern "C" */
