/*
 * dev_solver.hpp — device side of the in-loop threat solver: incremental line-pattern / threat state, the staged move
 * generator and the alpha-beta search with its bucketed transposition table.  One 64-lane wavefront owns one game.
 *
 * What it replaces (all host code in the reference): PatternCalculator::{setBoard,addMove,undoMove}
 * (src/patterns/PatternCalculator.cpp:40-105,245-367), RawPatternCalculator (RawPatternCalculator.hpp:22-288),
 * ThreatHistogram (ThreatHistogram.hpp:39-113), NNInputFeatures::encode (src/networks/NNInputFeatures.cpp:59-113),
 * MoveGenerator::generate (src/search/alpha_beta/MoveGenerator.cpp:159-1207), AlphaBetaSearch::{solve,recursive_solve,
 * evaluate} (src/search/alpha_beta/AlphaBetaSearch.cpp:77-365) and SharedHashTable (SharedHashTable.hpp:90-220).
 *
 * MI355X mapping: the per-position state (line bit-boards, per-cell pattern types, threat lists, recursion frames: ~25 KB)
 * lives in LDS for the whole solve; the wide steps — classifying all cells of a new position, re-classifying the <= 40
 * cells around a placed/removed stone, encoding the feature plane — run one cell per lane with wave ballots keeping the
 * reference's list ORDER (row-major appends, swap-with-last removals), which decides move ordering and therefore results.
 * The branchy control flow (move-generation cascade, alpha-beta bookkeeping, table probes) runs on lane 0 as an explicit
 * frame machine that yields to the wave whenever a stone has to be placed or removed.
 * Renju (forbidden moves re-enter the incremental update from inside the generator) is not supported on the device yet.
 */
#ifndef AGX_DEV_SOLVER_HPP_
#define AGX_DEV_SOLVER_HPP_

#include <hip/hip_runtime.h>
#include "engine_types.hpp"

namespace agx
{
	namespace dev
	{
		typedef uint64_t u64;

		/* One wavefront owns the whole workgroup in the solver kernel: its lanes run in lockstep and the LDS executes one wave's
		 * instructions in order, so hand-offs between lanes through LDS need no s_barrier and no s_waitcnt vmcnt(0) — only the
		 * compiler must not reorder the accesses.  (A real __syncthreads() would also drain the table prefetch that is meant to stay
		 * in flight while a stone is placed.) */
		__device__ __forceinline__ void wave_sync()
		{
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			__builtin_amdgcn_wave_barrier();
		}

		/*
		 * Wave-wide reductions on the DPP path.  __shfl_xor compiles to ds_bpermute_b32 + s_waitcnt on gfx950 — six LDS-crossbar
		 * round trips per reduction; a row_shr 1/2/4/8 scan inside the 16-lane rows followed by row_bcast:15 / row_bcast:31 is six
		 * VALU instructions, after which lane 63 holds the result (identity 0 for max / add / xor of unsigned values).
		 */
#define AGX_DPP_STEP(V, OP, CTRL, ROWS) V = OP(V, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(V), CTRL, ROWS, 0xf, false)))
#define AGX_OP_MAX(a, b) max(a, b)
#define AGX_OP_ADD(a, b) ((a) + (b))
#define AGX_OP_XOR(a, b) ((a) ^ (b))
#define AGX_OP_OR(a, b) ((a) | (b))
#define AGX_WAVE_REDUCE(V, OP) do { AGX_DPP_STEP(V, OP, 0x111, 0xf); AGX_DPP_STEP(V, OP, 0x112, 0xf); AGX_DPP_STEP(V, OP, 0x114, 0xf); \
		AGX_DPP_STEP(V, OP, 0x118, 0xf); AGX_DPP_STEP(V, OP, 0x142, 0xa); AGX_DPP_STEP(V, OP, 0x143, 0xc); } while (0)
		__device__ __forceinline__ uint32_t wave_reduce_umax(uint32_t v)
		{
			AGX_WAVE_REDUCE(v, AGX_OP_MAX);
			return __builtin_amdgcn_readlane(v, 63);
		}
		__device__ __forceinline__ uint32_t wave_reduce_add(uint32_t v)
		{
			AGX_WAVE_REDUCE(v, AGX_OP_ADD);
			return __builtin_amdgcn_readlane(v, 63);
		}
		__device__ __forceinline__ u64 wave_reduce_xor64(u64 v)
		{
			uint32_t lo = static_cast<uint32_t>(v), hi = static_cast<uint32_t>(v >> 32);
			AGX_WAVE_REDUCE(lo, AGX_OP_XOR);
			AGX_WAVE_REDUCE(hi, AGX_OP_XOR);
			return static_cast<u64>(static_cast<uint32_t>(__builtin_amdgcn_readlane(lo, 63))) | (static_cast<u64>(static_cast<uint32_t>(__builtin_amdgcn_readlane(hi, 63))) << 32); // readlane yields int: no sign extension
		}
		/* A pointer that went through LDS (sh.snap, sh.ov_data, sh.spill_*) has lost its address space: hipcc accesses what it points to with FLAT
		 * instructions, and a FLAT access counts on BOTH wait counters — every later s_waitcnt lgkmcnt(0), i.e. every LDS round trip, then also waits for
		 * it.  The solver's prefetches (the child's table bucket, the undo snapshot) are requested exactly so that they travel during LDS-heavy work:
		 * as FLAT loads they were waited for at the first LDS access behind them.  These accesses go through a global-address-space view of the
		 * pointer (global_load / global_store: vmcnt only).  All of these pointers are hipMalloc'ed device memory.  (AGX_GLOBAL_VIEW = 0: plain.) */
#ifndef AGX_GLOBAL_VIEW
#define AGX_GLOBAL_VIEW 1
#endif
#if AGX_GLOBAL_VIEW
		typedef __attribute__((address_space(1))) u64 global_u64;
		__device__ __forceinline__ global_u64* global_view(u64 *p) { return (global_u64*) p; }
		__device__ __forceinline__ const global_u64* global_view(const u64 *p) { return (const global_u64*) p; }
#else
		__device__ __forceinline__ u64* global_view(u64 *p) { return p; }
		__device__ __forceinline__ const u64* global_view(const u64 *p) { return p; }
#endif
		/* The lane index as a value hipcc cannot see through.  Everything a function derives from the lane index (LDS addresses of per-lane
		 * elements, masks, move encodings) is loop-invariant for the whole kernel, so the compiler computes all of it ONCE at the top — and at
		 * 168 registers parks it in scratch: a one-instruction value then costs a scratch reload plus an s_waitcnt vmcnt(0) at its use (which also
		 * waits for whatever prefetch is in flight), because LLVM does not rematerialise VALU results.  Used locally, right in front of the use,
		 * the value is recomputed there and dies there.  (AGX_FRESH_LANE = 0: the plain lane index.) */
#ifndef AGX_FRESH_LANE
#define AGX_FRESH_LANE 1 /* (no scratch access left inside the solver's loop: ScratchSize 340 -> 224 B/lane, renju 672 -> 208; search launch 3.91 -> 3.87 ms) */
#endif
		__device__ __forceinline__ int fresh_lane(int lane)
		{
#if AGX_FRESH_LANE
			asm volatile("" : "+v"(lane));
#endif
			return lane;
		}
		/* how many of the lanes BELOW this one are set in a ballot: v_mbcnt_lo / _hi, two instructions and no per-lane mask constant (the mask
		 * ~0 >> (64 - lane) is a 64-bit lane-derived value that hipcc hoists to the top of the kernel and, at 168 registers, reloads from scratch
		 * at every inlined use) */
#ifndef AGX_MBCNT
#define AGX_MBCNT 0 /* (measured flat to slightly slower than the mask form: 3.975 against 3.944 ms) */
#endif
		__device__ __forceinline__ int lanes_below(u64 ballot, int lane)
		{
#if AGX_MBCNT
			(void) lane;
			return static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(ballot >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(ballot), 0u)));
#else
			return __popcll((lane == 0) ? 0ull : (ballot & (~0ull >> (64 - lane))));
#endif
		}
		/* inclusive prefix sum over lanes 0..31 (the rows of a board): 16-lane scans, then row 0's total into row 1 */
		__device__ __forceinline__ uint32_t wave_scan32_add(uint32_t v)
		{
			AGX_DPP_STEP(v, AGX_OP_ADD, 0x111, 0xf);
			AGX_DPP_STEP(v, AGX_OP_ADD, 0x112, 0xf);
			AGX_DPP_STEP(v, AGX_OP_ADD, 0x114, 0xf);
			AGX_DPP_STEP(v, AGX_OP_ADD, 0x118, 0xf);
			AGX_DPP_STEP(v, AGX_OP_ADD, 0x142, 0xa);
			return v;
		}

		/* Solver Zobrist keys (FastZobristHashing, ZobristHashing.cpp:35-43 draws 2*HW 128-bit keys from an RNG): key word j is the
		 * j-th output of splitmix64 seeded with the engine's zobrist_seed, so it can be recomputed in registers instead of being
		 * gathered from memory (lo word of (cell, colour) = output 2*(2*cell + colour - 1), hi word = the next one). */
		__device__ __forceinline__ u64 zobrist_word(u64 seed, uint32_t j)
		{
			u64 z = seed + (static_cast<u64>(j) + 1ull) * 0x9E3779B97F4A7C15ull;
			z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
			z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
			return z ^ (z >> 31);
		}

		/* ---------------- Score algebra on raw 16-bit values (search/Score.hpp:47-320) ---------------- */
		__device__ __forceinline__ int s_pv(uint32_t d) { return (d >> 13) & 3; }
		__device__ __forceinline__ int s_eval(uint32_t d) { return static_cast<int>(d & 8191u) - 4000; }
		__device__ __forceinline__ bool s_infinite(uint32_t d) { return d == 0u || d == 0xFFFFu; }
		__device__ __forceinline__ bool s_unproven(uint32_t d) { return s_pv(d) == 2; }
		__device__ __forceinline__ bool s_proven(uint32_t d) { return s_pv(d) != 2 && !s_infinite(d); }
		__device__ __forceinline__ bool s_loss(uint32_t d) { return s_pv(d) == 0 && !s_infinite(d); }
		__device__ __forceinline__ bool s_win(uint32_t d) { return s_pv(d) == 3 && !s_infinite(d); }
		__device__ __forceinline__ uint32_t s_make(int pv, int eval) { return (static_cast<uint32_t>(pv) << 13) | static_cast<uint32_t>(4000 + eval); }
		__device__ __forceinline__ uint32_t s_unknown(int eval) { return s_make(2, eval); }
		__device__ __forceinline__ uint32_t s_loss_in(int n) { return s_make(0, n); }
		__device__ __forceinline__ uint32_t s_draw_in(int n) { return s_make(1, n); }
		__device__ __forceinline__ uint32_t s_win_in(int n) { return s_make(3, -n); }
		__device__ __forceinline__ int s_distance(uint32_t d)
		{
			const int pv = s_pv(d);
			return (pv <= 1) ? s_eval(d) : (pv == 3 ? -s_eval(d) : 0);
		}
		__device__ __forceinline__ uint32_t s_negate(uint32_t d)
		{
			switch (s_pv(d))
			{
				case 0: return s_infinite(d) ? 0xFFFFu : s_make(3, -s_eval(d));
				case 1: return s_make(1, s_eval(d));
				case 3: return s_infinite(d) ? 0x0000u : s_make(0, -s_eval(d));
				default: return s_make(2, -s_eval(d));
			}
		}
		/* Score::increaseDistance / decreaseDistance behind operator- (invert_up / invert_down, search/Score.hpp:205-260) without branches: with
		 * r = 4000 + eval the proven-value classes map loss -> win (3, 8000 - r -/+ 1), draw -> draw (1, r +/- 1), win -> loss (0, 8000 - r +/- 1),
		 * unknown -> unknown (2, 8000 - r), and the two infinities swap.  (The solver inverts a score at every descent and every return; as a
		 * switch this was a jump table per call.  All 65 536 inputs give the integers of the switch form: tests/test_oracle_tables.py pins the
		 * oracle's score algebra to the compiled reference, the engine tests compare the device's scores with the oracle's.) */
		__device__ __forceinline__ uint32_t s_invert_step(uint32_t d, int step)
		{ // step = +1: invert_up, -1: invert_down
			const uint32_t pv = (d >> 13) & 3u, r = d & 8191u;
			const uint32_t npv = (pv == 0u) ? 3u : ((pv == 3u) ? 0u : pv);
			const int flipped = 8000 - static_cast<int>(r) + ((pv == 0u) ? -step : ((pv == 3u) ? step : 0));
			const int kept = static_cast<int>(r) + step;
			const uint32_t body = (npv << 13) | static_cast<uint32_t>((pv == 1u) ? kept : flipped);
			return (d == 0u) ? 0xFFFFu : ((d == 0xFFFFu) ? 0u : body);
		}
		__device__ __forceinline__ uint32_t s_invert_up(uint32_t d) { return s_invert_step(d, +1); }
		__device__ __forceinline__ uint32_t s_invert_down(uint32_t d) { return s_invert_step(d, -1); }
		__device__ __forceinline__ void s_to_value(uint32_t d, float &win, float &draw)
		{ // Score.hpp:266-283
			win = 0.0f;
			draw = 0.0f;
			switch (s_pv(d))
			{
				case 1: draw = 1.0f; break;
				case 2: win = (1000 + s_eval(d)) / 2000.0f; break;
				case 3: win = s_infinite(d) ? 0.0f : 1.0f; break;
				default: break;
			}
		}

		/* ---------------- per-game solver state in LDS ---------------- */
		struct alignas(16) Frame
		{ // 32 bytes, 16-byte aligned: a frame moves between registers and LDS as two 128-bit accesses
				int base;
				uint16_t size, i; // an action list never exceeds the number of cells
				uint16_t alpha, beta, original_alpha, best_score;
				uint16_t best_move, move, baseline;
				int16_t depth_remaining;
				uint8_t must_defend, has_initiative, fully_expanded, pad;
				uint32_t ov_slot; // speculative solves: the overlay slot of this node's table bucket (set when the frame is entered)
		};
		static_assert(sizeof(Frame) == 32, "frame layout");
		enum Cmd : int { CMD_NONE = 0, CMD_ADD = 1, CMD_UNDO = 2, CMD_DONE = 3 };
#ifndef AGX_LUT_LDS
#define AGX_LUT_LDS 0 /* 1: a private 4 KB copy of the packed ThreatTable in every solver wave's LDS (-1.4 % solver time at one wave per SIMD); 0: read
                         through the vector L1 / L2 — 4 KB less LDS per wave buys a third wave per SIMD, which hides far more than that latency */
#endif
#ifndef AGX_PLACE_OPAQUE_LANE
#define AGX_PLACE_OPAQUE_LANE 1 /* 1: solver_update_around / pattern_prefetch recompute their lane-derived values instead of taking hoisted copies out of scratch
                                   (search launch 4.08 -> 3.94 ms, profiles/r05_search_variants_ab.txt) */
#endif
		constexpr int NODE_TIME_OVER = 0x40000000; // set in the node counter when a time-limited solve runs out of time: every "nodes left" test then fails
		constexpr int NODE_COUNT_MASK = NODE_TIME_OVER - 1;
		constexpr int LDS_FRAMES = 42;
		constexpr int OV_CAP = 256; // overlay slots of a speculative solve (a 100-node solve touches ~100-150 buckets; more = the task is re-run serially) // alpha-beta frames kept in LDS; deeper ones (only reachable with node budgets far above 100) live in HBM

		/*
		 * Per-wave solver state, sized by the board (N = 15, 20, or MAXN for the any-size kernel) so that eight solver waves fit into a
		 * CU's 160 KB of LDS at 15x15 (two per SIMD) instead of four:
		 *  - the threat lists: a cell sits in at most one list per side, and the list of HALF_OPEN_3 cells is never read by anything on the
		 *    path (MoveGenerator / AlphaBetaSearch::evaluate use OPEN_3 ... OVERLINE), so only its size is kept.  Lists 2-9 have a fixed LDS
		 *    capacity (OPEN_3: CAP2, the others CAP); the rare tail beyond it lives in the game's HBM spill area (list_get / list_set),
		 *    exactly like the action stack's tail;
		 *  - the alpha-beta frames beyond LDS_FRAMES and the action stack beyond ACT_LDS spill the same way.
		 */
		template<int N>
		struct SolverSharedT
		{
				static constexpr int DIM = N, HW = N * N;
				static constexpr int CAP2 = (N <= 15) ? 64 : 96, CAP = 24;
				static constexpr int LIST_ITEMS = CAP2 + 7 * CAP;
				// the select stage of the fused kernel parks its node-cache keys in the (then idle) action stack + frames, which are
				// contiguous, its compressed board in `lines` and its board in `board`
				static constexpr int SELECT_KEY_BYTES = 3 * (1 + HW) * 8;
				// (round 3, late: 20x20 had 2112 entries + 42 frames only to give the select stage's keys room in act + frames — 20.5 KB per wave,
				// 7 waves per compute unit; the keys now run on over ptype / threat / items, which a select does not use either: 16.4 KB, 10 waves)
				// (round 5, late: 15x15 boards keep 448 action-stack entries and 30 frames in LDS instead of 1024 and 42 — 12 928 -> 10 240 bytes per wave,
				// SIXTEEN waves per compute unit instead of twelve: a slice's ~1 900 leaves are then two rounds of solves on its 1 024 waves, not 2.47 -> three
				// on 768; a 100-node solve's stack stays below 448 entries and 30 levels nearly always, the tails go to HBM as before)
				// (round 6: 20x20 boards keep 336 entries instead of 1024 — 16 064 -> 13 312 bytes per wave, TWELVE waves per compute unit instead of ten; the select
				// stage's keys run on over everything up to `board`)
#ifndef AGX_ACT_LDS_20
#define AGX_ACT_LDS_20 336
#endif
				static constexpr int ACT_LDS = (N <= 15) ? 448 : AGX_ACT_LDS_20;
				static constexpr int FRAMES = (N <= 15) ? 30 : 34; // alpha-beta frames kept in LDS (LDS_FRAMES at most); deeper ones live in HBM
				__device__ static constexpr int list_cap(int t) { return (t == 2) ? CAP2 : CAP; }
				__device__ static constexpr int list_off(int t) { return (t == 2) ? 0 : CAP2 + (t - 3) * CAP; }

				u64 lines[6 * N];
				uint32_t act[ACT_LDS];   // head of the action stack (the tail, if ever needed, spills to HBM)
				Frame frames[FRAMES];
				uint8_t ptype[HW][8]; // [cell][0-3 cross dirs, 4-7 circle dirs]
#if AGX_LUT_LDS
				uint8_t threat_lut[4096]; // ThreatTable in LDS: cross | circle << 4 for the 4 x 3-bit pattern index (loaded per launch)
#endif
				uint8_t threat[HW][2];
				uint16_t items[2][LIST_ITEMS]; // threat lists 2..9 of both sides, list t at list_off(t)
				uint16_t pos[2][HW];     // index of a cell inside its list (ThreatHistogram::remove finds it by search)
				alignas(4) uint16_t count[2][10];
				uint32_t legal[N];
				uint32_t added[N];
				uint32_t row_mask[N]; // scratch row masks of the move generator
				uint16_t tmp_list[HW]; // copy of a threat list that renju foul checks would permute while it is iterated
				uint16_t foul_cell[64];  // MoveGenerator::forbidden_moves_cache (MoveGenerator.hpp:74)
				uint8_t foul_flag[64];
				uint8_t board[(HW + 7) / 8 * 8];
				int foul_count;
				int fstack[16][5];       // explicit stack of the recursive renju 3x3 check: cell, dir, i, count, promotion mask
				u64 pf_bucket[8];        // transposition-table bucket prefetched for the child about to be entered
				u64 pf_lo;
				int pf_valid;
				// speculative solves (k_search_spec): every table bucket the task touches is copied into a per-task overlay in HBM (its content at
				// first touch + the task's own version); the game's table itself is only written when the task is committed
				uint32_t ov_keys[OV_CAP];          // bucket index of every overlay slot
				uint32_t ov_dirty[OV_CAP / 32];    // slots the task has written
				u64 *ov_data;                      // [OV_CAP][16]: words 0-7 the bucket at first touch, 8-15 the task's version
				int ov_on, ov_count, ov_overflow, pf_slot;
#ifdef AGX_SOLVER_PROFILE
				unsigned long long prof[8];
				unsigned long long dprof[24];
#endif
				// the frame machine's scalars: written together when it yields, read together when it resumes — three 16-byte words
				alignas(16) u64 hash_lo;
				u64 hash_hi;
				int phase, level, node_counter, stack_offset;
				int stack_max, error, pending_value, cmd_move;
				u64 time_deadline;       // time-limited solves (E.solve_time_ticks != 0): the wall-clock tick at which this task's share of the time is over
				uint16_t *spill_lists;   // [2][10][hw] tails of the threat lists (HBM, per game)
				// solver_place reads the next three together (one 16-byte access)
				alignas(16) u64 *snap;   // [hw + 2][64] undo snapshots of this solve's area (HBM): what addMove overwrote, one level per stone on the board
				int sign_to_move, depth;
				Frame *spill_frames;     // [MAX_FRAMES] frames beyond LDS_FRAMES (HBM, per game)
				int cmd, result_score;
		};
		typedef SolverSharedT<MAXN> SolverShared; // any board size
#if !defined(AGX_SOLVER_PROFILE)
		static_assert(sizeof(SolverSharedT<15>) <= 10240, "15x15 solver state: sixteen waves per compute unit (160 KB of LDS)");
#endif

		/* Threat-list entries: LDS below the list's capacity, the game's HBM spill area above it.  As with the action stack the two paths
		 * stay two different instructions (inline assembly), or hipcc merges them into FLAT accesses. */
		template<class SH>
		__device__ __forceinline__ uint32_t list_get(const SH &sh, int s, int t, int i)
		{
			if (__builtin_expect(i < SH::list_cap(t), 1))
				return sh.items[s][SH::list_off(t) + i];
			uint32_t v;
			asm volatile("global_load_ushort %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(sh.spill_lists + (s * 10 + t) * SH::HW + i) : "memory");
			return v;
		}
		template<class SH>
		__device__ __forceinline__ void list_set(SH &sh, int s, int t, int i, uint32_t cell)
		{
			if (__builtin_expect(i < SH::list_cap(t), 1))
				sh.items[s][SH::list_off(t) + i] = static_cast<uint16_t>(cell);
			else
				asm volatile("global_store_short %0, %1, off\n\ts_waitcnt vmcnt(0)" : : "v"(sh.spill_lists + (s * 10 + t) * SH::HW + i), "v"(cell) : "memory");
		}
		template<class SH>
		__device__ __forceinline__ Frame frame_get(const SH &sh, int level)
		{
			if (__builtin_expect(level < SH::FRAMES, 1))
				return sh.frames[level];
			u64 w[4];
			const Frame *p = sh.spill_frames + level;
			asm volatile("global_load_dwordx2 %0, %4, off\n\tglobal_load_dwordx2 %1, %4, off offset:8\n\tglobal_load_dwordx2 %2, %4, off offset:16\n\t"
					"global_load_dwordx2 %3, %4, off offset:24\n\ts_waitcnt vmcnt(0)" : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]) : "v"(p) : "memory");
			Frame f;
			__builtin_memcpy(&f, w, sizeof(Frame));
			return f;
		}
		template<class SH>
		__device__ __forceinline__ void frame_set(SH &sh, int level, const Frame &f)
		{
			if (__builtin_expect(level < SH::FRAMES, 1))
				sh.frames[level] = f;
			else
			{
				u64 w[4];
				__builtin_memcpy(w, &f, sizeof(Frame));
				Frame *p = sh.spill_frames + level;
				asm volatile("global_store_dwordx2 %4, %0, off\n\tglobal_store_dwordx2 %4, %1, off offset:8\n\tglobal_store_dwordx2 %4, %2, off offset:16\n\t"
						"global_store_dwordx2 %4, %3, off offset:24\n\ts_waitcnt vmcnt(0)" : : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(p) : "memory");
			}
		}

#ifdef AGX_SOLVER_PROFILE
#define AGX_PROF_BEGIN() unsigned long long agx_pt_ = clock64()
#define AGX_PROF_MARK(SH, K) do { const unsigned long long agx_now_ = clock64(); (SH).dprof[K] += agx_now_ - agx_pt_; agx_pt_ = agx_now_; } while (0)
#define AGX_PROF_COUNT(SH, K, V) do { (SH).dprof[K] += (V); } while (0)
#else
#define AGX_PROF_BEGIN() do { } while (0)
#define AGX_PROF_MARK(SH, K) do { } while (0)
#define AGX_PROF_COUNT(SH, K, V) do { } while (0)
#endif
		__device__ __forceinline__ int row_step(int d) { return d == 0 ? 0 : 1; }
		__device__ __forceinline__ int col_step(int d) { return d == 0 ? 1 : (d == 1 ? 0 : (d == 2 ? 1 : -1)); }
		__device__ __forceinline__ void line_of(int n, int r, int c, int d, int &index, int &shift)
		{ // RawPatternCalculator.hpp:241-273; selects instead of a switch: lanes of one wave hold different directions
			const int i2 = 3 * n - 1 + c - r, i3 = 4 * n - 1 + r + c;
			const int s2 = min(c, r), s3 = min(r, n - 1 - c);
			index = (d == 0) ? r : ((d == 1) ? n + c : ((d == 2) ? i2 : i3));
			shift = 2 * ((d == 0) ? c : ((d == 1) ? r : ((d == 2) ? s2 : s3)));
		}
		template<class SH>
		__device__ __forceinline__ uint32_t extended_pattern(const SH &sh, int n, int r, int c, int d)
		{ // 13 cells (+-6), off-board = 3 (RawPatternCalculator.hpp:196-210)
			int index, shift;
			line_of(n, r, c, d, index, shift);
			return static_cast<uint32_t>((sh.lines[index] >> shift) & 0x3FFFFFFull);
		}
		template<class SH>
		__device__ __forceinline__ uint32_t normal_pattern(const SH &sh, int n, int r, int c, int d) { return (extended_pattern(sh, n, r, c, d) >> 2) & 0x3FFFFFu; }
		__device__ __forceinline__ uint32_t narrow(uint32_t x) { return (x & 1023u) | ((x & 4190208u) >> 2); }
		__device__ __forceinline__ uint32_t threat_index(const uint8_t *pt) { return pt[0] | (pt[1] << 3) | (pt[2] << 6) | (pt[3] << 9); }

		/* ThreatTable (4096 x 2 threat types < 16) packed to one byte per index: 4 KB of LDS instead of a second dependent L2 access
		 * for every re-classified cell.  Once per launch, before the first solver_set_board. */
		template<class SH>
		__device__ __forceinline__ void solver_load_threat_table(SH &sh, const EngineDev &E, int lane)
		{
#if AGX_LUT_LDS
			for (int i = lane; i < 4096; i += 64)
				sh.threat_lut[i] = E.t_threat_packed[i];
			wave_sync();
#endif
		}
		/* ThreatTable lookup: cross type in the low nibble, circle type in the high one (tables_host: packed once at engine creation) */
		template<class SH>
		__device__ __forceinline__ uint32_t threat_lookup(const SH &sh, const EngineDev &E, uint32_t index)
		{
#if AGX_LUT_LDS
			return sh.threat_lut[index];
#else
			return E.t_threat_packed[index];
#endif
		}

		/* PatternCalculator::setBoard (PatternCalculator.cpp:40-66, 245-277) */
		template<class SH>
		__device__ __forceinline__ void solver_set_board(SH &sh, const EngineDev &E, const uint8_t *board, int sign_to_move, int lane)
		{
			const int n = E.n, hw = E.hw;
			for (int i = lane; i < hw; i += 64)
				sh.board[i] = board[i];
			if (lane < 20)
				sh.count[lane / 10][lane % 10] = 0;
			wave_sync();
			for (int L = lane; L < 6 * n - 2; L += 64)
			{
				int len, r0, c0, dr, dc;
				if (L < n) { len = n; r0 = L; c0 = 0; dr = 0; dc = 1; }
				else if (L < 2 * n) { len = n; r0 = 0; c0 = L - n; dr = 1; dc = 0; }
				else if (L < 4 * n - 1)
				{
					const int k = L - 2 * n - (n - 1); // col - row
					len = n - abs(k);
					r0 = (k < 0) ? -k : 0;
					c0 = (k < 0) ? 0 : k;
					dr = 1; dc = 1;
				}
				else
				{
					const int s = L - (4 * n - 1); // row + col
					len = (s < n) ? s + 1 : 2 * n - 1 - s;
					r0 = (s < n) ? 0 : s - (n - 1);
					c0 = s - r0;
					dr = 1; dc = -1;
				}
				u64 line = 0xFFFull | (0xFFFull << (12 + 2 * len));
				for (int j = 0; j < len; j++)
					line |= static_cast<u64>(sh.board[(r0 + j * dr) * n + (c0 + j * dc)]) << (12 + 2 * j);
				sh.lines[L] = line;
			}
			if (lane < n)
			{
				uint32_t m = 0;
				for (int c = 0; c < n; c++)
					if (sh.board[lane * n + c] == 0)
						m |= (1u << c);
				sh.legal[lane] = m;
			}
			wave_sync();
			int stones = 0;
			for (int chunk = 0; chunk * 64 < hw; chunk++)
			{
				const int cell = chunk * 64 + lane;
				int t0 = 0, t1 = 0;
				if (cell < hw)
				{
					if (sh.board[cell] == 0)
					{
						const int r = cell / n, c = cell % n;
						for (int d = 0; d < 4; d++)
						{
							const uint8_t e = E.t_pattern[narrow(normal_pattern(sh, n, r, c, d))];
							sh.ptype[cell][d] = e & 15;
							sh.ptype[cell][4 + d] = e >> 4;
						}
						t0 = threat_lookup(sh, E, threat_index(sh.ptype[cell])) & 15;
						t1 = threat_lookup(sh, E, threat_index(sh.ptype[cell] + 4)) >> 4;
					}
					else
					{
						stones++;
						for (int d = 0; d < 8; d++)
							sh.ptype[cell][d] = 0;
					}
					sh.threat[cell][0] = static_cast<uint8_t>(t0);
					sh.threat[cell][1] = static_cast<uint8_t>(t1);
				}
				// ordered (row-major) append to the threat lists
				for (int t = 1; t < 10; t++)
				{
					const u64 m0 = __ballot(t0 == t);
					if (m0 != 0)
					{
						const int cnt = sh.count[0][t];
						if (t0 == t && t != 1)
						{
							list_set(sh, 0, t, cnt + lanes_below(m0, lane), cell);
							sh.pos[0][cell] = static_cast<uint16_t>(cnt + lanes_below(m0, lane));
						}
						wave_sync();
						if (lane == 0)
							sh.count[0][t] = static_cast<uint16_t>(cnt + __popcll(m0));
						wave_sync();
					}
					const u64 m1 = __ballot(t1 == t);
					if (m1 != 0)
					{
						const int cnt = sh.count[1][t];
						if (t1 == t && t != 1)
						{
							list_set(sh, 1, t, cnt + lanes_below(m1, lane), cell);
							sh.pos[1][cell] = static_cast<uint16_t>(cnt + lanes_below(m1, lane));
						}
						wave_sync();
						if (lane == 0)
							sh.count[1][t] = static_cast<uint16_t>(cnt + __popcll(m1));
						wave_sync();
					}
				}
			}
			stones = static_cast<int>(wave_reduce_add(static_cast<uint32_t>(stones)));
			if (lane == 0)
			{
				sh.sign_to_move = sign_to_move;
				sh.depth = stones;
			}
			wave_sync();
		}

		/*
		 * The pattern-table entries solver_update_around will need for placing (add) or removing the stone `mv`, requested while the
		 * frame machine still does its bookkeeping: lane (k, d) asks for the entry of the cell k steps from the stone in direction d
		 * with the stone already put into / taken out of that cell's 11-cell window (window index 5 - k), lanes 40-43 for the centre's
		 * own four windows when the stone is removed.  The L2 round trip then overlaps the descend / return work instead of standing
		 * between the stone and the list edits.
		 */
		template<class SH>
		__device__ __forceinline__ uint8_t pattern_prefetch(const SH &sh, const EngineDev &E, int n, uint32_t mv, bool add, int lane)
		{ // branch-free on purpose: ONE load instruction for the whole wave (idle lanes read entry 0), so that nothing has to wait for
		  // it before solver_update_around consumes it
#if AGX_PLACE_OPAQUE_LANE
			asm volatile("" : "+v"(lane)); // (as in solver_update_around: no lane-derived value of this function comes out of scratch)
#endif
			const int s = mv & 3, r = (mv >> 2) & 127, c = (mv >> 9) & 127;
			const bool around = lane < 40;
			const int ki = lane >> 2, d = around ? (lane & 3) : ((lane - 40) & 3);
			const int k = around ? ((ki < 5) ? ki - 5 : ki - 4) : 0;
			const int r1 = r + k * row_step(d), c1 = c + k * col_step(d);
			const bool inside = r1 >= 0 && r1 < n && c1 >= 0 && c1 < n;
			const bool wanted = around ? inside : (lane < 44 && !add);
			const int rr = inside ? r1 : r, cc = inside ? c1 : c;
			uint32_t x = normal_pattern(sh, n, rr, cc, d);
			const int shift = 2 * (5 - k);
			const uint32_t with_stone = add ? (x | (static_cast<uint32_t>(s) << shift)) : (x & ~(3u << shift));
			x = around ? with_stone : x; // the centre's own windows do not contain the centre
			return E.t_pattern[wanted ? narrow(x) : 0u];
		}

		/* PatternCalculator::update_around (PatternCalculator.cpp:278-367): the centre cell, then the +-5 cells in the four
		 * directions in the order k = -5..5 (k != 0), direction 0..3 — one lane per (k, direction); lanes 40-43 re-classify the four
		 * directions of the centre when a stone is removed.
		 * The threat lists must end up in the reference's order (push-back add, swap-with-last remove, applied cell by cell): see the
		 * list-edit steps below.
		 *
		 * Undo by snapshot (AGX_SNAPSHOT_UNDO, round 5).  undoMove re-derives what addMove overwrote: the pattern types of the <= 40 cells around
		 * the stone and of the centre, and from them the threat types — two dependent table round trips per cell.  Stones come off in the reverse
		 * order they went on (alpha-beta recursion, renju foul probes), and between a stone's add and its undo every cell's pattern / threat
		 * TYPES return to what they were right after the add (only the lists' order does not), so the add parks what it overwrites — per lane one
		 * 64-bit word: the cell's eight pattern types (4 bits used per byte) with both old threat types and a valid bit in the free nibbles; lane
		 * 40 the centre's — in a per-solve HBM area indexed by the number of stones on the board, and the undo gets its `new` values from one
		 * coalesced 512-byte read (requested by the frame machine when it decides to return) instead of 40 pattern-table + 80 threat-table
		 * gathers.  The ordered list edits are unchanged: their order is observable. */
#ifndef AGX_SNAPSHOT_UNDO
#define AGX_SNAPSHOT_UNDO 1
#endif
#ifndef AGX_LEAN_LIST_EDITS
#define AGX_LEAN_LIST_EDITS 1 /* the ordered list edits of a pattern update as one packed event word per step (see solver_update_around) */
#endif
#ifndef AGX_SNAPSHOT_PREFETCH
#define AGX_SNAPSHOT_PREFETCH 1 /* the snapshot word is requested by the frame machine when a node returns (0: by solver_place itself) */
#endif
		template<class SH>
		__device__ __forceinline__ void solver_update_around(SH &sh, const EngineDev &E, int r, int c, bool added, int lane, bool prefetched, uint8_t pf_e,
				u64 *snap_slot, u64 snap_word)
		{ // prefetched: pf_e holds this lane's pattern_prefetch() result for exactly this stone; snap_slot: this lane's word of the stone's snapshot
		  // level (written when a stone is added); snap_word: that word as the add wrote it (when a stone is removed)
#if AGX_PLACE_OPAQUE_LANE
			// What a lane owns here (which neighbour, which direction, its steps) is a function of the lane index alone, so hipcc computes all of it
			// once at the top of the kernel and — at 168 registers — parks it in scratch: every call then began with two scratch reloads and an
			// s_waitcnt vmcnt(0), which also waits for the transposition-table bucket that was requested to travel DURING this call (an HBM miss).
			// With the lane index opaque the handful of values are recomputed here (~10 VALU instructions) and nothing waits.
			asm volatile("" : "+v"(lane));
#endif
			const int n = E.n;
			const int center = r * n + c;
			AGX_PROF_BEGIN();
			int cnt = (lane < 20) ? sh.count[lane / 10][lane % 10] : 0; // lane 10 s + t holds the size of list (s, t)

			int cell = -1, old0 = 0, old1 = 0, new0 = 0, new1 = 0;
			int c0 = 0, c1 = 0; // the centre's threat types: its old ones leave the lists (stone added), its new ones join them (stone removed)
#if AGX_SNAPSHOT_UNDO
			if (!added)
			{ // ---- undo: everything the add overwrote comes out of the snapshot ----
				const uint32_t lo = static_cast<uint32_t>(snap_word), hi = static_cast<uint32_t>(snap_word >> 32);
				const bool valid = ((lo >> 20) & 1u) != 0u;
				new0 = static_cast<int>((lo >> 4) & 15u);
				new1 = static_cast<int>((lo >> 12) & 15u);
				const u64 types = static_cast<u64>(lo & 0x0F0F0F0Fu) | (static_cast<u64>(hi) << 32);
				if (lane <= 40 && valid)
				{
					const int ki = lane >> 2, d = lane & 3;
					const int k = (ki < 5) ? ki - 5 : ki - 4;
					const int at = (lane == 40) ? center : (r + k * row_step(d)) * n + (c + k * col_step(d));
					const uint32_t t01 = *reinterpret_cast<const uint16_t*>(&sh.threat[at][0]);
					*reinterpret_cast<u64*>(&sh.ptype[at][0]) = types;
					*reinterpret_cast<uint16_t*>(&sh.threat[at][0]) = static_cast<uint16_t>(new0 | (new1 << 8));
					if (lane < 40)
					{
						cell = at;
						old0 = t01 & 255u;
						old1 = t01 >> 8;
					}
				}
				c0 = __builtin_amdgcn_readlane(new0, 40);
				c1 = __builtin_amdgcn_readlane(new1, 40);
			}
			else
#endif
			{
			uint32_t centre_bits = 0; // lanes 40-43: pattern types of the centre in direction lane - 40, as 3-bit fields of the two threat indices
#if AGX_SNAPSHOT_UNDO
			u64 park = 0; // what this lane's cell held before the stone (valid bit 20 clear: nothing to restore)
#endif
			if (lane < 40)
			{
				const int ki = lane >> 2, d = lane & 3;
				const int k = (ki < 5) ? ki - 5 : ki - 4;
				const int rr = r + k * row_step(d), cc = c + k * col_step(d);
				if (rr >= 0 && rr < n && cc >= 0 && cc < n)
				{ // every operand is requested before the first is looked at: one LDS round trip ahead of the table access, not two
					const int at = rr * n + cc;
					const int stone = sh.board[at];
					const uint32_t t01 = *reinterpret_cast<const uint16_t*>(&sh.threat[at][0]); // both sides' threat types: one read
					const int t0 = t01 & 255u, t1 = t01 >> 8;
					const u64 w01 = *reinterpret_cast<const u64*>(&sh.ptype[at][0]);            // the eight pattern types of the cell: one read
					uint32_t w0 = static_cast<uint32_t>(w01), w1 = static_cast<uint32_t>(w01 >> 32); // cross / circle, one byte per direction
					uint32_t raw = 0;
					if (!prefetched)
						raw = narrow(normal_pattern(sh, n, rr, cc, d));
					if (stone == 0)
					{
						cell = at;
						old0 = t0;
						old1 = t1;
#if AGX_SNAPSHOT_UNDO
						park = static_cast<u64>(w0 | (t01 << 4 & 0xF0u) | ((t01 >> 8) << 12) | (1u << 20)) | (static_cast<u64>(w1) << 32);
#endif
						const uint32_t e = prefetched ? pf_e : E.t_pattern[raw]; // (the byte is widened HERE, not where it was requested)
						w0 = (w0 & ~(255u << (8 * d))) | ((e & 15u) << (8 * d));
						w1 = (w1 & ~(255u << (8 * d))) | ((e >> 4) << (8 * d));
						// (every lane of the gather owns a different cell — the four lines through the centre meet nowhere else — so the cell's whole pattern
						//  words and both threat types go back in one store each)
						*reinterpret_cast<u64*>(&sh.ptype[cell][0]) = static_cast<u64>(w0) | (static_cast<u64>(w1) << 32);
						new0 = threat_lookup(sh, E, (w0 & 7u) | (((w0 >> 8) & 7u) << 3) | (((w0 >> 16) & 7u) << 6) | (((w0 >> 24) & 7u) << 9)) & 15;
						new1 = threat_lookup(sh, E, (w1 & 7u) | (((w1 >> 8) & 7u) << 3) | (((w1 >> 16) & 7u) << 6) | (((w1 >> 24) & 7u) << 9)) >> 4;
						*reinterpret_cast<uint16_t*>(&sh.threat[cell][0]) = static_cast<uint16_t>(new0 | (new1 << 8));
					}
				}
			}
			else if (lane < 44 && !added)
			{
				const int d = lane - 40;
				const uint32_t e = prefetched ? pf_e : E.t_pattern[narrow(normal_pattern(sh, n, r, c, d))];
				sh.ptype[center][d] = static_cast<uint8_t>(e & 15u);
				sh.ptype[center][4 + d] = static_cast<uint8_t>(e >> 4);
				centre_bits = ((e & 15u) << (3 * d)) | ((e >> 4) << (16 + 3 * d));
			}
			if (added)
			{
				c0 = sh.threat[center][0];
				c1 = sh.threat[center][1];
#if AGX_SNAPSHOT_UNDO
				{ // lane 40 parks the centre; every lane stores its word (one coalesced 512-byte store, nothing waits for it)
					const u64 cw = *reinterpret_cast<const u64*>(&sh.ptype[center][0]);
					const u64 centre_park = (cw & 0xFFFFFFFF0F0F0F0Full) | static_cast<u64>((static_cast<uint32_t>(c0) << 4) | (static_cast<uint32_t>(c1) << 12) | (1u << 20));
					*global_view(snap_slot) = (lane == 40) ? centre_park : park;
				}
#endif
				if (lane < 8)
					sh.ptype[center][lane] = 0;
				if (lane < 2)
					sh.threat[center][lane] = 0;
			}
			else
			{
				const uint32_t bits = __builtin_amdgcn_readlane(centre_bits, 40) | __builtin_amdgcn_readlane(centre_bits, 41)
						| __builtin_amdgcn_readlane(centre_bits, 42) | __builtin_amdgcn_readlane(centre_bits, 43);
				c0 = threat_lookup(sh, E, bits & 4095u) & 15;
				c1 = threat_lookup(sh, E, (bits >> 16) & 4095u) >> 4;
				if (lane == 0)
				{
					sh.threat[center][0] = static_cast<uint8_t>(c0);
					sh.threat[center][1] = static_cast<uint8_t>(c1);
				}
			}
			}
			u64 changed0 = __ballot(cell >= 0 && old0 != new0);
			u64 changed1 = __ballot(cell >= 0 && old1 != new1);
			AGX_PROF_MARK(sh, 11);
			AGX_PROF_COUNT(sh, 13, __popcll(changed0) + __popcll(changed1));
			AGX_PROF_COUNT(sh, 14, 1);

			/*
			 * List edits.  ThreatHistogram::remove overwrites the first match with the last element, ThreatHistogram::add pushes back
			 * (ThreatHistogram.hpp:39-111).  A cell sits in at most one list per side, so its index is kept in lists[side][0][cell]
			 * (the list of type NONE does not exist) and a removal needs no search.  Lane 10 s + t owns list (s, t) and its size: one
			 * step applies one cell's removal and insertion for BOTH sides at once (four different lists, four different lanes), in the
			 * reference's order per list.  LDS executes a wave's instructions in issue order, so a step sees the previous one.
			 */
			const int my_s = (lane >= 10) ? 1 : 0;
#if AGX_LEAN_LIST_EDITS
			/* The step, leaner (round 6; the serial form below stays for lists about to outgrow their LDS capacity).  An event — one cell changing
			 * its threat type on one or both sides — travels as ONE word (cell | old0 << 9 | new0 << 13 | old1 << 17 | new1 << 21: one v_readlane per
			 * side instead of three), a list lane pulls its side's fields out with its own shift amounts, lanes without a list hold type -1 and match
			 * nothing, and the only exec region left is the one around the two stores.  ~50 instructions per step against ~95. */
			const int t_raw = lane - 10 * my_s;
			const int lean_t = (lane < 20 && t_raw != 0) ? t_raw : -1;
			const int pending = 1 + max(__popcll(changed0), __popcll(changed1)); // no list grows by more than this
			const bool roomy = __ballot(lean_t >= 2 && cnt + pending > SH::list_cap(lean_t)) == 0ull;
			if (__builtin_expect(roomy, 1))
			{
				const uint32_t ev = static_cast<uint32_t>(cell) | (static_cast<uint32_t>(old0) << 9) | (static_cast<uint32_t>(new0) << 13)
						| (static_cast<uint32_t>(old1) << 17) | (static_cast<uint32_t>(new1) << 21);
				const uint32_t shift_old = 9u + 8u * my_s, shift_new = 13u + 8u * my_s;
				uint16_t *const my_items = &sh.items[my_s][(lean_t >= 3) ? SH::list_off(lean_t) : 0];
				uint16_t *const my_pos = &sh.pos[my_s][0];
				const bool stored = (lean_t != 1); // the HALF_OPEN_3 list is only counted
				const uint32_t centre_event = static_cast<uint32_t>(center)
						| (added ? ((static_cast<uint32_t>(c0) << 9) | (static_cast<uint32_t>(c1) << 17)) : ((static_cast<uint32_t>(c0) << 13) | (static_cast<uint32_t>(c1) << 21)));
				uint32_t e0 = centre_event, e1 = centre_event; // the centre is the first event of both sides
				for (;;)
				{
					const uint32_t e = my_s ? e1 : e0;
					const int o = static_cast<int>((e >> shift_old) & 15u), nw = static_cast<int>((e >> shift_new) & 15u);
					const int cl = static_cast<int>(e & 511u);
					const bool take = (lean_t == o), put = (lean_t == nw);
					const int idx = my_pos[cl];
					const int last = my_items[max(cnt - 1, 0)];
					const int where = take ? idx : cnt, what = take ? last : cl;
					if ((take || put) && stored)
					{
						my_items[where] = static_cast<uint16_t>(what);
						if (put || last != cl)
							my_pos[what] = static_cast<uint16_t>(where);
					}
					cnt += (put ? 1 : 0) - (take ? 1 : 0);
					__builtin_amdgcn_wave_barrier();
					if ((changed0 | changed1) == 0ull)
						break;
					e0 = 0u;
					e1 = 0u;
					if (changed0 != 0ull)
					{
						const int src = __builtin_amdgcn_readfirstlane(__ffsll(static_cast<long long>(changed0)) - 1);
						changed0 &= changed0 - 1;
						e0 = __builtin_amdgcn_readlane(ev, src);
					}
					if (changed1 != 0ull)
					{
						const int src = __builtin_amdgcn_readfirstlane(__ffsll(static_cast<long long>(changed1)) - 1);
						changed1 &= changed1 - 1;
						e1 = __builtin_amdgcn_readlane(ev, src);
					}
				}
			}
			else
#endif
			{
			const int my_t = (lane < 20) ? lane - 10 * my_s : -1;
			auto edit = [&](int cell0, int o0, int n0, int cell1, int o1, int n1)
			{
				const int cl = my_s ? cell1 : cell0;
				const int o = my_s ? o1 : o0, nw = my_s ? n1 : n0;
				const bool take = (my_t == o) && (o != 0);
				const bool put = (my_t == nw) && (nw != 0);
				const bool stored = (my_t != 1); // the HALF_OPEN_3 list is only counted (nothing on the path reads its entries)
				int idx = 0, last = 0;
				if (take && stored)
				{
					idx = sh.pos[my_s][cl];
					last = static_cast<int>(list_get(sh, my_s, my_t, cnt - 1));
				}
				if (take || put)
				{
					if (stored)
					{
						list_set(sh, my_s, my_t, take ? idx : cnt, static_cast<uint32_t>(take ? last : cl));
						if (put || last != cl)
							sh.pos[my_s][take ? last : cl] = static_cast<uint16_t>(take ? idx : cnt);
					}
					cnt += put ? 1 : -1;
				}
				__builtin_amdgcn_wave_barrier();
			};
			if (added)
				edit(center, c0, 0, center, c1, 0);
			else
				edit(center, 0, c0, center, 0, c1);
			while ((changed0 | changed1) != 0)
			{
				int cell0 = 0, o0 = 0, n0 = 0, cell1 = 0, o1 = 0, n1 = 0;
				if (changed0 != 0)
				{
					const int src = __builtin_amdgcn_readfirstlane(__ffsll(static_cast<long long>(changed0)) - 1);
					changed0 &= changed0 - 1;
					cell0 = __builtin_amdgcn_readlane(cell, src);
					o0 = __builtin_amdgcn_readlane(old0, src);
					n0 = __builtin_amdgcn_readlane(new0, src);
				}
				if (changed1 != 0)
				{
					const int src = __builtin_amdgcn_readfirstlane(__ffsll(static_cast<long long>(changed1)) - 1);
					changed1 &= changed1 - 1;
					cell1 = __builtin_amdgcn_readlane(cell, src);
					o1 = __builtin_amdgcn_readlane(old1, src);
					n1 = __builtin_amdgcn_readlane(new1, src);
				}
				edit(cell0, o0, n0, cell1, o1, n1);
			}
			}
			if (lane < 20)
				sh.count[lane / 10][lane % 10] = static_cast<uint16_t>(cnt);
			wave_sync();
			AGX_PROF_MARK(sh, 12);
		}
		/* this lane's word of snapshot level `level` (wave-uniform base and level: a scalar base + the lane's offset) */
		template<class SH>
		__device__ __forceinline__ u64* snap_address(u64 *base, int level, int lane)
		{
			const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(base)));
			const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(base) >> 32));
			u64 *uniform = reinterpret_cast<u64*>((static_cast<uintptr_t>(hi) << 32) | lo);
			return uniform + (static_cast<size_t>(__builtin_amdgcn_readfirstlane(level)) * 64 + lane);
		}
		template<class SH>
		__device__ __forceinline__ void solver_place(SH &sh, const EngineDev &E, uint32_t move, bool add, int lane, bool prefetched = false,
				uint8_t pf_e = 0, bool have_snap = false, u64 pf_snap = 0)
		{ // PatternCalculator::addMove / undoMove (PatternCalculator.cpp:68-105); have_snap: pf_snap is snap_address(..)'s word for exactly this undo
			const int n = E.n;
			const int s = move & 3, r = (move >> 2) & 127, c = (move >> 9) & 127;
			u64 *const snap_base = sh.snap;
			const int to_move = sh.sign_to_move, stones = sh.depth;
			u64 *snap_slot = nullptr;
#if AGX_SNAPSHOT_UNDO
			snap_slot = snap_address<SH>(snap_base, add ? stones : stones - 1, lane);
			if (!add && !have_snap)
				pf_snap = *global_view(snap_slot);
#endif
			if (lane < 4)
			{
				int index, shift;
				line_of(n, r, c, lane, index, shift);
				if (add)
					sh.lines[index] |= static_cast<u64>(s) << (12 + shift);
				else
					sh.lines[index] &= ~(3ull << (12 + shift));
			}
			{ // wave-uniform values: every lane stores the same word to the same address (no exec masking around the stores)
				sh.board[r * n + c] = add ? static_cast<uint8_t>(s) : 0;
				const uint32_t row_bits = sh.legal[r];
				sh.legal[r] = add ? (row_bits & ~(1u << c)) : (row_bits | (1u << c));
			}
			wave_sync();
			solver_update_around(sh, E, r, c, add, lane, prefetched, pf_e, snap_slot, pf_snap);
			{
				sh.sign_to_move = 3 - to_move;
				sh.depth = stones + (add ? 1 : -1);
			}
			wave_sync();
		}

		/* getOpenThreePromotionMoves (DefensiveMoveTable.cpp:329-377): first matching shape wins */
		__device__ __forceinline__ uint32_t promotion_moves(uint32_t pattern)
		{ // The pattern is wave-uniform: twelve compare-and-select steps on the scalar unit with the shapes as immediates.  (As one shape per lane the
		  // three tables were loads from memory — two vector round trips and a scalar one per direction of every 3x3-fork test, nothing to hide them.)
			constexpr uint32_t PATTERNS[12] = { 320u, 4352u, 20480u, 80u, 16640u, 69632u, 272u, 4160u, 81920u, 320u, 4352u, 20480u };
			constexpr uint32_t MASKS[12] = { 65520u, 262080u, 1048320u, 16380u, 262080u, 1048320u, 16380u, 65520u, 1048320u, 16380u, 65520u, 262080u };
			constexpr uint32_t RESULTS[12] = { 196u, 392u, 784u, 82u, 328u, 656u, 74u, 148u, 592u, 70u, 140u, 280u };
			const uint32_t p = __builtin_amdgcn_readfirstlane(pattern);
			uint32_t result = 0u;
#pragma unroll
			for (int k = 11; k >= 0; k--) // (the last assignment is the first match in table order)
				result = ((p & MASKS[k]) == PATTERNS[k]) ? RESULTS[k] : result;
			return result;
		}
		/* RawPatternCalculator::isStraightFourAt on the line bit-boards: the 11-cell window around (r, c) in direction d (2 bits per cell, off-board 3),
		 * a cross stone assumed at the centre and, optionally, at window position `extra` (a stone that is on the reference's raw board but not in
		 * the pattern state: is_3x3_forbidden puts the fork stone there while it tries the promotion moves, PatternCalculator.cpp:222-231) */
		template<class SH>
		__device__ __forceinline__ bool straight_four_with(const SH &sh, int n, int r, int c, int d, int extra)
		{
			uint32_t line = normal_pattern(sh, n, r, c, d) | (1u << 10);
			if (extra >= 0)
				line |= 1u << (2 * extra);
			bool found = false;
#pragma unroll
			for (int k = 0; k < 7; k++, line >>= 2)
				found = found || ((line & 255u) == 85u);
			return found;
		}
		/*
		 * PatternCalculator::isForbidden + is_3x3_forbidden (PatternCalculator.hpp:161-177, PatternCalculator.cpp:213-244) for the cross player under
		 * renju.  The reference recurses through addMove / isForbidden / undoMove; here the recursion is an explicit machine whose CURRENT frame
		 * (cell, direction, promotion-move cursor, count of legal threes, promotion mask) lives in registers — wave-uniform values — and only a real
		 * recursion (a promotion cell that is itself a 3x3 fork: rare) goes through the LDS stack; a child that is decided by its threat type alone
		 * returns on the spot.  Stones are placed with the same wave-wide incremental update as everywhere else, so the threat lists are permuted
		 * exactly as in the reference.  Called by ALL lanes.
		 */
		template<class SH>
		__device__ __forceinline__ bool renju_is_forbidden(SH &sh, const EngineDev &E, int cell0, int lane)
		{
			const int n = E.n;
			if (sh.board[cell0] != 0)
				return false;
			{
				const int t = sh.threat[cell0][0];
				if (t == 9 || t == 6)
					return true;
				if (t != 3)
					return false;
			}
			int sp = 0, cell = cell0, dir = 0, count = 0, phase = 1; // phase 1: next direction, 2: next promotion move, 3: a child returned `ret`
			uint32_t candidates = 0; // promotion moves of (cell, dir) still to try: bit k = the empty cell k - 5 steps along the direction that would make a straight four
			bool ret = false;
			while (true)
			{
				const int r = cell / n, c = cell % n;
				if (phase == 1)
				{
					const uint32_t cross = *reinterpret_cast<const uint32_t*>(&sh.ptype[cell][0]); // the cell's four cross pattern types
					while (dir < 4 && ((cross >> (8 * dir)) & 255u) != 2u)
						dir++;
					if (dir >= 4)
					{ // every direction tried: forbidden iff two or more threes can become legal straight fours
						ret = count >= 2;
						if (sp == 0)
							return ret;
						sp--;
						cell = sh.fstack[sp][0];
						dir = sh.fstack[sp][1];
						candidates = static_cast<uint32_t>(sh.fstack[sp][2]);
						count = sh.fstack[sp][3];
						phase = 3;
						continue;
					}
					const uint32_t promo = promotion_moves(normal_pattern(sh, n, r, c, dir));
					{ // All promotion moves of the direction are looked at at once, one per lane (board cell, straight-four window: two LDS round trips for
					  // the direction instead of two per move); the stones that come and go between two of them leave board and lines as they were.
						const int k = fresh_lane(lane);
						bool makes_four = false;
						if (k < 11 && ((promo >> k) & 1u) != 0u)
						{
							const int rr = r + (k - 5) * row_step(dir), cc = c + (k - 5) * col_step(dir);
							makes_four = sh.board[rr * n + cc] == 0 && straight_four_with(sh, n, rr, cc, dir, 10 - k);
						}
						candidates = static_cast<uint32_t>(__ballot(makes_four));
					}
					phase = 2;
				}
				if (phase == 2)
				{
					if (candidates == 0u)
					{
						dir++;
						phase = 1;
						continue;
					}
					const int i = __ffs(static_cast<int>(candidates)) - 1 - 5;
					candidates &= candidates - 1u;
					const int found = (r + i * row_step(dir)) * n + (c + i * col_step(dir));
					if (sp + 1 >= 16)
					{
						sh.error = ERR_FRAMES;
						return true;
					}
					solver_place(sh, E, 1u | (static_cast<uint32_t>(r) << 2) | (static_cast<uint32_t>(c) << 9), true, lane);
					// isForbidden(found) with the fork stone on the board: decided by the threat type unless it is a 3x3 fork itself
					const int ct = sh.threat[found][0];
					if (ct != 3)
					{
						ret = (ct == 9 || ct == 6);
						phase = 3;
						continue;
					}
					sh.fstack[sp][0] = cell;
					sh.fstack[sp][1] = dir;
					sh.fstack[sp][2] = static_cast<int>(candidates);
					sh.fstack[sp][3] = count;
					wave_sync();
					sp++;
					cell = found;
					dir = 0;
					count = 0;
					phase = 1;
					continue;
				}
				// phase 3: the promotion move's cell returned `ret`; the fork stone comes off again
				solver_place(sh, E, 1u | (static_cast<uint32_t>(r) << 2) | (static_cast<uint32_t>(c) << 9), false, lane);
				if (!ret)
				{ // a legal straight four: this three counts, on to the next direction ("break")
					count++;
					dir++;
					phase = 1;
				}
				else
					phase = 2; // the next promotion move of the same direction
			}
		}
		/* NNInputFeatures::encode (NNInputFeatures.cpp:15-32,59-113) */
		template<class SH>
		__device__ __forceinline__ void solver_encode_features(const SH &sh, const EngineDev &E, uint32_t *out, int lane)
		{
			const int own = sh.sign_to_move;
			const uint32_t base = (1u << 3) | ((own == 1) ? (1u << 4) : (1u << 5));
			for (int cell = lane; cell < E.hw; cell += 64)
			{
				uint32_t r1 = 0, r2 = 0;
				for (int d = 0; d < 4; d++)
				{
					const int p1 = sh.ptype[cell][d], p2 = sh.ptype[cell][4 + d];
					r1 |= ((p1 == 2) ? (1u << d) : 0u) | ((p1 == 3) ? (16u << d) : 0u) | ((p1 >= 4) ? (1u << (4 + p1)) : 0u);
					r2 |= ((p2 == 2) ? (1u << d) : 0u) | ((p2 == 3) ? (16u << d) : 0u) | ((p2 >= 4) ? (1u << (4 + p2)) : 0u);
				}
				const uint32_t pat = (own == 1) ? ((r1 << 8) | (r2 << 20)) : ((r1 << 20) | (r2 << 8));
				const int v = sh.board[cell];
				const uint32_t stone = (v == 0) ? 1u : ((v == own) ? 2u : 4u);
				out[cell] = base | stone | pat;
			}
		}
		/* the renju part of encode (NNInputFeatures.cpp:105-112): bit 6 on every cell that is a foul for cross, probed in row-major
		 * order (the probes of 3x3 forks place and remove stones, i.e. permute the threat lists, exactly like the reference) */
		template<class SH>
		__device__ __forceinline__ void solver_encode_forbidden(SH &sh, const EngineDev &E, uint32_t *out, int lane)
		{
			if (E.rules != AGX_RENJU || sh.sign_to_move != 1)
				return;
			wave_sync();
			for (int base = 0; base < E.hw; base += 64)
			{ // candidates of 64 cells at a time (a probe restores the position, so the threat types seen here stay valid), probed in order
				const int mine = base + lane;
				const int t = (mine < E.hw) ? sh.threat[mine][0] : 0;
				u64 candidates = __ballot(mine < E.hw && sh.board[mine] == 0 && (t == 9 || t == 6 || t == 3));
				while (candidates != 0)
				{
					const int cell = base + __builtin_amdgcn_readfirstlane(__ffsll(static_cast<long long>(candidates)) - 1);
					candidates &= candidates - 1;
					if (renju_is_forbidden(sh, E, cell, lane))
					{
						if (lane == 0)
							out[cell] |= (1u << 6);
					}
				}
			}
			wave_sync();
		}

		/* ---------------- defensive-move lookup (DefensiveMoveTable.cpp:380-461) ---------------- */
		__constant__ const uint32_t STENCIL_BOX[7] = { 73u, 62u, 62u, 119u, 62u, 62u, 73u };   // MoveGenerator.cpp:1014-1023
		__constant__ const uint32_t STENCIL_STAR[7] = { 73u, 42u, 28u, 119u, 28u, 42u, 73u };  // MoveGenerator.cpp:1075-1084
		/* The 47 line shapes of getDefensiveMoves (DefensiveMoveTable.cpp:380-461), one per LANE: the shape as cross stones, where it starts in the
		 * 13-cell window and how long it is, its row in the defence table, the threat it answers (pattern type of the attacker's line: 6 FIVE,
		 * 4 OPEN_4, 5 DOUBLE_4, 3 HALF_OPEN_4, 2 OPEN_3) and the window position its table entry is stored for.  Lanes 47-63 answer nothing. */
		struct DefShape
		{
				uint32_t shape;
				uint8_t start, len, row, kind, ref, pad[3];
		};
		__constant__ const DefShape DEF_SHAPES[64] = {
			{ 85u, 2, 5, 0, 6, 2 }, { 277u, 3, 5, 1, 6, 3 }, { 325u, 4, 5, 2, 6, 4 }, { 337u, 5, 5, 3, 6, 5 }, { 340u, 6, 5, 4, 6, 6 },
			{ 84u, 2, 6, 5, 4, 2 }, { 276u, 3, 6, 6, 4, 3 }, { 324u, 4, 6, 7, 4, 4 }, { 336u, 5, 6, 8, 4, 5 },
			{ 4177u, 2, 7, 9, 5, 2 }, { 4369u, 3, 7, 10, 5, 3 }, { 4417u, 4, 7, 11, 5, 4 }, { 20549u, 2, 8, 12, 5, 2 }, { 20741u, 3, 8, 13, 5, 3 }, { 86037u, 2, 9, 14, 5, 2 },
			{ 21u, 3, 5, 0, 3, 2 }, { 69u, 4, 5, 0, 3, 2 }, { 81u, 5, 5, 0, 3, 2 }, { 84u, 6, 5, 0, 3, 2 },
			{ 21u, 2, 5, 1, 3, 3 }, { 261u, 4, 5, 1, 3, 3 }, { 273u, 5, 5, 1, 3, 3 }, { 276u, 6, 5, 1, 3, 3 },
			{ 69u, 2, 5, 2, 3, 4 }, { 261u, 3, 5, 2, 3, 4 }, { 321u, 5, 5, 2, 3, 4 }, { 324u, 6, 5, 2, 3, 4 },
			{ 81u, 2, 5, 3, 3, 5 }, { 273u, 3, 5, 3, 3, 5 }, { 321u, 4, 5, 3, 3, 5 }, { 336u, 6, 5, 3, 3, 5 },
			{ 84u, 2, 5, 4, 3, 6 }, { 276u, 3, 5, 4, 3, 6 }, { 324u, 4, 5, 4, 3, 6 }, { 336u, 5, 5, 4, 3, 6 },
			{ 20u, 3, 6, 5, 2, 2 }, { 68u, 4, 6, 5, 2, 2 }, { 80u, 5, 6, 5, 2, 2 },
			{ 20u, 2, 6, 6, 2, 3 }, { 260u, 4, 6, 6, 2, 3 }, { 272u, 5, 6, 6, 2, 3 },
			{ 68u, 2, 6, 7, 2, 4 }, { 260u, 3, 6, 7, 2, 4 }, { 320u, 5, 6, 7, 2, 4 },
			{ 80u, 2, 6, 8, 2, 5 }, { 272u, 3, 6, 8, 2, 5 }, { 320u, 4, 6, 8, 2, 5 } };
		/* All lanes hold the same arguments; lane i tests shape i against the window (one compare instead of a loop over up to 20 shapes run by
		 * every lane), a ballot finds the first match in table order, and the matching lane's defence-table entry is broadcast.  Same results as
		 * the loops of DefensiveMoveTable.cpp: first matching shape per threat kind; HALF_OPEN_4 under the caro rules ORs every match. */
		__device__ __forceinline__ uint32_t defensive_mask(const EngineDev &E, uint32_t pattern, int defender, int threat_to_defend)
		{
			if (threat_to_defend < 2 || threat_to_defend > 6)
				return 0u;
			const int lane = static_cast<int>(threadIdx.x);
			const int attacker = 3 - defender;
			const uint32_t mul = (attacker == 1) ? 1u : 2u;
			const DefShape c = DEF_SHAPES[lane];
			const int start = c.start, len = c.len;
			const uint32_t sub = (pattern >> (2 * start)) & ((1u << (2 * len)) - 1u);
			bool match = (static_cast<int>(c.kind) == threat_to_defend) && (sub == c.shape * mul);
			if (threat_to_defend == 3)
			{
				const bool allow_overline = (E.rules == AGX_FREESTYLE) || (E.rules == AGX_RENJU && attacker == 2) || (E.rules == AGX_CARO6);
				const bool allow_blocked = (E.rules != AGX_CARO5 && E.rules != AGX_CARO6);
				const int first = (pattern >> (2 * (start - 1))) & 3, last = (pattern >> (2 * (start + 5))) & 3;
				if (!allow_overline && (first == attacker || last == attacker))
					match = false;
				if (!allow_blocked && (first == defender && last == defender))
					match = false;
			}
			const uint32_t sides = ((pattern >> (2 * (start - 2))) & 15u) | (((pattern >> (2 * (start + len))) & 15u) << 4);
			uint32_t entry = E.t_defense[(static_cast<uint32_t>(c.row) * 256u + sides) * 2u + static_cast<uint32_t>(defender - 1)];
			const int sh = start - static_cast<int>(c.ref);
			entry = (sh >= 0) ? ((entry << sh) & 0xFFFFu) : (entry >> (-sh));
			const u64 m = __ballot(match);
			const uint32_t always = (threat_to_defend == 3) ? (1u << 6) : 0u; // HALF_OPEN_4: the answer starts from the centre cell
			if (m == 0ull)
				return always;
			if (threat_to_defend == 3 && (E.rules == AGX_CARO5 || E.rules == AGX_CARO6))
			{ // every matching shape contributes
				uint32_t acc = match ? entry : 0u;
				AGX_WAVE_REDUCE(acc, AGX_OP_OR);
				return always | static_cast<uint32_t>(__builtin_amdgcn_readlane(acc, 63));
			}
			const uint32_t found = static_cast<uint32_t>(__builtin_amdgcn_readlane(entry, __builtin_amdgcn_readfirstlane(__ffsll(static_cast<long long>(m)) - 1)));
			return found | ((threat_to_defend == 3 || threat_to_defend == 2) ? (1u << 6) : 0u);
		}

		/* ---------------- staged move generator, lane 0 only (MoveGenerator.cpp:159-1207, non-renju) ---------------- */
		struct SmallSet
		{ // StackVector<Location, N> semantics (patterns/common.hpp:154-245) with element i in LANE i of one register: the generator's sets hold at
		  // most a few dozen cells and every lane runs the generator with the same (wave-uniform) values, so membership is a ballot, an append a
		  // select and an element a v_readlane — no LDS round trips (kept in LDS every operation was one or two, in dependent chains: the
		  // defend-loss-in-2 / -4 / -6 stages of a node spent most of their time waiting for them)
				uint32_t v;
				int n, lane;
				__device__ explicit SmallSet(int ln) : v(0u), n(0), lane(ln) {}
				__device__ __forceinline__ int at(int i) const { return static_cast<int>(__builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(i))); }
				__device__ __forceinline__ bool contains(int x) const { return __ballot(lane < n && v == static_cast<uint32_t>(x)) != 0ull; }
				__device__ __forceinline__ void add(int x)
				{
					v = (lane == n) ? static_cast<uint32_t>(x) : v;
					n++;
				}
				__device__ __forceinline__ void remove_at(int i)
				{ // v[i] = v[--n]
					const uint32_t last = static_cast<uint32_t>(at(n - 1));
					v = (lane == i) ? last : v;
					n--;
				}
				__device__ __forceinline__ void remove(int x)
				{ // the first match
					const u64 m = __ballot(lane < n && v == static_cast<uint32_t>(x));
					if (m != 0ull)
						remove_at(__ffsll(static_cast<long long>(m)) - 1);
				}
		};

		/* The action stack: entries [0, ACT_LDS) live in LDS, deeper ones (a full board of candidate moves on top of a long
		 * forced line) in the per-game HBM spill area. */
		/* The action stack lives in LDS; only its tail (index >= ACT_LDS, rare) spills to HBM.  The two paths must stay two different
		 * instructions (hence the inline assembly): as plain C++ hipcc merges them into a FLAT load / store with a selected address, and a FLAT
		 * access to LDS costs a full vector-memory round trip on every action read (2300 cycles per move pick, measured). */
		template<class SH>
		__device__ __forceinline__ uint32_t act_get(const SH &sh, const uint32_t *spill, int i)
		{
			if (__builtin_expect(i < SH::ACT_LDS, 1))
				return sh.act[i];
			uint32_t v;
			asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(spill + i) : "memory");
			return v;
		}
		template<class SH>
		__device__ __forceinline__ void act_set(SH &sh, uint32_t *spill, int i, uint32_t v)
		{
			if (__builtin_expect(i < SH::ACT_LDS, 1))
				sh.act[i] = v;
			else
				asm volatile("global_store_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : : "v"(spill + i), "v"(v) : "memory");
		}
		template<class SH>
		__device__ __forceinline__ int act_find_move(const SH &sh, const uint32_t *spill, int begin, int end, uint32_t move, int lane)
		{
			lane = fresh_lane(lane);
			for (int base = begin; base < end; base += 64)
			{
				const int j = base + lane;
				const u64 m = __ballot(j < end && (act_get(sh, spill, j) & 0xFFFFu) == move);
				if (m != 0)
					return base + __ffsll(static_cast<long long>(m)) - 1;
			}
			return -1;
		}

		template<bool RENJU, class SH>
		struct MoveGen
		{
				SH &sh;
				const EngineDev &E;
				uint32_t *act;
				Frame &f;
				int n, own, opp, lane;
				int stack_offset, stack_max, board_depth;
				uint32_t cnt_w[2][5]; // list sizes, read once ([0] own, [1] opponent), two 16-bit counts per word as they lie in LDS: 10 registers, not 20

				__device__ __forceinline__ MoveGen(SH &s, const EngineDev &e, uint32_t *a, Frame &fr, int ln, int offset, int maximum) :
						sh(s), E(e), act(a), f(fr), n(e.n), own(s.sign_to_move), opp(3 - s.sign_to_move), lane(ln), stack_offset(offset), stack_max(maximum), board_depth(s.depth)
				{
const uint32_t *words = reinterpret_cast<const uint32_t*>(&s.count[0][0]);
#pragma unroll
for (int k = 0; k < 5; k++)
{
	cnt_w[0][k] = words[5 * (own - 1) + k];
	cnt_w[1][k] = words[5 * (opp - 1) + k];
}
				}
				__device__ __forceinline__ int item(int sign, int t, int k) const { return static_cast<int>(list_get(sh, sign - 1, t, k)); }
				__device__ __forceinline__ int count(int sign, int t) const
				{
					const uint32_t w = (sign == own) ? cnt_w[0][t >> 1] : cnt_w[1][t >> 1];
					return static_cast<int>((t & 1) ? (w >> 16) : (w & 0xFFFFu));
				}
				__device__ __forceinline__ const uint8_t* patterns(int sign, int cell) const { return sh.ptype[cell] + 4 * (sign - 1); }
				__device__ __forceinline__ int threat_at(int sign, int cell) const { return sh.threat[cell][sign - 1]; }
				__device__ __forceinline__ static int count_of(const uint8_t *g, int v) { return (g[0] == v) + (g[1] == v) + (g[2] == v) + (g[3] == v); }
				__device__ __forceinline__ static int direction_of(const uint8_t *g, int v)
				{
					for (int d = 0; d < 4; d++)
						if (g[d] == v)
							return d;
					return 0;
				}
				__device__ __forceinline__ bool has_any_four(int sign) const { return count(sign, 4) > 0 || count(sign, 5) > 0 || count(sign, 6) > 0 || count(sign, 7) > 0; }
				__device__ __forceinline__ bool fouls_possible_for(int sign) const { return RENJU && sign == 1; } // MoveGenerator.cpp:1155-1158
				__device__ __forceinline__ int available_fours(int sign) const
				{ // :1198-1207
					return count(sign, 7) + (fouls_possible_for(sign) ? 0 : count(sign, 6)) + count(sign, 5) + count(sign, 4);
				}
				__device__ __forceinline__ bool is_foul(int sign, int cell)
				{ // MoveGenerator::is_forbidden with its per-generate cache (:1159-1173)
					if (!fouls_possible_for(sign))
						return false;
					const int cached = sh.foul_count;
					{ // the cache, one entry per lane (its first match, as the scan in list order would find it): one LDS round trip, not one per entry
						const int k = fresh_lane(lane);
						const u64 hits = __ballot(k < cached && sh.foul_cell[k] == cell);
						if (hits != 0ull)
							return sh.foul_flag[__ffsll(static_cast<long long>(hits)) - 1] != 0;
					}
					#ifdef AGX_SOLVER_PROFILE
					const unsigned long long foul_t0 = clock64();
					#endif
					const bool r = renju_is_forbidden(sh, E, cell, lane);
					#ifdef AGX_SOLVER_PROFILE
					sh.dprof[18] += clock64() - foul_t0; // renju: uncached foul tests of the generator (each may place and remove stones)
					sh.dprof[19] += 1;
					#endif
					if (cached < 64)
					{
						sh.foul_cell[cached] = static_cast<uint16_t>(cell);
						sh.foul_flag[cached] = r ? 1 : 0;
						sh.foul_count = cached + 1;
					}
					wave_sync();
					return r;
				}
				__device__ __forceinline__ int copy_list(int sign, int t)
				{ // MoveGenerator::get_copy_of (:1174-1178)
					const int cnt = count(sign, t);
					for (int i = fresh_lane(lane); i < cnt; i += 64)
						sh.tmp_list[i] = item(sign, t, i);
					wave_sync();
					return cnt;
				}

				__device__ __forceinline__ void push(uint32_t move, uint32_t score, int num)
				{ // ActionList::add (ActionList.hpp:405-410)
					if (f.base + f.size + 1 >= E.act_cap)
					{
						sh.error = ERR_ACTION_STACK;
						return;
					}
					act_set(sh, act, f.base + f.size, move | (score << 16));
					f.size += num;
					stack_offset += num;
					stack_max = max(stack_max, stack_offset);
				}
				__device__ __forceinline__ uint32_t move_of(int cell) const { return static_cast<uint32_t>(own) | ((cell / n) << 2) | ((cell % n) << 9); }
				__device__ __forceinline__ void add_move(int cell, uint32_t score, bool override_duplicate)
				{ // MoveGenerator.cpp:228-249
					const int r = cell / n, c = cell % n;
					if ((sh.added[r] >> c) & 1)
					{
						if (override_duplicate)
						{
							const uint32_t m = move_of(cell);
							const int at = act_find_move(sh, act, f.base, f.base + f.size, m, lane);
							if (at >= 0)
								act_set(sh, act, at, m | (score << 16));
						}
					}
					else
					{
						push(move_of(cell), score, 1);
						sh.added[r] |= (1u << c);
					}
				}
				__device__ __forceinline__ void add_list(int sign, int t, uint32_t score, bool override_duplicate)
				{
					const int cnt = count(sign, t);
					if (override_duplicate)
					{
						for (int i = 0; i < cnt; i++)
							add_move(item(sign, t, i), score, true);
						return;
					}
					// the cells of one threat list are distinct, so the not-yet-added ones can be appended in list order by all lanes at once
					const int lane = fresh_lane(this->lane);
					for (int base = 0; base < cnt; base += 64)
					{
						const int k = base + lane;
						const int cell = (k < cnt) ? item(sign, t, k) : 0;
						const int r = cell / n, c = cell % n;
						const bool fresh = (k < cnt) && (((sh.added[r] >> c) & 1u) == 0u);
						const u64 m = __ballot(fresh);
						const int n_new = __popcll(m);
						if (n_new == 0)
							continue;
						if (f.base + f.size + n_new + 1 >= E.act_cap)
						{
							sh.error = ERR_ACTION_STACK;
							return;
						}
						if (fresh)
						{
							act_set(sh, act, f.base + f.size + lanes_below(m, lane), move_of(cell) | (score << 16));
							atomicOr(&sh.added[r], 1u << c);
						}
						f.size += n_new;
						stack_offset += n_new;
						stack_max = max(stack_max, stack_offset);
						wave_sync();
					}
				}
				__device__ __forceinline__ void defensive_moves(int defender, int cell, int dir, SmallSet &out) const
				{ // PatternCalculator::getDefensiveMoves (PatternCalculator.hpp:150-160)
					const int r = cell / n, c = cell % n;
					const uint32_t ext = extended_pattern(sh, n, r, c, dir);
					const int to_defend = patterns(3 - defender, cell)[dir];
					const uint32_t mask = defensive_mask(E, ext, defender, to_defend);
					out.n = 0;
					for (uint32_t m = mask & 0x1FFFu; m != 0u; m &= m - 1u)
					{ // the set bits in ascending order = the cells i = -6 .. 6 along the line
						const int i = __ffs(static_cast<int>(m)) - 1 - 6;
						out.add((r + i * row_step(dir)) * n + (c + i * col_step(dir)));
					}
				}
				__device__ __forceinline__ void get_defensive_moves(int cell, int dir, SmallSet &out)
				{ // MoveGenerator::get_defensive_moves (:263-308): defender is the side to move
					defensive_moves(own, cell, dir, out);
					if (fouls_possible_for(own))
					{
						int i = 0;
						while (i < out.n)
						{
							const int candidate = out.at(i);
							if (is_foul(own, candidate))
							{
								add_move(candidate, s_loss_in(1), true);
								out.remove_at(i);
							}
							else
								i++;
						}
					}
					else if (fouls_possible_for(opp))
					{
						if (patterns(opp, cell)[dir] == 4)
						{
							const int r = cell / n, c = cell % n;
							const uint32_t raw = extended_pattern(sh, n, r, c, dir);
							int type = 0;
							if ((raw & 65520u) == 1344u)
								type = -1;
							if ((raw & 4193280u) == 344064u)
								type = +1;
							if (type != 0)
							{
								const int far = (r + 4 * type * row_step(dir)) * n + (c + 4 * type * col_step(dir));
								if (is_foul(opp, far))
									out.add((r - type * row_step(dir)) * n + (c - type * col_step(dir)));
							}
						}
					}
				}
				__device__ __forceinline__ static void intersect(SmallSet &lhs, const SmallSet &rhs)
				{
					int i = 0;
					while (i < lhs.n)
					{
						if (rhs.contains(lhs.at(i)))
							i++;
						else
							lhs.remove_at(i);
					}
				}
				__device__ __forceinline__ static void intersect_init(SmallSet &dm, bool &initialized, const SmallSet &other)
				{ // DefensiveMoves::get_intersection_with (MoveGenerator.cpp:101-111)
					if (!initialized)
					{
						for (int i = 0; i < other.n; i++)
							dm.add(other.at(i));
						initialized = true;
					}
					else
						intersect(dm, other);
				}
				__device__ __forceinline__ uint32_t try_solve_own_fork_4x3(int cell)
				{ // :947-992
					if (fouls_possible_for(own))
						return s_unknown(15);
					const int dir = direction_of(patterns(own, cell), 3);
					SmallSet dm(lane);
					defensive_moves(opp, cell, dir, dm);
					dm.remove(cell);
					int best = 0;
					for (int i = 0; i < dm.n; i++)
					{
						const int tt = threat_at(opp, dm.at(i));
						if ((tt != 6 && tt != 9) || !fouls_possible_for(opp))
							best = max(best, tt);
					}
					switch (best)
					{
						case 4:
						case 5:
							return s_unknown(15);
						case 6:
						case 7:
							return s_loss_in(4);
						case 8:
						case 9:
							return s_loss_in(2);
						default:
							return s_win_in(5);
					}
				}
				__device__ __forceinline__ uint32_t add_own_4x3_forks()
				{ // :881-893
					uint32_t result = s_unknown(0);
					const int cnt = count(own, 5);
					for (int k = 0; k < cnt; k++)
					{
						const int cell = item(own, 5, k);
						const uint32_t solution = try_solve_own_fork_4x3(cell);
						add_move(cell, solution, true);
						if (s_proven(solution))
							result = max(result, solution);
					}
					return result;
				}
				__device__ __forceinline__ void add_own_half_open_fours()
				{ // :894-946
					int hidden = 0;
					if (fouls_possible_for(own))
					{ // a half-open four hidden inside a legal 3x3 fork
						const int cnt = copy_list(own, 3);
						for (int k = 0; k < cnt; k++)
						{
							const int cell = sh.tmp_list[k];
							if (count_of(patterns(own, cell), 3) > 0 && !is_foul(own, cell))
							{
								add_move(cell, s_unknown(14), false);
								hidden++;
							}
						}
					}
					add_list(own, 4, s_unknown(14), false);
					if (hidden + count(own, 4) > 0)
						f.has_initiative = 1;
				}
				/* 7x7 stencil (vertically and horizontally symmetric) OR-ed around every set bit of `occupied` rows: row R of the result
				 * only depends on rows R-3..R+3, so each lane builds one row (MoveGenerator.cpp:1011-1126 does it stone by stone). */
				__device__ __forceinline__ uint32_t stencil_row(const uint32_t *stencil, int which_sign) const
				{
					uint32_t m = 0;
					const int lane = fresh_lane(this->lane);
					if (lane < n)
						for (int dr = -3; dr <= 3; dr++)
						{
							const int rr = lane + dr;
							if (rr < 0 || rr >= n)
								continue;
							uint32_t occ = 0;
							if (which_sign == 0)
								occ = (~sh.legal[rr]) & ((1u << n) - 1u);
							else
								for (int c = 0; c < n; c++)
									if (sh.board[rr * n + c] == which_sign)
										occ |= (1u << c);
							const uint32_t p = stencil[3 - dr];
							for (int j = 0; j < 7; j++)
								if ((p >> j) & 1)
									m |= (j >= 3) ? (occ << (j - 3)) : (occ >> (3 - j));
						}
					return m;
				}
				__device__ __forceinline__ void create_remaining_moves(const uint32_t *mask, uint32_t score)
				{ // :1127-1137 — row-major append; one row per lane, offsets by a wave prefix sum
					const int lane = fresh_lane(this->lane);
					uint32_t bits = (lane < n) ? (mask[lane] & (~sh.added[lane])) : 0u;
					const int mine = __popc(bits);
					int offset = static_cast<int>(wave_scan32_add(static_cast<uint32_t>(mine)));
					const int total = __builtin_amdgcn_readlane(offset, 31);
					offset -= mine;
					if (f.base + f.size + total + 1 >= E.act_cap)
					{
						sh.error = ERR_ACTION_STACK;
						return;
					}
					int at = f.base + f.size + offset;
					while (bits != 0)
					{
						const int c = __ffs(static_cast<int>(bits)) - 1;
						bits &= bits - 1;
						act_set(sh, act, at++, (static_cast<uint32_t>(own) | (lane << 2) | (c << 9)) | (score << 16));
					}
					if (lane < n)
						sh.added[lane] |= mask[lane];
					f.size += total;
					stack_offset += total;
					stack_max = max(stack_max, stack_offset);
				}

				// each stage returns true when the cascade must continue; `result` receives the static score
				__device__ __forceinline__ bool try_win_in_1(uint32_t &result)
				{ // :355-371
					if (count(own, 8) > 0)
					{
						f.has_initiative = 1;
						add_list(own, 8, s_win_in(1), false);
						result = s_win_in(1);
						return false;
					}
					return true;
				}
				__device__ __forceinline__ bool try_draw_in_1(uint32_t &result)
				{ // :309-354
					f.baseline = static_cast<uint16_t>(s_draw_in(1));
					if (fouls_possible_for(own))
					{
						bool found = false;
						for (int cell = 0; cell < n * n; cell++)
							if (sh.board[cell] == 0)
							{
								const int t = threat_at(own, cell);
								if (t == 6 || t == 9)
									add_move(cell, s_loss_in(1), false);
								else if (t == 3 && is_foul(own, cell))
									add_move(cell, s_loss_in(1), false);
								else
								{
									add_move(cell, s_draw_in(1), false);
									found = true;
								}
							}
						result = found ? s_draw_in(1) : s_loss_in(1);
						return false;
					}
					create_remaining_moves(sh.legal, s_draw_in(1));
					result = s_draw_in(1);
					return false;
				}
				__device__ __forceinline__ bool defend_loss_in_2(uint32_t &result)
				{ // :372-463
					const int cnt = count(opp, 8);
					if (cnt == 0)
						return true;
					f.must_defend = 1;
					f.baseline = static_cast<uint16_t>(s_loss_in(2));
					SmallSet dm(lane), tmp(lane);
					bool initialized = false;
					for (int k = 0; k < cnt; k++)
					{
						const int cell = item(opp, 8, k);
						const int dir = direction_of(patterns(opp, cell), 6);
						get_defensive_moves(cell, dir, tmp);
						intersect_init(dm, initialized, tmp);
						if (dm.n == 0)
						{
							add_list(opp, 8, s_loss_in(2), false);
							result = s_loss_in(2);
							return false;
						}
					}
					uint32_t best = 0x0000u;
					for (int k = 0; k < dm.n; k++)
					{
						const int cell = dm.at(k);
						uint32_t response = s_unknown(0);
						switch (threat_at(own, cell))
						{
							case 3: // FORK_3x3
								if (fouls_possible_for(own))
								{
									if (count_of(patterns(own, cell), 4) > 0)
										response = s_win_in(3); // an open four hidden inside a legal 3x3 fork
								}
								else if (!has_any_four(opp))
									response = s_win_in(5);
								break;
							case 5: // FORK_4x3
							{
								const uint32_t solution = try_solve_own_fork_4x3(cell);
								response = s_proven(solution) ? solution : s_unknown(15);
								break;
							}
							case 6:
							case 7:
								response = s_win_in(3);
								break;
							default:
								if (count_of(patterns(own, cell), 3) > 0)
								{
									f.has_initiative = 1;
									response = s_unknown(14);
								}
								break;
						}
						if (s_win(response))
							f.has_initiative = 1;
						add_move(cell, response, false);
						best = max(best, response);
					}
					result = best;
					return false;
				}
				__device__ __forceinline__ bool try_win_in_3(uint32_t &result)
				{ // :464-555
					int threats = 0;
					if (fouls_possible_for(own))
					{ // an open four hidden inside a legal 3x3 fork
						const int cnt = copy_list(own, 3);
						for (int k = 0; k < cnt; k++)
						{
							const int cell = sh.tmp_list[k];
							if (count_of(patterns(own, cell), 4) > 0 && !is_foul(own, cell))
							{
								threats++;
								add_move(cell, s_win_in(3), false);
							}
						}
					}
					add_list(own, 7, s_win_in(3), false);
					threats += count(own, 7);
					if (count(own, 6) > 0 && !fouls_possible_for(own))
					{
						threats += count(own, 6);
						add_list(own, 6, s_win_in(3), false);
					}
					if (fouls_possible_for(opp))
					{ // a four whose only defence is a foul for the opponent (circle to move)
						const int cnt = copy_list(own, 4);
						for (int k = 0; k < cnt; k++)
						{
							const int cell = sh.tmp_list[k];
							const int dir = direction_of(patterns(own, cell), 3);
							bool winning = false;
							const int ot = threat_at(opp, cell);
							if (ot == 3)
								winning = (patterns(opp, cell)[dir] != 2) && is_foul(opp, cell);
							else if (ot == 6 || ot == 9)
								winning = true;
							if (winning)
							{
								SmallSet two(lane);
								defensive_moves(opp, cell, dir, two);
								const int original = (two.at(0) == cell) ? two.at(1) : two.at(0);
								add_move(original, s_win_in(3), false);
								result = s_win_in(3);
								return false;
							}
						}
					}
					if (threats > 0)
					{
						f.has_initiative = 1;
						result = s_win_in(3);
						return false;
					}
					return true;
				}
				__device__ __forceinline__ bool defend_loss_in_4_renju(bool any_four, uint32_t &result)
				{ // :621-677: no intersection of defences in renju, every defensive move of every threat is kept
					SmallSet tmp(lane);
					{
						const int cnt = copy_list(opp, 7);
						for (int k = 0; k < cnt; k++)
						{
							f.must_defend = 1;
							const int cell = sh.tmp_list[k];
							const int dir = direction_of(patterns(opp, cell), 4);
							get_defensive_moves(cell, dir, tmp);
							for (int i = 0; i < tmp.n; i++)
								add_move(tmp.at(i), s_unknown(0), false);
						}
					}
					if (fouls_possible_for(opp))
					{
						const int cnt = copy_list(opp, 3);
						for (int k = 0; k < cnt; k++)
						{
							const int cell = sh.tmp_list[k];
							if (count_of(patterns(opp, cell), 4) > 0 && !is_foul(opp, cell))
							{
								f.must_defend = 1;
								const int dir = direction_of(patterns(opp, cell), 4);
								get_defensive_moves(cell, dir, tmp);
								for (int i = 0; i < tmp.n; i++)
									add_move(tmp.at(i), s_unknown(0), false);
							}
						}
					}
					else
					{
						const int cnt = copy_list(opp, 6);
						for (int k = 0; k < cnt; k++)
						{
							f.must_defend = 1;
							const int cell = sh.tmp_list[k];
							for (int d = 0; d < 4; d++)
							{
								const int pt = patterns(opp, cell)[d];
								if (pt == 3 || pt == 4 || pt == 5)
								{
									get_defensive_moves(cell, d, tmp);
									for (int i = 0; i < tmp.n; i++)
										add_move(tmp.at(i), s_unknown(0), false);
								}
							}
						}
					}
					return finish_defend_loss_in_4(any_four, result);
				}
				__device__ __forceinline__ bool finish_defend_loss_in_4(bool any_four, uint32_t &result)
				{
					if (f.must_defend)
					{
						f.has_initiative = any_four ? 1 : 0;
						const uint32_t best = add_own_4x3_forks();
						add_own_half_open_fours();
						result = s_win(best) ? best : s_unknown(0);
						return false;
					}
					f.baseline = static_cast<uint16_t>(s_unknown(0));
					return true;
				}
				__device__ __forceinline__ bool defend_loss_in_4(uint32_t &result)
				{ // :556-689
					const bool any_four = has_any_four(own);
					f.baseline = static_cast<uint16_t>(s_loss_in(4));
					SmallSet dm(lane), tmp(lane), storage(lane);
					bool initialized = false;
					if (RENJU)
						return defend_loss_in_4_renju(any_four, result);
					const int n_open4 = count(opp, 7);
					for (int k = 0; k < n_open4; k++)
					{
						f.must_defend = 1;
						const int cell = item(opp, 7, k);
						const int dir = direction_of(patterns(opp, cell), 4);
						get_defensive_moves(cell, dir, tmp);
						intersect_init(dm, initialized, tmp);
						if (dm.n == 0 && !any_four)
						{
							add_list(opp, 7, s_loss_in(4), false);
							result = s_loss_in(4);
							return false;
						}
					}
					const int n_fork44 = count(opp, 6);
					for (int k = 0; k < n_fork44; k++)
					{
						f.must_defend = 1;
						const int cell = item(opp, 6, k);
						const uint8_t *group = patterns(opp, cell);
						for (int d = 0; d < 4; d++)
							if (group[d] == 4 || group[d] == 5)
							{
								get_defensive_moves(cell, d, tmp);
								intersect_init(dm, initialized, tmp);
							}
						if (count_of(group, 3) > 0)
						{
							storage.n = 0;
							for (int d = 0; d < 4; d++)
								if (group[d] == 3)
								{
									get_defensive_moves(cell, d, tmp);
									for (int i = 0; i < tmp.n; i++)
										if (!storage.contains(tmp.at(i)))
											storage.add(tmp.at(i));
								}
							intersect_init(dm, initialized, storage);
						}
						if (dm.n == 0 && !any_four)
						{
							add_list(opp, 6, s_loss_in(4), false);
							result = s_loss_in(4);
							return false;
						}
					}
					for (int i = 0; i < dm.n; i++)
						add_move(dm.at(i), s_unknown(0), false);
					if (f.must_defend)
					{
						f.has_initiative = any_four ? 1 : 0;
						const uint32_t best = add_own_4x3_forks();
						add_own_half_open_fours();
						result = s_win(best) ? best : s_unknown(0);
						return false;
					}
					f.baseline = static_cast<uint16_t>(s_unknown(0));
					return true;
				}
				__device__ __forceinline__ bool try_win_in_5(uint32_t &result)
				{ // :690-720
					uint32_t best = add_own_4x3_forks();
					if (!fouls_possible_for(own) && available_fours(opp) == 0 && count(own, 3) > 0)
					{
						add_list(own, 3, s_win_in(5), false);
						best = max(best, s_win_in(5));
					}
					if (s_win(best))
					{
						f.has_initiative = 1;
						result = best;
						return false;
					}
					return true;
				}
				__device__ __forceinline__ bool defend_loss_in_6(uint32_t &result)
				{ // :721-816
					if (available_fours(own) > 0)
						return true;
					const int n43 = count(opp, 5), n33 = count(opp, 3);
					if (n43 > 0 || n33 > 0)
					{
						f.must_defend = 1;
						f.baseline = static_cast<uint16_t>(s_loss_in(6));
					}
					SmallSet tmp(lane), half4(lane);
					for (int k = 0; k < n43; k++)
					{
						const int cell = item(opp, 5, k);
						const uint8_t *group = patterns(opp, cell);
						for (int d = 0; d < 4; d++)
							if (group[d] == 2)
							{
								get_defensive_moves(cell, d, tmp);
								for (int i = 0; i < tmp.n; i++)
									add_move(tmp.at(i), s_unknown(0), false);
							}
						const int dir = direction_of(group, 3);
						get_defensive_moves(cell, dir, half4);
						for (int i = 0; i < half4.n; i++)
							add_move(half4.at(i), s_unknown(0), false);
						for (int i = 0; i < half4.n; i++)
						{ // the 4 x 9 (direction, offset) candidates around a defensive move: one per lane, added in (d, j) order
							const int hr = half4.at(i) / n, hc = half4.at(i) % n;
							int l = -1;
							bool wanted = false;
							const int lane = fresh_lane(this->lane);
							if (lane < 36)
							{
								const int d = lane / 9, j = lane % 9 - 4;
								const uint32_t reduced = (extended_pattern(sh, n, hr, hc, d) >> 4) & 0x3FFFFu;
								if (((reduced >> (2 * (j + 4))) & 3u) == 0u)
								{
									const int rr = hr + j * row_step(d), cc = hc + j * col_step(d);
									l = rr * n + cc;
									wanted = patterns(own, l)[d] > 0 || ((E.t_ho3[narrow(normal_pattern(sh, n, rr, cc, d))] >> (own - 1)) & 1);
								}
							}
							u64 m = __ballot(wanted);
							while (m != 0)
							{
								const int src = __builtin_amdgcn_readfirstlane(__ffsll(static_cast<long long>(m)) - 1);
								m &= m - 1;
								add_move(__builtin_amdgcn_readlane(l, src), s_unknown(0), false);
							}
						}
					}
					for (int k = 0; k < n33; k++)
					{
						const int cell = item(opp, 3, k);
						const uint8_t *group = patterns(opp, cell);
						for (int d = 0; d < 4; d++)
							if (group[d] == 2)
							{
								get_defensive_moves(cell, d, tmp);
								for (int i = 0; i < tmp.n; i++)
									add_move(tmp.at(i), s_unknown(0), false);
							}
						add_list(own, 3, s_unknown(13), false);
						add_list(own, 2, s_unknown(1), false);
						if (fresh_lane(lane) < n)
							sh.row_mask[fresh_lane(lane)] = stencil_row(STENCIL_STAR, own);
						wave_sync();
						for (int base = 0; base < n * n; base += 64)
						{ // one cell per lane: inside the star mask, legal, not yet added, and a half-open three in some direction
							const int cell = base + fresh_lane(lane);
							bool wanted = false;
							if (cell < n * n)
							{
								const int r = cell / n, c = cell % n;
								if (((sh.row_mask[r] & sh.legal[r] & (~sh.added[r])) >> c) & 1u)
									for (int d = 0; d < 4; d++)
										wanted = wanted || ((E.t_ho3[narrow(normal_pattern(sh, n, r, c, d))] >> (own - 1)) & 1);
							}
							u64 m = __ballot(wanted);
							while (m != 0)
							{
								const int src = __builtin_amdgcn_readfirstlane(__ffsll(static_cast<long long>(m)) - 1);
								m &= m - 1;
								add_move(base + src, s_unknown(1), false);
							}
						}
					}
					if (f.must_defend)
					{
						add_own_half_open_fours();
						result = s_unknown(0);
						return false;
					}
					return true;
				}
				__device__ __forceinline__ void mark_neighborhood() const
				{ // :1011-1071 -> sh.row_mask
					uint32_t m = stencil_row(STENCIL_BOX, 0);
					const int lane = fresh_lane(this->lane);
					if (board_depth == 0 && lane == n / 2)
						m |= (1u << (n / 2));
					if (lane < n)
						sh.row_mask[lane] = m & sh.legal[lane];
				}

				/* MoveGenerator::generate (:159-223); mode 1 = THREATS, 2 = OPTIMAL */
				__device__ __forceinline__ void mark_forbidden_moves()
				{ // :993-1010
					add_list(own, 9, s_loss_in(1), true);
					add_list(own, 6, s_loss_in(1), true);
					const int cnt = copy_list(own, 3);
					for (int k = 0; k < cnt; k++)
						if (is_foul(1, sh.tmp_list[k]))
							add_move(sh.tmp_list[k], s_loss_in(1), true);
				}
				__device__ __forceinline__ uint32_t generate(int mode)
				{
					const int distance_to_draw = E.draw_after - board_depth;
					if (distance_to_draw <= 0)
						return s_make(1, 0);
					if (fresh_lane(lane) < n)
						sh.added[fresh_lane(lane)] = 0; // one row per lane (LDS runs a wave's accesses in issue order: the generator's later reads see it)
					sh.foul_count = 0;
					uint32_t result = s_unknown(0);
					bool go = true;
					AGX_PROF_BEGIN();
					if (go && distance_to_draw >= 1) go = try_win_in_1(result);
					if (go && distance_to_draw == 1) go = try_draw_in_1(result);
					AGX_PROF_MARK(sh, 0);
					if (go && distance_to_draw >= 2) go = defend_loss_in_2(result);
					AGX_PROF_MARK(sh, 1);
					if (go && distance_to_draw >= 3) go = try_win_in_3(result);
					AGX_PROF_MARK(sh, 2);
					if (go && distance_to_draw >= 4) go = defend_loss_in_4(result);
					AGX_PROF_MARK(sh, 3);
					if (go && distance_to_draw >= 5) go = try_win_in_5(result);
					AGX_PROF_MARK(sh, 4);
					if (go && distance_to_draw >= 6) go = defend_loss_in_6(result);
					AGX_PROF_MARK(sh, 5);
					if (go && distance_to_draw >= 3) add_own_half_open_fours();
					AGX_PROF_MARK(sh, 6);
					AGX_PROF_COUNT(sh, 8, 1);
					if (go && mode >= 2)
					{
						if (distance_to_draw >= 6)
						{
							add_list(opp, 3, s_unknown(3), false);
							add_list(opp, 2, s_unknown(2), false);
						}
						if (distance_to_draw >= 5)
						{
							add_list(own, 3, s_unknown(13), false);
							add_list(own, 2, s_unknown(1), false);
						}
						if (distance_to_draw >= 3)
							add_list(opp, 4, s_unknown(4), false);
						mark_neighborhood();
						create_remaining_moves(sh.row_mask, s_unknown(0));
					}
					AGX_PROF_MARK(sh, 7);
					if (fouls_possible_for(own))
						mark_forbidden_moves();
					AGX_PROF_MARK(sh, 9); // renju: mark_forbidden_moves
					f.fully_expanded = (f.must_defend || mode >= 2) ? 1 : 0;
					return result;
				}
		};

		/* ---------------- speculative table overlay ----------------
		 * The tasks of a game's batch share its transposition table and the reference solves them one after the other (Search.cpp:159-183).
		 * k_search_spec solves them in parallel, each against the table as it was BEFORE the batch: a task never writes the table, it
		 * copies every bucket it touches into its overlay (first-touch content + own version) and works on the copy.  When the whole batch
		 * is solved the tasks are committed in batch order: a task whose first-touch copies still equal the table's buckets has seen exactly
		 * what it would have seen in its turn, so its version of the written buckets IS the serial result; any other task is solved again,
		 * serially, on the then-current table (solve_spec_commit, engine.hip). */
		template<class SH>
		__device__ __forceinline__ int ov_find(const SH &sh, uint32_t bucket, int lane)
		{ // slot of `bucket` or -1: the keys are searched 64 per step, one per lane
			const int cnt = sh.ov_count;
			lane = fresh_lane(lane);
			for (int base = 0; base < cnt; base += 64)
			{
				const u64 m = __ballot(base + lane < cnt && sh.ov_keys[base + lane] == bucket);
				if (m != 0)
					return base + __ffsll(static_cast<long long>(m)) - 1;
			}
			return -1;
		}
		/* a new slot for `bucket` whose eight words lanes 0-7 and again lanes 8-15 hold in `word`; -1 (and ov_overflow) when the overlay is full */
		template<class SH>
		__device__ __forceinline__ int ov_create(SH &sh, uint32_t bucket, u64 word, int lane)
		{
			const int p = sh.ov_count;
			if (p >= OV_CAP)
			{
				sh.ov_overflow = 1;
				return -1;
			}
			if (lane < 16)
				global_view(sh.ov_data)[p * 16 + lane] = word; // lanes 0-7: first-touch copy, lanes 8-15 (same words): working copy
			if (lane == 0)
			{
				sh.ov_keys[p] = bucket;
				sh.ov_count = p + 1;
			}
			wave_sync();
			return p;
		}

		/* ---------------- transposition table (SharedHashTable.hpp:27-220), 4 x 16-byte entries per bucket ---------------- */
		__device__ __forceinline__ u64 tt_pack(int bound, int depth, uint32_t score, uint32_t move)
		{
			return static_cast<u64>(bound) | (static_cast<u64>(depth) << 8) | (static_cast<u64>(score) << 16) | (static_cast<u64>(move) << 32);
		}
		__device__ __forceinline__ u64 tt_seek(const u64 *tt, u64 bucket_mask, u64 lo, u64 hi)
		{
			const u64 *bucket = tt + 8 * (lo & bucket_mask);
			const u64 KEY = 0xFFFF000000000000ull;
			for (int i = 0; i < 4; i++)
				if (bucket[2 * i] == hi && (bucket[2 * i + 1] & KEY) == (lo & KEY))
					return bucket[2 * i + 1];
			return tt_pack(0, 0, s_unknown(0), 0);
		}
		__device__ __forceinline__ void tt_insert(u64 *tt, u64 bucket_mask, u64 lo, u64 hi, u64 value, int generation)
		{
			const u64 KEY = 0xFFFF000000000000ull;
			value &= ~(KEY | 0xFCull);
			value |= static_cast<u64>(generation) << 2;
			value |= (lo & KEY);
			u64 *bucket = tt + 8 * (lo & bucket_mask);
			const uint32_t score = static_cast<uint32_t>((value >> 16) & 65535u);
			if (s_proven(score) || (value & 3ull) == 3ull)
				for (int i = 0; i < 4; i++)
					if (bucket[2 * i] == hi && (bucket[2 * i + 1] & KEY) == (lo & KEY))
					{
						bucket[2 * i] = hi;
						bucket[2 * i + 1] = value;
						return;
					}
			int idx = 0, best_worth = 0;
			for (int i = 0; i < 4; i++)
			{
				const u64 d = bucket[2 * i + 1];
				const int worth = static_cast<int>((d >> 8) & 255) - (generation - static_cast<int>((d >> 2) & 63));
				if (i == 0 || worth < best_worth)
				{
					best_worth = worth;
					idx = i;
				}
			}
			bucket[2 * idx] = hi;
			bucket[2 * idx + 1] = value;
		}

		/* The same two operations with the bucket spread over lanes (AGX_TT_LANES): lane i < 8 holds word i of the bucket — the way the prefetch
		 * delivers it.  The forms above are run by all 64 lanes redundantly and walk the four entries one after the other: up to eight DEPENDENT
		 * round trips (LDS for the prefetched bucket, L2 / HBM for an insert) for what is one key compare per lane and a ballot. */
#ifndef AGX_TT_LANES
#define AGX_TT_LANES 1 /* (search launch 3.95 -> 3.92 ms; ScratchSize 396 -> 340 B/lane) */
#endif
		__device__ __forceinline__ u64 wave_read64(u64 v, int src_lane)
		{
			const int l = __builtin_amdgcn_readfirstlane(src_lane);
			const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(v)), l));
			const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(v >> 32)), l));
			return static_cast<u64>(lo) | (static_cast<u64>(hi) << 32);
		}
		/* index (0..3) of the FIRST entry whose key matches (SharedHashTable.hpp:150-165: hi word equal, top 16 bits of the data word equal to
		 * the top 16 bits of lo), or -1 */
		__device__ __forceinline__ int tt_match_lanes(u64 word, u64 lo, u64 hi, int lane)
		{
			const u64 KEY = 0xFFFF000000000000ull;
			const bool mine = (lane < 8) && (((lane & 1) == 0) ? (word == hi) : ((word & KEY) == (lo & KEY)));
			const uint32_t m = static_cast<uint32_t>(__ballot(mine));
			const uint32_t both = m & (m >> 1) & 0x55u; // bit 2k: entry k's key word (lane 2k) and tag (lane 2k + 1) both match
			return (both != 0u) ? ((__ffs(static_cast<int>(both)) - 1) >> 1) : -1;
		}
		__device__ __forceinline__ u64 tt_seek_lanes(u64 word, u64 lo, u64 hi, int lane)
		{
			const int k = tt_match_lanes(word, lo, hi, lane);
			return (k >= 0) ? wave_read64(word, 2 * k + 1) : tt_pack(0, 0, s_unknown(0), 0);
		}
		__device__ __forceinline__ void tt_insert_lanes(u64 *bucket, u64 lo, u64 hi, u64 value, int generation, int lane)
		{ // `bucket`: the eight words (wave-uniform pointer)
			const u64 KEY = 0xFFFF000000000000ull;
			value &= ~(KEY | 0xFCull);
			value |= static_cast<u64>(generation) << 2;
			value |= (lo & KEY);
			u64 word = 0;
			if (lane < 8)
				word = global_view(bucket)[lane];
			const uint32_t score = static_cast<uint32_t>((value >> 16) & 65535u);
			int idx = -1;
			if (s_proven(score) || (value & 3ull) == 3ull)
				idx = tt_match_lanes(word, lo, hi, lane);
			if (idx < 0)
			{ // the first minimum of depth - age over the four entries (their data words sit in the odd lanes)
				const int worth = static_cast<int>((word >> 8) & 255) - (generation - static_cast<int>((word >> 2) & 63));
				const int w0 = __builtin_amdgcn_readlane(worth, 1), w1 = __builtin_amdgcn_readlane(worth, 3), w2 = __builtin_amdgcn_readlane(worth, 5),
						w3 = __builtin_amdgcn_readlane(worth, 7);
				idx = 0;
				int best = w0;
				if (w1 < best) { best = w1; idx = 1; }
				if (w2 < best) { best = w2; idx = 2; }
				if (w3 < best) { best = w3; idx = 3; }
			}
			if (lane < 2)
				global_view(bucket)[2 * idx + lane] = (lane == 0) ? hi : value;
		}

		/* AlphaBetaSearch::evaluate (AlphaBetaSearch.cpp:345-365) */
		template<class SH>
		__device__ __forceinline__ uint32_t solver_evaluate(const SH &sh)
		{ // one (side, threat type) per lane: lane 10 s + t weighs count[s][t] (types 0, 1 and 9 weigh nothing), a wave sum adds the 20 products — two's
		  // complement, so the unsigned reduction IS the signed sum — instead of 14 LDS reads and 14 constant loads run by every lane
			const int lane = static_cast<int>(threadIdx.x);
			const int own = sh.sign_to_move - 1;
			int product = 0;
			if (lane < 20)
			{
				const int side = (lane >= 10) ? 1 : 0, t = lane - 10 * side;
				// AlphaBetaSearch.cpp:356-357: own { 0, 0, 19, 49, 76, 170, 33, 159, 252, 0 }, opponent { 0, 0, -1, -50, -45, -135, -14, -154, -496, 0 } — types 2 .. 8
				// as 8-bit fields / negated as 9-bit fields of one 64-bit constant each (no loads from memory in front of the wave sum)
				constexpr u64 OWN_WEIGHTS = 19ull | (49ull << 8) | (76ull << 16) | (170ull << 24) | (33ull << 32) | (159ull << 40) | (252ull << 48);
				constexpr u64 OPP_WEIGHTS = 1ull | (50ull << 9) | (45ull << 18) | (135ull << 27) | (14ull << 36) | (154ull << 45) | (496ull << 54);
				const bool mine = (side == own), weighs = (t >= 2 && t <= 8);
				const int field = weighs ? t - 2 : 0;
				const int magnitude = static_cast<int>(((mine ? OWN_WEIGHTS : OPP_WEIGHTS) >> ((mine ? 8 : 9) * field)) & (mine ? 255ull : 511ull));
				const int weight = weighs ? (mine ? magnitude : -magnitude) : 0;
				product = weight * static_cast<int>(sh.count[side][t]);
			}
			const int result = 12 + static_cast<int>(wave_reduce_add(static_cast<uint32_t>(product)));
			return s_unknown(max(-1000, min(1000, result)));
		}

		/*
		 * AlphaBetaSearch::recursive_solve (AlphaBetaSearch.cpp:185-339) as an explicit frame machine run by lane 0.
		 * Runs until a stone must be placed/removed (returns CMD_ADD / CMD_UNDO with sh.cmd_move) or the root returns (CMD_DONE).
		 * phase: 0 = enter frame `level`, 1 = resume frame `level` after its child returned sh.pending_value.
		 */
		template<bool RENJU, class SH>
		__device__ __forceinline__ int solver_run(SH &sh, const EngineDev &E, uint32_t *act, u64 *tt, int generation, int lane, u64 &pf_word,
				uint8_t &pf_pattern, int &pf_pattern_tag, u64 &pf_snap)
		{ // executed by ALL lanes with identical (wave-uniform) state: stores are same-address / same-value, scans are lane-parallel.
		  // The scalars of the machine and the current frame are held in registers and written back to LDS only when the machine yields.
			AGX_PROF_BEGIN();
			const u64 zseed = E.zobrist_seed;
			const int n = E.n;
			int phase = sh.phase, level = sh.level;
			int node_counter = sh.node_counter, stack_offset = sh.stack_offset, stack_max = sh.stack_max, error = sh.error;
			u64 hash_lo = sh.hash_lo, hash_hi = sh.hash_hi;
			uint32_t value = static_cast<uint32_t>(sh.pending_value);
			Frame f = frame_get(sh, level);
			AGX_PROF_MARK(sh, 15); // resume: machine state back into registers
			auto yield = [&](int cmd, int move)
			{
				sh.phase = phase;
				sh.level = level;
				sh.node_counter = node_counter;
				sh.stack_offset = stack_offset;
				sh.stack_max = stack_max;
				sh.error = error;
				sh.hash_lo = hash_lo;
				sh.hash_hi = hash_hi;
				sh.pending_value = static_cast<int>(value);
				sh.cmd_move = move;
				return cmd;
			};
			uint32_t cur_action = 0; // the action list's entry at f.i as last written / picked (phase 3 reads it)
			while (true)
			{
				bool returning = false;
				AGX_PROF_COUNT(sh, 21, 1);
#ifdef AGX_SOLVER_PROFILE
				agx_pt_ = clock64();
				const unsigned long long p0 = wall_clock64();
				unsigned long long p1 = p0, p2 = p0;
#endif
				if (phase == 0)
				{ // ---- enter ----
					f.best_move = 0;
					u64 entry;
					const bool have_pf = sh.pf_valid && sh.pf_lo == hash_lo;
					if (have_pf || sh.ov_on)
					{ // bucket fetched while the stone was being placed (lanes 0-7 still hold its words in a register)
						if (sh.ov_on)
						{ // speculative solve: the bucket comes out of / goes into the task's overlay
							const uint32_t bucket = static_cast<uint32_t>(hash_lo & E.tt_bucket_mask);
							int slot = have_pf ? sh.pf_slot : ov_find(sh, bucket, lane);
							if (!have_pf && lane < 16)
								pf_word = (slot >= 0) ? global_view(sh.ov_data)[slot * 16 + 8 + (lane & 7)] : global_view(tt)[8 * static_cast<u64>(bucket) + (lane & 7)];
							if (slot < 0)
								slot = ov_create(sh, bucket, pf_word, lane);
							f.ov_slot = static_cast<uint32_t>(slot);
						}
#if AGX_TT_LANES
						entry = tt_seek_lanes(pf_word, hash_lo, hash_hi, lane);
					}
					else
					{
						u64 word = 0;
						if (lane < 8)
							word = tt[8 * (hash_lo & E.tt_bucket_mask) + lane];
						entry = tt_seek_lanes(word, hash_lo, hash_hi, lane);
					}
#else
						if (lane < 8)
							sh.pf_bucket[lane] = pf_word;
						wave_sync();
						const u64 KEY = 0xFFFF000000000000ull;
						entry = tt_pack(0, 0, s_unknown(0), 0);
						for (int k = 3; k >= 0; k--)
							if (sh.pf_bucket[2 * k] == hash_hi && (sh.pf_bucket[2 * k + 1] & KEY) == (hash_lo & KEY))
								entry = sh.pf_bucket[2 * k + 1];
					}
					else
						entry = tt_seek(tt, E.tt_bucket_mask, hash_lo, hash_hi);
#endif
					sh.pf_valid = 0;
					bool early = false;
					if (sh.ov_on && sh.ov_overflow)
					{ // the overlay is full: the solve is abandoned (every frame returns at once) and repeated serially, straight on the table
						error = ERR_OVERLAY;
						value = s_unknown(0);
						early = true;
					}
					else if ((entry & 3ull) != 0ull)
					{
						f.best_move = static_cast<uint16_t>((entry >> 32) & 65535u);
						if (level != 0)
						{
							const uint32_t tt_s = static_cast<uint32_t>((entry >> 16) & 65535u);
							const int b = static_cast<int>(entry & 3ull);
							if (s_proven(tt_s))
							{
								value = tt_s;
								early = true;
							}
							else if (static_cast<int>((entry >> 8) & 255) >= f.depth_remaining && (b == 3 || (b == 1 && tt_s >= f.beta) || (b == 2 && tt_s <= f.alpha)))
							{
								value = tt_s;
								early = true;
							}
						}
					}
#ifdef AGX_SOLVER_PROFILE
					p1 = wall_clock64();
					sh.prof[0] += p1 - p0; // table seek
#endif
					if (!early)
					{
						node_counter++;
						if (E.solve_time_ticks != 0ull && wall_clock64() >= sh.time_deadline)
							node_counter |= NODE_TIME_OVER; // AlphaBetaSearch.cpp:110-111,277: (getTime() - start_time) >= max_time
						if (f.size == 0)
						{
							MoveGen<RENJU, SH> gen(sh, E, act, f, lane, stack_offset, stack_max);
							const uint32_t static_score = gen.generate(level == 0 ? 2 : 1);
							stack_offset = gen.stack_offset;
							stack_max = gen.stack_max;
							if (sh.error != 0)
								error = sh.error;
							if (s_proven(static_score))
							{
								value = static_score;
								early = true;
							}
						}
					}
#ifdef AGX_SOLVER_PROFILE
					p2 = wall_clock64();
					sh.prof[1] += p2 - p1; // move generation
#endif
					if (!early && f.depth_remaining <= 0)
					{
						value = solver_evaluate(sh);
						early = true;
					}
					if (early)
						returning = true;
					else
					{
						f.original_alpha = f.alpha;
						f.best_score = 0x0000u;
						f.i = 0;
						phase = 2; // iterate
					}
				}
				else if (phase == 1)
				{ // ---- child returned ----
					const uint32_t mv = f.move; // == the move of the entry at f.i (set when the machine descended)
					cur_action = mv | (s_invert_up(value) << 16);
					act_set(sh, act, f.base + f.i, cur_action);
					const int cell = ((mv >> 2) & 127) * n + ((mv >> 9) & 127);
					{ // the key index is wave-uniform: in an SGPR the splitmix64 rounds run on the scalar unit
						const uint32_t zi = __builtin_amdgcn_readfirstlane(2 * (2 * cell + ((mv & 3) - 1)));
						hash_lo ^= zobrist_word(zseed, zi);
						hash_hi ^= zobrist_word(zseed, zi + 1);
					}
					phase = 3; // post-child bookkeeping
					AGX_PROF_MARK(sh, 20); // child returned: score write-back + hash
				}
#ifdef AGX_SOLVER_PROFILE
				const unsigned long long p3 = wall_clock64();
				agx_pt_ = clock64();
#endif
				if (!returning && phase == 2)
				{ // ---- pick the next action (:253-266) ----
					if (f.i >= f.size)
						phase = 4;
					else
					{
						uint32_t picked = 0;
						bool have_picked = false;
						const uint32_t bm = f.best_move;
						const bool tt_move_legal = ((bm & 3u) == static_cast<uint32_t>(sh.sign_to_move)) && sh.board[((bm >> 2) & 127) * n + ((bm >> 9) & 127)] == 0;
						if (f.i == 0 && tt_move_legal)
						{
							const int at = act_find_move(sh, act, f.base, f.base + f.size, bm, lane);
							if (at >= 0)
							{
								const uint32_t t = act_get(sh, act, f.base);
								act_set(sh, act, f.base, act_get(sh, act, at));
								act_set(sh, act, at, t);
							}
						}
						else if (f.size - f.i <= 64)
						{ // first maximum of the remaining actions, one per lane: the key is (score, lowest index), one DPP reduction picks the
						  // winner, and the two entries that trade places come out of the lanes' registers (one LDS round trip, not four)
							const int j = f.i + fresh_lane(lane);
							const uint32_t mine = (j < f.size) ? act_get(sh, act, f.base + j) : 0u;
							uint32_t key = (j < f.size) ? (((mine >> 16) << 16) | static_cast<uint32_t>(0xFFFF - j)) : 0u;
							key = wave_reduce_umax(key);
							const int idx = 0xFFFF - static_cast<int>(key & 0xFFFFu);
							picked = __builtin_amdgcn_readlane(mine, __builtin_amdgcn_readfirstlane(idx - f.i));
							if (idx != f.i)
							{
								act_set(sh, act, f.base + f.i, picked);
								act_set(sh, act, f.base + idx, __builtin_amdgcn_readlane(mine, 0));
							}
							have_picked = true;
						}
						else
						{ // more than 64 actions left: every lane folds its actions into one key
							uint32_t key = 0;
							for (int j = f.i + lane; j < f.size; j += 64)
								key = max(key, ((act_get(sh, act, f.base + j) >> 16) << 16) | static_cast<uint32_t>(0xFFFF - j));
							key = wave_reduce_umax(key);
							const int idx = 0xFFFF - static_cast<int>(key & 0xFFFFu);
							if (idx != f.i)
							{
								const uint32_t t = act_get(sh, act, f.base + f.i);
								act_set(sh, act, f.base + f.i, act_get(sh, act, f.base + idx));
								act_set(sh, act, f.base + idx, t);
							}
						}
						AGX_PROF_MARK(sh, 16); // pick: table move or first maximum, swap to the front
						AGX_PROF_COUNT(sh, 22, 1);
						const uint32_t a = have_picked ? picked : act_get(sh, act, f.base + f.i);
						cur_action = a;
						if (s_unproven(a >> 16) && node_counter < E.tss_max_nodes)
						{ // descend (:268-298)
							if (level + 1 >= MAX_FRAMES)
							{
								error = ERR_FRAMES;
								phase = 3;
							}
							else
							{
								const uint32_t mv = a & 0xFFFFu;
								pf_pattern = pattern_prefetch(sh, E, n, mv, true, lane); // consumed by the solver_place this yield asks for
								pf_pattern_tag = static_cast<int>(mv) | 0x10000;
								const int cell = ((mv >> 2) & 127) * n + ((mv >> 9) & 127);
								const uint32_t zi = __builtin_amdgcn_readfirstlane(2 * (2 * cell + ((mv & 3) - 1)));
								hash_lo ^= zobrist_word(zseed, zi);
								hash_hi ^= zobrist_word(zseed, zi + 1);
								f.move = static_cast<uint16_t>(mv);
								frame_set(sh, level, f);
								Frame child;
								child.base = f.base + f.size; // == stack offset: lists are strictly nested
								child.size = 0;
								child.i = 0;
								child.depth_remaining = f.depth_remaining - 1;
								child.alpha = static_cast<uint16_t>(s_invert_down(f.beta));
								child.beta = static_cast<uint16_t>(s_invert_down(f.alpha));
								child.original_alpha = child.alpha;
								child.best_score = 0;
								child.best_move = 0;
								child.move = 0;
								child.baseline = static_cast<uint16_t>(s_unknown(0));
								child.must_defend = child.has_initiative = child.fully_expanded = child.pad = 0;
								frame_set(sh, level + 1, child);
								level++;
								phase = 0;
								// SharedHashTable::prefetch (AlphaBetaSearch.cpp:273): fetch the child's bucket now, it is consumed
								// when the child frame is entered after the stone has been placed
								if (sh.ov_on)
								{
									const uint32_t bucket = static_cast<uint32_t>(hash_lo & E.tt_bucket_mask);
									const int slot = ov_find(sh, bucket, lane);
									if (lane < 16)
										pf_word = (slot >= 0) ? global_view(sh.ov_data)[slot * 16 + 8 + (lane & 7)] : global_view(tt)[8 * static_cast<u64>(bucket) + (lane & 7)];
									sh.pf_slot = slot;
								}
								else if (lane < 8)
									pf_word = tt[8 * (hash_lo & E.tt_bucket_mask) + lane];
								sh.pf_lo = hash_lo;
								sh.pf_valid = 1;
								AGX_PROF_MARK(sh, 17); // descend: frames, hash, prefetch
#ifdef AGX_SOLVER_PROFILE
								sh.prof[2] += wall_clock64() - p3; // ordering + descend bookkeeping
#endif
								return yield(CMD_ADD, static_cast<int>(mv));
							}
						}
						else
							phase = 3;
					}
				}
				if (!returning && phase == 3)
				{ // ---- after the action has its score (:299-307): the entry at f.i, still in a register from the pick or the child's return ----
					const uint32_t a = cur_action;
					const uint32_t sc = a >> 16;
					f.best_score = static_cast<uint16_t>(max(static_cast<uint32_t>(f.best_score), sc));
					if (sc > f.alpha)
					{
						f.alpha = static_cast<uint16_t>(sc);
						f.best_move = static_cast<uint16_t>(a & 0xFFFFu);
					}
					if (sc >= f.beta || s_win(sc) || error != 0)
						phase = 4;
					else
					{
						f.i++;
						phase = 2;
#ifdef AGX_SOLVER_PROFILE
						sh.prof[2] += wall_clock64() - p3;
#endif
						continue;
					}
				}
#ifdef AGX_SOLVER_PROFILE
				const unsigned long long p4 = wall_clock64();
				sh.prof[2] += p4 - p3;
#endif
				if (!returning && phase == 4)
				{ // ---- finish the node (:308-338) ----
					uint32_t best = f.best_score;
					if (f.size == 0 || (s_loss(best) && !f.fully_expanded))
						best = solver_evaluate(sh);
					if (level > 0)
					{ // the stone of the parent's move comes off next: its pattern entries travel while the table is updated
						const uint32_t umv = frame_get(sh, level - 1).move;
#if AGX_SNAPSHOT_UNDO && AGX_SNAPSHOT_PREFETCH
						pf_snap = *global_view(snap_address<SH>(sh.snap, sh.depth - 1, lane)); // what this node's stone overwrote (solver_update_around)
						pf_pattern_tag = static_cast<int>(umv);
#elif !AGX_SNAPSHOT_UNDO
						pf_pattern = pattern_prefetch(sh, E, n, umv, false, lane);
						pf_pattern_tag = static_cast<int>(umv);
#endif
					}
					int bound;
					if (best <= f.original_alpha)
						bound = 2;
					else
						bound = (best >= f.beta) ? 1 : 3;
					if (sh.ov_on)
					{ // the node's bucket is in the overlay since the frame was entered: update the task's version of it
						if (error != ERR_OVERLAY)
						{
							const int slot = static_cast<int>(f.ov_slot);
#if AGX_TT_LANES
							tt_insert_lanes(sh.ov_data + slot * 16 + 8, hash_lo, hash_hi, tt_pack(bound, f.depth_remaining, best, f.best_move), generation, lane);
#else
							tt_insert(sh.ov_data + slot * 16 + 8, 0ull, hash_lo, hash_hi, tt_pack(bound, f.depth_remaining, best, f.best_move), generation);
#endif
							if (lane == 0)
								sh.ov_dirty[slot >> 5] |= 1u << (slot & 31);
						}
					}
					else
#if AGX_TT_LANES
						tt_insert_lanes(tt + 8 * (hash_lo & E.tt_bucket_mask), hash_lo, hash_hi, tt_pack(bound, f.depth_remaining, best, f.best_move), generation, lane);
#else
						tt_insert(tt, E.tt_bucket_mask, hash_lo, hash_hi, tt_pack(bound, f.depth_remaining, best, f.best_move), generation);
#endif
					value = best;
					returning = true;
#ifdef AGX_SOLVER_PROFILE
					sh.prof[3] += wall_clock64() - p4; // evaluate + table insert
#endif
				}
				if (returning)
				{
					if (level == 0)
					{
						sh.frames[0] = f;
						sh.result_score = static_cast<int>(value);
						return yield(CMD_DONE, 0);
					}
					stack_offset -= f.size; // ~ActionList (ActionList.hpp:344-347)
					level--;
					phase = 1;
					return yield(CMD_UNDO, frame_get(sh, level).move);
				}
			}
		}
	}
}

#endif
