/*
 * nn_forward.hip — whole-tower policy/value forward for one board per workgroup, gfx950 (CDNA4).
 *
 * Replaces `graph.predict` of the ResnetPV / ResnetPVQ graphs (src/networks/networks.cpp:71-93, :143-168; layers from
 * src/networks/blocks.cpp:32-38 input block, :45-55 residual block, :99-107 policy head, :108-118 value head, :119-127
 * action-values head)
 * after optimize(2) folded every BatchNormalization into the preceding layer (AGNetwork.cpp:136-160), plus
 * `ml::unpackInput` (AGNetwork.cpp:249-258) which expands the bit-packed feature word of each cell
 * (NNInputFeatures.cpp:59-113) to Cin = 32 channels of {0,1}.
 *
 * MI355X-first design (not how MinML runs it):
 *   - A 15x15 board with F = 128 channels in fp16 is 57.6 KB, so TWO activation planes of a board fit in the
 *     160 KB LDS of one CU.  One workgroup (8 waves, two per SIMD) therefore carries a board through the
 *     entire tower — conv5x5, every residual block, both heads — without a single activation byte touching
 *     HBM.  HBM traffic per position is the algorithmic minimum: 4*HW bytes in, 4*(HW+3) bytes out.
 *   - Each convolution is an implicit GEMM on the matrix cores: D[out-ch][position] += W[out-ch][k] * X[k][position]
 *     with v_mfma_f32_16x16x32_f16 (fp16 in, fp32 accumulate).  The WEIGHTS are the A operand and the
 *     activations the B operand, so an accumulator lane ends up with 4 consecutive output channels of one
 *     position = one 8-byte LDS store in NHWC order.
 *   - The board is stored with a row stride of S = cols + 1 positions and one spare row above and below; the
 *     spare column / rows are kept at zero, so a 3x3 tap is just a constant offset dy*S + dx in the flattened
 *     position index — no bounds tests in the MFMA loop.
 *   - LDS rows are XOR-swizzled at 16-byte granularity so that the 16 lanes of a ds_read_b128 group hit 16
 *     different bank quads.
 *   - Weights are pre-packed on the host in MFMA A-fragment order; each wave streams only the fragments of its
 *     own output channels from L2 with fully coalesced 16-byte loads, fetched a few k-steps ahead (no LDS round trip).
 *   - The grid is persistent (one workgroup per CU) and strides over the batch (a device-side slot list, so the search
 *     kernels hand positions over without a host round trip).
 *   - 20x20 boards do not fit two planes: the INPLACE variant computes every layer over its own input (all outputs stay in
 *     accumulators until every wave has passed a barrier) and parks the residual input in a per-workgroup global scratch.
 *   - Template flags keep the variants apart: <F, ROWS, COLS, INPLACE, QHEAD>; the pv kernels carry no q-head code.
 */
#include "agx_internal.hpp"

#include <vector>
#include <mutex>
#include <utility>
#include <cstring>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#ifndef AGX_NN_COLS_AHEAD
#define AGX_NN_COLS_AHEAD 4 // activation fragments in flight per wave in the column-tile k-loop
#endif
#ifndef AGX_NN_COLS_PAD
#define AGX_NN_COLS_PAD 1 // 16-byte units of padding per stored position on column-tile boards (Geometry::PAD): 1 or 2
#endif
#ifndef AGX_NN_ROW_GROUPS
#define AGX_NN_ROW_GROUPS 8 // channel groups of a workgroup's 8 waves on 15-column boards with 128 filters (Geometry::CG): 8 (x 1 position group) or 4 (x 2)
#endif
#ifndef AGX_NN_PAIR_BALANCE
#define AGX_NN_PAIR_BALANCE 1 // 1: the two waves of a SIMD steer their priorities by each other's progress in the row-stationary k-loop (+0.8 %)
#endif
#ifndef AGX_NN_AHEAD
#define AGX_NN_AHEAD 4 // activation fragments in flight per wave in the row-stationary k-loop
#endif
/* (Variants measured flat or negative and removed from the source — the records are in profiles/: a ring of three weight stages
 *  (r04_nn_ab*.txt: spills), the pair-balance ticks in the column-tile loop (-0.6 %), the interleaved read order of the column loop, the
 *  conditional layer prefetch / input prefetch / fused residual add / two-pass conv5x5 switches (all on since round 4), the
 *  operand-traffic timing experiments AGX_NN_DBG_NOLDS / _NOW (r03), the 12-wave tile and the accumulator initialisation inside the first
 *  stage (r05_nn_ab.txt); round 6: a row's fragment read once per channel chunk and SHIFTED BY LANES (DPP row_shr / row_shl) for the taps
 *  dx = -1, +1 — a third of the LDS reads for 8 vector moves per row, one between every two MFMAs: 20-23 % slower, vector instructions in
 *  the MFMA stream cost far more than the LDS reads they replace (r06_nn_ab6_lane_shifted_rows_negative.txt).) */

namespace
{
	typedef _Float16 half_t;
	typedef _Float16 half8 __attribute__((ext_vector_type(8)));
	typedef _Float16 half4 __attribute__((ext_vector_type(4)));
	typedef _Float16 half2 __attribute__((ext_vector_type(2)));
	typedef float floatx4 __attribute__((ext_vector_type(4)));

	struct NetParams
	{
			const half8 *w_in;      // packed conv5x5 fragments
			const half8 *w_tower;   // packed 3x3 fragments: 2*blocks layers, then the policy conv
			const float *bias;      // [1 + 2*blocks + 1][F]
			const float *wp2;       // [F]
			const float *wv1;       // [F][4]
			const half_t *wv2;      // value-head dense weights in MFMA A-fragment order [KPAD/32][D/16][lane][8] (value_head_kernel)
			half_t *vhead_x;        // [batch][KPAD]: the value head's conv1x1 output of every board of the launch, input of value_head_kernel
			const float *bv2;       // [D]
			const float *wv3;       // [D][3]
			float bp2;
			float bv1[4];
			float bv3[3];
			int blocks;
			int batch;
			const int *slot_list; // optional: batch element i is slot slot_list[i]
			const int *count_ptr; // optional: batch size read on the device
			const float *wq2;     // [F][4] action-values head 1x1 weights (3 outputs, padded), null without the head
			float bq2[3];
			float *q;             // action values out: float[slots][HW][2] = (win, draw) per cell, null = head not evaluated
			half4 *skip;          // single-plane variant only: residual inputs in accumulator layout, [workgroup][wave][MT][NTW][lane]
	};

#ifdef AGX_NN_PROFILE
	// profile builds only: shader cycles per phase, waves 0 and 4 of every workgroup
	__device__ unsigned long long g_nn_prof[2][16];
	struct NnStamp
	{
			unsigned long long last;
			bool on;
			int row;
			__device__ NnStamp(int wave, int lane) :
					last(clock64()), on(lane == 0 && (wave & 3) == 0), row(wave >> 2)
			{
			}
			__device__ void mark(int k)
			{
				const unsigned long long now = clock64();
				if (on)
					atomicAdd(&g_nn_prof[row][k], now - last);
				last = now;
			}
	};
#define AGX_NN_MARK(K) stamp.mark(K)
#define AGX_NN_STAMP_PARAM , NnStamp &stamp
#define AGX_NN_STAMP_ARG , stamp
#else
#define AGX_NN_MARK(K) do { } while (0)
#define AGX_NN_STAMP_PARAM
#define AGX_NN_STAMP_ARG
#endif

	template<int F, int ROWS, int COLS>
	struct Geometry
	{
			static constexpr int S = COLS + 1;                                   // row stride in positions
			static constexpr int NT = (ROWS * S + 15) / 16;                      // 16-position tiles of the output
			static constexpr int NPOS = 1 + S + NT * 16 + S + 2;                  // stored positions (index = position + 1)
			static constexpr int CH = F / 8;                                     // 16-byte chunks per position
			static constexpr int PPR = (16 / CH) > 0 ? (16 / CH) : 1;            // positions per 256-byte bank row
			// The 8 waves of a workgroup are CG channel groups x PG position groups: 4 x 2, except for 64-filter nets on boards whose tiles are
			// not rows (20x20: the tap-major k-loop, one activation fragment read per (tap, tile)), where 4 groups would leave a wave a single
			// 16-channel tile (MT = 1) per fragment read: 2 x 4 there (+12 % measured).  128-filter nets keep 4 x 2 on every board: with
			// 2 x 4 (MT = 4) each weight fragment is fetched by four waves instead of two and the kernel lost 8 %.
			// Row tiles with 128 filters (the 15x15 towers of BASELINE configs 1, 2, 4): 8 x 1 — a wave owns ONE 16-channel tile of ALL rows.  Each
			// weight fragment is then fetched by one wave instead of two (4 x 2: the two position groups; fetched by four, 2 x 4 lost 8 %), 3 per
			// stage and wave instead of 6: half the vector-memory instructions and L2 -> CU bytes, which ran at half the CU's L2 bandwidth at full
			// MFMA rate; the price is LDS reads — an activation fragment feeds 3 MFMAs instead of 6 (17 reads per 45 MFMAs instead of 10 per 48,
			// the LDS array 22 % -> 38 % busy at full MFMA rate) — and the gain besides: all eight waves carry the same 15 rows (no 8 / 7 split,
			// no runtime tile counts), 60 accumulator registers instead of 64.
			// Column tiles (below) keep 4 x 2: with eight groups the tail tiles' fragments feed ONE MFMA each and the LDS array carries 40 reads per 78
			// MFMAs only with the conflict-free PAD 2 plane (whose epilogue stores are 4-way): measured 2.7 % (whole chip) to 7 % (64 CUs) slower,
			// profiles/r06_nn_ab8_column_tiles_eight_groups_negative.txt.
			static constexpr bool COLT = (ROWS == 20 && COLS == 20 && F == 128);
			static constexpr int CG = (S == 16 && F >= 128) ? AGX_NN_ROW_GROUPS : ((S == 16 || F >= 128) ? 4 : 2);
			static constexpr int PG = 8 / CG;
			static constexpr int MT = F / (16 * CG);                             // 16-channel output tiles per wave
			// stages of weight fragments a wave holds in the row-stationary k-loop: 2 = a stage's fragments are requested one stage ahead.  (With one
			// channel tile per wave a set is 12 registers and deeper rings fit: 3 measured equal to 2, 4 is 18 % slower — profiles/r06_nn_ab4_weight_ring.txt;
			// with two tiles per wave a third set spilled, profiles/r04_nn_ab*.txt.  An L2 round trip is covered by one stage of MFMAs.)
			static constexpr int WRING = 2;
			static constexpr int STAGE_TAPS = 3;                                 // weight fragments per stage and channel tile (the row / column loops)
			// 20x20 boards with 128 filters: the tiles of rows 0..15 are COLUMNS (16 cells of one board column: lane r = row r) — the
			// neighbouring column is the same tile shifted by one position, so one activation fragment feeds the three taps dx = -1, 0, +1 like
			// a row's fragment feeds dy on 15x15 boards (conv3x3_mac_cols) — and rows 16..19 stay six ordinary tiles of consecutive positions
			// (16 rows x 21 = 336 positions = 21 whole tiles lie in front of them).  A position group owns COLS / PG columns + 6 / PG of those.
			static constexpr int COL_TILES = COLS / PG, TAIL_TILES = 6 / PG, TAIL_FIRST = 21;
			/* Bytes of a stored position.  Row tiles (S = 16: a tile's 16 positions are whole 256-byte bank rows) keep F halves per position and the
			 * XOR chunk swizzle below — every LDS address of the k-loop is one base + immediates anyway.  On column tiles the lanes of a fragment
			 * are 21 positions apart and the XOR term differs from fragment to fragment: 4.2 vector instructions of address arithmetic per
			 * ds_read_b128 (176 beside the 156 MFMAs of two stages, round 5), and as many per epilogue store.  There a position is PADDED by
			 * PAD x 16 bytes instead: chunk c of position i sits at i * POS_BYTES + 16 c — bank group (PAD * i + c) mod 16, the additive
			 * counterpart of the swizzle — so every fragment of a stage is ONE per-lane base + an immediate offset, and so is every epilogue
			 * store.  Bank model (MI355X_MICROARCH.md, LDS; scripts/nn_lds_banks.py): PAD 1 = the XOR swizzle's conflicts (column and tail
			 * reads 2-way, 8-byte stores 2-way), PAD 2 = every read conflict-free, stores 4-way.  (The one-position pad also keeps planes
			 * 16-byte aligned; 8-byte pads would give conflict-free stores but misalign ds_read_b128.) */
			static constexpr int PAD = COLT ? AGX_NN_COLS_PAD : 0;
			static constexpr int POS_BYTES = F * 2 + 16 * PAD;
			static constexpr int PLANE_BYTES = NPOS * POS_BYTES;
			static constexpr int NTW = COLT ? (COL_TILES + TAIL_TILES) : (NT + PG - 1) / PG; // position tiles per wave
			static constexpr int THREADS = 512;                                  // 8 waves, 2 per SIMD
			// (shifts and masks on purpose: written with / and % the 15x15 kernels came out 12 % (6x128) and 60 x (2x64) slower)
			__device__ static __forceinline__ int channel_group(int wave) { return wave & (CG - 1); }
			__device__ static __forceinline__ int first_tile(int wave) { return (wave >> (CG == 8 ? 3 : (CG == 4 ? 2 : 1))) * NTW; }
			/* Which 16-byte slot of its bank row a position's chunk c lives in: c ^ swizzle(stored index).  A ds_read_b128 is served in four
			 * groups of 16 lanes — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS): a group is 8 lanes
			 * of one 8-channel chunk and 8 of the next, and is conflict-free when its 16 slots differ.  index mod CH: consecutive positions
			 * take consecutive slots (unshifted tiles conflict-free, tiles shifted by an odd dx 2-way in two of the four groups).  Padded
			 * positions (PAD > 0) are not swizzled. */
			__device__ static __forceinline__ int swizzle(int index)
			{
				if constexpr (PAD > 0)
					return 0;
				else
					return (index / PPR) % CH;
			}
			// (Row tiles: slot (chunk + 2 * index) mod 16 instead of the XOR makes every fragment read of the row-stationary loop conflict-free — the XOR
			//  form reads one column shift in three 2-way — but the 8-byte epilogue stores 4-way: measured 3.6-3.9 % slower,
			//  profiles/r06_nn_ab12_rotation_swizzle_negative.txt; no linear XOR swizzle does better than the identity, scripts/nn_lds_banks.py.)
			__device__ static __forceinline__ int slot(int index, int chunk) { return chunk ^ swizzle(index); }
			/* row-major position (stride S, 0 = cell (0, 0)) of lane r's cell in tile n of the wave */
			__device__ static __forceinline__ int tile_position(int wave, int n, int r)
			{
				if constexpr (COLT)
				{
					const int pg = wave >> (CG == 8 ? 3 : 2);
					return (n < COL_TILES) ? (r * S + pg * COL_TILES + n) : ((TAIL_FIRST + pg * TAIL_TILES + (n - COL_TILES)) * 16 + r);
				}
				else
					return (first_tile(wave) + n) * 16 + r;
			}
			__device__ static __forceinline__ int tile_count(int wave)
			{
				if constexpr (COLT)
					return NTW;
				else if constexpr (PG == 1)
					return NT;
				else if constexpr (PG == 2)
					return (wave >> 2) ? (NT - NTW) : NTW;
				else
					return ((wave >> 1) == PG - 1) ? (NT - (PG - 1) * NTW) : NTW;
			}
			static constexpr int MTILES = F / 16;
			static constexpr int KC = F / 32;                                    // k-steps per tap
			// row stride of the padded input plane.  Row tiles with 128 filters: 32 positions, so that the plane's chunk swizzle ((index >> 2) & 3) does
			// not depend on the ROW — a lane's fragments of one column shift are one address + immediates (with stride S + 4 every fragment read of the
			// input conv cost ~15 vector instructions of address arithmetic: 380 beside the 75 MFMAs of a column shift)
			static constexpr int S5 = (S == 16 && F >= 128) ? 32 : S + 4;
			static constexpr int NPOS5 = (ROWS + 4) * S5 + 4;
			static constexpr int HW = ROWS * COLS;
			static constexpr int D = (2 * F < 256) ? 2 * F : 256;
			static constexpr int KPAD = (HW * 4 + 31) / 32 * 32;                 // value-head dense input length, padded to whole MFMA k-steps
			static constexpr int SCRATCH_FLOATS = HW * 4 + D + 8 + 256 + 8 + F * 4 + F + F * 4;
			static constexpr int LDS_BYTES = 2 * PLANE_BYTES + SCRATCH_FLOATS * 4;
			// single-plane variant (boards whose two planes do not fit): one plane + scratch + policy partial sums [CG][NT*16] + action-value sums [3][NT*16]
			static constexpr int LDS_BYTES_INPLACE = PLANE_BYTES + SCRATCH_FLOATS * 4 + CG * NT * 16 * 4 + 3 * NT * 16 * 4;
			static constexpr int SKIP_PER_WG = 8 * MT * NTW * 64;               // half4 elements of residual scratch per workgroup
	};

	/* A workgroup barrier for hand-offs through LDS only: orders (and waits for) this wave's LDS accesses, not its global stores in flight.
	 * __syncthreads() is s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier: in the single-plane kernel every wave reaches the layer barrier right
	 * behind the 2 * NTW global stores of its residual values (read back only by the same lane, a layer later) and would sit there for a
	 * store round trip, twice per residual block. */
	__device__ __forceinline__ void lds_barrier()
	{
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
		__builtin_amdgcn_s_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
	}
	/*
	 * The two waves of a SIMD (wave w and w + 4: the same channel group, the two position groups) share its MFMA pipe, and the hardware
	 * issues oldest-first: left alone the older wave runs ahead, reaches the layer barrier early and the younger one finishes ALONE — a
	 * lone wave hides none of its LDS / L2 latencies.  In-kernel stamps: the older wave waits 9–12 % of a board's time at layer barriers.
	 * Each wave publishes how far it is (a tick per loop turn, across layers and boards) and reads its partner's tick one turn later
	 * (the read is requested at the end of a turn and consumed at the start of the next: it has come back with the turn's last
	 * fragments); the wave that is ahead lowers its priority, the one behind raises it.
	 */
	__device__ __forceinline__ int* pair_progress()
	{ // one word per wave of the workgroup; zeroed by the kernel before its first board
		__shared__ int progress[8];
		return progress;
	}
	struct PairBalance
	{
			// (LDS address space spelled out: through plain `volatile int*` the two accesses of done() were FLAT instructions — to LDS by way of the
			//  vector-memory path, counted on both wait counters — in the middle of the k-loop's counted lgkmcnt / vmcnt waits)
			typedef __attribute__((address_space(3))) int lds_int;
			volatile lds_int *mine;
			const volatile lds_int *partner;
			int tick, seen;
			__device__ __forceinline__ PairBalance(int *progress, int wave) :
					mine((volatile lds_int*) (progress + wave)), partner((const volatile lds_int*) (progress + (wave ^ 4))), tick(progress[wave]), seen(progress[wave ^ 4])
			{
			}
			__device__ __forceinline__ void turn()
			{
#if AGX_NN_PAIR_BALANCE
				const int ahead = __builtin_amdgcn_readfirstlane(tick - seen); // wave-uniform: a scalar compare and branch
				if (ahead > 0)
					__builtin_amdgcn_s_setprio(1);
				else if (ahead == 0)
					__builtin_amdgcn_s_setprio(2);
				else
					__builtin_amdgcn_s_setprio(3);
#endif
			}
			__device__ __forceinline__ void done()
			{
#if AGX_NN_PAIR_BALANCE
				tick++;
				*mine = tick;
				seen = *partner;
#endif
			}
	};
	template<typename G>
	__device__ __forceinline__ int plane_offset(int index, int chunk)
	{ // byte offset of a 16-byte chunk of stored position `index` (= position + 1)
		return index * G::POS_BYTES + G::slot(index, chunk) * 16;
	}

	/* The weight fragments of a layer's first stage(s), requested by the layer in front of it: a layer that fetches them itself starts with an
	 * L2 / MALL round trip that nothing hides (every wave of the workgroup has just left the layer barrier).  `next` = the packed weights of
	 * the layer that follows. */
	template<typename G>
	struct WeightCarry
	{
			half8 a[G::WRING - 1][G::STAGE_TAPS][G::MT]; // the first WRING - 1 stages of the layer
			const half8 *next;
	};
	/* A layer's bias values, requested by the layer in front of it behind its k-loop: requested at the layer's own top — straight behind the
	 * layer barrier — every wave of the workgroup waits out an L2 round trip there, per channel tile, with nothing to hide it (the
	 * accumulators start from the bias).  Carried across the epilogue and the barrier only, where the k-loop's weight registers are free. */
	template<int MT>
	struct BiasCarry
	{
			floatx4 b[MT];
	};
	/* The residual input of a block's second layer (single-plane kernels: parked in the workgroup's global scratch), requested by the block's FIRST
	 * layer behind its k-loop — its accumulator registers are free there — so that the round trips ride through that layer's barrier and plane
	 * write instead of standing in front of the second layer's k-loop, where every wave of the workgroup waited for them. */
	template<typename G>
	struct SkipCarry
	{
			uint2 v[G::MT][G::NTW];
	};
	template<typename G>
	__device__ __forceinline__ void request_bias(const float *__restrict__ bias, int wave, int lane, BiasCarry<G::MT> &carry)
	{
#pragma unroll
		for (int i = 0; i < G::MT; i++)
			carry.b[i] = *reinterpret_cast<const floatx4*>(bias + (G::channel_group(wave) * G::MT + i) * 16 + 4 * (lane >> 4));
	}
	template<int F, int ROWS, int COLS>
	__device__ __forceinline__ void request_first_stage(const half8 *__restrict__ wpk, int wave, int lane, WeightCarry<Geometry<F, ROWS, COLS>> &carry)
	{
		typedef Geometry<F, ROWS, COLS> G;
		const half8 *wl = wpk + __builtin_amdgcn_readfirstlane(G::channel_group(wave) * G::STAGE_TAPS * G::MT * 64);
#pragma unroll
		for (int u = 0; u < G::WRING - 1; u++)
#pragma unroll
			for (int t = 0; t < G::STAGE_TAPS; t++)
#pragma unroll
				for (int i = 0; i < G::MT; i++)
					carry.a[u][t][i] = wl[u * (G::CG * G::STAGE_TAPS * G::MT * 64) + (t * G::MT + i) * 64 + lane];
	}

	/*
	 * 3x3 convolution + bias (+ skip) + ReLU over one board held in LDS.
	 * src, dst: activation planes; if SKIP the residual input is read from (and the result written to) dst.
	 */
	/* the default row epilogue of the row-stationary loop: none (the caller finishes the layer behind the k-loop) */
	struct NoRowEpilogue
	{
			static constexpr bool FUSED = false;
			__device__ __forceinline__ void operator()(int) const
			{
			}
	};
	template<int F, int ROWS, int COLS, bool FINISH = false, typename Epi = NoRowEpilogue>
	__device__ __forceinline__ void conv3x3_rows_stage(const char *src, const half8 *__restrict__ wnext, int kc, int dxi, int index_base, int q4,
			int my_tiles, int lane, const half8 (&a_cur)[3][Geometry<F, ROWS, COLS>::MT], half8 (&a_next)[3][Geometry<F, ROWS, COLS>::MT],
			floatx4 (&acc)[Geometry<F, ROWS, COLS>::MT][Geometry<F, ROWS, COLS>::NTW], const Epi &epi = Epi())
	{ // dxi is a compile-time constant at every call (conv3x3_mac_rows unrolls the three shifts of a chunk)
	  // FINISH: the layer's last stage — output row o is complete behind input row o + 1; epi(o) converts and stores it one row later, beside the
	  // MFMAs of the rows that follow (its vector instructions would otherwise wait for the matrix pipe to deliver the row)
		typedef Geometry<F, ROWS, COLS> G;
		static_assert(((G::KC - 1) << 6) < G::CH * 16, "a stage's chunk kc * 4 lies inside the chunk field of a position's bytes");
		// the next stage's 3 * MT weight fragments: contiguous for this wave (pack_conv_rows), one scalar base + small offsets
#pragma unroll
		for (int dyi = 0; dyi < 3; dyi++)
#pragma unroll
			for (int i = 0; i < G::MT; i++)
				a_next[dyi][i] = wnext[(dyi * G::MT + i) * 64 + lane];
		const int index0 = index_base + (dxi - 1);
		const int swz0 = (index0 / G::PPR) % G::CH; // invariant over rows: 16 positions == whole bank rows
		// (kc * 4 + q4) ^ swz0 == (kc * 4) ^ (q4 ^ swz0): the lane's part does not depend on the stage — one register per shift for the whole
		// kernel — and the stage's chunk is ONE exclusive-or with a scalar
		const int lane_part = index0 * G::CH * 16 + ((q4 ^ swz0) * 16);
		const char *src0 = src + (lane_part ^ (kc << 6));
		// input rows j = -1 .. NTW, each fragment used by its own three taps only: a window of AHEAD fragments in flight
		constexpr int AHEAD = AGX_NN_AHEAD;
		half8 b[AHEAD];
#pragma unroll
		for (int u = 0; u < AHEAD - 1; u++)
			b[u] = *reinterpret_cast<const half8*>(src0 + (u - 1) * (16 * G::CH * 16));
#pragma unroll
		for (int j = -1; j <= G::NTW; j++)
		{
			const int jn = j + AHEAD - 1; // the row requested now
			if (jn <= G::NTW)
				b[(jn + 1) % AHEAD] = *reinterpret_cast<const half8*>(src0 + ((jn <= my_tiles) ? jn : 0) * (16 * G::CH * 16));
#pragma unroll
			for (int dyi = 0; dyi < 3; dyi++)
			{
				const int o = j - (dyi - 1); // output row fed by input row j through the tap dy = dyi - 1
				if (o >= 0 && o < G::NTW && o < my_tiles)
				{
#pragma unroll
					for (int i = 0; i < G::MT; i++)
						acc[i][o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur[dyi][i], b[(j + 1) % AHEAD], acc[i][o], 0, 0, 0);
				}
			}
			if constexpr (FINISH)
			{
				const int done = j - 2;
				if (done >= 0 && done < my_tiles)
					epi(done);
			}
			// nothing crosses a row boundary: the fragment requested in this turn is the one used AHEAD - 1 turns later, so the wait in
			// front of a turn's MFMAs leaves the younger requests in flight (left to itself the scheduler sinks every request to just
			// before its use and a wave running alone on its SIMD stalls on each one)
			__builtin_amdgcn_sched_barrier(0);
		}
		if constexpr (FINISH)
		{
			if (G::NTW - 1 < my_tiles)
				epi(G::NTW - 1);
		}
	}

	/*
	 * The k-loop of a 3x3 convolution when a 16-position tile IS a board row (row stride S = 16, i.e. 15-column boards).
	 *
	 * The straightforward loop (conv3x3_mac_taps) walks the 9 taps and reads, per tap and 32-channel chunk, one activation
	 * fragment per output tile from LDS: every fragment feeds only MT MFMAs (8 KB of ds_read per 16 MFMAs per wave).  Here the loop is
	 * input-row stationary: for a 32-channel chunk and a column shift dx the fragment of INPUT row j (shifted by dx) is read once and
	 * multiplied by the three taps (dy = -1, 0, +1) of that column, accumulating into OUTPUT rows j + 1, j, j - 1.  A wave that owns 8
	 * output rows reads 10 input rows x 3 shifts = 30 fragments per chunk instead of 72: LDS traffic / 2.4, the same MFMAs.  Weight
	 * fragments are packed in consumption order (pack_conv_rows: stage = (chunk, dx), then channel group, dy, tile), so a stage's
	 * fragments are 3 * MT consecutive KB behind one scalar base.
	 */
	template<int F, int ROWS, int COLS, bool ZERO = true, typename Epi = NoRowEpilogue>
	__device__ __forceinline__ void conv3x3_mac_rows(const char *src, const half8 *__restrict__ wpk, int wave, int lane,
			floatx4 (&acc)[Geometry<F, ROWS, COLS>::MT][Geometry<F, ROWS, COLS>::NTW], WeightCarry<Geometry<F, ROWS, COLS>> *carry = nullptr, const Epi &epi = Epi())
	{ // epi (Epi::FUSED): the caller's epilogue of one output row, run inside the layer's last stage
		typedef Geometry<F, ROWS, COLS> G;
		static_assert(G::S == 16 && G::PAD == 0, "a position tile must be a board row (whole 256-byte bank rows, XOR chunk swizzle)");
		const int r = lane & 15;
		const int q4 = lane >> 4;
		const int mg = G::channel_group(wave);   // output channels [mg * 16 * MT, (mg + 1) * 16 * MT)
		const int n0 = G::first_tile(wave);
		const int my_tiles = G::tile_count(wave);

		if (ZERO)
		{
#pragma unroll
			for (int i = 0; i < G::MT; i++)
#pragma unroll
				for (int n = 0; n < G::NTW; n++)
					acc[i][n] = floatx4 { 0.0f, 0.0f, 0.0f, 0.0f };
		}

		constexpr int STAGES = 3 * G::KC;               // stage = (32-channel chunk kc, column shift dx)
		constexpr int STAGE_FRAGS = G::CG * 3 * G::MT * 64; // half8 elements of one stage: channel groups x 3 taps x MT tiles x 64 lanes
		constexpr int RING = G::WRING;                  // sets of weight fragments: a stage's are requested RING - 1 stages ahead
		static_assert(G::STAGE_TAPS == 3, "stage = (chunk, column shift): three taps");
		static_assert(STAGES % RING == 0, "RING stages per loop turn (static ring index)");
		const half8 *wl = wpk + __builtin_amdgcn_readfirstlane(mg * 3 * G::MT * 64); // wave-uniform: scalar base + lane offset
		half8 a[RING][3][G::MT];
		// (the layer's last requests have to ask for SOMETHING, see below: the next layer's first stages when there is a carry, its own otherwise)
		const half8 *wrap = (carry != nullptr) ? carry->next + __builtin_amdgcn_readfirstlane(mg * 3 * G::MT * 64) : wl;
#pragma unroll
		for (int u = 0; u < RING - 1; u++)
#pragma unroll
			for (int dyi = 0; dyi < 3; dyi++)
#pragma unroll
				for (int i = 0; i < G::MT; i++)
					a[u][dyi][i] = (carry != nullptr) ? carry->a[u][dyi][i] : wl[u * STAGE_FRAGS + (dyi * G::MT + i) * 64 + lane];
		const int index_base = 1 + G::S + n0 * 16 + r; // stored index of this lane's position in the wave's first output row
#if AGX_NN_PAIR_BALANCE
		PairBalance balance(pair_progress(), wave);
#endif
		// A loop turn is TURN = 6 stages — two channel chunks x the three column shifts — so that a stage's shift and its set of weight fragments
		// are compile-time constants: no scalar divisions by three, the LDS address of a stage is one exclusive-or, the weight requests are scalar
		// base + immediate.  (Vector instructions in the MFMA stream are expensive: profiles/r06_nn_ab6_lane_shifted_rows_negative.txt.)
		constexpr int TURN = 6;
		static_assert(STAGES % TURN == 0 && TURN % RING == 0, "whole turns, static ring index");
		auto turn = [&](int s, auto last_turn)
		{
			constexpr bool LAST_TURN = decltype(last_turn)::value;
#pragma unroll
			for (int u = 0; u < TURN; u++)
			{
				// The two waves of a SIMD are issued oldest-first: left alone, the older one runs ahead, finishes its k-loop early and waits at
				// the layer barrier while the younger one finishes ALONE (a lone wave hides none of its LDS / L2 latencies: measured 2.1 x its
				// MFMA time).  Priority by remaining work — the wave that is behind goes first — keeps the pair together to the end.
				// (a static bias towards the younger wave of a pair was measured worse)
				if (u % 2 == 0)
				{
#if AGX_NN_PAIR_BALANCE
					balance.turn();
#else
					if (3 * (s + u) < STAGES)
						__builtin_amdgcn_s_setprio(3);
					else if (3 * (s + u) < 2 * STAGES)
						__builtin_amdgcn_s_setprio(2);
					else
						__builtin_amdgcn_s_setprio(1);
#endif
				}
				// stage s + u computes with set u % RING and requests stage s + u + RING - 1 into the set that stage s + u - 1 has just finished
				// with.  Past the layer's end the requests go on with the NEXT layer's first stages (its own again without a carry) instead of
				// branching around the fetch: with a conditional fetch the wait for THIS stage's fragments has to assume the newer loads were
				// never issued (vmcnt(0)), which serialises fetch and MFMAs in every turn
				const int ahead = s + u + RING - 1;
				const half8 *wnext = (u + RING - 1 < TURN || ahead < STAGES) ? wl + ahead * STAGE_FRAGS : wrap + (ahead - STAGES) * STAGE_FRAGS;
				if (LAST_TURN && Epi::FUSED && u == TURN - 1) // (a constant once the turn is unrolled)
					conv3x3_rows_stage<F, ROWS, COLS, true, Epi>(src, wnext, s / 3 + u / 3, u % 3, index_base, q4, my_tiles, lane, a[u % RING], a[(u + RING - 1) % RING], acc, epi);
				else
					conv3x3_rows_stage<F, ROWS, COLS, false, Epi>(src, wnext, s / 3 + u / 3, u % 3, index_base, q4, my_tiles, lane, a[u % RING], a[(u + RING - 1) % RING], acc, epi);
#if AGX_NN_PAIR_BALANCE
				if (u % 2 == 1)
					balance.done();
#endif
			}
		};
		// (the layer's last turn is its own copy of the code: its last stage carries the caller's row epilogue)
#pragma unroll 1
		for (int s = 0; s + TURN < STAGES; s += TURN)
			turn(s, std::false_type());
		turn(STAGES - TURN, std::true_type());
		__builtin_amdgcn_s_setprio(0);
		if (carry != nullptr)
		{ // (behind the last turn sets 0 .. RING - 2 hold the next layer's stages 0 .. RING - 2)
#pragma unroll
			for (int u = 0; u < RING - 1; u++)
#pragma unroll
				for (int dyi = 0; dyi < 3; dyi++)
#pragma unroll
					for (int i = 0; i < G::MT; i++)
						carry->a[u][dyi][i] = a[u][dyi][i];
		}
	}

	template<int F, int ROWS, int COLS, bool ZERO = true>
	__device__ __forceinline__ void conv3x3_mac_taps(const char *src, const half8 *__restrict__ wpk, int wave, int lane,
			floatx4 (&acc)[Geometry<F, ROWS, COLS>::MT][Geometry<F, ROWS, COLS>::NTW])
	{
		typedef Geometry<F, ROWS, COLS> G;
		static_assert(G::PAD == 0, "tiles of consecutive positions in unpadded planes (the XOR chunk swizzle is invariant over a wave's tiles)");
		const int r = lane & 15;
		const int q4 = lane >> 4;
		const int mg = G::channel_group(wave);   // output channels [mg * 16 * MT, (mg + 1) * 16 * MT)
		const int n0 = G::first_tile(wave);
		const int my_tiles = G::tile_count(wave);

		if (ZERO)
		{
#pragma unroll
			for (int i = 0; i < G::MT; i++)
#pragma unroll
				for (int n = 0; n < G::NTW; n++)
					acc[i][n] = floatx4 { 0.0f, 0.0f, 0.0f, 0.0f };
		}

		/*
		 * 9*KC k-steps (one k-step = 32 input channels of one tap).  Two waves share a SIMD, so while one waits for its LDS /
		 * L2 operands the other issues MFMAs; inside a wave the activation fragments of step s+1 and the weight fragments of
		 * step s+1 are requested before the MFMAs of step s (two register sets each).
		 */
		constexpr int STEPS = 9 * G::KC;
		constexpr int RING = (G::KC == 4) ? 4 : 2; // weight fragments are fetched RING-1 k-steps ahead (L2 latency > one k-step of MFMAs)
		static_assert(G::KC % RING == 0, "ring index must be static inside the unrolled k loop");
		const half8 *wp = wpk + (mg * G::MT) * 64 + lane;
		half8 a_ring[RING][G::MT];
#pragma unroll
		for (int u = 0; u < RING - 1; u++)
#pragma unroll
			for (int i = 0; i < G::MT; i++)
				a_ring[u][i] = wp[(u * G::MTILES + i) * 64];
#pragma unroll 3
		for (int t = 0; t < 9; t++)
		{
			const int off = (t / 3 - 1) * G::S + (t % 3 - 1);
			const int index0 = 1 + G::S + n0 * 16 + r + off;    // stored index of this lane's position in the wave's first tile
			const int swz0 = (index0 / G::PPR) % G::CH;          // invariant over tiles: 16 positions == whole bank rows
			const char *src0 = src + index0 * G::CH * 16;
#pragma unroll
			for (int kc = 0; kc < G::KC; kc++)
			{
				const int ahead = t * G::KC + kc + RING - 1;
				if (ahead < STEPS)
				{
#pragma unroll
					for (int i = 0; i < G::MT; i++)
						a_ring[(kc + RING - 1) % RING][i] = wp[(ahead * G::MTILES + i) * 64];
				}
				// activation fragments in groups of at most 8 tiles (one group on 15x15; two on 20x20, where 14 live fragments at
				// once would push the kernel over its register budget)
				constexpr int BG = 8;
#pragma unroll
				for (int g0 = 0; g0 < G::NTW; g0 += BG)
				{
					half8 b[BG];
#pragma unroll
					for (int n = 0; n < BG; n++)
						if (g0 + n < G::NTW)
							b[n] = *reinterpret_cast<const half8*>(src0 + ((g0 + n < my_tiles) ? (g0 + n) : 0) * (16 * G::CH * 16) + (((kc * 4 + q4) ^ swz0) * 16));
					// scheduling hint: hipcc otherwise threads the MFMAs of a k-step through its operand loads; iglp_opt(0) (the built-in
					// DS-read / MFMA interleave for small GEMM loops) keeps the cluster together: +6-8 % on the 15x15 kernels (the same as
					// s_setprio 1 / 0 around the cluster) and +3 % on 20x20, measured A/B on one box
					__builtin_amdgcn_iglp_opt(0);
#pragma unroll
					for (int n = 0; n < BG; n++)
						if (g0 + n < G::NTW && g0 + n < my_tiles)
						{
#pragma unroll
							for (int i = 0; i < G::MT; i++)
								acc[i][g0 + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_ring[kc % RING][i], b[n], acc[i][g0 + n], 0, 0, 0);
						}
				}
			}
		}
	}

	/*
	 * The k-loop of a 3x3 convolution on column tiles (Geometry::COLT: 20x20 boards, 128 filters).
	 *
	 * A 20-column board has row stride 21, so a 16-position tile is not a row and conv3x3_mac_rows' trick — one fragment of an input row
	 * feeds three output rows — has nothing to hold on to; the tap-major loop reads one 1 KB fragment per (tap, tile) for MT = 2 MFMAs,
	 * and at 14 tiles per wave the LDS reads of a k-step take as long as its MFMAs (the kernel sat at 0.39 of the launch's MFMA peak).
	 * With tiles that are COLUMNS (lane r = row r of column x) the neighbouring column is the same tile one position on: for a 32-channel
	 * chunk and a row shift dy the fragment of INPUT column c is read once and multiplied by the three taps dx = -1, 0, +1 of that row
	 * shift, accumulating into OUTPUT columns c + 1, c, c - 1.  A wave owns 10 output columns (12 input fragments per stage instead of
	 * 30) and three ordinary tiles of rows 16..19, which take their three shifted fragments from the same stage's weights: 21 fragment
	 * reads per stage of 78 MFMAs, against 42 per 84 before.  Lanes of a fragment are 21 positions apart: 21 r mod 16 = 5 r mod 16 is a
	 * permutation, so the padded positions (Geometry::PAD: bank group (PAD * position + chunk) mod 16) spread the 16 lanes of a read over all banks.
	 * Weight fragments in consumption order (pack_conv_rows with the roles of dx and dy exchanged): stage = (chunk, dy), then channel
	 * group, dx, tile.
	 */
	template<int F, int ROWS, int COLS>
	__device__ __forceinline__ void conv3x3_cols_stage(const char *src, const half8 *__restrict__ wnext, int kc, int dyi, int col_base, int tail_base, int q4, int lane,
			const half8 (&a_cur)[3][Geometry<F, ROWS, COLS>::MT], half8 (&a_next)[3][Geometry<F, ROWS, COLS>::MT],
			floatx4 (&acc)[Geometry<F, ROWS, COLS>::MT][Geometry<F, ROWS, COLS>::NTW])
	{
		typedef Geometry<F, ROWS, COLS> G;
		static_assert(G::PAD > 0, "padded positions: every fragment of a stage is a per-lane base + an immediate offset");
#pragma unroll
		for (int dxi = 0; dxi < 3; dxi++)
#pragma unroll
			for (int i = 0; i < G::MT; i++)
				a_next[dxi][i] = wnext[(dxi * G::MT + i) * 64 + lane];
		constexpr int NCOL = G::COL_TILES + 2;             // input columns x0 - 1 .. x0 + COL_TILES
		constexpr int NFRAG = NCOL + 3 * G::TAIL_TILES;     // then (tail tile, dx) pairs
		// the stage's two bases: this lane's cell in input column x0 - 1 and in the wave's first tail tile, both shifted by dy rows, chunk kc * 4 + q4
		const int stage_off = (dyi - 1) * G::S * G::POS_BYTES + kc * 64; // wave-uniform
		const char *col_ptr = src + col_base + stage_off;
		const char *tail_ptr = src + tail_base + stage_off;
		auto fragment = [&](int k) -> half8
		{ // k < NCOL: input column k; else the pair (tail tile, dx) = ((k - NCOL) / 3, (k - NCOL) % 3)
			if (k < NCOL)
				return *reinterpret_cast<const half8*>(col_ptr + k * G::POS_BYTES);
			return *reinterpret_cast<const half8*>(tail_ptr + (((k - NCOL) / 3) * 16 + ((k - NCOL) % 3 - 1)) * G::POS_BYTES);
		};
		constexpr int AHEAD = AGX_NN_COLS_AHEAD;
		half8 b[AHEAD];
#pragma unroll
		for (int u = 0; u < AHEAD - 1; u++)
			b[u] = fragment(u);
#pragma unroll
		for (int f = 0; f < NFRAG; f++)
		{
			const int fn = f + AHEAD - 1;
			if (fn < NFRAG)
				b[fn % AHEAD] = fragment(fn);
			const int k = f;
			if (k < NCOL)
			{
#pragma unroll
				for (int dxi = 0; dxi < 3; dxi++)
				{
					const int o = k - dxi; // output column (relative to the wave's first) fed by input column k - 1 through the tap dx = dxi - 1
					if (o >= 0 && o < G::COL_TILES)
					{
#pragma unroll
						for (int i = 0; i < G::MT; i++)
							acc[i][o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur[dxi][i], b[f % AHEAD], acc[i][o], 0, 0, 0);
					}
				}
			}
			else
			{
				const int t = (k - NCOL) / 3, dxi = (k - NCOL) % 3;
#pragma unroll
				for (int i = 0; i < G::MT; i++)
					acc[i][G::COL_TILES + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur[dxi][i], b[f % AHEAD], acc[i][G::COL_TILES + t], 0, 0, 0);
			}
			__builtin_amdgcn_sched_barrier(0); // (as in conv3x3_rows_stage: keep AHEAD - 1 requests in flight behind every turn's MFMAs)
		}
	}
	template<int F, int ROWS, int COLS, bool ZERO = true>
	__device__ __forceinline__ void conv3x3_mac_cols(const char *src, const half8 *__restrict__ wpk, int wave, int lane,
			floatx4 (&acc)[Geometry<F, ROWS, COLS>::MT][Geometry<F, ROWS, COLS>::NTW], WeightCarry<Geometry<F, ROWS, COLS>> *carry = nullptr)
	{ // carry: as in conv3x3_mac_rows — the first stage's fragments come from the layer in front, the last requests are the next layer's first stage
		typedef Geometry<F, ROWS, COLS> G;
		static_assert(G::COLT && (G::CG == 4 || G::CG == 8), "column tiles: four channel groups x two position groups, or eight x one");
		const int r = lane & 15;
		const int q4 = lane >> 4;
		const int mg = G::channel_group(wave);
		const int pg = wave >> (G::CG == 8 ? 3 : 2);
		if (ZERO)
		{
#pragma unroll
			for (int i = 0; i < G::MT; i++)
#pragma unroll
				for (int n = 0; n < G::NTW; n++)
					acc[i][n] = floatx4 { 0.0f, 0.0f, 0.0f, 0.0f };
		}
		constexpr int STAGES = 3 * G::KC;               // stage = (32-channel chunk kc, row shift dy)
		constexpr int STAGE_FRAGS = G::CG * 3 * G::MT * 64;
		static_assert(STAGES % 2 == 0, "two stages per loop turn (static ring index)");
		const half8 *wl = wpk + __builtin_amdgcn_readfirstlane(mg * 3 * G::MT * 64);
		static_assert(G::WRING == 2, "two sets of weight fragments");
		const half8 *wrap = (carry != nullptr) ? carry->next + __builtin_amdgcn_readfirstlane(mg * 3 * G::MT * 64) : wl;
		half8 a0[3][G::MT], a1[3][G::MT];
#pragma unroll
		for (int dxi = 0; dxi < 3; dxi++)
#pragma unroll
			for (int i = 0; i < G::MT; i++)
				a0[dxi][i] = (carry != nullptr) ? carry->a[0][dxi][i] : wl[(dxi * G::MT + i) * 64 + lane];
		// byte offsets (chunk q4) of this lane's cell (row r, column x0 - 1) and of its cell in the wave's first tail tile
		const int col_base = plane_offset<G>(1 + G::S + r * G::S + pg * G::COL_TILES - 1, q4);
		const int tail_base = plane_offset<G>(1 + G::S + (G::TAIL_FIRST + pg * G::TAIL_TILES) * 16 + r, q4);
#pragma unroll 1
		for (int s = 0; s + 2 < STAGES; s += 2)
		{
			if (3 * s < STAGES) // (priority by remaining work)
				__builtin_amdgcn_s_setprio(3);
			else if (3 * s < 2 * STAGES)
				__builtin_amdgcn_s_setprio(2);
			else
				__builtin_amdgcn_s_setprio(1);
			conv3x3_cols_stage<F, ROWS, COLS>(src, wl + (s + 1) * STAGE_FRAGS, s / 3, s % 3, col_base, tail_base, q4, lane, a0, a1, acc);
			conv3x3_cols_stage<F, ROWS, COLS>(src, wl + (s + 2) * STAGE_FRAGS, (s + 1) / 3, (s + 1) % 3, col_base, tail_base, q4, lane, a1, a0, acc);
		}
		// (the layer's last two stages: the very last requests the NEXT layer's first stage — its own again without a carry — see conv3x3_mac_rows)
		conv3x3_cols_stage<F, ROWS, COLS>(src, wl + (STAGES - 1) * STAGE_FRAGS, (STAGES - 2) / 3, (STAGES - 2) % 3, col_base, tail_base, q4, lane, a0, a1, acc);
		conv3x3_cols_stage<F, ROWS, COLS>(src, wrap, (STAGES - 1) / 3, (STAGES - 1) % 3, col_base, tail_base, q4, lane, a1, a0, acc);
		__builtin_amdgcn_s_setprio(0);
		if (carry != nullptr)
		{ // (behind the last stage set 0 holds the next layer's first stage)
#pragma unroll
			for (int dxi = 0; dxi < 3; dxi++)
#pragma unroll
				for (int i = 0; i < G::MT; i++)
					carry->a[0][dxi][i] = a0[dxi][i];
		}
	}

	template<int F, int ROWS, int COLS, bool ZERO = true, typename Epi = NoRowEpilogue>
	__device__ __forceinline__ void conv3x3_mac(const char *src, const half8 *__restrict__ wpk, int wave, int lane,
			floatx4 (&acc)[Geometry<F, ROWS, COLS>::MT][Geometry<F, ROWS, COLS>::NTW], WeightCarry<Geometry<F, ROWS, COLS>> *carry = nullptr, const Epi &epi = Epi())
	{
		static_assert(!Epi::FUSED || Geometry<F, ROWS, COLS>::S == 16, "only the row-stationary loop runs a row epilogue inside its last stage");
		if constexpr (Geometry<F, ROWS, COLS>::S == 16)
			conv3x3_mac_rows<F, ROWS, COLS, ZERO, Epi>(src, wpk, wave, lane, acc, carry, epi);
		else if constexpr (Geometry<F, ROWS, COLS>::COLT)
			conv3x3_mac_cols<F, ROWS, COLS, ZERO>(src, wpk, wave, lane, acc, carry);
		else
			conv3x3_mac_taps<F, ROWS, COLS, ZERO>(src, wpk, wave, lane, acc);
	}

	__device__ __forceinline__ float half_plus_float_lo(uint32_t packed_halves, float addend)
	{
		float d;
		asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(packed_halves), "v"(addend));
		return d;
	}
	__device__ __forceinline__ float half_plus_float_hi(uint32_t packed_halves, float addend)
	{
		float d;
		asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(packed_halves), "v"(addend));
		return d;
	}
	template<bool TANH>
	__device__ __forceinline__ float activation(float x)
	{ // ReLU of the tower / policy head, tanh of the action-values head (blocks.cpp:119-127)
		return TANH ? tanhf(x) : fmaxf(x, 0.0f);
	}

	/* The epilogue of ONE output row of a row-tile layer (conv3x3): convert, ReLU on packed halves, store — the spare column's lanes sit the
	 * store out.  Run by the row-stationary loop inside the layer's last stage (conv3x3_rows_stage<.., FINISH>). */
	template<typename G>
	struct RowEpilogue
	{
			static constexpr bool FUSED = true;
			floatx4 (&acc)[G::MT][G::NTW];
			char *(&out0)[G::MT]; // this lane's cell in the wave's first row, per channel tile
			int r;
			__device__ __forceinline__ void operator()(int n) const
			{
				if (r < G::S - 1)
				{
#pragma unroll
					for (int i = 0; i < G::MT; i++)
					{
						const floatx4 v = acc[i][n];
						half2 lo { static_cast<half_t>(v[0]), static_cast<half_t>(v[1]) }, hi { static_cast<half_t>(v[2]), static_cast<half_t>(v[3]) };
						const half2 zero2 { static_cast<half_t>(0.0f), static_cast<half_t>(0.0f) };
						lo = __builtin_elementwise_max(lo, zero2);
						hi = __builtin_elementwise_max(hi, zero2);
						*reinterpret_cast<uint2*>(out0[i] + n * 16 * G::POS_BYTES) = uint2 { __builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi) };
					}
				}
			}
	};
	template<int F, int ROWS, int COLS, bool SKIP, bool TANH = false>
	__device__ __forceinline__ void conv3x3(const char *src, char *dst, const half8 *__restrict__ wpk, const float *__restrict__ bias, int wave,
			int lane AGX_NN_STAMP_PARAM, WeightCarry<Geometry<F, ROWS, COLS>> *carry = nullptr, BiasCarry<Geometry<F, ROWS, COLS>::MT> *bias_carry = nullptr,
			const float *__restrict__ next_bias = nullptr)
	{ // bias_carry: holds this layer's bias values on entry and the next layer's (next_bias) on return
		typedef Geometry<F, ROWS, COLS> G;
		const int r = lane & 15;
		const int q4 = lane >> 4;
		const int mg = G::channel_group(wave);   // output channels [mg * 16 * MT, (mg + 1) * 16 * MT)
		const int n0 = G::first_tile(wave);
		const int my_tiles = G::tile_count(wave);
		// The accumulators START from bias (+ residual input): the adds of the epilogue move to the layer's beginning, where they run in
		// the shadow of the first weight fetch, and the epilogue — which both waves of a SIMD reach with no MFMAs left to hide behind —
		// shrinks to ReLU, convert, mask, store.
		floatx4 acc[G::MT][G::NTW];
		// Row tiles (S = 16): tile n of the wave is 16 whole bank rows behind tile 0 and the chunk swizzle does not depend on the tile, so a lane's
		// accesses to its cells are ONE address per channel tile + immediates (n * 4096 bytes) — written out here: left to plane_offset() of each
		// tile's position the compiler kept MT * NTW per-lane offsets in registers for the whole kernel and added one to the plane per access.
		constexpr bool ROW_TILES = (G::S == 16);
		const int index0 = 1 + G::S + n0 * 16 + r; // stored index of this lane's cell in the wave's first tile (row tiles)
#pragma unroll
		for (int i = 0; i < G::MT; i++)
		{
			const int ch = (mg * G::MT + i) * 16 + 4 * q4;
			const floatx4 bv = (bias_carry != nullptr) ? bias_carry->b[i] : *reinterpret_cast<const floatx4*>(bias + ch);
			const char *skip0 = dst + plane_offset<G>(index0, ch / 8) + (ch % 8) * 2;
#pragma unroll
			for (int n = 0; n < G::NTW; n++)
			{
				floatx4 v = bv;
				if (SKIP && n < my_tiles)
				{
					const int pos = G::S + G::tile_position(wave, n, r);
					// bias + (float) residual as ONE instruction per value: v_fma_mix_f32 widens the fp16 operand itself (h * 1.0 + b rounds once,
					// exactly like the conversion followed by the add)
					const uint2 sk = ROW_TILES ? *reinterpret_cast<const uint2*>(skip0 + n * 16 * G::POS_BYTES)
							: *reinterpret_cast<const uint2*>(dst + plane_offset<G>(pos + 1, ch / 8) + (ch % 8) * 2);
					v[0] = half_plus_float_lo(sk.x, bv[0]);
					v[1] = half_plus_float_hi(sk.x, bv[1]);
					v[2] = half_plus_float_lo(sk.y, bv[2]);
					v[3] = half_plus_float_hi(sk.y, bv[3]);
				}
				acc[i][n] = v;
			}
		}
		AGX_NN_MARK(2);
		if constexpr (ROW_TILES && !TANH)
		{
			// Every tile of a wave is a board row (NT == ROWS): the only cell of a tile that is not on the board is the spare column's, lane
			// r == COLS of every tile.  Those lanes sit the stores out (an exec mask) instead of every tile masking its values: the spare column
			// stays zero from the plane's clearing.  ReLU after the conversion (rounding is monotonic, so max(cvt(x), 0) == cvt(max(x, 0))) on
			// packed halves: 4 vector instructions per tile (2 conversions, 2 packed max) + one ds_write_b64 at an immediate offset.  The rows
			// are finished INSIDE the layer's last stage, each beside the MFMAs of the rows behind it (conv3x3_rows_stage<.., FINISH>): nobody
			// reads `dst` during this layer, and behind the k-loop both waves of a SIMD would convert and store with no MFMA left in flight.
			static_assert(G::NT == ROWS && G::S == COLS + 1, "a tile is a board row + the spare column");
			char *out0[G::MT];
#pragma unroll
			for (int i = 0; i < G::MT; i++)
			{
				const int ch = (mg * G::MT + i) * 16 + 4 * q4;
				out0[i] = dst + plane_offset<G>(index0, ch / 8) + (ch % 8) * 2;
			}
			const RowEpilogue<G> epilogue { acc, out0, r };
			if constexpr (G::MT == 1)
			{ // (one channel tile: the next layer's bias values are four registers — requested here they have the whole k-loop to arrive)
				if (bias_carry != nullptr)
					request_bias<G>(next_bias, wave, lane, *bias_carry);
			}
			conv3x3_mac<F, ROWS, COLS, false, RowEpilogue<G>>(src, wpk, wave, lane, acc, carry, epilogue);
			AGX_NN_MARK(3);
			if constexpr (G::MT != 1)
			{
				if (bias_carry != nullptr)
					request_bias<G>(next_bias, wave, lane, *bias_carry);
			}
			AGX_NN_MARK(4);
			return;
		}
		conv3x3_mac<F, ROWS, COLS, false>(src, wpk, wave, lane, acc, carry);
		AGX_NN_MARK(3);
		if (bias_carry != nullptr)
			request_bias<G>(next_bias, wave, lane, *bias_carry);

		// epilogue: lane holds out-channels 4*q4 .. 4*q4+3 of tile i for position r of tile n (boards whose tiles are not rows, tanh layers)
#pragma unroll
		for (int i = 0; i < G::MT; i++)
		{
			const int ch = (mg * G::MT + i) * 16 + 4 * q4;
#pragma unroll
			for (int n = 0; n < G::NTW; n++)
				if (n < my_tiles)
				{
					const int pos = G::S + G::tile_position(wave, n, r);
					const int x = pos % G::S;
					const int y = pos / G::S - 1;
					const bool valid = (x < COLS) && (y < ROWS);
					char *ptr = dst + plane_offset<G>(pos + 1, ch / 8) + (ch % 8) * 2;
					const floatx4 v = acc[i][n];
					if (TANH)
					{
						half4 o;
						o[0] = static_cast<half_t>(valid ? activation<TANH>(v[0]) : 0.0f);
						o[1] = static_cast<half_t>(valid ? activation<TANH>(v[1]) : 0.0f);
						o[2] = static_cast<half_t>(valid ? activation<TANH>(v[2]) : 0.0f);
						o[3] = static_cast<half_t>(valid ? activation<TANH>(v[3]) : 0.0f);
						*reinterpret_cast<half4*>(ptr) = o;
					}
					else
					{ // (boards whose tiles are not rows: the spare column / overhang cells masked to zero by an AND)
						half2 lo { static_cast<half_t>(v[0]), static_cast<half_t>(v[1]) }, hi { static_cast<half_t>(v[2]), static_cast<half_t>(v[3]) };
						const half2 zero2 { static_cast<half_t>(0.0f), static_cast<half_t>(0.0f) };
						lo = __builtin_elementwise_max(lo, zero2);
						hi = __builtin_elementwise_max(hi, zero2);
						const uint32_t keep = valid ? 0xFFFFFFFFu : 0u;
						uint2 packed;
						packed.x = __builtin_bit_cast(uint32_t, lo) & keep;
						packed.y = __builtin_bit_cast(uint32_t, hi) & keep;
						*reinterpret_cast<uint2*>(ptr) = packed;
					}
				}
		}
		AGX_NN_MARK(4);
	}

	/*
	 * Single-plane variant of the 3x3 convolution for boards whose two planes do not fit into LDS (20x20 with F = 128: one plane is
	 * 119 KB).  A wave keeps ALL its outputs in accumulators until every wave has finished reading the plane, so the layer can
	 * be written over its own input.  The residual input cannot stay in LDS then: each lane parks the values it will need again
	 * (same accumulator layout, 8 bytes per lane and tile) in a per-workgroup scratch in global memory — written and read by the
	 * same lane, fully coalesced, 2 x 110 KB per residual block against ~120 MFLOP of MFMA work.
	 * MODE 0: first conv of a block (ReLU)   MODE 1: second conv (+ skip, ReLU, new skip saved)
	 * MODE 2: policy conv + ReLU folded with the 1x1 policy conv: per-channel-group partial logits into `ppart` [CG][NT*16].
	 * MODE 3: action-values conv + tanh folded with its 1x1 conv to 3 outputs: `ppart` is [3][NT*16], the channel groups
	 *         add their partial sums one after the other (fixed order, so results do not depend on wave timing).
	 */
	template<int F, int ROWS, int COLS, int MODE>
	__device__ __forceinline__ void conv3x3_inplace(char *plane, const half8 *__restrict__ wpk, BiasCarry<Geometry<F, ROWS, COLS>::MT> &bias_carry,
			const float *__restrict__ next_bias, half4 *skip, const float *__restrict__ wp2, float *ppart, int wave, int lane AGX_NN_STAMP_PARAM,
			SkipCarry<Geometry<F, ROWS, COLS>> *skip_carry = nullptr, WeightCarry<Geometry<F, ROWS, COLS>> *carry = nullptr)
	{ // carry: the layer's first weight fragments on entry, the next layer's on return (conv3x3_mac_rows / _cols)
	  // bias_carry: this layer's bias values on entry, the next layer's (next_bias, may be null: nothing follows) on return
	  // skip_carry: mode 0 fills it behind its k-loop with the block's residual input, mode 1 starts its accumulators from it
		typedef Geometry<F, ROWS, COLS> G;
		const int r = lane & 15;
		const int q4 = lane >> 4;
		const int mg = G::channel_group(wave);   // output channels [mg * 16 * MT, (mg + 1) * 16 * MT)
		const int n0 = G::first_tile(wave);
		const int my_tiles = G::tile_count(wave);
		floatx4 acc[G::MT][G::NTW];
		// (the 2 * MT * NTW addresses of a lane's residual values do not depend on the layer: hipcc computed all of them — 26 register pairs on 20x20 —
		//  once per kernel, kept what fitted and reloaded the rest from scratch one by one, each reload followed by s_waitcnt vmcnt(0), in front of
		//  every layer's residual loads and stores.  With the LANE's part of the address opaque they are this layer's own: a wave-uniform base in
		//  scalar registers + one 32-bit lane offset + immediates (the scalar-base form of global_load / global_store), no 64-bit vector address
		//  arithmetic per access.)
		typedef __attribute__((address_space(1))) half4 global_half4;
		int skip_lane = lane;
		asm volatile("" : "+v"(skip_lane));
		global_half4 *my_skip = (global_half4*) (skip + __builtin_amdgcn_readfirstlane(wave * G::MT * G::NTW * 64)) + skip_lane;
		if (MODE == 0 || MODE == 1)
		{ // The accumulators START from bias (+ the residual input, fetched from the workgroup's scratch): requested here, the 2 * NTW loads of a
		  // lane are in flight together behind the layer's first weight fetch.  In the epilogue — where nothing is left to hide a global
		  // round trip behind — they were 36 % of a 20x20 board's time (in-kernel stamps).
#pragma unroll
			for (int i = 0; i < G::MT; i++)
			{
				const floatx4 bv = bias_carry.b[i];
#pragma unroll
				for (int n = 0; n < G::NTW; n++)
				{
					floatx4 v = bv;
					if (MODE == 1)
					{ // bias + (float) residual as one v_fma_mix_f32 per value (conv3x3: h * 1.0 + b rounds once, like the conversion followed by the add)
						const uint2 sk = (skip_carry != nullptr) ? skip_carry->v[i][n] : __builtin_bit_cast(uint2, my_skip[(i * G::NTW + n) * 64]);
						v[0] = half_plus_float_lo(sk.x, bv[0]);
						v[1] = half_plus_float_hi(sk.x, bv[1]);
						v[2] = half_plus_float_lo(sk.y, bv[2]);
						v[3] = half_plus_float_hi(sk.y, bv[3]);
					}
					acc[i][n] = v;
				}
			}
			AGX_NN_MARK(2);
			conv3x3_mac<F, ROWS, COLS, false>(plane, wpk, wave, lane, acc, carry);
		}
		else
		{
			AGX_NN_MARK(2);
			conv3x3_mac<F, ROWS, COLS>(plane, wpk, wave, lane, acc, carry);
		}
		AGX_NN_MARK(3);
		const BiasCarry<G::MT> bias_now = bias_carry; // (modes 2 and 3 add the bias behind the k-loop)
		if (next_bias != nullptr)
			request_bias<G>(next_bias, wave, lane, bias_carry);

		if (MODE == 3)
		{
			float part[3][G::NTW];
#pragma unroll
			for (int o = 0; o < 3; o++)
#pragma unroll
				for (int n = 0; n < G::NTW; n++)
					part[o][n] = 0.0f;
#pragma unroll
			for (int i = 0; i < G::MT; i++)
			{
				const int ch = (mg * G::MT + i) * 16 + 4 * q4;
				const floatx4 bv = bias_now.b[i];
#pragma unroll
				for (int n = 0; n < G::NTW; n++)
				{
					const floatx4 v = acc[i][n] + bv;
#pragma unroll
					for (int j = 0; j < 4; j++)
					{
						const float t = static_cast<float>(static_cast<half_t>(tanhf(v[j]))); // fp16 like the two-plane kernel's plane
						const floatx4 w = *reinterpret_cast<const floatx4*>(wp2 + (ch + j) * 4);
						part[0][n] += t * w[0];
						part[1][n] += t * w[1];
						part[2][n] += t * w[2];
					}
				}
			}
#pragma unroll
			for (int o = 0; o < 3; o++)
#pragma unroll
				for (int n = 0; n < G::NTW; n++)
				{
					part[o][n] += __shfl_xor(part[o][n], 16);
					part[o][n] += __shfl_xor(part[o][n], 32);
				}
			for (int group = 0; group < G::CG; group++)
			{
				if (mg == group && q4 == 0)
				{
#pragma unroll
					for (int o = 0; o < 3; o++)
#pragma unroll
						for (int n = 0; n < G::NTW; n++)
							if (n < my_tiles)
							{
								float *dst = ppart + o * (G::NT * 16) + G::tile_position(wave, n, r);
								*dst = (group == 0) ? part[o][n] : (*dst + part[o][n]);
							}
				}
				__syncthreads();
			}
			return;
		}
		if (MODE == 2)
		{
			float part[G::NTW];
#pragma unroll
			for (int n = 0; n < G::NTW; n++)
				part[n] = 0.0f;
#pragma unroll
			for (int i = 0; i < G::MT; i++)
			{
				const int ch = (mg * G::MT + i) * 16 + 4 * q4;
				const floatx4 bv = bias_now.b[i];
				const floatx4 wv = *reinterpret_cast<const floatx4*>(wp2 + ch);
#pragma unroll
				for (int n = 0; n < G::NTW; n++)
				{
					const floatx4 v = acc[i][n] + bv;
					// the two-plane kernel rounds the ReLU output to fp16 before the 1x1 conv; keep that rounding so both agree
					part[n] += static_cast<float>(static_cast<half_t>(fmaxf(v[0], 0.0f))) * wv[0];
					part[n] += static_cast<float>(static_cast<half_t>(fmaxf(v[1], 0.0f))) * wv[1];
					part[n] += static_cast<float>(static_cast<half_t>(fmaxf(v[2], 0.0f))) * wv[2];
					part[n] += static_cast<float>(static_cast<half_t>(fmaxf(v[3], 0.0f))) * wv[3];
				}
			}
			int r_part = r;
			asm volatile("" : "+v"(r_part)); // (the NTW partial-sum addresses are this call's own, not kernel-wide constants reloaded from scratch one by one)
#pragma unroll
			for (int n = 0; n < G::NTW; n++)
			{
				float s = part[n];
				s += __shfl_xor(s, 16);
				s += __shfl_xor(s, 32);
				if (q4 == 0 && n < my_tiles)
					ppart[mg * (G::NT * 16) + G::tile_position(wave, n, r_part)] = s;
			}
			return;
		}

		// The cell masks and LDS addresses below depend on the lane only, not on the layer or the board: left alone the compiler computes all
		// 2 * MT * NTW of them once per kernel and keeps them in scratch — a reload (a global round trip, behind every store in flight) in
		// front of every LDS write of every layer.  An opaque copy of the lane's row makes them this layer's own few VALU instructions.
		int r_mask = r;
		asm volatile("" : "+v"(r_mask));
		// ReLU after the conversion on packed halves (rounding is monotonic: max(cvt(x), 0) == cvt(max(x, 0))): 4 vector instructions per tile
		// instead of 6.  Cells that are not on the board (spare column, overhang of the last tile: tail tiles only) are masked to zero by an AND.
		uint2 out[G::MT][G::NTW];
		bool valid[G::NTW];
#pragma unroll
		for (int n = 0; n < G::NTW; n++)
		{
			valid[n] = (n < my_tiles);
			if (!(G::COLT && n < G::COL_TILES))
			{ // (a column tile holds 16 cells of the board: nothing to mask)
				const int pos = G::S + G::tile_position(wave, n, r_mask);
				const int x = pos % G::S;
				const int y = pos / G::S - 1;
				valid[n] = valid[n] && (x < COLS) && (y < ROWS);
			}
		}
#pragma unroll
		for (int i = 0; i < G::MT; i++)
		{
#pragma unroll
			for (int n = 0; n < G::NTW; n++)
			{
				const floatx4 v = acc[i][n];
				half2 lo { static_cast<half_t>(v[0]), static_cast<half_t>(v[1]) }, hi { static_cast<half_t>(v[2]), static_cast<half_t>(v[3]) };
				const half2 zero2 { static_cast<half_t>(0.0f), static_cast<half_t>(0.0f) };
				lo = __builtin_elementwise_max(lo, zero2);
				hi = __builtin_elementwise_max(hi, zero2);
				uint2 packed { __builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi) };
				if (!(G::COLT && n < G::COL_TILES))
				{
					const uint32_t keep = valid[n] ? 0xFFFFFFFFu : 0u;
					packed.x &= keep;
					packed.y &= keep;
				}
				out[i][n] = packed;
				if (MODE == 1)
					my_skip[(i * G::NTW + n) * 64] = __builtin_bit_cast(half4, packed);
			}
		}
		if (MODE == 0 && skip_carry != nullptr)
		{ // (the accumulators are dead, the packed outputs hold half their registers: room for the next layer's residual input)
#pragma unroll
			for (int i = 0; i < G::MT; i++)
#pragma unroll
				for (int n = 0; n < G::NTW; n++)
					skip_carry->v[i][n] = __builtin_bit_cast(uint2, my_skip[(i * G::NTW + n) * 64]);
		}
		AGX_NN_MARK(4);
		lds_barrier(); // every wave has consumed the plane: it can be overwritten now
		AGX_NN_MARK(5);
		int r_write = r;
		asm volatile("" : "+v"(r_write));
		// (padded positions: the stores of a lane are two bases — its cell in the wave's first column tile and in its first tail tile — + immediates)
#pragma unroll
		for (int i = 0; i < G::MT; i++)
		{
			const int ch = (mg * G::MT + i) * 16 + 4 * q4;
#pragma unroll
			for (int n = 0; n < G::NTW; n++)
				if (n < my_tiles)
				{
					const int pos = G::S + G::tile_position(wave, n, r_write);
					*reinterpret_cast<uint2*>(plane + plane_offset<G>(pos + 1, ch / 8) + (ch % 8) * 2) = out[i][n];
				}
		}
		AGX_NN_MARK(4);
	}

	/*
	 * Input block: bit-unpack + 5x5 convolution (Cin = 32) + bias + ReLU.  `in5` is the padded input plane
	 * (stride S5, 64 bytes per position, chunk-swizzled by (index >> 2) & 3).
	 * RAW (ResnetPVraw, networks.cpp:107-129: Cin = 8, the low byte of the feature word): a position is 16 bytes (8 halves) and one
	 * MFMA's K = 32 carries FOUR horizontal taps x 8 channels — lane group q4 reads the cell q4 columns further right — so the 25 taps
	 * are 5 rows x 2 column groups (dx 0-3, dx 4 + three zero-weight taps) = 10 k-steps instead of 25; no chunk swizzle (16 consecutive
	 * positions x 16 bytes already cover all 64 banks once).
	 */
	template<int F, int ROWS, int COLS, bool INPLACE, bool RAW>
	__device__ __forceinline__ void conv5x5_input(const char *in5, char *dst, const half8 *__restrict__ wpk, const float *__restrict__ bias,
			half4 *skip, int wave, int lane)
	{
		typedef Geometry<F, ROWS, COLS> G;
		int r = lane & 15;
		if (INPLACE)
			asm volatile("" : "+v"(r)); // (per-lane cell indices and addresses are recomputed per board instead of living in scratch: conv3x3_inplace)
		const int q4 = lane >> 4;
		const int mg = G::channel_group(wave);   // output channels [mg * 16 * MT, (mg + 1) * 16 * MT)
		const int n0 = G::first_tile(wave);
		const int my_tiles = G::tile_count(wave);

		floatx4 acc[G::MT][G::NTW];
#pragma unroll
		for (int i = 0; i < G::MT; i++)
#pragma unroll
			for (int n = 0; n < G::NTW; n++)
				acc[i][n] = floatx4 { 0.0f, 0.0f, 0.0f, 0.0f };

		if constexpr (RAW)
		{
			int q0[G::NTW];
	#pragma unroll
			for (int n = 0; n < G::NTW; n++)
			{
				const int pos = G::S + G::tile_position(wave, n, r);
				q0[n] = (pos / G::S - 1 + 2) * G::S5 + (pos % G::S + 2) + q4; // (dy = 0, dx = 0) cell of this lane's position, q4 columns on
			}
			const half8 *wp = wpk + (mg * G::MT) * 64 + lane;
			half8 a_next[G::MT];
	#pragma unroll
			for (int i = 0; i < G::MT; i++)
				a_next[i] = wp[i * 64];
	#pragma unroll 1
			for (int t = 0; t < 10; t++)
			{ // t = 2 * dy + column group
				const int off = (t / 2 - 2) * G::S5 + (4 * (t % 2) - 2);
				half8 a[G::MT];
	#pragma unroll
				for (int i = 0; i < G::MT; i++)
					a[i] = a_next[i];
				const int tn = (t + 1 < 10) ? (t + 1) : 0;
	#pragma unroll
				for (int i = 0; i < G::MT; i++)
					a_next[i] = wp[(tn * G::MTILES + i) * 64];
	#pragma unroll
				for (int n = 0; n < G::NTW; n++)
					if (n < my_tiles)
					{
						int q = q0[n] + off;
						q = (q < 0) ? 0 : ((q >= G::NPOS5) ? (G::NPOS5 - 1) : q); // dummy positions and the zero-weight taps of column group 1 only
						const half8 b = *reinterpret_cast<const half8*>(in5 + q * 16);
	#pragma unroll
						for (int i = 0; i < G::MT; i++)
							acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b, acc[i][n], 0, 0, 0);
					}
			}
		}
		else if constexpr (G::S == 16 && G::MT == 1) // (one channel tile per wave: 2 x 5 weight fragments in flight; with two tiles they spill: the two-pass form below)
		{
			// Input-row stationary like conv3x3_mac_rows: for a column shift dx the fragment of padded input row j is read once and feeds the
			// five taps dy = -2 .. 2 (output rows j + 2 .. j - 2): 5 x (NTW + 4) fragment reads instead of 25 x NTW, and — the padded plane being
			// 2 cells wider than any shift on every side — no index clamping at all.
			const half8 *wl = wpk + __builtin_amdgcn_readfirstlane(mg * G::MT * 64); // + lane; tap (dy, dx), tile i at ((dy * 5 + dx) * MTILES + i) * 64
			half8 a_cur[5][G::MT], a_next[5][G::MT];
#pragma unroll
			for (int dyi = 0; dyi < 5; dyi++)
#pragma unroll
				for (int i = 0; i < G::MT; i++)
					a_cur[dyi][i] = wl[((dyi * 5 + 0) * G::MTILES + i) * 64 + lane];
#pragma unroll 1
			for (int dxi = 0; dxi < 5; dxi++)
			{ // one loop body for the five column shifts (unrolled five times the 2 x 10 weight fragments in flight spill)
				const int dxn = (dxi + 1 < 5) ? (dxi + 1) : 0;
#pragma unroll
				for (int dyi = 0; dyi < 5; dyi++)
#pragma unroll
					for (int i = 0; i < G::MT; i++)
						a_next[dyi][i] = wl[((dyi * 5 + dxn) * G::MTILES + i) * 64 + lane];
				// stored cell of this lane in padded row (n0 + j + 2): column r + (dxi - 2) + 2
				const int column = r + dxi;
				const char *column_ptr = in5 + (column * 4 + (q4 ^ ((column >> 2) & 3))) * 16; // (row stride a multiple of 16 positions: the swizzle is the column's)
				auto fragment = [&](int j) -> half8
				{
					const int jj = (j <= my_tiles + 1) ? j : 0; // rows past the wave's last output row + 2 are not needed (and would leave the plane)
					if constexpr (G::S5 % 16 == 0)
						return *reinterpret_cast<const half8*>(column_ptr + (n0 + jj + 2) * (G::S5 * 64));
					const int q = (n0 + jj + 2) * G::S5 + (r + dxi);
					return *reinterpret_cast<const half8*>(in5 + (q * 4 + (q4 ^ ((q >> 2) & 3))) * 16);
				};
				constexpr int AHEAD = AGX_NN_AHEAD;
				half8 b[AHEAD];
#pragma unroll
				for (int u = 0; u < AHEAD - 1; u++)
					b[u] = fragment(u - 2);
#pragma unroll
				for (int j = -2; j <= G::NTW + 1; j++)
				{
					const int jn = j + AHEAD - 1;
					if (jn <= G::NTW + 1)
						b[(jn + 2) % AHEAD] = fragment(jn);
#pragma unroll
					for (int dyi = 0; dyi < 5; dyi++)
					{
						const int o = j - (dyi - 2);
						if (o >= 0 && o < G::NTW && o < my_tiles)
						{
#pragma unroll
							for (int i = 0; i < G::MT; i++)
								acc[i][o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_cur[dyi][i], b[(j + 2) % AHEAD], acc[i][o], 0, 0, 0);
						}
					}
					__builtin_amdgcn_sched_barrier(0);
				}
#pragma unroll
				for (int dyi = 0; dyi < 5; dyi++)
#pragma unroll
					for (int i = 0; i < G::MT; i++)
						a_cur[dyi][i] = a_next[dyi][i];
			}
		}
		else if constexpr (G::S == 16 && G::MT == 2)
		{
			// 128 filters: the row-stationary loop above with the five vertical taps of a column shift taken in two passes — dy = -2, -1, 0
			// (input rows -2 .. NTW - 1 of the wave) and dy = +1, +2 (rows 1 .. NTW + 1) — so that only 3 * MT + 2 * MT weight fragments are
			// held at a time (the 2 x 5 * MT of the single pass spill): 5 x (2 NTW + 3) fragment reads instead of 25 x NTW, the tap-major
			// loop's LDS traffic halved (it is bound by it: one ds_read_b128 per MT = 2 MFMAs).
			const half8 *wl = wpk + __builtin_amdgcn_readfirstlane(mg * G::MT * 64); // + lane; tap (dy, dx), tile i at ((dy * 5 + dx) * MTILES + i) * 64
			half8 a_up[3][G::MT], a_down[2][G::MT];
#pragma unroll
			for (int dyi = 0; dyi < 3; dyi++)
#pragma unroll
				for (int i = 0; i < G::MT; i++)
					a_up[dyi][i] = wl[((dyi * 5 + 0) * G::MTILES + i) * 64 + lane];
#pragma unroll 1
			for (int dxi = 0; dxi < 5; dxi++)
			{
				const int dxn = (dxi + 1 < 5) ? (dxi + 1) : 0; // (the last turn requests the first shift's fragments again instead of branching)
				auto fragment = [&](int j) -> half8
				{ // stored cell of this lane in padded row (n0 + j + 2): column r + (dxi - 2) + 2
					const int jj = (j <= my_tiles + 1) ? j : 0; // rows past the wave's last output row + 2 are not needed (and would leave the plane)
					const int q = (n0 + jj + 2) * G::S5 + (r + dxi);
					return *reinterpret_cast<const half8*>(in5 + (q * 4 + (q4 ^ ((q >> 2) & 3))) * 16);
				};
				constexpr int AHEAD = AGX_NN_AHEAD;
				{ // pass 1: dy = -2, -1, 0 (dyi 0 .. 2) — while it runs the two lower taps' fragments of this shift arrive
#pragma unroll
					for (int dyi = 0; dyi < 2; dyi++)
#pragma unroll
						for (int i = 0; i < G::MT; i++)
							a_down[dyi][i] = wl[(((3 + dyi) * 5 + dxi) * G::MTILES + i) * 64 + lane];
					half8 b[AHEAD];
#pragma unroll
					for (int u = 0; u < AHEAD - 1; u++)
						b[u] = fragment(u - 2);
#pragma unroll
					for (int j = -2; j <= G::NTW - 1; j++)
					{
						const int jn = j + AHEAD - 1;
						if (jn <= G::NTW - 1)
							b[(jn + 2) % AHEAD] = fragment(jn);
#pragma unroll
						for (int dyi = 0; dyi < 3; dyi++)
						{
							const int o = j - (dyi - 2);
							if (o >= 0 && o < G::NTW && o < my_tiles)
							{
#pragma unroll
								for (int i = 0; i < G::MT; i++)
									acc[i][o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_up[dyi][i], b[(j + 2) % AHEAD], acc[i][o], 0, 0, 0);
							}
						}
						__builtin_amdgcn_sched_barrier(0);
					}
				}
				{ // pass 2: dy = +1, +2 (dyi 3, 4) — and the next shift's upper taps are requested
#pragma unroll
					for (int dyi = 0; dyi < 3; dyi++)
#pragma unroll
						for (int i = 0; i < G::MT; i++)
							a_up[dyi][i] = wl[((dyi * 5 + dxn) * G::MTILES + i) * 64 + lane];
					half8 b[AHEAD];
#pragma unroll
					for (int u = 0; u < AHEAD - 1; u++)
						b[u] = fragment(u + 1);
#pragma unroll
					for (int j = 1; j <= G::NTW + 1; j++)
					{
						const int jn = j + AHEAD - 1;
						if (jn <= G::NTW + 1)
							b[(jn - 1) % AHEAD] = fragment(jn);
#pragma unroll
						for (int dyi = 3; dyi < 5; dyi++)
						{
							const int o = j - (dyi - 2);
							if (o >= 0 && o < G::NTW && o < my_tiles)
							{
#pragma unroll
								for (int i = 0; i < G::MT; i++)
									acc[i][o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_down[dyi - 3][i], b[(j - 1) % AHEAD], acc[i][o], 0, 0, 0);
							}
						}
						__builtin_amdgcn_sched_barrier(0);
					}
				}
			}
		}
		else
		{
			// padded-plane index of the (dy = 0, dx = 0) input cell of this lane's position in every tile
			int q0[G::NTW];
	#pragma unroll
			for (int n = 0; n < G::NTW; n++)
			{
				const int pos = G::S + G::tile_position(wave, n, r);
				const int x = pos % G::S;
				const int y = pos / G::S - 1;
				q0[n] = (y + 2) * G::S5 + (x + 2);
			}

			const half8 *wp = wpk + (mg * G::MT) * 64 + lane;
			half8 a_next[G::MT]; // the next tap's weight fragments are requested one tap ahead (an L2 round trip is longer than a tap's MFMAs)
	#pragma unroll
			for (int i = 0; i < G::MT; i++)
				a_next[i] = wp[i * 64];
	#pragma unroll 1
			for (int t = 0; t < 25; t++)
			{
				const int off = (t / 5 - 2) * G::S5 + (t % 5 - 2);
				half8 a[G::MT];
	#pragma unroll
				for (int i = 0; i < G::MT; i++)
					a[i] = a_next[i];
				const int tn = (t + 1 < 25) ? (t + 1) : 0;
	#pragma unroll
				for (int i = 0; i < G::MT; i++)
					a_next[i] = wp[(tn * G::MTILES + i) * 64];
	#pragma unroll
				for (int n = 0; n < G::NTW; n++)
					if (n < my_tiles)
					{
						int q = q0[n] + off;
						q = (q < 0) ? 0 : ((q >= G::NPOS5) ? (G::NPOS5 - 1) : q); // only dummy (spare-column / overhang) positions can fall outside
						const half8 b = *reinterpret_cast<const half8*>(in5 + (q * 4 + (q4 ^ ((q >> 2) & 3))) * 16);
	#pragma unroll
						for (int i = 0; i < G::MT; i++)
							acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b, acc[i][n], 0, 0, 0);
					}
			}
		}
		if (INPLACE)
		{ // the padded input plane aliases the output plane: wait for every wave, clear the plane (zero borders), then write
			__syncthreads();
			const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
			for (int i = wave * 64 + lane; i < G::PLANE_BYTES / 16; i += G::THREADS)
				reinterpret_cast<uint4*>(dst)[i] = zero4;
			__syncthreads();
		}
		typedef __attribute__((address_space(1))) half4 global_half4; // (scalar base + opaque lane offset: conv3x3_inplace)
		int skip_lane = lane;
		asm volatile("" : "+v"(skip_lane));
		global_half4 *my_skip = (global_half4*) (skip + __builtin_amdgcn_readfirstlane(wave * G::MT * G::NTW * 64)) + skip_lane;
		if constexpr (G::S == 16 && !INPLACE)
		{ // row tiles, two planes: conv3x3's epilogue (RowEpilogue) with the bias added — one address per channel tile + n * 4096, the spare column's lanes
		  // sit the stores out (the plane's border stays zero), ReLU on packed halves
			static_assert(G::NT == ROWS, "a tile is a board row + the spare column");
			if (r < COLS)
			{
#pragma unroll
				for (int i = 0; i < G::MT; i++)
				{
					const int ch = (mg * G::MT + i) * 16 + 4 * q4;
					const floatx4 bv = *reinterpret_cast<const floatx4*>(bias + ch);
					char *out0 = dst + plane_offset<G>(1 + G::S + n0 * 16 + r, ch / 8) + (ch % 8) * 2;
#pragma unroll
					for (int n = 0; n < G::NTW; n++)
						if (n < my_tiles)
						{
							const floatx4 v = acc[i][n];
							half2 lo { static_cast<half_t>(v[0] + bv[0]), static_cast<half_t>(v[1] + bv[1]) }, hi { static_cast<half_t>(v[2] + bv[2]), static_cast<half_t>(v[3] + bv[3]) };
							const half2 zero2 { static_cast<half_t>(0.0f), static_cast<half_t>(0.0f) };
							lo = __builtin_elementwise_max(lo, zero2);
							hi = __builtin_elementwise_max(hi, zero2);
							*reinterpret_cast<uint2*>(out0 + n * 16 * G::POS_BYTES) = uint2 { __builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi) };
						}
				}
			}
			return;
		}
#pragma unroll
		for (int i = 0; i < G::MT; i++)
		{
			const int ch = (mg * G::MT + i) * 16 + 4 * q4;
			const floatx4 bv = *reinterpret_cast<const floatx4*>(bias + ch);
#pragma unroll
			for (int n = 0; n < G::NTW; n++)
				if (n < my_tiles)
				{
					const int pos = G::S + G::tile_position(wave, n, r);
					const int x = pos % G::S;
					const int y = pos / G::S - 1;
					const bool valid = (x < COLS) && (y < ROWS);
					char *ptr = dst + plane_offset<G>(pos + 1, ch / 8) + (ch % 8) * 2;
					const floatx4 v = acc[i][n] + bv;
					half4 o;
					o[0] = static_cast<half_t>(valid ? fmaxf(v[0], 0.0f) : 0.0f);
					o[1] = static_cast<half_t>(valid ? fmaxf(v[1], 0.0f) : 0.0f);
					o[2] = static_cast<half_t>(valid ? fmaxf(v[2], 0.0f) : 0.0f);
					o[3] = static_cast<half_t>(valid ? fmaxf(v[3], 0.0f) : 0.0f);
					*reinterpret_cast<half4*>(ptr) = o;
					if (INPLACE)
						my_skip[(i * G::NTW + n) * 64] = o; // residual input of the first block
				}
		}
	}

	__device__ __forceinline__ float block_reduce_max(float v, float *red, int tid)
	{
#pragma unroll
		for (int o = 32; o > 0; o >>= 1)
			v = fmaxf(v, __shfl_xor(v, o));
		__syncthreads();
		if ((tid & 63) == 0)
			red[tid >> 6] = v;
		__syncthreads();
		return fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
	}
	__device__ __forceinline__ float block_reduce_sum(float v, float *red, int tid)
	{
#pragma unroll
		for (int o = 32; o > 0; o >>= 1)
			v += __shfl_xor(v, o);
		__syncthreads();
		if ((tid & 63) == 0)
			red[tid >> 6] = v;
		__syncthreads();
		return ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
	}

	template<int F, int ROWS, int COLS, bool INPLACE, bool QHEAD, bool RAW = false>
#ifndef AGX_NN_WAVES_PER_EU
#define AGX_NN_WAVES_PER_EU 2 /* register budget of the tower: 2 = 256 registers per wave (its 8 waves fill a compute unit's register file); 3 = 168, which
                                 leaves room for one 168-register solver wave per SIMD next to a tower workgroup (the co-resident pairing experiment:
                                 profiles/r05_coresident_ab.txt, together with AGX_NN_SINGLE_PLANE=1 for the LDS) */
#endif
	// (a waves-per-EU request above what the kernel's LDS allows is ignored by hipcc; declaring a larger maximum block — 12 waves = 3 per SIMD — is what
	//  caps the allocation at 168 registers.  The kernel is always launched with 512 threads.)
	__global__ __launch_bounds__(256 * AGX_NN_WAVES_PER_EU) void nn_tower_kernel(NetParams p, const uint32_t *__restrict__ features, float *__restrict__ policy,
			float *__restrict__ value)
	{
		typedef Geometry<F, ROWS, COLS> G;
		constexpr int LDS_TOTAL = INPLACE ? G::LDS_BYTES_INPLACE : G::LDS_BYTES;
		static_assert(LDS_TOTAL <= 163840, "board does not fit in LDS");
		static_assert(G::NPOS5 * 64 <= G::PLANE_BYTES, "padded input plane must fit into an activation plane");
		__shared__ __attribute__((aligned(16))) char lds[LDS_TOTAL];
		char *plane_x = lds;
		char *plane_t = INPLACE ? lds : lds + G::PLANE_BYTES; // single-plane variant: every layer is computed in place
		float *vbuf = reinterpret_cast<float*>(lds + (INPLACE ? 1 : 2) * G::PLANE_BYTES); // [HW*4] + hid [D]: now the value head's conv1x1 weights as MFMA A fragments
		half8 *s_wv1f = reinterpret_cast<half8*>(vbuf);                    // [KC][64]: lane l = unit (l & 15) (4 real, 12 zero), inputs kc*32 + 8*(l >> 4) .. + 7
		static_assert(G::KC * 64 * 16 <= (G::HW * 4 + G::D) * 4, "value-head fragments must fit into the former vbuf + hid area");
		float *hid = vbuf + G::HW * 4;                                     // [D]
		float *red = hid + G::D;                                           // [8 + 256 + 8]: [0..7] wave partials, [8..8+D) value-head partials, [264..266] value logits
		float *s_wv1 = red + 8 + 256 + 8;                                      // [F][4] value-head 1x1 weights (kept in LDS, not in registers)
		float *s_wp2 = s_wv1 + F * 4;                                      // [F] policy-head 1x1 weights
		float *s_wq2 = s_wp2 + F;                                          // [F][4] action-values head 1x1 weights
		float *ppart = s_wq2 + F * 4;                                      // [CG][NT*16] policy partial logits (single-plane variant only)
		float *qpart = ppart + G::CG * G::NT * 16;                         // [3][NT*16] action-value logits (single-plane variant only)
		half4 *skip = INPLACE ? (p.skip + static_cast<size_t>(blockIdx.x) * G::SKIP_PER_WG) : nullptr;

		const int tid = threadIdx.x;
		const int wave = tid >> 6;
		const int lane = tid & 63;
		const int layer_halves8 = 9 * G::KC * G::MTILES * 64; // half8 elements per packed 3x3 layer

		const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
		for (int i = tid; i < G::PLANE_BYTES / 16; i += G::THREADS)
			reinterpret_cast<uint4*>(plane_x)[i] = zero4;
		for (int i = tid; i < G::KC * 64; i += G::THREADS)
		{ // A fragments of the F x 4 value-head conv1x1 (p.wv1 is [F][4] fp32)
			const int kc = i / 64, l = i % 64, unit = l & 15;
			half8 f;
#pragma unroll
			for (int j = 0; j < 8; j++)
				f[j] = static_cast<half_t>((unit < 4) ? p.wv1[(kc * 32 + 8 * (l >> 4) + j) * 4 + unit] : 0.0f);
			s_wv1f[i] = f;
		}
		for (int i = tid; i < F; i += G::THREADS)
			s_wp2[i] = p.wp2[i];
		if (QHEAD)
			for (int i = tid; i < F * 4; i += G::THREADS)
				s_wq2[i] = p.wq2[i];

#if AGX_NN_PAIR_BALANCE
		if (tid < 8)
			pair_progress()[tid] = 0;
#endif
		const int batch = (p.count_ptr != nullptr) ? min(*p.count_ptr, p.batch) : p.batch;
#ifdef AGX_NN_PROFILE
		NnStamp stamp(wave, lane);
#endif
		static_assert(G::HW <= G::THREADS, "one feature word per thread");
		uint32_t next_word = 0; // this thread's feature word of the board about to be staged
		if (static_cast<int>(blockIdx.x) < batch && tid < G::HW)
			next_word = features[static_cast<size_t>((p.slot_list != nullptr) ? p.slot_list[blockIdx.x] : static_cast<int>(blockIdx.x)) * G::HW + tid];
		for (int bi = blockIdx.x; bi < batch; bi += gridDim.x)
		{
			const int b = (p.slot_list != nullptr) ? p.slot_list[bi] : bi;
			// ---- stage the bit-unpacked input into the padded plane (aliases plane_t) ----
			__syncthreads();
			for (int i = tid; i < G::NPOS5 * (RAW ? 1 : 4); i += G::THREADS)
				reinterpret_cast<uint4*>(plane_t)[i] = zero4;
			__syncthreads();
			for (int c = tid; c < G::HW; c += G::THREADS)
			{
				const uint32_t word = next_word;
				const int q = (c / COLS + 2) * G::S5 + (c % COLS + 2);
#pragma unroll
				for (int k = 0; k < (RAW ? 1 : 4); k++)
				{
					const uint32_t bits = (word >> (8 * k)) & 255u;
					uint4 v;
					v.x = ((bits & 1u) ? 0x3C00u : 0u) | ((bits & 2u) ? 0x3C000000u : 0u);
					v.y = ((bits & 4u) ? 0x3C00u : 0u) | ((bits & 8u) ? 0x3C000000u : 0u);
					v.z = ((bits & 16u) ? 0x3C00u : 0u) | ((bits & 32u) ? 0x3C000000u : 0u);
					v.w = ((bits & 64u) ? 0x3C00u : 0u) | ((bits & 128u) ? 0x3C000000u : 0u);
					if (RAW) // ml::unpackInput into 8 channels (AGNetwork.cpp:249-258): the low byte of the word, 16 bytes per position
						*reinterpret_cast<uint4*>(plane_t + q * 16) = v;
					else
						*reinterpret_cast<uint4*>(plane_t + (q * 4 + (k ^ ((q >> 2) & 3))) * 16) = v;
				}
			}
			__syncthreads();
			AGX_NN_MARK(0);
			conv5x5_input<F, ROWS, COLS, INPLACE, RAW>(plane_t, plane_x, p.w_in, p.bias, skip, wave, lane);
			BiasCarry<G::MT> bias_carry; // the first tower layer's bias values (with no block: the policy conv's), then each layer's successor's
			request_bias<G>(p.bias + F, wave, lane, bias_carry);
			__syncthreads();
			AGX_NN_MARK(1);
			if (!INPLACE)
			{
				for (int i = tid; i < G::PLANE_BYTES / 16; i += G::THREADS)
					reinterpret_cast<uint4*>(plane_t)[i] = zero4; // restore the zero border of plane_t
				__syncthreads();
			}

			AGX_NN_MARK(9);
			// ---- residual tower ----
			constexpr bool CARRY = INPLACE ? (G::COLT && !RAW) : G::S == 16;
			WeightCarry<G> carry_store;
			WeightCarry<G> *carry = CARRY ? &carry_store : nullptr;
			if constexpr (CARRY)
				request_first_stage<F, ROWS, COLS>(p.w_tower, wave, lane, carry_store);
			for (int blk = 0; blk < p.blocks; blk++)
			{
				// (bias_carry: every layer requests the next layer's bias values behind its k-loop; behind the last block follows the policy conv)
				if (INPLACE)
				{
					SkipCarry<G> skip_carry; // the block's residual input: requested by its first layer, consumed by its second
					if constexpr (CARRY)
						carry_store.next = p.w_tower + (2 * blk + 1) * layer_halves8;
					conv3x3_inplace<F, ROWS, COLS, 0>(plane_x, p.w_tower + (2 * blk) * layer_halves8, bias_carry, p.bias + (2 + 2 * blk) * F, skip, nullptr, nullptr, wave,
							lane AGX_NN_STAMP_ARG, &skip_carry, carry);
					lds_barrier();
					if constexpr (CARRY)
						carry_store.next = p.w_tower + (2 * blk + 2) * layer_halves8; // (behind the last block: the policy conv)
					conv3x3_inplace<F, ROWS, COLS, 1>(plane_x, p.w_tower + (2 * blk + 1) * layer_halves8, bias_carry, p.bias + (3 + 2 * blk) * F, skip, nullptr, nullptr, wave,
							lane AGX_NN_STAMP_ARG, &skip_carry, carry);
					lds_barrier();
				}
				else
				{
					if constexpr (CARRY)
						carry_store.next = p.w_tower + (2 * blk + 1) * layer_halves8;
					conv3x3<F, ROWS, COLS, false>(plane_x, plane_t, p.w_tower + (2 * blk) * layer_halves8, p.bias + (1 + 2 * blk) * F, wave, lane AGX_NN_STAMP_ARG, carry, &bias_carry,
							p.bias + (2 + 2 * blk) * F);
					__syncthreads();
					AGX_NN_MARK(5);
					if constexpr (CARRY)
						carry_store.next = p.w_tower + (2 * blk + 2) * layer_halves8; // (behind the last block: the policy conv)
					conv3x3<F, ROWS, COLS, true>(plane_t, plane_x, p.w_tower + (2 * blk + 1) * layer_halves8, p.bias + (2 + 2 * blk) * F, wave, lane AGX_NN_STAMP_ARG, carry, &bias_carry,
							p.bias + (3 + 2 * blk) * F);
					__syncthreads();
					AGX_NN_MARK(5);
				}
			}

			// ---- value head, stage 1: conv1x1 F->4 + ReLU (NHWC flatten order).  The dense layers behind it run for ALL boards of the launch in
			//      value_head_kernel: inside this kernel every board streamed the 0.46 MB of dense weights through one dependent chain per
			//      thread (measured: 9 % of a board's time), batched they are one small GEMM on the matrix cores ----
			{ // conv1x1 F -> 4 as one 16 x 16 output tile per 16 positions (4 of the 16 "channels" are real): K = F in KC steps; the 15 (NT)
			  // position tiles are dealt round-robin to the 8 waves.  (As a per-thread dot product this stage took 4 % of a board's time with
			  // more than half of the threads idle.)
				const int r = lane & 15, q4 = lane >> 4;
				for (int n = wave; n < G::NT; n += G::THREADS / 64)
				{
					floatx4 v { p.bv1[0], p.bv1[1], p.bv1[2], p.bv1[3] };
					const int index0 = 1 + G::S + n * 16 + r;
#pragma unroll
					for (int kc = 0; kc < G::KC; kc++)
					{
						const half8 bfrag = *reinterpret_cast<const half8*>(plane_x + plane_offset<G>(index0, kc * 4 + q4));
						v = __builtin_amdgcn_mfma_f32_16x16x32_f16(s_wv1f[kc * 64 + lane], bfrag, v, 0, 0, 0);
					}
					// lanes with q4 == 0 hold outputs 0 .. 3 of position r of tile n (the bias of the other, unused rows is irrelevant)
					const int pos = G::S + n * 16 + r;
					const int x = pos % G::S, y = pos / G::S - 1;
					if (q4 == 0 && x < COLS && y < ROWS)
					{
						half4 o;
						o[0] = static_cast<half_t>(fmaxf(v[0], 0.0f));
						o[1] = static_cast<half_t>(fmaxf(v[1], 0.0f));
						o[2] = static_cast<half_t>(fmaxf(v[2], 0.0f));
						o[3] = static_cast<half_t>(fmaxf(v[3], 0.0f));
						*reinterpret_cast<half4*>(p.vhead_x + static_cast<size_t>(bi) * G::KPAD + (y * COLS + x) * 4) = o;
					}
				}
			}
			AGX_NN_MARK(6);
			// ---- policy head: conv3x3 + ReLU into plane_t ----
			if (INPLACE)
			{
				if constexpr (CARRY) // (nothing follows: its own first stage again)
					carry_store.next = p.w_tower + (2 * p.blocks) * layer_halves8;
				conv3x3_inplace<F, ROWS, COLS, 2>(plane_x, p.w_tower + (2 * p.blocks) * layer_halves8, bias_carry, QHEAD ? p.bias + (2 + 2 * p.blocks) * F : nullptr, nullptr, s_wp2,
						ppart, wave, lane AGX_NN_STAMP_ARG, nullptr, carry);
			}
			else
			{
				if constexpr (CARRY)
					carry_store.next = p.w_tower + (2 * p.blocks) * layer_halves8; // nothing follows: its last turn requests its own first stage again
				conv3x3<F, ROWS, COLS, false>(plane_x, plane_t, p.w_tower + (2 * p.blocks) * layer_halves8, p.bias + (1 + 2 * p.blocks) * F, wave, lane AGX_NN_STAMP_ARG, carry,
						&bias_carry, p.bias + (1 + 2 * p.blocks) * F); // (nothing follows: its own values again)
			}
			__syncthreads();
			AGX_NN_MARK(7);
			// the next board's input: requested here, consumed by the staging loop at the top — the round trip rides under the heads
			if (bi + static_cast<int>(gridDim.x) < batch && tid < G::HW)
			{
				const int nb = bi + static_cast<int>(gridDim.x);
				next_word = features[static_cast<size_t>((p.slot_list != nullptr) ? p.slot_list[nb] : nb) * G::HW + tid];
			}

			// ---- policy head: conv1x1 F->1 + bias, softmax over the board ----
			{
				float logit = -3.0e38f;
				const int c = tid;
				if (c < G::HW)
				{
					float s = p.bp2;
					if (INPLACE)
					{
						const int idx = (c / COLS) * G::S + (c % COLS);
						constexpr int PS = G::NT * 16; // the channel groups' partial sums, added in a fixed order
						if constexpr (G::CG == 8)
							s += ((ppart[idx] + ppart[PS + idx]) + (ppart[2 * PS + idx] + ppart[3 * PS + idx]))
									+ ((ppart[4 * PS + idx] + ppart[5 * PS + idx]) + (ppart[6 * PS + idx] + ppart[7 * PS + idx]));
						else if constexpr (G::CG == 4)
							s += (ppart[idx] + ppart[PS + idx]) + (ppart[2 * PS + idx] + ppart[3 * PS + idx]);
						else
							s += ppart[idx] + ppart[PS + idx];
					}
					else
					{
						const int index = 1 + G::S + (c / COLS) * G::S + (c % COLS);
						for (int k = 0; k < G::CH; k++)
						{
							const half8 tv = *reinterpret_cast<const half8*>(plane_t + plane_offset<G>(index, k));
#pragma unroll
							for (int j = 0; j < 8; j++)
								s += static_cast<float>(tv[j]) * s_wp2[k * 8 + j];
						}
					}
					logit = s;
				}
				static_assert(G::HW <= G::THREADS, "policy softmax assumes one cell per thread");
				const float m = block_reduce_max(logit, red, tid);
				const float e = (c < G::HW) ? __expf(logit - m) : 0.0f;
				const float sum = block_reduce_sum(e, red, tid);
				if (c < G::HW)
					policy[static_cast<size_t>(b) * G::HW + c] = e / sum;
			}

			// ---- action-values head (blocks.cpp:119-127): conv3x3 + tanh, conv1x1 F -> 3 + bias, softmax over the 3 per cell ----
			if (QHEAD)
			{
				const half8 *wq1 = p.w_tower + (2 * p.blocks + 1) * layer_halves8;
				const float *bq1 = p.bias + (2 + 2 * p.blocks) * F;
				__syncthreads(); // the policy head is done with plane_t / the partial-sum buffers
				if (INPLACE)
					conv3x3_inplace<F, ROWS, COLS, 3>(plane_x, wq1, bias_carry, nullptr, nullptr, s_wq2, qpart, wave, lane AGX_NN_STAMP_ARG);
				else
				{
					conv3x3<F, ROWS, COLS, false, true>(plane_x, plane_t, wq1, bq1, wave, lane AGX_NN_STAMP_ARG);
					__syncthreads();
				}
				const int c = tid;
				if (c < G::HW)
				{
					float z0 = p.bq2[0], z1 = p.bq2[1], z2 = p.bq2[2];
					if (INPLACE)
					{
						const int idx = (c / COLS) * G::S + (c % COLS);
						z0 += qpart[idx];
						z1 += qpart[G::NT * 16 + idx];
						z2 += qpart[2 * G::NT * 16 + idx];
					}
					else
					{
						const int index = 1 + G::S + (c / COLS) * G::S + (c % COLS);
						for (int k = 0; k < G::CH; k++)
						{
							const half8 tv = *reinterpret_cast<const half8*>(plane_t + plane_offset<G>(index, k));
#pragma unroll
							for (int j = 0; j < 8; j++)
							{
								const float t = static_cast<float>(tv[j]);
								const floatx4 w = *reinterpret_cast<const floatx4*>(s_wq2 + (k * 8 + j) * 4);
								z0 += t * w[0];
								z1 += t * w[1];
								z2 += t * w[2];
							}
						}
					}
					const float m = fmaxf(z0, fmaxf(z1, z2));
					const float e0 = __expf(z0 - m), e1 = __expf(z1 - m), e2 = __expf(z2 - m);
					const float inv = 1.0f / (e0 + e1 + e2);
					float *out = p.q + (static_cast<size_t>(b) * G::HW + c) * 2;
					out[0] = e0 * inv; // win
					out[1] = e1 * inv; // draw
				}
			}
			AGX_NN_MARK(8);
		}
	}

	/*
	 * Value head behind the tower, for every board of a launch: hidden = ReLU(W2^T x + b2) (4 HW -> D), out = softmax(W3^T hidden + b3)
	 * (createValueHead, blocks.cpp:108-118).  One workgroup = 16 boards (one MFMA position tile) x all D hidden units: wave w owns the
	 * 16-unit tiles w * D/64 .. and walks K in 32-input steps (weights pre-packed in A-fragment order, the boards' inputs are rows of
	 * KPAD halves).  3.1 GFLOP for a whole self-play batch — microseconds.
	 */
	template<int KPAD, int D>
	__global__ __launch_bounds__(256) void value_head_kernel(const half_t *__restrict__ x, const half8 *__restrict__ w2, const float *__restrict__ b2,
			const float *__restrict__ w3, float b30, float b31, float b32, const int *__restrict__ slot_list, const int *__restrict__ count_ptr, int batch_cap,
			float *__restrict__ value)
	{
		constexpr int MTW = D / 64; // 16-unit tiles per wave
		__shared__ float hid[16][D + 1];
		const int batch = (count_ptr != nullptr) ? min(*count_ptr, batch_cap) : batch_cap;
		const int b0 = blockIdx.x * 16;
		if (b0 >= batch)
			return;
		const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, q4 = lane >> 4;
		const int board = min(b0 + r, batch - 1); // the tail tile repeats the last board (never stored)
		floatx4 acc[MTW];
#pragma unroll
		for (int i = 0; i < MTW; i++)
			acc[i] = floatx4 { 0.0f, 0.0f, 0.0f, 0.0f };
		const half8 *xb = reinterpret_cast<const half8*>(x + static_cast<size_t>(board) * KPAD) + q4;
		const half8 *wp = w2 + (wave * MTW) * 64 + lane;
#pragma unroll 4
		for (int kc = 0; kc < KPAD / 32; kc++)
		{
			const half8 bfrag = xb[kc * 4];
#pragma unroll
			for (int i = 0; i < MTW; i++)
				acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wp[(kc * (D / 16) + i) * 64], bfrag, acc[i], 0, 0, 0);
		}
		// lane holds hidden units 4 * q4 .. + 3 of tile i for board r
#pragma unroll
		for (int i = 0; i < MTW; i++)
		{
			const int u = (wave * MTW + i) * 16 + 4 * q4;
#pragma unroll
			for (int e = 0; e < 4; e++)
				hid[r][u + e] = fmaxf(acc[i][e] + b2[u + e], 0.0f);
		}
		__syncthreads();
		// 16 boards x 3 outputs: threads 0 .. 47 each reduce one (board, output) pair in a fixed order
		__shared__ float logits[16][3];
		if (tid < 48)
		{
			const int bb = tid / 3, o = tid % 3;
			float sacc = 0.0f;
			for (int j = 0; j < D; j++)
				sacc += hid[bb][j] * w3[j * 3 + o];
			logits[bb][o] = sacc + ((o == 0) ? b30 : ((o == 1) ? b31 : b32));
		}
		__syncthreads();
		if (tid < 16 && b0 + tid < batch)
		{
			const int bi = b0 + tid;
			const int slot = (slot_list != nullptr) ? slot_list[bi] : bi;
			const float z0 = logits[tid][0], z1 = logits[tid][1], z2 = logits[tid][2];
			const float m = fmaxf(z0, fmaxf(z1, z2));
			const float e0 = __expf(z0 - m), e1 = __expf(z1 - m), e2 = __expf(z2 - m);
			const float inv = 1.0f / (e0 + e1 + e2);
			value[static_cast<size_t>(slot) * 3 + 0] = e0 * inv;
			value[static_cast<size_t>(slot) * 3 + 1] = e1 * inv;
			value[static_cast<size_t>(slot) * 3 + 2] = e2 * inv;
		}
	}

	/*
	 * Host side: pack canonical [kh][kw][cin][cout] fp32 weights into MFMA A-fragment order
	 * [tap][kc][mtile][lane][8]: lane l holds out-channel mtile*16 + (l & 15), in-channels kc*32 + 8*(l >> 4) + j.
	 */
	void pack_conv(const float *w, int taps, int cin, int cout, std::vector<half_t> &dst)
	{
		const int kcs = cin / 32;
		const int mtiles = cout / 16;
		const size_t base = dst.size();
		dst.resize(base + static_cast<size_t>(taps) * kcs * mtiles * 512);
		half_t *out = dst.data() + base;
		for (int t = 0; t < taps; t++)
			for (int kc = 0; kc < kcs; kc++)
				for (int mt = 0; mt < mtiles; mt++)
					for (int lane = 0; lane < 64; lane++)
						for (int j = 0; j < 8; j++)
						{
							const int oc = mt * 16 + (lane & 15);
							const int ic = kc * 32 + 8 * (lane >> 4) + j;
							const float v = w[(static_cast<size_t>(t) * cin + ic) * cout + oc];
							out[(((static_cast<size_t>(t) * kcs + kc) * mtiles + mt) * 64 + lane) * 8 + j] = static_cast<half_t>(v);
						}
	}
}

namespace
{
	/*
	 * The same fragments in the order the row-stationary k-loop consumes them (conv3x3_mac_rows): [kc][dx][channel group][dy][tile][lane][8],
	 * so the 3 * MT fragments a wave needs for one stage are consecutive.  3x3 kernels only; MT = cout / (16 * groups) tiles per channel group.
	 */
	void pack_conv_rows(const float *w, int cin, int cout, std::vector<half_t> &dst, bool by_row_shift = false, int groups = 4)
	{ // by_row_shift: the column-tile loop (conv3x3_mac_cols) — stage = (chunk, dy), inside it dx: the same order with the roles exchanged
	  // groups: channel groups of the kernel's geometry (Geometry::CG)
		const int kcs = cin / 32, mt_per_group = cout / (16 * groups);
		const size_t base = dst.size();
		dst.resize(base + static_cast<size_t>(9) * kcs * (cout / 16) * 512);
		half_t *out = dst.data() + base;
		size_t frag = 0;
		for (int kc = 0; kc < kcs; kc++)
			for (int dx = 0; dx < 3; dx++)
				for (int mg = 0; mg < groups; mg++)
					for (int dy = 0; dy < 3; dy++)
						for (int i = 0; i < mt_per_group; i++, frag++)
							for (int lane = 0; lane < 64; lane++)
								for (int j = 0; j < 8; j++)
								{
									const int t = by_row_shift ? (dx * 3 + dy) : (dy * 3 + dx); // (loop variable "dx" is the stage's shift)
									const int oc = (mg * mt_per_group + i) * 16 + (lane & 15);
									const int ic = kc * 32 + 8 * (lane >> 4) + j;
									out[(frag * 64 + lane) * 8 + j] = static_cast<half_t>(w[(static_cast<size_t>(t) * cin + ic) * cout + oc]);
								}
	}
}

namespace
{
	/* conv5x5 of an 8-channel input (ResnetPVraw) for conv5x5_input<.., RAW>: k-step t = 2 * dy + g holds the four horizontal taps
	 * dx = 4 g + (lane >> 4) x 8 channels (taps beyond dx = 4 are zero): [10][cout / 16][lane][8] */
	void pack_conv5x5_raw(const float *w, int cout, std::vector<half_t> &dst)
	{
		const int mtiles = cout / 16;
		dst.assign(static_cast<size_t>(10) * mtiles * 512, static_cast<half_t>(0.0f));
		for (int t = 0; t < 10; t++)
			for (int mt = 0; mt < mtiles; mt++)
				for (int lane = 0; lane < 64; lane++)
				{
					const int dy = t / 2, dx = 4 * (t % 2) + (lane >> 4);
					if (dx >= 5)
						continue;
					for (int j = 0; j < 8; j++)
						dst[((static_cast<size_t>(t) * mtiles + mt) * 64 + lane) * 8 + j] = static_cast<half_t>(w[(static_cast<size_t>(dy * 5 + dx) * 8 + j) * cout + mt * 16 + (lane & 15)]);
				}
	}
}

struct AgxNet
{
		AgxNetDesc desc;
		bool loaded = false;
		void *d_w_in = nullptr;
		void *d_w_tower = nullptr;
		void *d_bias = nullptr;
		void *d_wp2 = nullptr;
		void *d_wv1 = nullptr;
		void *d_wv2 = nullptr;
		void *d_bv2 = nullptr;
		void *d_wv3 = nullptr;
		void *d_wq2 = nullptr;  // action-values head 1x1 weights [F][4] (padded), only with desc.action_values
		float bq2[3] = { 0, 0, 0 };
		// single-plane variant: residual scratch, one slice per workgroup of the persistent grid.  Launches on DIFFERENT streams may run
		// concurrently (pool slices driven from several streams share one network), so every stream gets its own scratch; launches on
		// one stream are ordered and share theirs.
		std::mutex skip_mutex;
		struct StreamScratch
		{
				hipStream_t stream;
				void *skip;   // single-plane variant: residual scratch
				void *vhead;  // value-head inputs of one launch: [boards][KPAD] halves
				int vhead_boards;
		};
		std::vector<StreamScratch> scratch_by_stream;
		size_t skip_bytes = 0;
		bool inplace = false;
		float bp2 = 0.0f;
		float bv1[4] = { 0, 0, 0, 0 };
		float bv3[3] = { 0, 0, 0 };
		int num_cus = 256;
		int launch_width = 0; // 0 = every CU
};

namespace
{
	bool is_supported(const AgxNetDesc &d)
	{
		return ((d.rows == 15 && d.cols == 15) || (d.rows == 20 && d.cols == 20)) && (d.filters == 64 || d.filters == 128) && d.blocks >= 0
				&& (d.in_channels == 32 || (d.in_channels == 8 && d.action_values == 0)) // 8: ResnetPVraw (networks.cpp:107-129); the raw PVQ variant is off the path
				&& d.value_hidden == ((2 * d.filters < 256) ? 2 * d.filters : 256) && (d.action_values == 0 || d.action_values == 1);
	}
	void free_net_buffers(AgxNet *net)
	{
		void **ptrs[] = { &net->d_w_in, &net->d_w_tower, &net->d_bias, &net->d_wp2, &net->d_wv1, &net->d_wv2, &net->d_bv2, &net->d_wv3, &net->d_wq2 };
		for (void **p : ptrs)
		{
			if (*p != nullptr)
				(void) hipFree(*p);
			*p = nullptr;
		}
		std::lock_guard<std::mutex> lock(net->skip_mutex);
		for (auto &slice : net->scratch_by_stream)
		{
			if (slice.skip != nullptr)
				(void) hipFree(slice.skip);
			if (slice.vhead != nullptr)
				(void) hipFree(slice.vhead);
		}
		net->scratch_by_stream.clear();
	}
	template<typename T>
	int upload(void **dst, const std::vector<T> &src)
	{
		AGX_HIP_CHECK(hipMalloc(dst, src.size() * sizeof(T)));
		AGX_HIP_CHECK(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
		return AGX_OK;
	}
}

extern "C" {

size_t agx_net_blob_floats(const AgxNetDesc *d)
{
	if (d == nullptr)
		return 0;
	const size_t F = d->filters, C = d->in_channels, HW = static_cast<size_t>(d->rows) * d->cols, D = d->value_hidden;
	size_t n = 25 * C * F + F;
	n += static_cast<size_t>(d->blocks) * 2 * (9 * F * F + F);
	n += 9 * F * F + F + F + 1;
	n += F * 4 + 4 + HW * 4 * D + D + D * 3 + 3;
	if (d->action_values)
		n += 9 * F * F + F + F * 3 + 3;
	return n;
}

int agx_net_create(const AgxNetDesc *desc, AgxNet **out)
{
	AGX_REQUIRE(desc != nullptr && out != nullptr, AGX_ERR_INVALID, "agx_net_create: null argument");
	AGX_REQUIRE(is_supported(*desc), AGX_ERR_UNSUPPORTED,
			"agx_net_create: unsupported network %dx%d blocks=%d filters=%d cin=%d hidden=%d (supported: 15x15 / 20x20, F in {64,128}, cin 32, or cin 8 without the action-values head)", desc->rows,
			desc->cols, desc->blocks, desc->filters, desc->in_channels, desc->value_hidden);
	AgxNet *net = new AgxNet();
	net->desc = *desc;
	int dev = 0;
	hipDeviceProp_t prop;
	if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
		net->num_cus = prop.multiProcessorCount;
	*out = net;
	return AGX_OK;
}

int agx_net_load_weights(AgxNet *net, const float *h_blob, size_t n_floats)
{
	AGX_REQUIRE(net != nullptr && h_blob != nullptr, AGX_ERR_INVALID, "agx_net_load_weights: null argument");
	AGX_REQUIRE(n_floats == agx_net_blob_floats(&net->desc), AGX_ERR_INVALID, "agx_net_load_weights: blob has %zu floats, expected %zu", n_floats,
			agx_net_blob_floats(&net->desc));
	free_net_buffers(net);
	net->loaded = false;

	const int F = net->desc.filters, C = net->desc.in_channels, HW = net->desc.rows * net->desc.cols, D = net->desc.value_hidden;
	const int blocks = net->desc.blocks;
	const float *ptr = h_blob;

	std::vector<half_t> w_in, w_tower, wv2;
	std::vector<float> bias, wp2, wv1, bv2, wv3;

	if (C == 8)
		pack_conv5x5_raw(ptr, F, w_in);
	else
		pack_conv(ptr, 25, C, F, w_in);
	ptr += 25 * C * F;
	bias.insert(bias.end(), ptr, ptr + F);
	ptr += F;
	// 15-column boards (row stride 16 = one MFMA tile) run the row-stationary k-loop, which reads the fragments in its own order
	const bool row_order = (net->desc.cols + 1 == 16);
	// 20x20 boards with 128 filters run the column-tile k-loop (Geometry::COLT)
	const bool column_order = (net->desc.rows == 20 && net->desc.cols == 20 && F == 128);
	auto pack3x3 = [&](const float *w)
	{
		if (row_order)
			pack_conv_rows(w, F, F, w_tower, false, (F == 128) ? Geometry<128, 15, 15>::CG : Geometry<64, 15, 15>::CG);
		else if (column_order)
			pack_conv_rows(w, F, F, w_tower, true, Geometry<128, 20, 20>::CG);
		else
			pack_conv(w, 9, F, F, w_tower);
	};
	for (int l = 0; l < 2 * blocks + 1; l++)
	{
		pack3x3(ptr);
		ptr += 9 * F * F;
		bias.insert(bias.end(), ptr, ptr + F);
		ptr += F;
	}
	wp2.assign(ptr, ptr + F);
	ptr += F;
	net->bp2 = *ptr++;
	wv1.assign(ptr, ptr + F * 4);
	ptr += F * 4;
	for (int i = 0; i < 4; i++)
		net->bv1[i] = *ptr++;
	{ // dense 4 HW -> D in MFMA A-fragment order [k-step][16-unit tile][lane][8] (value_head_kernel), zero beyond the last input
		const int kpad = (HW * 4 + 31) / 32 * 32, mtiles = D / 16;
		wv2.assign(static_cast<size_t>(kpad) * D, static_cast<half_t>(0.0f));
		for (int kc = 0; kc < kpad / 32; kc++)
			for (int mt = 0; mt < mtiles; mt++)
				for (int lane = 0; lane < 64; lane++)
					for (int j = 0; j < 8; j++)
					{
						const int unit = mt * 16 + (lane & 15), k = kc * 32 + 8 * (lane >> 4) + j;
						if (k < HW * 4)
							wv2[((static_cast<size_t>(kc) * mtiles + mt) * 64 + lane) * 8 + j] = static_cast<half_t>(ptr[static_cast<size_t>(k) * D + unit]);
					}
	}
	ptr += static_cast<size_t>(HW) * 4 * D;
	bv2.assign(ptr, ptr + D);
	ptr += D;
	wv3.assign(ptr, ptr + D * 3);
	ptr += D * 3;
	for (int i = 0; i < 3; i++)
		net->bv3[i] = *ptr++;
	std::vector<float> wq2;
	if (net->desc.action_values)
	{ // createActionValuesHead (blocks.cpp:119-127): the 3x3 conv joins the packed tower layers, the 1x1 conv is kept in fp32
		pack3x3(ptr);
		ptr += 9 * F * F;
		bias.insert(bias.end(), ptr, ptr + F);
		ptr += F;
		wq2.assign(static_cast<size_t>(F) * 4, 0.0f);
		for (int c = 0; c < F; c++)
			for (int o = 0; o < 3; o++)
				wq2[c * 4 + o] = ptr[c * 3 + o];
		ptr += F * 3;
		for (int i = 0; i < 3; i++)
			net->bq2[i] = *ptr++;
	}

	// 20x20 boards use the single-plane kernel (two planes do not fit into LDS); AGX_NN_SINGLE_PLANE=1 selects it for 15x15 too
	const char *force = getenv("AGX_NN_SINGLE_PLANE");
	net->inplace = (net->desc.rows == 20) || (force != nullptr && force[0] == '1');
	if (net->inplace)
	{
		const size_t per_wg = (net->desc.rows == 20) ? ((F == 128) ? Geometry<128, 20, 20>::SKIP_PER_WG : Geometry<64, 20, 20>::SKIP_PER_WG)
				: ((F == 128) ? Geometry<128, 15, 15>::SKIP_PER_WG : Geometry<64, 15, 15>::SKIP_PER_WG);
		net->skip_bytes = per_wg * 8 * static_cast<size_t>(net->num_cus);
	}
	int status = AGX_OK;
	if ((status = upload(&net->d_w_in, w_in)) != AGX_OK || (status = upload(&net->d_w_tower, w_tower)) != AGX_OK
			|| (status = upload(&net->d_bias, bias)) != AGX_OK || (status = upload(&net->d_wp2, wp2)) != AGX_OK
			|| (status = upload(&net->d_wv1, wv1)) != AGX_OK || (status = upload(&net->d_wv2, wv2)) != AGX_OK
			|| (status = upload(&net->d_bv2, bv2)) != AGX_OK || (status = upload(&net->d_wv3, wv3)) != AGX_OK
			|| (!wq2.empty() && (status = upload(&net->d_wq2, wq2)) != AGX_OK))
	{
		free_net_buffers(net);
		return status;
	}
	net->loaded = true;
	return AGX_OK;
}

static int launch_forward(AgxNet *net, const uint32_t *d_features, const int *d_slot_list, const int *d_count, int batch, float *d_policy, float *d_value,
		float *d_action_values, void *stream)
{
	AGX_REQUIRE(net != nullptr, AGX_ERR_INVALID, "agx_nn_forward: null network");
	AGX_REQUIRE(net->loaded, AGX_ERR_STATE, "agx_nn_forward: weights not loaded");
	AGX_REQUIRE(batch >= 0, AGX_ERR_INVALID, "agx_nn_forward: negative batch");
	if (batch == 0)
		return AGX_OK;
	AGX_REQUIRE(d_features != nullptr && d_policy != nullptr && d_value != nullptr, AGX_ERR_INVALID, "agx_nn_forward: null buffer");

	NetParams p;
	p.w_in = static_cast<const half8*>(net->d_w_in);
	p.w_tower = static_cast<const half8*>(net->d_w_tower);
	p.bias = static_cast<const float*>(net->d_bias);
	p.wp2 = static_cast<const float*>(net->d_wp2);
	p.wv1 = static_cast<const float*>(net->d_wv1);
	p.wv2 = static_cast<const half_t*>(net->d_wv2);
	p.bv2 = static_cast<const float*>(net->d_bv2);
	p.wv3 = static_cast<const float*>(net->d_wv3);
	p.bp2 = net->bp2;
	for (int i = 0; i < 4; i++)
		p.bv1[i] = net->bv1[i];
	for (int i = 0; i < 3; i++)
		p.bv3[i] = net->bv3[i];
	p.blocks = net->desc.blocks;
	p.batch = batch;
	p.slot_list = d_slot_list;
	p.count_ptr = d_count;
	AGX_REQUIRE(d_action_values == nullptr || net->desc.action_values, AGX_ERR_INVALID, "agx_nn_forward: the network has no action-values head");
	p.q = d_action_values;
	p.wq2 = (d_action_values != nullptr) ? static_cast<const float*>(net->d_wq2) : nullptr;
	for (int i = 0; i < 3; i++)
		p.bq2[i] = net->bq2[i];

	// persistent grid: one workgroup per CU it may use.  A pool stepped as pipelined slices narrows the launch (agx_net_set_launch_width) so that
	// the CUs it leaves free run the OTHER slice's search kernels at the same time: a tower workgroup fills its CU's LDS and registers, so
	// the two kinds of work never share a CU, they share the chip
	const int width = (net->launch_width > 0 && net->launch_width < net->num_cus) ? net->launch_width : net->num_cus;
	const int grid = (batch < width) ? batch : width;
	hipStream_t s = static_cast<hipStream_t>(stream);
	p.skip = nullptr;
	const int kpad = (net->desc.rows * net->desc.cols * 4 + 31) / 32 * 32;
	{ // per-stream scratch: launches on one stream are ordered and share it, launches on different streams may overlap
		std::lock_guard<std::mutex> lock(net->skip_mutex);
		AgxNet::StreamScratch *mine = nullptr;
		for (auto &slice : net->scratch_by_stream)
			if (slice.stream == s)
				mine = &slice;
		if (mine == nullptr)
		{
			net->scratch_by_stream.push_back(AgxNet::StreamScratch { s, nullptr, nullptr, 0 });
			mine = &net->scratch_by_stream.back();
		}
		if (net->inplace && mine->skip == nullptr)
			AGX_HIP_CHECK(hipMalloc(&mine->skip, net->skip_bytes));
		if (mine->vhead_boards < batch)
		{ // grows to the largest launch seen on this stream (a pool's launches all have the same capacity)
			if (mine->vhead != nullptr)
			{
				AGX_HIP_CHECK(hipStreamSynchronize(s));
				AGX_HIP_CHECK(hipFree(mine->vhead));
				mine->vhead = nullptr;
			}
			const size_t bytes = static_cast<size_t>(batch) * kpad * sizeof(half_t);
			AGX_HIP_CHECK(hipMalloc(&mine->vhead, bytes));
			AGX_HIP_CHECK(hipMemsetAsync(mine->vhead, 0, bytes, s)); // the k-padding stays zero: the tower kernel only writes the first 4 HW halves of a row
			mine->vhead_boards = batch;
		}
		p.skip = static_cast<half4*>(mine->skip);
		p.vhead_x = static_cast<half_t*>(mine->vhead);
	}
	const bool big = (net->desc.rows == 20), wide = (net->desc.filters == 128), qhead = (p.q != nullptr), raw = (net->desc.in_channels == 8);
	const dim3 g(grid), t(512);
#define AGX_LAUNCH_TOWER(FF, NN, IP, QH, RW) hipLaunchKernelGGL((nn_tower_kernel<FF, NN, NN, IP, QH, RW>), g, t, 0, s, p, d_features, d_policy, d_value)
#define AGX_LAUNCH_HEADS(FF, NN, IP) do { if (qhead) AGX_LAUNCH_TOWER(FF, NN, IP, true, false); else if (raw) AGX_LAUNCH_TOWER(FF, NN, IP, false, true); \
		else AGX_LAUNCH_TOWER(FF, NN, IP, false, false); } while (0)
	if (net->inplace)
	{
		if (big && wide)
			AGX_LAUNCH_HEADS(128, 20, true);
		else if (big)
			AGX_LAUNCH_HEADS(64, 20, true);
		else if (wide)
			AGX_LAUNCH_HEADS(128, 15, true);
		else
			AGX_LAUNCH_HEADS(64, 15, true);
	}
	else if (wide)
		AGX_LAUNCH_HEADS(128, 15, false);
	else
		AGX_LAUNCH_HEADS(64, 15, false);
#undef AGX_LAUNCH_HEADS
#undef AGX_LAUNCH_TOWER
	{ // the value head's dense layers for all boards of the launch
		const dim3 vg((batch + 15) / 16), vt(256);
#define AGX_LAUNCH_VALUE(KP, DD) hipLaunchKernelGGL((value_head_kernel<KP, DD>), vg, vt, 0, s, p.vhead_x, reinterpret_cast<const half8*>(p.wv2), p.bv2, p.wv3, \
		p.bv3[0], p.bv3[1], p.bv3[2], d_slot_list, d_count, batch, d_value)
		if (big && wide)
			AGX_LAUNCH_VALUE(1600, 256);
		else if (big)
			AGX_LAUNCH_VALUE(1600, 128);
		else if (wide)
			AGX_LAUNCH_VALUE(928, 256);
		else
			AGX_LAUNCH_VALUE(928, 128);
#undef AGX_LAUNCH_VALUE
	}
	AGX_HIP_CHECK(hipGetLastError());
	return AGX_OK;
}

int agx_nn_forward(AgxNet *net, const uint32_t *d_features, int batch, float *d_policy, float *d_value, void *stream)
{
	return launch_forward(net, d_features, nullptr, nullptr, batch, d_policy, d_value, nullptr, stream);
}
int agx_nn_forward_pvq(AgxNet *net, const uint32_t *d_features, int batch, float *d_policy, float *d_value, float *d_action_values, void *stream)
{
	return launch_forward(net, d_features, nullptr, nullptr, batch, d_policy, d_value, d_action_values, stream);
}
int agx_nn_forward_indirect(AgxNet *net, const uint32_t *d_features, const int *d_slot_list, const int *d_count, int max_batch, float *d_policy,
		float *d_value, void *stream)
{
	AGX_REQUIRE(d_slot_list != nullptr && d_count != nullptr, AGX_ERR_INVALID, "agx_nn_forward_indirect: null list");
	return launch_forward(net, d_features, d_slot_list, d_count, max_batch, d_policy, d_value, nullptr, stream);
}
int agx_nn_forward_indirect_pvq(AgxNet *net, const uint32_t *d_features, const int *d_slot_list, const int *d_count, int max_batch, float *d_policy,
		float *d_value, float *d_action_values, void *stream)
{
	AGX_REQUIRE(d_slot_list != nullptr && d_count != nullptr, AGX_ERR_INVALID, "agx_nn_forward_indirect_pvq: null list");
	return launch_forward(net, d_features, d_slot_list, d_count, max_batch, d_policy, d_value, d_action_values, stream);
}

#ifdef AGX_NN_PROFILE
/* profile builds only: shader cycles per phase summed over the workgroups (waves 0 and 4), then reset.  out[2][16] */
int agx_debug_nn_profile(unsigned long long *out)
{
	AGX_HIP_CHECK(hipDeviceSynchronize());
	AGX_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nn_prof), sizeof(unsigned long long) * 32));
	unsigned long long zero[32] = { 0 };
	AGX_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_nn_prof), zero, sizeof(zero)));
	return AGX_OK;
}
#endif

int agx_net_set_launch_width(AgxNet *net, int workgroups)
{
	AGX_REQUIRE(net != nullptr && workgroups >= 0, AGX_ERR_INVALID, "agx_net_set_launch_width: invalid argument");
	net->launch_width = workgroups;
	return AGX_OK;
}

int agx_net_description(const AgxNet *net, AgxNetDesc *out)
{
	AGX_REQUIRE(net != nullptr && out != nullptr, AGX_ERR_INVALID, "agx_net_description: null argument");
	*out = net->desc;
	return AGX_OK;
}

int agx_net_destroy(AgxNet *net)
{
	if (net == nullptr)
		return AGX_OK;
	free_net_buffers(net);
	delete net;
	return AGX_OK;
}

} /* extern "C" */
