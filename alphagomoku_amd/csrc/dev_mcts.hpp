/*
 * dev_mcts.hpp — device side of the graph-MCTS: PUCT select with wave-level argmax, transposition lookup, expand,
 * backup, information-leak repair.  One 64-lane wavefront owns one game; the edges of a node are scanned one per lane.
 *
 * Replaces (host code in the reference): Tree::{select,expand,backup,correctInformationLeak,cancelVirtualLoss}
 * (src/search/monte_carlo/Tree.cpp:226-384), PUCTSelector / PUCT_q_head / PUCT / BestEdge
 * (EdgeSelector.cpp:27-32,335-361,389-424,515-536,562-586,1123-1166), NodeCache::{seek,insert} (NodeCache.cpp:250-297),
 * UnifiedGenerator::generate (EdgeGenerator.cpp:23-127,269-303), Search::{select,expand,backup} (Search.cpp:117-232).
 *
 * Floating point is restated operation by operation (the library is built with -ffp-contract=off) so that visit counts
 * and move choices are bit-identical to the CPU oracle.
 */
#ifndef AGX_DEV_MCTS_HPP_
#define AGX_DEV_MCTS_HPP_

#include "dev_solver.hpp"
#include "root_noise.hpp"
#include "symmetry.hpp"

namespace agx
{
	namespace dev
	{
		/* a game's arenas are regions of the pool-wide heaps; kernels fetch these bases once per launch.  NB E.node_cap / E.edge_cap /
		 * E.ht_cap are the CLASS-0 capacities in the host's copy: a kernel that works on one game overrides them in its by-value copy of E
		 * with that game's capacities (use_game_arenas) before calling anything below. */
		__device__ __forceinline__ DNode* nodes_of(const EngineDev &E, int g, int arena) { return E.nodes + E.games[g].node_off[arena]; }
		__device__ __forceinline__ DEdge* edges_of(const EngineDev &E, int g, int arena) { return E.edges + E.games[g].edge_off[arena]; }
		__device__ __forceinline__ int* ht_of(const EngineDev &E, int g) { return E.ht + E.games[g].ht_off; }
		__device__ __forceinline__ void use_game_arenas(EngineDev &E, int g)
		{
			E.node_cap = E.games[g].node_cap;
			E.edge_cap = E.games[g].edge_cap;
			E.ht_cap = E.games[g].ht_cap;
		}

		/* NodeCache::seek (NodeCache.cpp:250-264): hash, side to move and the FULL (compressed) board must match */
		__device__ inline int cache_seek(const EngineDev &E, const DNode *nodes, const int *ht, u64 hash, const u64 *cboard, int sign, int lane)
		{
			const int mask = E.ht_cap - 1;
			int slot = static_cast<int>(hash & static_cast<u64>(mask));
			for (int probes = 0; probes < E.ht_cap; probes++)
			{
				const int idx = ht[slot];
				if (idx == 0)
					return -1;
				const DNode &nd = nodes[idx - 1];
				bool same = (nd.hash == hash) && (nd.sign_to_move == sign);
				if (same)
				{
					const bool word_ok = (lane < BWORDS) ? (nd.cboard[lane] == cboard[lane]) : true;
					same = (__ballot(!word_ok) == 0);
				}
				if (same)
					return idx - 1;
				slot = (slot + 1) & mask;
			}
			return -1;
		}
		/* the fields of a node record that the descent reads (everything but the compressed board), as one group of loads */
		__device__ __forceinline__ void node_head(DNode &dst, const DNode &src)
		{
			dst.edge_begin = src.edge_begin;
			dst.win = src.win;
			dst.draw = src.draw;
			dst.moves_left = src.moves_left;
			dst.visits = src.visits;
			dst.score = src.score;
			dst.n_edges = src.n_edges;
			dst.depth = src.depth;
			dst.vl = src.vl;
			dst.sign_to_move = src.sign_to_move;
			dst.flags = src.flags;
			dst.hash = src.hash;
		}
		/* cache_seek for the descent: the candidate's head is requested together with its board words, so the node found costs no
		 * further round trip (the leak test and the next level's selection read `head`) */
		__device__ inline int cache_seek_head(const EngineDev &E, const DNode *nodes, const int *ht, u64 hash, const u64 *cboard, int sign, int lane, DNode &head)
		{
			const int mask = E.ht_cap - 1;
			int slot = static_cast<int>(hash & static_cast<u64>(mask));
			for (int probes = 0; probes < E.ht_cap; probes++)
			{
				const int idx = ht[slot];
				if (idx == 0)
					return -1;
				const DNode &nd = nodes[idx - 1];
				DNode h;
				node_head(h, nd);
				const u64 word = (lane < BWORDS) ? nd.cboard[lane] : 0ull;
				bool same = (h.hash == hash) && (h.sign_to_move == sign);
				if (same)
				{
					const bool word_ok = (lane < BWORDS) ? (word == cboard[lane]) : true;
					same = (__ballot(!word_ok) == 0);
				}
				if (same)
				{
					head = h;
					return idx - 1;
				}
				slot = (slot + 1) & mask;
			}
			return -1;
		}
		__device__ inline void cache_insert(const EngineDev &E, int *ht, u64 hash, int node)
		{ // single lane
			const int mask = E.ht_cap - 1;
			int slot = static_cast<int>(hash & static_cast<u64>(mask));
			while (ht[slot] != 0)
				slot = (slot + 1) & mask;
			ht[slot] = node + 1;
		}

		/* has_information_leak (Tree.cpp:75-85) */
		__device__ inline bool has_leak(const EngineDev &E, const DEdge &e, bool node_found, uint32_t node_score, float node_win, float node_draw)
		{
			if (!node_found || E.leak_threshold >= 1.0f)
				return false;
			if (e.score != s_invert_up(node_score))
				return true;
			const float inv_win = 1.0f - (node_win + node_draw);
			const float dw = e.win - inv_win, dd = e.draw - node_draw;
			return (fabsf(dw) + fabsf(dd)) > E.leak_threshold;
		}

		/* the symmetry maps themselves: symmetry.hpp (agx::symmetry_source / inverse_symmetry / shuffle_feature_directions) */
		__device__ __forceinline__ unsigned long long symmetry_mix(unsigned long long z)
		{
			z += 0x9E3779B97F4A7C15ull;
			z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
			z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
			return z ^ (z >> 31);
		}

		/* wave-wide (value, index) argmax: largest value, lowest index among equals (EdgeSelector.cpp:562-586 scans in order with '>') */
		__device__ __forceinline__ void wave_argmax(float &value, int &index)
		{
			// order-preserving map of the float bits to unsigned, DPP max; then the lowest index among the lanes that hold the maximum
			const uint32_t bits = __float_as_uint(value);
			const uint32_t key = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
			const uint32_t best = wave_reduce_umax(key);
			const uint32_t mine = (key == best) ? (0x7FFFFFFFu - static_cast<uint32_t>(index)) : 0u;
			index = static_cast<int>(0x7FFFFFFFu - wave_reduce_umax(mine));
			value = __uint_as_float((best & 0x80000000u) ? (best & 0x7FFFFFFFu) : ~best);
		}
		__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
		{
			return wave_reduce_umax(v);
		}

		/* PUCTSelector::select without root noise (EdgeSelector.cpp:1123-1166) */
		__device__ inline int select_edge(const EngineDev &E, const DNode &nd, const DEdge *edges, int lane, unsigned long long &edge_reads, const float *noisy_priors)
		{
			const int total = nd.visits + nd.vl;
			float c_puct = E.c_puct;
			if (E.c_scale != 0.0f)
				c_puct = static_cast<float>(static_cast<double>(E.c_puct) + static_cast<double>(E.c_scale) * det_log(static_cast<double>(total))); // series log: same bits as the host
			const float parent_sqrt_visit = static_cast<float>(static_cast<double>(c_puct) * sqrt(static_cast<double>(total)));
			float initial_q = 0.0f;
			if (E.init_to == 1)
				initial_q = nd.win + 0.5f * nd.draw;
			else if (E.init_to == 2)
				initial_q = 0.5f;

			float best_value = -3.402823466e+38f;
			int best = 0x7FFFFFFF;
			for (int i = lane; i < nd.n_edges; i += 64)
			{
				const DEdge e = edges[nd.edge_begin + i];
				float value;
				switch (s_pv(e.score))
				{
					case 0:
						value = -1000.0f + s_distance(e.score);
						break;
					case 1:
						value = 0.5f;
						break;
					case 3:
						value = +1000.0f - s_distance(e.score);
						break;
					default:
					{
						const int vl = e.flag_vl & 0x7FFF;
						const bool being_expanded = (e.flag_vl & 0x8000u) != 0;
						const float visits = 1.0e-8f + e.visits;
						const float virtual_loss = static_cast<float>(vl);
						const float vl_factor = visits / (visits + virtual_loss);
						const float expectation = e.win + 0.5f * e.draw;
						float Q;
						if (E.init_to == 0)
							Q = being_expanded ? -1000.0f : expectation * vl_factor;
						else
						{
							Q = initial_q;
							if (being_expanded)
								Q = -1000.0f;
							else if (e.visits > 0)
								Q = expectation * vl_factor;
						}
						const float prior = (noisy_priors != nullptr) ? noisy_priors[i] : e.prior; // find_best_edge_impl<Op, UseNoise> (:562-586)
						const float U = prior * parent_sqrt_visit / (1.0f + e.visits + vl);
						value = Q + U;
						break;
					}
				}
				if (value > best_value)
				{
					best_value = value;
					best = i;
				}
			}
			edge_reads += nd.n_edges;
			wave_argmax(best_value, best);
			return nd.edge_begin + best;
		}

		/* Tree::correctInformationLeak (Tree.cpp:352-376).  Every lane carries the same arithmetic (lane 0 stores it); a level reads
		 * its node head and edge once and the child's value / score travel up in registers.  final_node exists (a leak needs one). */
		__device__ inline void correct_information_leak(DNode *nodes, DEdge *edges, const DTask &t, int path_len, int final_node, int lane)
		{
			float child_win = nodes[final_node].win, child_draw = nodes[final_node].draw;
			uint32_t child_score = nodes[final_node].score;
			for (int i = path_len - 1; i >= 0; i--)
			{
				const int node = t.path_node[i], e = t.path_edge[i];
				DNode nd;
				node_head(nd, nodes[node]);
				DEdge ed = edges[e];
				const float cw = ed.win, cd = ed.draw;
				const float tw = 1.0f - (child_win + child_draw), td = child_draw;
				const float scale = static_cast<float>(ed.visits) / static_cast<float>(nd.visits);
				const float nw = nd.win + (tw - cw) * scale, ndr = nd.draw + (td - cd) * scale;
				ed.win = tw;
				ed.draw = td;
				nd.win = nw;
				nd.draw = ndr;
				const uint32_t new_score = s_invert_up(child_score);
				ed.score = static_cast<uint16_t>(new_score);
				// update_score(Node*) (Tree.cpp:93-104)
				uint32_t result = 0;
				for (int j = lane; j < nd.n_edges; j += 64)
					result = max(result, (nd.edge_begin + j == e) ? new_score : static_cast<uint32_t>(edges[nd.edge_begin + j].score));
				result = wave_max_u32(result);
				if (((nd.flags & 4) != 0) || s_win(result) || s_unproven(result))
					nd.score = static_cast<uint16_t>(result);
				if (lane == 0)
				{
					nodes[node].win = nd.win;
					nodes[node].draw = nd.draw;
					nodes[node].score = nd.score;
					edges[e].win = ed.win;
					edges[e].draw = ed.draw;
					edges[e].score = ed.score;
				}
				child_win = nd.win;
				child_draw = nd.draw;
				child_score = nd.score;
			}
		}
		/* Tree::cancelVirtualLoss (Tree.cpp:377-384): one level per lane (the nodes and edges of a path are all different) */
		__device__ inline void cancel_virtual_loss(DNode *nodes, DEdge *edges, const DTask &t, int path_len, int lane)
		{
			for (int i = lane; i < path_len; i += 64)
			{
				nodes[t.path_node[i]].vl--;
				DEdge &e = edges[t.path_edge[i]];
				e.flag_vl = static_cast<uint16_t>((e.flag_vl & 0x8000u) | (((e.flag_vl & 0x7FFF) - 1) & 0x7FFF));
			}
		}

		__device__ inline u64 full_hash(const EngineDev &E, const uint8_t *board, int sign, int lane)
		{ // FullZobristHashing::getHash (ZobristHashing.cpp:21-33)
			u64 h = 0;
			for (int i = lane; i < E.hw; i += 64)
				h ^= E.nc_keys[3 + 3 * i + board[i]];
			return wave_reduce_xor64(h) ^ E.nc_keys[sign];
		}
	}
}

#endif
