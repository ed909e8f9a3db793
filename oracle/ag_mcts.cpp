/*
 * oracle/ag_mcts.cpp — TEST INFRASTRUCTURE ONLY.  Graph-MCTS (select / expand / backup with transpositions, virtual
 * loss, proven scores and information-leak repair), the batch driver and one self-play game.
 * Follows src/search/monte_carlo/{Tree,NodeCache,EdgeSelector,EdgeGenerator,Search}.cpp and
 * src/selfplay/GameGenerator.cpp.  PARITY UNPINNED by reference tests (test_Tree.cpp etc. are commented out).
 */
#include "agoracle.hpp"
#include "ag_noise.hpp"

#include <random>

namespace ago
{
	/* ------------------------------------------------ Tree ------------------------------------------------ */
	Tree::Tree(GameConfig c, const SearchConfig &sc) : cfg(c), scfg(sc)
	{
		// FullZobristHashing (ZobristHashing.cpp:15-33): 3*(1+HW) 64-bit keys.  Key values do not influence results
		// because NodeCache::seek compares the full board (NodeCache.cpp:257).
		keys.resize(3 + 3 * c.rows * c.cols);
		uint64_t st = sc.zobrist_seed ^ 0x5851F42D4C957F2Dull;
		for (size_t i = 0; i < keys.size(); i++)
			keys[i] = splitmix64(st);
		bins.assign(1u << 12, std::vector<int>());
		base_board.assign(c.rows * c.cols, NONE);
	}
	void Tree::clear()
	{
		nodes.clear();
		edges.clear();
		node_boards.clear();
		node_hash.clear();
		for (auto &b : bins)
			b.clear();
		root = -1;
	}
	uint64_t full_zobrist_hash(const uint64_t *keys, const Sign *board, int cells, Sign to_move)
	{ // FullZobristHashing::getHash (ZobristHashing.cpp:21-33): the key of the side to move, then one key per cell and content
		uint64_t h = keys[to_move];
		for (int i = 0, k = 3; i < cells; i++, k += 3)
			h ^= keys[k + board[i]];
		return h;
	}
	uint64_t Tree::hash_of(const Sign *board, Sign to_move) const
	{
		return full_zobrist_hash(keys.data(), board, cfg.rows * cfg.cols, to_move);
	}
	int Tree::seek(const Sign *board, Sign to_move) const
	{ // NodeCache.cpp:250-264
		const uint64_t h = hash_of(board, to_move);
		const std::vector<int> &bin = bins[h & (bins.size() - 1)];
		for (int n : bin)
			if (node_hash[n] == h && nodes[n].sign_to_move == to_move && std::memcmp(node_boards[n].data(), board, cfg.rows * cfg.cols) == 0)
				return n;
		return -1;
	}
	void Tree::set_board(const Sign *board, Sign to_move)
	{ // Tree.cpp:128-151; NodeCache::cleanup (NodeCache.cpp:221-249) keeps an entry iff every stone of the new root
	  // position is present, with the same colour, in the stored position (isTransitionPossibleFrom :95-115).
		const int hw = cfg.rows * cfg.cols;
		std::vector<Node> new_nodes;
		std::vector<Edge> new_edges;
		std::vector<std::vector<Sign>> new_boards;
		std::vector<uint64_t> new_hash;
		for (size_t n = 0; n < nodes.size(); n++)
		{
			bool keep = true;
			for (int i = 0; i < hw && keep; i++)
				if (board[i] != NONE && node_boards[n][i] != board[i])
					keep = false;
			if (!keep)
				continue;
			Node copy = nodes[n];
			const int begin = static_cast<int>(new_edges.size());
			new_edges.insert(new_edges.end(), edges.begin() + copy.edge_begin, edges.begin() + copy.edge_begin + copy.n_edges);
			copy.edge_begin = begin;
			new_nodes.push_back(copy);
			new_boards.push_back(node_boards[n]);
			new_hash.push_back(node_hash[n]);
		}
		nodes.swap(new_nodes);
		edges.swap(new_edges);
		node_boards.swap(new_boards);
		node_hash.swap(new_hash);
		for (auto &b : bins)
			b.clear();
		for (size_t n = 0; n < nodes.size(); n++)
			bins[node_hash[n] & (bins.size() - 1)].push_back(static_cast<int>(n));

		base_board.assign(board, board + hw);
		sign_to_move = to_move;
		root = seek(board, to_move);
		if (root >= 0)
			nodes[root].flags |= 2; // markAsRoot
		max_depth = 0; // Tree.cpp:150
	}
	bool Tree::has_information_leak(const Edge &e, int node) const
	{ // Tree.cpp:75-85
		if (node < 0 || scfg.information_leak_threshold >= 1.0f)
			return false;
		if (e.score != invert_up(nodes[node].score))
			return true;
		const Value diff = e.value - nodes[node].value.inverted();
		return diff.abs() > scfg.information_leak_threshold;
	}
	void Tree::update_node_score(int node)
	{ // Tree.cpp:93-104
		Node &n = nodes[node];
		Score result = Score::minus_inf();
		for (int i = 0; i < n.n_edges; i++)
			result = std::max(result, edges[n.edge_begin + i].score);
		if (n.fully_expanded() || result.is_win() || result.is_unproven())
			n.score = result;
	}
	int Tree::select_edge(int node) const
	{ // PUCTSelector::select (EdgeSelector.cpp:1123-1166) without root noise; ops PUCT_q_head (:335-361) / PUCT (:389-424)
		const Node &n = nodes[node];
		// std::log in the reference; the series log of ag_noise.hpp (error < 1e-15) is used so that the device can reproduce the bits
		const float c_puct = (scfg.exploration_scaling == 0.0f) ? scfg.exploration_constant
				: static_cast<float>(static_cast<double>(scfg.exploration_constant) + static_cast<double>(scfg.exploration_scaling) * det_log(static_cast<double>(n.visits + n.vl)));
		const float parent_sqrt_visit = static_cast<float>(c_puct * std::sqrt(static_cast<double>(n.visits + n.vl)));
		float initial_q = 0.0f;
		if (scfg.init_to == 1)
			initial_q = n.value.expectation();
		else if (scfg.init_to == 2)
			initial_q = 0.5f;

		const bool use_noise = (node == root) && scfg.noise_type != 0 && scfg.noise_weight > 0.0f; // EdgeSelector.cpp:1127-1137
		if (use_noise && noisy_policy.empty())
		{
			std::vector<float> priors(n.n_edges);
			for (int i = 0; i < n.n_edges; i++)
				priors[i] = edges[n.edge_begin + i].prior;
			noisy_policy.resize(n.n_edges);
			make_root_noise(scfg.noise_type, scfg.noise_weight, scfg.noise_seed, noise_serial, noise_move, n.n_edges, [&](int i) { return priors[i]; },
					noisy_policy.data());
		}

		int best = -1;
		float best_value = std::numeric_limits<float>::lowest();
		for (int i = 0; i < n.n_edges; i++)
		{
			const Edge &e = edges[n.edge_begin + i];
			const float policy_prior = use_noise ? noisy_policy[i] : e.prior; // find_best_edge_impl<Op, UseNoise> (:562-586)
			float value;
			switch (e.score.pv())
			{
				case PV_LOSS:
					value = -1000.0f + e.score.distance();
					break;
				case PV_DRAW:
					value = 0.5f;
					break;
				case PV_WIN:
					value = +1000.0f - e.score.distance();
					break;
				default:
				{
					const float visits = 1.0e-8f + e.visits; // getVirtualLoss (:27-32)
					const float virtual_loss = static_cast<float>(e.vl());
					const float vl_factor = visits / (visits + virtual_loss);
					float Q;
					if (scfg.init_to == 0)
						Q = e.being_expanded() ? -1000.0f : e.value.expectation() * vl_factor;
					else
					{
						Q = initial_q;
						if (e.being_expanded())
							Q = -1000.0f;
						else if (e.visits > 0)
							Q = e.value.expectation() * vl_factor;
					}
					const float U = policy_prior * parent_sqrt_visit / (1.0f + e.visits + e.vl());
					value = Q + U;
					break;
				}
			}
			stats.select_edges++;
			if (value > best_value)
			{ // first strict maximum (:568-580)
				best_value = value;
				best = i;
			}
		}
		stats.select_levels++;
		return n.edge_begin + best;
	}
	int Tree::select_final_edge(int node, int selector) const
	{ // MaxVisit / MinVisit / MaxValue / MaxPolicy ops under find_best_edge (EdgeSelector.cpp:476-514,562-586): first maximum wins
		if (selector == 0)
			return select_best_edge(node);
		const Node &n = nodes[node];
		int best = -1;
		float best_value = std::numeric_limits<float>::lowest();
		for (int i = 0; i < n.n_edges; i++)
		{
			const Edge &e = edges[n.edge_begin + i];
			float value;
			switch (selector)
			{
				case 1:
					value = e.visits;
					break;
				case 2:
					value = -e.visits;
					break;
				case 3:
					switch (e.score.pv())
					{
						case PV_LOSS:
							value = -1000.0f + e.score.distance();
							break;
						case PV_DRAW:
							value = Value(0.0f, 1.0f).expectation();
							break;
						case PV_WIN:
							value = +1000.0f - e.score.distance();
							break;
						default:
							value = e.value.expectation();
							break;
					}
					break;
				case 5:
				{ // LCB op (EdgeSelector.cpp:446-475) with the parent value as initial Q; ln by the series of ag_noise.hpp
					if (e.score.pv() == PV_LOSS)
						value = -1.0e6f + e.score.distance() + e.prior;
					else if (e.score.pv() == PV_WIN)
						value = +1.0e6f - e.score.distance() + e.prior;
					else
					{
						const float visits = 1.0e-8f + e.visits;
						const float vl_factor = visits / (visits + static_cast<float>(e.vl()));
						const float Q = (e.visits > 0) ? e.value.expectation() : n.value.expectation();
						const float parent_log_visit = static_cast<float>(det_log(static_cast<double>(n.visits + n.vl)));
						const float U = scfg.exploration_constant * std::sqrt(parent_log_visit / (1.0f + e.visits + e.vl()));
						value = Q * vl_factor - U;
					}
					break;
				}
				default:
					value = e.prior;
					break;
			}
			if (value > best_value)
			{
				best_value = value;
				best = i;
			}
		}
		return n.edge_begin + best;
	}
	int Tree::select_best_edge(int node) const
	{ // BestEdgeSelector / BestEdge (EdgeSelector.cpp:515-536)
		const Node &n = nodes[node];
		int best = -1;
		float best_value = std::numeric_limits<float>::lowest();
		for (int i = 0; i < n.n_edges; i++)
		{
			const Edge &e = edges[n.edge_begin + i];
			float value;
			switch (e.score.pv())
			{
				case PV_LOSS:
					value = -1.0e8f + e.score.distance();
					break;
				case PV_WIN:
					value = +1.0e8f - e.score.distance();
					break;
				default:
					value = e.visits + e.value.expectation() * n.visits + 0.001f * e.prior;
					break;
			}
			if (value > best_value)
			{
				best_value = value;
				best = i;
			}
		}
		return n.edge_begin + best;
	}
	static void task_reset(Task &t, const std::vector<Sign> &base, Sign to_move)
	{ // SearchTask::set (SearchTask.cpp:32-50) — the two "solved" flags are NOT reset there
		const size_t hw = base.size();
		t.path.clear();
		t.edges.clear();
		t.board = base;
		t.features.assign(hw, 0u);
		t.policy.assign(hw, 0.0f);
		t.action_values.assign(hw, Value());
		t.value = Value();
		t.action_scores.assign(hw, Score());
		t.score = Score();
		t.final_node = -1;
		t.sign_to_move = to_move;
		t.must_defend = false;
		t.by_network = false;
		t.by_solver = false;
		t.skip_edge_generation = false;
	}
	int Tree::select(Task &t)
	{ // Tree.cpp:226-251
		task_reset(t, base_board, sign_to_move);
		int node = root;
		while (node >= 0)
		{
			const int e = select_edge(node);
			const Move m = edges[e].move;
			t.board[m.row * cfg.cols + m.col] = m.sign; // SearchTask::append (SearchTask.cpp:52-60)
			t.sign_to_move = invert_sign(m.sign);
			t.path.push_back(std::make_pair(node, e));
			nodes[node].vl++;
			edges[e].flag_vl = static_cast<uint16_t>((edges[e].flag_vl & 0x8000u) | ((edges[e].vl() + 1) & 0x7FFF));

			if (edges[e].score.is_proven())
				return 2;
			node = seek(t.board.data(), t.sign_to_move);
			t.final_node = node;
			if (node < 0)
				edges[e].flag_vl |= 0x8000u;
			if (has_information_leak(edges[e], node))
				return 1;
		}
		max_depth = std::max(max_depth, static_cast<int>(t.path.size())); // Tree.cpp:249
		return 0;
	}
	void Tree::generate_edges(Task &t) const
	{ // UnifiedGenerator::generate (EdgeGenerator.cpp:269-303) for tasks that went through the solver, temperature 1
		for (Edge &e : t.edges)
		{ // initialize_edges (:88-127)
			const int i = e.move.row * cfg.cols + e.move.col;
			e.prior = t.policy[i];
			e.value = t.action_values[i];
			e.score = t.action_scores[i];
		}
		if (scfg.policy_temperature == 0.0f)
		{ // :90-100: prior 1 where the policy holds its maximum over the whole plane
			float max_p = t.policy[0];
			for (float p : t.policy)
				max_p = std::max(max_p, p);
			for (Edge &e : t.edges)
				e.prior = (e.prior == max_p) ? 1.0f : 0.0f;
		}
		else if (scfg.policy_temperature != 1.0f)
		{ // :111-117: std::pow(p, 1 / T) — evaluated here (and on the device) as exp(log(p) / T) with the fixed series of
		  // oracle/ag_noise.hpp, so both sides agree bit for bit; libm's powf differs in the last bits
			const float inv_t = 1.0f / scfg.policy_temperature;
			for (Edge &e : t.edges)
				e.prior = (e.prior > 0.0f) ? static_cast<float>(det_exp(det_log(static_cast<double>(e.prior)) * static_cast<double>(inv_t))) : 0.0f;
		}
		const bool expand_fully = t.path.empty() && scfg.force_expand_root != 0; // relative depth 0 and force_expand_root == true (GameGenerator.cpp:183-184)
		if (!expand_fully)
		{ // prune_weak_moves (:49-86)
			size_t idx = 0;
			bool erase = true;
			if (t.score.is_proven())
			{
				Score best = Score::loss_in(0);
				for (const Edge &e : t.edges)
					best = std::max(best, e.score);
				for (size_t i = 0; i < t.edges.size(); i++)
					if (t.edges[i].score == best)
					{
						std::swap(t.edges[i], t.edges[idx]);
						idx++;
					}
			}
			else
			{
				const size_t max_edges = static_cast<size_t>(scfg.max_children);
				if (t.edges.size() <= max_edges || t.must_defend)
					erase = false;
				else
				{ // EdgeComparator<MaxPolicyPrior> (Edge.hpp:156-172)
					auto cmp = [](const Edge &a, const Edge &b)
					{
						if ((a.score.is_proven() || b.score.is_proven()) && a.score != b.score)
							return a.score > b.score;
						return a.prior > b.prior;
					};
					// the reference calls std::partial_sort, which leaves the order of equal elements unspecified; a stable sort of the
					// whole list gives the same first max_edges elements whenever no two edges compare equal, and a defined order otherwise
					std::stable_sort(t.edges.begin(), t.edges.end(), cmp);
					float sum = 0.0f;
					for (size_t i = 0; i < max_edges; i++)
						sum += t.edges[i].prior;
					const float threshold = scfg.policy_expansion_threshold * sum;
					for (size_t i = 0; i < max_edges; i++)
						if (t.edges[i].prior >= threshold)
							idx++;
				}
			}
			if (erase)
				t.edges.erase(t.edges.begin() + idx, t.edges.end());
		}
		// renormalize_policy (:23-40)
		float sum = 0.0f;
		for (const Edge &e : t.edges)
			sum += e.prior;
		if (sum == 0.0f)
		{
			const float u = 1.0f / t.edges.size();
			for (Edge &e : t.edges)
				e.prior = u;
		}
		else
		{
			const float inv = 1.0f / sum;
			for (Edge &e : t.edges)
				e.prior = e.prior * inv;
		}
	}
	int Tree::expand(Task &t)
	{ // Tree.cpp:257-298
		if (t.edges.empty())
			return 2;
		int node = seek(t.board.data(), t.sign_to_move);
		if (node < 0)
		{
			Node n;
			n.edge_begin = static_cast<int>(edges.size());
			n.n_edges = static_cast<int16_t>(t.edges.size());
			edges.insert(edges.end(), t.edges.begin(), t.edges.end());
			int stones = 0;
			for (Sign s : t.board)
				stones += (s != NONE);
			n.depth = static_cast<int16_t>(stones);
			n.sign_to_move = t.sign_to_move;
			n.update_value(t.value);
			n.moves_left += (t.moves_left - n.moves_left) / n.visits; // Node::updateMovesLeft (Node.hpp:275-278)
			if (t.must_defend || (n.n_edges + n.depth) == static_cast<int>(t.board.size()))
				n.flags |= 4;
			n.flags |= (t.statically_solved ? 8 : 0) | (t.recursively_solved ? 16 : 0) | (t.must_defend ? 32 : 0);
			node = static_cast<int>(nodes.size());
			nodes.push_back(n);
			node_boards.push_back(t.board);
			const uint64_t h = hash_of(t.board.data(), t.sign_to_move);
			node_hash.push_back(h);
			if (nodes.size() > 2 * bins.size())
			{
				bins.assign(bins.size() * 4, std::vector<int>());
				for (size_t k = 0; k + 1 < nodes.size(); k++)
					bins[node_hash[k] & (bins.size() - 1)].push_back(static_cast<int>(k));
			}
			bins[h & (bins.size() - 1)].push_back(node);
			update_node_score(node);
			t.final_node = node;
			if (t.path.empty())
			{
				root = node;
				nodes[root].flags |= 2;
			}
			return 0;
		}
		t.final_node = node;
		if (!t.path.empty() && has_information_leak(edges[t.path.back().second], node))
			correct_information_leak(t);
		return 1;
	}
	void Tree::backup(const Task &t)
	{ // Tree.cpp:299-351
		float moves_left = t.moves_left;
		for (int i = static_cast<int>(t.path.size()) - 1; i >= 0; i--)
		{
			const int node = t.path[i].first, e = t.path[i].second;
			const int next = (i == static_cast<int>(t.path.size()) - 1) ? t.final_node : t.path[i + 1].first;
			const Value value = (nodes[node].sign_to_move == t.sign_to_move) ? t.value : t.value.inverted();
			nodes[node].update_value(value);
			edges[e].update_value(value);
			nodes[node].moves_left += (moves_left - nodes[node].moves_left) / nodes[node].visits;
			moves_left += 1.0f;
			if (next >= 0)
				edges[e].score = invert_up(nodes[next].score);
			update_node_score(node);
			nodes[node].vl--;
			edges[e].flag_vl = static_cast<uint16_t>((edges[e].vl() - 1) & 0x7FFF); // decreaseVirtualLoss + clearFlags
		}
	}
	void Tree::correct_information_leak(const Task &t)
	{ // Tree.cpp:352-376
		for (int i = static_cast<int>(t.path.size()) - 1; i >= 0; i--)
		{
			const int node = t.path[i].first, e = t.path[i].second;
			const int next = (i == static_cast<int>(t.path.size()) - 1) ? t.final_node : t.path[i + 1].first;
			const Value current = edges[e].value;
			const Value target = nodes[next].value.inverted();
			const float scale = static_cast<float>(edges[e].visits) / static_cast<float>(nodes[node].visits);
			const Value target_node = nodes[node].value + (target - current) * scale;
			edges[e].value = target;
			nodes[node].value = target_node;
			edges[e].score = invert_up(nodes[next].score);
			update_node_score(node);
		}
	}
	void Tree::cancel_virtual_loss(const Task &t)
	{ // Tree.cpp:377-384
		for (const auto &p : t.path)
		{
			nodes[p.first].vl--;
			edges[p.second].flag_vl = static_cast<uint16_t>((edges[p.second].flag_vl & 0x8000u) | ((edges[p.second].vl() - 1) & 0x7FFF));
		}
	}

	/* ------------------------------------------------ Search ------------------------------------------------ */
	Search::Search(GameConfig c, const SearchConfig &sc) : cfg(c), scfg(sc), solver(c, sc.tss_table_entries, sc.zobrist_seed)
	{
		tasks.resize(sc.max_batch_size);
		solver.max_nodes = sc.tss_max_positions;
	}
	void Search::select(Tree &tree, int max_simulations)
	{ // Search.cpp:117-158
		int trials = 2 * static_cast<int>(tasks.size());
		while (stored < static_cast<int>(tasks.size()) && tree.simulation_count() <= max_simulations)
		{
			Task &t = tasks[stored++];
			const int out = tree.select(t);
			if (t.path.empty())
				break;
			for (int i = 0; i < stored - 1; i++)
				if (!tasks[i].path.empty() && tasks[i].path.back().second == t.path.back().second)
				{
					stats.duplicates++;
					break;
				}
			if (out == 1)
			{
				tree.correct_information_leak(t);
				tree.cancel_virtual_loss(t);
				stats.leaks++;
				stored--;
			}
			if (out == 2)
			{
				const Score s = tree.edges[t.path.back().second].score;
				t.final_node = -1;
				t.sign_to_move = invert_sign(t.sign_to_move);
				t.score = s;
				t.value = s.to_value();
				t.by_solver = true;
				t.skip_edge_generation = true;
				stats.proven++;
			}
			if (--trials <= 0)
				break;
		}
	}
	void Search::solve()
	{ // Search.cpp:159-183 + AlphaBetaSearch::solve's writes into the task (AlphaBetaSearch.cpp:114-139)
		for (int i = 0; i < stored; i++)
		{
			Task &t = tasks[i];
			if (t.by_solver)
				continue;
			Solver::Output out;
			solver.solve(t.board.data(), t.sign_to_move, t.features.data(), out);
			for (const Action &a : out.actions)
			{
				const int c = a.move.row * cfg.cols + a.move.col;
				t.action_scores[c] = a.score;
				if (a.score.is_proven())
					t.action_values[c] = a.score.to_value();
				Edge e;
				e.move = a.move;
				t.edges.push_back(e);
			}
			t.score = out.score;
			if (t.score.is_proven())
			{
				t.value = t.score.to_value();
				t.moves_left = static_cast<float>(t.score.distance());
			}
			if (out.must_defend)
				t.must_defend = true;
			if (out.score.is_proven())
				t.recursively_solved = true;
			if (out.nodes <= 1)
				t.statically_solved = true;
			t.by_solver = true;
			stats.solver_nodes += out.nodes;
		}
	}
	int Search::schedule(std::vector<int> &out) const
	{ // Search.cpp:184-199
		out.clear();
		for (int i = 0; i < stored; i++)
			if (tasks[i].path.empty() || !tasks[i].score.is_proven())
				out.push_back(i);
		return static_cast<int>(out.size());
	}
	void Search::generate_edges(const Tree &tree)
	{
		for (int i = 0; i < stored; i++)
			if (!tasks[i].skip_edge_generation)
				tree.generate_edges(tasks[i]);
	}
	void Search::expand(Tree &tree)
	{
		for (int i = 0; i < stored; i++)
			if (tree.expand(tasks[i]) == 1)
				stats.wasted++;
	}
	void Search::backup(Tree &tree)
	{
		stats.nodes += stored;
		for (int i = 0; i < stored; i++)
			tree.backup(tasks[i]);
		stored = 0;
	}
	void Search::cleanup(Tree &tree)
	{ // Search.cpp:233-242: both buffers
		for (int b = 0; b < 2; b++)
		{
			for (int i = 0; i < stored; i++)
				tree.cancel_virtual_loss(tasks[i]);
			stored = 0;
			switch_buffer();
		}
	}
	void Search::switch_buffer()
	{ // Search.cpp:248-251 (the buffers are swapped instead of indexed)
		if (other_tasks.size() != tasks.size())
			other_tasks.resize(tasks.size());
		std::swap(tasks, other_tasks);
		std::swap(stored, other_stored);
	}

	/* ------------------------------------------------ Game ------------------------------------------------ */
	Game::Game(GameConfig c, const SearchConfig &sc) : cfg(c), scfg(sc), tree(c, sc), search(c, sc)
	{
		board.assign(c.rows * c.cols, NONE);
	}
	void Game::begin(const std::vector<Move> &opening)
	{ // GameGenerator.cpp:48-77 (GAME_NOT_STARTED -> loadOpening -> prepare_search)
		std::fill(board.begin(), board.end(), NONE);
		moves.clear();
		records.clear();
		outcome = O_UNKNOWN;
		queued = 0;
		tree.clear();
		for (int l = 0; l < lanes(); l++)
		{
			lane(l).solver.clear();
			lane(l).stored = 0;
			lane(l).other_stored = 0;
		}
		for (const Move &m : opening)
		{
			board[m.row * cfg.cols + m.col] = m.sign;
			moves.push_back(m);
		}
		sign_to_move = moves.empty() ? CROSS : invert_sign(moves.back().sign); // Game::getSignToMove (Game.cpp:60-69)
		prepare_search();
	}
	void Game::set_search_threads(int count)
	{
		more_searches.clear();
		for (int i = 1; i < count; i++)
			more_searches.push_back(std::make_unique<Search>(cfg, scfg));
	}
	void Game::prepare_search()
	{ // GameGenerator.cpp:174-185
		for (int l = 0; l < lanes(); l++)
			lane(l).cleanup(tree);
		tree.set_board(board.data(), sign_to_move);
		tree.noisy_policy.clear(); // a fresh EdgeSelector per move (GameGenerator.cpp:181-183)
		tree.noise_serial = serial;
		tree.noise_move = static_cast<int>(moves.size());
		for (int l = 0; l < lanes(); l++)
			lane(l).solver.increase_generation();
	}
	int Game::step_select(std::vector<uint32_t> &features_out)
	{ // GameGenerator.cpp:79-86
		// SearchThread::serial_run (SearchThread.cpp:121-146) of every thread in lock-step: select under the tree lock in thread order ...
		for (int l = 0; l < lanes(); l++)
			lane(l).select(tree, scfg.max_simulations);
		scheduled.clear();
		scheduled_lane.clear();
		for (int l = 0; l < lanes(); l++)
		{ // ... each thread's own solver on its own batch, its leaves queued for the shared evaluator
			lane(l).solve();
			std::vector<int> mine;
			const int m = lane(l).schedule(mine);
			lane(l).stats.nn_evals += m;
			for (int idx : mine)
			{
				scheduled.push_back(idx);
				scheduled_lane.push_back(l);
			}
		}
		const int n = static_cast<int>(scheduled.size());
		const int hw = cfg.rows * cfg.cols;
		features_out.resize(static_cast<size_t>(n) * hw);
		symmetries.assign(n, 0);
		for (int i = 0; i < n; i++)
		{
			Task &t = lane(scheduled_lane[i]).tasks[scheduled[i]];
			if (scfg.use_symmetries)
			{ // NNEvaluator::addToQueue + pack_to_network: features.augment(symmetry) (NNEvaluator.cpp:134-141,244-262)
				const int s = pick_symmetry(scfg.symmetry_seed, serial, queued);
				symmetries[i] = s;
				if (s != 0)
				{
					const std::vector<uint32_t> src = t.features;
					for (int r = 0; r < cfg.rows; r++)
						for (int c = 0; c < cfg.cols; c++)
						{
							int sr, sc;
							symmetry_source(s, cfg.rows, r, c, sr, sc);
							t.features[r * cfg.cols + c] = shuffle_feature_directions(src[sr * cfg.cols + sc], s);
						}
				}
			}
			queued++;
			std::memcpy(features_out.data() + static_cast<size_t>(i) * hw, t.features.data(), hw * sizeof(uint32_t));
		}
		return n;
	}
	int Game::step_expand(const float *policy, const float *value, const float *action_values)
	{
		unpack_network(policy, value, action_values);
		return expand_and_move();
	}
	int Game::async_step(std::vector<uint32_t> &features_out)
	{ // SearchThread.cpp:152-170: generateEdges / expand / backup (current buffer), stop condition, select, solve, scheduleToNN
		expand_and_move();
		if (outcome != O_UNKNOWN)
		{
			features_out.clear();
			scheduled.clear();
			scheduled_lane.clear();
			return 0;
		}
		return step_select(features_out);
	}
	void Game::async_provide(const float *policy, const float *value)
	{ // SearchThread.cpp:171-174: the batch goes to the network (its answer is read by the expand of this buffer's next turn), switchBuffer
		unpack_network(policy, value, nullptr);
		for (int l = 0; l < lanes(); l++)
			lane(l).switch_buffer();
	}
	void Game::unpack_network(const float *policy, const float *value, const float *action_values)
	{ // NNEvaluator::unpack_from_network (NNEvaluator.cpp:263-286) with symmetry 0 and a 'pv' network (no 'q'/'m' heads:
	  // those output tensors stay zero, NetworkDataPack.cpp:122-126,214-235)
		const int hw = cfg.rows * cfg.cols;
		for (size_t i = 0; i < scheduled.size(); i++)
		{
			Task &t = lane(scheduled_lane.empty() ? 0 : scheduled_lane[i]).tasks[scheduled[i]];
			const int inv = inverse_symmetry(symmetries.empty() ? 0 : symmetries[i]);
			for (int k = 0; k < hw; k++)
			{ // apply_symmetry(task policy, network policy, inverse symmetry) (NNEvaluator.cpp:277-279)
				int sr, sc;
				symmetry_source(inv, cfg.rows, k / cfg.cols, k % cfg.cols, sr, sc);
				t.policy[k] = policy[i * hw + sr * cfg.cols + sc];
				// the 'q' tensor overwrites every action value (zero-filled for a 'pv' network), NNEvaluator.cpp:279
				t.action_values[k] = (action_values != nullptr) ? Value(action_values[(i * hw + sr * cfg.cols + sc) * 2], action_values[(i * hw + sr * cfg.cols + sc) * 2 + 1]) : Value();
			}
			t.value = Value(value[2 * i], value[2 * i + 1]);
			if (t.score.is_unproven())
				t.moves_left = 0.0f;
			t.by_network = true;
		}
	}
	int Game::expand_and_move()
	{ // GameGenerator.cpp:88-118
		for (int l = 0; l < lanes(); l++)
		{ // each thread in turn, under the tree lock: expand its batch, back it up (SearchThread.cpp:135-141)
			lane(l).generate_edges(tree);
			lane(l).expand(tree);
			lane(l).backup(tree);
		}

		if (tree.root < 0)
			return 0; // (the first iterations of the double-buffered loop: nothing expanded yet)
		const float draw_rate = tree.nodes[tree.root].value.draw;
		// get_simulations_for_move (utils/misc.cpp:171-179)
		const float reduction = std::max(0.0f, std::min(1.0f, (draw_rate - 0.75f) / (1.0f - 0.75f)));
		const int simulations = static_cast<int>(scfg.max_simulations - reduction * (scfg.max_simulations - 50));
		if (tree.simulation_count() > simulations || tree.root_proven())
		{
			make_move();
			if (external_opponent)
				awaiting = true; // EvaluationGame.cpp:126-143: the OTHER player's setBoard follows, this one waits for its next turn
			else if (outcome == O_UNKNOWN)
				prepare_search();
			return 1;
		}
		return 0;
	}
	void Game::match_begin(const std::vector<Move> &opening)
	{ // EvaluationGame.cpp:77-106: getSolver().clear(), newGame(), loadOpening; only the player to move gets setBoard (take_turn)
		external_opponent = true;
		awaiting = true;
		std::fill(board.begin(), board.end(), NONE);
		moves.clear();
		records.clear();
		outcome = O_UNKNOWN;
		queued = 0;
		search.solver.clear();
		for (const Move &m : opening)
		{
			board[m.row * cfg.cols + m.col] = m.sign;
			moves.push_back(m);
		}
		sign_to_move = moves.empty() ? CROSS : invert_sign(moves.back().sign);
	}
	void Game::take_turn()
	{ // Player::setBoard (Player.cpp:98-110): cleanup, Tree::setBoard on the current board (two plies ahead of this player's last
	  // search), Search::setBoard -> increaseGeneration, fresh selector
		prepare_search();
		awaiting = false;
	}
	void Game::external_move(Move m)
	{ // Game::makeMove (game/Game.cpp:104-122) by the other player
		board[m.row * cfg.cols + m.col] = m.sign;
		moves.push_back(m);
		sign_to_move = invert_sign(m.sign);
		outcome = get_outcome(cfg.rules, board.data(), cfg.rows, cfg.cols, m, cfg.draw_after);
	}
	void Game::make_move()
	{ // GameGenerator.cpp:145-173
		const Node &r = tree.nodes[tree.root];
		MoveRecord rec;
		rec.root_visits = r.visits;
		rec.root_value = r.value;
		rec.root_score = r.score;
		rec.root_flags = (r.flags >> 3) & 7; // wasStaticallySolved, wasRecursivelySolved, mustDefend (Node.hpp:26-42 flag bits 8, 16, 32)
		rec.stones = static_cast<int>(moves.size());
		rec.root_edges.assign(tree.edges.begin() + r.edge_begin, tree.edges.begin() + r.edge_begin + r.n_edges);
		const Move m = tree.edges[tree.select_final_edge(tree.root, scfg.final_selector)].move;
		rec.move = m;
		records.push_back(rec);
		board[m.row * cfg.cols + m.col] = m.sign;
		moves.push_back(m);
		sign_to_move = invert_sign(m.sign);
		outcome = get_outcome(cfg.rules, board.data(), cfg.rows, cfg.cols, m, cfg.draw_after);
	}

	/* ------------------------------------------------ openings ------------------------------------------------ */
	std::vector<Move> prepare_opening(GameConfig cfg, uint32_t seed)
	{ // utils/misc.cpp:108-170.  Same distribution; the random stream is std::mt19937(seed) consumed in the order below
	  // (the reference uses thread-local time-seeded generators, utils/random.cpp:17-23, so streams cannot be replayed).
		std::mt19937 rng(seed);
		auto rand_int = [&](int n) { return static_cast<int>(rng() % static_cast<uint32_t>(n)); };
		auto rand_float = [&]() { return static_cast<float>(rng() >> 8) * (1.0f / 16777216.0f); };
		const int hw = cfg.rows * cfg.cols;
		std::vector<float> dist(hw);
		std::vector<Sign> board(hw);
		while (true)
		{
			std::vector<Move> result;
			std::fill(board.begin(), board.end(), NONE); // NB the distance map is NOT cleared between attempts (misc.cpp:144)
			Sign to_move = CROSS;
			int opening_moves = std::max(1, rand_int(6) + rand_int(6) + rand_int(6));
			if (rand_int(1000) == 0)
				opening_moves = 0;
			for (int k = 0; k < opening_moves; k++)
			{
				if (result.empty())
				{ // generateOpeningMap on an empty board (:111-120); NB accumulates into dist (+=)
					for (int i = 0; i < cfg.rows; i++)
						for (int j = 0; j < cfg.cols; j++)
						{
							const float d = static_cast<float>(std::hypot(0.5 + i - 0.5 * cfg.rows, 0.5 + j - 0.5 * cfg.cols) - 1);
							dist[i * cfg.cols + j] += static_cast<float>(std::pow(1.5f, -d));
						}
				}
				else
				{
					for (int i = 0; i < hw; i++)
						dist[i] = (board[i] != NONE) ? 0.0f : 1.0e-6f;
					const float tmp = 2.0f + rand_float();
					for (int p = 0; p < cfg.rows; p++)
						for (int q = 0; q < cfg.cols; q++)
							if (board[p * cfg.cols + q] != NONE)
								for (int i = 0; i < cfg.rows; i++)
									for (int j = 0; j < cfg.cols; j++)
										if (board[i * cfg.cols + j] == NONE)
										{
											const float d = static_cast<float>(std::hypot(static_cast<double>(i - p), static_cast<double>(j - q)) - 1);
											dist[i * cfg.cols + j] += static_cast<float>(std::pow(tmp, -d));
										}
				}
				// randomizeMove (:84-103)
				float r = 0.0f;
				for (int i = 0; i < hw; i++)
					r += dist[i];
				int pick;
				if (r == 0.0f)
					pick = rand_int(hw);
				else
				{
					r *= rand_float();
					float sum = 0.0f;
					pick = 0;
					for (; pick < hw; pick++)
					{
						sum += dist[pick];
						if (r < sum)
							break;
					}
					if (pick >= hw)
						pick = hw - 1;
					while (board[pick] != NONE && pick > 0)
						pick--; // unreachable in exact arithmetic; guards the rounding tail
				}
				const Move m(to_move, pick / cfg.cols, pick % cfg.cols);
				result.push_back(m);
				board[pick] = to_move;
				to_move = invert_sign(to_move);
			}
			if (result.empty())
				return result;
			if (get_outcome(cfg.rules, board.data(), cfg.rows, cfg.cols, result.back(), -1) == O_UNKNOWN)
				return result;
		}
	}
}
