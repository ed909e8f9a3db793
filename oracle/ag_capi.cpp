/*
 * oracle/ag_capi.cpp — TEST INFRASTRUCTURE ONLY.  extern "C" surface of the CPU oracle for ctypes (tests, smoke, and
 * the cpu_baseline leg of bench.py).
 */
#include "agoracle.hpp"
#include "ag_dataset.hpp"
#include "ag_noise.hpp"

#include <chrono>
#include <ctime>
#include <atomic>
#include <thread>

using namespace ago;

namespace
{
	GameConfig make_cfg(int rules, int rows, int cols)
	{
		GameConfig c;
		c.rules = static_cast<Rules>(rules);
		c.rows = rows;
		c.cols = cols;
		c.draw_after = rows * cols;
		return c;
	}
	struct GameHandle
	{
			Game game;
			std::vector<uint32_t> features;
			GameHandle(GameConfig c, const SearchConfig &sc) : game(c, sc) {}
	};
}

extern "C" {

struct AgoSearchConfig
{
		int max_batch_size;
		float exploration_constant;
		float exploration_scaling;
		int init_to;
		int max_children;
		float policy_expansion_threshold;
		float information_leak_threshold;
		int tss_max_positions;
		uint64_t tss_table_entries;
		int max_simulations;
		uint64_t zobrist_seed;
		int final_selector;
		int use_symmetries;
		uint64_t symmetry_seed;
		int noise_type;
		float noise_weight;
		uint64_t noise_seed;
};

static SearchConfig convert(const AgoSearchConfig *c)
{
	SearchConfig s;
	s.max_batch_size = c->max_batch_size;
	s.exploration_constant = c->exploration_constant;
	s.exploration_scaling = c->exploration_scaling;
	s.init_to = c->init_to;
	s.max_children = c->max_children;
	s.policy_expansion_threshold = c->policy_expansion_threshold;
	s.information_leak_threshold = c->information_leak_threshold;
	s.tss_max_positions = c->tss_max_positions;
	s.tss_table_entries = c->tss_table_entries;
	s.max_simulations = c->max_simulations;
	s.zobrist_seed = c->zobrist_seed;
	s.final_selector = c->final_selector;
	s.use_symmetries = c->use_symmetries;
	s.symmetry_seed = c->symmetry_seed;
	s.noise_type = c->noise_type;
	s.noise_weight = c->noise_weight;
	s.noise_seed = c->noise_seed;
	return s;
}

void ago_tables(int rules, uint8_t *types, uint8_t *half_open_3, uint8_t *threats)
{
	const Tables &t = Tables::get(static_cast<Rules>(rules));
	std::memcpy(types, t.pattern_types.data(), 1u << 20);
	std::memcpy(half_open_3, t.half_open_3.data(), 1u << 20);
	std::memcpy(threats, t.threats, 4096 * 2);
}
/* raw defence tables in the product's row order: rows 0-4 five, 5-8 open four, 9-14 double four; [row][sides][defender-1] */
void ago_defense_tables(int rules, uint16_t *out)
{
	const Tables &t = Tables::get(static_cast<Rules>(rules));
	for (int j = 0; j < 256; j++)
		for (int d = 0; d < 2; d++)
		{
			for (int i = 0; i < 5; i++)
				out[((0 + i) * 256 + j) * 2 + d] = t.five_defense[i][j][d];
			for (int i = 0; i < 4; i++)
				out[((5 + i) * 256 + j) * 2 + d] = t.open_four_defense[i][j][d];
			for (int i = 0; i < 6; i++)
				out[((9 + i) * 256 + j) * 2 + d] = t.double_four_defense[i][j][d];
		}
}
uint16_t ago_defensive_moves(int rules, uint32_t extended_pattern, int defender, int pattern_type)
{
	return Tables::get(static_cast<Rules>(rules)).defensive_moves(extended_pattern, static_cast<Sign>(defender), static_cast<PatternType>(pattern_type));
}
uint16_t ago_open_three_promotion_moves(uint32_t normal_pattern)
{
	return open_three_promotion_moves(normal_pattern);
}
uint16_t ago_score_op(uint16_t raw, int op)
{
	const Score s = Score::raw(raw);
	switch (op)
	{
		case 0: return invert_up(s).d;
		case 1: return invert_down(s).d;
		case 2: return negate(s).d;
		default: return raw;
	}
}
int ago_score_info(uint16_t raw, int *out_distance, float *out_value)
{
	const Score s = Score::raw(raw);
	*out_distance = s.distance();
	const Value v = s.to_value();
	out_value[0] = v.win;
	out_value[1] = v.draw;
	return (s.is_proven() ? 1 : 0) | (s.is_win() ? 2 : 0) | (s.is_loss() ? 4 : 0) | (s.is_draw() ? 8 : 0) | (s.is_unproven() ? 16 : 0) | (s.is_infinite() ? 32 : 0);
}
uint16_t ago_score_make(int pv, int eval)
{
	return Score(static_cast<ProvenValue>(pv), eval).d;
}
void ago_edge_update_value(float *win_draw, int *visits, float ew, float ed)
{
	Edge e;
	e.value = Value(win_draw[0], win_draw[1]);
	e.visits = *visits;
	e.update_value(Value(ew, ed));
	win_draw[0] = e.value.win;
	win_draw[1] = e.value.draw;
	*visits = e.visits;
}
void ago_node_update_value(float *win_draw, int *visits, float ew, float ed)
{
	Node n;
	n.value = Value(win_draw[0], win_draw[1]);
	n.visits = *visits;
	n.update_value(Value(ew, ed));
	win_draw[0] = n.value.win;
	win_draw[1] = n.value.draw;
	*visits = n.visits;
}
uint16_t ago_move_to_short(int sign, int row, int col)
{
	return Move(static_cast<Sign>(sign), row, col).to_short();
}

void ago_encode_features(int rules, int rows, int cols, const uint8_t *board, int sign_to_move, uint32_t *out)
{
	Calc calc(make_cfg(rules, rows, cols));
	calc.set_board(board, static_cast<Sign>(sign_to_move));
	encode_features(calc, out);
}
/* per cell: pattern types [cell][2][4], threats [cell][2]; lists: for each sign, for each threat type 0..9: count then (row, col) pairs */
int ago_pattern_state(int rules, int rows, int cols, const uint8_t *board, int sign_to_move, const uint16_t *moves, int n_moves, uint8_t *ptypes,
		uint8_t *threats, int16_t *lists, int lists_capacity)
{ // applies add_move for moves[i] (Move::to_short encoding); a move with sign 0 means "undo the most recent not-yet-undone move"
	Calc calc(make_cfg(rules, rows, cols));
	calc.set_board(board, static_cast<Sign>(sign_to_move));
	std::vector<Move> done;
	for (int i = 0; i < n_moves; i++)
	{
		const Move m = Move::from_short(moves[i]);
		if (m.sign == NONE)
		{
			calc.undo_move(done.back());
			done.pop_back();
		}
		else
		{
			calc.add_move(m);
			done.push_back(m);
		}
	}
	for (int i = 0; i < rows * cols; i++)
	{
		std::memcpy(ptypes + i * 8, calc.ptype[i], 8);
		threats[2 * i] = calc.threat[i][0];
		threats[2 * i + 1] = calc.threat[i][1];
	}
	int pos = 0;
	for (int s = 0; s < 2; s++)
		for (int t = 0; t < 10; t++)
		{
			const LocList &l = calc.hist[s][t];
			if (pos + 1 + 2 * l.size() > lists_capacity)
				return -1;
			lists[pos++] = static_cast<int16_t>(l.size());
			for (const Loc &x : l.v)
			{
				lists[pos++] = x.row;
				lists[pos++] = x.col;
			}
		}
	return pos;
}
int ago_outcome(int rules, int rows, int cols, const uint8_t *board, int sign, int row, int col, int draw_after)
{
	return get_outcome(static_cast<Rules>(rules), board, rows, cols, Move(static_cast<Sign>(sign), row, col), draw_after);
}
int ago_is_forbidden(int rows, int cols, const uint8_t *board, int sign, int row, int col)
{
	return is_forbidden_static(board, rows, cols, Move(static_cast<Sign>(sign), row, col)) ? 1 : 0;
}
/* returns the number of actions; moves as Move::to_short, scores raw; flags: bit0 must_defend, bit1 has_initiative, bit2 fully expanded */
int ago_movegen(int rules, int rows, int cols, const uint8_t *board, int sign_to_move, int mode, int draw_after, uint16_t *moves, uint16_t *scores,
		int *flags, uint16_t *result_score)
{
	GameConfig cfg = make_cfg(rules, rows, cols);
	if (draw_after > 0)
		cfg.draw_after = draw_after;
	Calc calc(cfg);
	calc.set_board(board, static_cast<Sign>(sign_to_move));
	MoveGen gen(cfg, calc);
	ActionStack stack;
	stack.data.resize(4096);
	ActionList list;
	list.stack = &stack;
	const Score s = gen.generate(list, static_cast<GenMode>(mode));
	for (int i = 0; i < list.size; i++)
	{
		moves[i] = list[i].move.to_short();
		scores[i] = list[i].score.d;
	}
	*flags = (list.must_defend ? 1 : 0) | (list.has_initiative ? 2 : 0) | (list.is_fully_expanded ? 4 : 0);
	*result_score = s.d;
	return list.size;
}

/* persistent solver (keeps its transposition table across calls like one AlphaBetaSearch per game) */
void* ago_solver_create(int rules, int rows, int cols, uint64_t table_entries, uint64_t zobrist_seed, int max_nodes)
{
	Solver *s = new Solver(make_cfg(rules, rows, cols), table_entries, zobrist_seed);
	s->max_nodes = max_nodes;
	return s;
}
/* transposition table alone (SharedHashTable.hpp): value packing as SharedTableData(bound, depth, score, move) */
void ago_solver_tt_insert(void *h, uint64_t lo, uint64_t hi, int bound, int depth, uint16_t score_raw, uint16_t move_short)
{
	Key128 k;
	k.lo = lo;
	k.hi = hi;
	const uint64_t value = static_cast<uint64_t>(bound) | (static_cast<uint64_t>(depth) << 8) | (static_cast<uint64_t>(score_raw) << 16)
			| (static_cast<uint64_t>(move_short) << 32);
	static_cast<Solver*>(h)->tt_insert(k, value);
}
uint64_t ago_solver_tt_seek(void *h, uint64_t lo, uint64_t hi)
{
	Key128 k;
	k.lo = lo;
	k.hi = hi;
	return static_cast<const Solver*>(h)->tt_seek(k);
}
/* line patterns of every cell after a move sequence (0 = undo): out[(cell * 4 + dir) * 2 + {0, 1}] = normal / extended pattern */
void ago_raw_patterns(int rows, int cols, const uint8_t *board, const uint16_t *moves, int n_moves, uint32_t *out)
{
	Calc calc(make_cfg(0, rows, cols));
	calc.set_board(board, CROSS);
	std::vector<Move> done;
	for (int i = 0; i < n_moves; i++)
	{
		if (moves[i] == 0)
		{
			calc.undo_move(done.back());
			done.pop_back();
		}
		else
		{
			const Move m = Move::from_short(moves[i]);
			calc.add_move(m);
			done.push_back(m);
		}
	}
	for (int r = 0; r < rows; r++)
		for (int c = 0; c < cols; c++)
			for (int d = 0; d < 4; d++)
			{
				out[((r * cols + c) * 4 + d) * 2 + 0] = calc.raw_pattern(r, c, d, 5);
				out[((r * cols + c) * 4 + d) * 2 + 1] = calc.raw_pattern(r, c, d, 6);
			}
}
int ago_is_straight_four(int rows, int cols, const uint8_t *board, int row, int col, int dir)
{
	Calc calc(make_cfg(2, rows, cols));
	calc.set_board(board, CROSS);
	return is_straight_four_at(calc, row, col, dir) ? 1 : 0;
}
/* threat lists alone (ThreatHistogram.hpp): ops[4 i ..] = (1 add / 0 remove, threat type, row, col) */
int ago_threat_histogram(const int *ops, int n_ops, int16_t *out)
{
	LocList lists[10];
	for (int i = 0; i < n_ops; i++)
	{
		const int t = ops[4 * i + 1];
		if (t == 0)
			continue; // ThreatType::NONE is never stored
		const Loc l(ops[4 * i + 2], ops[4 * i + 3]);
		if (ops[4 * i])
			lists[t].add(l);
		else
			lists[t].remove(l);
	}
	int pos = 0;
	for (int t = 0; t < 10; t++)
	{
		out[pos++] = static_cast<int16_t>(lists[t].size());
		for (const Loc &l : lists[t].v)
		{
			out[pos++] = l.row;
			out[pos++] = l.col;
		}
	}
	return pos;
}
void ago_solver_destroy(void *h)
{
	delete static_cast<Solver*>(h);
}
void ago_solver_zobrist(void *h, uint64_t *out)
{
	Solver *s = static_cast<Solver*>(h);
	for (size_t i = 0; i < s->zobrist.size(); i++)
	{
		out[2 * i] = s->zobrist[i].lo;
		out[2 * i + 1] = s->zobrist[i].hi;
	}
}
void ago_solver_new_generation(void *h)
{
	static_cast<Solver*>(h)->increase_generation();
}
int ago_solver_solve(void *h, const uint8_t *board, int sign_to_move, uint32_t *features, uint16_t *moves, uint16_t *scores, int *flags,
		uint16_t *result_score, int *nodes)
{
	Solver *s = static_cast<Solver*>(h);
	Solver::Output out;
	s->solve(board, static_cast<Sign>(sign_to_move), features, out);
	for (size_t i = 0; i < out.actions.size(); i++)
	{
		moves[i] = out.actions[i].move.to_short();
		scores[i] = out.actions[i].score.d;
	}
	*flags = out.must_defend ? 1 : 0;
	*result_score = out.score.d;
	*nodes = out.nodes;
	return static_cast<int>(out.actions.size());
}

/* deterministic stand-in evaluator: a pure function of the feature words (used where the NN itself is not under test) */
void ago_fake_eval(int count, int hw, const uint32_t *features, float *policy, float *value)
{
	for (int b = 0; b < count; b++)
	{
		const uint32_t *f = features + static_cast<size_t>(b) * hw;
		float *p = policy + static_cast<size_t>(b) * hw;
		uint32_t acc = 2166136261u;
		float sum = 0.0f;
		for (int i = 0; i < hw; i++)
		{
			uint32_t h = (f[i] ^ (static_cast<uint32_t>(i) * 2654435761u)) * 2246822519u;
			h ^= h >> 15;
			acc = (acc ^ f[i]) * 16777619u;
			float w = 0.0f;
			if (f[i] & 1u)
			{
				const int threat_bits = __builtin_popcount(f[i] >> 8);
				w = 1.0f + static_cast<float>(h % 1024u) * (1.0f / 1024.0f) + 2.0f * threat_bits;
			}
			p[i] = w;
			sum += w;
		}
		if (sum > 0.0f)
		{
			const float inv = 1.0f / sum;
			for (int i = 0; i < hw; i++)
				p[i] *= inv;
		}
		acc ^= acc >> 13;
		value[2 * b] = 0.25f + 0.5f * static_cast<float>(acc % 4096u) * (1.0f / 4096.0f);
		value[2 * b + 1] = 0.125f;
	}
}

int ago_prepare_opening(int rules, int rows, int cols, uint32_t seed, uint16_t *moves)
{
	const std::vector<Move> o = prepare_opening(make_cfg(rules, rows, cols), seed);
	for (size_t i = 0; i < o.size(); i++)
		moves[i] = o[i].to_short();
	return static_cast<int>(o.size());
}

void* ago_game_create(int rules, int rows, int cols, const AgoSearchConfig *cfg)
{
	return new GameHandle(make_cfg(rules, rows, cols), convert(cfg));
}
/* same with GameConfig::draw_after set (a game is a draw once that many stones are on the board, rules.cpp:129-131) */
void* ago_game_create_ex(int rules, int rows, int cols, int draw_after, const AgoSearchConfig *cfg)
{
	GameConfig c = make_cfg(rules, rows, cols);
	if (draw_after > 0)
		c.draw_after = draw_after;
	return new GameHandle(c, convert(cfg));
}
void ago_game_destroy(void *h)
{
	delete static_cast<GameHandle*>(h);
}
/* root noise alone (oracle/ag_noise.hpp): out[i] = noisy prior; also the two series it is built on */
void ago_root_noise(int type, float weight, uint64_t seed, int serial, int move_number, int n, const float *priors, float *out)
{
	make_root_noise(type, weight, seed, serial, move_number, n, [&](int i) { return priors[i]; }, out);
}
double ago_det_log(double x)
{
	return det_log(x);
}
double ago_det_exp(double x)
{
	return det_exp(x);
}
/* tournament search on one tree: `count` search threads (Search objects with their own solver and table) in lock-step; before ago_game_begin */
void ago_game_set_search_threads(void *h, int count)
{
	static_cast<GameHandle*>(h)->game.set_search_threads(count);
}
void ago_game_set_serial(void *h, int serial)
{
	static_cast<GameHandle*>(h)->game.serial = serial;
}
/* out[r * n + c] = in[source of (r, c) under symmetry s], with the feature direction bits shuffled when `features` != 0 */
void ago_apply_symmetry(int n, int s, int features, const uint32_t *in, uint32_t *out)
{
	for (int r = 0; r < n; r++)
		for (int c = 0; c < n; c++)
		{
			int sr, sc;
			symmetry_source(s, n, r, c, sr, sc);
			out[r * n + c] = features ? shuffle_feature_directions(in[sr * n + sc], s) : in[sr * n + sc];
		}
}
int ago_inverse_symmetry(int s)
{
	return inverse_symmetry(s);
}
void ago_game_begin(void *h, const uint16_t *opening, int n)
{
	std::vector<Move> o;
	for (int i = 0; i < n; i++)
		o.push_back(Move::from_short(opening[i]));
	static_cast<GameHandle*>(h)->game.begin(o);
}
/* ---- evaluation match: one handle per Player (own tree + solver), the game is mirrored by external moves ---- */
void ago_game_set_force_expand_root(void *h, int value)
{ // UnifiedGenerator's forceExpandRoot: 1 self-play (default), 0 evaluation Player (Player.cpp:111)
	GameHandle *g = static_cast<GameHandle*>(h);
	g->game.scfg.force_expand_root = value;
	g->game.tree.scfg.force_expand_root = value;
	g->game.search.scfg.force_expand_root = value;
}
int ago_game_record_flags(void *h, int index)
{
	return static_cast<GameHandle*>(h)->game.records[index].root_flags;
}
void ago_game_set_policy_temperature(void *h, float value)
{
	GameHandle *g = static_cast<GameHandle*>(h);
	g->game.scfg.policy_temperature = value;
	g->game.tree.scfg.policy_temperature = value;
	g->game.search.scfg.policy_temperature = value;
}
void ago_game_match_begin(void *h, const uint16_t *opening, int n)
{
	std::vector<Move> o;
	for (int i = 0; i < n; i++)
		o.push_back(Move::from_short(opening[i]));
	static_cast<GameHandle*>(h)->game.match_begin(o);
}
void ago_game_take_turn(void *h)
{
	static_cast<GameHandle*>(h)->game.take_turn();
}
void ago_game_external_move(void *h, int move)
{
	static_cast<GameHandle*>(h)->game.external_move(Move::from_short(static_cast<uint16_t>(move)));
}
int ago_game_last_move(void *h)
{
	const Game &g = static_cast<GameHandle*>(h)->game;
	return g.moves.empty() ? -1 : static_cast<int>(g.moves.back().to_short());
}
int ago_game_sign_to_move(void *h)
{
	return static_cast<int>(static_cast<GameHandle*>(h)->game.sign_to_move);
}
/* returns number of positions to evaluate; features copied to out (capacity in positions) */
int ago_game_step_select(void *h, uint32_t *features_out, int capacity)
{
	GameHandle *g = static_cast<GameHandle*>(h);
	const int n = g->game.step_select(g->features);
	if (n > capacity)
		return -1;
	std::memcpy(features_out, g->features.data(), g->features.size() * sizeof(uint32_t));
	return n;
}
/* double-buffered tournament search (SearchThread::asynchronous_run): one loop iteration, then the network's answer for its batch */
int ago_game_async_step(void *h, uint32_t *features_out, int capacity)
{
	GameHandle *g = static_cast<GameHandle*>(h);
	const int n = g->game.async_step(g->features);
	if (n > capacity)
		return -1;
	std::memcpy(features_out, g->features.data(), g->features.size() * sizeof(uint32_t));
	return n;
}
void ago_game_async_provide(void *h, const float *policy, const float *value)
{
	static_cast<GameHandle*>(h)->game.async_provide(policy, value);
}
int ago_game_step_expand(void *h, const float *policy, const float *value)
{
	return static_cast<GameHandle*>(h)->game.step_expand(policy, value);
}
int ago_game_step_expand_q(void *h, const float *policy, const float *value, const float *action_values)
{
	return static_cast<GameHandle*>(h)->game.step_expand(policy, value, action_values);
}
int ago_game_outcome(void *h)
{
	return static_cast<GameHandle*>(h)->game.outcome;
}
int ago_game_num_records(void *h)
{
	return static_cast<int>(static_cast<GameHandle*>(h)->game.records.size());
}
/* record i: move (short), root visits, root value, root score, and per edge: move, visits, prior bits, win bits, draw bits, score */
int ago_game_record(void *h, int index, uint16_t *move, int *root_visits, float *root_value, uint16_t *root_score, uint16_t *edge_moves,
		int32_t *edge_visits, float *edge_prior, float *edge_value, uint16_t *edge_score, int capacity)
{
	const Game::MoveRecord &r = static_cast<GameHandle*>(h)->game.records.at(index);
	*move = r.move.to_short();
	*root_visits = r.root_visits;
	root_value[0] = r.root_value.win;
	root_value[1] = r.root_value.draw;
	*root_score = r.root_score.d;
	const int n = static_cast<int>(r.root_edges.size());
	if (n > capacity)
		return -1;
	for (int i = 0; i < n; i++)
	{
		edge_moves[i] = r.root_edges[i].move.to_short();
		edge_visits[i] = r.root_edges[i].visits;
		edge_prior[i] = r.root_edges[i].prior;
		edge_value[2 * i] = r.root_edges[i].value.win;
		edge_value[2 * i + 1] = r.root_edges[i].value.draw;
		edge_score[i] = r.root_edges[i].score.d;
	}
	return n;
}
/* ---- self-play record sink (ag_dataset.hpp) ---- */
} /* extern "C" */
namespace
{
	template<typename F>
	uint32_t lowfp_dispatch(int format, F f)
	{
		switch (format)
		{
			case 0: return f(score_format());
			case 1: return f(visit_format());
			case 2: return f(policy_format());
			default: return f(fp16_format());
		}
	}
	/* SearchDataPack(const Node&, board) (data_packs.cpp:24-43) from a root snapshot; only the NUMBER of stones of the board matters to format 201 */
	SearchDataPack make_pack(int rows, int cols, int stones, int n_edges, const uint16_t *edge_moves, const int32_t *edge_visits, const float *edge_prior,
			const float *edge_value, const uint16_t *edge_score, uint16_t root_score, int root_flags)
	{
		SearchDataPack pack(rows, cols);
		for (int i = 0; i < stones && i < rows * cols; i++)
			pack.board[i] = CROSS;
		for (int i = 0; i < n_edges; i++)
		{
			const Move m = Move::from_short(edge_moves[i]);
			const int cell = m.row * cols + m.col;
			pack.policy_prior[cell] = edge_prior[i];
			pack.visit_count[cell] = edge_visits[i];
			pack.action_values[cell] = Value(edge_value[2 * i], edge_value[2 * i + 1]);
			pack.action_scores[cell] = Score::raw(edge_score[i]);
		}
		pack.minimax_score = Score::raw(root_score);
		pack.flags = static_cast<uint16_t>(root_flags);
		return pack;
	}
	SearchDataStorage_v201 storage_of_record(const Game &g, const Game::MoveRecord &r)
	{
		SearchDataPack pack(g.cfg.rows, g.cfg.cols);
		for (int i = 0; i < r.stones; i++)
			pack.board[i] = CROSS;
		for (const Edge &e : r.root_edges)
		{
			const int cell = e.move.row * g.cfg.cols + e.move.col;
			pack.policy_prior[cell] = e.prior;
			pack.visit_count[cell] = e.visits;
			pack.action_values[cell] = e.value;
			pack.action_scores[cell] = e.score;
		}
		pack.minimax_value = r.root_value;
		pack.minimax_score = r.root_score;
		pack.flags = static_cast<uint16_t>(r.root_flags);
		SearchDataStorage_v201 s;
		s.load_from(pack);
		return s;
	}
}
extern "C" {
uint32_t ago_lowfp_to_lowp(int format, float x)
{
	return lowfp_dispatch(format, [x](auto f) { return decltype(f)::to_lowp(x); });
}
float ago_lowfp_to_fp32(int format, uint32_t code)
{
	float out = 0.0f;
	lowfp_dispatch(format, [&](auto f) { out = decltype(f)::to_fp32(code); return 0u; });
	return out;
}
float ago_lowfp_max(int format)
{
	float out = 0.0f;
	lowfp_dispatch(format, [&](auto f) { out = decltype(f)::max(); return 0u; });
	return out;
}
int ago_score_to_int8(uint16_t raw)
{
	return score_to_int8(Score::raw(raw));
}
uint16_t ago_int8_to_score(int code)
{
	return int8_to_score(static_cast<uint8_t>(code)).d;
}
/* SearchDataStorage_v201::loadFrom + serialize of one root snapshot; returns the number of bytes (-1: does not fit) */
int ago_sample_v201_pack(int rows, int cols, int stones, int n_edges, const uint16_t *edge_moves, const int32_t *edge_visits, const float *edge_prior,
		const float *edge_value, const uint16_t *edge_score, uint16_t root_score, int root_flags, uint8_t *out, int capacity)
{
	const SearchDataPack pack = make_pack(rows, cols, stones, n_edges, edge_moves, edge_visits, edge_prior, edge_value, edge_score, root_score, root_flags);
	SearchDataStorage_v201 s;
	s.load_from(pack);
	std::vector<uint8_t> bytes;
	s.serialize(bytes);
	if (static_cast<int>(bytes.size()) > capacity)
		return -1;
	std::memcpy(out, bytes.data(), bytes.size());
	return static_cast<int>(bytes.size());
}
/* parse + storeTo: visits int[hw], prior float[hw], value float[hw][2], score u16[hw], header int[3] = (minimax score, move number, flags),
 * minimax value float[2]; returns the number of bytes consumed */
int ago_sample_v201_unpack(const uint8_t *bytes, int rows, int cols, int32_t *visits, float *prior, float *value, uint16_t *score, int *header, float *minimax_value)
{
	SearchDataStorage_v201 s;
	const size_t used = s.parse(bytes, 0);
	SearchDataPack pack(rows, cols);
	s.store_to(pack);
	for (int i = 0; i < rows * cols; i++)
	{
		visits[i] = pack.visit_count[i];
		prior[i] = pack.policy_prior[i];
		value[2 * i] = pack.action_values[i].win;
		value[2 * i + 1] = pack.action_values[i].draw;
		score[i] = pack.action_scores[i].d;
	}
	header[0] = pack.minimax_score.d;
	header[1] = s.move_number;
	header[2] = pack.flags;
	minimax_value[0] = pack.minimax_value.win;
	minimax_value[1] = pack.minimax_value.draw;
	return static_cast<int>(used);
}
/* record i of the oracle game as SearchDataStorage_v201 bytes */
int ago_game_record_v201(void *h, int index, uint8_t *out, int capacity)
{
	const Game &g = static_cast<GameHandle*>(h)->game;
	std::vector<uint8_t> bytes;
	storage_of_record(g, g.records.at(index)).serialize(bytes);
	if (static_cast<int>(bytes.size()) > capacity)
		return -1;
	std::memcpy(out, bytes.data(), bytes.size());
	return static_cast<int>(bytes.size());
}
/* the whole game as GameDataStorage::serialize (format 201) bytes: what GameGenerator hands to GeneratorManager::addToBuffer when the
 * game is over (GameGenerator.cpp:104-114): samples, ALL moves of the game (opening included, Game::getMoves), outcome, rows, cols */
int ago_game_storage_v201(void *h, uint8_t *out, int capacity)
{
	const Game &g = static_cast<GameHandle*>(h)->game;
	std::vector<SearchDataStorage_v201> samples;
	for (const Game::MoveRecord &r : g.records)
		samples.push_back(storage_of_record(g, r));
	std::vector<uint16_t> played;
	for (const Move &m : g.moves)
		played.push_back(m.to_short());
	std::vector<uint8_t> bytes;
	serialize_game_v201(samples, played, static_cast<int>(g.outcome), g.cfg.rows, g.cfg.cols, bytes);
	if (static_cast<int>(bytes.size()) > capacity)
		return -1;
	std::memcpy(out, bytes.data(), bytes.size());
	return static_cast<int>(bytes.size());
}
/* stats: nodes, nn_evals, leaks, duplicates, proven, wasted, solver_nodes, select_levels, select_edges, tree nodes, tree edges */
void ago_game_stats(void *h, uint64_t *out)
{
	GameHandle *g = static_cast<GameHandle*>(h);
	const Stats &s = g->game.search.stats;
	out[0] = s.nodes;
	out[1] = s.nn_evals;
	out[2] = s.leaks;
	out[3] = s.duplicates;
	out[4] = s.proven;
	out[5] = s.wasted;
	out[6] = s.solver_nodes;
	out[7] = g->game.tree.stats.select_levels;
	out[8] = g->game.tree.stats.select_edges;
	out[9] = g->game.tree.nodes.size();
	out[10] = g->game.tree.edges.size();
}
/* current root snapshot (for step-by-step parity): returns number of edges */
int ago_game_root(void *h, int *root_visits, float *root_value, uint16_t *root_score, uint16_t *edge_moves, int32_t *edge_visits, float *edge_prior,
		float *edge_value, uint16_t *edge_score, uint16_t *edge_flag_vl, int capacity)
{
	Game &g = static_cast<GameHandle*>(h)->game;
	if (g.tree.root < 0)
		return 0;
	const Node &r = g.tree.nodes[g.tree.root];
	*root_visits = r.visits;
	root_value[0] = r.value.win;
	root_value[1] = r.value.draw;
	*root_score = r.score.d;
	if (r.n_edges > capacity)
		return -1;
	for (int i = 0; i < r.n_edges; i++)
	{
		const Edge &e = g.tree.edges[r.edge_begin + i];
		edge_moves[i] = e.move.to_short();
		edge_visits[i] = e.visits;
		edge_prior[i] = e.prior;
		edge_value[2 * i] = e.value.win;
		edge_value[2 * i + 1] = e.value.draw;
		edge_score[i] = e.score.d;
		edge_flag_vl[i] = e.flag_vl;
	}
	return r.n_edges;
}

/* Tree::getMovesLeft / getMaximumDepth / hasAllMovesProven / hasSingleMove / hasSingleNonLosingMove / getNodeCount (Tree.cpp:173-224):
 * out_f[0] = moves left of the root, out_i = { max depth, all moves proven, single move, single non-losing move, stored nodes } */
void ago_game_tree_info(void *h, float *out_f, int *out_i)
{
	Game &g = static_cast<GameHandle*>(h)->game;
	out_f[0] = 0.0f;
	out_i[0] = g.tree.max_depth;
	out_i[1] = out_i[2] = out_i[3] = 0;
	out_i[4] = static_cast<int>(g.tree.nodes.size());
	if (g.tree.root < 0)
		return;
	const Node &r = g.tree.nodes[g.tree.root];
	out_f[0] = r.moves_left; // Tree.cpp:350 (the root's value after the last backup; the device reads the root record)
	bool all_proven = true;
	int non_losing = 0;
	for (int i = 0; i < r.n_edges; i++)
	{
		const Edge &e = g.tree.edges[r.edge_begin + i];
		all_proven = all_proven && e.score.is_proven();
		non_losing += e.score.is_loss() ? 0 : 1;
	}
	out_i[1] = all_proven ? 1 : 0;
	out_i[2] = (r.n_edges == 1) ? 1 : 0;
	out_i[3] = (non_losing == 1) ? 1 : 0;
}

/*
 * CPU baseline: plays `games_per_thread` self-play games on each of `threads` host threads with the stand-in evaluator
 * (network cost = 0), for at most `max_seconds`; returns evaluated nodes, completed games and moves made.
 */
static void cpu_baseline_impl(int rules, int rows, int cols, const AgoSearchConfig *cfg, int threads, int games_per_thread, double max_seconds, uint64_t *out_nodes,
		uint64_t *out_games, uint64_t *out_moves, double *out_seconds, uint64_t *out_stats, double *out_thread_cpu_seconds);
void ago_cpu_baseline(int rules, int rows, int cols, const AgoSearchConfig *cfg, int threads, int games_per_thread, double max_seconds, uint64_t *out_nodes,
		uint64_t *out_games, uint64_t *out_moves, double *out_seconds, uint64_t *out_stats)
{
	cpu_baseline_impl(rules, rows, cols, cfg, threads, games_per_thread, max_seconds, out_nodes, out_games, out_moves, out_seconds, out_stats, nullptr);
}
/* the same with the CPU time every worker thread consumed inside the timed region (CLOCK_THREAD_CPUTIME_ID): out_thread_cpu_seconds[threads].
 * Σ cpu / (threads x wall) well below 1 means the threads did not get the cores they were started on (a cgroup CPU quota, oversubscription) */
void ago_cpu_baseline_ex(int rules, int rows, int cols, const AgoSearchConfig *cfg, int threads, int games_per_thread, double max_seconds, uint64_t *out_nodes,
		uint64_t *out_games, uint64_t *out_moves, double *out_seconds, uint64_t *out_stats, double *out_thread_cpu_seconds)
{
	cpu_baseline_impl(rules, rows, cols, cfg, threads, games_per_thread, max_seconds, out_nodes, out_games, out_moves, out_seconds, out_stats, out_thread_cpu_seconds);
}
static void cpu_baseline_impl(int rules, int rows, int cols, const AgoSearchConfig *cfg, int threads, int games_per_thread, double max_seconds, uint64_t *out_nodes,
		uint64_t *out_games, uint64_t *out_moves, double *out_seconds, uint64_t *out_stats, double *out_thread_cpu_seconds)
{
	const GameConfig gc = make_cfg(rules, rows, cols);
	const SearchConfig sc = convert(cfg);
	Tables::get(gc.rules);
	std::vector<uint64_t> nodes(threads, 0), games(threads, 0), moves(threads, 0);
	std::vector<std::vector<uint64_t>> st(threads, std::vector<uint64_t>(9, 0));
	// the clock starts when every thread has built its Game (tree, 64 MB solver table ...): set-up is not part of the sample
	std::atomic<int> ready(0);
	std::atomic<bool> go(false);
	std::chrono::steady_clock::time_point t0;
	auto worker = [&](int tid)
	{
		const int hw = rows * cols;
		std::vector<uint32_t> features;
		std::vector<float> policy(static_cast<size_t>(sc.max_batch_size) * hw), value(2 * sc.max_batch_size);
		Game game(gc, sc);
		game.begin(prepare_opening(gc, 1000u * tid)); // touches the tables once
		ready.fetch_add(1);
		while (!go.load(std::memory_order_acquire))
			std::this_thread::yield();
		timespec cpu0;
		clock_gettime(CLOCK_THREAD_CPUTIME_ID, &cpu0);
		for (int gi = 0; gi < games_per_thread; gi++)
		{
			game.begin(prepare_opening(gc, 1000u * tid + gi));
			while (!game.is_over())
			{
				const int n = game.step_select(features);
				ago_fake_eval(n, hw, features.data(), policy.data(), value.data());
				moves[tid] += game.step_expand(policy.data(), value.data());
				const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
				if (el > max_seconds)
					break;
			}
			if (game.is_over())
				games[tid]++;
			const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			if (el > max_seconds)
				break;
		}
		timespec cpu1;
		clock_gettime(CLOCK_THREAD_CPUTIME_ID, &cpu1);
		if (out_thread_cpu_seconds != nullptr)
			out_thread_cpu_seconds[tid] = static_cast<double>(cpu1.tv_sec - cpu0.tv_sec) + 1.0e-9 * static_cast<double>(cpu1.tv_nsec - cpu0.tv_nsec);
		nodes[tid] = game.search.stats.nodes;
		const Stats &s = game.search.stats;
		const uint64_t vals[9] = { s.nodes, s.nn_evals, s.leaks, s.duplicates, s.proven, s.wasted, s.solver_nodes, game.tree.stats.select_levels, game.tree.stats.select_edges };
		for (int k = 0; k < 9; k++)
			st[tid][k] = vals[k];
	};
	std::vector<std::thread> pool;
	for (int t = 0; t < threads; t++)
		pool.emplace_back(worker, t);
	while (ready.load() < threads)
		std::this_thread::sleep_for(std::chrono::milliseconds(1));
	t0 = std::chrono::steady_clock::now();
	go.store(true, std::memory_order_release);
	for (auto &t : pool)
		t.join();
	*out_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	*out_nodes = *out_games = *out_moves = 0;
	for (int k = 0; k < 9; k++)
		out_stats[k] = 0;
	for (int t = 0; t < threads; t++)
	{
		*out_nodes += nodes[t];
		*out_games += games[t];
		*out_moves += moves[t];
		for (int k = 0; k < 9; k++)
			out_stats[k] += st[t][k];
	}
}


/* ---- the hashing formulas and the action-list mechanics with CALLER-SUPPLIED keys / scripts: tests feed them what the compiled
 * reference produced (oracle/ref_driver.cpp: ref_full_zobrist, ref_fast_zobrist, ref_action_list_script) ---- */
uint64_t ago_full_zobrist_with_keys(const uint64_t *keys, int cells, const uint8_t *board, int sign_to_move)
{
	std::vector<Sign> b(cells);
	for (int i = 0; i < cells; i++)
		b[i] = static_cast<Sign>(board[i]);
	return full_zobrist_hash(keys, b.data(), cells, static_cast<Sign>(sign_to_move));
}
/* keys: [2 * cells][2] (lo, hi); out: the hash of the board (2 words), then the hash after every move of `moves` (2 words each) */
void ago_fast_zobrist_with_keys(const uint64_t *keys, int rows, int cols, const uint8_t *board, const uint16_t *moves, int n_moves, uint64_t *out)
{
	const int cells = rows * cols;
	std::vector<Key128> k(2 * cells);
	for (int i = 0; i < 2 * cells; i++)
	{
		k[i].lo = keys[2 * i];
		k[i].hi = keys[2 * i + 1];
	}
	std::vector<Sign> b(cells);
	for (int i = 0; i < cells; i++)
		b[i] = static_cast<Sign>(board[i]);
	Key128 h = fast_zobrist_hash(k.data(), b.data(), cells);
	out[0] = h.lo;
	out[1] = h.hi;
	for (int i = 0; i < n_moves; i++)
	{
		fast_zobrist_update(k.data(), cols, h, Move::from_short(moves[i]));
		out[2 * (1 + i)] = h.lo;
		out[2 * (1 + i) + 1] = h.hi;
	}
}
/* the script format of ref_bitmask_script (utils/BitMask.hpp on the oracle's plain words) */
int ago_bitmask_script(const int *ops, int n_ops, int rows, int cols, uint32_t *out, int capacity)
{
	uint16_t m = 0, last = 0;
	uint32_t g[32] = { 0 }, h[32] = { 0 };
	(void) rows;
	(void) cols;
	int at = 0, pos = 0;
	for (int k = 0; k < n_ops; k++)
	{
		const int op = ops[at++];
		switch (op)
		{
			case 1: mask_set(m, ops[at], ops[at + 1] != 0); at += 2; break;
			case 2: m = mask_flip16(m, ops[at++]); break;
			case 3: m = static_cast<uint16_t>(m << ops[at++]); break;
			case 4: m = static_cast<uint16_t>(m >> ops[at++]); break;
			case 5: last = static_cast<uint16_t>(ops[at++]); m &= last; break;
			case 6: last = static_cast<uint16_t>(ops[at++]); m |= last; break;
			case 7: mask_set(g[ops[at]], ops[at + 1], ops[at + 2] != 0); at += 3; break;
			case 8: mask_set(h[ops[at]], ops[at + 1], ops[at + 2] != 0); at += 3; break;
			case 9: for (int r = 0; r < 32; r++) g[r] &= h[r]; break;
			case 10: for (int r = 0; r < 32; r++) g[r] |= h[r]; break;
			default: { const uint32_t v = (ops[at++] != 0) ? 0xFFFFFFFFu : 0u; for (int r = 0; r < 32; r++) g[r] = v; } break;
		}
		if (pos + 34 > capacity)
			return -1;
		out[pos++] = m;
		out[pos++] = (m == last) ? 1u : 0u;
		for (int r = 0; r < 32; r++)
			out[pos++] = g[r];
	}
	return pos;
}
/* the script format of ref_action_list_script */
int ago_action_list_script(const int *ops, int n_ops, int *out, int capacity)
{
	ActionStack stack;
	stack.data.resize(4096);
	std::vector<ActionList> lists(1);
	lists[0].stack = &stack;
	lists[0].base = 0;
	int pos = 0, at = 0;
	for (int k = 0; k < n_ops; k++)
	{
		const int op = ops[at++];
		ActionList &top = lists.back();
		if (op == 1)
		{
			top.add(Move::from_short(static_cast<uint16_t>(ops[at])), Score::raw(static_cast<uint16_t>(ops[at + 1])), ops[at + 2]);
			at += 3;
		}
		else if (op == 2)
		{ // ActionList(stack, parent, i) (ActionList.hpp:337-343), as Solver::recursive_solve opens a child
			at++;
			ActionList next;
			next.stack = &stack;
			next.base = stack.offset;
			next.distance_from_root = top.distance_from_root + 1;
			lists.push_back(next);
		}
		else if (op == 3)
		{
			lists.back().release();
			lists.pop_back();
		}
		else
		{
			move_closer_to_front(top, Move::from_short(static_cast<uint16_t>(ops[at])), ops[at + 1]);
			at += 2;
		}
		if (pos + 4 > capacity)
			return -1;
		out[pos++] = static_cast<int>(stack.offset);
		out[pos++] = static_cast<int>(stack.max_offset);
		out[pos++] = lists.back().size;
		out[pos++] = lists.back().distance_from_root;
	}
	for (const ActionList &l : lists)
		for (int i = 0; i < l.size; i++)
		{
			if (pos + 2 > capacity)
				return -1;
			out[pos++] = l[i].move.to_short();
			out[pos++] = l[i].score.d;
		}
	return pos;
}
} /* extern "C" */
