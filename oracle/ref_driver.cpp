/*
 * oracle/ref_driver.cpp — TEST INFRASTRUCTURE ONLY.
 *
 * Thin extern "C" driver over the subset of the reference that compiles from its own sources WITHOUT the absent
 * MinML library (no stand-in headers are written): src/patterns/{PatternTable,PatternClassifier,ThreatTable,
 * DefensiveMoveTable}.cpp, src/game/Move.cpp, src/search/{Score,Value,ZobristHashing}.cpp,
 * src/search/monte_carlo/{Edge,Node}.cpp, src/utils/random.cpp, and the header-only utils/augmentations.hpp,
 * search/alpha_beta/SharedHashTable.hpp, search/alpha_beta/ActionList.hpp, patterns/RawPatternCalculator.hpp, patterns/ThreatHistogram.hpp,
 * utils/low_precision.hpp
 * (the LowFP formats of the dataset quantiser; the quantiser itself, src/dataset/SearchDataStorage.cpp, includes
 * <minml/utils/serialization.hpp> and is unbuildable here).
 * Built by oracle/Makefile into oracle/_ref/libagref.so straight from /root/reference; used by tests to pin the
 * restatement in oracle/ (tables, score algebra, struct layouts) and to generate tests/golden fixtures.
 * Everything that includes utils/configs.hpp (PatternCalculator, rules, MoveGenerator, AlphaBetaSearch, Tree, Search,
 * NNInputFeatures ...) pulls in <minml/...> and is therefore unbuildable here.
 */
#include <alphagomoku/patterns/PatternTable.hpp>
#include <alphagomoku/patterns/ThreatTable.hpp>
#include <alphagomoku/patterns/DefensiveMoveTable.hpp>
#include <alphagomoku/search/Score.hpp>
#include <alphagomoku/search/Value.hpp>
#include <alphagomoku/search/monte_carlo/Edge.hpp>
#include <alphagomoku/search/monte_carlo/Node.hpp>
#include <alphagomoku/game/rules.hpp>
#include <alphagomoku/utils/augmentations.hpp>
#include <alphagomoku/search/alpha_beta/SharedHashTable.hpp>
#include <alphagomoku/patterns/RawPatternCalculator.hpp>
#include <alphagomoku/patterns/ThreatHistogram.hpp>
#include <alphagomoku/utils/low_precision.hpp>
#include <alphagomoku/search/ZobristHashing.hpp>
#include <alphagomoku/search/alpha_beta/ActionList.hpp>
#include <alphagomoku/utils/matrix.hpp>
#include <alphagomoku/utils/BitMask.hpp>
#include <memory>
#include <vector>

#include <cstdint>
#include <cstring>

using namespace ag;

extern "C" {

/* out[i] = PatternEncoding byte (cross | circle << 4) for the 20-bit narrowed index i (center removed) */
void ref_pattern_table(int rules, uint8_t *types, uint8_t *half_open_3)
{
	const PatternTable &t = PatternTable::get(static_cast<GameRules>(rules));
	for (uint32_t i = 0; i < (1u << 20); i++)
	{
		const uint32_t expanded = (i & 1023u) | ((i & 1047552u) << 2u);
		const PatternEncoding e = t.getPatternType(NormalPattern(expanded));
		types[i] = static_cast<uint8_t>(e.forCross()) | (static_cast<uint8_t>(e.forCircle()) << 4);
		half_open_3[i] = (t.isHalfOpenThree(NormalPattern(expanded), Sign::CROSS) ? 1 : 0) | (t.isHalfOpenThree(NormalPattern(expanded), Sign::CIRCLE) ? 2 : 0);
	}
}
/* out[2*i] = threat for cross, out[2*i+1] = threat for circle; i = h | v<<3 | d<<6 | a<<9 */
void ref_threat_table(int rules, uint8_t *out)
{
	const ThreatTable &t = ThreatTable::get(static_cast<GameRules>(rules));
	for (int i = 0; i < 4096; i++)
	{
		DirectionGroup<PatternType> g;
		g.horizontal = static_cast<PatternType>(i & 7);
		g.vertical = static_cast<PatternType>((i >> 3) & 7);
		g.diagonal = static_cast<PatternType>((i >> 6) & 7);
		g.antidiagonal = static_cast<PatternType>((i >> 9) & 7);
		out[2 * i] = static_cast<uint8_t>(t.getThreat<Sign::CROSS>(g));
		out[2 * i + 1] = static_cast<uint8_t>(t.getThreat<Sign::CIRCLE>(g));
	}
}
uint16_t ref_defensive_moves(int rules, uint32_t extended_pattern, int defender_sign, int pattern_type)
{
	const DefensiveMoveTable &t = DefensiveMoveTable::get(static_cast<GameRules>(rules));
	return t.getMoves(ExtendedPattern(extended_pattern), static_cast<Sign>(defender_sign), static_cast<PatternType>(pattern_type)).raw();
}
uint16_t ref_open_three_promotion_moves(uint32_t normal_pattern)
{
	return getOpenThreePromotionMoves(NormalPattern(normal_pattern)).raw();
}
/* score algebra: op 0 invert_up, 1 invert_down, 2 negate, 3 +1, 4 -1, 5 increaseDistance, 6 decreaseDistance */
uint16_t ref_score_op(uint16_t raw, int op)
{
	Score s = Score::from_short(raw);
	switch (op)
	{
		case 0: return Score::to_short(invert_up(s));
		case 1: return Score::to_short(invert_down(s));
		case 2: return Score::to_short(-s);
		case 3: return Score::to_short(s + 1);
		case 4: return Score::to_short(s - 1);
		case 5: s.increaseDistance(); return Score::to_short(s);
		case 6: s.decreaseDistance(); return Score::to_short(s);
		default: return raw;
	}
}
/* flags: bit0 isProven, bit1 isWin, bit2 isLoss, bit3 isDraw, bit4 isUnproven, bit5 isInfinite; distance in out_distance */
int ref_score_info(uint16_t raw, int *out_distance, float *out_value)
{
	const Score s = Score::from_short(raw);
	*out_distance = s.getDistance();
	const Value v = s.convertToValue();
	out_value[0] = v.win_rate;
	out_value[1] = v.draw_rate;
	return (s.isProven() ? 1 : 0) | (s.isWin() ? 2 : 0) | (s.isLoss() ? 4 : 0) | (s.isDraw() ? 8 : 0) | (s.isUnproven() ? 16 : 0) | (s.isInfinite() ? 32 : 0);
}
uint16_t ref_score_make(int proven_value, int eval)
{
	return Score::to_short(Score(static_cast<ProvenValue>(proven_value), eval));
}
int ref_sizeof(int what)
{
	switch (what)
	{
		case 0: return sizeof(Edge);
		case 1: return sizeof(Node);
		case 2: return sizeof(Move);
		case 3: return sizeof(Score);
		case 4: return sizeof(Value);
		default: return -1;
	}
}
/* running-mean updates exactly as Edge::updateValue / Node::updateValue (Edge.hpp:111-117, Node.hpp:268-274) */
void ref_edge_update_value(float *win_draw, int *visits, float ew, float ed)
{
	Edge e;
	e.setValue(Value(win_draw[0], win_draw[1]));
	for (int i = 0; i < *visits; i++)
		e.updateValue(e.getValue()); // restores the visit counter without changing the value
	e.setValue(Value(win_draw[0], win_draw[1]));
	e.updateValue(Value(ew, ed));
	win_draw[0] = e.getValue().win_rate;
	win_draw[1] = e.getValue().draw_rate;
	*visits = e.getVisits();
}
void ref_node_update_value(float *win_draw, int *visits, float ew, float ed)
{
	Node n;
	n.setValue(Value(win_draw[0], win_draw[1]));
	for (int i = 0; i < *visits; i++)
		n.updateValue(n.getValue());
	n.setValue(Value(win_draw[0], win_draw[1]));
	n.updateValue(Value(ew, ed));
	win_draw[0] = n.getValue().win_rate;
	win_draw[1] = n.getValue().draw_rate;
	*visits = n.getVisits();
}
uint16_t ref_move_to_short(int sign, int row, int col)
{
	return Move(row, col, static_cast<Sign>(sign)).toShort();
}
/* utils/augmentations.hpp: the two matrix forms (copying and in place) and the inverse map, on an n x n matrix of words */
void ref_apply_symmetry(int n, int s, int in_place, const uint32_t *in, uint32_t *out)
{
	matrix<uint32_t> src(n, n), dst(n, n);
	std::memcpy(src.data(), in, sizeof(uint32_t) * n * n);
	if (in_place)
	{
		apply_symmetry_in_place(src, int_to_symmetry(s));
		std::memcpy(out, src.data(), sizeof(uint32_t) * n * n);
	}
	else
	{
		apply_symmetry(dst, src, int_to_symmetry(s));
		std::memcpy(out, dst.data(), sizeof(uint32_t) * n * n);
	}
}
int ref_inverse_symmetry(int s)
{
	return static_cast<int>(get_inverse_symmetry(int_to_symmetry(s)));
}

/* ---- SharedHashTable (search/alpha_beta/SharedHashTable.hpp): the solver's transposition table ---- */
void* ref_tt_create(int rows, int cols, uint64_t entries)
{
	return new SharedHashTable(rows, cols, entries);
}
void ref_tt_destroy(void *h)
{
	delete static_cast<SharedHashTable*>(h);
}
void ref_tt_increase_generation(void *h)
{
	static_cast<SharedHashTable*>(h)->increaseGeneration();
}
void ref_tt_insert(void *h, uint64_t lo, uint64_t hi, int bound, int depth, uint16_t score_raw, uint16_t move_short)
{
	static_cast<SharedHashTable*>(h)->insert(HashKey128(HashKey64(lo), HashKey64(hi)),
			SharedTableData(static_cast<Bound>(bound), depth, Score::from_short(score_raw), Move(move_short)));
}
uint64_t ref_tt_seek(void *h, uint64_t lo, uint64_t hi)
{
	return static_cast<uint64_t>(static_cast<const SharedHashTable*>(h)->seek(HashKey128(HashKey64(lo), HashKey64(hi))));
}

/* ---- RawPatternCalculator: line bit-boards; out[(cell * 4 + dir) * 2 + {0, 1}] = normal (11 cells) / extended (13 cells) pattern.
 * moves: Move::toShort words, 0 = undo the most recent not-yet-undone move ---- */
void ref_raw_patterns(int n, const uint8_t *board, const uint16_t *moves, int n_moves, uint32_t *out)
{
	matrix<Sign> b(n, n);
	for (int i = 0; i < n * n; i++)
		b[i] = static_cast<Sign>(board[i]);
	RawPatternCalculator calc(n, n);
	calc.set(b);
	std::vector<Move> done;
	for (int i = 0; i < n_moves; i++)
	{
		if (moves[i] == 0)
		{
			calc.undoMove(done.back());
			done.pop_back();
		}
		else
		{
			const Move m(moves[i]);
			calc.addMove(m);
			done.push_back(m);
		}
	}
	for (int r = 0; r < n; r++)
		for (int c = 0; c < n; c++)
			for (int d = 0; d < 4; d++)
			{
				out[((r * n + c) * 4 + d) * 2 + 0] = calc.getRawPatternAt<NormalPattern>(r, c, static_cast<Direction>(d));
				out[((r * n + c) * 4 + d) * 2 + 1] = calc.getRawPatternAt<ExtendedPattern>(r, c, static_cast<Direction>(d));
			}
}
int ref_is_straight_four(int n, const uint8_t *board, int row, int col, int dir)
{
	matrix<Sign> b(n, n);
	for (int i = 0; i < n * n; i++)
		b[i] = static_cast<Sign>(board[i]);
	return RawPatternCalculator::isStraightFourAt(b, Move(row, col, Sign::CROSS), static_cast<Direction>(dir)) ? 1 : 0;
}

/* ---- ThreatHistogram: ops[4 * i ..] = (1 add / 0 remove, threat type, row, col); out: for every threat type its count, then (row, col) pairs ---- */
int ref_threat_histogram(const int *ops, int n_ops, int16_t *out)
{
	ThreatHistogram h;
	for (int i = 0; i < n_ops; i++)
	{
		const ThreatType t = static_cast<ThreatType>(ops[4 * i + 1]);
		const Location l(ops[4 * i + 2], ops[4 * i + 3]);
		if (ops[4 * i])
			h.add(t, l);
		else
			h.remove(t, l);
	}
	int pos = 0;
	for (int t = 0; t < 10; t++)
	{
		const LocationList &list = h.get(static_cast<ThreatType>(t));
		out[pos++] = static_cast<int16_t>(list.size());
		for (size_t i = 0; i < list.size(); i++)
		{
			out[pos++] = list[i].row;
			out[pos++] = list[i].col;
		}
	}
	return pos;
}

/* ---- LowFP (utils/low_precision.hpp): the four formats of src/dataset/SearchDataStorage.cpp:22,161-164.
 * format 0 = score_format <1,3,2,-8>, 1 = visit_format <0,3,5,-8>, 2 = policy/value_format <0,4,4,-16>, 3 = fp16_format <0,5,11,-16> ---- */
uint32_t ref_lowfp_to_lowp(int format, float x)
{
	switch (format)
	{
		case 0: return LowFP<1, 3, 2, -8>::to_lowp(x);
		case 1: return LowFP<0, 3, 5, -8>::to_lowp(x);
		case 2: return LowFP<0, 4, 4, -16>::to_lowp(x);
		default: return LowFP<0, 5, 11, -16>::to_lowp(x);
	}
}
float ref_lowfp_to_fp32(int format, uint32_t code)
{
	switch (format)
	{
		case 0: return LowFP<1, 3, 2, -8>::to_fp32(code);
		case 1: return LowFP<0, 3, 5, -8>::to_fp32(code);
		case 2: return LowFP<0, 4, 4, -16>::to_fp32(code);
		default: return LowFP<0, 5, 11, -16>::to_fp32(code);
	}
}
float ref_lowfp_max(int format)
{
	switch (format)
	{
		case 0: return LowFP<1, 3, 2, -8>::max();
		case 1: return LowFP<0, 3, 5, -8>::max();
		case 2: return LowFP<0, 4, 4, -16>::max();
		default: return LowFP<0, 5, 11, -16>::max();
	}
}


/* ---- FullZobristHashing / FastZobristHashing (search/ZobristHashing.cpp:15-68): ONE hashing object per call (its keys are random), the
 * hashes of `n` boards; tests derive the keys from single-stone boards and check that the oracle's formulas reproduce every other hash ---- */
static matrix<Sign> to_matrix(int rows, int cols, const uint8_t *cells)
{
	matrix<Sign> m(rows, cols);
	for (int i = 0; i < rows * cols; i++)
		m[i] = static_cast<Sign>(cells[i]);
	return m;
}
void ref_full_zobrist(int rows, int cols, const uint8_t *boards, const int *signs, int n, uint64_t *out)
{
	const FullZobristHashing hashing(rows, cols);
	for (int b = 0; b < n; b++)
		out[b] = static_cast<uint64_t>(hashing.getHash(to_matrix(rows, cols, boards + static_cast<size_t>(b) * rows * cols), static_cast<Sign>(signs[b])));
}
/* out[2 b], out[2 b + 1] = low / high word of getHash(board b); then, starting from the hash of board 0, the hash after every updateHash(move)
 * of `moves` (placing or removing: the same XOR), two words each */
void ref_fast_zobrist(int rows, int cols, const uint8_t *boards, int n, const uint16_t *moves, int n_moves, uint64_t *out)
{
	const FastZobristHashing hashing(rows, cols);
	for (int b = 0; b < n; b++)
	{
		const HashKey128 h = hashing.getHash(to_matrix(rows, cols, boards + static_cast<size_t>(b) * rows * cols));
		out[2 * b] = static_cast<uint64_t>(h.getLow());
		out[2 * b + 1] = static_cast<uint64_t>(h.getHigh());
	}
	HashKey128 h = hashing.getHash(to_matrix(rows, cols, boards));
	for (int i = 0; i < n_moves; i++)
	{
		hashing.updateHash(h, Move(moves[i]));
		out[2 * (n + i)] = static_cast<uint64_t>(h.getLow());
		out[2 * (n + i) + 1] = static_cast<uint64_t>(h.getHigh());
	}
}

/* ---- ActionStack / ActionList (search/alpha_beta/ActionList.hpp:247-470) driven by a script of operations on a stack of nested lists:
 *   1 move score num : top.add(Move(move), Score::from_short(score), num)        2 index : open a child list of top at move `index`
 *   3                : close the top list (its destructor releases its actions)   4 move offset : top.moveCloserToFront(Move(move), offset)
 * After every operation out gets (stack offset, stack max_offset, size of the top list, its distance from the root); at the end the
 * (move, score) of every action of every open list, root first.  Returns the number of ints written. ---- */
/* ---- utils/BitMask.hpp: BitMask1D<uint16_t> (a line's move mask) and BitMask2D<uint32_t, 32> (a board's move mask) under a script.
 * ops: 1 idx v : m.at(idx) = v      2 length : m.flip(length)     3 s : m = m << s      4 s : m = m >> s      5 x : m &= x      6 x : m |= x
 *      7 r c v : g.at(r, c) = v     8 r c v : h.at(r, c) = v      9 : g &= h            10 : g |= h           11 b : g.fill(b)
 * After every operation out gets m.raw(), m == BitMask1D(x of the last op 5 / 6), and the 32 row words of g. ---- */
int ref_bitmask_script(const int *ops, int n_ops, int rows, int cols, uint32_t *out, int capacity)
{
	BitMask1D<uint16_t> m, last;
	BitMask2D<uint32_t, 32> g(rows, cols), h(rows, cols);
	int at = 0, pos = 0;
	for (int k = 0; k < n_ops; k++)
	{
		const int op = ops[at++];
		switch (op)
		{
			case 1: m.at(ops[at]) = (ops[at + 1] != 0); at += 2; break;
			case 2: m.flip(ops[at++]); break;
			case 3: m = m << ops[at++]; break;
			case 4: m = m >> ops[at++]; break;
			case 5: last = BitMask1D<uint16_t>(static_cast<uint16_t>(ops[at++])); m &= last; break;
			case 6: last = BitMask1D<uint16_t>(static_cast<uint16_t>(ops[at++])); m |= last; break;
			case 7: g.at(ops[at], ops[at + 1]) = (ops[at + 2] != 0); at += 3; break;
			case 8: h.at(ops[at], ops[at + 1]) = (ops[at + 2] != 0); at += 3; break;
			case 9: g &= h; break;
			case 10: g |= h; break;
			default: g.fill(ops[at++] != 0); break;
		}
		if (pos + 34 > capacity)
			return -1;
		out[pos++] = m.raw();
		out[pos++] = (m == last) ? 1u : 0u;
		for (int r = 0; r < 32; r++)
			out[pos++] = g.data()[r];
	}
	return pos;
}
int ref_action_list_script(const int *ops, int n_ops, int *out, int capacity)
{
	ActionStack stack(4096);
	std::vector<std::unique_ptr<ActionList>> lists;
	lists.push_back(std::make_unique<ActionList>(stack));
	int pos = 0, at = 0;
	for (int k = 0; k < n_ops; k++)
	{
		const int op = ops[at++];
		ActionList &top = *lists.back();
		if (op == 1)
		{
			top.add(Move(static_cast<uint16_t>(ops[at])), Score::from_short(static_cast<uint16_t>(ops[at + 1])), ops[at + 2]);
			at += 3;
		}
		else if (op == 2)
			lists.push_back(std::make_unique<ActionList>(stack, top, ops[at++]));
		else if (op == 3)
			lists.pop_back();
		else
		{
			top.moveCloserToFront(Move(static_cast<uint16_t>(ops[at])), ops[at + 1]);
			at += 2;
		}
		if (pos + 4 > capacity)
			return -1;
		out[pos++] = static_cast<int>(stack.offset());
		out[pos++] = static_cast<int>(stack.max_offset());
		out[pos++] = lists.back()->size();
		out[pos++] = lists.back()->distanceFromRoot();
	}
	for (const auto &l : lists)
		for (int i = 0; i < l->size(); i++)
		{
			if (pos + 2 > capacity)
				return -1;
			out[pos++] = (*l)[i].move.toShort();
			out[pos++] = Score::to_short((*l)[i].score);
		}
	while (lists.size() > 1)
		lists.pop_back(); // children before parents: each releases its own actions
	return pos;
}
} /* extern "C" */
