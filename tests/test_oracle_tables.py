"""Pins the oracle's tables / score algebra / record arithmetic against the REAL reference code compiled from its own
sources (oracle/_ref/libagref.so, built by oracle/Makefile from /root/reference without any stand-in header), and —
where the prebuilt reference library is unavailable — against checksums committed in tests/golden/tables.json that were
produced from it (tests/golden/make_tables_golden.py)."""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RULES = ["FREESTYLE", "STANDARD", "RENJU", "CARO5", "CARO6"]


@pytest.fixture(scope="module")
def oracle():
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libagoracle.so"])
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libagoracle.so"))
    lib.ago_defensive_moves.restype = ctypes.c_uint16
    lib.ago_open_three_promotion_moves.restype = ctypes.c_uint16
    lib.ago_score_op.restype = ctypes.c_uint16
    lib.ago_score_make.restype = ctypes.c_uint16
    lib.ago_move_to_short.restype = ctypes.c_uint16
    return lib


@pytest.fixture(scope="module")
def ref():
    path = os.path.join(ROOT, "oracle", "_ref", "libagref.so")
    if not os.path.exists(path):
        if os.path.isdir("/root/reference/src"):
            import subprocess
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/libagref.so"])
        else:
            pytest.skip("reference library not built")
    lib = ctypes.CDLL(path)
    lib.ref_defensive_moves.restype = ctypes.c_uint16
    lib.ref_open_three_promotion_moves.restype = ctypes.c_uint16
    lib.ref_score_op.restype = ctypes.c_uint16
    lib.ref_score_make.restype = ctypes.c_uint16
    lib.ref_move_to_short.restype = ctypes.c_uint16
    return lib


def oracle_tables(oracle, rules):
    types = np.zeros(1 << 20, np.uint8)
    ho3 = np.zeros(1 << 20, np.uint8)
    thr = np.zeros(4096 * 2, np.uint8)
    oracle.ago_tables(rules, types.ctypes.data_as(ctypes.c_void_p), ho3.ctypes.data_as(ctypes.c_void_p), thr.ctypes.data_as(ctypes.c_void_p))
    return types, ho3, thr


def valid_extended_patterns(rng, count):
    """random 13-cell lines whose off-board cells (3) are contiguous from the outside and whose centre is empty"""
    out = []
    for _ in range(count):
        cells = rng.integers(0, 3, size=13)
        left = int(rng.integers(0, 7)) if rng.random() < 0.3 else 0
        right = int(rng.integers(0, 7)) if rng.random() < 0.3 else 0
        cells[:left] = 3
        if right:
            cells[13 - right:] = 3
        cells[6] = 0
        out.append(int(sum(int(c) << (2 * i) for i, c in enumerate(cells))))
    return out


@pytest.mark.parametrize("rules", range(5))
def test_pattern_and_threat_tables_equal_reference(oracle, ref, rules):
    types, ho3, thr = oracle_tables(oracle, rules)
    rt = np.zeros(1 << 20, np.uint8)
    rh = np.zeros(1 << 20, np.uint8)
    rthr = np.zeros(4096 * 2, np.uint8)
    ref.ref_pattern_table(rules, rt.ctypes.data_as(ctypes.c_void_p), rh.ctypes.data_as(ctypes.c_void_p))
    ref.ref_threat_table(rules, rthr.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(types, rt)
    assert np.array_equal(ho3, rh)
    assert np.array_equal(thr, rthr)


@pytest.mark.parametrize("rules", range(5))
def test_table_checksums_match_golden(oracle, rules):
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "tables.json")))
    types, ho3, thr = oracle_tables(oracle, rules)
    g = golden[RULES[rules]]
    assert hashlib.sha256(types.tobytes()).hexdigest() == g["pattern_types_sha256"]
    assert hashlib.sha256(ho3.tobytes()).hexdigest() == g["half_open_3_sha256"]
    assert hashlib.sha256(thr.tobytes()).hexdigest() == g["threats_sha256"]
    rng = np.random.default_rng(12345)
    pats = valid_extended_patterns(rng, 4000)
    acc = hashlib.sha256()
    for p in pats:
        for defender in (1, 2):
            for pt in (2, 3, 4, 5, 6):
                acc.update(int(oracle.ago_defensive_moves(rules, p, defender, pt)).to_bytes(2, "little"))
    assert acc.hexdigest() == g["defensive_moves_sha256"]


@pytest.mark.parametrize("rules", range(5))
def test_defensive_moves_equal_reference(oracle, ref, rules):
    rng = np.random.default_rng(777 + rules)
    for p in valid_extended_patterns(rng, 6000):
        for defender in (1, 2):
            for pt in (2, 3, 4, 5, 6):
                assert oracle.ago_defensive_moves(rules, p, defender, pt) == ref.ref_defensive_moves(rules, p, defender, pt), (p, defender, pt)


def test_open_three_promotion_moves_equal_reference(oracle, ref):
    # every 11-cell line that contains one of the four open-three shapes for cross, centre empty
    shapes = [[0, 1, 1, 1, 0, 0], [0, 1, 1, 0, 1, 0], [0, 1, 0, 1, 1, 0], [0, 0, 1, 1, 1, 0]]
    checked = 0
    for shape in shapes:
        for start in range(0, 6):
            cells = [0] * 11
            cells[start:start + 6] = shape
            # the centre must be one of the cross stones of the shape, removed (patterns have an empty centre)
            if not (start <= 5 < start + 6) or cells[5] != 1:
                continue
            cells[5] = 0
            p = sum(c << (2 * i) for i, c in enumerate(cells))
            assert oracle.ago_open_three_promotion_moves(p) == ref.ref_open_three_promotion_moves(p)
            checked += 1
    assert checked >= 9


def test_score_algebra_equals_reference_on_every_raw_value(oracle, ref):
    d1, d2 = ctypes.c_int(), ctypes.c_int()
    v1, v2 = (ctypes.c_float * 2)(), (ctypes.c_float * 2)()
    for raw in range(0, 65536):
        pv, ev = (raw >> 13) & 3, (raw & 8191) - 4000
        if not (raw in (0, 0xFFFF) or -3000 <= ev <= 3000):
            continue  # constructors assert |eval| <= 4000; +-1 steps must stay inside
        for op in (0, 1, 2):
            assert oracle.ago_score_op(raw, op) == ref.ref_score_op(raw, op), (raw, op)
        f1 = oracle.ago_score_info(raw, ctypes.byref(d1), v1)
        f2 = ref.ref_score_info(raw, ctypes.byref(d2), v2)
        assert f1 == f2 and d1.value == d2.value and tuple(v1) == tuple(v2), raw
    for pv in range(4):
        for ev in (-1000, -1, 0, 1, 7, 1000):
            assert oracle.ago_score_make(pv, ev) == ref.ref_score_make(pv, ev)


def test_running_means_equal_reference(oracle, ref):
    rng = np.random.default_rng(5)
    for _ in range(500):
        w, d = float(np.float32(rng.random() * 0.8)), float(np.float32(rng.random() * 0.2))
        ew, ed = float(np.float32(rng.random() * 0.7)), float(np.float32(rng.random() * 0.3))
        visits = int(rng.integers(0, 2000))
        for name in ("edge", "node"):
            a = (ctypes.c_float * 2)(w, d)
            b = (ctypes.c_float * 2)(w, d)
            va, vb = ctypes.c_int(visits), ctypes.c_int(visits)
            getattr(oracle, "ago_%s_update_value" % name)(a, ctypes.byref(va), ctypes.c_float(ew), ctypes.c_float(ed))
            getattr(ref, "ref_%s_update_value" % name)(b, ctypes.byref(vb), ctypes.c_float(ew), ctypes.c_float(ed))
            assert tuple(a) == tuple(b) and va.value == vb.value


def test_record_layouts(ref, oracle):
    assert [ref.ref_sizeof(i) for i in range(5)] == [24, 40, 4, 2, 8]
    for sign, r, c in [(1, 0, 0), (2, 14, 3), (1, 19, 19)]:
        assert oracle.ago_move_to_short(sign, r, c) == ref.ref_move_to_short(sign, r, c)


def test_board_symmetries_equal_reference(oracle, ref):
    """The 8 dihedral maps used to augment network inputs (utils/augmentations.hpp): copying form, in-place form, inverse."""
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    for n in (5, 15, 20):
        src = np.arange(n * n, dtype=np.uint32) * 7 + 3
        for s in range(8):
            a, b, c = np.zeros_like(src), np.zeros_like(src), np.zeros_like(src)
            oracle.ago_apply_symmetry(n, s, 0, p(src), p(a))
            ref.ref_apply_symmetry(n, s, 0, p(src), p(b))
            ref.ref_apply_symmetry(n, s, 1, p(src), p(c))
            assert np.array_equal(a, b) and np.array_equal(a, c), (n, s)
            assert oracle.ago_inverse_symmetry(s) == ref.ref_inverse_symmetry(s)
            back = np.zeros_like(src)
            oracle.ago_apply_symmetry(n, oracle.ago_inverse_symmetry(s), 0, p(a), p(back))
            assert np.array_equal(back, src)


def test_feature_direction_shuffle(oracle):
    """NNInputFeatures::augment (NNInputFeatures.cpp:33-50,114-154): a reflection swaps the two diagonal-direction bits, a
    transposition swaps horizontal/vertical, a quarter turn swaps both pairs; every other bit stays."""
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    n = 5
    base = 0xF00F00FF
    for s, perm in [(0, (0, 1, 2, 3)), (3, (0, 1, 2, 3)), (1, (0, 1, 3, 2)), (2, (0, 1, 3, 2)), (4, (1, 0, 2, 3)), (5, (1, 0, 2, 3)),
                    (6, (1, 0, 3, 2)), (7, (1, 0, 3, 2))]:
        for group in (8, 12, 20, 24):
            for d in range(4):
                src = np.full(n * n, base | (1 << (group + d)), dtype=np.uint32)
                out = np.zeros_like(src)
                oracle.ago_apply_symmetry(n, s, 1, p(src), p(out))
                want = base | (1 << (group + perm.index(d)))
                assert (out == want).all(), (s, group, d)


def test_transposition_table_equals_reference(oracle, ref):
    """SharedHashTable (search/alpha_beta/SharedHashTable.hpp) compiled from the reference vs the oracle's table: bucket choice,
    key matching, replacement by depth - age, in-place update for proven / exact entries, generations."""
    oracle.ago_solver_create.restype = ctypes.c_void_p
    oracle.ago_solver_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int]
    oracle.ago_solver_tt_insert.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_uint16, ctypes.c_uint16]
    oracle.ago_solver_tt_seek.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
    oracle.ago_solver_tt_seek.restype = ctypes.c_uint64
    oracle.ago_solver_new_generation.argtypes = [ctypes.c_void_p]
    oracle.ago_solver_destroy.argtypes = [ctypes.c_void_p]
    ref.ref_tt_create.restype = ctypes.c_void_p
    ref.ref_tt_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint64]
    ref.ref_tt_insert.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_uint16, ctypes.c_uint16]
    ref.ref_tt_seek.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
    ref.ref_tt_seek.restype = ctypes.c_uint64
    ref.ref_tt_increase_generation.argtypes = [ctypes.c_void_p]
    ref.ref_tt_destroy.argtypes = [ctypes.c_void_p]
    rng = np.random.default_rng(77)
    entries = 64   # 16 buckets: plenty of collisions and replacements
    a = oracle.ago_solver_create(0, 15, 15, entries, 1, 100)
    b = ref.ref_tt_create(15, 15, entries)
    keys = [(int(rng.integers(0, 1 << 62)) * 4 + int(rng.integers(0, 4)), int(rng.integers(0, 1 << 63))) for _ in range(200)]
    # a few keys that share the bucket AND the 16 high bits of the low word but differ in the high word
    keys += [(keys[0][0], keys[0][1] ^ 1), (keys[1][0] ^ (1 << 20), keys[1][1])]
    proven_scores = [oracle.ago_score_make(pv, ev) for pv in (0, 1, 3) for ev in (0, 3, 17)]
    checked = hits = 0
    for step in range(6000):
        lo, hi = keys[int(rng.integers(0, len(keys)))]
        action = rng.random()
        if action < 0.45:
            score = int(proven_scores[int(rng.integers(0, len(proven_scores)))]) if rng.random() < 0.3 else int(oracle.ago_score_make(2, int(rng.integers(-500, 500))))
            bound, depth, move = int(rng.integers(1, 4)), int(rng.integers(0, 40)), int(rng.integers(1, 1 << 16))
            oracle.ago_solver_tt_insert(a, lo, hi, bound, depth, score, move)
            ref.ref_tt_insert(b, lo, hi, bound, depth, score, move)
        elif action < 0.47:
            oracle.ago_solver_new_generation(a)
            ref.ref_tt_increase_generation(b)
        else:
            x, y = oracle.ago_solver_tt_seek(a, lo, hi), ref.ref_tt_seek(b, lo, hi)
            assert x == y, (step, hex(x), hex(y))
            checked += 1
            hits += int((x & 3) != 0)
    assert checked > 2000 and hits > 500
    oracle.ago_solver_destroy(a)
    ref.ref_tt_destroy(b)


def test_line_patterns_equal_reference(oracle, ref):
    """RawPatternCalculator (line bit-boards, incremental add / undo, 11- and 13-cell windows) and isStraightFourAt"""
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    rng = np.random.default_rng(5)
    for n in (15, 20, 9):
        for trial in range(20):
            board = np.zeros(n * n, np.uint8)
            stones = rng.permutation(n * n)[:int(rng.integers(0, n * n // 2))]
            board[stones] = rng.integers(1, 3, len(stones))
            cur, moves, done = board.copy(), [], []
            for _ in range(30):
                if done and rng.random() < 0.3:
                    moves.append(0)
                    cur[done.pop()] = 0
                else:
                    cell = int(rng.choice(np.flatnonzero(cur == 0)))
                    s = int(rng.integers(1, 3))
                    moves.append(s | ((cell // n) << 2) | ((cell % n) << 9))
                    done.append(cell)
                    cur[cell] = s
            mv = np.array(moves, np.uint16)
            a, b = np.zeros(n * n * 8, np.uint32), np.zeros(n * n * 8, np.uint32)
            oracle.ago_raw_patterns(n, n, p(board), p(mv), len(mv), p(a))
            ref.ref_raw_patterns(n, p(board), p(mv), len(mv), p(b))
            assert np.array_equal(a, b), (n, trial)
            for _ in range(40):
                cell = int(rng.choice(np.flatnonzero(cur == 0)))
                d = int(rng.integers(0, 4))
                assert oracle.ago_is_straight_four(n, n, p(cur), cell // n, cell % n, d) == ref.ref_is_straight_four(n, p(cur), cell // n, cell % n, d)
    # a position where the answer is yes: X X . X on a row, playing the gap
    row = np.zeros(15 * 15, np.uint8)
    row[7 * 15 + 3] = row[7 * 15 + 4] = row[7 * 15 + 6] = 1
    assert ref.ref_is_straight_four(15, p(row), 7, 5, 0) == 1 and oracle.ago_is_straight_four(15, 15, p(row), 7, 5, 0) == 1


def test_threat_lists_equal_reference(oracle, ref):
    """ThreatHistogram: push-back add, first-match swap-with-last remove (the AVX2 search of the reference included): the ORDER of
    every list after long random add / remove sequences"""
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    rng = np.random.default_rng(8)
    for trial in range(30):
        present = {t: [] for t in range(10)}
        ops = []
        for _ in range(int(rng.integers(10, 600))):
            t = int(rng.integers(0, 10))
            if present[t] and rng.random() < 0.45:
                r, c = present[t].pop(int(rng.integers(0, len(present[t]))))
                ops += [0, t, r, c]
            else:
                r, c = int(rng.integers(0, 20)), int(rng.integers(0, 20))
                if (r, c) in present[t]:
                    continue
                if t != 0:
                    present[t].append((r, c))
                ops += [1, t, r, c]
        arr = np.array(ops, np.int32)
        a, b = np.zeros(4096, np.int16), np.zeros(4096, np.int16)
        na = oracle.ago_threat_histogram(p(arr), len(ops) // 4, p(a))
        nb = ref.ref_threat_histogram(p(arr), len(ops) // 4, p(b))
        assert na == nb and np.array_equal(a[:na], b[:nb]), trial


# ---- round 3: hashing formulas and the solver's action lists against the compiled reference ----

def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("n", [9, 15, 20])
def test_full_zobrist_formula_matches_compiled_reference(oracle, ref, n):
    """FullZobristHashing::getHash (ZobristHashing.cpp:21-33).  The reference's keys are random per object, so the hashes of the empty board
    and of every single-stone board give the key table (up to the XOR of the empty-cell keys, which is folded into the sign keys); the
    oracle's formula (Tree::hash_of) on that table must then reproduce the reference's hash of arbitrary boards, for all three signs."""
    rng = np.random.default_rng(n)
    hw = n * n
    boards, signs = [np.zeros(hw, np.uint8)] * 3, [0, 1, 2]
    for cell in range(hw):
        for v in (1, 2):
            b = np.zeros(hw, np.uint8)
            b[cell] = v
            boards.append(b)
            signs.append(1)
    probes = []
    for _ in range(60):
        b = rng.integers(0, 3, hw).astype(np.uint8) * (rng.random(hw) < rng.random()).astype(np.uint8)
        probes.append(b)
        boards.append(b)
        signs.append(int(rng.integers(0, 3)))
    flat = np.ascontiguousarray(np.stack(boards))
    sg = np.array(signs, np.int32)
    out = np.zeros(len(boards), np.uint64)
    ref.ref_full_zobrist(n, n, _ptr(flat), _ptr(sg), len(boards), _ptr(out))
    keys = np.zeros(3 + 3 * hw, np.uint64)
    keys[0:3] = out[0:3]                                   # sign key ^ XOR of all empty-cell keys
    for cell in range(hw):
        for v in (1, 2):
            keys[3 + 3 * cell + v] = out[3 + 2 * cell + (v - 1)] ^ out[1]     # cell key relative to the empty cell's
    oracle.ago_full_zobrist_with_keys.restype = ctypes.c_uint64
    for k, b in enumerate(probes):
        idx = 3 + 2 * hw + k
        got = oracle.ago_full_zobrist_with_keys(_ptr(keys), hw, _ptr(np.ascontiguousarray(b)), int(sg[idx]))
        assert got == int(out[idx]), k


@pytest.mark.parametrize("n", [15, 20])
def test_fast_zobrist_formula_matches_compiled_reference(oracle, ref, n):
    """FastZobristHashing::getHash / updateHash (ZobristHashing.cpp:49-68, ZobristHashing.hpp:125-129): the solver's 128-bit keys"""
    rng = np.random.default_rng(100 + n)
    hw = n * n
    start = (rng.integers(0, 3, hw) * (rng.random(hw) < 0.2)).astype(np.uint8)
    boards = [start]
    for cell in range(hw):
        for v in (1, 2):
            b = np.zeros(hw, np.uint8)
            b[cell] = v
            boards.append(b)
    probes = [(rng.integers(0, 3, hw) * (rng.random(hw) < rng.random())).astype(np.uint8) for _ in range(40)]
    boards += probes
    # a sequence of placements and removals on the first board
    moves, cur = [], start.copy()
    for _ in range(80):
        cell = int(rng.integers(0, hw))
        v = int(cur[cell]) if cur[cell] else int(rng.integers(1, 3))
        cur[cell] = 0 if cur[cell] else v
        moves.append(v | ((cell // n) << 2) | ((cell % n) << 9))
    mv = np.array(moves, np.uint16)
    flat = np.ascontiguousarray(np.stack(boards))
    out = np.zeros(2 * (len(boards) + len(moves)), np.uint64)
    ref.ref_fast_zobrist(n, n, _ptr(flat), len(boards), _ptr(mv), len(moves), _ptr(out))
    keys = np.zeros((2 * hw, 2), np.uint64)
    for cell in range(hw):
        for v in (1, 2):
            i = 1 + 2 * cell + (v - 1)
            keys[2 * cell + (v - 1)] = out[2 * i:2 * i + 2]          # the empty board hashes to zero: a single stone's hash IS its key
    for k, b in enumerate(probes):
        got = np.zeros(2, np.uint64)
        oracle.ago_fast_zobrist_with_keys(_ptr(keys), n, n, _ptr(np.ascontiguousarray(b)), None, 0, _ptr(got))
        i = 1 + 2 * hw + k
        assert np.array_equal(got, out[2 * i:2 * i + 2]), k
    got = np.zeros(2 * (1 + len(moves)), np.uint64)
    oracle.ago_fast_zobrist_with_keys(_ptr(keys), n, n, _ptr(start), _ptr(mv), len(moves), _ptr(got))
    assert np.array_equal(got[:2], out[:2])
    assert np.array_equal(got[2:], out[2 * len(boards):])             # every add / undo step


def test_action_list_mechanics_match_compiled_reference(oracle, ref):
    """ActionStack / ActionList (search/alpha_beta/ActionList.hpp:247-470): add (incl. num = 0: the slot is written, the size is not),
    nested child lists on the shared stack, release on close, moveCloserToFront(move, offset) — stack offsets, high-water mark, list sizes
    and the final ORDER of every list, on random scripts."""
    rng = np.random.default_rng(7)
    for trial in range(200):
        ops, n_ops, depth, sizes, seen = [], 0, 0, [0], []
        for _ in range(int(rng.integers(5, 60))):
            r = rng.random()
            if r < 0.55 or sizes[-1] == 0:
                num = 0 if rng.random() < 0.1 else 1
                move = int(rng.integers(1, 3)) | (int(rng.integers(0, 15)) << 2) | (int(rng.integers(0, 15)) << 9)
                seen.append(move)
                ops += [1, move, int(rng.integers(0, 65536)), num]
                sizes[-1] += num
            elif r < 0.7 and depth < 6:
                ops += [2, int(rng.integers(0, sizes[-1]))]
                sizes.append(0)
                depth += 1
            elif r < 0.8 and depth > 0:
                ops += [3]
                sizes.pop()
                depth -= 1
            else:   # mostly a move that is somewhere on the stack (the table move of the solver), sometimes one that is not
                move = seen[int(rng.integers(0, len(seen)))] if rng.random() < 0.8 else int(rng.integers(1, 3)) | (int(rng.integers(0, 15)) << 2)
                ops += [4, move, int(rng.integers(0, max(1, sizes[-1])))]
            n_ops += 1
        arr = np.array(ops, np.int32)
        a, b = np.zeros(8192, np.int32), np.zeros(8192, np.int32)
        na = ref.ref_action_list_script(_ptr(arr), n_ops, _ptr(a), len(a))
        nb = oracle.ago_action_list_script(_ptr(arr), n_ops, _ptr(b), len(b))
        assert na == nb and na > 0 and np.array_equal(a[:na], b[:nb]), trial


def test_bitmask_matches_reference(ref, oracle):
    """utils/BitMask.hpp (the compiled reference header): BitMask1D<uint16_t> — at / reference assignment, flip(length) over reverse_bits,
    shifts, &=, |=, == — and BitMask2D<uint32_t, 32> — at(row, col), fill, &=, |= — against the oracle's plain-word restatement (the move
    masks of a line and of a board in its move generator and defensive-move tables), on random scripts."""
    rng = np.random.default_rng(11)
    for trial in range(100):
        rows, cols = int(rng.integers(5, 21)), int(rng.integers(5, 21))
        ops, n_ops = [], 0
        for _ in range(int(rng.integers(10, 120))):
            op = int(rng.integers(1, 12))
            if op == 1:
                ops += [1, int(rng.integers(0, 16)), int(rng.integers(0, 2))]
            elif op == 2:
                ops += [2, int(rng.integers(1, 17))]
            elif op in (3, 4):
                ops += [op, int(rng.integers(0, 16))]
            elif op in (5, 6):
                ops += [op, int(rng.integers(0, 65536))]
            elif op in (7, 8):
                ops += [op, int(rng.integers(0, rows)), int(rng.integers(0, cols)), int(rng.integers(0, 2))]
            elif op in (9, 10):
                ops += [op]
            else:
                ops += [11, int(rng.integers(0, 2))]
            n_ops += 1
        arr = np.array(ops, np.int32)
        a, b = np.zeros(34 * n_ops, np.uint32), np.zeros(34 * n_ops, np.uint32)
        na = ref.ref_bitmask_script(_ptr(arr), n_ops, rows, cols, _ptr(a), len(a))
        nb = oracle.ago_bitmask_script(_ptr(arr), n_ops, rows, cols, _ptr(b), len(b))
        assert na == nb == 34 * n_ops and np.array_equal(a, b), trial
    # reverse_bits itself, every 16-bit word: flip(16) of a fresh mask
    words = np.arange(65536, dtype=np.int64)
    expect = np.zeros(65536, np.int64)
    for i in range(16):
        expect |= ((words >> i) & 1) << (15 - i)
    for x in (0, 1, 0x8000, 0x1234, 0xFFFF, 0xA5A5):
        arr = np.array([6, x, 2, 16], np.int32)
        a = np.zeros(68, np.uint32)
        assert ref.ref_bitmask_script(_ptr(arr), 2, 15, 15, _ptr(a), 68) == 68 and a[34] == expect[x]
