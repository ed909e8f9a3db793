"""The CPU oracle against the known-answer cases of the reference's own unit tests (restated as data in
tests/golden/ref_*_cases.json by tests/golden/extract_reference_fixtures.py): game rules incl. renju fouls, NN input
bit layout, staged move generator (37 test functions)."""
import ctypes
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return json.load(open(os.path.join(GOLDEN, name)))


@pytest.fixture(scope="module")
def lib():
    return ol.load()


@pytest.mark.parametrize("case", load("ref_rules_cases.json"), ids=lambda c: c["name"])
def test_rules(lib, case):
    b = ol.board_array(case["board"])
    rows, cols = b.shape
    for chk in case["checks"]:
        m = chk["move"]
        if chk["kind"] == "outcome":
            got = lib.ago_outcome(ol.RULES[chk["rules"]], rows, cols, ol.ptr(b), m["sign"], m["row"], m["col"], -1)
            assert got == ol.OUTCOMES[chk["expected"]], chk
        else:
            got = lib.ago_is_forbidden(rows, cols, ol.ptr(b), m["sign"], m["row"], m["col"])
            assert bool(got) == chk["expected"], chk
            # the incremental calculator must agree with the static rule (test/game/test_renju.cpp:45-51): bit 6 of the features
            if b[m["row"], m["col"]] == 0:
                f = ol.encode_features(lib, ol.RULES["RENJU"], case["board"], 1)
                assert bool((int(f[m["row"], m["col"]]) >> 6) & 1) == chk["expected"], chk


@pytest.mark.parametrize("case", load("ref_features_cases.json"), ids=lambda c: c["name"])
def test_features(lib, case):
    sign = ol.SIGNS[case["sign_to_move"]]
    f = ol.encode_features(lib, ol.RULES[case["rules"]], case["board"], sign)
    b = np.array(case["board"])
    own, opp = sign, 3 - sign
    # generic per-cell layout (test/networks/test_NNInputFeatures.cpp:112-141)
    assert np.array_equal((f >> 0) & 1, (b == 0).astype(np.uint32))
    assert np.array_equal((f >> 1) & 1, (b == own).astype(np.uint32))
    assert np.array_equal((f >> 2) & 1, (b == opp).astype(np.uint32))
    assert ((f >> 3) & 1).all()
    assert (((f >> 4) & 1) == (1 if sign == 1 else 0)).all()
    assert (((f >> 5) & 1) == (1 if sign == 2 else 0)).all()
    assert not ((f >> 6) & 1).any() and not ((f >> 7) & 1).any()
    for chk in case["bits"]:
        assert bool((int(f[chk["row"], chk["col"]]) >> chk["bit"]) & 1) == chk["expected"], chk


MOVEGEN = load("ref_movegen_cases.json")


def run_movegen(lib, case):
    return ol.movegen(lib, ol.RULES[case["rules"]], case["board"], ol.SIGNS[case["sign_to_move"]], ol.MODES[case["mode"]])


@pytest.mark.parametrize("case", MOVEGEN, ids=lambda c: c["name"])
def test_move_generator(lib, case):
    got = run_movegen(lib, case)
    locs = [(m >> 2 & 127, m >> 9 & 127) for m in got["moves"]]
    assert len(set(locs)) == len(locs)
    if case["size"] is not None:
        assert len(got["moves"]) == case["size"], (locs,)
    if case["must_defend"] is not None:
        assert got["must_defend"] == case["must_defend"]
    if case["has_initiative"] is not None:
        assert got["has_initiative"] == case["has_initiative"]
    for m in case["contains"]:
        assert ol.move_short(m) in got["moves"], (m, locs)
    # ORDER: no case of test/search/alpha_beta/test_move_generator.cpp reads an action by index (its 40 test functions use size(), contains(),
    # getScoreOf(), equals() — membership and scores only), so the reference fixes the ORDER of a list only where the list has ONE move: there
    # the whole list is pinned (8 of the 57 action lists; DESIGN.md §4 names them)
    if case["size"] == 1 and len(case["contains"]) == 1:
        assert got["moves"] == [ol.move_short(case["contains"][0])]
    for m in case["not_contains"]:
        assert ol.move_short(m) not in got["moves"], (m, locs)
    for s in case["scores"]:
        idx = got["moves"].index(ol.move_short(s["move"]))
        expected = lib.ago_score_make({"loss_in": 0, "draw_in": 1, "win_in": 3}[s["kind"]], s["n"] if s["kind"] != "win_in" else -s["n"])
        assert got["scores"][idx] == expected
    if "equals" in case:
        other = run_movegen(lib, [c for c in MOVEGEN if c["name"] == case["equals"]][0])
        assert sorted(other["moves"]) == sorted(got["moves"])
        assert (other["must_defend"], other["has_initiative"], other["fully_expanded"]) == (got["must_defend"], got["has_initiative"], got["fully_expanded"])


@pytest.mark.parametrize("rules", ["FREESTYLE", "STANDARD", "RENJU", "CARO5"])
def test_augment_equals_encode_of_the_symmetric_board(lib, rules):
    """The property test/networks/test_NNInputFeatures.cpp:234-279 checks for the reference: augmenting the features of a board
    with symmetry s (cell permutation + direction-bit shuffle) gives the features of the symmetric board.  This is what pins the
    direction shuffle of oracle/agoracle.hpp (its TU needs MinML, so it cannot be compiled) to the pinned feature encoder."""
    rng = np.random.default_rng(11)
    lib.ago_apply_symmetry.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    seen_directional = 0
    for n in (12, 15):
        for trial in range(12):
            b = np.zeros((n, n), np.uint32)
            # clustered stones so that threes and fours (the per-direction bits) actually occur
            r, c = n // 2, n // 2
            for k in range(int(rng.integers(6, 40))):
                for _ in range(50):
                    rr, cc = r + int(rng.integers(-2, 3)), c + int(rng.integers(-2, 3))
                    if 0 <= rr < n and 0 <= cc < n and b[rr, cc] == 0:
                        b[rr, cc] = 1 + (k & 1)
                        r, c = rr, cc
                        break
            sign = 1 if int((b != 0).sum()) % 2 == 0 else 2
            f = np.ascontiguousarray(ol.encode_features(lib, ol.RULES[rules], b.tolist(), sign).reshape(-1), dtype=np.uint32)
            for s in range(8):
                bs = np.zeros(n * n, np.uint32)
                lib.ago_apply_symmetry(n, s, 0, ol.ptr(np.ascontiguousarray(b.reshape(-1))), ol.ptr(bs))
                want = ol.encode_features(lib, ol.RULES[rules], bs.reshape(n, n).tolist(), sign).reshape(-1)
                got = np.zeros(n * n, np.uint32)
                lib.ago_apply_symmetry(n, s, 1, ol.ptr(f), ol.ptr(got))
                assert np.array_equal(got, want), (rules, n, trial, s)
                seen_directional += int(np.count_nonzero(want & 0x0FF0FF00))
    assert seen_directional > 200   # the boards do carry per-direction threat bits, so the shuffle is exercised
