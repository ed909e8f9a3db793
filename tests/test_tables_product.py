"""The product's own host-side table construction (alphagomoku_amd/csrc/tables_host.cpp, a bit-parallel formulation
independent of the oracle's) against the oracle tables — which tests/test_oracle_tables.py pins to the compiled reference —
and against the golden checksums derived from the reference."""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RULES = ["FREESTYLE", "STANDARD", "RENJU", "CARO5", "CARO6"]


@pytest.mark.parametrize("rules", range(5))
def test_host_tables_equal_oracle_and_golden(agx_lib, rules):
    pattern = np.zeros(1 << 20, np.uint8)
    ho3 = np.zeros(1 << 20, np.uint8)
    threat = np.zeros(8192, np.uint8)
    defense = np.zeros(15 * 256 * 2, np.uint16)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    assert agx_lib.agx_host_tables(rules, p(pattern), p(ho3), p(threat), p(defense)) == 0
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "tables.json")))[RULES[rules]]
    assert hashlib.sha256(pattern.tobytes()).hexdigest() == golden["pattern_types_sha256"]
    assert hashlib.sha256(ho3.tobytes()).hexdigest() == golden["half_open_3_sha256"]
    assert hashlib.sha256(threat.tobytes()).hexdigest() == golden["threats_sha256"]
    oracle = ol.load()
    t2 = np.zeros(1 << 20, np.uint8)
    h2 = np.zeros(1 << 20, np.uint8)
    thr2 = np.zeros(8192, np.uint8)
    oracle.ago_tables(rules, p(t2), p(h2), p(thr2))
    assert np.array_equal(pattern, t2) and np.array_equal(ho3, h2) and np.array_equal(threat, thr2)
    # the raw defence tables (the oracle's lookups over them are pinned to the reference on random lines)
    d2 = np.zeros(15 * 256 * 2, np.uint16)
    oracle.ago_defense_tables(rules, p(d2))
    assert np.array_equal(defense, d2)
    assert np.count_nonzero(defense) > 1000


def _product_outcome(agx_lib, rules, b, sign, row, col, draw_after=-1):
    out = ctypes.c_int(-1)
    n = b.shape[0]
    assert agx_lib.agx_get_outcome(rules, n, b.ctypes.data_as(ctypes.c_void_p), sign, row, col, draw_after, ctypes.byref(out)) == 0
    return out.value


def test_get_outcome_on_reference_fixtures(agx_lib):
    """agx_get_outcome (win / renju foul / draw, the test k_advance runs on the device) on the boards of the reference's own
    rules tests (test/game/test_*.cpp, extracted into tests/golden/ref_rules_cases.json)."""
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_rules_cases.json")))
    checked = fouls = 0
    for case in cases:
        b = ol.board_array(case["board"])
        if b.shape[0] != b.shape[1]:
            continue
        for chk in case["checks"]:
            m = chk["move"]
            if chk["kind"] == "outcome":
                assert _product_outcome(agx_lib, ol.RULES[chk["rules"]], b, m["sign"], m["row"], m["col"]) == ol.OUTCOMES[chk["expected"]], chk
            else:
                got = _product_outcome(agx_lib, ol.RULES["RENJU"], b, m["sign"], m["row"], m["col"])
                assert (got == 3) == chk["expected"], (case["name"], chk)
                fouls += int(chk["expected"])
            checked += 1
    assert checked > 100 and fouls > 10


@pytest.mark.parametrize("rules", range(5))
def test_get_outcome_equals_oracle_on_random_boards(agx_lib, rules):
    oracle = ol.load()
    rng = np.random.default_rng(500 + rules)
    decided = 0
    for trial in range(300):
        n = 15
        b = np.zeros((n, n), np.uint8)
        # dense clusters so that fives, overlines and (for renju) 3x3 / 4x4 forks actually occur
        k = int(rng.integers(10, 120))
        r0, c0 = rng.integers(3, 12, 2)
        cells = set()
        while len(cells) < k:
            r, c = int(np.clip(r0 + rng.normal(0, 3), 0, n - 1)), int(np.clip(c0 + rng.normal(0, 3), 0, n - 1))
            cells.add((r, c))
        for i, (r, c) in enumerate(cells):
            b[r, c] = 1 if rng.random() < 0.6 else 2
        for _ in range(6):
            r, c = int(rng.integers(0, n)), int(rng.integers(0, n))
            sign = 1 + int(rng.integers(0, 2))
            bb = b.copy()
            bb[r, c] = sign
            want = oracle.ago_outcome(rules, n, n, ol.ptr(bb), sign, r, c, 225)
            assert _product_outcome(agx_lib, rules, bb, sign, r, c, 225) == want, (trial, r, c, sign)
            decided += int(want != 0)
    assert decided > 20


def test_renju_openings_are_legal_and_undecided(agx_lib):
    oracle = ol.load()
    for seed in range(200):
        op = np.zeros(32, np.uint16)
        assert agx_lib.agx_make_opening(2, 15, seed, op.ctypes.data_as(ctypes.c_void_p)) == 0
        b = np.zeros((15, 15), np.uint8)
        last = None
        for k in range(int(op[0])):
            m = int(op[1 + k])
            assert (m & 3) == 1 + (k & 1) and b[(m >> 2) & 127, (m >> 9) & 127] == 0
            b[(m >> 2) & 127, (m >> 9) & 127] = m & 3
            last = m
        if last is not None:
            assert oracle.ago_outcome(2, 15, 15, ol.ptr(b), last & 3, (last >> 2) & 127, (last >> 9) & 127, -1) == 0
