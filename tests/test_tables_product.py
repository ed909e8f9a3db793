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
