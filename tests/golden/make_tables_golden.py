"""Generates tests/golden/tables.json from the REAL reference (oracle/_ref/libagref.so built from /root/reference by
oracle/Makefile).  Run in the build container only; the JSON holds checksums (data), no reference source."""
import ctypes
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_oracle_tables import RULES, valid_extended_patterns  # noqa: E402

ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libagref.so"))
ref.ref_defensive_moves.restype = ctypes.c_uint16
out = {}
for rules, name in enumerate(RULES):
    t = np.zeros(1 << 20, np.uint8)
    h = np.zeros(1 << 20, np.uint8)
    thr = np.zeros(8192, np.uint8)
    ref.ref_pattern_table(rules, t.ctypes.data_as(ctypes.c_void_p), h.ctypes.data_as(ctypes.c_void_p))
    ref.ref_threat_table(rules, thr.ctypes.data_as(ctypes.c_void_p))
    rng = np.random.default_rng(12345)
    acc = hashlib.sha256()
    for p in valid_extended_patterns(rng, 4000):
        for defender in (1, 2):
            for pt in (2, 3, 4, 5, 6):
                acc.update(int(ref.ref_defensive_moves(rules, p, defender, pt)).to_bytes(2, "little"))
    out[name] = dict(pattern_types_sha256=hashlib.sha256(t.tobytes()).hexdigest(),
                     half_open_3_sha256=hashlib.sha256(h.tobytes()).hexdigest(),
                     threats_sha256=hashlib.sha256(thr.tobytes()).hexdigest(),
                     defensive_moves_sha256=acc.hexdigest())
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "tables.json"), "w"), indent=1)
print("written")
