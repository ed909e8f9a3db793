"""Dynamic instruction counts of ONE place / undo of the solver (PatternCalculator::addMove / undoMove on the device: solver_place ->
solver_update_around), by difference of two launches of the debug kernel k_debug_pattern_state on the same 2048 positions: (A) set_board only,
(B) set_board + 32 stones placed and removed again.  Run under the profiler, then summarise:

    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES -d gpurun_out/place_pmc -o p -- python3 scripts/place_pmc.py run
    python3 scripts/place_pmc.py summary gpurun_out/place_pmc/p_results.db gpurun_out/r04_place_pmc.json

(The debug kernel is the any-size instantiation of the pattern state; the place / undo code has no division by the board size, so its counts
are those of the 15x15 search kernel.)"""
import json
import os
import sqlite3
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N, BOARDS, PAIRS = 15, 2048, 32


def positions():
    rng = np.random.default_rng(7)
    boards, signs, plain, busy = [], [], [], []
    for g in range(BOARDS):
        b = np.zeros(N * N, np.uint8)
        stones = int(rng.integers(10, 60))
        # clustered stones around the centre: threats exist, lists change when a stone lands next to them (like the solver's positions)
        cells = set()
        while len(cells) < stones:
            r, c = int(np.clip(rng.normal(7, 3), 0, 14)), int(np.clip(rng.normal(7, 3), 0, 14))
            cells.add(r * N + c)
        for k, cell in enumerate(sorted(cells, key=lambda x: rng.random())):
            b[cell] = 1 + k % 2
        sign = 1 if stones % 2 == 0 else 2
        seq, s, cur = [], sign, b.copy()
        for _ in range(PAIRS):   # one stone next to the cluster, then taken off again (the solver's descend / return)
            empties = np.flatnonzero(cur == 0)
            near = [int(x) for x in empties if abs(x // N - 7) <= 4 and abs(x % N - 7) <= 4]
            cell = int(rng.choice(near if near else empties))
            seq += [s | ((cell // N) << 2) | ((cell % N) << 9), 0]
        boards.append(b)
        signs.append(sign)
        plain.append([0xFFFF] * (2 * PAIRS))
        busy.append(seq)
    return np.array(boards), signs, np.array(plain, np.uint16), np.array(busy, np.uint16)


def run():
    from alphagomoku_amd import selfplay
    pool = selfplay.GeneratorPool(selfplay.default_config(n_games=BOARDS, max_batch_size=1, tss_table_entries=1 << 12, node_capacity=256, edge_capacity=4096))
    boards, signs, plain, busy = positions()
    for _ in range(2):
        pool.debug_pattern_state(boards, signs, plain)
        pool.debug_pattern_state(boards, signs, busy)
    pool.close()


def summary(db, out_path):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, dispatch_id, counter_name, sum(counter_value) from pmc_events group by name, dispatch_id, counter_name order by dispatch_id").fetchall()
    per = {}
    for name, dispatch, counter, value in rows:
        if "k_debug_pattern_state" in name:
            per.setdefault(dispatch, {})[counter] = value
    launches = [per[k] for k in sorted(per)]
    assert len(launches) == 4, len(launches)
    a, b = launches[2], launches[3]     # the second pair (the first warms the caches)
    places = BOARDS * PAIRS * 2
    out = {"_comment": __doc__, "boards": BOARDS, "places_and_undos": places,
           "set_board_only_per_wave": {k: v / BOARDS for k, v in a.items()},
           "per_place_or_undo": {k: (b[k] - a[k]) / places for k in a if k.startswith("SQ_INSTS")}}
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "_comment"}, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        summary(sys.argv[2], sys.argv[3])
