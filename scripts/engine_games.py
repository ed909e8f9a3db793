"""Dev script: whole-game parity device vs oracle, every step, with the stand-in evaluator."""
import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from alphagomoku_amd import selfplay, lib, check

olib = ol.load()
RULES = int(os.environ.get("RULES", "0")); N = 15; HW = N * N
G = int(os.environ.get("G", "8")); B = int(os.environ.get("B", "4")); SIMS = int(os.environ.get("SIMS", "100"))
MAXSTEPS = int(os.environ.get("STEPS", "400"))


def oracle_root(h):
    rv = ctypes.c_int(); rval = (ctypes.c_float * 2)(); rs = ctypes.c_uint16()
    em = np.zeros(512, np.uint16); ev = np.zeros(512, np.int32); ep = np.zeros(512, np.float32); evl = np.zeros(1024, np.float32)
    es = np.zeros(512, np.uint16); ef = np.zeros(512, np.uint16)
    n = olib.ago_game_root(h, ctypes.byref(rv), rval, ctypes.byref(rs), ol.ptr(em), ol.ptr(ev), ol.ptr(ep), ol.ptr(evl), ol.ptr(es), ol.ptr(ef), 512)
    return dict(n=n, visits=rv.value, win=np.float32(rval[0]), draw=np.float32(rval[1]), score=rs.value, moves=em[:n].copy(), ev=ev[:n].copy(), prior=ep[:n].copy(),
                val=evl[:2 * n].copy(), es=es[:n].copy(), ef=ef[:n].copy())


def run(compare):
    cfg = selfplay.default_config(rules=RULES, n_games=G, max_batch_size=B, max_simulations=SIMS, tss_table_entries=1 << 16,
                                  node_capacity=4096, edge_capacity=65536)
    pool = selfplay.GeneratorPool(cfg)
    ocfg = ol.default_search_config(max_batch_size=B, max_simulations=SIMS, table_entries=1 << 16)
    openings, games = [], []
    for g in range(G):
        op = np.zeros(64, np.uint16); k = olib.ago_prepare_opening(RULES, N, N, 100 + g, ol.ptr(op))
        openings.append([int(x) for x in op[:k]])
        if compare:
            h = olib.ago_game_create(RULES, N, N, ctypes.byref(ocfg)); olib.ago_game_begin(h, ol.ptr(op), k); games.append(h)
    pool.begin(selfplay.pack_openings(openings))
    trace = []
    mismatch = None
    for step in range(MAXSTEPS):
        pool.select_solve()
        slots, feats = pool.scheduled()
        pol = np.zeros((len(slots), HW), np.float32); val = np.zeros((len(slots), 2), np.float32)
        if len(slots):
            olib.ago_fake_eval(len(slots), HW, ol.ptr(np.ascontiguousarray(feats)), ol.ptr(pol), ol.ptr(val))
        v3 = np.concatenate([val, 1 - val.sum(1, keepdims=True)], 1).astype(np.float32)
        pool.provide(slots, pol, v3)
        if compare:
            dev_counts = np.bincount(np.array(slots, np.int64) // B, minlength=G)
            for g in range(G):
                if olib.ago_game_outcome(games[g]) != 0:
                    continue
                f = np.zeros((B, HW), np.uint32)
                c = olib.ago_game_step_select(games[g], ol.ptr(f), B)
                dslots = sorted(int(s) for s in slots if s // B == g)
                dfe = np.array([feats[list(slots).index(s)] for s in dslots], np.uint32).reshape(-1, HW)
                if c != len(dslots) or not np.array_equal(dfe, f[:c]):
                    mismatch = "step %d game %d: scheduled %d vs oracle %d, features equal %s" % (step, g, len(dslots), c, np.array_equal(dfe, f[:c]) if c == len(dslots) else None)
                p = np.zeros((c, HW), np.float32); v = np.zeros((c, 2), np.float32)
                if c:
                    olib.ago_fake_eval(c, HW, ol.ptr(np.ascontiguousarray(f[:c])), ol.ptr(p), ol.ptr(v))
                olib.ago_game_step_expand(games[g], ol.ptr(p), ol.ptr(v))
        pool.expand_backup()
        for g in range(G):
            info = pool.game_info(g)
            key = (info["n_moves"], info["root_visits"], float(info["root_win"]), float(info["root_draw"]), info["root_score"],
                   tuple((e["move"], e["visits"], e["win"], e["draw"], e["prior"], e["score"]) for e in info["edges"]))
            trace.append(key)
            if compare and mismatch is None and info["opening_id"] == g and info["active"] and olib.ago_game_outcome(games[g]) == 0:
                r = oracle_root(games[g])
                dm = np.array([e["move"] for e in info["edges"]], np.uint16); dv = np.array([e["visits"] for e in info["edges"]], np.int32)
                dw = np.array([[e["win"], e["draw"]] for e in info["edges"]], np.float32).reshape(-1); dp = np.array([e["prior"] for e in info["edges"]], np.float32)
                ds = np.array([e["score"] for e in info["edges"]], np.uint16)
                checks = dict(n=r["n"] == info["root_edges"], visits=r["visits"] == info["root_visits"], moves=np.array_equal(dm, r["moves"]),
                              ev=np.array_equal(dv, r["ev"]), win=r["win"] == np.float32(info["root_win"]), draw=r["draw"] == np.float32(info["root_draw"]),
                              score=r["score"] == info["root_score"], val=np.array_equal(dw, r["val"]), prior=np.array_equal(dp, r["prior"]), es=np.array_equal(ds, r["es"]),
                              nmoves=info["n_moves"] == len(openings[g]) + olib.ago_game_num_records(games[g]))
                if not all(checks.values()):
                    mismatch = "step %d game %d: %s | visits %d/%d root win %r/%r" % (step, g, [k for k, v in checks.items() if not v], info["root_visits"], r["visits"], info["root_win"], r["win"])
                    if not checks["val"] and checks["n"]:
                        idx = np.flatnonzero(dw != r["val"])[:4]
                        mismatch += " val idx %s dev %s or %s" % (idx, dw[idx], r["val"][idx])
        if mismatch:
            break
    st = pool.stats()
    pool.close()
    return trace, mismatch, st


t0 = time.time()
tr1, mm, st = run(True)
print("compare run: steps %d mismatch: %s (%.1fs)" % (len(tr1) // G, mm, time.time() - t0))
print(st)
if os.environ.get("DET", "1") == "1":
    tr2, _, _ = run(False)
    tr3, _, _ = run(False)
    n = min(len(tr2), len(tr3))
    first = next((i for i in range(n) if tr2[i] != tr3[i]), None)
    print("determinism: two device-only runs identical:", first is None, "first diff at entry", first, "(step %s game %s)" % ((first // G, first % G) if first is not None else (None, None)))
