"""Dev script: device engine vs CPU oracle on the GPU box (pattern state, solver, whole games)."""
import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from alphagomoku_amd import selfplay, lib, check

olib = ol.load()
RULES = int(os.environ.get("RULES", "0"))
N = 15
HW = N * N
rng = np.random.default_rng(1)


def random_board(stones):
    b = np.zeros(HW, np.uint8)
    idx = rng.permutation(HW)[:stones]
    for k, i in enumerate(idx):
        b[i] = 1 + (k & 1)
    return b


def clustered_board(stones):
    b = np.zeros((N, N), np.uint8)
    r, c = 7, 7
    for k in range(stones):
        for _ in range(100):
            rr, cc = r + rng.integers(-2, 3), c + rng.integers(-2, 3)
            if 0 <= rr < N and 0 <= cc < N and b[rr, cc] == 0:
                b[rr, cc] = 1 + (k & 1); r, c = rr, cc
                break
    return b.reshape(-1)


def oracle_pattern_state(board, sign, moves):
    pt = np.zeros((HW, 8), np.uint8); th = np.zeros((HW, 2), np.uint8); lists = np.zeros(4096, np.int16)
    mv = np.array(moves, np.uint16)
    n = olib.ago_pattern_state(RULES, N, N, ol.ptr(board), sign, ol.ptr(mv), len(moves), ol.ptr(pt), ol.ptr(th), ol.ptr(lists), 4096)
    return pt, th, lists[:n]


cfg = selfplay.default_config(rules=RULES, n_games=64, max_batch_size=4, max_simulations=100, tss_table_entries=1 << 16,
                              node_capacity=4096, edge_capacity=65536)
pool = selfplay.GeneratorPool(cfg)

# ---------------- 1. pattern state ----------------
G = 64
boards, signs, moves = [], [], []
NM = 12
for g in range(G):
    b = clustered_board(int(rng.integers(0, 40))) if g % 2 else random_board(int(rng.integers(0, 60)))
    stones = int((b != 0).sum())
    sign = 1 if stones % 2 == 0 else 2
    seq, cur, s, done = [], b.copy(), sign, []
    for k in range(NM):
        if done and rng.random() < 0.35:
            seq.append(0); m = done.pop(); cur[(m >> 2 & 127) * N + (m >> 9 & 127)] = 0; s = 3 - s
        else:
            empt = np.flatnonzero(cur == 0); cell = int(rng.choice(empt)); m = s | ((cell // N) << 2) | ((cell % N) << 9)
            seq.append(m); done.append(m); cur[cell] = s; s = 3 - s
    boards.append(b); signs.append(sign); moves.append(seq)
pt, th, lists = pool.debug_pattern_state(np.array(boards), signs, np.array(moves, np.uint16))
bad = 0
for g in range(G):
    opt, oth, ol_ = oracle_pattern_state(boards[g], signs[g], moves[g])
    n = int(lists[g, -1])
    ok = np.array_equal(pt[g], opt) and np.array_equal(th[g], oth) and np.array_equal(lists[g, :n], ol_)
    if not ok:
        bad += 1
        if bad < 3:
            print("pattern mismatch game", g, np.array_equal(pt[g], opt), np.array_equal(th[g], oth), n, len(ol_))
print("pattern state: %d/%d match" % (G - bad, G))

# ---------------- 2. solver on positions ----------------
pool.begin(selfplay.pack_openings([[] for _ in range(64)]))
check(lib.agx_device_synchronize())
boards, signs = [], []
for g in range(64):
    b = clustered_board(int(rng.integers(2, 50)))
    boards.append(b); signs.append(1 if int((b != 0).sum()) % 2 == 0 else 2)
t0 = time.time()
out = pool.debug_solve(np.array(boards), signs)
print("device solve time %.3fs" % (time.time() - t0))
zob = pool.zobrist()
bad = 0
for g in range(64):
    s = olib.ago_solver_create(RULES, N, N, 1 << 16, cfg.zobrist_seed, 100)
    z = np.zeros(4 * HW, np.uint64); olib.ago_solver_zobrist(s, ol.ptr(z))
    assert np.array_equal(z, zob)
    feat = np.zeros(HW, np.uint32); mv = np.zeros(HW, np.uint16); sc = np.zeros(HW, np.uint16)
    fl = ctypes.c_int(); rs = ctypes.c_uint16(); nodes = ctypes.c_int()
    n = olib.ago_solver_solve(s, ol.ptr(boards[g]), signs[g], ol.ptr(feat), ol.ptr(mv), ol.ptr(sc), ctypes.byref(fl), ctypes.byref(rs), ctypes.byref(nodes))
    olib.ago_solver_destroy(s)
    ok = (n == out["counts"][g] and np.array_equal(mv[:n], out["moves"][g, :n]) and np.array_equal(sc[:n], out["scores"][g, :n])
          and rs.value == out["results"][g] and np.array_equal(feat, out["features"][g]) and bool(fl.value & 1) == bool(out["flags"][g] & 1))
    if not ok:
        bad += 1
        if bad < 4:
            print("solve mismatch", g, "n", n, out["counts"][g], "result", rs.value, out["results"][g], "nodes", nodes.value,
                  "feat", np.array_equal(feat, out["features"][g]), "moves eq", np.array_equal(mv[:n], out["moves"][g, :n]) if n == out["counts"][g] else None)
print("solver: %d/64 match" % (64 - bad))
pool.close()

# ---------------- 3. whole games with the stand-in evaluator ----------------
G, B, SIMS = 8, 4, 100
cfg = selfplay.default_config(rules=RULES, n_games=G, max_batch_size=B, max_simulations=SIMS, tss_table_entries=1 << 16,
                              node_capacity=4096, edge_capacity=65536)
pool = selfplay.GeneratorPool(cfg)
ocfg = ol.default_search_config(max_batch_size=B, max_simulations=SIMS, table_entries=1 << 16)
openings = []
games = []
for g in range(G):
    op = np.zeros(64, np.uint16); k = olib.ago_prepare_opening(RULES, N, N, 100 + g, ol.ptr(op))
    openings.append([int(x) for x in op[:k]])
    h = olib.ago_game_create(RULES, N, N, ctypes.byref(ocfg)); olib.ago_game_begin(h, ol.ptr(op), k); games.append(h)
pool.begin(selfplay.pack_openings(openings))


def oracle_root(h):
    rv = ctypes.c_int(); rval = (ctypes.c_float * 2)(); rs = ctypes.c_uint16()
    em = np.zeros(512, np.uint16); ev = np.zeros(512, np.int32); ep = np.zeros(512, np.float32); evl = np.zeros(1024, np.float32)
    es = np.zeros(512, np.uint16); ef = np.zeros(512, np.uint16)
    n = olib.ago_game_root(h, ctypes.byref(rv), rval, ctypes.byref(rs), ol.ptr(em), ol.ptr(ev), ol.ptr(ep), ol.ptr(evl), ol.ptr(es), ol.ptr(ef), 512)
    return dict(n=n, visits=rv.value, win=rval[0], draw=rval[1], score=rs.value, moves=em[:n].copy(), ev=ev[:n].copy(), prior=ep[:n].copy(),
                val=evl[:2 * n].copy(), es=es[:n].copy(), ef=ef[:n].copy())


steps = 0
mismatch = None
t0 = time.time()
while steps < 3000:
    pool.select_solve()
    slots, feats = pool.scheduled()
    # oracle side
    ofeat = {}
    for g in range(G):
        if olib.ago_game_outcome(games[g]) != 0:
            continue
        f = np.zeros((B, HW), np.uint32)
        c = olib.ago_game_step_select(games[g], ol.ptr(f), B)
        ofeat[g] = f[:c]
    n_or = sum(len(v) for v in ofeat.values())
    if n_or != len(slots):
        mismatch = "step %d: scheduled %d vs oracle %d" % (steps, len(slots), n_or)
        dev_counts = np.bincount(np.array(slots) // B, minlength=G)
        for g in range(G):
            info = pool.game_info(g, with_edges=False)
            oc = len(ofeat.get(g, []))
            print("  game", g, "dev sched", dev_counts[g], "oracle sched", oc, "dev active", info["active"], "outcome", info["outcome"], "n_moves", info["n_moves"],
                  "opening", info["opening_id"], "root_visits", info["root_visits"], "| oracle outcome", olib.ago_game_outcome(games[g]),
                  "moves", len(openings[g]) + olib.ago_game_num_records(games[g]))
        break
    # evaluate (stand-in evaluator) for the device slots
    pol = np.zeros((len(slots), HW), np.float32); val = np.zeros((len(slots), 2), np.float32)
    if len(slots):
        olib.ago_fake_eval(len(slots), HW, ol.ptr(np.ascontiguousarray(feats)), ol.ptr(pol), ol.ptr(val))
    v3 = np.concatenate([val, 1 - val.sum(1, keepdims=True)], 1).astype(np.float32)
    pool.provide(slots, pol, v3)
    pool.expand_backup()
    for g, f in ofeat.items():
        p = np.zeros((len(f), HW), np.float32); v = np.zeros((len(f), 2), np.float32)
        if len(f):
            olib.ago_fake_eval(len(f), HW, ol.ptr(np.ascontiguousarray(f)), ol.ptr(p), ol.ptr(v))
        olib.ago_game_step_expand(games[g], ol.ptr(p), ol.ptr(v))
    steps += 1
    if steps % 25 == 0 or steps < 4:
        for g in range(G):
            info = pool.game_info(g)
            if info["error"]:
                mismatch = "device error %d game %d" % (info["error"], g); break
            if olib.ago_game_outcome(games[g]) != 0 or info["opening_id"] != g:
                continue
            r = oracle_root(games[g])
            dm = np.array([e["move"] for e in info["edges"]], np.uint16)
            dv = np.array([e["visits"] for e in info["edges"]], np.int32)
            ok = (r["n"] == info["root_edges"] and r["visits"] == info["root_visits"] and np.array_equal(dm, r["moves"]) and np.array_equal(dv, r["ev"])
                  and np.float32(r["win"]) == np.float32(info["root_win"]) and info["n_moves"] == len(openings[g]) + olib.ago_game_num_records(games[g]))
            if not ok:
                mismatch = "step %d game %d: root mismatch: edges %d/%d visits %d/%d moves %d/%d win %r/%r" % (
                    steps, g, info["root_edges"], r["n"], info["root_visits"], r["visits"], info["n_moves"],
                    len(openings[g]) + olib.ago_game_num_records(games[g]), info["root_win"], r["win"])
                break
        if mismatch:
            break
    if all(olib.ago_game_outcome(h) != 0 for h in games):
        break
print("whole games: steps %d, %.1fs, mismatch: %s" % (steps, time.time() - t0, mismatch))
st = pool.stats()
print({k: v for k, v in st.items()})
recs, redges = pool.records()
print("records", len(recs), "oracle moves", sum(olib.ago_game_num_records(h) for h in games))
# compare played moves per game
bad = 0
for g in range(G):
    dev_moves = [r.move for r in recs if r.game_serial == g]
    om = []
    for i in range(olib.ago_game_num_records(games[g])):
        mv = ctypes.c_uint16(); rv = ctypes.c_int(); rval = (ctypes.c_float * 2)(); rs = ctypes.c_uint16()
        em = np.zeros(512, np.uint16); ev = np.zeros(512, np.int32); ep = np.zeros(512, np.float32); evl = np.zeros(1024, np.float32); es = np.zeros(512, np.uint16)
        olib.ago_game_record(games[g], i, ctypes.byref(mv), ctypes.byref(rv), rval, ctypes.byref(rs), ol.ptr(em), ol.ptr(ev), ol.ptr(ep), ol.ptr(evl), ol.ptr(es), 512)
        om.append(mv.value)
    if dev_moves[:len(om)] != om[:len(dev_moves)] or len(dev_moves) != len(om):
        bad += 1
        print("game", g, "moves differ: dev", len(dev_moves), "oracle", len(om))
print("games with identical move lists: %d/%d" % (G - bad, G))
