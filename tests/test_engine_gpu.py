"""Bit-exact parity of the device search engine (through the C ABI) against the CPU oracle.

  1. incremental pattern / threat state after random place / remove sequences  (PatternCalculator)
  2. threat solver per position: action ORDER, scores, flags, result, and the encoded NN features (AlphaBetaSearch + MoveGenerator)
  3. whole self-play games, compared after EVERY step: scheduled positions and their features, root edge list (moves, visit
     counts, priors, values, scores), root value, moves played — with the same evaluator outputs fed to both sides
  4. the same with the HIP network in the loop (oracle fed with the device network's outputs)
Integer / index data must be identical; floats are compared bit for bit (the tree arithmetic is restated operation by operation)."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol
from alphagomoku_amd import synthetic

pytestmark = pytest.mark.gpu
N = 15
HW = N * N


@pytest.fixture(scope="module")
def olib():
    return ol.load()


def clustered_board(rng, stones, n=N):
    b = np.zeros((n, n), np.uint8)
    r, c = n // 2, n // 2
    for k in range(stones):
        for _ in range(100):
            rr, cc = r + rng.integers(-2, 3), c + rng.integers(-2, 3)
            if 0 <= rr < n and 0 <= cc < n and b[rr, cc] == 0:
                b[rr, cc] = 1 + (k & 1)
                r, c = rr, cc
                break
    return b.reshape(-1)


def random_board(rng, stones, n=N):
    b = np.zeros(n * n, np.uint8)
    for k, i in enumerate(rng.permutation(n * n)[:stones]):
        b[i] = 1 + (k & 1)
    return b



def _split_threat_lists(flat):
    """[count, (row, col) * count] for side 0/1 x threat type 0..9 -> {(side, type): [(row, col), ...]}"""
    out, pos = {}, 0
    for side in range(2):
        for t in range(10):
            cnt = int(flat[pos])
            pos += 1
            out[(side, t)] = [(int(flat[pos + 2 * k]), int(flat[pos + 2 * k + 1])) for k in range(cnt)]
            pos += 2 * cnt
    assert pos == len(flat)
    return out

@pytest.mark.parametrize("rules", [0, 1, 2, 3, 4])
def test_pattern_state_after_move_sequences(agx_lib, olib, rules):
    from alphagomoku_amd import selfplay
    rng = np.random.default_rng(10 + rules)
    pool = selfplay.GeneratorPool(selfplay.default_config(rules=rules, n_games=64, max_batch_size=2, tss_table_entries=1 << 12,
                                                          node_capacity=256, edge_capacity=4096))
    G, NM = 64, 14
    boards, signs, moves = [], [], []
    for g in range(G):
        b = clustered_board(rng, int(rng.integers(0, 45))) if g % 2 else random_board(rng, int(rng.integers(0, 80)))
        if g == 0:
            b[:] = 0  # empty board edge case
        sign = 1 if int((b != 0).sum()) % 2 == 0 else 2
        seq, cur, s, done = [], b.copy(), sign, []
        for _ in range(NM):
            if done and rng.random() < 0.35:
                seq.append(0)
                m = done.pop()
                cur[(m >> 2 & 127) * N + (m >> 9 & 127)] = 0
            else:
                cell = int(rng.choice(np.flatnonzero(cur == 0)))
                m = s | ((cell // N) << 2) | ((cell % N) << 9)
                seq.append(m)
                done.append(m)
                cur[cell] = s
            s = 3 - s
        boards.append(b)
        signs.append(sign)
        moves.append(seq)
    _compare_pattern_state_with_the_oracle(pool, olib, rules, boards, signs, moves)
    pool.close()


def _compare_pattern_state_with_the_oracle(pool, olib, rules, boards, signs, moves):
    pt, th, lists = pool.debug_pattern_state(np.array(boards), signs, np.array(moves, np.uint16))
    sizes = []
    for g in range(len(boards)):
        opt = np.zeros((HW, 8), np.uint8)
        oth = np.zeros((HW, 2), np.uint8)
        olists = np.zeros(4096, np.int16)
        mv = np.array(moves[g], np.uint16)
        n = olib.ago_pattern_state(rules, N, N, ol.ptr(boards[g]), signs[g], ol.ptr(mv), len(mv), ol.ptr(opt), ol.ptr(oth), ol.ptr(olists), 4096)
        assert np.array_equal(pt[g], opt), g
        assert np.array_equal(th[g], oth), g
        assert int(lists[g, -1]) == n, g
        dl, rl = _split_threat_lists(lists[g, :n]), _split_threat_lists(olists[:n])
        for key in rl:
            if key[1] == 1:  # HALF_OPEN_3: nothing on the path reads this list, the device keeps only its size (and reports the cells row-major)
                assert sorted(dl[key]) == sorted(rl[key]), (g, key)
            else:            # every list the move generator / evaluation reads: same cells in the same ORDER
                assert dl[key] == rl[key], (g, key)
                sizes.append(len(rl[key]))
    return sizes


def _random_move_sequence(rng, board, sign, length):
    seq, cur, s, done = [], board.copy(), sign, []
    for _ in range(length):
        if done and rng.random() < 0.35:
            seq.append(0)
            m = done.pop()
            cur[(m >> 2 & 127) * N + (m >> 9 & 127)] = 0
        else:
            cell = int(rng.choice(np.flatnonzero(cur == 0)))
            m = s | ((cell // N) << 2) | ((cell % N) << 9)
            seq.append(m)
            done.append(m)
            cur[cell] = s
        s = 3 - s
    return seq


@pytest.mark.parametrize("rules", [0, 2])
def test_pattern_state_with_lists_beyond_their_lds_capacity(agx_lib, olib, rules):
    """Threat lists keep 24 entries (OPEN_3: 64) in LDS and the tail in HBM; the ordered list edits of a place / undo take their lean loop only
    while no list can outgrow its LDS part in the call (solver_update_around).  Boards of many parallel threes / twos hold lists of 30-56 cells:
    stones going on and coming off around them cross that boundary in both directions, and every list must still equal the oracle's in ORDER."""
    from alphagomoku_amd import selfplay
    rng = np.random.default_rng(70 + rules)
    pool = selfplay.GeneratorPool(selfplay.default_config(rules=rules, n_games=32, max_batch_size=2, tss_table_entries=1 << 12, node_capacity=256, edge_capacity=4096))
    boards, signs, moves = [], [], []
    for g in range(32):
        b = np.zeros(HW, np.uint8)
        side = 1 + g % 2
        if (g >> 1) % 2 == 0:   # two threes in every other row: OPEN_4 / HALF_OPEN_4 / 4x4-fork cells by the dozen
            for r in range(0, N, 2):
                for c0 in (2, 9):
                    b[r * N + c0:r * N + c0 + 3] = side
        else:                   # three twos in every other row: ~56 OPEN_3 cells
            for r in range(0, N, 2):
                for c0 in (1, 6, 11):
                    b[r * N + c0:r * N + c0 + 2] = side
        empties = np.flatnonzero(b == 0)
        for cell in rng.choice(empties, size=int(rng.integers(0, 6)), replace=False):   # a few stones of the other side in between
            b[cell] = 3 - side
        sign = 1 if int((b != 0).sum()) % 2 == 0 else 2
        boards.append(b)
        signs.append(sign)
        moves.append(_random_move_sequence(rng, b, sign, 14))
    sizes = _compare_pattern_state_with_the_oracle(pool, olib, rules, boards, signs, moves)
    assert max(sizes) > 40 and sum(1 for x in sizes if x > 24) >= 32, (max(sizes), sum(1 for x in sizes if x > 24))
    pool.close()


@pytest.mark.parametrize("rules,max_nodes", [(0, 100), (1, 100), (2, 100), (3, 100), (0, 1000), (2, 1000)])
def test_solver_matches_oracle_per_position(agx_lib, olib, rules, max_nodes):
    """max_nodes 1000 is the budget OpeningGenerator gives the solver (OpeningGenerator.cpp:61)"""
    from alphagomoku_amd import selfplay, lib, check
    rng = np.random.default_rng(20 + rules)
    cfg = selfplay.default_config(rules=rules, n_games=128, max_batch_size=2, tss_table_entries=1 << 16, node_capacity=256, edge_capacity=4096,
                                  tss_max_positions=max_nodes)
    pool = selfplay.GeneratorPool(cfg)
    pool.begin(selfplay.pack_openings([[] for _ in range(128)]))   # clears the transposition tables
    check(lib.agx_device_synchronize())
    boards, signs = [], []
    for g in range(128):
        b = clustered_board(rng, int(rng.integers(0, 60))) if g % 4 else random_board(rng, int(rng.integers(0, 120)))
        boards.append(b)
        signs.append(1 if int((b != 0).sum()) % 2 == 0 else 2)
    out = pool.debug_solve(np.array(boards), signs)
    zob = pool.zobrist()
    proven = fouls = 0
    for g in range(128):
        s = olib.ago_solver_create(rules, N, N, 1 << 16, cfg.zobrist_seed, max_nodes)
        z = np.zeros(4 * HW, np.uint64)
        olib.ago_solver_zobrist(s, ol.ptr(z))
        assert np.array_equal(z, zob)
        feat = np.zeros(HW, np.uint32)
        mv = np.zeros(HW, np.uint16)
        sc = np.zeros(HW, np.uint16)
        fl, rs, nodes = ctypes.c_int(), ctypes.c_uint16(), ctypes.c_int()
        n = olib.ago_solver_solve(s, ol.ptr(boards[g]), signs[g], ol.ptr(feat), ol.ptr(mv), ol.ptr(sc), ctypes.byref(fl), ctypes.byref(rs), ctypes.byref(nodes))
        olib.ago_solver_destroy(s)
        assert n == out["counts"][g], g
        assert np.array_equal(mv[:n], out["moves"][g, :n]), g          # same actions in the same ORDER
        assert np.array_equal(sc[:n], out["scores"][g, :n]), g
        assert rs.value == out["results"][g], g
        assert nodes.value == int(out["nodes"][g]), g                 # AlphaBetaSearch::solve's return value
        assert bool(fl.value & 1) == bool(out["flags"][g] & 1), g     # must_defend
        assert np.array_equal(feat, out["features"][g]), g
        proven += int(((rs.value >> 13) & 3) != 2)
        fouls += int(((feat >> 6) & 1).sum())
    assert proven > 3   # the sample must exercise proven results too
    assert rules != 2 or fouls > 20   # ... and, under renju, forbidden cells (overline, 4x4 and recursive 3x3 checks)
    pool.close()


def _compare_record_sink(olib, pool, handles, rules, n, record_format):
    """the samples quantised by k_advance to dataset format 201 and the finished games framed by the library
    (GameDataStorage::serialize) must equal the oracle's restatement byte for byte"""
    from alphagomoku_amd import selfplay
    recs, edges, samples, ends = pool.fetch_records()
    one = np.zeros(16 + 6 * n * n + 16, np.uint8)
    checked = entries = 0
    for g, h in enumerate(handles):
        mine = sorted((r.move_number, i) for i, r in enumerate(recs) if r.game_serial == g)
        assert len(mine) == olib.ago_game_num_records(h)
        for k, (_, i) in enumerate(mine):
            r = recs[i]
            assert r.game_slot == g and r.game_index == 0 and r.sample_offset >= 0 and r.sample_offset % 4 == 0
            size = olib.ago_game_record_v201(h, k, ol.ptr(one), one.size)
            dev = samples[r.sample_offset:r.sample_offset + r.sample_bytes]
            assert size == r.sample_bytes and np.array_equal(dev, one[:size]), (g, k)
            if record_format & 1:   # ... and the same bytes from the device's own raw snapshot through the oracle's quantiser
                e = edges[r.edge_offset:r.edge_offset + r.n_edges]
                size2 = olib.ago_sample_v201_pack(n, n, r.move_number, len(e), ol.ptr(np.array([x.move for x in e], np.uint16)),
                                                  ol.ptr(np.array([x.visits for x in e], np.int32)), ol.ptr(np.array([x.prior for x in e], np.float32)),
                                                  ol.ptr(np.array([[x.win, x.draw] for x in e], np.float32).reshape(-1)),
                                                  ol.ptr(np.array([x.score for x in e], np.uint16)), r.root_score, r.root_flags, ol.ptr(one), one.size)
                assert size2 == r.sample_bytes and np.array_equal(dev, one[:size2]), (g, k)
            assert (r.outcome != 0) == (k == len(mine) - 1 and olib.ago_game_outcome(h) != 0), (g, k)
            entries += int(dev[12:16].view(np.uint32)[0])
            checked += 1
    finished = [g for g, h in enumerate(handles) if olib.ago_game_outcome(h) != 0]
    assert sorted(e.game_slot for e in ends) == finished
    buffer = selfplay.GameBuffer(rules, n, n)
    assert buffer.collect(pool) == len(finished)
    big = np.zeros(1 << 20, np.uint8)
    by_first = {}
    for i in range(len(finished)):
        data = buffer.game(i)
        by_first[bytes(data)] = i
    for g in finished:
        size = olib.ago_game_storage_v201(handles[g], ol.ptr(big), big.size)
        assert bytes(big[:size]) in by_first, g      # GameDataStorage::serialize, byte for byte
    st = buffer.stats()
    assert st["games"] == len(finished) and st["samples"] == sum(olib.ago_game_num_records(handles[g]) for g in finished)
    assert st["cross_win"] + st["draws"] + st["circle_win"] == len(finished)
    buffer.close()
    assert checked > 0 and entries > checked


def _oracle_root(olib, h):
    rv, rs = ctypes.c_int(), ctypes.c_uint16()
    rval = (ctypes.c_float * 2)()
    em = np.zeros(512, np.uint16)
    ev = np.zeros(512, np.int32)
    ep = np.zeros(512, np.float32)
    evl = np.zeros(1024, np.float32)
    es = np.zeros(512, np.uint16)
    ef = np.zeros(512, np.uint16)
    n = olib.ago_game_root(h, ctypes.byref(rv), rval, ctypes.byref(rs), ol.ptr(em), ol.ptr(ev), ol.ptr(ep), ol.ptr(evl), ol.ptr(es), ol.ptr(ef), 512)
    return dict(n=n, visits=rv.value, win=np.float32(rval[0]), draw=np.float32(rval[1]), score=rs.value, moves=em[:n].copy(), ev=ev[:n].copy(),
                prior=ep[:n].copy(), val=evl[:2 * n].copy(), es=es[:n].copy())


def _play_and_compare(olib, rules, games, batch, sims, max_steps, evaluator, table_entries=1 << 16, n=N, final_selector=0, use_symmetries=0,
                      action_values=0, noise_weight=0.0, noise_type=1, exploration_scaling=0.0, draw_after=0, max_children=0, policy_temperature=1.0,
                      record_format=1, node_capacity=4096, edge_capacity=0, arena_reserve=1.0, groups=1, restore_from=None, **engine_options):
    """evaluator(features uint32 [n][HW]) -> (policy [n][HW] f32, value [n][2] f32 (win, draw)[, q [n][HW][2]]); used for BOTH sides"""
    from alphagomoku_amd import selfplay
    N, HW = n, n * n   # noqa: N806 (shadow the 15x15 module defaults)
    cfg = selfplay.default_config(rules=rules, board_size=n, draw_after=draw_after if draw_after > 0 else n * n, n_games=games, max_batch_size=batch, max_simulations=sims,
                                  tss_table_entries=table_entries, node_capacity=node_capacity,
                                  edge_capacity=edge_capacity if edge_capacity > 0 else (65536 if n <= 15 else 131072), arena_reserve=arena_reserve,
                                  final_selector=final_selector, use_symmetries=use_symmetries, action_values=action_values,
                                  noise_type=noise_type if noise_weight > 0 else 0, noise_weight=noise_weight,
                                  exploration_scaling=exploration_scaling, max_children=max_children, policy_temperature=policy_temperature,
                                  record_format=record_format, record_edge_capacity=games * HW * HW,   # a root edge per empty cell at worst
                                  **engine_options)
    pool = selfplay.GeneratorPool(cfg)
    ocfg = ol.default_search_config(max_batch_size=batch, max_simulations=sims, table_entries=table_entries, final_selector=final_selector,
                                    use_symmetries=use_symmetries, noise_type=noise_type if noise_weight > 0 else 0, noise_weight=noise_weight)
    ocfg.exploration_scaling = exploration_scaling
    if "tss_max_positions" in engine_options:
        ocfg.tss_max_positions = engine_options["tss_max_positions"]
    if max_children > 0:
        ocfg.max_children = max_children
    openings, handles = [], []
    for g in range(games):
        op = np.zeros(256, np.uint16)
        if restore_from is not None:
            # games in flight saved by another engine (agx_engine_save_games) continue here: for the oracle the saved moves are the "opening"
            k = len(restore_from[g]["moves"])
            op[:k] = restore_from[g]["moves"]
        else:
            k = olib.ago_prepare_opening(rules, N, N, 100 + g, ol.ptr(op))
        openings.append([int(x) for x in op[:k]])
        h = olib.ago_game_create_ex(rules, N, N, draw_after, ctypes.byref(ocfg))
        olib.ago_game_set_serial(h, g)   # the device keys the symmetry hash by the opening id
        olib.ago_game_set_policy_temperature(h, policy_temperature)
        olib.ago_game_begin(h, ol.ptr(op), k)
        handles.append(h)
    if restore_from is not None:
        pool.begin(selfplay.pack_openings([[] for _ in range(games)]))
        for g in range(games):
            pool.restore_game(dict(restore_from[g], opening_id=g, nn_queued=0), slot=g)
    else:
        pool.begin(selfplay.pack_openings(openings))
    compared = 0
    deferred = [False] * games
    # groups > 1: the pool stepped as slices on CU-masked streams, the way bench.py and ag::GeneratorThread run it
    streams = selfplay.chip_slices(groups)[0] if groups > 1 else [None]
    for step in range(max_steps):
        if groups > 1:
            for k in range(groups):
                pool.select_solve_group(k, groups, streams[k])
            parts = [pool.scheduled_group(k, groups) for k in range(groups)]
            slots, feats = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
        else:
            pool.select_solve()
            slots, feats = pool.scheduled()
        out = evaluator(feats) if len(slots) else (np.zeros((0, HW), np.float32), np.zeros((0, 2), np.float32), np.zeros((0, HW, 2), np.float32))
        pol, val = out[0], out[1]
        qv = np.ascontiguousarray(out[2], dtype=np.float32) if action_values else None
        v3 = np.concatenate([val, 1 - val.sum(1, keepdims=True)], 1).astype(np.float32)
        pool.provide(slots, pol, v3, qv)
        by_slot = {int(s): i for i, s in enumerate(slots)}
        for g in range(games):
            if olib.ago_game_outcome(handles[g]) != 0:
                continue
            if deferred[g]:
                # the game's previous batch waited for larger arenas: this step the device only expands it (no select, nothing scheduled);
                # the oracle already holds that batch's results
                assert not any(s // batch == g for s in by_slot), (step, g)
                continue
            f = np.zeros((batch, HW), np.uint32)
            c = olib.ago_game_step_select(handles[g], ol.ptr(f), batch)
            mine = sorted(s for s in by_slot if s // batch == g)
            assert c == len(mine), (step, g)
            idx = [by_slot[s] for s in mine]
            assert np.array_equal(feats[idx], f[:c]), (step, g)                      # same leaves, same features
            p = np.ascontiguousarray(pol[idx])
            v = np.ascontiguousarray(val[idx])
            if action_values:
                olib.ago_game_step_expand_q(handles[g], ol.ptr(p), ol.ptr(v), ol.ptr(np.ascontiguousarray(qv[idx])))
            else:
                olib.ago_game_step_expand(handles[g], ol.ptr(p), ol.ptr(v))
        if groups > 1:
            for k in range(groups):
                pool.expand_backup_group(k, groups, streams[k])
        else:
            pool.expand_backup()
        for g in range(games):
            info = pool.game_info(g)
            assert info["error"] == 0
            deferred[g] = info["grow_pending"] != 0
            if deferred[g]:
                continue   # compared again once the batch has been expanded
            if info["opening_id"] != g or not info["active"] or olib.ago_game_outcome(handles[g]) != 0:
                continue
            r = _oracle_root(olib, handles[g])
            e = info["edges"]
            assert info["n_moves"] == len(openings[g]) + olib.ago_game_num_records(handles[g]), (step, g)
            assert r["n"] == info["root_edges"] and r["visits"] == info["root_visits"], (step, g)
            assert np.array_equal(np.array([x["move"] for x in e], np.uint16), r["moves"]), (step, g)
            assert np.array_equal(np.array([x["visits"] for x in e], np.int32), r["ev"]), (step, g)   # visit counts
            assert np.array_equal(np.array([x["score"] for x in e], np.uint16), r["es"]), (step, g)
            assert np.array_equal(np.array([x["prior"] for x in e], np.float32), r["prior"]), (step, g)
            assert np.array_equal(np.array([[x["win"], x["draw"]] for x in e], np.float32).reshape(-1), r["val"]), (step, g)
            assert np.float32(info["root_win"]) == r["win"] and np.float32(info["root_draw"]) == r["draw"] and info["root_score"] == r["score"], (step, g)
            # Tree::getMovesLeft / getMaximumDepth (Tree.cpp:173-176,188-191): the root's running mean of the backed-up moves-left estimates,
            # the longest select path that reached a leaf since the last setBoard
            tree_f, tree_i = np.zeros(1, np.float32), np.zeros(5, np.int32)
            olib.ago_game_tree_info(handles[g], ol.ptr(tree_f), ol.ptr(tree_i))
            assert np.float32(info["root_moves_left"]) == tree_f[0] and info["max_depth"] == tree_i[0], (step, g, info["root_moves_left"], tree_f[0], info["max_depth"], tree_i[0])
            compared += 1
        if all(olib.ago_game_outcome(h) != 0 for h in handles) and not any(deferred):
            break
    # the played moves (the path's output records) must be identical
    recs, _ = pool.records()
    for g in range(games):
        dev_moves = [r.move for r in recs if r.game_serial == g]
        n = olib.ago_game_num_records(handles[g])
        assert [r.root_flags for r in recs if r.game_serial == g] == [olib.ago_game_record_flags(handles[g], i) for i in range(n)], g
        om = []
        for i in range(n):
            mv, rv, rs = ctypes.c_uint16(), ctypes.c_int(), ctypes.c_uint16()
            rval = (ctypes.c_float * 2)()
            em = np.zeros(512, np.uint16)
            ev = np.zeros(512, np.int32)
            ep = np.zeros(512, np.float32)
            evl = np.zeros(1024, np.float32)
            es = np.zeros(512, np.uint16)
            olib.ago_game_record(handles[g], i, ctypes.byref(mv), ctypes.byref(rv), rval, ctypes.byref(rs), ol.ptr(em), ol.ptr(ev), ol.ptr(ep), ol.ptr(evl), ol.ptr(es), 512)
            om.append(mv.value)
        assert dev_moves == om, g
    if record_format & 2:
        _compare_record_sink(olib, pool, handles, rules, N, record_format)
    stats = pool.stats()
    pool.close()
    for h in handles:
        olib.ago_game_destroy(h)
    return compared, stats


def _stand_in_evaluator(olib, hw=HW):
    def f(feats):
        feats = np.ascontiguousarray(feats, dtype=np.uint32)
        pol = np.zeros((len(feats), hw), np.float32)
        val = np.zeros((len(feats), 2), np.float32)
        olib.ago_fake_eval(len(feats), hw, ol.ptr(feats), ol.ptr(pol), ol.ptr(val))
        return pol, val
    return f


@pytest.mark.parametrize("rules,batch,sims", [(0, 1, 100), (0, 8, 100), (1, 4, 100), (2, 4, 100), (2, 8, 60), (3, 4, 60)])
def test_whole_games_bit_exact_with_stand_in_evaluator(agx_lib, olib, rules, batch, sims):
    compared, stats = _play_and_compare(olib, rules, games=6, batch=batch, sims=sims, max_steps=4000, evaluator=_stand_in_evaluator(olib))
    assert compared > 500
    assert stats["games_finished"] == 6 and stats["information_leaks"] > 0 and stats["proven_edge_visits"] > 0


@pytest.mark.parametrize("rules,n,per_cu", [(0, 15, 16), (2, 15, 16), (3, 20, 12), (0, 12, 12)])
def test_default_search_waves_are_what_stays_resident(agx_lib, rules, n, per_cu):
    """AgxEngineConfig.speculative_waves = 0: the search launch gets as many one-wave workgroups as its kernel instantiation keeps resident — sixteen per
    compute unit on 15x15 boards (10 240 B of LDS state, 128 registers), twelve for the 20x20 and the any-size kernels (13 312 B, 168 registers) — and
    agx_engine_speculative_waves reports the number; an explicit count is taken as given, the serial solver reports 0."""
    from alphagomoku_amd import selfplay, lib, check
    cus = ctypes.c_int()
    check(lib.agx_device_cu_count(ctypes.byref(cus)))
    common = dict(rules=rules, board_size=n, n_games=1024, max_batch_size=8, max_simulations=50, tss_table_entries=1 << 12, node_capacity=256, edge_capacity=8192)
    pool = selfplay.GeneratorPool(selfplay.default_config(speculative_solver=1, **common))
    assert pool.speculative_waves() == per_cu * cus.value
    pool.close()
    pool = selfplay.GeneratorPool(selfplay.default_config(speculative_solver=1, speculative_waves=96, **common))
    assert pool.speculative_waves() == 96
    pool.close()
    pool = selfplay.GeneratorPool(selfplay.default_config(speculative_solver=0, **common))
    assert pool.speculative_waves() == 0
    pool.close()


@pytest.mark.parametrize("rules,n,batch,sims,symmetries,table_entries", [(0, 15, 8, 100, 0, 1 << 16), (1, 15, 8, 100, 1, 1 << 10), (2, 15, 8, 60, 0, 1 << 16),
                                                                         (3, 20, 8, 60, 0, 1 << 12), (4, 15, 4, 60, 1, 1 << 16), (0, 12, 8, 60, 0, 1 << 8)])
def test_speculative_solver_plays_the_same_games(agx_lib, olib, rules, n, batch, sims, symmetries, table_entries):
    """AgxEngineConfig.speculative_solver: the leaves of a batch are solved in parallel against the pre-batch transposition table and committed
    in batch order (k_search_spec) — the oracle solves them one after the other (Search::solve, Search.cpp:159-183) and every leaf's features,
    every root after every step and every played move must still be identical.  Small tables (down to 64 buckets) force bucket collisions
    between the leaves of a batch, i.e. the conflict / serial re-run path."""
    compared, stats = _play_and_compare(olib, rules, games=16, batch=batch, sims=sims, max_steps=400, evaluator=_stand_in_evaluator(olib, n * n), n=n,
                                        use_symmetries=symmetries, table_entries=table_entries, speculative_solver=1, speculative_waves=96)
    assert compared > 100 and stats["first_error"] == 0
    assert stats["speculative_solves"] > 0
    if table_entries <= 1 << 10:
        assert stats["speculative_reruns"] > 0   # the re-run path was exercised


@pytest.mark.parametrize("rules,speculative", [(0, 1), (2, 0)])
def test_sliced_pool_matches_the_oracle(agx_lib, olib, rules, speculative):
    """the pool stepped as 4 slices on CU-masked streams (bench.py's default) against the ORACLE, step by step"""
    compared, stats = _play_and_compare(olib, rules, games=16, batch=8, sims=60, max_steps=300, evaluator=_stand_in_evaluator(olib), groups=4,
                                        speculative_solver=speculative, speculative_waves=192, record_format=3)
    assert compared > 100 and stats["first_error"] == 0


def test_speculative_solver_with_a_full_overlay(agx_lib, olib):
    """a solver budget of 250 positions touches more buckets than a task's overlay holds now and then: those solves are abandoned and
    repeated serially, straight on the table"""
    compared, stats = _play_and_compare(olib, 0, games=8, batch=8, sims=60, max_steps=150, evaluator=_stand_in_evaluator(olib), speculative_solver=1,
                                        speculative_waves=64, tss_max_positions=250)
    assert compared > 50 and stats["first_error"] == 0 and stats["speculative_solves"] > 0


@pytest.mark.parametrize("rules", [0, 2])
def test_separate_select_and_solve_launches(agx_lib, olib, rules, monkeypatch):
    """AGX_FUSE_SELECT=0: Search::select and the solver as two launches (k_select + k_solve<.., false>, what agx_engine_select_group /
    agx_engine_solve_group and tournament pools use) play the same games as the fused launch that every other test goes through"""
    monkeypatch.setenv("AGX_FUSE_SELECT", "0")   # read by agx_engine_create
    compared, stats = _play_and_compare(olib, rules, games=4, batch=8, sims=80, max_steps=4000, evaluator=_stand_in_evaluator(olib))
    assert compared > 300 and stats["games_finished"] == 4


@pytest.mark.parametrize("selector", [1, 2, 3, 4, 5])
def test_final_move_selectors(agx_lib, olib, selector):
    """GameGenerator::make_move with final_selector max_visit / min_visit / max_value / max_policy / lcb instead of "best" """
    compared, stats = _play_and_compare(olib, 0, games=4, batch=4, sims=60, max_steps=4000, evaluator=_stand_in_evaluator(olib), final_selector=selector)
    assert compared > 200 and stats["games_finished"] == 4


@pytest.mark.parametrize("rules,weight,kind", [(0, 0.25, 1), (1, 0.5, 1), (0, 0.25, 2), (2, 0.25, 2), (0, 0.5, 3)])
def test_root_noise(agx_lib, olib, rules, weight, kind):
    """EdgeSelectorConfig noise_type "custom" / "dirichlet" / "gumbel": the root priors of every move are mixed with noise drawn
    once per move by the first select that sees an expanded root; the games differ from the noise-free ones and match the oracle
    bit for bit (the generators use only IEEE-exact operations, csrc/root_noise.hpp)"""
    compared, stats = _play_and_compare(olib, rules, games=4, batch=4, sims=60, max_steps=4000, evaluator=_stand_in_evaluator(olib), noise_weight=weight,
                                        noise_type=kind)
    assert compared > 200 and stats["games_finished"] == 4
    _, plain = _play_and_compare(olib, rules, games=4, batch=4, sims=60, max_steps=4000, evaluator=_stand_in_evaluator(olib))
    assert (plain["moves_played"], plain["evaluated_nodes"]) != (stats["moves_played"], stats["evaluated_nodes"])


@pytest.mark.parametrize("rules,draw_after", [(0, 28), (2, 40)])
def test_short_draw_limit(agx_lib, olib, rules, draw_after):
    """GameConfig::draw_after far below the board size: most games end as draws, the solver's distance-to-draw stages
    (MoveGenerator.cpp:159-223) and the draw-rate reduction of the playout budget (misc.cpp:171-179) are exercised"""
    compared, stats = _play_and_compare(olib, rules, games=6, batch=4, sims=60, max_steps=4000, evaluator=_stand_in_evaluator(olib), draw_after=draw_after)
    assert compared > 100 and stats["games_finished"] == 6


@pytest.mark.parametrize("rules,max_children", [(0, 12), (1, 30), (2, 5)])
def test_max_children_pruning(agx_lib, olib, rules, max_children):
    """MCTSConfig::max_children: non-root nodes keep the best max_children edges (proven scores first, then prior) and of those
    only the ones above the scaled expansion threshold (prune_weak_moves, EdgeGenerator.cpp:69-83); the root is never pruned"""
    compared, stats = _play_and_compare(olib, rules, games=4, batch=4, sims=80, max_steps=4000, evaluator=_stand_in_evaluator(olib),
                                        max_children=max_children)
    assert compared > 200 and stats["games_finished"] == 4
    _, plain = _play_and_compare(olib, rules, games=4, batch=4, sims=80, max_steps=4000, evaluator=_stand_in_evaluator(olib))
    assert plain["peak_edges"] > stats["peak_edges"]      # pruned trees are smaller


@pytest.mark.parametrize("rules,temperature", [(0, 0.5), (1, 2.0), (0, 0.0)])
def test_policy_temperature(agx_lib, olib, rules, temperature):
    """MCTSConfig::policy_temperature (initialize_edges, EdgeGenerator.cpp:88-127): priors policy^(1/T), or the arg-max indicator for
    T = 0; the power is exp(log(p)/T) with the fixed series on both sides, so the games match the oracle bit for bit"""
    compared, stats = _play_and_compare(olib, rules, games=4, batch=4, sims=60, max_steps=4000, evaluator=_stand_in_evaluator(olib),
                                        policy_temperature=temperature, draw_after=90)
    assert compared > 150 and stats["games_finished"] == 4
    _, plain = _play_and_compare(olib, rules, games=4, batch=4, sims=60, max_steps=4000, evaluator=_stand_in_evaluator(olib), draw_after=90)
    assert (plain["moves_played"], plain["evaluated_nodes"]) != (stats["moves_played"], stats["evaluated_nodes"])


def test_exploration_scaling(agx_lib, olib):
    """c_puct = c0 + c1 * ln(N + vl) (EdgeSelector.cpp:1139): the logarithm is the fixed series on both sides"""
    compared, stats = _play_and_compare(olib, 0, games=4, batch=4, sims=80, max_steps=4000, evaluator=_stand_in_evaluator(olib), exploration_scaling=0.35)
    assert compared > 200 and stats["games_finished"] == 4


@pytest.mark.parametrize("rules,n", [(0, 15), (2, 15), (3, 20)])
def test_input_symmetries(agx_lib, olib, rules, n):
    """NNEvaluator::useSymmetries: features handed to the network are augmented (board symmetry + direction-bit shuffle), the
    policy is mapped back with the inverse symmetry; the stand-in evaluator is not equivariant, so any slip changes the games"""
    compared, stats = _play_and_compare(olib, rules, games=4, batch=4, sims=60, max_steps=6000, evaluator=_stand_in_evaluator(olib, n * n), n=n,
                                        use_symmetries=1)
    assert compared > 200 and stats["games_finished"] == 4


@pytest.mark.parametrize("rules,n", [(0, 12), (1, 19)])
def test_whole_games_on_other_board_sizes(agx_lib, olib, rules, n):
    """boards other than 15x15 / 20x20 run the generic solver kernel (board size not a compile-time constant)"""
    compared, stats = _play_and_compare(olib, rules, games=4, batch=4, sims=50, max_steps=6000, evaluator=_stand_in_evaluator(olib, n * n), n=n)
    assert compared > 200 and stats["games_finished"] == 4


@pytest.mark.parametrize("rules,batch,sims", [(3, 8, 60), (0, 4, 60)])
def test_whole_games_on_the_20x20_board(agx_lib, olib, rules, batch, sims):
    """BASELINE configs[3] shape (caro, 20x20): solver lists, node-cache board words and record sizes at the largest board"""
    compared, stats = _play_and_compare(olib, rules, games=4, batch=batch, sims=sims, max_steps=6000, evaluator=_stand_in_evaluator(olib, 400), n=20)
    assert compared > 300
    assert stats["games_finished"] == 4


@pytest.mark.parametrize("rules,n,record_format", [(0, 15, 3), (2, 15, 3), (3, 20, 3), (1, 15, 2), (0, 20, 2)])
def test_record_sink_format_201(agx_lib, olib, rules, n, record_format):
    """every played move's root quantised ON THE DEVICE to SearchDataStorage_v201 bytes (3 scales, score, move number, flags,
    6-byte entries; 20x20 boards exercise the ">= 255 cells" filler rule) == the oracle's loadFrom + serialize; finished games framed as
    GameDataStorage::serialize == the oracle's (dataset/SearchDataStorage.cpp:326-419, dataset/GameDataStorage.cpp:217-250)"""
    compared, stats = _play_and_compare(olib, rules, games=5, batch=4, sims=60, max_steps=6000, evaluator=_stand_in_evaluator(olib, n * n), n=n,
                                        record_format=record_format)
    assert compared > 200 and stats["games_finished"] == 5


@pytest.mark.parametrize("rules,batch", [(0, 8), (2, 4)])
def test_arenas_grow_on_demand(agx_lib, olib, rules, batch):
    """NodeCache::resize / ObjectPool growth (NodeCache.cpp:320-355, utils/ObjectPool.hpp:74-289): with class-0 arenas far too small for a
    search (64 nodes, 1024 edges per game) every game has to move into larger bundles several times — the games must still be the
    oracle's, step by step, no game may stop, and finished games hand their grown bundles back"""
    compared, stats = _play_and_compare(olib, rules, games=6, batch=batch, sims=100, max_steps=6000, evaluator=_stand_in_evaluator(olib), node_capacity=64,
                                        edge_capacity=1024, arena_reserve=60.0)
    assert compared > 300 and stats["games_finished"] == 6 and stats["first_error"] == 0
    assert stats["arena_grows"] >= 12 and stats["arena_failures"] == 0      # two size classes or more per game
    assert stats["arena_releases"] >= 1          # finished games hand their grown bundles back (and sit in class 0 again)


def test_exhausted_arena_reserve_is_reported(agx_lib, olib):
    """without a reserve the old behaviour remains: a tree that outgrows its arenas stops its game with an error code, the others play on"""
    from alphagomoku_amd import selfplay
    cfg = selfplay.default_config(n_games=4, max_batch_size=4, max_simulations=100, tss_table_entries=1 << 12, node_capacity=64, edge_capacity=1024,
                                  arena_reserve=0.0)
    pool = selfplay.GeneratorPool(cfg)
    pool.begin(selfplay.pack_openings(synthetic.make_openings(15, 4, seed0=7)))
    ev = _stand_in_evaluator(olib)
    for _ in range(200):
        pool.select_solve()
        slots, feats = pool.scheduled()
        if len(slots):
            pol, val = ev(feats)
            pool.provide(slots, pol, np.concatenate([val, 1 - val.sum(1, keepdims=True)], 1).astype(np.float32))
        pool.expand_backup()
    st = pool.stats()
    assert st["first_error"] in (1, 2, 5) and st["arena_failures"] > 0 and st["arena_grows"] == 0
    pool.close()


@pytest.mark.parametrize("rules,threads,batch,speculative", [(0, 4, 8, 0), (1, 8, 4, 0), (2, 3, 4, 0), (0, 4, 8, 1), (2, 3, 4, 1)])
def test_tournament_search_on_one_tree(agx_lib, olib, rules, threads, batch, speculative):
    """SURVEY row f4 (player/SearchThread.cpp:121-180): ONE tree searched by several SearchThreads, each with its own Search (task buffer,
    threat solver, table).  The device runs the threads in lock-step — select in thread order under the tree 'lock' (one wave), the
    solvers in parallel (one wave per thread), one network launch, expand + backup in thread order — and the oracle plays the same
    schedule: same leaves, same features, same root after every step, same moves, for whole games."""
    from alphagomoku_amd import selfplay
    sims = 300
    cfg = selfplay.default_config(rules=rules, n_games=threads, search_threads=threads, max_batch_size=batch, max_simulations=sims,
                                  tss_table_entries=1 << 16, node_capacity=4096, edge_capacity=65536, speculative_solver=speculative)   # (1: every thread's leaves solved in parallel)
    pool = selfplay.GeneratorPool(cfg)
    ocfg = ol.default_search_config(max_batch_size=batch, max_simulations=sims, table_entries=1 << 16)
    op = np.zeros(64, np.uint16)
    k = olib.ago_prepare_opening(rules, N, N, 77 + rules, ol.ptr(op))
    h = olib.ago_game_create_ex(rules, N, N, 0, ctypes.byref(ocfg))
    olib.ago_game_set_search_threads(h, threads)
    olib.ago_game_set_serial(h, 0)
    olib.ago_game_begin(h, ol.ptr(op), k)
    pool.begin(selfplay.pack_openings([[int(x) for x in op[:k]]]))
    ev = _stand_in_evaluator(olib)
    compared, widest = 0, 0
    for step in range(3000):
        pool.select_solve()
        slots, feats = pool.scheduled()
        order = np.argsort(slots)                       # thread-major, task order inside a thread: the oracle's queue order
        slots, feats = slots[order], feats[order]
        pol, val = ev(feats) if len(slots) else (np.zeros((0, HW), np.float32), np.zeros((0, 2), np.float32))
        pool.provide(slots, pol, np.concatenate([val, 1 - val.sum(1, keepdims=True)], 1).astype(np.float32))
        f = np.zeros((threads * batch, HW), np.uint32)
        c = olib.ago_game_step_select(h, ol.ptr(f), threads * batch)
        assert c == len(slots), step
        assert np.array_equal(feats, f[:c]), step
        widest = max(widest, len({int(s) // batch for s in slots}))
        olib.ago_game_step_expand(h, ol.ptr(np.ascontiguousarray(pol)), ol.ptr(np.ascontiguousarray(val)))
        pool.expand_backup()
        info = pool.game_info(0)
        assert info["error"] == 0 and info["grow_pending"] == 0
        if olib.ago_game_outcome(h) != 0:
            break
        r = _oracle_root(olib, h)
        e = info["edges"]
        assert info["n_moves"] == k + olib.ago_game_num_records(h), step
        assert r["n"] == info["root_edges"] and r["visits"] == info["root_visits"], step
        assert np.array_equal(np.array([x["move"] for x in e], np.uint16), r["moves"]), step
        assert np.array_equal(np.array([x["visits"] for x in e], np.int32), r["ev"]), step
        assert np.array_equal(np.array([x["score"] for x in e], np.uint16), r["es"]), step
        assert np.array_equal(np.array([[x["win"], x["draw"]] for x in e], np.float32).reshape(-1), r["val"]), step
        compared += 1
    assert olib.ago_game_outcome(h) != 0 and pool.game_info(0, with_edges=False)["outcome"] == olib.ago_game_outcome(h)
    recs, _ = pool.records()
    assert len(recs) == olib.ago_game_num_records(h)
    for i, r in enumerate(sorted(recs, key=lambda x: x.move_number)):
        mv, rv, rs = ctypes.c_uint16(), ctypes.c_int(), ctypes.c_uint16()
        rval = (ctypes.c_float * 2)()
        em, evv, ep, evl, es = np.zeros(512, np.uint16), np.zeros(512, np.int32), np.zeros(512, np.float32), np.zeros(1024, np.float32), np.zeros(512, np.uint16)
        olib.ago_game_record(h, i, ctypes.byref(mv), ctypes.byref(rv), rval, ctypes.byref(rs), ol.ptr(em), ol.ptr(evv), ol.ptr(ep), ol.ptr(evl), ol.ptr(es), 512)
        assert r.move == mv.value and r.root_visits == rv.value
    assert compared > 100 and widest == threads      # every thread found work in some step
    pool.close()
    olib.ago_game_destroy(h)


@pytest.mark.parametrize("rules,threads,batch,node_capacity,seed,speculative", [(0, 1, 8, 4096, 177, 1), (1, 4, 4, 4096, 181, 0), (2, 3, 8, 4096, 182, 1),
                                                                                (0, 2, 8, 256, 181, 1), (0, 1, 8, 4096, 184, 0)])
def test_double_buffered_tournament_search(agx_lib, olib, rules, threads, batch, node_capacity, seed, speculative):
    """SURVEY row f4, the double buffering (player/SearchThread.cpp:148-180 asynchronous_run, Search.cpp:243-252 useBuffer / switchBuffer):
    every search thread has two task buffers; while buffer b's leaves are with the network — virtual losses applied — buffer 1 - b is
    expanded, backed up, selected and solved.  The device steps the pool buffer by buffer (group b of 2: expand_backup, select_solve,
    network), the oracle runs asynchronous_run's loop body per thread with the same fixed thread order: the same leaves and features in
    every iteration, the same root after it, the same moves, for whole games.  When the move rule fires, the other buffer's leaves are
    dropped and their virtual losses taken back (Search::cleanup).  The last case starts with small arenas: a buffer whose expansion
    waits for larger ones must still go before the other buffer.  speculative: the leaves of every buffer solved in parallel (k_search_spec
    behind the one-wave select)."""
    from alphagomoku_amd import selfplay
    sims = 300
    cfg = selfplay.default_config(rules=rules, n_games=2 * threads, search_threads=threads, search_buffers=2, max_batch_size=batch, max_simulations=sims,
                                  tss_table_entries=1 << 16, node_capacity=node_capacity, edge_capacity=65536 if node_capacity >= 4096 else 8192,
                                  arena_reserve=1.0 if node_capacity >= 4096 else 8.0, speculative_solver=speculative)
    pool = selfplay.GeneratorPool(cfg)
    ocfg = ol.default_search_config(max_batch_size=batch, max_simulations=sims, table_entries=1 << 16)
    op = np.zeros(64, np.uint16)
    k = olib.ago_prepare_opening(rules, N, N, seed, ol.ptr(op))   # (openings whose games last 120-670 iterations)
    h = olib.ago_game_create_ex(rules, N, N, 0, ctypes.byref(ocfg))
    olib.ago_game_set_search_threads(h, threads)
    olib.ago_game_set_serial(h, 0)
    olib.ago_game_begin(h, ol.ptr(op), k)
    pool.begin(selfplay.pack_openings([[int(x) for x in op[:k]]]))
    with pytest.raises(RuntimeError, match="buffer by buffer"):
        pool.select_solve()                                   # a double-buffered pool is stepped as two groups
    ev = _stand_in_evaluator(olib)
    compared, dropped, widest, sat_out = 0, 0, 0, 0
    for step in range(6000):
        b = step % 2
        before = pool.game_info(0, with_edges=False)["n_moves"]
        pool.expand_backup_group(b, 2)
        if pool.game_info(0, with_edges=False)["n_moves"] != before and pool.game_info(threads * (1 - b), with_edges=False)["outcome"] == 0:
            dropped += 1                                      # a move was made with the other buffer in flight
        growing = pool.game_info(0, with_edges=False)["grow_pending"] != 0
        pool.select_solve_group(b, 2)
        slots, feats = pool.scheduled_group(b, 2)
        if growing:
            # a buffer's expansion waits for larger arenas: the tree sits out this iteration AND the other buffer's next one (the waiting
            # buffer must be expanded first, as in the oracle's order), nothing is selected or scheduled meanwhile
            assert len(slots) == 0, step
            sat_out += 1
            continue
        order = np.argsort(slots)                             # thread-major, task order inside a thread: the oracle's queue order
        slots, feats = slots[order], feats[order]
        assert all(b * threads * batch <= int(s) < (b + 1) * threads * batch for s in slots), step
        pol, val = ev(feats) if len(slots) else (np.zeros((0, HW), np.float32), np.zeros((0, 2), np.float32))
        pool.provide(slots, pol, np.concatenate([val, 1 - val.sum(1, keepdims=True)], 1).astype(np.float32))
        f = np.zeros((threads * batch, HW), np.uint32)
        c = olib.ago_game_async_step(h, ol.ptr(f), threads * batch)
        assert c == len(slots), step
        assert np.array_equal(feats, f[:c]), step
        olib.ago_game_async_provide(h, ol.ptr(np.ascontiguousarray(pol)), ol.ptr(np.ascontiguousarray(val)))
        widest = max(widest, len({int(s) // batch for s in slots}))
        info = pool.game_info(0)
        assert info["error"] == 0, step
        if olib.ago_game_outcome(h) != 0:
            break
        if info["root_edges"] == 0:
            continue                                          # nothing expanded yet (the first iterations of a search)
        r = _oracle_root(olib, h)
        e = info["edges"]
        assert info["n_moves"] == k + olib.ago_game_num_records(h), step
        assert r["n"] == info["root_edges"] and r["visits"] == info["root_visits"], step
        assert np.array_equal(np.array([x["move"] for x in e], np.uint16), r["moves"]), step
        assert np.array_equal(np.array([x["visits"] for x in e], np.int32), r["ev"]), step
        assert np.array_equal(np.array([x["score"] for x in e], np.uint16), r["es"]), step
        assert np.array_equal(np.array([[x["win"], x["draw"]] for x in e], np.float32).reshape(-1), r["val"]), step
        compared += 1
    assert olib.ago_game_outcome(h) != 0 and pool.game_info(0, with_edges=False)["outcome"] == olib.ago_game_outcome(h)
    recs, _ = pool.records()
    assert len(recs) == olib.ago_game_num_records(h)
    for i, r in enumerate(sorted(recs, key=lambda x: x.move_number)):
        mv, rv, rs = ctypes.c_uint16(), ctypes.c_int(), ctypes.c_uint16()
        rval = (ctypes.c_float * 2)()
        em, evv, ep, evl, es = np.zeros(512, np.uint16), np.zeros(512, np.int32), np.zeros(512, np.float32), np.zeros(1024, np.float32), np.zeros(512, np.uint16)
        olib.ago_game_record(h, i, ctypes.byref(mv), ctypes.byref(rv), rval, ctypes.byref(rs), ol.ptr(em), ol.ptr(evv), ol.ptr(ep), ol.ptr(evl), ol.ptr(es), 512)
        assert r.move == mv.value and r.root_visits == rv.value
    st = pool.stats()
    assert compared > 100 and widest == threads and dropped > 0
    if node_capacity < 4096:
        assert st["arena_grows"] > 0 and st["arena_failures"] == 0 and sat_out == 2 * st["arena_grows"]
    pool.close()
    olib.ago_game_destroy(h)


def _compare_records_with_fresh_oracle_games(olib, pool, rules, openings, batch, ocfg, ev):
    """every game of `openings` played by the pool to its end: moves, root visits and root edge visits of every record against an oracle game
    played from the same opening (the pacing of the pool — yields, deferrals, parked solves — must not show)"""
    games = len(openings)
    recs, edges = pool.records()
    for g in range(games):
        h = olib.ago_game_create(rules, N, N, ctypes.byref(ocfg))
        op = np.array(openings[g] + [0] * (64 - len(openings[g])), np.uint16)
        olib.ago_game_begin(h, ol.ptr(op), len(openings[g]))
        f = np.zeros((batch, HW), np.uint32)
        while olib.ago_game_outcome(h) == 0:
            c = olib.ago_game_step_select(h, ol.ptr(f), batch)
            p, v = ev(f[:c]) if c else (np.zeros((0, HW), np.float32), np.zeros((0, 2), np.float32))
            olib.ago_game_step_expand(h, ol.ptr(np.ascontiguousarray(p)), ol.ptr(np.ascontiguousarray(v)))
        mine = sorted((r.move_number, r) for r in recs if r.game_serial == g)
        assert len(mine) == olib.ago_game_num_records(h), g
        for i, (_, r) in enumerate(mine):
            mv, rv, rs = ctypes.c_uint16(), ctypes.c_int(), ctypes.c_uint16()
            rval = (ctypes.c_float * 2)()
            em, ev_ = np.zeros(512, np.uint16), np.zeros(512, np.int32)
            ep, evl, es = np.zeros(512, np.float32), np.zeros(1024, np.float32), np.zeros(512, np.uint16)
            ne = olib.ago_game_record(h, i, ctypes.byref(mv), ctypes.byref(rv), rval, ctypes.byref(rs), ol.ptr(em), ol.ptr(ev_), ol.ptr(ep), ol.ptr(evl), ol.ptr(es), 512)
            assert (r.move, r.root_visits, r.n_edges) == (mv.value, rv.value, ne), (g, i)
            assert [e.visits for e in edges[r.edge_offset:r.edge_offset + r.n_edges]] == [int(x) for x in ev_[:ne]], (g, i)
        olib.ago_game_destroy(h)


@pytest.mark.parametrize("fraction,speculative,table_bits,rules", [(0.5, 0, 16, 0), (0.5, 1, 16, 0), (0.25, 1, 10, 0), (0.9, 1, 22, 0), (0.5, 1, 16, 2), (0.9, 1, 22, 2),
                                                                   (0.5, 1, 16, 1)])
def test_yielding_pool_gives_the_same_games(agx_lib, olib, fraction, speculative, table_bits, rules):
    """solver_yield_fraction only changes the pacing (stragglers sit out a step): every game must still produce exactly the
    oracle's moves and root visit counts.  Compared through the output records, game by game.  With the speculative solver the rule
    defers a batch whose commit needs a serial re-run while the rest of the launch is done (small tables provoke those), and a solve that is
    still running when all but the last game of the launch are done is PARKED and taken up again by the next launch (engine.hip "Parking":
    renju pools; with 12 games from 11 done on — nearly every launch here)."""
    from alphagomoku_amd import selfplay
    games, batch, sims = 12, 8, 60
    cfg = selfplay.default_config(rules=rules, n_games=games, max_batch_size=batch, max_simulations=sims, tss_table_entries=1 << table_bits, node_capacity=4096,
                                  edge_capacity=65536, solver_yield_fraction=fraction, speculative_solver=speculative, speculative_waves=48)
    pool = selfplay.GeneratorPool(cfg)
    ocfg = ol.default_search_config(max_batch_size=batch, max_simulations=sims, table_entries=1 << table_bits)
    ev = _stand_in_evaluator(olib)
    openings = []
    for g in range(games):
        op = np.zeros(64, np.uint16)
        k = olib.ago_prepare_opening(rules, N, N, 300 + g, ol.ptr(op))
        openings.append([int(x) for x in op[:k]])
    pool.begin(selfplay.pack_openings(openings))
    for _ in range(6000):
        pool.select_solve()
        slots, feats = pool.scheduled()
        if len(slots):
            p, v = ev(feats)
            pool.provide(slots, p, np.concatenate([v, 1 - v.sum(1, keepdims=True)], 1).astype(np.float32))
        pool.expand_backup()
        if pool.stats()["active_games"] == 0:
            break
    st = pool.stats()
    assert st["first_error"] == 0 and st["games_finished"] == games
    if speculative:
        assert st["speculative_solves"] > 0 and (st["speculative_deferrals"] > 0 or table_bits > 16)   # the deferral path ran
        print("solves %d, re-runs %d, deferrals %d, parked %d" % (st["speculative_solves"], st["speculative_reruns"], st["speculative_deferrals"], st["speculative_parks"]))
        assert (st["speculative_parks"] > 0) == (rules == 2)   # ... and so did parking (renju pools only)
    _compare_records_with_fresh_oracle_games(olib, pool, rules, openings, batch, ocfg, ev)
    pool.close()


@pytest.mark.parametrize("how", ["begin", "serial"])
def test_interrupted_speculative_launches_leave_nothing_behind(agx_lib, olib, how):
    """A parked solve leaves its game's OTHER leaves of that launch "solved speculatively, not committed" across launches.  Whoever ends such a
    batch from outside must end that state too: (begin) the pool is restarted with new openings right behind a launch that parked solves;
    (serial) a speculative pool is stepped with the serial launches (Search::solve with a deadline: agx_engine_select_group +
    agx_engine_solve_timed_group at the configured node limit) whenever the previous launch has left parked solves.  Either way the games
    are the oracle's, record by record."""
    from alphagomoku_amd import selfplay
    rules, games, batch, sims, table_bits = 2, 12, 8, 60, 16
    cfg = selfplay.default_config(rules=rules, n_games=games, max_batch_size=batch, max_simulations=sims, tss_table_entries=1 << table_bits, node_capacity=4096,
                                  edge_capacity=65536, solver_yield_fraction=0.5, speculative_solver=1, speculative_waves=48)
    pool = selfplay.GeneratorPool(cfg)
    ocfg = ol.default_search_config(max_batch_size=batch, max_simulations=sims, table_entries=1 << table_bits)
    ev = _stand_in_evaluator(olib)

    def make(seed0):
        out = []
        for g in range(games):
            op = np.zeros(64, np.uint16)
            k = olib.ago_prepare_opening(rules, N, N, seed0 + g, ol.ptr(op))
            out.append([int(x) for x in op[:k]])
        return out

    def finish_step():
        slots, feats = pool.scheduled()
        if len(slots):
            p, v = ev(feats)
            pool.provide(slots, p, np.concatenate([v, 1 - v.sum(1, keepdims=True)], 1).astype(np.float32))
        pool.expand_backup()

    openings = make(300)
    pool.begin(selfplay.pack_openings(openings))
    parks, interrupted, serial_steps = 0, 0, 0
    for _ in range(6000):
        if how == "serial" and pool.stats()["speculative_parks"] > parks:
            # the last launch parked solves: this step is the serial pair of launches
            parks = pool.stats()["speculative_parks"]
            pool.select_group(0, 1)
            pool.solve_timed_group(0, 1, 100, 60.0)
            serial_steps += 1
        else:
            pool.select_solve()
        if how == "begin" and interrupted < 3 and pool.stats()["speculative_parks"] > parks + 2:
            # parked solves and their solved siblings are in flight: start over with other games (three times, then play to the end)
            interrupted += 1
            openings = make(300 + 40 * interrupted)
            pool.begin(selfplay.pack_openings(openings))
            parks = pool.stats()["speculative_parks"]
            continue
        finish_step()
        if pool.stats()["active_games"] == 0:
            break
    st = pool.stats()
    assert st["first_error"] == 0 and st["games_finished"] == games
    assert (interrupted == 3) if how == "begin" else (serial_steps > 5)
    _compare_records_with_fresh_oracle_games(olib, pool, rules, openings, batch, ocfg, ev)
    pool.close()


def test_chip_slices_play_the_same_games(agx_lib):
    """The pool stepped as 4 slices on CU-masked streams (selfplay.chip_slices: how bench.py and ag::GeneratorThread run it) with the HIP
    network in the loop plays, game by game, exactly the moves, root visits and root values of the same pool stepped in one piece."""
    from alphagomoku_amd import selfplay, lib, check
    from alphagomoku_amd.networks import AGNetwork
    games, batch, sims, steps = 64, 8, 60, 260
    desc = synthetic.net_desc(blocks=2, filters=64)
    blob, _ = synthetic.make_weights(desc)
    net = AGNetwork(desc)
    net.loadWeights(blob)
    openings = selfplay.pack_openings(synthetic.make_openings(N, games, seed0=900))

    def play(slices):
        cfg = selfplay.default_config(n_games=games, max_batch_size=batch, max_simulations=sims, tss_table_entries=1 << 16, node_capacity=4096,
                                      edge_capacity=65536, solver_yield_fraction=0.75)
        pool = selfplay.GeneratorPool(cfg)
        pool.begin(openings)
        if slices > 1:
            streams, per = selfplay.chip_slices(slices)
            check(lib.agx_net_set_launch_width(net._net, per))
        else:
            streams = [None]
            check(lib.agx_net_set_launch_width(net._net, 0))
        for _ in range(steps):
            for g in range(slices):
                pool.step_group(net, g, slices, streams[g])
        check(lib.agx_device_synchronize())
        st = pool.stats()
        assert st["first_error"] == 0
        recs, _ = pool.records()
        out = {}
        for r in recs:
            out.setdefault(r.game_serial, []).append((r.move_number, r.move, r.root_visits, r.root_win, r.root_draw, r.root_score))
        pool.close()
        return {k: sorted(v) for k, v in out.items()}, st

    whole, st1 = play(1)
    sliced, st4 = play(4)
    check(lib.agx_net_set_launch_width(net._net, 0))
    net.close()
    assert st1["moves_played"] > 200
    # the slices pace their games differently (yielding is per slice), so after a fixed number of steps a game may be a move ahead in one
    # run: every game's records must agree over the moves both runs have played
    compared = 0
    for serial, a in whole.items():
        b = sliced.get(serial, [])
        n = min(len(a), len(b))
        assert n > 0 and a[:n] == b[:n], serial
        compared += n
    assert compared > 150


def test_games_bit_exact_with_the_hip_network_in_the_loop(agx_lib, olib):
    """C1-shaped plumbing check: 2-block / 64-filter network evaluated by the HIP tower; the oracle tree is fed the same outputs."""
    from alphagomoku_amd.networks import AGNetwork
    d = synthetic.net_desc(blocks=2, filters=64)
    blob, _ = synthetic.make_weights(d)
    net = AGNetwork(d)
    net.loadWeights(blob)

    def evaluator(feats):
        p, v = net.forward(np.ascontiguousarray(feats, dtype=np.uint32))
        return p, np.ascontiguousarray(v[:, :2])
    compared, stats = _play_and_compare(olib, 0, games=4, batch=4, sims=100, max_steps=250, evaluator=evaluator)
    assert compared > 300 and stats["moves_played"] > 0
    net.close()


def test_games_bit_exact_with_the_raw_network_in_the_loop(agx_lib, olib):
    """ResnetPVraw (networks.cpp:107-129) as the evaluator of a pool: the tower reads the 8 low bits of the feature words the
    search kernels write; the pool stepped with the network on the device (GeneratorPool.step) plays the games of the oracle tree fed the
    same network's outputs."""
    from alphagomoku_amd.networks import AGNetwork
    d = synthetic.net_desc(blocks=2, filters=64, in_channels=8)
    blob, _ = synthetic.make_weights(d)
    net = AGNetwork(d)
    net.loadWeights(blob)

    def evaluator(feats):
        p, v = net.forward(np.ascontiguousarray(feats, dtype=np.uint32))
        return p, np.ascontiguousarray(v[:, :2])
    compared, stats = _play_and_compare(olib, 0, games=4, batch=4, sims=100, max_steps=250, evaluator=evaluator, speculative_solver=1, speculative_waves=48)
    assert compared > 300 and stats["moves_played"] > 0
    net.close()


@pytest.mark.parametrize("rules,speculative", [(0, 1), (2, 0)])
def test_restored_games_continue_like_the_oracle(agx_lib, olib, rules, speculative):
    """GeneratorThread::saveGames / loadGames on the C ABI (agx_engine_save_games / agx_engine_restore_game; GameGenerator::save / load,
    GameGenerator.cpp:122-141): a pool plays for a while, its games in flight are saved; a NEW engine gets them back and must go on exactly as
    the oracle does from those positions with empty trees (the reference's load() calls prepare_search on a fresh Tree) — every leaf, every
    root, every move to the end of the games."""
    from alphagomoku_amd import selfplay
    games = 6
    ev = _stand_in_evaluator(olib)
    cfg = selfplay.default_config(rules=rules, n_games=games, max_batch_size=4, max_simulations=60, tss_table_entries=1 << 14, node_capacity=4096, edge_capacity=65536)
    first = selfplay.GeneratorPool(cfg)
    first.begin(selfplay.pack_openings(synthetic.make_openings(15, 3 * games, seed0=40, rules=rules)))   # (a slot whose game ends takes the next opening)
    for _ in range(140):
        first.select_solve()
        slots, feats = first.scheduled()
        if len(slots):
            pol, val = ev(feats)
            first.provide(slots, pol, np.concatenate([val, 1 - val.sum(1, keepdims=True)], 1).astype(np.float32))
        first.expand_backup()
    saved = first.save_games()
    infos = [first.game_info(g, with_edges=False) for g in range(games)]
    first.close()
    assert len(saved) >= games - 2 and all(len(g["moves"]) == infos[g["game_slot"]]["n_moves"] for g in saved)
    assert max(len(g["moves"]) for g in saved) > 8          # the games have left their openings behind
    by_slot = {g["game_slot"]: g for g in saved}
    restore = [by_slot.get(g, saved[0]) for g in range(games)]   # (a slot whose game had just ended takes a copy of another game)
    compared, stats = _play_and_compare(olib, rules, games=games, batch=4, sims=60, max_steps=2500, evaluator=ev, table_entries=1 << 14,
                                        restore_from=restore, speculative_solver=speculative, speculative_waves=48)
    assert compared > 200 and stats["first_error"] == 0 and stats["games_finished"] >= games


def test_time_limited_solve(agx_lib, olib):
    """Search::solve(endTime >= 0) (Search.cpp:159-183, SearchThread::asynchronous_run): node limit + a wall-clock budget shared out over the
    leaves.  (a) With a generous budget and the configured node limit the launch is Search::solve(): the pool plays the oracle's games.  (b) With
    no time left every leaf stops after its first node (AlphaBetaSearch.cpp:110-111: the check behind the first iteration).  (c) A 10 000-node
    limit visits more nodes than the 100-node one on the same leaves."""
    from alphagomoku_amd import selfplay
    games, batch = 8, 4
    ev = _stand_in_evaluator(olib)
    openings = synthetic.make_openings(15, games, seed0=60)

    def run(mode, steps):
        pool = selfplay.GeneratorPool(selfplay.default_config(n_games=games, max_batch_size=batch, max_simulations=80, tss_table_entries=1 << 14, node_capacity=4096,
                                                              edge_capacity=65536, speculative_solver=0, solver_yield_fraction=0.0))
        pool.begin(selfplay.pack_openings(openings))
        roots = []
        for _ in range(steps):
            if mode == "plain":
                pool.select_solve()
            else:
                pool.select_group(0, 1)
                pool.solve_timed_group(0, 1, {"generous": 100, "expired": 100, "deep": 10000}[mode], 0.0 if mode == "expired" else 30.0)
            slots, feats = pool.scheduled()
            if len(slots):
                pol, val = ev(feats)
                pool.provide(slots, pol, np.concatenate([val, 1 - val.sum(1, keepdims=True)], 1).astype(np.float32))
            pool.expand_backup()
            roots.append([(pool.game_info(g)["root_visits"], [(e["move"], e["visits"], e["score"]) for e in pool.game_info(g)["edges"]]) for g in range(games)])
        st = pool.stats()
        pool.close()
        return roots, st
    plain, st_plain = run("plain", 25)
    generous, st_generous = run("generous", 25)
    assert plain == generous and st_plain["solver_nodes"] == st_generous["solver_nodes"]
    _, st_expired = run("expired", 6)
    _, st_six = run("plain", 6)
    assert 0 < st_expired["solver_nodes"] < st_six["solver_nodes"]
    assert st_expired["solver_nodes"] <= 6 * games * batch      # one node per leaf: at most steps x games x batch
    _, st_deep = run("deep", 6)
    assert st_deep["solver_nodes"] > 2 * st_six["solver_nodes"] and st_deep["first_error"] == 0


def test_select_stage_follows_set_batch_size(agx_lib, olib):
    """Search::setBatchSize (Search.cpp:252-255; SearchThread.cpp:125-126): a pool created with max_batch_size 8 whose select stage is told to
    take 3 leaves per game plays exactly the games of a pool created with max_batch_size 3"""
    from alphagomoku_amd import selfplay
    games = 6
    ev = _stand_in_evaluator(olib)
    openings = synthetic.make_openings(15, games, seed0=70)

    def run(max_batch, limit):
        pool = selfplay.GeneratorPool(selfplay.default_config(n_games=games, max_batch_size=max_batch, max_simulations=60, tss_table_entries=1 << 14, node_capacity=4096,
                                                              edge_capacity=65536))
        pool.begin(selfplay.pack_openings(openings))
        if limit:
            pool.set_batch_size(limit)
        trace = []
        for _ in range(60):
            pool.select_solve()
            slots, feats = pool.scheduled()
            if len(slots):
                pol, val = ev(feats)
                pool.provide(slots, pol, np.concatenate([val, 1 - val.sum(1, keepdims=True)], 1).astype(np.float32))
            pool.expand_backup()
            trace.append([(i["root_visits"], i["n_moves"], [(e["move"], e["visits"]) for e in i["edges"]]) for i in (pool.game_info(g) for g in range(games))])
        pool.close()
        return trace
    assert run(8, 3) == run(3, 0)
    assert run(8, 3) != run(8, 0)


@pytest.mark.parametrize("rules,symmetries", [(0, 0), (1, 1), (2, 0)])
def test_whole_games_with_action_values(agx_lib, olib, rules, symmetries):
    """'pvq' network semantics (ResnetPVQ): every edge of a network-evaluated node starts from the 'q' output of its cell, which
    is what the default init_to = "q_head" selector reads (EdgeSelector.cpp:335-361, EdgeGenerator.cpp:119-124)"""
    base = _stand_in_evaluator(olib)

    def evaluator(feats):
        pol, val = base(feats)
        h = (np.ascontiguousarray(feats, dtype=np.uint32).astype(np.uint64) * np.uint64(2654435761) + np.arange(HW, dtype=np.uint64) * np.uint64(40503)) % np.uint64(1 << 20)
        w = (h.astype(np.float32) / np.float32(1 << 20)) * np.float32(0.8)
        d = (np.float32(1.0) - w) * np.float32(0.25)
        return pol, val, np.stack([w, d], axis=2).astype(np.float32)
    compared, stats = _play_and_compare(olib, rules, games=4, batch=4, sims=80, max_steps=4000, evaluator=evaluator, use_symmetries=symmetries,
                                        action_values=1)
    assert compared > 200 and stats["games_finished"] == 4


def test_pvq_network_in_the_pool(agx_lib):
    """agx_engine_evaluate with a ResnetPVQ network == network outputs (policy, value, q) written by hand into the slot buffers"""
    from alphagomoku_amd import selfplay, AgxError
    from alphagomoku_amd.networks import AGNetwork
    d = synthetic.net_desc(blocks=2, filters=64, action_values=1)
    blob, _ = synthetic.make_weights(d)
    net = AGNetwork(d)
    net.loadWeights(blob)
    openings = synthetic.make_openings(15, 8, seed0=3)
    pools = []
    for _ in range(2):
        pool = selfplay.GeneratorPool(selfplay.default_config(n_games=8, max_batch_size=4, max_simulations=60, tss_table_entries=1 << 14,
                                                              node_capacity=2048, edge_capacity=32768, action_values=1))
        pool.begin(selfplay.pack_openings(openings))
        pools.append(pool)
    for step in range(40):
        pools[0].step(net)
        pools[1].select_solve()
        slots, feats = pools[1].scheduled()
        if len(slots):
            p, v, q = net.forward(feats)
            pools[1].provide(slots, p, v, q)
        pools[1].expand_backup()
        for g in range(8):
            a, b = pools[0].game_info(g), pools[1].game_info(g)
            assert a["root_visits"] == b["root_visits"] and a["n_moves"] == b["n_moves"], (step, g)
            assert [(e["move"], e["visits"], e["win"], e["draw"], e["prior"]) for e in a["edges"]] == \
                   [(e["move"], e["visits"], e["win"], e["draw"], e["prior"]) for e in b["edges"]], (step, g)
    assert any(e["win"] != 0.0 for e in pools[0].game_info(0)["edges"])   # the q outputs really seed the edges
    pv = AGNetwork(synthetic.net_desc(blocks=2, filters=64))
    pv.loadWeights(blob[:pv.blobFloats()])
    with pytest.raises(AgxError):
        pools[0].step(pv)          # a 'pv' network cannot feed a pool configured for action values
    for pool in pools:
        pool.close()
    net.close()
    pv.close()


@pytest.mark.parametrize("rules", [0, 2])
def test_opening_generator(agx_lib, olib, rules):
    """OpeningGenerator::generate on the device: every opening is a legal undecided position that the solver (1000 nodes) does not
    prove and that the network rates as balanced: |E - 0.5| < 0.1 + 0.01 * trials (trials <= number of rejections)"""
    from alphagomoku_amd import selfplay
    from alphagomoku_amd.networks import AGNetwork
    from oracle import nn_ref
    d = synthetic.net_desc(blocks=2, filters=64)
    blob, _ = synthetic.make_weights(d)
    net = AGNetwork(d)
    net.loadWeights(blob)
    pool = selfplay.GeneratorPool(selfplay.default_config(rules=rules, n_games=64, max_batch_size=2, tss_table_entries=1 << 16, node_capacity=256,
                                                          edge_capacity=4096))
    openings, st = pool.generate_openings(net, 40, seed=7)
    assert len(openings) == 40 and st["candidates"] >= 40 and st["network_evaluations"] >= 40
    assert st["candidates"] == st["proven_by_solver"] + st["network_evaluations"]
    feats = []
    for op in openings:
        b = np.zeros((N, N), np.uint8)
        for k, m in enumerate(op):
            assert (m & 3) == 1 + (k & 1) and b[(m >> 2) & 127, (m >> 9) & 127] == 0
            b[(m >> 2) & 127, (m >> 9) & 127] = m & 3
        if op:
            last = op[-1]
            assert olib.ago_outcome(rules, N, N, ol.ptr(b), last & 3, (last >> 2) & 127, (last >> 9) & 127, -1) == 0
        sign = 1 if len(op) % 2 == 0 else 2
        feats.append(ol.encode_features(olib, rules, b.tolist(), sign).reshape(-1))
    _, v = nn_ref.forward(d, blob, np.array(feats, np.uint32))
    balance = np.abs(v[:, 0] + 0.5 * v[:, 1] - 0.5)
    assert (balance < 0.1 + 0.01 * st["unbalanced"] + 4e-3).all()
    # the pool is usable afterwards, and the generator refuses to run under a playing pool
    pool.begin(selfplay.pack_openings(openings))
    for _ in range(3):
        pool.step(net)
    assert pool.stats()["first_error"] == 0
    from alphagomoku_amd import AgxError
    with pytest.raises(AgxError):
        pool.generate_openings(net, 1)
    pool.close()
    net.close()


def test_pool_step_with_device_network_is_deterministic_and_consistent(agx_lib):
    """agx_engine_step (network evaluated on the device through the indirect slot list) twice from the same openings gives identical
    records; counters are consistent (size-independent properties that also hold at BASELINE sizes)."""
    from alphagomoku_amd import selfplay
    from alphagomoku_amd.networks import AGNetwork
    d = synthetic.net_desc(blocks=2, filters=64)
    blob, _ = synthetic.make_weights(d)
    net = AGNetwork(d)
    net.loadWeights(blob)
    openings = selfplay.pack_openings(synthetic.make_openings(N, 64, seed0=5))

    def run():
        pool = selfplay.GeneratorPool(selfplay.default_config(n_games=32, max_batch_size=8, max_simulations=60, tss_table_entries=1 << 14,
                                                              node_capacity=2048, edge_capacity=32768))
        pool.begin(openings)
        for _ in range(120):
            pool.step(net)
        st = pool.stats()
        recs, edges = pool.records()
        out = sorted((r.game_serial, r.move_number, r.move, r.root_visits, r.n_edges,
                      tuple((e.move, e.visits) for e in edges[r.edge_offset:r.edge_offset + r.n_edges])) for r in recs)
        pool.close()
        return st, out
    s1, r1 = run()
    s2, r2 = run()
    assert r1 == r2 and s1 == s2
    assert s1["first_error"] == 0 and s1["moves_played"] == len(r1) and s1["moves_played"] > 32
    assert s1["network_evaluations"] <= s1["evaluated_nodes"] + 32
    for serial, number, move, visits, n_edges, edges in r1:
        assert sum(v for _, v in edges) <= visits        # edge visits never exceed the root's
        assert any(m == move for m, _ in edges)         # the played move is one of the root edges
    net.close()


FULL_SIZE = {
    # BASELINE.json configs at bench.py's sizes: rules, board, network, playouts, steps of the big pool, steps of the small pool
    # (+ the least number of move records that must have been compared over the 128 games looked at: 2 to 3 moves per game)
    "C2-freestyle-15x15-6x128-400": dict(rules=0, n=15, blocks=6, sims=400, big_steps=320, small_steps=510, compared=300),
    "C3-standard-15x15-10x128-800": dict(rules=1, n=15, blocks=10, sims=800, big_steps=350, small_steps=560, compared=200),
    "C4-caro5-20x20-10x128-400": dict(rules=3, n=20, blocks=10, sims=400, big_steps=240, small_steps=390, compared=270),
    "C5-renju-15x15-10x128-1600": dict(rules=2, n=15, blocks=10, sims=1600, big_steps=500, small_steps=810, compared=170),
}


@pytest.mark.parametrize("name", list(FULL_SIZE))
def test_full_size_pool_matches_a_small_pool_game_by_game(agx_lib, name):
    """BASELINE configs[1..4] at FULL size — 1024 games, the config's network in the loop, its playout budget, batch 8, yielding on, the
    reference's 4 Mi-entry solver table per game and bench.py's arena sizes: games are independent, so every game of the big pool must play
    exactly what the same opening plays in a 128-game pool stepped in one piece, serial solver, no yielding (whose behaviour the other tests pin
    to the oracle step by step) — a size-independent property checked at full size, on the path bench.py times; the arenas must hold."""
    from alphagomoku_amd import selfplay, lib, check
    from alphagomoku_amd.networks import AGNetwork
    c = FULL_SIZE[name]
    n, sims = c["n"], c["sims"]
    d = synthetic.net_desc(rows=n, cols=n, blocks=c["blocks"], filters=128)
    blob, _ = synthetic.make_weights(d)
    net = AGNetwork(d)
    net.loadWeights(blob)
    openings = synthetic.make_openings(n, 1024, seed0=900, rules=c["rules"])

    def run(games, steps, as_bench):
        # as_bench: exactly bench.py's way of running the pool — 4 chip slices on CU-masked streams, the speculative solver, yielding at 0.5 (renju: parking),
        # format-201 samples on (record_format 3 = samples + the raw root edges this test compares), records drained as it goes
        pool = selfplay.GeneratorPool(selfplay.default_config(rules=c["rules"], board_size=n, draw_after=n * n, n_games=games, max_batch_size=8, max_simulations=sims,
                                                              tss_table_entries=4 * 1024 * 1024, node_capacity=max(4096, 8 * sims),
                                                              edge_capacity=max(65536, 192 * sims), arena_reserve=3.0, solver_yield_fraction=0.5 if as_bench else 0.0,
                                                              speculative_solver=1 if as_bench else 0, record_format=3 if as_bench else 1,
                                                              record_capacity=games * 64, record_edge_capacity=games * 64 * n * n))
        pool.begin(selfplay.pack_openings(openings[:games]))   # no spare openings: a finished game stays finished
        slices = 4 if as_bench else 1
        if as_bench:
            streams, per = selfplay.chip_slices(slices)
            check(lib.agx_net_set_launch_width(net._net, per))
        else:
            streams = [None]
            check(lib.agx_net_set_launch_width(net._net, 0))
        per_game = {}

        def collect(drain):
            recs, edges = pool.records(drain=drain)
            for r in recs:
                per_game.setdefault(r.game_serial, []).append((r.move_number, r.move, r.root_visits, r.n_edges,
                                                               tuple((e.move, e.visits, e.score) for e in edges[r.edge_offset:r.edge_offset + r.n_edges])))
        for i in range(steps):
            for g in range(slices):
                pool.step_group(net, g, slices, streams[g])
            if as_bench and i % 64 == 63:
                collect(True)
        check(lib.agx_device_synchronize())
        st = pool.stats()
        collect(False)
        pool.close()
        return st, {g: sorted(v) for g, v in per_game.items()}
    big_stats, big = run(1024, c["big_steps"], True)
    small_stats, small = run(128, c["small_steps"], False)
    check(lib.agx_net_set_launch_width(net._net, 0))
    assert big_stats["first_error"] == 0 and small_stats["first_error"] == 0 and big_stats["arena_failures"] == 0
    assert big_stats["moves_played"] > 1024 and big_stats["evaluated_nodes"] > 1024 * sims and big_stats["speculative_solves"] > 0
    assert (big_stats["speculative_parks"] > 0) == (c["rules"] == 2)   # renju launches park the solves that outlast them (a dozen per slice launch), the others never
    compared = 0
    for g in range(128):
        a, b = big.get(g, []), small.get(g, [])
        k = min(len(a), len(b))
        assert k >= 1, g
        assert a[:k] == b[:k], g           # same moves, same root visit counts, same edge visits and scores
        compared += k
    assert compared >= c["compared"], compared
    net.close()


def test_whole_games_with_the_reference_table_size(agx_lib, olib):
    """the solver's transposition table at the reference's size (4 Mi entries = 64 MB per game, AlphaBetaSearch.cpp:59): bucket choice and
    replacement depend on the table size, so the BASELINE size gets its own step-by-step comparison with the oracle"""
    compared, stats = _play_and_compare(olib, 0, games=3, batch=8, sims=100, max_steps=4000, evaluator=_stand_in_evaluator(olib), table_entries=4 * 1024 * 1024)
    assert compared > 300 and stats["games_finished"] == 3
    compared, stats = _play_and_compare(olib, 2, games=2, batch=4, sims=100, max_steps=4000, evaluator=_stand_in_evaluator(olib), table_entries=4 * 1024 * 1024)
    assert compared > 150 and stats["games_finished"] == 2


def test_engine_error_paths(agx_lib):
    from alphagomoku_amd import selfplay, AgxError
    with pytest.raises(AgxError):
        selfplay.GeneratorPool(selfplay.default_config(rules=7))                      # unknown rules
    with pytest.raises(AgxError):
        selfplay.GeneratorPool(selfplay.default_config(board_size=25))
    pool = selfplay.GeneratorPool(selfplay.default_config(n_games=2, tss_table_entries=1 << 10, node_capacity=64, edge_capacity=1024))
    with pytest.raises(AgxError):
        pool.select_solve()                                                           # begin() not called
    # capacity overflow must be reported, not silently corrupt the tree
    pool.begin(selfplay.pack_openings([[], []]))
    olib = ol.load()
    ev = _stand_in_evaluator(olib)
    for _ in range(200):
        pool.select_solve()
        slots, feats = pool.scheduled()
        if len(slots):
            p, v = ev(feats)
            pool.provide(slots, p, np.concatenate([v, 1 - v.sum(1, keepdims=True)], 1).astype(np.float32))
        pool.expand_backup()
    assert pool.stats()["first_error"] in (1, 2, 5)
    pool.close()


def test_native_cpp_driver_over_the_c_abi(agx_lib):
    """alphagomoku_amd/agx_selfplay: the C++ host loop (include/agx.hpp facade: AGNetwork + GeneratorPool, the stand-ins of the
    reference's AGNetwork / GeneratorThread) drives the same library without Python"""
    import json
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "alphagomoku_amd", "agx_selfplay")
    assert os.path.exists(exe), "native driver not built (python -m alphagomoku_amd.build)"
    out = subprocess.run([exe, "--games", "16", "--steps", "30", "--warmup", "3", "--sims", "50", "--batch", "4", "--blocks", "2", "--filters", "64",
                          "--rules", "1"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["simulations_per_sec"] > 0 and line["network_evaluations"] > 0
    # the long-running shape: balanced openings from the device generator, pvq network, symmetries, draining and refilling
    out = subprocess.run([exe, "--games", "8", "--steps", "400", "--warmup", "2", "--sims", "20", "--batch", "8", "--blocks", "2", "--filters", "64",
                          "--balanced-openings", "1", "--drain-every", "20", "--pvq", "1", "--symmetries", "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["samples_drained"] > 50 and line["games_finished"] > 4 and line["opening_refills"] >= 1
    # evaluation matches from the same loop: 8 pairs of players, two networks
    out = subprocess.run([exe, "--match", "1", "--games", "8", "--steps", "2500", "--warmup", "2", "--sims", "50", "--batch", "8", "--blocks", "2",
                          "--filters", "64"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["games_finished"] > 0 and sum(line["first_player_won_drawn_lost"]) >= line["games_finished"]   # (the score also counts warm-up games)
    bad = subprocess.run([exe, "--filters", "96", "--steps", "1"], capture_output=True, text=True, timeout=300)
    assert bad.returncode == 1 and "unsupported network" in bad.stderr   # errors surface as exceptions with the library's message


def test_long_running_loop_drains_records_and_refills_openings(agx_lib):
    """A generator thread that runs for hours: records are handed over and the device pools emptied (agx_engine_drain_records),
    openings are appended on demand (agx_engine_add_openings); finished games take openings in GAME order, so two runs agree."""
    from alphagomoku_amd import selfplay
    from alphagomoku_amd.networks import AGNetwork
    d = synthetic.net_desc(blocks=2, filters=64)
    blob, _ = synthetic.make_weights(d)
    net = AGNetwork(d)
    net.loadWeights(blob)
    first = synthetic.make_openings(15, 12, seed0=50)
    more = synthetic.make_openings(15, 30, seed0=500)

    def run():
        pool = selfplay.GeneratorPool(selfplay.default_config(n_games=8, max_batch_size=8, max_simulations=30, tss_table_entries=1 << 14,
                                                              node_capacity=2048, edge_capacity=32768, record_capacity=4096))
        pool.begin(selfplay.pack_openings(first))
        drained, history, added = [], [], False
        for step in range(1500):
            pool.step(net)
            if step % 50 == 49:
                recs, edges = pool.records(drain=True)
                for r in recs:
                    e = edges[r.edge_offset:r.edge_offset + r.n_edges]
                    drained.append((r.game_serial, r.move_number, r.move, r.root_visits, tuple(x.visits for x in e)))
                st = pool.stats()
                assert st["records_used"] == 0 and st["first_error"] == 0
                history.append((st["games_finished"], st["openings_taken"], tuple(pool.game_info(g)["opening_id"] for g in range(8))))
                if st["games_finished"] >= 6 and not added:
                    # slot s plays openings s, s + 8, s + 16, ...: with 12 openings in the list the slots 4-7 wait after their first game
                    # (a waiting slot is one whose next opening, slot + 8 x games played, is not in the list of 12 yet)
                    waiting = [g for g in range(8) if not pool.game_info(g)["active"]]
                    assert all(g + 8 * pool.game_info(g)["games_done"] >= 12 for g in waiting)
                    pool.add_openings(selfplay.pack_openings(more))
                    added = True
                if st["games_finished"] >= 30:
                    break
        st = pool.stats()
        pool.close()
        return drained, history, st
    d1, h1, st1 = run()
    d2, h2, st2 = run()
    assert st1["games_finished"] >= 30 and st1["openings_taken"] > 12
    assert len(d1) == st1["moves_played"]                                   # every played move was handed over exactly once
    assert len({(s, m) for s, m, _, _, _ in d1}) == len(d1)
    assert sorted(d1) == sorted(d2) and h1 == h2                            # reproducible, including which slot got which opening
    net.close()


def test_wave_reduction_helpers(agx_lib, tmp_path):
    """scripts/dpp_check.hip: the DPP reductions / scan / arg-max that replaced the ds_bpermute shuffles, against a scalar loop"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "dpp_check")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(root, "include"),
                           "-o", exe, os.path.join(root, "scripts", "dpp_check.hip")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    assert out.stdout.strip().splitlines()[-1] == "bad: umax 0 add 0 xor 0 scan 0 argmax value 0 index 0", out.stdout


# ---------------------------------------------------------------------------------------------------------------------------------
# evaluation matches: two players, two networks, one game per pair of trees


def _best_edge(root_visits, edges):
    """BestEdgeSelector (EdgeSelector.cpp:515-536) on a root snapshot, in float32 like the reference: first maximum wins"""
    best, best_value = -1, np.float32(-3.0e38)
    for i, e in enumerate(edges):
        pv, raw_eval = (e["score"] >> 13) & 3, (e["score"] & 8191) - 4000
        if pv == 0:
            value = np.float32(-1.0e8) + np.float32(raw_eval)
        elif pv == 3:
            value = np.float32(1.0e8) - np.float32(-raw_eval)
        else:
            expectation = np.float32(e["win"]) + np.float32(0.5) * np.float32(e["draw"])
            value = np.float32(e["visits"]) + expectation * np.float32(root_visits) + np.float32(0.001) * np.float32(e["prior"])
        if value > best_value:
            best, best_value = i, value
    return best


@pytest.mark.parametrize("rules,sims,speculative", [(0, 100, 0), (1, 80, 1), (2, 80, 0)])
def test_player_api_drives_a_game_from_outside(agx_lib, olib, rules, sims, speculative):
    """Two evaluation Players, each a Tree / Search pair of its own (a one-game engine), driven exactly like evaluation/Player.cpp:100-129 and
    EvaluationGame.cpp:100-143 drive them: setBoard (cleanup + Tree::setBoard + Search::setBoard) for the player to move, then select / solve /
    evaluate / expand / backup until isSearchOver, the move by the final selector on a copy of the root.  Every leaf, every root and every move
    must equal the oracle's two players (Game::match_begin / take_turn / external_move), for a whole game."""
    from alphagomoku_amd import selfplay
    batch = 8
    evaluators = [_stand_in_evaluator(olib), _second_evaluator(olib)]
    ocfg = ol.default_search_config(max_batch_size=batch, max_simulations=sims, table_entries=1 << 16)
    op = np.zeros(64, np.uint16)
    k = olib.ago_prepare_opening(rules, N, N, 4242, ol.ptr(op))
    opening = [int(x) for x in op[:k]]
    pools, handles = [], []
    for _ in range(2):
        cfg = selfplay.default_config(rules=rules, n_games=1, max_batch_size=batch, max_simulations=1 << 24, tss_table_entries=1 << 16, node_capacity=4096,
                                      edge_capacity=65536, force_expand_root=0, speculative_solver=speculative, speculative_waves=16)
        pool = selfplay.GeneratorPool(cfg)
        pool.begin(selfplay.pack_openings([[]]))     # an engine has to be begun; the first set_board replaces the empty position
        pool.set_max_simulations(sims)               # Search::select(tree, constraints.max_simulations)
        pools.append(pool)
        h = olib.ago_game_create_ex(rules, N, N, 0, ctypes.byref(ocfg))
        olib.ago_game_set_force_expand_root(h, 0)
        olib.ago_game_match_begin(h, ol.ptr(np.array(opening + [0], np.uint16)), len(opening))
        handles.append(h)
    board = np.zeros(HW, np.uint8)
    for m in opening:
        board[(m >> 2 & 127) * N + (m >> 9 & 127)] = m & 3
    sign = 1 if not opening else 3 - (opening[-1] & 3)
    who = 0 if sign == 1 else 1     # the first player holds cross
    plies = compared = 0
    while True:
        pool, h = pools[who], handles[who]
        pool.set_board(0, board, sign)               # Player::setBoard
        olib.ago_game_take_turn(h)
        while True:                                  # EvaluationGame: selectSolveEvaluate, expandBackup, isSearchOver
            pool.select_solve()
            slots, feats = pool.scheduled()
            f = np.zeros((batch, HW), np.uint32)
            c = olib.ago_game_step_select(h, ol.ptr(f), batch)
            assert c == len(slots) and np.array_equal(feats, f[:c]), plies
            pol, val = evaluators[who](feats) if c else (np.zeros((0, HW), np.float32), np.zeros((0, 2), np.float32))
            pool.provide(slots, pol, np.concatenate([val, 1 - val.sum(1, keepdims=True)], 1).astype(np.float32))
            pool.expand_only()
            moved = olib.ago_game_step_expand(h, ol.ptr(np.ascontiguousarray(pol)), ol.ptr(np.ascontiguousarray(val)))
            info = pool.game_info(0)
            assert info["error"] == 0
            proven = ((info["root_score"] >> 13) & 3) != 2 and info["root_score"] not in (0, 0xFFFF)
            reduction = np.float32(max(0.0, min(1.0, (np.float32(info["root_draw"]) - np.float32(0.75)) / np.float32(0.25))))
            budget = int(np.float32(sims) - reduction * np.float32(sims - 50))      # get_simulations_for_move (utils/misc.cpp:171-179)
            over = proven or info["root_visits"] > budget                           # Player::isSearchOver (Player.cpp:152-160)
            assert over == bool(moved), plies
            if not over:
                r = _oracle_root(olib, h)
                assert r["n"] == info["root_edges"] and r["visits"] == info["root_visits"], plies
                assert np.array_equal(np.array([x["visits"] for x in info["edges"]], np.int32), r["ev"]), plies
                compared += 1
                continue
            mv = info["edges"][_best_edge(info["root_visits"], info["edges"])]["move"]     # Player::getMove
            assert mv == olib.ago_game_last_move(h), plies
            break
        board[(mv >> 2 & 127) * N + (mv >> 9 & 127)] = mv & 3
        sign = 3 - (mv & 3)
        olib.ago_game_external_move(handles[1 - who], mv)
        plies += 1
        if olib.ago_game_outcome(h) != 0:
            assert olib.ago_game_outcome(handles[1 - who]) == olib.ago_game_outcome(h)
            break
        who = 1 - who
    assert plies >= 10 and compared > 60
    for pool in pools:
        pool.close()
    for h in handles:
        olib.ago_game_destroy(h)


def _second_evaluator(olib, hw=HW):
    """a different deterministic 'network' for the second player: sharper policy, shifted value"""
    base = _stand_in_evaluator(olib, hw)

    def f(feats):
        pol, val = base(feats)
        pol = (pol * pol).astype(np.float32)
        s = pol.sum(1, keepdims=True, dtype=np.float32)
        pol = np.where(s > 0, pol / np.where(s > 0, s, 1), np.float32(1.0 / hw)).astype(np.float32)
        val = np.stack([np.float32(0.9) * val[:, 0] + np.float32(0.05), np.float32(0.5) * val[:, 1]], 1).astype(np.float32)
        return pol, val
    return f


def _play_matches_and_compare(olib, rules, pairs, n_openings, batch, sims, max_steps, max_children=0, n=N, draw_after=0, merged=False, use_symmetries=0):
    from alphagomoku_amd import selfplay
    N, HW = n, n * n   # noqa: N806
    cfg = selfplay.default_config(rules=rules, board_size=n, draw_after=draw_after if draw_after > 0 else n * n, n_games=2 * pairs, max_batch_size=batch,
                                  max_simulations=sims, tss_table_entries=1 << 16, node_capacity=4096, edge_capacity=65536, match_mode=1,
                                  max_children=max_children, use_symmetries=use_symmetries)
    pool = selfplay.GeneratorPool(cfg)
    ocfg = ol.default_search_config(max_batch_size=batch, max_simulations=sims, table_entries=1 << 16, use_symmetries=use_symmetries)
    if max_children > 0:
        ocfg.max_children = max_children
    evaluators = [_stand_in_evaluator(olib, HW), _second_evaluator(olib, HW)]
    openings = []
    for g in range(n_openings):
        op = np.zeros(64, np.uint16)
        k = olib.ago_prepare_opening(rules, N, N, 500 + g, ol.ptr(op))
        openings.append([int(x) for x in op[:k]])
    players = []   # [pair][0 first / 1 second player]
    for m in range(pairs):
        hs = []
        for _ in range(2):
            h = olib.ago_game_create_ex(rules, N, N, draw_after, ctypes.byref(ocfg))
            olib.ago_game_set_force_expand_root(h, 0)   # Player::setBoard builds UnifiedGenerator without forceExpandRoot
            hs.append(h)
        players.append(hs)
    games_done = [0] * pairs
    mover = [None] * pairs       # 0 / 1: whose turn; None: the pair waits for an opening
    opening_of = [None] * pairs
    results = []

    def start_game(m, oid):
        opening_of[m] = oid
        op = np.array(openings[oid] + [0], np.uint16)
        first_sign = 1 if games_done[m] % 2 == 0 else 2      # EvaluationGame.cpp:59-71
        for h in players[m]:
            olib.ago_game_set_serial(h, oid)
            olib.ago_game_match_begin(h, ol.ptr(op), len(openings[oid]))
        mover[m] = 0 if olib.ago_game_sign_to_move(players[m][0]) == first_sign else 1
        olib.ago_game_take_turn(players[m][mover[m]])

    def restart_waiting():   # what k_assign_openings + k_match_restart do at the end of the first players' phase
        for m in range(pairs):
            if mover[m] is None:
                if games_done[m] % 2 == 1:
                    start_game(m, opening_of[m])
                elif m + pairs * (games_done[m] // 2) < n_openings:
                    # pair m plays openings m, m + pairs, m + 2 pairs, ...: a function of the pair and of how many matches it has played, not of
                    # which pair finished first (k_assign_openings)
                    start_game(m, m + pairs * (games_done[m] // 2))

    pool.begin(selfplay.pack_openings(openings))
    restart_waiting()
    compared = 0
    # merged: every stage one launch over both players' trees (agx_engine_step_match); else two group steps one after the other
    phases = [(0, 1)] if merged else [(0,), (1,)]
    for step in range(max_steps):
        for phase in phases:
            if merged:
                pool.select_solve_match()
            else:
                pool.select_solve_group(phase[0], 2)
            by_slot, pol_of, val_of, feat_of = {}, {}, {}, {}
            all_slots, all_pol, all_val = [], [], []
            for player in phase:
                slots, feats = pool.scheduled_group(player, 2)
                pol, val = evaluators[player](feats) if len(slots) else (np.zeros((0, HW), np.float32), np.zeros((0, 2), np.float32))
                for i, sl in enumerate(slots):
                    assert (int(sl) // batch >= pairs) == (player == 1)      # each player's list holds only its own trees' leaves
                    by_slot[int(sl)] = (feats[i], pol[i], val[i])
                all_slots.append(slots)
                all_pol.append(pol)
                all_val.append(val)
            slots = np.concatenate(all_slots)
            pol = np.concatenate(all_pol)
            val = np.concatenate(all_val)
            v3 = np.concatenate([val, 1 - val.sum(1, keepdims=True)], 1).astype(np.float32)
            pool.provide(slots, pol, v3)
            seen = 0
            movers_now = list(mover)
            for m in range(pairs):
                who = movers_now[m]
                if who is None or who not in phase:
                    continue
                tree = m + who * pairs
                h = players[m][who]
                f = np.zeros((batch, HW), np.uint32)
                c = olib.ago_game_step_select(h, ol.ptr(f), batch)
                mine = sorted(sl for sl in by_slot if sl // batch == tree)
                assert c == len(mine), (step, phase, m)
                assert np.array_equal(np.array([by_slot[sl][0] for sl in mine], np.uint32).reshape(c, HW), f[:c]), (step, phase, m)
                seen += c
                p_in = np.ascontiguousarray(np.array([by_slot[sl][1] for sl in mine], np.float32).reshape(c, HW))
                v_in = np.ascontiguousarray(np.array([by_slot[sl][2] for sl in mine], np.float32).reshape(c, 2))
                moved = olib.ago_game_step_expand(h, ol.ptr(p_in), ol.ptr(v_in))
                if moved:
                    other = players[m][1 - who]
                    olib.ago_game_external_move(other, olib.ago_game_last_move(h))
                    assert olib.ago_game_outcome(other) == olib.ago_game_outcome(h)
                    if olib.ago_game_outcome(h) != 0:
                        results.append((m, opening_of[m], games_done[m] % 2, olib.ago_game_outcome(h)))
                        games_done[m] += 1
                        mover[m] = None
                    else:
                        olib.ago_game_take_turn(other)
                        mover[m] = 1 - who
            assert seen == len(slots), (step, phase)      # idle trees schedule nothing
            if merged:
                pool.expand_backup_match()
            else:
                pool.expand_backup_group(phase[0], 2)
            if merged or phase[0] == 0:
                restart_waiting()
            for m in range(pairs):
                infos = [pool.game_info(m), pool.game_info(m + pairs)]
                assert infos[0]["error"] == 0 and infos[1]["error"] == 0
                if mover[m] is None:
                    assert not infos[0]["active"] and not infos[1]["active"], (step, phase, m)
                    continue
                info = infos[mover[m]]
                assert info["active"] and not infos[1 - mover[m]]["active"], (step, phase, m)   # exactly the player to move searches
                assert info["opening_id"] == opening_of[m] and info["games_done"] == games_done[m], (step, phase, m)
                r = _oracle_root(olib, players[m][mover[m]])
                e = info["edges"]
                assert r["n"] == info["root_edges"] and r["visits"] == info["root_visits"], (step, phase, m)
                assert np.array_equal(np.array([x["move"] for x in e], np.uint16), r["moves"]), (step, phase, m)
                assert np.array_equal(np.array([x["visits"] for x in e], np.int32), r["ev"]), (step, phase, m)
                assert np.array_equal(np.array([x["score"] for x in e], np.uint16), r["es"]), (step, phase, m)
                assert np.array_equal(np.array([x["prior"] for x in e], np.float32), r["prior"]), (step, phase, m)
                assert np.array_equal(np.array([[x["win"], x["draw"]] for x in e], np.float32).reshape(-1), r["val"]), (step, phase, m)
                compared += 1
        if all(x is None for x in mover) and all(d % 2 == 0 and m + pairs * (d // 2) >= n_openings for m, d in enumerate(games_done)):
            break
    stats = pool.stats()
    pool.close()
    for hs in players:
        for h in hs:
            olib.ago_game_destroy(h)
    return compared, stats, results


@pytest.mark.parametrize("rules,batch,sims,max_children,merged,symmetries", [(0, 4, 60, 0, False, 0), (1, 8, 80, 20, False, 0), (2, 4, 60, 0, False, 0),
                                                                              (0, 8, 60, 0, True, 0), (1, 4, 80, 20, True, 1), (2, 4, 60, 0, True, 0)])
def test_evaluation_matches_bit_exact(agx_lib, olib, rules, batch, sims, max_children, merged, symmetries):
    """match_mode: EvaluationGame + Player (evaluation/EvaluationGame.cpp:77-146, evaluation/Player.cpp:98-216): two players with
    their own trees, solvers and networks share a game; every opening is played twice with the colours swapped; a player's tree
    jumps two plies per setBoard and survives from game to game; the root is pruned like any node.  Device vs oracle after
    every half-step (features of every scheduled leaf, root edges of the searching tree), for two pairs playing two matches each;
    stepped as two groups and as merged launches over both players' trees (agx_engine_step_match)."""
    compared, stats, results = _play_matches_and_compare(olib, rules, pairs=2, n_openings=4, batch=batch, sims=sims, max_steps=6000, max_children=max_children,
                                                         draw_after=80, merged=merged, use_symmetries=symmetries)
    assert len(results) == 8 and compared > 200                      # 4 openings x 2 games
    assert stats["games_finished"] == 8
    by_opening = {}
    for m, oid, k, outcome in results:
        by_opening.setdefault(oid, []).append(k)
    assert all(sorted(v) == [0, 1] for v in by_opening.values())     # each opening once per colour assignment


def test_evaluation_matches_with_two_networks(agx_lib):
    """match_mode end to end on the device: two HIP networks of different weights, 32 pairs; every pair plays whole matches, the
    per-pair scores add up and each finished game was recorded move by move"""
    from alphagomoku_amd import selfplay, networks, synthetic
    pairs, n_openings = 32, 48
    cfg = selfplay.default_config(rules=0, board_size=N, draw_after=60, n_games=2 * pairs, max_batch_size=8, max_simulations=50, tss_table_entries=1 << 16,
                                  node_capacity=4096, edge_capacity=65536, match_mode=1, solver_yield_fraction=0.75,
                                  record_capacity=2 * n_openings * 64, record_edge_capacity=2 * n_openings * 64 * 230)
    pool = selfplay.GeneratorPool(cfg)
    nets = []
    for seed in (1234, 4321):
        d = synthetic.net_desc(blocks=2, filters=64)
        net = networks.AGNetwork(d)
        net.loadWeights(synthetic.make_weights(d, seed=seed)[0])
        nets.append(net)
    openings = []
    for g in range(n_openings):
        op = np.zeros(selfplay.OPENING_CAP, np.uint16)
        agx_lib.agx_make_opening(0, N, 900 + g, op.ctypes.data_as(ctypes.c_void_p))
        openings.append([int(x) for x in op[1:1 + int(op[0])]])
    pool.begin(selfplay.pack_openings(openings))
    for step in range(6000):
        pool.step_match(nets[0], nets[1])
        if step % 50 == 49:
            res = pool.match_results()
            if int(res[:, 3].sum()) == 2 * n_openings:
                break
    res = pool.match_results()
    stats = pool.stats()
    assert int(res[:, 3].sum()) == 2 * n_openings == stats["games_finished"]
    assert np.array_equal(res[:, :3].sum(1), res[:, 3]) and np.all(res[:, 3] % 2 == 0)
    recs, _ = pool.records()
    assert len(recs) == stats["moves_played"]
    per_opening = {}
    for r in recs:
        per_opening.setdefault(r.game_serial, []).append(r.move_number)
    assert sorted(per_opening) == list(range(n_openings))
    for oid, numbers in per_opening.items():   # two games per opening, each recorded from the opening's length on without gaps
        first = len(openings[oid])
        assert numbers.count(first) == 2 and max(numbers) <= 60
    for g in range(2 * pairs):
        assert pool.game_info(g, with_edges=False)["error"] == 0
    pool.close()
    for net in nets:
        net.close()


def test_configuration_errors_are_reported(agx_lib):
    """the boundary's error behaviour: invalid configurations fail with a status and a message, nothing is launched"""
    from alphagomoku_amd import selfplay, synthetic
    from alphagomoku_amd._lib import AgxError
    from alphagomoku_amd.networks import AGNetwork
    with pytest.raises(AgxError, match="n_games must be even"):
        selfplay.GeneratorPool(selfplay.default_config(n_games=7, match_mode=1, tss_table_entries=1 << 12))
    with pytest.raises(AgxError, match="policy_temperature"):
        selfplay.GeneratorPool(selfplay.default_config(n_games=4, policy_temperature=-1.0, tss_table_entries=1 << 12))
    with pytest.raises(AgxError, match="final_selector"):
        selfplay.GeneratorPool(selfplay.default_config(n_games=4, final_selector=9, tss_table_entries=1 << 12))
    pool = selfplay.GeneratorPool(selfplay.default_config(n_games=4, max_simulations=50, tss_table_entries=1 << 12, node_capacity=1024, edge_capacity=16384))
    d = synthetic.net_desc(blocks=2, filters=64)
    net = AGNetwork(d)
    net.loadWeights(synthetic.make_weights(d)[0])
    with pytest.raises(AgxError, match="agx_engine_begin has not been called"):
        pool.step(net)
    pool.begin(selfplay.pack_openings([[], [], [], []]))
    with pytest.raises(AgxError, match="match_mode"):
        pool.step_match(net, net)                       # a self-play pool has no pairs
    with pytest.raises(AgxError, match="match_mode"):
        pool.match_results()
    pool.step(net)                                      # and still works afterwards
    assert pool.stats()["first_error"] == 0
    pool.close()
    net.close()


@pytest.mark.parametrize("overrides", [dict(), dict(rules=2, solver_yield_fraction=0.5, speculative_solver=1), dict(board_size=20, rules=3, speculative_solver=1),
                                       dict(speculative_solver=0, max_batch_size=4)])
def test_sizing_pass_equals_what_an_engine_allocates(agx_lib, overrides):
    """agx_engine_estimate_device_bytes (no device touched: bench.py --plan-only sizes the ranks of a multi-GPU job with it) adds up exactly the
    allocations agx_engine_create makes on this device"""
    from alphagomoku_amd import selfplay, lib, check
    cus = ctypes.c_int()
    check(lib.agx_device_cu_count(ctypes.byref(cus)))
    base = dict(n_games=24, max_batch_size=8, max_simulations=100, tss_table_entries=1 << 14, node_capacity=4096, edge_capacity=131072)
    base.update(overrides)
    cfg = selfplay.default_config(**base)
    need = ctypes.c_ulonglong()
    check(lib.agx_engine_estimate_device_bytes(ctypes.byref(cfg), cus.value, ctypes.byref(need)))
    pool = selfplay.GeneratorPool(cfg)
    assert pool.device_bytes() == need.value and need.value > 0
    pool.close()
