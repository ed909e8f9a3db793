"""CPU checks of the oracle's evaluation-match and policy-temperature restatements (the device is compared against them in
tests/test_engine_gpu.py; here: internal consistency, no GPU)."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol

N, HW = 15, 225


@pytest.fixture(scope="module")
def lib():
    return ol.load()


def _root(lib, h):
    rv, rs = ctypes.c_int(), ctypes.c_uint16()
    rval = (ctypes.c_float * 2)()
    em, ev, ep, evl, es, ef = (np.zeros(512, np.uint16), np.zeros(512, np.int32), np.zeros(512, np.float32), np.zeros(1024, np.float32),
                               np.zeros(512, np.uint16), np.zeros(512, np.uint16))
    n = lib.ago_game_root(h, ctypes.byref(rv), rval, ctypes.byref(rs), ol.ptr(em), ol.ptr(ev), ol.ptr(ep), ol.ptr(evl), ol.ptr(es), ol.ptr(ef), 512)
    return n, rv.value, em[:n].copy(), ep[:n].copy()


def _fake(lib, f, c):
    pol = np.zeros((max(c, 1), HW), np.float32)
    val = np.zeros((max(c, 1), 2), np.float32)
    if c:
        lib.ago_fake_eval(c, HW, ol.ptr(f), ol.ptr(pol), ol.ptr(val))
    return pol, val


def test_match_of_two_players_colours_swap_and_copies_agree(lib):
    """EvaluationGame (evaluation/EvaluationGame.cpp:44-146): one opening, two games with the colours swapped; each player searches
    only on its own turns and both copies of the game always agree"""
    cfg = ol.default_search_config(max_batch_size=4, max_simulations=60, table_entries=1 << 14)   # (>= 50: the draw-rate rule of misc.cpp:171-179 assumes it)
    players = []
    for _ in range(2):
        h = lib.ago_game_create_ex(0, N, N, 40, ctypes.byref(cfg))
        lib.ago_game_set_force_expand_root(h, 0)
        players.append(h)
    op = np.zeros(64, np.uint16)
    k = lib.ago_prepare_opening(0, N, N, 4242, ol.ptr(op))
    turns = [[0, 0], [0, 0]]   # [game][player] number of moves made
    for game in range(2):
        for h in players:
            lib.ago_game_match_begin(h, ol.ptr(op), k)
        first_sign = 1 if game == 0 else 2
        mover = 0 if lib.ago_game_sign_to_move(players[0]) == first_sign else 1
        lib.ago_game_take_turn(players[mover])
        for _ in range(20000):
            h = players[mover]
            f = np.zeros((4, HW), np.uint32)
            c = lib.ago_game_step_select(h, ol.ptr(f), 4)
            pol, val = _fake(lib, f, c)
            if lib.ago_game_step_expand(h, ol.ptr(pol), ol.ptr(val)):
                mv = lib.ago_game_last_move(h)
                assert (mv & 3) == (first_sign if mover == 0 else 3 - first_sign)      # a player only ever plays its own colour
                turns[game][mover] += 1
                other = players[1 - mover]
                lib.ago_game_external_move(other, mv)
                assert lib.ago_game_outcome(other) == lib.ago_game_outcome(h)
                assert lib.ago_game_sign_to_move(other) == lib.ago_game_sign_to_move(h)
                if lib.ago_game_outcome(h) != 0:
                    break
                lib.ago_game_take_turn(other)
                mover = 1 - mover
        assert lib.ago_game_outcome(players[0]) != 0
        assert abs(turns[game][0] - turns[game][1]) <= 1
    assert sum(turns[0]) + k <= 40 and sum(turns[1]) + k <= 40      # draw_after 40 bounds both games
    for h in players:
        lib.ago_game_destroy(h)


@pytest.mark.parametrize("temperature", [0.0, 0.5, 2.0])
def test_policy_temperature_of_the_root_priors(lib, temperature):
    """initialize_edges (EdgeGenerator.cpp:88-127): root priors are policy^(1/T) renormalised (arg-max indicator for T = 0)"""
    cfg = ol.default_search_config(max_batch_size=1, max_simulations=10, table_entries=1 << 12)
    op = np.zeros(64, np.uint16)
    k = lib.ago_prepare_opening(0, N, N, 99, ol.ptr(op))
    roots = {}
    for t in (1.0, temperature):
        h = lib.ago_game_create_ex(0, N, N, 0, ctypes.byref(cfg))
        lib.ago_game_set_policy_temperature(h, t)
        lib.ago_game_begin(h, ol.ptr(op), k)
        f = np.zeros((1, HW), np.uint32)
        c = lib.ago_game_step_select(h, ol.ptr(f), 1)
        assert c == 1                                    # the root itself
        pol, val = _fake(lib, f, c)
        lib.ago_game_step_expand(h, ol.ptr(pol), ol.ptr(val))
        roots[t] = _root(lib, h)
        lib.ago_game_destroy(h)
    n1, _, moves1, p1 = roots[1.0]
    nt, _, movest, pt = roots[temperature]
    assert n1 == nt and np.array_equal(moves1, movest)
    if temperature == 0.0:
        # one-hot over the edges that hold the maximum of the WHOLE policy plane; if that maximum is not an edge every prior is 0
        # and renormalize_policy makes the priors uniform
        assert np.allclose(pt.sum(), 1.0, atol=1e-5)
        assert set(np.unique(pt)).issubset({np.float32(0.0), np.float32(1.0)}) or np.allclose(pt, 1.0 / nt, atol=1e-6)
    else:
        want = p1.astype(np.float64) ** (1.0 / temperature)
        want /= want.sum()
        assert np.allclose(pt, want, rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("rules,threads,batch", [(0, 1, 8), (1, 3, 4)])
def test_double_buffered_schedule_is_the_serial_loop_shifted_by_one_batch(lib, rules, threads, batch):
    """SearchThread::asynchronous_run (player/SearchThread.cpp:148-180; Search::useBuffer / switchBuffer / cleanup, Search.cpp:233-252) in the
    oracle: the first iteration of a search expands nothing, every later one expands the batch selected two iterations earlier — so while a
    search lasts, two batches hold virtual losses at select time and the tree's visit count lags the serial loop's by exactly one batch;
    a move drops the batch still in flight (the network evaluated it: positions scheduled > nodes backed up); the game ends like any game.
    (The device's double-buffered pool is compared with this schedule leaf by leaf in tests/test_engine_gpu.py.)"""
    sims = 120
    cfg = ol.default_search_config(max_batch_size=batch, max_simulations=sims, table_entries=1 << 14)
    op = np.zeros(64, np.uint16)
    k = lib.ago_prepare_opening(rules, N, N, 181, ol.ptr(op))
    games = []
    for _ in range(2):
        h = lib.ago_game_create_ex(rules, N, N, 0, ctypes.byref(cfg))
        lib.ago_game_set_search_threads(h, threads)
        lib.ago_game_set_serial(h, 0)
        lib.ago_game_begin(h, ol.ptr(op), k)
        games.append(h)
    a, s = games   # asynchronous / serial
    # the first search of both, iteration by iteration, until the serial one moves
    scheduled_async, first = [], True
    for it in range(400):
        f = np.zeros((threads * batch, HW), np.uint32)
        c = lib.ago_game_async_step(a, ol.ptr(f), threads * batch)
        pol, val = _fake(lib, f, c)
        lib.ago_game_async_provide(a, ol.ptr(pol), ol.ptr(val))
        scheduled_async.append(c)
        n_a, visits_a, _, _ = _root(lib, a)
        if lib.ago_game_num_records(a) > 0:
            break
        if it == 0:
            assert n_a == 0 and visits_a == 0 and c == threads    # nothing expanded yet; every thread sent the root itself to the network
        if it == 1:
            assert visits_a == 0 and c == threads                 # buffer 1 selected the (still unexpanded) root too: one leaf per thread
    assert lib.ago_game_num_records(a) == 1 and it > 3
    # the same game through the serial loop: its first move comes from a tree with at most one batch per thread less of lag
    for it_s in range(400):
        f = np.zeros((threads * batch, HW), np.uint32)
        c = lib.ago_game_step_select(s, ol.ptr(f), threads * batch)
        pol, val = _fake(lib, f, c)
        if lib.ago_game_step_expand(s, ol.ptr(pol), ol.ptr(val)):
            break
    assert lib.ago_game_num_records(s) == 1
    assert it >= it_s + 1                                         # the double-buffered loop needs at least one more iteration: its expansions lag
    # play the asynchronous game to its end: every search stops by the same rule, the game ends, statistics stay consistent
    for _ in range(40000):
        if lib.ago_game_outcome(a) != 0:
            break
        f = np.zeros((threads * batch, HW), np.uint32)
        c = lib.ago_game_async_step(a, ol.ptr(f), threads * batch)
        pol, val = _fake(lib, f, c)
        lib.ago_game_async_provide(a, ol.ptr(pol), ol.ptr(val))
    assert lib.ago_game_outcome(a) in (1, 2, 3) and lib.ago_game_num_records(a) >= 5
    st = np.zeros(16, np.uint64)
    lib.ago_game_stats(a, ol.ptr(st))
    for h in games:
        lib.ago_game_destroy(h)
