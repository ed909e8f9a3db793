"""The reference-named C++ boundary (include/alphagomoku_agx/, libagx_ag.so) driven by a C++ program that is the reference's own call
chain (tests/cpp/boundary_main.cpp: training_launcher/launcher.cpp:63-72 -> TrainingManager::generateGames -> GeneratorManager::generate
with one GeneratorThread per device), on the GPU box.  Two generator threads share device 0 here (there is one GPU)."""
import ctypes
import json
import os
import subprocess
import zlib

import numpy as np
import pytest

import oracle_lib as ol
from alphagomoku_amd import synthetic

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINARY = os.path.join(ROOT, "alphagomoku_amd", "agx_boundary_test")


def run(mode, *args):
    p = subprocess.run([BINARY, mode] + [str(a) for a in args], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:] + p.stdout[-2000:]
    lines = [x for x in p.stdout.splitlines() if x.startswith('{"mode"')]
    assert len(lines) == 1, p.stdout[-3000:]
    return json.loads(lines[0]), p.stdout


@pytest.fixture(scope="module")
def network_file(agx_lib, tmp_path_factory):
    d = synthetic.net_desc(blocks=2, filters=64)
    blob, _ = synthetic.make_weights(d)
    path = tmp_path_factory.mktemp("net") / "network.agxw"
    synthetic.save_weights(path, d, blob)
    return str(path), d, blob


def parse_game(data, n):
    """GameDataStorage (format 201) parsing constructor (dataset/GameDataStorage.cpp:27-69)"""
    off = 0
    n_samples = int(data[off:off + 4].view(np.uint32)[0])
    off += 4
    samples = []
    for _ in range(n_samples):
        count = int(data[off + 12:off + 16].view(np.uint32)[0])
        samples.append(data[off:off + 16 + 6 * count])
        off += 16 + 6 * count
    n_moves = int(data[off:off + 4].view(np.uint32)[0])
    moves = data[off + 4:off + 4 + 2 * n_moves].view(np.uint16)
    off += 4 + 2 * n_moves
    outcome, rows, cols = (int(x) for x in data[off:off + 12].view(np.int32))
    assert off + 12 == data.size and rows == n and cols == n
    return samples, moves, outcome


def test_generator_manager_call_chain(network_file, tmp_path):
    path, d, _ = network_file
    olib = ol.load()
    out = tmp_path / "work"
    out.mkdir()
    line, stdout = run("generate", "--network", path, "--games", 24, "--games-per-thread", 32, "--devices", "0,0", "--sims", 60, "--batch", 4,
                       "--out", out, "--nn-batch", 64)   # 32 games x 4 = 128 slots, 64 per launch -> 2 slices per thread, pipelined on 2 streams
    assert line["threads"] == 2 and line["games"] >= 24 and line["samples"] > line["games"]
    assert line["cross_win"] + line["draws"] + line["circle_win"] == line["games"]
    assert "Played games" in stdout and "----SearchStats----" in stdout and "----NNEvaluator----" in stdout       # printStats
    # the saved buffer: GameDataBuffer::save = JSON header line + the games' bytes, zlib-wrapped
    blob = zlib.decompress((out / "buffer_0.bin").read_bytes())
    header, _, body = blob.partition(b"\n")
    meta = json.loads(header)
    assert meta["format"] == 201 and meta["config"] == {"rules": "FREESTYLE", "rows": 15, "cols": 15, "draw_after": 225}
    assert len(meta["offsets"]) == line["games"] and (out / "saved_state" / "buffer.bin").exists()
    raw = np.frombuffer((out / "games.raw").read_bytes(), np.uint8)
    pos, total_samples = 0, 0
    for i in range(line["games"]):
        size = int(raw[pos:pos + 4].view(np.uint32)[0])
        game = raw[pos + 4:pos + 4 + size]
        pos += 4 + size
        end = meta["offsets"][i + 1] if i + 1 < line["games"] else len(body)
        assert bytes(game) == body[meta["offsets"][i]:end]
        samples, moves, outcome = parse_game(game, 15)
        total_samples += len(samples)
        assert outcome in (1, 2, 3) and len(moves) >= len(samples) >= 1
        # replaying the moves gives the recorded outcome, and only on the last move (getOutcome, rules.cpp:110-133)
        board = np.zeros(225, np.uint8)
        for k, m in enumerate(moves):
            s, r, c = int(m) & 3, (int(m) >> 2) & 127, (int(m) >> 9) & 127
            assert board[r * 15 + c] == 0 and s == 1 + (k % 2)
            board[r * 15 + c] = s
            res = olib.ago_outcome(0, 15, 15, ol.ptr(board), s, r, c, 225)
            assert (res != 0) == (k == len(moves) - 1) and (res == 0 or res == outcome)
        # every sample decodes (storeTo) on the position it was taken from: move_number stones, entries on empty cells only
        first = len(moves) - len(samples)
        for k, smp in enumerate(samples):
            hw = 225
            visits, prior, value, score = np.zeros(hw, np.int32), np.zeros(hw, np.float32), np.zeros((hw, 2), np.float32), np.zeros(hw, np.uint16)
            header3, mm = np.zeros(3, np.int32), np.zeros(2, np.float32)
            used = olib.ago_sample_v201_unpack(ol.ptr(np.ascontiguousarray(smp)), 15, 15, ol.ptr(visits), ol.ptr(prior), ol.ptr(value), ol.ptr(score),
                                               ol.ptr(header3), ol.ptr(mm))
            assert used == smp.size and header3[1] == first + k
            occupied = np.zeros(hw, bool)
            for m in moves[:first + k]:
                occupied[((int(m) >> 2) & 127) * 15 + ((int(m) >> 9) & 127)] = True
            assert not visits[occupied].any()
            assert visits.sum() > 0 or ((header3[0] >> 13) & 3) != 2      # no visits at all only under a proven root
    assert total_samples == line["samples"]


def test_nn_evaluator_with_host_tasks(network_file, agx_lib, tmp_path):
    """NNEvaluator::addToQueue(task, symmetry) / evaluateGraph / asyncEvaluateGraphLaunch + Join (NNEvaluator.cpp:134-286): the task gets the
    network's output of the AUGMENTED features mapped back by the inverse symmetry"""
    from alphagomoku_amd.networks import AGNetwork
    path, d, blob = network_file
    olib = ol.load()
    feats = synthetic.random_features(1, 15, 15, seed=5)[0]
    fpath = tmp_path / "features.bin"
    fpath.write_bytes(feats.tobytes())
    opath = tmp_path / "outputs.bin"
    line, _ = run("evaluator", "--network", path, "--features", fpath, "--out", opath)
    assert line == {"mode": "evaluator", "queue_full": 1, "queued": 16, "processed": 1, "second_launch_refused": 1, "samples": 19, "outputs": "pv"}
    got = np.frombuffer(opath.read_bytes(), np.float32).reshape(19, 227)
    net = AGNetwork(d)
    net.loadWeights(blob)
    for i, s in enumerate([k % 8 for k in range(16)] + [5, 5, 5]):
        aug = np.zeros(225, np.uint32)
        olib.ago_apply_symmetry(15, s, 1, ol.ptr(feats), ol.ptr(aug))          # NNInputFeatures::augment
        p, v = net.forward(aug.reshape(1, 225))
        back = np.zeros(225, np.uint32)
        olib.ago_apply_symmetry(15, olib.ago_inverse_symmetry(s), 0, ol.ptr(np.ascontiguousarray(p[0]).view(np.uint32)), ol.ptr(back))
        assert np.array_equal(back.view(np.float32), got[i, :225]), (i, s)       # same kernel, same bits
        assert np.array_equal(v[0, :2], got[i, 225:]), (i, s)
    net.close()


def test_repeated_generate_calls_reuse_their_streams(network_file, tmp_path):
    """GeneratorManager::generate once per training iteration: every call sets the generator threads (and their CU-masked slice streams) up
    again.  Such streams cannot be destroyed (ROCm 7.2), so they are cached per (device, mask): 12 iterations end with exactly the streams
    of one (2 slices per thread: 2 masks)."""
    path, _, _ = network_file
    out = tmp_path / "work"
    out.mkdir()
    line, _ = run("generate", "--network", path, "--games", 12, "--iterations", 12, "--games-per-thread", 16, "--devices", "0", "--sims", 30, "--batch", 4,
                  "--out", out, "--nn-batch", 32)
    assert line["iterations"] == 12 and line["games"] >= 12
    assert line["masked_streams"] == 2


def test_players_with_the_reference_constructors(network_file, agx_lib, tmp_path):
    """evaluation/Player.cpp:64-129,205-212 compiled against include/alphagomoku_agx/ as written — Tree(const TreeConfig&),
    Search(const GameConfig&, const SearchConfig&), cleanup / setBoard / setEdgeSelector / setEdgeGenerator / select / solve / scheduleToNN /
    generateEdges / expand / backup, Tree::getInfo({}), EdgeSelector::create(final)->select(&root) — two such players play a game against each
    other (EvaluationGame.cpp:77-143).  The same opening, networks and budgets on the match-mode pool (whose every step the engine tests
    compare with the oracle) must give the same moves."""
    from alphagomoku_amd import selfplay
    from alphagomoku_amd.networks import AGNetwork
    path, d, blob = network_file
    blob2, _ = synthetic.make_weights(d, seed=77)
    path2 = tmp_path / "second.agxw"
    synthetic.save_weights(path2, d, blob2)
    seed, sims, batch = 11, 60, 4
    line, _ = run("player", "--network", path, "--network2", path2, "--sims", sims, "--batch", batch, "--opening-seed", seed, "--table-entries", 1 << 16)
    assert line["outcome"] in (1, 2, 3) and len(line["moves"]) >= 20
    # the same game on the match-mode engine
    nets = []
    for b in (blob, blob2):
        net = AGNetwork(d)
        net.loadWeights(b)
        nets.append(net)
    opening = synthetic.make_openings(15, 1, seed0=seed)
    assert len(opening[0]) == line["opening_stones"]
    cfg = selfplay.default_config(n_games=2, max_batch_size=batch, max_simulations=sims, tss_table_entries=1 << 16, node_capacity=4096, edge_capacity=65536,
                                  match_mode=1)
    pool = selfplay.GeneratorPool(cfg)
    pool.begin(selfplay.pack_openings(opening))
    for _ in range(20000):
        pool.step_match(nets[0], nets[1])
        if pool.stats()["games_finished"] >= 1:
            break
    recs, _ = pool.records()
    first_game = sorted((r.move_number, r.move) for r in recs if r.game_index == 0)
    pool.close()
    for net in nets:
        net.close()
    assert [m for _, m in first_game] == line["moves"]


@pytest.mark.parametrize("asynchronous", [0, 1])
def test_search_thread_loops_with_the_reference_constructors(network_file, agx_lib, asynchronous):
    """player/SearchThread.cpp:121-199 compiled against include/alphagomoku_agx/ as written — serial_run, and asynchronous_run with
    Search::useBuffer / switchBuffer, NNEvaluator::asyncEvaluateGraphLaunch / Join and the stop condition read through Tree::getNodeCount /
    getSimulationCount / isRootProven — drives a game (tests/cpp/boundary_main.cpp, mode thread).  The same procedure through the C ABI on a
    double-buffered engine (search_buffers = 2, whose every iteration test_double_buffered_tournament_search compares with the oracle), one
    stream, everything in order, must give the same moves and root visit counts: the classes' two streams and events change when things run,
    not what is computed."""
    from alphagomoku_amd import selfplay, lib, check
    from alphagomoku_amd.networks import AGNetwork
    from test_engine_gpu import _best_edge
    path, d, blob = network_file
    seed, sims, batch, plies, n = 11, 150, 8, 14, 15
    line, _ = run("thread", "--network", path, "--sims", sims, "--batch", batch, "--opening-seed", seed, "--table-entries", 1 << 16, "--plies", plies, "--async", asynchronous)
    assert line["asynchronous"] == asynchronous and len(line["moves"]) >= 1
    net = AGNetwork(d)
    net.loadWeights(blob)
    opening = synthetic.make_openings(n, 1, seed0=seed)[0]
    assert len(opening) == line["opening_stones"]
    cfg = selfplay.default_config(n_games=2, search_buffers=2, max_batch_size=batch, max_simulations=1 << 24, tss_table_entries=1 << 16, node_capacity=4096,
                                  edge_capacity=65536, force_expand_root=0, speculative_solver=1)
    pool = selfplay.GeneratorPool(cfg)
    pool.begin(selfplay.pack_openings([[]]))
    board = np.zeros(n * n, np.uint8)
    sign = 1
    for m in opening:
        board[((m >> 2) & 127) * n + ((m >> 9) & 127)] = m & 3
        sign = 3 - (m & 3)
    out4 = (ctypes.c_int * 4)()

    def stop():
        check(lib.agx_engine_root_summary(pool._h, 0, None, out4))
        return out4[2] != 0 and (out4[0] >= sims or out4[1] != 0)

    moves, visits, outcome, iterations = [], [], 0, 0
    while outcome == 0 and len(moves) < plies:
        pool.set_board(0, board, sign)
        if not stop():
            b = 0
            while True:
                if asynchronous:
                    check(lib.agx_engine_expand_group(pool._h, b, 2, None))
                    iterations += 1
                    if stop():
                        break
                pool.set_max_simulations(sims)
                pool.select_solve_group(b, 2)
                pool.evaluate_group(net, b, 2)
                if asynchronous:
                    b = 1 - b
                else:
                    check(lib.agx_engine_expand_group(pool._h, 0, 2, None))
                    iterations += 1
                    if stop():
                        break
        pool.cancel_pending()
        info = pool.game_info(0)
        e = info["edges"][_best_edge(info["root_visits"], info["edges"])]
        mv = int(e["move"])
        moves.append(mv)
        visits.append(info["root_visits"])
        row, col = (mv >> 2) & 127, (mv >> 9) & 127
        board[row * n + col] = mv & 3
        sign = 3 - (mv & 3)
        res = ctypes.c_int(0)
        check(lib.agx_get_outcome(0, n, board.ctypes.data_as(ctypes.c_void_p), mv & 3, row, col, n * n, ctypes.byref(res)))
        outcome = res.value
    st = pool.stats()
    pool.close()
    net.close()
    assert moves == line["moves"] and visits == line["root_visits"] and outcome == line["outcome"]
    assert iterations == line["iterations"] and st["evaluated_nodes"] == line["simulations"]
    assert min(visits) >= sims or outcome != 0


def test_game_generators_with_the_reference_constructor(network_file, agx_lib):
    """GameGenerator(gameOptions, selfplayOptions, manager, evaluator) (selfplay/GameGenerator.hpp:54): generators of one game each, driven by
    the reference's generator-thread loop, hand their finished games to the manager's buffer"""
    path, _, _ = network_file
    line, _ = run("generator", "--network", path, "--generators", 3, "--games", 2, "--sims", 40)
    assert line["generators"] == 3 and line["games"] >= 2 and line["samples"] > line["games"]


def test_boundary_error_behaviour(agx_lib):
    line, _ = run("errors")
    assert line["caught"] == 31
