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


@pytest.mark.parametrize("rules,games_per_thread", [(0, 32), (2, 64)])
def test_generator_manager_call_chain(network_file, tmp_path, rules, games_per_thread):
    """training_launcher's call chain on the reference-named classes; (2, 64): renju with pools large enough for GeneratorThread's pacing rules
    (64 games: yield fraction 0.5, and the renju search launches park the solves that outlast them) — the games must still be legal games whose
    recorded outcome is what replaying their moves gives"""
    path, d, _ = network_file
    olib = ol.load()
    out = tmp_path / "work"
    out.mkdir()
    line, stdout = run("generate", "--network", path, "--rules", rules, "--games", 24, "--games-per-thread", games_per_thread, "--devices", "0,0", "--sims", 60, "--batch", 4,
                       "--out", out, "--nn-batch", 2 * games_per_thread)   # 32 games x 4 = 128 slots, 64 per launch -> 2 slices per thread, pipelined on 2 streams
    assert line["threads"] == 2 and line["games"] >= 24 and line["samples"] > line["games"]
    assert line["cross_win"] + line["draws"] + line["circle_win"] == line["games"]
    assert "Played games" in stdout and "----SearchStats----" in stdout and "----NNEvaluator----" in stdout       # printStats
    # the saved buffer: GameDataBuffer::save = JSON header line + the games' bytes, zlib-wrapped
    blob = zlib.decompress((out / "buffer_0.bin").read_bytes())
    header, _, body = blob.partition(b"\n")
    meta = json.loads(header)
    assert meta["format"] == 201 and meta["config"] == {"rules": "RENJU" if rules == 2 else "FREESTYLE", "rows": 15, "cols": 15, "draw_after": 225}
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
            res = olib.ago_outcome(rules, 15, 15, ol.ptr(board), s, r, c, 225)
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


def parse_saved_games(path):
    """saved_state/thread_<i>.bin (GeneratorThread::saveGames): "AGXS", u32 version, u32 generators, per generator u32 count and per game an
    AgxSavedGame (6 ints, 400 u16), u64 n, n bytes of { i32 move number, u32 bytes, format-201 sample } records"""
    data = np.frombuffer(path.read_bytes(), np.uint8)
    assert bytes(data[:4]) == b"AGXS" and int(data[4:8].view(np.uint32)[0]) == 1
    generators = int(data[8:12].view(np.uint32)[0])
    off, games = 12, []
    for _ in range(generators):
        count = int(data[off:off + 4].view(np.uint32)[0])
        off += 4
        for _ in range(count):
            slot, index, opening_id, sign, nn_queued, n_moves = (int(x) for x in data[off:off + 24].view(np.int32))
            moves = [int(m) for m in data[off + 24:off + 24 + 2 * n_moves].view(np.uint16)]
            off += 24 + 800
            n = int(data[off:off + 8].view(np.uint64)[0])
            off += 8
            rec, samples = off, []
            while rec < off + n:
                move_number = int(data[rec:rec + 4].view(np.int32)[0])
                size = int(data[rec + 4:rec + 8].view(np.uint32)[0])
                samples.append((move_number, bytes(data[rec + 8:rec + 8 + size])))
                rec += 8 + size
            assert rec == off + n
            off += n
            games.append(dict(slot=slot, index=index, sign=sign, moves=moves, samples=samples))
    assert off == data.size
    return games


def read_games(out, n):
    raw = np.frombuffer((out / "games.raw").read_bytes(), np.uint8)
    pos, games = 0, []
    while pos < raw.size:
        size = int(raw[pos:pos + 4].view(np.uint32)[0])
        games.append(parse_game(raw[pos + 4:pos + 4 + size], n))
        pos += 4 + size
    return games


def test_generate_stops_on_sigint_and_the_next_start_resumes(network_file, tmp_path):
    """GeneratorManager::generate (GeneratorManager.cpp:196-208) with the process's SIGINT handler installed the way TrainingManager does
    (utils/os_utils.hpp:62-63): a captured SIGINT stops the generator threads, generate() returns, the caller's saveState(was_interrupted)
    writes the buffer AND the games in flight, the process ends normally — and the next start loads both and goes on.  The statistics are
    printed periodically while it runs (every 60 s in the reference; 2 s here)."""
    import signal
    import time
    path, d, _ = network_file
    out = tmp_path / "work"
    out.mkdir()
    common = ["--network", path, "--games-per-thread", 16, "--devices", "0,0", "--sims", 50, "--batch", 4, "--out", out, "--nn-batch", 32]
    cmd = [str(a) for a in [BINARY, "generate", "--games", 1000000, "--interruptible", 1, "--stats-period", 2] + common]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        deadline = time.time() + 600
        while time.time() < deadline:
            line = p.stdout.readline()
            if line.startswith("generating") or line == "":
                break
        assert p.poll() is None, p.stderr.read()[-2000:]
        time.sleep(12.0)                     # set-up (networks, openings) + a few seconds of play
        p.send_signal(signal.SIGINT)
        stdout, stderr = p.communicate(timeout=300)
    finally:
        if p.poll() is None:
            p.kill()
    assert p.returncode == 0, stderr[-3000:] + stdout[-2000:]
    assert "Caught interruption signal" in stdout and "Generators stopped" in stdout and "Saving buffer" in stdout
    assert stdout.count("Played games = ") >= 2          # the periodic statistics + the caller's printStats
    first = json.loads([x for x in stdout.splitlines() if x.startswith('{"mode"')][0])
    assert first["interrupted"] is True and 0 < first["games"] < 1000000
    state = out / "saved_state"
    assert (state / "buffer.bin").exists() and (state / "thread_0.bin").exists() and (state / "thread_1.bin").exists()
    in_flight = sum(len(parse_saved_games(state / ("thread_%d.bin" % t))) for t in range(2))
    assert in_flight >= 16
    second, stdout2 = run("generate", "--games", first["games"] + 8, "--interruptible", 1, *common)
    assert second["interrupted"] is False and "Loaded buffer" in stdout2 and "thread_1.bin" in stdout2 and second["games"] >= first["games"] + 8
    # an uninterrupted run's saveState(false) leaves no buffer behind (TrainingManager.cpp:207), only the games in flight
    assert not (state / "buffer.bin").exists() and (state / "thread_0.bin").exists()


def test_generator_manager_restart_continues_the_games_in_flight(network_file, tmp_path):
    """GeneratorManager::saveState / loadState (GeneratorManager.cpp:241-290) and GameGenerator::save / load (GameGenerator.cpp:122-141) across
    two PROCESSES: the first plays 12 games and stops — its buffer goes to saved_state/buffer.bin, the games still in flight (their moves and
    the samples collected so far) to saved_state/thread_<i>.bin; the second loads both and goes on to 30 games.  The buffer's games come back
    byte for byte, every game that was in flight continues from its saved move prefix (trees are rebuilt, as in the reference) and, when it
    ends, carries the samples of BOTH processes."""
    path, d, _ = network_file
    out = tmp_path / "work"
    out.mkdir()
    common = ["--network", path, "--games-per-thread", 16, "--devices", "0,0", "--sims", 50, "--batch", 4, "--out", out, "--nn-batch", 32]
    first, _ = run("generate", "--games", 12, *common)
    assert first["games"] >= 12
    games_1 = read_games(out, 15)
    state = out / "saved_state"
    assert (state / "buffer.bin").exists() and (state / "thread_0.bin").exists() and (state / "thread_1.bin").exists()
    in_flight = [parse_saved_games(state / ("thread_%d.bin" % t)) for t in range(2)]
    assert sum(len(x) for x in in_flight) >= 16      # most of the 2 x 16 slots hold a game when the threads stop
    assert any(len(g["samples"]) > 0 for t in in_flight for g in t)
    for t in in_flight:
        for g in t:
            stones = len(g["moves"])
            assert [mn for mn, _ in g["samples"]] == list(range(stones - len(g["samples"]), stones))   # one sample per move played so far
            assert g["sign"] == (1 if stones % 2 == 0 else 2)

    second, stdout = run("generate", "--games", 30, *common)
    assert "Loaded buffer" in stdout and "thread_0.bin" in stdout and second["games"] >= 30
    games_2 = read_games(out, 15)
    # the loaded buffer: the first process's games, in order, byte for byte
    for a, b in zip(games_1, games_2):
        assert a[2] == b[2] and np.array_equal(a[1], b[1]) and len(a[0]) == len(b[0]) and all(np.array_equal(x, y) for x, y in zip(a[0], b[0]))
    assert not (state / "buffer.bin").exists() or second["games"] == len(games_2)   # (loadState removes the file, saveState wrote a new one)
    later = games_2[len(games_1):]
    still = [parse_saved_games(state / ("thread_%d.bin" % t)) for t in range(2)]
    resumed = 0
    for t in range(2):
        for g in in_flight[t]:
            prefix = g["moves"]
            ended = [x for x in later if len(x[1]) >= len(prefix) and [int(m) for m in x[1][:len(prefix)]] == prefix]
            going = [x for x in still[t] if x["slot"] == g["slot"] and x["moves"][:len(prefix)] == prefix]
            assert len(ended) + len(going) >= 1, (t, g["slot"])
            for samples, moves, outcome in ended[:1]:
                # the samples of the first process lead the game's sample list, bit for bit, and every later move added one
                assert len(samples) == len(g["samples"]) + (len(moves) - len(prefix))
                assert all(bytes(samples[k]) == g["samples"][k][1] for k in range(len(g["samples"])))
                resumed += 1
    assert resumed >= 4   # (how many of the in-flight games END among the second process's 18 depends on the pacing: 7-12 seen; every one of them is checked above)


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
    # (--fixed-batch 1 --solve-deadline 0: the two clock-dependent pieces of the loops — sqrt batch sizes are not, but the replay below uses one
    #  size — are switched off for THIS comparison; test_search_thread_as_written_with_deadlines runs them as written)
    line, _ = run("thread", "--network", path, "--sims", sims, "--batch", batch, "--opening-seed", seed, "--table-entries", 1 << 16, "--plies", plies, "--async", asynchronous,
                  "--fixed-batch", 1, "--solve-deadline", 0)
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


@pytest.mark.parametrize("asynchronous", [0, 1])
def test_search_thread_as_written_with_deadlines(network_file, agx_lib, asynchronous):
    """player/SearchThread.cpp:84-199 AS WRITTEN: the lock scopes on Tree::low_priority_lock (utils/PriorityMutex.hpp), get_batch_size =
    sqrt(simulations) through Search::setBatchSize, and in asynchronous_run the estimated end of the network launch
    (NNEvaluator::asyncEvaluateGraphLaunch's return value) as the deadline of Search::solve (Search.cpp:159-183: node limit 10 000, the time
    left shared out over the leaves).  The results depend on the clock, as in the reference; what must hold: a legal game whose every search
    reaches its simulation count (or a proven root), and with sqrt batches more iterations than the fixed-batch run needs."""
    path, d, blob = network_file
    seed, sims, batch, plies, n = 11, 150, 8, 10, 15
    line, _ = run("thread", "--network", path, "--sims", sims, "--batch", batch, "--opening-seed", seed, "--table-entries", 1 << 16, "--plies", plies, "--async", asynchronous)
    fixed, _ = run("thread", "--network", path, "--sims", sims, "--batch", batch, "--opening-seed", seed, "--table-entries", 1 << 16, "--plies", plies, "--async", asynchronous,
                   "--fixed-batch", 1, "--solve-deadline", 0)
    assert len(line["moves"]) == plies or line["outcome"] != 0
    opening = synthetic.make_openings(n, 1, seed0=seed)[0]
    board = np.zeros(n * n, np.uint8)
    for m in opening:
        board[((m >> 2) & 127) * n + ((m >> 9) & 127)] = m & 3
    sign = 3 - (opening[-1] & 3) if opening else 1
    for mv in line["moves"]:
        cell = ((mv >> 2) & 127) * n + ((mv >> 9) & 127)
        assert board[cell] == 0 and (mv & 3) == sign
        board[cell] = mv & 3
        sign = 3 - sign
    assert all(v >= sims for v in line["root_visits"][:-1]) or line["outcome"] != 0
    # batch sizes 1, 2, 3, ... up to 8 while the simulation count grows: more, smaller iterations than 8 leaves at a time
    assert line["iterations"] > fixed["iterations"]


def test_tree_accessors_of_the_reference(network_file, agx_lib):
    """Tree::getMovesLeft / getMaximumDepth / hasAllMovesProven / hasSingleMove / hasSingleNonLosingMove / clear / clearNodeCacheStats and the
    priority locks (Tree.hpp:70,79-87,100-103), read after every search of a short game (boundary_main.cpp, mode tree)"""
    path, _, _ = network_file
    line, _ = run("tree", "--network", path, "--sims", 120, "--batch", 4, "--opening-seed", 5, "--plies", 6)
    assert line["searches"] == 6
    assert all(1 <= d_ <= 60 for d_ in line["max_depth"])          # every search descended at least one level
    assert all(0.0 <= m <= 225.0 for m in line["moves_left"])
    assert line["single_move_matches_edges"] == 1 and line["all_proven_matches_edges"] == 1 and line["non_losing_matches_edges"] == 1
    assert line["nodes_after_clear"] == 0 and line["depth_after_set_board"] == 0
    assert line["high_priority_passed_low"] == 1                   # the PriorityMutex order: a waiting high-priority locker goes first


@pytest.mark.parametrize("rules,nodes", [(0, 1000), (2, 200), (3, 1000)])
def test_alpha_beta_search_object_matches_the_oracle(agx_lib, tmp_path, rules, nodes):
    """AlphaBetaSearch(const GameConfig&) with solve(SearchTask&) / increaseGeneration / setNodeLimit (AlphaBetaSearch.hpp:54-66): one solver
    object — one table, the reference's 4 Mi entries — over a sequence of positions, the table aged every fourth one; node counts, scores,
    marks, the actions in their order and the feature words against ONE oracle solver given the same sequence"""
    from alphagomoku_amd import selfplay
    olib = ol.load()
    n, hw = 15, 225
    rng = np.random.default_rng(60 + rules)
    boards, signs = [], []
    for g in range(24):
        b = np.zeros(hw, np.uint8)
        stones = int(rng.integers(4, 50))
        cells = set()
        while len(cells) < stones:
            r, c = int(np.clip(rng.normal(7, 2.5), 0, n - 1)), int(np.clip(rng.normal(7, 2.5), 0, n - 1))
            cells.add(r * n + c)
        for k, cell in enumerate(sorted(cells, key=lambda x: rng.random())):
            b[cell] = 1 + k % 2
        boards.append(b)
        signs.append(1 if stones % 2 == 0 else 2)
    boards.append(boards[3].copy())    # a position seen before: its table entries are one generation older now
    signs.append(signs[3])
    path = tmp_path / "positions.txt"
    path.write_text("".join("%d %s\n" % (s_, "".join(str(int(x)) for x in b)) for b, s_ in zip(boards, signs)))
    p = subprocess.run([BINARY, "solver", "--rules", str(rules), "--board", str(n), "--nodes", str(nodes), "--positions", str(path)], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(x) for x in p.stdout.splitlines() if x.startswith('{"index"')]
    assert len(lines) == len(boards) and "AlphaBetaSearch :" in p.stdout
    s = olib.ago_solver_create(rules, n, n, 4 * 1024 * 1024, selfplay.default_config().zobrist_seed, nodes)
    proven = searched = 0
    for g, (b, sign) in enumerate(zip(boards, signs)):
        if g % 4 == 3:
            olib.ago_solver_new_generation(s)
        feat = np.zeros(hw, np.uint32)
        mv = np.zeros(hw, np.uint16)
        sc = np.zeros(hw, np.uint16)
        fl, rs, nd = ctypes.c_int(), ctypes.c_uint16(), ctypes.c_int()
        k = olib.ago_solver_solve(s, ol.ptr(b), sign, ol.ptr(feat), ol.ptr(mv), ol.ptr(sc), ctypes.byref(fl), ctypes.byref(rs), ctypes.byref(nd))
        line = lines[g]
        assert line["nodes"] == nd.value and line["score"] == rs.value and line["processed"] == 1, (g, line["nodes"], nd.value)
        assert line["edges"] == [[int(mv[i]), int(sc[i])] for i in range(k)], g
        assert line["must_defend"] == (fl.value & 1), g
        is_proven = ((rs.value >> 13) & 3) != 2 and rs.value not in (0, 0xFFFF)
        assert line["recursively_solved"] == int(is_proven) and line["statically_solved"] == int(nd.value <= 1), g   # AlphaBetaSearch.cpp:131-134
        assert line["feature_sum"] == int((feat.astype(np.uint64) * np.arange(1, hw + 1, dtype=np.uint64)).sum()), g
        proven += int(is_proven)
        searched += int(nd.value > 1)
    olib.ago_solver_destroy(s)
    assert proven >= 2 and searched >= 8


def test_game_generators_with_the_reference_constructor(network_file, agx_lib):
    """GameGenerator(gameOptions, selfplayOptions, manager, evaluator) (selfplay/GameGenerator.hpp:54): generators of one game each, driven by
    the reference's generator-thread loop, hand their finished games to the manager's buffer"""
    path, _, _ = network_file
    line, _ = run("generator", "--network", path, "--generators", 3, "--games", 2, "--sims", 40)
    assert line["generators"] == 3 and line["games"] >= 2 and line["samples"] > line["games"]


def test_boundary_error_behaviour(agx_lib, tmp_path):
    line, _ = run("errors", "--out", tmp_path)
    assert line["caught"] == 63     # five of the reference's exceptions + (32) a checkpoint written before the first generate() loads again
