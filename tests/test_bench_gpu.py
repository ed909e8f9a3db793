"""bench.py on the GPU box: the N > 1 flow (SURVEY row e — independent game pools, no collective) exercised on ONE GPU by putting both
ranks on device 0 (AGX_FORCE_DEVICE), and the line's contract fields."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--games", "128", "--sims", "100", "--steps", "60", "--warmup", "5", "--table-entries", "65536", "--no-cpu-baseline", "--yield-fraction", "0"]


def run_bench(extra, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=e, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [x for x in p.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]            # rank 0 prints ONE line
    return json.loads(lines[0])


def test_two_ranks_on_one_gpu_report_the_whole_job(agx_lib):
    one = run_bench(["--gpus", "1"] + SMALL)
    two = run_bench(["--gpus", "2"] + SMALL, env={"AGX_FORCE_DEVICE": "0"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "weak"
    ranks = two["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and all(r["simulations"] > 0 for r in ranks)
    assert ranks[0]["opening_seed_base"] != ranks[1]["opening_seed_base"]          # the pools play different games
    total = sum(r["simulations"] for r in ranks)
    assert abs(two["value"] * (two["ms_per_step"] * 1e-3 * two["steps"]) - total) <= 1e-6 * total   # value = all ranks' work / max time
    assert abs(ranks[0]["simulations"] - ranks[1]["simulations"]) < 0.25 * total   # same shape of work per rank
    # the same game pool as the single run on rank 0 (seed base 0): identical work, whatever the pacing
    assert one["ranks"][0]["simulations"] == ranks[0]["simulations"]
    for line in (one, two):
        assert line["roofline"]["bound"] == "mfma" and 0 < line["roofline"]["frac"] < 1
        assert line["roofline_solver"]["kernel"].startswith(("k_solve", "k_search_spec")) and line["longest_kernel"]["name"] in line["kernel_ms_per_step"]


def test_pool_rate_floor(agx_lib):
    """the default workload for 150 steps of an aged pool (round 6: the boxes reach 950-1000 k simulations/s there): a tripwire for regressions
    that leave every parity test green, not a benchmark"""
    line = run_bench(["--steps", "150", "--warmup", "20", "--age-steps", "1500", "--no-cpu-baseline"])
    assert line["config"]["games_per_gpu"] == 1024 and line["n_gpus"] == 1 and line["aged_steps"] == 1500
    assert line["value"] >= 800e3, line["value"]
    assert line["speculative_solver"]["enabled"] and line["speculative_solver"]["leaves_solved"] > 0
    assert line["slices"]["count"] == 4 and line["slices"]["cus_per_slice"] * 4 <= 256
    assert line["roofline"]["whole_chip_equivalent"] >= 800.0 and 0 < line["roofline"]["frac"] < 1, line["roofline"]
    assert line["peak_tree_per_game"]["arena_failures"] == 0


def test_eight_ranks_on_one_gpu(agx_lib):
    """BASELINE configs[2]'s launch shape — `bench.py --gpus 8 --config C3` = one rank per GPU, games sharded, no collective on the data path
    (GeneratorManager.cpp:146-152) — on ONE GPU: all eight ranks on device 0 with small pools.  No scaling is measured here (there is one
    GPU); what is checked is the whole-job line: eight ranks with disjoint openings, value = the ranks' simulations / the slowest rank's time,
    the host CPU seconds of every rank on the line."""
    line = run_bench(["--gpus", "8", "--config", "C3", "--games", "128", "--steps", "40", "--warmup", "5", "--age-steps", "0", "--table-entries", "65536",
                      "--no-cpu-baseline", "--slices", "1"], env={"AGX_FORCE_DEVICE": "0"})
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and "standard" in json.dumps(line["config"]).lower()
    ranks = line["ranks"]
    assert [r["rank"] for r in ranks] == list(range(8)) and all(r["device"] == 0 for r in ranks)
    assert len({r["opening_seed_base"] for r in ranks}) == 8
    total = sum(r["simulations"] for r in ranks)
    assert all(r["simulations"] > 0 for r in ranks)
    assert abs(line["value"] * (line["ms_per_step"] * 1e-3 * line["steps"]) - total) <= 1e-6 * total
    # host pacing (bench.py --host-pacing 2, the default): the launch loop naps on an event two steps behind instead of spinning on a full
    # launch queue — a rank needs a fraction of a CPU (0.12 measured for one rank per GPU), two whole ones without the pacing
    assert line["slices"]["host_steps_ahead"] == 2
    assert all(r["host_cpu_seconds"] > 0.0 and r["host_cpu_utilisation"] < 0.8 for r in ranks), [r["host_cpu_utilisation"] for r in ranks]
    # every rank reports what it holds of its GPU and how many host threads it runs
    assert all(r["device_bytes_allocated"] > (64 << 20) and 1 <= r["host_threads"] <= 64 for r in ranks), [(r["device_bytes_allocated"], r["host_threads"]) for r in ranks]
    # configs[2] at full size is 1024 games per GPU with the reference's 4 Mi-entry solver tables: ONE such rank's engine is created here (the
    # eight of a node sit on eight GPUs) — its footprint must fit a GPU's 288 GB with room for the network and the runtime
    from alphagomoku_amd import selfplay
    full = selfplay.GeneratorPool(selfplay.default_config(n_games=1024, rules=1, max_batch_size=8, max_simulations=800, tss_table_entries=4 * 1024 * 1024,
                                                          node_capacity=4096, edge_capacity=76800, speculative_solver=1))
    footprint = full.device_bytes()
    full.close()
    print("C3 per-rank device footprint at 1024 games: %.1f GB" % (footprint / 1e9))
    assert 64e9 < footprint < 0.6 * 288e9, footprint     # 64 GB of solver tables alone; well inside one GPU's HBM (8 ranks = 8 GPUs x 288 GB)


def test_host_pacer_sleeps_behind_the_stream(agx_lib):
    """agx_event_create_blocking / agx_event_synchronize and selfplay.HostPacer (agx.hpp: HostPacer, ag::GameGenerator::generate): a loop paced two
    steps behind its stream completes, waits on never-recorded events return at once, and the waiting thread does not spin"""
    import ctypes
    import time
    from alphagomoku_amd import selfplay, synthetic, lib, check
    from alphagomoku_amd.networks import AGNetwork
    ev = ctypes.c_void_p()
    check(lib.agx_event_create_blocking(ctypes.byref(ev)))
    check(lib.agx_event_synchronize(ev))          # never recorded: complete
    check(lib.agx_event_record(ev, None))
    check(lib.agx_event_synchronize(ev))
    check(lib.agx_event_destroy(ev))
    d = synthetic.net_desc(blocks=2, filters=64)
    blob, _ = synthetic.make_weights(d)
    net = AGNetwork(d)
    net.loadWeights(blob)
    pool = selfplay.GeneratorPool(selfplay.default_config(n_games=256, max_batch_size=8, max_simulations=200, tss_table_entries=1 << 16))
    pool.begin(selfplay.pack_openings(synthetic.make_openings(15, 512, seed0=77)))
    pacer = selfplay.HostPacer(2)
    wall0, cpu0 = time.perf_counter(), time.thread_time()
    for _ in range(300):
        pool.step(net)
        pacer.step(None)
    check(lib.agx_device_synchronize())
    wall, cpu = time.perf_counter() - wall0, time.thread_time() - cpu0
    st = pool.stats()
    assert st["first_error"] == 0 and st["evaluated_nodes"] > 300 * 256
    assert pacer.count == 300
    assert cpu < 0.6 * wall, (cpu, wall)          # the loop's thread naps while the device works (a spinning wait would be ~1.0)
    pool.close()
    net.close()
