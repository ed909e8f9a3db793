"""Row f4's figure: ONE game searched on ONE tree, SearchThread::serial_run against SearchThread::asynchronous_run
(player/SearchThread.cpp:121-180) on the device engine — how long N simulations take when the network's launch for buffer b overlaps
the expand / backup / select / solve of buffer 1 - b (AgxEngineConfig.search_buffers = 2) instead of running in between.

  serial : expand_backup(0) -> select_solve(0) -> network(0), one stream, one buffer (serial_run's loop on the same engine and kernels)
  async  : the search stages of buffer b on the search stream, the network of buffer b on a second stream behind an event; the search
           stream waits for buffer b's network only when b's turn comes again (asynchronous_run's Join / Launch / switchBuffer)
  pool   : the ordinary one-game self-play engine (select + speculative solve in one launch), for reference
each of serial / async with the buffer's leaves solved one after the other by one wave per search thread (k_solve) and in parallel
(speculative_solver: k_search_spec behind the one-wave select) — on one tree the threat solver, not the network, is most of an iteration.

usage: python scripts/tournament_latency.py [--threads 1 4] [--batch 8 16] [--iterations 600] [--blocks 6] [--filters 128]"""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, nargs="+", default=[1, 4])
    ap.add_argument("--batch", type=int, nargs="+", default=[8, 16])
    ap.add_argument("--iterations", type=int, default=600, help="loop iterations (task buffers expanded) per measurement")
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--filters", type=int, default=128)
    ap.add_argument("--rules", type=int, default=0)
    args = ap.parse_args()
    from alphagomoku_amd import build
    build.build(verbose=False)
    from alphagomoku_amd import lib, check, synthetic, selfplay
    from alphagomoku_amd.networks import AGNetwork
    check(lib.agx_set_device(0))
    desc = synthetic.net_desc(blocks=args.blocks, filters=args.filters)
    net = AGNetwork(desc)
    net.loadWeights(synthetic.make_weights(desc, seed=1234)[0])
    vp = ctypes.c_void_p
    streams = []
    for _ in range(2):
        s = vp()
        check(lib.agx_stream_create(ctypes.byref(s)))
        streams.append(s)
    search_stream, net_stream = streams
    opening = selfplay.pack_openings(synthetic.make_openings(15, 1, seed0=4242, rules=args.rules))

    def make_pool(threads, batch, buffers, speculative=1):
        cfg = selfplay.default_config(rules=args.rules, board_size=15, n_games=threads * buffers, search_threads=threads if (threads > 1 or buffers == 2) else 0,
                                      search_buffers=buffers, max_batch_size=batch, max_simulations=1 << 22,   # one long search: the move rule never fires
                                      tss_table_entries=1 << 20, node_capacity=1 << 18, edge_capacity=1 << 23, speculative_solver=speculative)
        pool = selfplay.GeneratorPool(cfg)
        pool.begin(opening)
        return pool

    def measure(kind, threads, batch, speculative):
        if kind == "pool":
            if threads > 1 or not speculative:
                return None
            pool = make_pool(1, batch, 1)
        else:
            pool = make_pool(threads, batch, 2, speculative)
        sched = [vp(), vp()]
        done = [vp(), vp()]
        for e in sched + done:
            check(lib.agx_event_create(ctypes.byref(e)))

        def iteration(i):
            if kind == "pool":
                pool.step(net, search_stream)
            elif kind == "serial":
                pool.expand_backup_group(0, 2, search_stream)
                pool.select_solve_group(0, 2, search_stream)
                pool.evaluate_group(net, 0, 2, search_stream)
            else:
                b = i % 2
                check(lib.agx_stream_wait_event(search_stream, done[b]))          # asyncEvaluateGraphJoin of this buffer's previous batch
                pool.expand_backup_group(b, 2, search_stream)
                pool.select_solve_group(b, 2, search_stream)
                check(lib.agx_event_record(sched[b], search_stream))
                check(lib.agx_stream_wait_event(net_stream, sched[b]))
                pool.evaluate_group(net, b, 2, net_stream)                        # asyncEvaluateGraphLaunch
                check(lib.agx_event_record(done[b], net_stream))

        for i in range(50):
            iteration(i)
        check(lib.agx_device_synchronize())
        s0 = pool.stats()
        t0 = time.perf_counter()
        for i in range(args.iterations):
            iteration(i)
        check(lib.agx_device_synchronize())
        dt = time.perf_counter() - t0
        s1 = pool.stats()
        if s1["first_error"] != 0:
            raise RuntimeError("device engine stopped with error code %d" % s1["first_error"])
        sims = s1["evaluated_nodes"] - s0["evaluated_nodes"]
        out = dict(kind=kind, threads=threads, batch=batch, parallel_leaves=speculative, iterations=args.iterations, us_per_iteration=1e6 * dt / args.iterations, simulations=int(sims),
                   simulations_per_sec=sims / dt, network_evaluations=int(s1["network_evaluations"] - s0["network_evaluations"]), moves=int(s1["moves_played"] - s0["moves_played"]),
                   tree_nodes=int(s1["peak_nodes"]))
        for e in sched + done:
            check(lib.agx_event_destroy(e))
        pool.close()
        return out

    rows = []
    for threads in args.threads:
        for batch in args.batch:
            got = {}
            for speculative in (0, 1):
                for kind in ("pool", "serial", "async"):
                    r = measure(kind, threads, batch, speculative)
                    if r is not None:
                        got[kind, speculative] = r
                        rows.append(r)
                        print(json.dumps(r), flush=True)
            rate = {k: v["simulations_per_sec"] for k, v in got.items()}
            print(json.dumps(dict(threads=threads, batch=batch, async_over_serial=rate["async", 0] / rate["serial", 0],
                                  async_over_serial_parallel_leaves=rate["async", 1] / rate["serial", 1],
                                  parallel_leaves_over_serial_leaves=rate["async", 1] / rate["async", 0])), flush=True)
    # the same two loops as compiled C++ against the reference-named classes (tests/cpp/boundary_main.cpp mode thread: SearchThread::serial_run /
    # asynchronous_run as written, with the stop condition read from the tree in every iteration)
    import subprocess
    import tempfile
    binary = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "alphagomoku_amd", "agx_boundary_test")
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "network.agxw")
        synthetic.save_weights(path, desc, synthetic.make_weights(desc, seed=1234)[0])
        for batch in args.batch:
            got = {}
            for asynchronous in (0, 1):
                p = subprocess.run([binary, "thread", "--network", path, "--sims", "4000", "--batch", str(batch), "--opening-seed", "4242", "--plies", "3", "--async", str(asynchronous),
                                    "--table-entries", str(1 << 20), "--nodes", str(1 << 16), "--edges", str(1 << 21)], capture_output=True, text=True, timeout=600)
                if p.returncode != 0:
                    raise RuntimeError(p.stderr[-2000:])
                line = json.loads([x for x in p.stdout.splitlines() if x.startswith('{"mode"')][0])
                r = dict(kind="SearchThread classes", asynchronous=asynchronous, batch=batch, iterations=line["iterations"], seconds=line["seconds"], simulations=line["simulations"],
                         us_per_iteration=1e6 * line["seconds"] / line["iterations"], simulations_per_sec=line["simulations"] / line["seconds"], moves=len(line["moves"]))
                got[asynchronous] = r
                print(json.dumps(r), flush=True)
            print(json.dumps(dict(kind="SearchThread classes", batch=batch, async_over_serial=got[1]["simulations_per_sec"] / got[0]["simulations_per_sec"])), flush=True)
    print(json.dumps(dict(workload="one game, one tree, %dx%d network, 15x15 rules %d" % (args.blocks, args.filters, args.rules), rows=len(rows))))


if __name__ == "__main__":
    main()
