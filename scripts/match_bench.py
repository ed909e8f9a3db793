"""Throughput of evaluation matches on one GPU (not the headline bench): `pairs` pairs of players = 2 * pairs trees, two networks
of the bench architecture with different weights, each step = one launch per stage over both players' trees (agx_engine_step_match).
--slices N (default 4): the pairs are split over N engines, each stepped on a stream that owns 1 / N of the chip's compute units
(selfplay.chip_slices) — the same effect as the sliced self-play pool: the network launches are confined to a part of the chip at a time.
usage: python scripts/match_bench.py [--pairs 1024] [--steps 100] [--warmup 30] [--sims 400] [--slices 4]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--filters", type=int, default=128)
    ap.add_argument("--yield-fraction", type=float, default=0.6)
    ap.add_argument("--speculative", type=int, default=1, help="the leaves of a batch solved in parallel (AgxEngineConfig.speculative_solver)")
    ap.add_argument("--slices", type=int, default=4)
    args = ap.parse_args()
    from alphagomoku_amd import build
    build.build(verbose=False)
    from alphagomoku_amd import lib, check, synthetic, selfplay
    from alphagomoku_amd.networks import AGNetwork
    check(lib.agx_set_device(0))
    desc = synthetic.net_desc(blocks=args.blocks, filters=args.filters)
    nets = []
    for seed in (1234, 4321):
        net = AGNetwork(desc)
        net.loadWeights(synthetic.make_weights(desc, seed=seed)[0])
        nets.append(net)
    slices = args.slices if (args.slices > 1 and args.pairs % args.slices == 0) else 1
    streams, per = [None], 0
    if slices > 1:
        streams, per = selfplay.chip_slices(slices)
        for net in nets:
            check(lib.agx_net_set_launch_width(net._net, per))
    pools = []
    for k in range(slices):
        cfg = selfplay.default_config(rules=0, board_size=15, n_games=2 * args.pairs // slices, max_batch_size=8, max_simulations=args.sims,
                                      tss_table_entries=4 * 1024 * 1024, solver_yield_fraction=args.yield_fraction, match_mode=1,
                                      speculative_solver=args.speculative, speculative_waves=(12 * per if slices > 1 else 0))
        pool = selfplay.GeneratorPool(cfg)
        pool.begin(selfplay.pack_openings(synthetic.make_openings(15, args.pairs // slices * 3, seed0=7000 + 100000 * k, rules=0)))
        pools.append(pool)

    def step():
        for k, pool in enumerate(pools):
            pool.step_match(nets[0], nets[1], streams[k])

    def totals():
        out = {}
        for pool in pools:
            for key, value in pool.stats().items():
                out[key] = (out.get(key, 0) + value) if key != "first_error" else max(out.get(key, 0), value)
        return out

    for _ in range(args.warmup):
        step()
    check(lib.agx_device_synchronize())
    s0 = totals()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    check(lib.agx_device_synchronize())
    dt = time.perf_counter() - t0
    s1 = totals()
    if s1["first_error"] != 0:
        raise RuntimeError("device engine stopped with error code %d" % s1["first_error"])
    import numpy as np
    res = np.concatenate([pool.match_results() for pool in pools])
    print(json.dumps(dict(workload="evaluation matches, freestyle 15x15, %dx%d nets, %d playouts, %d pairs" % (args.blocks, args.filters, args.sims, args.pairs),
                          simulations_per_sec=(s1["evaluated_nodes"] - s0["evaluated_nodes"]) / dt, ms_per_step=1e3 * dt / args.steps,
                          moves_per_sec=(s1["moves_played"] - s0["moves_played"]) / dt, games_finished=int(res[:, 3].sum()),
                          first_player_score=[int(x) for x in res[:, :3].sum(0)], slices=slices)))


if __name__ == "__main__":
    main()
