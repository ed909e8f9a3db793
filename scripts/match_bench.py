"""Throughput of evaluation matches on one GPU (not the headline bench): `pairs` pairs of players = 2 * pairs trees, two networks
of the bench architecture with different weights, each step = first players' group then second players' group on one stream.
usage: python scripts/match_bench.py [--pairs 1024] [--steps 100] [--warmup 30] [--sims 400]"""
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
    ap.add_argument("--yield-fraction", type=float, default=0.75)
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
    cfg = selfplay.default_config(rules=0, board_size=15, n_games=2 * args.pairs, max_batch_size=8, max_simulations=args.sims,
                                  tss_table_entries=4 * 1024 * 1024, solver_yield_fraction=args.yield_fraction, match_mode=1)
    pool = selfplay.GeneratorPool(cfg)
    pool.begin(selfplay.pack_openings(synthetic.make_openings(15, args.pairs * 3, seed0=7000, rules=0)))
    for _ in range(args.warmup):
        pool.step_match(nets[0], nets[1])
    check(lib.agx_device_synchronize())
    s0 = pool.stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pool.step_match(nets[0], nets[1])
    check(lib.agx_device_synchronize())
    dt = time.perf_counter() - t0
    s1 = pool.stats()
    if s1["first_error"] != 0:
        raise RuntimeError("device engine stopped with error code %d" % s1["first_error"])
    res = pool.match_results()
    print(json.dumps(dict(workload="evaluation matches, freestyle 15x15, %dx%d nets, %d playouts, %d pairs" % (args.blocks, args.filters, args.sims, args.pairs),
                          simulations_per_sec=(s1["evaluated_nodes"] - s0["evaluated_nodes"]) / dt, ms_per_step=1e3 * dt / args.steps,
                          moves_per_sec=(s1["moves_played"] - s0["moves_played"]) / dt, games_finished=int(res[:, 3].sum()),
                          first_player_score=[int(x) for x in res[:, :3].sum(0)])))


if __name__ == "__main__":
    main()
