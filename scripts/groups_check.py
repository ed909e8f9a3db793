"""Dev script: throughput of the pool when driven as N independent slices on N streams."""
import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from alphagomoku_amd import lib, check, synthetic, selfplay
from alphagomoku_amd.networks import AGNetwork

games = int(os.environ.get("GAMES", "1024"))
desc = synthetic.net_desc(blocks=6, filters=128)
blob, _ = synthetic.make_weights(desc)
nets = []
for _ in range(8):
    nn = AGNetwork(desc); nn.loadWeights(blob); nets.append(nn)   # one network object per slice: the single-plane kernel owns a residual scratch
openings = selfplay.pack_openings(synthetic.make_openings(15, games * 2, seed0=0))
for n_groups in [int(x) for x in os.environ.get("GROUPS", "1,2,4,8").split(",")]:
    pool = selfplay.GeneratorPool(selfplay.default_config(n_games=games, max_batch_size=8, max_simulations=400, solver_yield_fraction=float(os.environ.get('YIELD', '0.75'))))
    pool.begin(openings)
    streams = []
    for _ in range(n_groups):
        s = ctypes.c_void_p(); check(lib.agx_stream_create(ctypes.byref(s))); streams.append(s)
    check(lib.agx_device_synchronize())
    for _ in range(20):
        for g in range(n_groups):
            pool.step_group(nets[g], g, n_groups, streams[g])
    check(lib.agx_device_synchronize())
    s0 = pool.stats()
    t0 = time.perf_counter()
    steps = 60
    for _ in range(steps):
        for g in range(n_groups):
            pool.step_group(nets[g], g, n_groups, streams[g])
    check(lib.agx_device_synchronize())
    dt = time.perf_counter() - t0
    s1 = pool.stats()
    print("groups %d: %.2f ms/step, %.0f simulations/s, errors %d" % (n_groups, 1e3 * dt / steps, (s1["evaluated_nodes"] - s0["evaluated_nodes"]) / dt, s1["first_error"]), flush=True)
    for s in streams:
        check(lib.agx_stream_destroy(s))
    pool.close()
