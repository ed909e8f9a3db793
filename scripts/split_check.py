"""Dev script: the pool as two slices of 512 games on two streams that own disjoint halves of the chip (CU masks), half a step out of
phase — while one slice's tower (MFMA, power-limited) runs on its 128 CUs the other slice's solver runs on the other 128."""
import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from alphagomoku_amd import lib, check, synthetic, selfplay
from alphagomoku_amd.networks import AGNetwork

games = 1024
steps = int(os.environ.get("STEPS", "400"))
desc = synthetic.net_desc(blocks=6, filters=128)
blob, _ = synthetic.make_weights(desc)
net = AGNetwork(desc); net.loadWeights(blob)
openings = selfplay.pack_openings(synthetic.make_openings(15, games * 3, seed0=0))

STREAMS = {}

def mask_words(cus):
    words = [0] * 8
    for c in cus:
        words[c // 32] |= 1 << (c % 32)
    return (ctypes.c_uint32 * 8)(*words)

def run(mode):
    cfg = selfplay.default_config(n_games=games, max_batch_size=8, max_simulations=400, solver_yield_fraction=float(os.environ.get('YIELD', '0.75')), node_capacity=4096, edge_capacity=76800,
                                  arena_reserve=3.0, record_format=2, tss_table_entries=4 << 20)
    pool = selfplay.GeneratorPool(cfg)
    pool.begin(openings)
    check(lib.agx_device_synchronize())
    if mode == "single":
        check(lib.agx_net_set_launch_width(net._net, 0))
        for _ in range(30):
            pool.step(net)
        check(lib.agx_device_synchronize())
        s0 = pool.stats(); t0 = time.perf_counter()
        for _ in range(steps):
            pool.step(net)
        check(lib.agx_device_synchronize())
    else:
        layout = mode.split("+")[0]
        parts = {"interleaved": [[c for c in range(256) if c % 2 == 0], [c for c in range(256) if c % 2 == 1]],
                 "blocks": [list(range(0, 128)), list(range(128, 256))],
                 "xcdhalf": [[c for c in range(256) if c % 32 < 16], [c for c in range(256) if c % 32 >= 16]],
                 "quads": [list(range(64 * k, 64 * k + 64)) for k in range(4)],
                 "octs": [list(range(32 * k, 32 * k + 32)) for k in range(8)],
                 "quadsx": [[c for c in range(256) if (c % 32) // 8 == k] for k in range(4)],
                 "quadsp": [[c for c in range(256) if (c // 32) % 4 == k] for k in range(4)],
                 "unmasked": [None, None]}[layout]
        n = len(parts)
        key = layout
        if key not in STREAMS:
            made = []
            for h in parts:
                s = ctypes.c_void_p()
                if h is None:
                    check(lib.agx_stream_create(ctypes.byref(s)))
                else:
                    check(lib.agx_stream_create_with_cu_mask(ctypes.byref(s), mask_words(h), 8))
                made.append(s)
            STREAMS[key] = made
        streams = STREAMS[key]
        check(lib.agx_net_set_launch_width(net._net, 256 // n))
        def step(g):
            pool.step_group(net, g, n, streams[g])
        for _ in range(30):
            for g in range(n):
                step(g)
        check(lib.agx_device_synchronize())
        if "offset" in mode:   # slice g falls g / n of a step behind: its first launches wait behind extra tower launches of its own
            for g in range(1, n):
                for _ in range(g if n > 2 else 1):
                    pool.evaluate_group(net, g, n, streams[g])   # (re-evaluates its last batch: same outputs)
        s0 = pool.stats(); t0 = time.perf_counter()
        for _ in range(steps):
            for g in range(n):
                step(g)
        check(lib.agx_device_synchronize())
        # (masked streams are never destroyed here: hipStreamDestroy of a CU-masked stream hung the process on this ROCm build)
    dt = time.perf_counter() - t0
    s1 = pool.stats()
    print("%-22s %.2f ms/step, %.0f simulations/s, errors %d" % (mode, 1e3 * dt / steps, (s1["evaluated_nodes"] - s0["evaluated_nodes"]) / dt, s1["first_error"]), flush=True)
    pool.close()

for mode in os.environ.get("MODES", "quads,quads+offset,quadsx+offset,quadsp+offset").split(","):
    run(mode)
