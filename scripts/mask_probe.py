"""Dev script: does a CU-masked stream run our kernels at all?  (short, under `timeout`)"""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alphagomoku_amd import lib, check, synthetic, selfplay
from alphagomoku_amd.networks import AGNetwork, DeviceBuffer
def mask_words(cus):
    words = [0] * 8
    for c in cus:
        words[c // 32] |= 1 << (c % 32)
    return (ctypes.c_uint32 * 8)(*words)
desc = synthetic.net_desc(blocks=2, filters=64)
blob, _ = synthetic.make_weights(desc)
net = AGNetwork(desc); net.loadWeights(blob)
B = 2048
fb = synthetic.random_features(B, 15, 15, seed=5)
df = DeviceBuffer(fb.nbytes); df.upload(fb)
dp = DeviceBuffer(B * 225 * 4); dv = DeviceBuffer(B * 3 * 4)
keep = []
for name, cus in (("first 128", range(128)), ("second 128", range(128, 256))):
    s = ctypes.c_void_p()
    check(lib.agx_stream_create_with_cu_mask(ctypes.byref(s), mask_words(cus), 8))
    keep.append(s)
    print("created", name, flush=True)
for (name, s) in zip(("first 128", "second 128"), keep):
    check(lib.agx_net_set_launch_width(net._net, 0))
    t0 = time.perf_counter()
    for _ in range(5):
        net.forwardDevice(df.ptr, B, dp.ptr, dv.ptr, stream=s)
    check(lib.agx_stream_synchronize(s))
    print("mask %-10s tower x5 on the masked stream: %.2f ms" % (name, 1e3 * (time.perf_counter() - t0)), flush=True)
print("network ok", flush=True)
cfg = selfplay.default_config(n_games=64, max_batch_size=8, max_simulations=100, tss_table_entries=1 << 16, node_capacity=4096, edge_capacity=65536)
pool = selfplay.GeneratorPool(cfg)
pool.begin(selfplay.pack_openings(synthetic.make_openings(15, 256, seed0=0)))
streams = keep
for stage in ("select_solve_group", "evaluate_group", "expand_backup_group"):
    for g in range(2):
        if stage == "evaluate_group":
            pool.evaluate_group(net, g, 2, streams[g])
        else:
            getattr(pool, stage)(g, 2, streams[g])
        check(lib.agx_stream_synchronize(streams[g]))
        print(stage, g, "done", flush=True)
for i in range(20):
    for g in range(2):
        pool.step_group(net, g, 2, streams[g])
for s in streams:
    check(lib.agx_stream_synchronize(s))
print("20 steps of two masked slices ok:", pool.stats()["evaluated_nodes"], flush=True)
