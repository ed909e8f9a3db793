"""Dev script: why is the tower slower inside the pool step than stand-alone?  Times the evaluate stage right after k_solve, a second evaluate
of the same batch right behind it, and a third after an idle gap."""
import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alphagomoku_amd import synthetic, lib, check, selfplay
from alphagomoku_amd.networks import AGNetwork
desc = synthetic.net_desc(blocks=6, filters=128)
blob, _ = synthetic.make_weights(desc)
net = AGNetwork(desc); net.loadWeights(blob)
cfg = selfplay.default_config(rules=0, board_size=15, n_games=1024, max_batch_size=8, max_simulations=400, tss_table_entries=4 << 20,
                              solver_yield_fraction=0.75, node_capacity=4096, edge_capacity=76800, arena_reserve=3.0, record_format=2)
pool = selfplay.GeneratorPool(cfg)
pool.begin(selfplay.pack_openings(synthetic.make_openings(15, 3072, seed0=1, rules=0)))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 400):
    pool.step(net)
check(lib.agx_device_synchronize())
def timer():
    t = ctypes.c_void_p(); check(lib.agx_timer_create(ctypes.byref(t))); return t
def ms(t):
    v = ctypes.c_float(); check(lib.agx_timer_elapsed_ms(t, ctypes.byref(v))); return v.value
N = 40
ta, tb, tc, td = [timer() for _ in range(N)], [timer() for _ in range(N)], [timer() for _ in range(N)], [timer() for _ in range(N)]
counts = []
for i in range(N):
    pool.select_solve()
    check(lib.agx_timer_start(ta[i], None)); pool.evaluate(net); check(lib.agx_timer_stop(ta[i], None))
    check(lib.agx_timer_start(tb[i], None)); pool.evaluate(net); check(lib.agx_timer_stop(tb[i], None))
    check(lib.agx_timer_start(td[i], None)); pool.evaluate(net); check(lib.agx_timer_stop(td[i], None))
    check(lib.agx_device_synchronize())
    c = np.zeros(1, np.int32); check(lib.agx_memcpy_d2h(c.ctypes.data_as(ctypes.c_void_p), pool.buffers.d_nn_count, 4)); counts.append(int(c[0]))
    time.sleep(0.01)
    check(lib.agx_timer_start(tc[i], None)); pool.evaluate(net); check(lib.agx_timer_stop(tc[i], None))
    pool.expand_backup()
check(lib.agx_device_synchronize())
a, b, d, c = [np.array([ms(t) for t in ts]) for ts in (ta, tb, td, tc)]
n = np.array(counts)
print("positions per launch: mean %.0f min %d max %d" % (n.mean(), n.min(), n.max()))
for name, v in (("after k_solve", a), ("2nd back-to-back", b), ("3rd back-to-back", d), ("after 10 ms idle", c)):
    print("%-18s %.3f ms   %.1f us / (position per CU-round)  %.0f TFLOP/s" % (name, v.mean(), 1e3 * (v / np.ceil(n / 256)).mean(), (n * 909.45e6 / (v * 1e-3)).mean() / 1e12))
