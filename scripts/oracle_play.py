import sys, os, time, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import oracle_lib as ol
lib = ol.load()
rules = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 100
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n = 15
cfg = ol.default_search_config(max_batch_size=batch, max_simulations=sims, table_entries=1 << 20)
g = lib.ago_game_create(rules, n, n, ctypes.byref(cfg))
op = np.zeros(64, np.uint16)
k = lib.ago_prepare_opening(rules, n, n, 7, ol.ptr(op))
print("opening", [ol.short_to_move(int(x)) for x in op[:k]])
lib.ago_game_begin(g, ol.ptr(op), k)
feat = np.zeros((batch, n * n), np.uint32)
pol = np.zeros((batch, n * n), np.float32)
val = np.zeros((batch, 2), np.float32)
t0 = time.time()
steps = 0
while lib.ago_game_outcome(g) == 0:
    c = lib.ago_game_step_select(g, ol.ptr(feat), batch)
    lib.ago_fake_eval(c, n * n, ol.ptr(feat), ol.ptr(pol), ol.ptr(val))
    lib.ago_game_step_expand(g, ol.ptr(pol), ol.ptr(val))
    steps += 1
dt = time.time() - t0
st = np.zeros(11, np.uint64)
lib.ago_game_stats(g, ol.ptr(st))
names = "nodes nn_evals leaks duplicates proven wasted solver_nodes select_levels select_edges tree_nodes tree_edges".split()
print("outcome", lib.ago_game_outcome(g), "moves", lib.ago_game_num_records(g), "steps", steps, "time %.2fs" % dt)
print(dict(zip(names, [int(x) for x in st])))
print("nodes/s %.0f" % (st[0] / dt), "mean depth %.2f" % (st[7] / max(1, st[0] + st[2])), "mean edges/level %.1f" % (st[8] / max(1, st[7])))
# visit trace hash
import hashlib
h = hashlib.sha256()
for i in range(lib.ago_game_num_records(g)):
    mv = ctypes.c_uint16(); rv = ctypes.c_int(); rval = (ctypes.c_float * 2)(); rs = ctypes.c_uint16()
    em = np.zeros(512, np.uint16); ev = np.zeros(512, np.int32); ep = np.zeros(512, np.float32); evl = np.zeros(1024, np.float32); es = np.zeros(512, np.uint16)
    ne = lib.ago_game_record(g, i, ctypes.byref(mv), ctypes.byref(rv), rval, ctypes.byref(rs), ol.ptr(em), ol.ptr(ev), ol.ptr(ep), ol.ptr(evl), ol.ptr(es), 512)
    h.update(em[:ne].tobytes()); h.update(ev[:ne].tobytes()); h.update(ep[:ne].tobytes()); h.update(evl[:2 * ne].tobytes()); h.update(es[:ne].tobytes())
    h.update(bytes([mv.value & 255, mv.value >> 8]))
print("trace", h.hexdigest()[:16])
