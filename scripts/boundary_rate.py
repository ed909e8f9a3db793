"""Throughput of the reference-shaped C++ path — GeneratorManager::generate with one GeneratorThread on device 0 (tests/cpp/boundary_main.cpp, mode
generate) — on BASELINE configs[1]'s shape: nb_node_count of printStats / the wall time of the whole call chain (set-up, opening generation and the
save at the end included, so it reads LOW for short runs).  usage: python scripts/boundary_rate.py [games to finish] [nn batch]"""
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alphagomoku_amd import synthetic

games = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
nn_batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
with tempfile.TemporaryDirectory() as tmp:
    d = synthetic.net_desc(blocks=6, filters=128)
    blob, _ = synthetic.make_weights(d)
    net = os.path.join(tmp, "net.agxw")
    synthetic.save_weights(net, d, blob)
    t0 = time.perf_counter()
    cpu0 = os.times()
    p = subprocess.run([os.path.join(ROOT, "alphagomoku_amd", "agx_boundary_test"), "generate", "--network", net, "--games", str(games), "--games-per-thread", "1024",
                        "--devices", "0", "--sims", "400", "--batch", "8", "--nn-batch", str(nn_batch), "--symmetries", "0", "--table-entries", str(4 * 1024 * 1024),
                        "--out", tmp], capture_output=True, text=True, timeout=1500)
    elapsed = time.perf_counter() - t0
    cpu1 = os.times()
    assert p.returncode == 0, p.stderr[-2000:]
    nodes = int(re.search(r"nb_node_count\s*=\s*(\d+)", p.stdout).group(1))
    line = [x for x in p.stdout.splitlines() if x.startswith('{"mode"')][0]
    print("games asked %d, nn batch %d: %.1f s wall, %d simulations -> %.0f simulations/s over the whole call chain; child CPU %.1f s user + %.1f s system; %s" % (
        games, nn_batch, elapsed, nodes, nodes / elapsed, cpu1.children_user - cpu0.children_user, cpu1.children_system - cpu0.children_system, line))
