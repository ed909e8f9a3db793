"""Timeline of ONE search launch from a profile build's per-wave trace (AGX_SPEC_TRACE=<file>: <file>.waves has one line per wave of the last
launches — select phase over, exit, leaves solved, ticks waited for items; 100 MHz ticks) -> how many waves are alive over the launch, where
the launch's time goes that the solves themselves do not explain (front, quantisation, tail).
usage: python scripts/spec_waves.py TRACE.waves [waves_per_group]"""
import sys
import numpy as np
t = np.loadtxt(sys.argv[1], dtype=np.uint64).astype(np.int64)
per = int(sys.argv[2]) if len(sys.argv) > 2 else 768
for g in range(len(t) // per):
    w = t[g * per:(g + 1) * per]
    w = w[w[:, 1] > 0]
    if len(w) == 0:
        continue
    t0 = w[:, 0].min() - 1      # (the first wave out of its select phase; the launch began a little earlier)
    sel = (w[:, 0] - t0) / 1e5
    ex = (w[:, 1] - t0) / 1e5
    print("group %d: %d waves, leaves %d (per wave: %s), launch end %.2f ms" % (g, len(w), w[:, 2].sum(), np.bincount(w[:, 2]).tolist(), ex.max()))
    print("  select phase over  pct 10/50/90/100: %s ms" % np.percentile(sel, [10, 50, 90, 100]).round(2))
    print("  wave exit          pct 10/25/50/75/90/100: %s ms" % np.percentile(ex, [10, 25, 50, 75, 90, 100]).round(2))
    print("  waited for items   mean %.2f ms per wave" % (w[:, 3].mean() / 1e5))
    grid = np.linspace(0, ex.max(), 21)
    alive = [(ex > x).sum() for x in grid]
    print("  waves alive at", " ".join("%.1f:%d" % (x, a) for x, a in zip(grid, alive)))
    busy = (ex - sel).sum() - w[:, 3].sum() / 1e5
    print("  wave-ms: alive %.0f, of it solving/committing %.0f; a perfectly packed launch of this work on %d waves = %.2f ms" % (ex.sum(), busy, len(w), busy / len(w)))
