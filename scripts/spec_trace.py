import sys
import numpy as np
t = np.loadtxt(sys.argv[1], dtype=np.uint64).astype(np.int64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
sel = (t[:, 0] - t0) / 1e5; c0 = (t[:, 1] - t0) / 1e5; c1 = (t[:, 2] - t0) / 1e5
ok = t[:, 1] > 0
print("games", len(t), "with commit", ok.sum(), "(times relative to the first select done)")
for name, a in (("select done", sel), ("commit begin", c0[ok]), ("commit end", c1[ok]), ("commit dur", (c1 - c0)[ok])):
    print(name, "pct 50/75/90/95/99/100:", np.percentile(a, [50, 75, 90, 95, 99, 100]).round(2))
rr = t[:, 3] // 16; lv = t[:, 3] % 16
print("reruns hist", np.bincount(rr[ok]), "leaves mean", lv[ok].mean())
