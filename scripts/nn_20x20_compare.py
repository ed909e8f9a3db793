import sys, numpy as np
names = sys.argv[1:]
o = np.load("gpurun_out/nn20_oracle.npz")
for n in names:
    d = np.load("gpurun_out/nn20_%s.npz" % n)
    e = np.abs(d["p"] - o["p"]).reshape(-1, 20, 20)
    print(n, "vs oracle: max %.3e mean %.3e; per-row max" % (e.max(), e.mean()), np.round(e.max(axis=(0, 2)) * 1e3, 1), "per-col max", np.round(e.max(axis=(0, 1)) * 1e3, 1))
    print("   oracle policy: max %.3e mean %.3e argmax agree %d / %d" % (o["p"].max(), o["p"].mean(), int((d["p"].argmax(1) == o["p"].argmax(1)).sum()), len(o["p"])))
a, b = np.load("gpurun_out/nn20_%s.npz" % names[0]), np.load("gpurun_out/nn20_%s.npz" % names[1])
e = np.abs(a["p"] - b["p"]).reshape(-1, 20, 20)
print(names[0], "vs", names[1], ": max %.3e mean %.3e; per-row max" % (e.max(), e.mean()), np.round(e.max(axis=(0, 2)) * 1e3, 1), "per-col max", np.round(e.max(axis=(0, 1)) * 1e3, 1))
