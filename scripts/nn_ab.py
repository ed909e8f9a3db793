"""Dev script: A/B of network-kernel variants on one box, interleaved rounds, one process per (variant, round).

usage: python scripts/nn_ab.py [--rounds R] [--configs 15x6,15x10,20x10] [--width W] base NAME ...
A variant NAME is alphagomoku_amd/libagx_NAME.so (scripts/ab_nn_build.sh), `base` the shipped libagx.so; the child process loads it through
AGX_LIB_PATH (no file is copied over the shipped library).  Per variant and config: max error of policy / value against the fp32 oracle on 8
boards, ms per launch and algorithmic TFLOP/s of back-to-back launches of the in-loop size.  --width W narrows the persistent grid to W
workgroups (a slice's launch: 64)."""
import sys, os, json, subprocess, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {"15x6": (15, 6, 128, 6744), "15x10": (15, 10, 128, 6744), "20x10": (20, 10, 128, 4096), "15x2x64": (15, 2, 64, 6744)}


def child(configs, width, launches):
    import ctypes
    import numpy as np
    from alphagomoku_amd import synthetic, lib, check
    from alphagomoku_amd.networks import AGNetwork, DeviceBuffer
    from oracle import nn_ref
    out = {}
    for name in configs:
        rows, blocks, filters, B = CONFIGS[name]
        if width:
            B = max(width, B * width // 256)
        d = synthetic.net_desc(rows=rows, cols=rows, blocks=blocks, filters=filters)
        blob, _ = synthetic.make_weights(d)
        net = AGNetwork(d)
        net.loadWeights(blob)
        f = synthetic.random_features(8, rows, rows, seed=3)
        p, v = net.forward(f)
        pr, vr = nn_ref.forward(d, blob, f)
        err = (float(np.abs(p - pr).max()), float(np.abs(v - vr).max()), bool((p.argmax(1) == pr.argmax(1)).all()))
        if width:
            check(lib.agx_net_set_launch_width(net._net, width))
        fb = synthetic.random_features(B, rows, rows, seed=5)
        df = DeviceBuffer(fb.nbytes)
        df.upload(fb)
        dp = DeviceBuffer(B * rows * rows * 4)
        dv = DeviceBuffer(B * 3 * 4)
        t = ctypes.c_void_p()
        check(lib.agx_timer_create(ctypes.byref(t)))
        for _ in range(3):
            net.forwardDevice(df.ptr, B, dp.ptr, dv.ptr)
        check(lib.agx_device_synchronize())
        check(lib.agx_timer_start(t, None))
        for _ in range(launches):
            net.forwardDevice(df.ptr, B, dp.ptr, dv.ptr)
        check(lib.agx_timer_stop(t, None))
        ms = ctypes.c_float()
        check(lib.agx_timer_elapsed_ms(t, ctypes.byref(ms)))
        hw, F, D = rows * rows, filters, min(256, 2 * filters)
        flops = 2 * hw * (25 * 32 * F + blocks * 2 * 9 * F * F + 9 * F * F + F + 4 * F) + 2 * 4 * hw * D + 6 * D
        per = ms.value / launches
        out[name] = {"err_policy": err[0], "err_value": err[1], "argmax": err[2], "ms": per, "boards": B, "tflops": B * flops / per / 1e9}
        net.close()
    print("NNAB " + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="*")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--configs", default="15x6,20x10")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--launches", type=int, default=10)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    configs = a.configs.split(",")
    if a.child:
        return child(configs, a.width, a.launches)
    results = {v: {c: [] for c in configs} for v in a.variants}
    for r in range(a.rounds):
        for v in a.variants:
            env = dict(os.environ)
            env["AGX_NO_BUILD"] = "1"
            if v != "base":
                env["AGX_LIB_PATH"] = os.path.join(ROOT, "alphagomoku_amd", "libagx_%s.so" % v)
            try:
                res = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--configs", a.configs, "--width", str(a.width), "--launches", str(a.launches)],
                                     env=env, capture_output=True, text=True, timeout=int(os.environ.get("NN_AB_CHILD_TIMEOUT", "600")))
            except subprocess.TimeoutExpired:
                print("%s round %d: TIMEOUT" % (v, r), flush=True)
                continue
            line = [l for l in res.stdout.splitlines() if l.startswith("NNAB ")]
            if not line:
                print("%s round %d: FAILED\n%s" % (v, r, (res.stdout + res.stderr)[-1500:]), flush=True)
                continue
            d = json.loads(line[0][5:])
            for c in configs:
                results[v][c].append(d[c])
                print("round %d %-10s %-6s %8.3f ms %7.0f TFLOP/s  err p %.2e v %.2e argmax %s" % (r, v, c, d[c]["ms"], d[c]["tflops"], d[c]["err_policy"], d[c]["err_value"], d[c]["argmax"]), flush=True)
    print("== best of %d rounds (TFLOP/s; relative to the first variant)" % a.rounds)
    for c in configs:
        ref = None
        for v in a.variants:
            runs = results[v][c]
            if not runs:
                continue
            best = max(x["tflops"] for x in runs)
            ref = ref or best
            print("%-6s %-10s %7.0f  %+.1f %%" % (c, v, best, 100.0 * (best / ref - 1.0)))


if __name__ == "__main__":
    main()
