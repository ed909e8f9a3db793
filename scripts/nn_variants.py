"""Single-plane vs two-plane network kernel: parity against the fp32 oracle and timing (15x15 and 20x20)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alphagomoku_amd import synthetic, lib, check
from alphagomoku_amd.networks import AGNetwork, DeviceBuffer
from oracle import nn_ref

def run(rows, blocks, filters, single):
    os.environ["AGX_NN_SINGLE_PLANE"] = "1" if single else "0"
    d = synthetic.net_desc(rows=rows, cols=rows, blocks=blocks, filters=filters)
    blob, _ = synthetic.make_weights(d)
    net = AGNetwork(d); net.loadWeights(blob)
    f = synthetic.random_features(8, rows, rows, seed=3)
    p, v = net.forward(f)
    pr, vr = nn_ref.forward(d, blob, f)
    err = (np.abs(p - pr).max(), np.abs(v - vr).max(), bool((p.argmax(1) == pr.argmax(1)).all()))
    B = 4096
    fb = synthetic.random_features(B, rows, rows, seed=5)
    df = DeviceBuffer(fb.nbytes); df.upload(fb)
    dp = DeviceBuffer(B * rows * rows * 4); dv = DeviceBuffer(B * 3 * 4)
    t = ctypes.c_void_p(); check(lib.agx_timer_create(ctypes.byref(t)))
    for _ in range(2):
        net.forwardDevice(df.ptr, B, dp.ptr, dv.ptr)
    check(lib.agx_device_synchronize())
    check(lib.agx_timer_start(t, None))
    for _ in range(3):
        net.forwardDevice(df.ptr, B, dp.ptr, dv.ptr)
    check(lib.agx_timer_stop(t, None))
    ms = ctypes.c_float(); check(lib.agx_timer_elapsed_ms(t, ctypes.byref(ms)))
    hw = rows * rows
    F = filters
    D = min(256, 2 * F)
    flops = 2 * hw * (25 * 32 * F + blocks * 2 * 9 * F * F + 9 * F * F + F + 4 * F) + 2 * 4 * hw * D + 6 * D
    per = ms.value / 3
    print("%dx%d %dx%d %s: err policy %.2e value %.2e argmax %s | %.3f ms / %d boards = %.0f TFLOP/s" % (
        rows, rows, blocks, filters, "single-plane" if single else "two-plane", err[0], err[1], err[2], per, B, B * flops / per / 1e9))
    net.close()

for cfg in [(15, 6, 128, False), (15, 6, 128, True), (15, 2, 64, True), (20, 2, 64, True), (20, 10, 128, True), (15, 10, 128, False), (15, 10, 128, True)]:
    run(*cfg)
