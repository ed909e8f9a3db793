"""Dev script: NN forward parity + timing on the GPU box."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alphagomoku_amd import synthetic, lib, check
from alphagomoku_amd.networks import AGNetwork, DeviceBuffer
from oracle import nn_ref

for blocks, filters in [(2, 64), (6, 128)]:
    d = synthetic.net_desc(blocks=blocks, filters=filters)
    blob, _ = synthetic.make_weights(d)
    net = AGNetwork(d)
    net.loadWeights(blob)
    f = synthetic.random_features(300, 15, 15, seed=3)
    p, v = net.forward(f)
    pr, vr = nn_ref.forward(d, blob, f[:16])
    print(blocks, filters, "policy maxabs", np.abs(p[:16] - pr).max(), "value maxabs", np.abs(v[:16] - vr).max(),
          "psum", p.sum(1)[:3], "argmax agree", (p[:16].argmax(1) == pr.argmax(1)).mean())
    # the grid-stride path (boards >= 256) must agree with the first pass of the same boards
    f2 = np.concatenate([f[:16], f[:284]])
    p2, v2 = net.forward(f2)
    print("  stride consistency", np.abs(p2[16:32] - p[:16]).max(), np.abs(p2[:16] - p[:16]).max())
    B = 8192
    fb = synthetic.random_features(B, 15, 15, seed=5)
    df = DeviceBuffer(fb.nbytes); df.upload(fb)
    dp = DeviceBuffer(B * 225 * 4); dv = DeviceBuffer(B * 3 * 4)
    t = ctypes.c_void_p(); check(lib.agx_timer_create(ctypes.byref(t)))
    for _ in range(2):
        net.forwardDevice(df.ptr, B, dp.ptr, dv.ptr)
    check(lib.agx_device_synchronize())
    check(lib.agx_timer_start(t, None))
    n = 5
    for _ in range(n):
        net.forwardDevice(df.ptr, B, dp.ptr, dv.ptr)
    check(lib.agx_timer_stop(t, None))
    ms = ctypes.c_float(); check(lib.agx_timer_elapsed_ms(t, ctypes.byref(ms)))
    F, C, HW, D = filters, 32, 225, d["value_hidden"]
    flops = 2 * HW * (25 * C * F + blocks * 2 * 9 * F * F + 9 * F * F + F + 4 * F) + 2 * 4 * HW * D + 6 * D
    per = ms.value / n
    print("  batch %d: %.3f ms/forward, %.0f pos/s, %.1f TFLOP/s" % (B, per, B / per * 1e3, B * flops / per / 1e9))
