"""Dev script: stand-alone network launches of the in-loop size (6744 positions: 26.3 boards per CU) — direct and through a slot list."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alphagomoku_amd import synthetic, lib, check
from alphagomoku_amd.networks import AGNetwork, DeviceBuffer
d = synthetic.net_desc(blocks=6, filters=128)
blob, _ = synthetic.make_weights(d)
net = AGNetwork(d); net.loadWeights(blob)
slots = 8192
fb = synthetic.random_features(slots, 15, 15, seed=5)
df = DeviceBuffer(fb.nbytes); df.upload(fb)
dp = DeviceBuffer(slots * 225 * 4); dv = DeviceBuffer(slots * 3 * 4)
t = ctypes.c_void_p(); check(lib.agx_timer_create(ctypes.byref(t)))
F, C, HW, D, blocks = 128, 32, 225, 256, 6
flops = 2 * HW * (25 * C * F + blocks * 2 * 9 * F * F + 9 * F * F + F + 4 * F) + 2 * 4 * HW * D + 6 * D
for B in (8192, 6744, 6656, 6912):
    for _ in range(2):
        net.forwardDevice(df.ptr, B, dp.ptr, dv.ptr)
    check(lib.agx_device_synchronize())
    check(lib.agx_timer_start(t, None))
    for _ in range(10):
        net.forwardDevice(df.ptr, B, dp.ptr, dv.ptr)
    check(lib.agx_timer_stop(t, None))
    ms = ctypes.c_float(); check(lib.agx_timer_elapsed_ms(t, ctypes.byref(ms)))
    print("direct   B=%d: %.3f ms  %.0f TFLOP/s" % (B, ms.value / 10, B * flops / (ms.value / 10) / 1e9))
rng = np.random.default_rng(1)
lst = rng.permutation(slots)[:6744].astype(np.int32)
dl = DeviceBuffer(lst.nbytes); dl.upload(lst)
cnt = np.array([6744], np.int32); dc = DeviceBuffer(4); dc.upload(cnt)
for _ in range(2):
    check(lib.agx_nn_forward_indirect(net._net, df.ptr, dl.ptr, dc.ptr, slots, dp.ptr, dv.ptr, None))
check(lib.agx_device_synchronize())
check(lib.agx_timer_start(t, None))
for _ in range(10):
    check(lib.agx_nn_forward_indirect(net._net, df.ptr, dl.ptr, dc.ptr, slots, dp.ptr, dv.ptr, None))
check(lib.agx_timer_stop(t, None))
ms = ctypes.c_float(); check(lib.agx_timer_elapsed_ms(t, ctypes.byref(ms)))
print("indirect B=6744 (capacity 8192): %.3f ms  %.0f TFLOP/s" % (ms.value / 10, 6744 * flops / (ms.value / 10) / 1e9))
