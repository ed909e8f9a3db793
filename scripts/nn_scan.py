import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alphagomoku_amd import synthetic, lib, check
from alphagomoku_amd.networks import AGNetwork, DeviceBuffer
B = 8192
fb = synthetic.random_features(B, 15, 15, seed=5)
df = DeviceBuffer(fb.nbytes); df.upload(fb)
dp = DeviceBuffer(B * 225 * 4); dv = DeviceBuffer(B * 3 * 4)
t = ctypes.c_void_p(); check(lib.agx_timer_create(ctypes.byref(t)))
for blocks in [0, 1, 6, 12]:
    d = synthetic.net_desc(blocks=blocks, filters=128)
    blob, _ = synthetic.make_weights(d)
    net = AGNetwork(d); net.loadWeights(blob)
    for _ in range(2):
        net.forwardDevice(df.ptr, B, dp.ptr, dv.ptr)
    check(lib.agx_device_synchronize())
    check(lib.agx_timer_start(t, None))
    for _ in range(5):
        net.forwardDevice(df.ptr, B, dp.ptr, dv.ptr)
    check(lib.agx_timer_stop(t, None))
    ms = ctypes.c_float(); check(lib.agx_timer_elapsed_ms(t, ctypes.byref(ms)))
    print("blocks %2d: %.3f ms per 8192 boards = %.1f us per board per CU" % (blocks, ms.value / 5, ms.value / 5 / 32 * 1e3))
    net.close()
