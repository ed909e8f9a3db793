"""Dev script: the 20x20 10x128 tower (BASELINE configs[3]) stand-alone, whole chip: parity against the fp32 oracle on 8 boards and TFLOP/s on 4096."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alphagomoku_amd import synthetic, lib, check
from alphagomoku_amd.networks import AGNetwork, DeviceBuffer
from oracle import nn_ref
rows, blocks, filters = 20, 10, 128
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
for _ in range(5):
    net.forwardDevice(df.ptr, B, dp.ptr, dv.ptr)
check(lib.agx_timer_stop(t, None))
ms = ctypes.c_float(); check(lib.agx_timer_elapsed_ms(t, ctypes.byref(ms)))
hw, F, D = rows * rows, filters, min(256, 2 * filters)
flops = 2 * hw * (25 * 32 * F + blocks * 2 * 9 * F * F + 9 * F * F + F + 4 * F) + 2 * 4 * hw * D + 6 * D
per = ms.value / 5
print("%s 20x20 10x128: err policy %.2e value %.2e argmax %s | %.3f ms / %d boards = %.0f TFLOP/s" % (os.environ.get("AGX_VARIANT", ""), err[0], err[1], err[2], per, B, B * flops / per / 1e9), flush=True)
