"""Dev script for rocprofv3: a few forwards of one network shape (default 6x128, 15x15, 8192 boards)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphagomoku_amd import synthetic, lib, check
from alphagomoku_amd.networks import AGNetwork, DeviceBuffer
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 6
filters = int(sys.argv[2]) if len(sys.argv) > 2 else 128
n = int(sys.argv[3]) if len(sys.argv) > 3 else 15
B = 8192
fb = synthetic.random_features(B, n, n, seed=5)
df = DeviceBuffer(fb.nbytes); df.upload(fb)
dp = DeviceBuffer(B * n * n * 4); dv = DeviceBuffer(B * 3 * 4)
d = synthetic.net_desc(rows=n, cols=n, blocks=blocks, filters=filters)
blob, _ = synthetic.make_weights(d)
net = AGNetwork(d); net.loadWeights(blob)
for _ in range(6):
    net.forwardDevice(df.ptr, B, dp.ptr, dv.ptr)
check(lib.agx_device_synchronize())
if hasattr(lib._get(), "agx_debug_nn_profile"):
    import numpy as np
    out = np.zeros((2, 16), np.uint64)
    lib._get().agx_debug_nn_profile(out.ctypes.data_as(ctypes.c_void_p))   # discard warm-up
    for _ in range(4):
        net.forwardDevice(df.ptr, B, dp.ptr, dv.ptr)
    lib._get().agx_debug_nn_profile(out.ctypes.data_as(ctypes.c_void_p))
    names = ["stage input", "conv5x5+barrier", "conv3x3 preinit", "conv3x3 k-loop", "conv3x3 epilogue", "layer barrier", "value stage 1", "policy conv (rest)+barrier", "heads", "plane restore"]
    boards = 4 * B
    for row in range(2):
        total = out[row].sum()
        print("wave %d: %.0f cycles per board" % (4 * row, total / boards))
        for k, nm in enumerate(names):
            print("   %-28s %9.0f cycles/board  %5.1f %%" % (nm, out[row][k] / boards, 100.0 * out[row][k] / total))
