"""Dev script: policy / value of the 20x20 10x128 tower on 16 boards -> gpurun_out/nn20_<AGX_VARIANT>.npz (+ the fp32 oracle's once)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alphagomoku_amd import synthetic
from alphagomoku_amd.networks import AGNetwork
rows, blocks, filters = 20, int(os.environ.get("AGX_BLOCKS", "10")), 128
d = synthetic.net_desc(rows=rows, cols=rows, blocks=blocks, filters=filters)
blob, _ = synthetic.make_weights(d)
net = AGNetwork(d); net.loadWeights(blob)
f = synthetic.random_features(16, rows, rows, seed=3)
p, v = net.forward(f)
v_name = os.environ.get("AGX_VARIANT", "X")
np.savez("gpurun_out/nn20_%s.npz" % v_name, p=p, v=v)
if not os.path.exists("gpurun_out/nn20_oracle.npz"):
    from oracle import nn_ref
    pr, vr = nn_ref.forward(d, blob, f)
    np.savez("gpurun_out/nn20_oracle.npz", p=pr, v=vr)
print("dumped", v_name, p.shape)
