"""The fp32 numpy network oracle against an independent torch-CPU implementation of the same layer definitions."""
import numpy as np
import torch
import torch.nn.functional as Fn

from alphagomoku_amd import synthetic
from oracle import nn_ref


def torch_forward(d, blob, f):
    it = iter(nn_ref.split_blob(d, blob))
    rows, cols = d["rows"], d["cols"]

    def conv(x, w, b):
        return Fn.conv2d(x, torch.from_numpy(w).permute(3, 2, 0, 1), torch.from_numpy(b), padding=w.shape[0] // 2)

    x = torch.from_numpy(nn_ref.unpack_input(f, rows, cols, d["in_channels"])).permute(0, 3, 1, 2)
    x = torch.relu(conv(x, next(it), next(it)))
    for _ in range(d["blocks"]):
        w1, b1, w2, b2 = next(it), next(it), next(it), next(it)
        x = torch.relu(x + conv(torch.relu(conv(x, w1, b1)), w2, b2))
    wp1, bp1, wp2, bp2 = next(it), next(it), next(it), next(it)
    p = torch.relu(conv(x, wp1, bp1)).permute(0, 2, 3, 1) @ torch.from_numpy(wp2) + float(bp2[0])
    policy = torch.softmax(p.reshape(p.shape[0], -1), 1)
    wv1, bv1, wv2, bv2, wv3, bv3 = [torch.from_numpy(next(it)) for _ in range(6)]
    v = torch.relu(x.permute(0, 2, 3, 1) @ wv1 + bv1).reshape(x.shape[0], -1)
    value = torch.softmax(torch.relu(v @ wv2 + bv2) @ wv3 + bv3, 1)
    return policy.numpy(), value.numpy()


def test_numpy_oracle_matches_torch():
    d = synthetic.net_desc(blocks=2, filters=64)
    blob, _ = synthetic.make_weights(d, seed=7)
    f = synthetic.random_features(3, 15, 15, seed=11)
    p, v = nn_ref.forward(d, blob, f)
    pt, vt = torch_forward(d, blob, f)
    assert np.abs(p - pt).max() < 1e-6
    assert np.abs(v - vt).max() < 1e-5
    assert np.allclose(p.sum(1), 1.0, atol=1e-5)


def test_unpack_input_bits():
    f = np.array([[0x80000001, 0x00000100]], dtype=np.uint32)
    x = nn_ref.unpack_input(np.tile(f, (1, 1)).repeat(1, 0)[:, :2].reshape(1, 2), 1, 2)
    assert x[0, 0, 0, 0] == 1 and x[0, 0, 0, 31] == 1 and x[0, 0, 0, 1:31].sum() == 0
    assert x[0, 0, 1, 8] == 1 and x[0, 0, 1].sum() == 1


def test_raw_input_oracle_matches_torch():
    """ResnetPVraw (networks.cpp:107-129): 8 input channels = bits 0-7 of the feature word"""
    d = synthetic.net_desc(blocks=2, filters=64, in_channels=8)
    blob, _ = synthetic.make_weights(d, seed=8)
    f = synthetic.random_features(3, 15, 15, seed=12)
    p, v = nn_ref.forward(d, blob, f)
    pt, vt = torch_forward(d, blob, f)
    assert np.abs(p - pt).max() < 1e-6 and np.abs(v - vt).max() < 1e-5
    p8, v8 = nn_ref.forward(d, blob, f & np.uint32(0xFF))
    assert np.array_equal(p, p8) and np.array_equal(v, v8)
    assert not np.array_equal(p, nn_ref.forward(d, blob, f ^ np.uint32(1))[0])


def test_fp16_storage_mode_rounds_where_the_kernel_does():
    """storage="fp16": weights of the MFMA layers and every activation plane are fp16 values, the result stays within the format's
    tolerance of the fp32 oracle and is not identical to it"""
    d = synthetic.net_desc(blocks=2, filters=64)
    blob, _ = synthetic.make_weights(d, seed=7)
    f = synthetic.random_features(4, 15, 15, seed=13)
    p32, v32 = nn_ref.forward(d, blob, f)
    p16, v16 = nn_ref.forward(d, blob, f, storage="fp16")
    assert not np.array_equal(p32, p16)
    assert np.abs(p32 - p16).max() < 4e-3 and np.abs(v32 - v16).max() < 4e-3
    # a blob whose values are already fp16-representable and all-zero biases: only the activations are rounded
    blob16 = blob.astype(np.float16).astype(np.float32)
    pa, va = nn_ref.forward(d, blob16, f, storage="fp16")
    pb, vb = nn_ref.forward(d, blob, f, storage="fp16")
    parts = nn_ref.split_blob(d, blob)
    # (policy 1x1 / last dense weights and biases stay fp32 in the kernel, so the two blobs may differ there)
    assert np.abs(pa - pb).max() < 1e-3 and np.abs(va - vb).max() < 1e-3 and len(parts) > 0


def test_whole_graph_fp16_mode():
    """storage="fp16_all" restates graph.convertTo(FLOAT16) over the WHOLE graph (AGNetwork.cpp:157): with it a blob and its fp16-rounded copy
    give IDENTICAL results (every parameter is rounded on entry, the kernel-format mode keeps the 1x1 head / last dense weights and the biases
    in fp32), and it stays within the format's tolerance of the kernel-format mode."""
    d = synthetic.net_desc(blocks=2, filters=64, action_values=1)
    blob, _ = synthetic.make_weights(d, seed=9)
    f = synthetic.random_features(3, 15, 15, seed=21)
    blob16 = blob.astype(np.float16).astype(np.float32)
    a = nn_ref.forward(d, blob, f, storage="fp16_all")
    b = nn_ref.forward(d, blob16, f, storage="fp16_all")
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    k = nn_ref.forward(d, blob, f, storage="fp16")
    assert not np.array_equal(a[0], k[0])
    assert all(np.abs(x - y).max() < 4e-3 for x, y in zip(a, k))
