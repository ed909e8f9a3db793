"""Synthetic inputs for tests and the benchmark (there is no network access for checkpoints or datasets).

Weights: He-normal fp32, fixed seed; BatchNorm already folded (scale folded into the weights = 1, shift = small
normal bias), i.e. the state of the graph after AGNetwork::optimize(2) (src/networks/AGNetwork.cpp:136-160).
"""
import numpy as np


def net_desc(rows=15, cols=15, blocks=6, filters=128, in_channels=32, action_values=0):
    return dict(rows=rows, cols=cols, blocks=blocks, filters=filters, in_channels=in_channels,
                value_hidden=min(256, 2 * filters), action_values=action_values)


def make_weights(desc, seed=1234, residual_gain=1.0, policy_gain=1.0):
    """Returns (blob, parts): the canonical fp32 blob of include/agx.h and the list of named arrays in blob order.
    residual_gain scales the second convolution of every residual block: 1.0 is plain He-init, under which the un-normalised residual
    sums double their variance per block (a 10-block tower ends with near one-hot policies); a gain < 1 keeps activations of order one
    like a trained tower's (used by the deep-network tolerance test).
    policy_gain scales the policy head's final 1x1 convolution, i.e. the logits: under plain He-init (1.0) a 6x128 net gives ~170 of ~215 legal
    cells a prior above the expansion threshold (1e-4 of the total), so trees are six times wider than the reference's self-play trees; 2.5
    leaves ~30 — the edges per stored node a TRAINED network's peaked policy gives (22-67) — a "trained-like" tree shape."""
    rng = np.random.default_rng(seed)
    F, C, HW, D = desc["filters"], desc["in_channels"], desc["rows"] * desc["cols"], desc["value_hidden"]

    def he(shape, fan_in):
        return (rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)

    def shift(n):
        return (0.01 * rng.standard_normal(n)).astype(np.float32)

    parts = [("conv_in.w", he((5, 5, C, F), 25 * C)), ("conv_in.b", shift(F))]
    for i in range(desc["blocks"]):
        parts += [("block%d.w1" % i, he((3, 3, F, F), 9 * F)), ("block%d.b1" % i, shift(F)),
                  ("block%d.w2" % i, he((3, 3, F, F), 9 * F) * np.float32(residual_gain)), ("block%d.b2" % i, shift(F))]
    parts += [("policy.w1", he((3, 3, F, F), 9 * F)), ("policy.b1", shift(F)),
              ("policy.w2", he((F,), F) * np.float32(policy_gain)), ("policy.b2", shift(1))]
    parts += [("value.w1", he((F, 4), F)), ("value.b1", shift(4)),
              ("value.w2", he((HW * 4, D), HW * 4)), ("value.b2", shift(D)),
              ("value.w3", he((D, 3), D)), ("value.b3", shift(3))]
    if desc.get("action_values", 0):
        parts += [("q.w1", he((3, 3, F, F), 9 * F)), ("q.b1", shift(F)), ("q.w2", he((F, 3), F)), ("q.b2", shift(3))]
    blob = np.concatenate([p[1].reshape(-1) for p in parts]).astype(np.float32)
    return blob, parts


def random_features(batch, rows, cols, seed=0):
    """Random bit-packed feature words with plausible sparsity (stones/legal bits dense, threat bits sparse)."""
    rng = np.random.default_rng(seed)
    hw = rows * cols
    stone = rng.integers(0, 3, size=(batch, hw))  # 0 empty, 1 own, 2 opponent
    words = np.zeros((batch, hw), dtype=np.uint32)
    words |= (stone == 0).astype(np.uint32) << 0
    words |= (stone == 1).astype(np.uint32) << 1
    words |= (stone == 2).astype(np.uint32) << 2
    words |= np.uint32(1) << 3
    colour = rng.integers(0, 2, size=(batch, 1)).astype(np.uint32)
    words |= (colour << 4) | ((1 - colour) << 5)
    sparse = (rng.random((batch, hw, 24)) < 0.03)
    for bit in range(24):
        words |= (sparse[:, :, bit] & (stone == 0)).astype(np.uint32) << np.uint32(8 + bit)
    return words


def make_openings(n, count, seed0=0, rules=0):
    """`count` synthetic random openings from the library's restatement of the reference's prepareOpening
    (agx_make_opening, csrc/host_util.cpp); returns lists of Move::toShort words (cross first)."""
    import ctypes
    from ._lib import lib, check
    out = []
    buf = np.zeros(32, dtype=np.uint16)
    for i in range(count):
        check(lib.agx_make_opening(rules, n, seed0 + i, buf.ctypes.data_as(ctypes.c_void_p)))
        out.append([int(x) for x in buf[1:1 + int(buf[0])]])
    return out


def save_weights(path, desc, blob):
    """the weight container ag::AGNetwork::loadFromFile reads (include/alphagomoku_agx/networks.hpp): "AGXW", AgxNetDesc (7 ints), u64 count, fp32 blob"""
    import struct
    with open(path, "wb") as f:
        f.write(b"AGXW")
        f.write(struct.pack("<7i", desc["rows"], desc["cols"], desc["blocks"], desc["filters"], desc["in_channels"], desc["value_hidden"],
                            desc.get("action_values", 0)))
        f.write(struct.pack("<Q", blob.size))
        f.write(np.ascontiguousarray(blob, dtype=np.float32).tobytes())
