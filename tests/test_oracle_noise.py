"""Root-noise generators of the oracle (oracle/ag_noise.hpp; the device computes the same series, tests/test_engine_gpu.py checks
that bit for bit): accuracy of the deterministic log / exp, and the distributions the reference draws from (utils/random.cpp:89-124)."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol


@pytest.fixture(scope="module")
def lib():
    lib = ol.load()
    lib.ago_det_log.restype = ctypes.c_double
    lib.ago_det_log.argtypes = [ctypes.c_double]
    lib.ago_det_exp.restype = ctypes.c_double
    lib.ago_det_exp.argtypes = [ctypes.c_double]
    lib.ago_root_noise.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return lib


def noise(lib, kind, weight, priors, serial=0, move=0, seed=12345):
    p = np.ascontiguousarray(priors, dtype=np.float32)
    out = np.zeros_like(p)
    lib.ago_root_noise(kind, weight, seed, serial, move, len(p), ol.ptr(p), ol.ptr(out))
    return out


def test_series_accuracy(lib):
    rng = np.random.default_rng(0)
    for x in np.concatenate([10.0 ** rng.uniform(-300, 300, 2000), rng.uniform(0.5, 2.0, 2000), [1.0, 2.0, 0.5, 1.1920929e-07]]):
        assert abs(lib.ago_det_log(float(x)) - np.log(x)) <= 4e-15 * max(1.0, abs(np.log(x)))
    for x in np.concatenate([rng.uniform(-600, 600, 2000), rng.uniform(-1, 1, 2000), [0.0]]):
        assert abs(lib.ago_det_exp(float(x)) - np.exp(x)) <= 1e-14 * np.exp(x)


@pytest.mark.parametrize("kind", [1, 2, 3])
def test_noise_is_a_distribution_and_reproducible(lib, kind):
    rng = np.random.default_rng(kind)
    for n in (1, 2, 7, 60, 225):
        priors = rng.random(n).astype(np.float32)
        priors /= priors.sum()
        a = noise(lib, kind, 0.25, priors, serial=3, move=n)
        assert np.isfinite(a).all() and (a >= 0).all() and float(a.sum()) < 1.0 + 1e-4
        assert kind == 1 or abs(float(a.sum()) - 1.0) < 1e-4     # the stick-breaking vector keeps a remainder, exactly like the reference's
        assert np.array_equal(a, noise(lib, kind, 0.25, priors, serial=3, move=n))
        if n > 2:
            assert not np.array_equal(a, noise(lib, kind, 0.25, priors, serial=4, move=n))
        if kind != 3:
            assert np.array_equal(noise(lib, kind, 0.0, priors), priors)          # weight 0 leaves the priors alone


def test_custom_noise_moments(lib):
    """u^4 stick breaking: E[first stick] = E[u^4] = 1/5; after the shuffle every position has the same mean 1/n"""
    n, runs = 20, 4000
    acc = np.zeros(n)
    top = []
    for r in range(runs):
        a = noise(lib, 1, 1.0, np.zeros(n, np.float32), serial=r)
        acc += a
        top.append(a.max())
    assert np.abs(acc / runs - acc.sum() / runs / n).max() < 0.01
    assert 0.15 < np.mean(top) < 0.6


def test_dirichlet_noise_moments(lib):
    """normalised Gamma(0.05) draws: Dirichlet(0.05): mean 1/n, variance (1/n)(1 - 1/n) / (n * 0.05 + 1)"""
    n, runs = 10, 6000
    samples = np.array([noise(lib, 2, 1.0, np.zeros(n, np.float32), serial=r) for r in range(runs)])
    assert np.abs(samples.mean(0) - 1.0 / n).max() < 0.015
    want = (1.0 / n) * (1 - 1.0 / n) / (n * 0.05 + 1)
    assert np.abs(samples.var(0) - want).max() < 0.15 * want
    assert np.abs(samples.sum(1) - 1).max() < 1e-5


def test_gumbel_noise_moments(lib):
    """softmax(log p + g) with standard Gumbel g picks arg max with probability p (Gumbel-max trick)"""
    p = np.array([0.5, 0.3, 0.15, 0.05], np.float32)
    runs = 8000
    wins = np.zeros(4)
    for r in range(runs):
        wins[int(np.argmax(noise(lib, 3, 1.0, p, serial=r)))] += 1
    assert np.abs(wins / runs - p).max() < 0.02
