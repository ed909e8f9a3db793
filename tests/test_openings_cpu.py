"""the product's opening generator (agx_make_opening, csrc/host_util.cpp, host-only) and the oracle's
(ago_prepare_opening, oracle/ag_mcts.cpp) are two restatements of prepareOpening (src/utils/misc.cpp:142-170) that consume
std::mt19937(seed) in the same order: they must agree stone for stone, incl. the reference's quirk that the distance map is not
cleared between rejected attempts (:144, generateOpeningMap accumulates on an empty board, :111-120)."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol


@pytest.mark.parametrize("rules", range(5))
def test_product_openings_equal_oracle_openings(agx_lib, rules):
    from alphagomoku_amd import check
    olib = ol.load()
    lengths = []
    for n in (15, 20):
        for seed in range(1000):
            a = np.zeros(64, np.uint16)
            b = np.zeros(32, np.uint16)
            k = olib.ago_prepare_opening(rules, n, n, seed, ol.ptr(a))
            check(agx_lib.agx_make_opening(rules, n, seed, b.ctypes.data_as(ctypes.c_void_p)))
            assert k == int(b[0]) and list(a[:k]) == list(b[1:1 + k]), (rules, n, seed)
            lengths.append(k)
    # the distribution of prepareOpening: max(1, U[0,6) + U[0,6) + U[0,6)) stones, one opening in a thousand empty
    assert 6.5 < np.mean(lengths) < 8.5 and max(lengths) <= 15 and min(lengths) >= 0
