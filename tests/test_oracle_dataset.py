"""Self-play record sink, CPU side.

1. pins the oracle's LowFP formats (oracle/ag_dataset.hpp) against the REAL reference header utils/low_precision.hpp compiled into
   oracle/_ref/libagref.so: every code of the four formats of dataset/SearchDataStorage.cpp:22,161-164 and 200 k floats per format;
2. checks the product's host-side reader (agx_sample_v201_unpack, csrc/game_buffer.cpp + csrc/sample_v201.hpp) against the oracle's
   parse + storeTo on oracle-made samples, incl. the 20x20 ">= 255 cells" filler rule and empty / proven-only roots;
3. storeTo(loadFrom(x)) round trip: the entries come back on the cells they were taken from, within the quantisation steps.
The device quantiser is compared with the same oracle in tests/test_engine_gpu.py."""
import ctypes
import os

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FORMATS = {0: (1, 3, 2), 1: (0, 3, 5), 2: (0, 4, 4), 3: (0, 5, 11)}  # id -> (sign, exponent, mantissa bits)


@pytest.fixture(scope="module")
def oracle():
    lib = ol.load()
    lib.ago_lowfp_to_lowp.restype = ctypes.c_uint32
    lib.ago_lowfp_to_lowp.argtypes = [ctypes.c_int, ctypes.c_float]
    lib.ago_lowfp_to_fp32.restype = ctypes.c_float
    lib.ago_lowfp_to_fp32.argtypes = [ctypes.c_int, ctypes.c_uint32]
    lib.ago_lowfp_max.restype = ctypes.c_float
    lib.ago_int8_to_score.restype = ctypes.c_uint16
    return lib


@pytest.fixture(scope="module")
def ref():
    path = os.path.join(ROOT, "oracle", "_ref", "libagref.so")
    if not os.path.exists(path):
        if os.path.isdir("/root/reference/src"):
            import subprocess
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/libagref.so"])
        else:
            pytest.skip("reference library not built")
    lib = ctypes.CDLL(path)
    if not hasattr(lib, "ref_lowfp_to_lowp"):
        pytest.skip("prebuilt reference library predates the LowFP pins")
    lib.ref_lowfp_to_lowp.restype = ctypes.c_uint32
    lib.ref_lowfp_to_lowp.argtypes = [ctypes.c_int, ctypes.c_float]
    lib.ref_lowfp_to_fp32.restype = ctypes.c_float
    lib.ref_lowfp_to_fp32.argtypes = [ctypes.c_int, ctypes.c_uint32]
    lib.ref_lowfp_max.restype = ctypes.c_float
    return lib


def bits(x):
    return np.float32(x).view(np.uint32)


def test_lowfp_decode_matches_reference_for_every_code(oracle, ref):
    for fmt, (s, e, m) in FORMATS.items():
        for code in range(1 << (s + e + m)):
            assert bits(oracle.ago_lowfp_to_fp32(fmt, code)) == bits(ref.ref_lowfp_to_fp32(fmt, code)), (fmt, code)
        assert bits(oracle.ago_lowfp_max(fmt)) == bits(ref.ref_lowfp_max(fmt))


def test_lowfp_encode_matches_reference(oracle, ref):
    rng = np.random.default_rng(201)
    for fmt, (s, e, m) in FORMATS.items():
        top = float(ref.ref_lowfp_max(fmt))
        samples = np.concatenate([
            rng.random(60000, dtype=np.float32) * np.float32(top * 1.1),              # the working range, a little beyond the maximum
            np.exp(rng.uniform(-30.0, np.log(top * 4.0), 60000)).astype(np.float32),  # every exponent, subnormal range included
            np.array([0.0, top, np.nextafter(np.float32(top), np.float32(0)), np.nextafter(np.float32(top), np.float32(1e9)), 1.0, 0.5,
                      2.0 ** -8, 2.0 ** -9, 2.0 ** -16, 2.0 ** -17, 1e-30], dtype=np.float32),
            # exact representable values and the midpoints between neighbours (rounding direction)
            np.array([ref.ref_lowfp_to_fp32(fmt, c) for c in range(1 << (e + m))], dtype=np.float32),
            np.array([0.5 * (ref.ref_lowfp_to_fp32(fmt, c) + ref.ref_lowfp_to_fp32(fmt, c + 1)) for c in range((1 << (e + m)) - 1)], dtype=np.float32),
        ])
        if s:
            samples = np.concatenate([samples, -samples])
        for x in samples[:200000]:
            assert oracle.ago_lowfp_to_lowp(fmt, float(x)) == ref.ref_lowfp_to_lowp(fmt, float(x)), (fmt, float(x))


def test_score_codes(oracle):
    """score_to_int8 / int8_to_score (SearchDataStorage.cpp:24-50): proven scores keep value and distance (clamped to 63), unproven ones
    keep their evaluation to the score format's precision"""
    for n in [0, 1, 5, 63, 64, 200]:
        for pv, make in [(0, lambda k: (0 << 13) | (4000 + k)), (1, lambda k: (1 << 13) | (4000 + k)), (3, lambda k: (3 << 13) | (4000 - k))]:
            code = oracle.ago_score_to_int8(make(n))
            assert code == (pv << 6) | min(n, 63)
            assert oracle.ago_int8_to_score(code) == make(min(n, 63))
    for ev in [-1000, -300, -17, -1, 0, 1, 12, 250, 999, 1000]:
        code = oracle.ago_score_to_int8((2 << 13) | (4000 + ev))
        assert code >> 6 == 2
        back = oracle.ago_int8_to_score(code)
        assert back >> 13 == 2
        got = (back & 8191) - 4000
        assert abs(got - ev) <= max(4, abs(ev) * 0.15), (ev, got)   # 2 mantissa bits
    assert oracle.ago_score_to_int8(0x0000) >> 6 == 0 and oracle.ago_score_to_int8(0xFFFF) >> 6 == 3   # infinities are not "proven"


def random_root(rng, n, n_edges, visited_fraction=0.6, proven_fraction=0.1):
    hw = n * n
    cells = rng.choice(hw, size=n_edges, replace=False)
    moves = np.array([1 | ((c // n) << 2) | ((c % n) << 9) for c in cells], np.uint16)
    visits = np.where(rng.random(n_edges) < visited_fraction, rng.integers(1, 900, n_edges), 0).astype(np.int32)
    prior = rng.random(n_edges).astype(np.float32)
    prior /= max(prior.sum(), np.float32(1e-9))
    value = rng.random((n_edges, 2)).astype(np.float32) * np.float32(0.5)
    score = np.full(n_edges, (2 << 13) | 4000, np.uint16)
    for i in range(n_edges):
        r = rng.random()
        if r < proven_fraction:
            k = int(rng.integers(1, 70))
            score[i] = [(0 << 13) | (4000 + k), (1 << 13) | (4000 + k), (3 << 13) | (4000 - k)][int(rng.integers(0, 3))]
        elif r < 0.5:
            score[i] = (2 << 13) | (4000 + int(rng.integers(-1000, 1001)))
    return moves, visits, prior.astype(np.float32), value, score


def pack(oracle, n, stones, root, root_score=(2 << 13) | 4000, flags=0):
    moves, visits, prior, value, score = root
    out = np.zeros(16 + 6 * n * n + 64, np.uint8)
    k = oracle.ago_sample_v201_pack(n, n, stones, len(moves), ol.ptr(moves), ol.ptr(visits), ol.ptr(prior), ol.ptr(np.ascontiguousarray(value)), ol.ptr(score),
                                    root_score, flags, ol.ptr(out), out.size)
    assert k >= 16
    return out[:k].copy()


def unpack_oracle(oracle, sample, n):
    hw = n * n
    visits, prior, value, score = np.zeros(hw, np.int32), np.zeros(hw, np.float32), np.zeros((hw, 2), np.float32), np.zeros(hw, np.uint16)
    header, mm = np.zeros(3, np.int32), np.zeros(2, np.float32)
    used = oracle.ago_sample_v201_unpack(ol.ptr(sample), n, n, ol.ptr(visits), ol.ptr(prior), ol.ptr(value), ol.ptr(score), ol.ptr(header), ol.ptr(mm))
    return used, visits, prior, value, score, header, mm


def unpack_product(lib, sample, n):
    from alphagomoku_amd import check
    hw = n * n
    visits, prior, value, score = np.zeros(hw, np.int32), np.zeros(hw, np.float32), np.zeros((hw, 2), np.float32), np.zeros(hw, np.uint16)
    header, mm = np.zeros(3, np.int32), np.zeros(2, np.float32)
    used = ctypes.c_size_t()
    check(lib.agx_sample_v201_unpack(ol.ptr(sample), sample.size, n, n, ol.ptr(visits), ol.ptr(prior), ol.ptr(value), ol.ptr(score), ol.ptr(header), ol.ptr(mm),
                                     ctypes.byref(used)))
    return used.value, visits, prior, value, score, header, mm


@pytest.mark.parametrize("n", [15, 20])
def test_product_reader_matches_oracle_store_to(oracle, agx_lib, n):
    rng = np.random.default_rng(7 + n)
    for trial in range(300):
        n_edges = int(rng.integers(0, min(n * n, 140)))
        root = random_root(rng, n, n_edges, visited_fraction=float(rng.random()), proven_fraction=float(rng.random()) * 0.3)
        rs = int([(2 << 13) | 4000, (3 << 13) | 3995, (0 << 13) | 4006, (1 << 13) | 4010][trial % 4])
        sample = pack(oracle, n, int(rng.integers(0, 40)), root, root_score=rs, flags=trial % 8)
        a = unpack_oracle(oracle, sample, n)
        b = unpack_product(agx_lib, sample, n)
        assert a[0] == b[0] == sample.size
        for x, y in zip(a[1:], b[1:]):
            assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), trial


def test_filler_entries_on_a_20x20_board(oracle):
    """an entry is forced on any cell 255 or more cells past the previous entry (loadFrom :333-337), whether or not it holds an edge"""
    n = 20
    mk = lambda cell: 1 | ((cell // n) << 2) | ((cell % n) << 9)  # noqa: E731
    # one visited edge on cell 300, one unvisited edge on cell 255 (the filler lands exactly on it and must carry its prior)
    moves = np.array([mk(300), mk(255)], np.uint16)
    visits = np.array([7, 0], np.int32)
    prior = np.array([0.25, 0.75], np.float32)
    value = np.array([[0.5, 0.1], [0.2, 0.3]], np.float32)
    score = np.array([(2 << 13) | 4000] * 2, np.uint16)
    sample = pack(oracle, n, 3, (moves, visits, prior, value, score))
    count = int(sample[12:16].view(np.uint32)[0])
    entries = sample[16:].reshape(-1, 6)
    assert count == 2 and list(entries[:, 0]) == [255, 45]
    assert entries[0, 1] == 0 and entries[0, 2] > 0            # filler: no visits, but the edge's prior
    _, v, p, q, s, header, _ = unpack_oracle(oracle, sample, n)
    assert v[300] == 7 and v[255] == 0 and p[255] > 0.7 and header[1] == 3
    # nothing at all: two fillers (255, 399 is only 144 further) -> exactly one entry
    empty = pack(oracle, n, 0, (np.zeros(0, np.uint16), np.zeros(0, np.int32), np.zeros(0, np.float32), np.zeros((0, 2), np.float32), np.zeros(0, np.uint16)),
                 root_score=(3 << 13) | 3999)
    assert int(empty[12:16].view(np.uint32)[0]) == 1 and empty[16] == 255
    # 15x15 never needs one
    empty15 = pack(oracle, 15, 0, (np.zeros(0, np.uint16), np.zeros(0, np.int32), np.zeros(0, np.float32), np.zeros((0, 2), np.float32), np.zeros(0, np.uint16)),
                   root_score=(3 << 13) | 3999)
    assert empty15.size == 16
    _, v, p, q, s, header, mm = unpack_oracle(oracle, empty15, 15)
    assert not v.any() and header[0] == (3 << 13) | 3999 and mm[0] == 1.0    # no visits: the value comes from the proven score


@pytest.mark.parametrize("n", [15, 20])
def test_round_trip_keeps_cells_and_values_within_a_quantisation_step(oracle, n):
    rng = np.random.default_rng(99)
    for trial in range(100):
        root = random_root(rng, n, int(rng.integers(1, 120)))
        moves, visits, prior, value, score = root
        sample = pack(oracle, n, 5, root)
        _, v, p, q, s, header, mm = unpack_oracle(oracle, sample, n)
        cells = ((moves >> 2) & 127).astype(int) * n + ((moves >> 9) & 127).astype(int)
        kept = (visits > 0) | np.array([(x >> 13) != 2 and x not in (0, 0xFFFF) for x in score])
        vmax, pmax, qmax = max(1, visits.max()), prior.max(), value.max()
        for i, c in enumerate(cells):
            if not kept[i]:
                continue
            assert abs(int(v[c]) - int(visits[i])) <= max(1, 0.02 * vmax + 0.04 * visits[i])     # 5 mantissa bits
            assert abs(p[c] - prior[i]) <= 0.04 * max(prior[i], pmax / 16) + 1e-6                 # 4 mantissa bits
            assert abs(q[c, 0] - value[i, 0]) <= 0.04 * max(value[i, 0], qmax / 16) + 1e-6
        assert np.count_nonzero(v) <= kept.sum() and header[1] == 5
        # a second pass through the quantiser is a fixed point for the visited cells' codes
        again = pack(oracle, n, 5, (moves[kept], v[cells[kept]].astype(np.int32), p[cells[kept]], q[cells[kept]], s[cells[kept]]))
        assert int(again[12:16].view(np.uint32)[0]) >= np.count_nonzero(v[cells[kept]] > 0)


def test_game_storage_layout(oracle):
    """GameDataStorage::serialize (format 201) of an oracle game: sample count, samples, moves (opening included), outcome, rows, cols"""
    n = 15
    cfg = ol.default_search_config(max_batch_size=4, max_simulations=60, table_entries=1 << 12)
    h = oracle.ago_game_create(0, n, n, ctypes.byref(cfg))
    op = np.zeros(64, np.uint16)
    k = oracle.ago_prepare_opening(0, n, n, 31, ol.ptr(op))
    oracle.ago_game_begin(h, ol.ptr(op), k)
    feats = np.zeros((8, n * n), np.uint32)
    for _ in range(4000):
        m = oracle.ago_game_step_select(h, ol.ptr(feats), 8)
        pol, val = np.zeros((max(m, 1), n * n), np.float32), np.zeros((max(m, 1), 2), np.float32)
        oracle.ago_fake_eval(m, n * n, ol.ptr(feats), ol.ptr(pol), ol.ptr(val))
        oracle.ago_game_step_expand(h, ol.ptr(pol), ol.ptr(val))
        if oracle.ago_game_outcome(h) != 0:
            break
    assert oracle.ago_game_outcome(h) != 0
    buf = np.zeros(1 << 20, np.uint8)
    size = oracle.ago_game_storage_v201(h, ol.ptr(buf), buf.size)
    data = buf[:size]
    n_samples = int(data[0:4].view(np.uint32)[0])
    assert n_samples == oracle.ago_game_num_records(h) > 0
    off = 4
    one = np.zeros(4096, np.uint8)
    for i in range(n_samples):
        count = int(data[off + 12:off + 16].view(np.uint32)[0])
        length = 16 + 6 * count
        got = oracle.ago_game_record_v201(h, i, ol.ptr(one), one.size)
        assert got == length and np.array_equal(one[:got], data[off:off + length])
        assert int(data[off + 8:off + 10].view(np.uint16)[0]) == k + i          # move_number = stones on the board
        off += length
    n_moves = int(data[off:off + 4].view(np.uint32)[0])
    assert n_moves == k + n_samples
    moves = data[off + 4:off + 4 + 2 * n_moves].view(np.uint16)
    assert list(moves[:k]) == list(op[:k])
    off += 4 + 2 * n_moves
    tail = data[off:off + 12].view(np.int32)
    assert off + 12 == size and list(tail) == [oracle.ago_game_outcome(h), n, n]
    oracle.ago_game_destroy(h)
