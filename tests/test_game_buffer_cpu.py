"""Host side of the record sink's checkpoint functions (csrc/game_buffer.cpp), no GPU involved: GameDataBuffer::load of files in
GameDataBuffer::save's layout (GameDataBuffer.cpp:97-131; GeneratorManager::loadState, GeneratorManager.cpp:263-275) and the pending-sample
hand-over of games in flight (GameGenerator::save / load, GameGenerator.cpp:122-141)."""
import ctypes
import json
import struct
import zlib

import numpy as np
import pytest


def make_game(n_samples, moves, outcome, rows=15, cols=15, seed=0):
    """GameDataStorage::serialize, format 201 (GameDataStorage.cpp:217-250): u32 samples, the samples (16-byte header + 6 bytes per entry), u32 moves,
    u16 per move, int outcome, rows, cols"""
    rng = np.random.default_rng(seed)
    out = struct.pack("<I", n_samples)
    for k in range(n_samples):
        count = int(rng.integers(1, 9))
        out += struct.pack("<6HI", 1, 2, 3, 0x4FA0, len(moves) - n_samples + k, 0, count) + bytes(rng.integers(0, 255, 6 * count, dtype=np.uint8))
    out += struct.pack("<I", len(moves)) + b"".join(struct.pack("<H", m) for m in moves)
    out += struct.pack("<3i", outcome, rows, cols)
    return out


def write_buffer_file(path, games, compressed, rules="FREESTYLE", rows=15, cols=15):
    offsets, pos = [], 0
    for g in games:
        offsets.append(pos)
        pos += len(g)
    header = json.dumps({"format": 201, "config": {"rules": rules, "rows": rows, "cols": cols, "draw_after": rows * cols}, "offsets": offsets})
    raw = header.encode() + b"\n" + b"".join(games)
    path.write_bytes(zlib.compress(raw) if compressed else raw)


def stats_of(lib, buf):
    from alphagomoku_amd._lib import AgxGameBufferStats
    s = AgxGameBufferStats()
    assert lib.agx_game_buffer_stats(buf, ctypes.byref(s)) == 0
    return {n: getattr(s, n) for n, _ in s._fields_}


def game_bytes(lib, buf, index):
    size = ctypes.c_size_t()
    assert lib.agx_game_buffer_game(buf, index, None, 0, ctypes.byref(size)) == 0
    out = (ctypes.c_uint8 * size.value)()
    assert lib.agx_game_buffer_game(buf, index, out, size.value, ctypes.byref(size)) == 0
    return bytes(out)


@pytest.mark.parametrize("compressed", [False, True])
def test_buffer_load_appends_the_games_of_a_saved_file(agx_lib, tmp_path, compressed):
    lib = agx_lib
    cross, circle = 1 | (7 << 2) | (7 << 9), 2 | (7 << 2) | (8 << 9)
    games = [make_game(2, [cross, circle, cross + 4], 2, seed=1), make_game(1, [cross, circle], 1, seed=2), make_game(3, [cross, circle, cross + 8, circle + 8], 3, seed=3)]
    path = tmp_path / "buffer.bin"
    write_buffer_file(path, games, compressed)
    buf = ctypes.c_void_p()
    assert lib.agx_game_buffer_create(0, 15, 15, 225, ctypes.byref(buf)) == 0
    assert lib.agx_game_buffer_load(buf, str(path).encode()) == 0
    assert stats_of(lib, buf) == {"games": 3, "samples": 6, "cross_win": 1, "draws": 1, "circle_win": 1, "game_length": 9}
    assert [game_bytes(lib, buf, i) for i in range(3)] == games
    # save -> load into a second buffer: the same games; loading twice appends
    saved = tmp_path / "again.bin"
    assert lib.agx_game_buffer_save(buf, str(saved).encode(), 1) == 0
    second = ctypes.c_void_p()
    assert lib.agx_game_buffer_create(0, 15, 15, 225, ctypes.byref(second)) == 0
    assert lib.agx_game_buffer_load(second, str(saved).encode()) == 0 and lib.agx_game_buffer_load(second, str(saved).encode()) == 0
    assert stats_of(lib, second)["games"] == 6 and [game_bytes(lib, second, i) for i in range(6)] == games + games
    # refusals: another board, other rules, a truncated game
    other = ctypes.c_void_p()
    assert lib.agx_game_buffer_create(0, 20, 20, 400, ctypes.byref(other)) == 0
    assert lib.agx_game_buffer_load(other, str(path).encode()) != 0 and b"15x15" in lib.agx_last_error()
    renju = ctypes.c_void_p()
    assert lib.agx_game_buffer_create(2, 15, 15, 225, ctypes.byref(renju)) == 0
    assert lib.agx_game_buffer_load(renju, str(path).encode()) != 0
    broken = tmp_path / "broken.bin"
    write_buffer_file(broken, [games[0][:-5]], compressed)
    assert lib.agx_game_buffer_load(buf, str(broken).encode()) != 0 and stats_of(lib, buf)["games"] == 3
    assert lib.agx_game_buffer_load(buf, str(tmp_path / "missing.bin").encode()) != 0
    for b in (buf, second, other, renju):
        lib.agx_game_buffer_destroy(b)


def test_pending_samples_travel_between_engines(agx_lib):
    """the samples of a game in flight leave the buffer with take_pending (keyed by engine, slot, index) and come back under another engine's key"""
    lib = agx_lib
    buf = ctypes.c_void_p()
    assert lib.agx_game_buffer_create(0, 15, 15, 225, ctypes.byref(buf)) == 0
    old_engine, new_engine = ctypes.c_void_p(0x1000), ctypes.c_void_p(0x2000)     # (keys only: the functions never look behind them)
    sample = lambda k, count: struct.pack("<6HI", 1, 2, 3, 0x4FA0, k, 0, count) + bytes(range(6 * count))   # noqa: E731
    records = b"".join(struct.pack("<iI", 8 + k, len(sample(8 + k, 2 + k))) + sample(8 + k, 2 + k) for k in range(3))
    assert lib.agx_game_buffer_restore_pending(buf, old_engine, 5, 2, records, len(records)) == 0
    size = ctypes.c_size_t()
    assert lib.agx_game_buffer_take_pending(buf, old_engine, 5, 2, None, 0, ctypes.byref(size)) == 0 and size.value == len(records)
    out = (ctypes.c_uint8 * size.value)()
    assert lib.agx_game_buffer_take_pending(buf, old_engine, 5, 2, out, size.value, ctypes.byref(size)) == 0 and bytes(out) == records
    assert lib.agx_game_buffer_take_pending(buf, old_engine, 5, 2, None, 0, ctypes.byref(size)) == 0 and size.value == 0      # taken means gone
    assert lib.agx_game_buffer_restore_pending(buf, new_engine, 5, 0, records, len(records)) == 0
    assert lib.agx_game_buffer_take_pending(buf, new_engine, 5, 0, None, 0, ctypes.byref(size)) == 0 and size.value == len(records)
    assert lib.agx_game_buffer_take_pending(buf, new_engine, 4, 0, None, 0, ctypes.byref(size)) == 0 and size.value == 0
    assert lib.agx_game_buffer_forget_engine(buf, new_engine) == 0
    assert lib.agx_game_buffer_take_pending(buf, new_engine, 5, 0, None, 0, ctypes.byref(size)) == 0 and size.value == 0
    assert lib.agx_game_buffer_restore_pending(buf, new_engine, 1, 0, records[:-3], len(records) - 3) != 0     # a truncated record is refused
    lib.agx_game_buffer_destroy(buf)
