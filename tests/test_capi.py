"""CPU-side checks of the C ABI: the library builds, loads, and exports every symbol include/agx.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for fname in os.listdir(os.path.join(ROOT, "include")):
        if fname.endswith(".h"):
            text = open(os.path.join(ROOT, "include", fname)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            names.update(re.findall(r"\b(agx_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_header_declares_entry_points():
    names = declared_symbols()
    assert "agx_nn_forward" in names and "agx_net_load_weights" in names


def test_library_exports_every_declared_symbol(agx_lib):
    from alphagomoku_amd import _lib
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in declared_symbols() if not hasattr(cdll, n)]
    assert missing == []


def test_library_carries_the_hash_of_its_sources(agx_lib, monkeypatch):
    from alphagomoku_amd import _lib, build
    assert agx_lib.agx_build_hash().decode() == build.source_hash()
    monkeypatch.setattr(build, "source_hash", lambda: "0" * 16)
    try:
        _lib.require_current_build()
    except _lib.AgxError as e:
        assert "other sources" in str(e)
    else:
        raise AssertionError("a library built from other sources must be refused")


def test_blob_size_matches_layout(agx_lib):
    from alphagomoku_amd import synthetic, _lib
    for blocks, filters in [(2, 64), (6, 128), (10, 128)]:
        d = synthetic.net_desc(blocks=blocks, filters=filters)
        blob, _ = synthetic.make_weights(d, seed=1)
        cdesc = _lib.AgxNetDesc(d["rows"], d["cols"], d["blocks"], d["filters"], d["in_channels"], d["value_hidden"])
        assert agx_lib.agx_net_blob_floats(ctypes.byref(cdesc)) == blob.size


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from alphagomoku_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib._load()
    except _lib.AgxError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("expected AgxError")


def test_header_is_plain_c(tmp_path):
    """the boundary is a C ABI: include/agx.h must compile as C99 (no C++ types leak into the signatures)"""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "agx.h"\nint main(void) { AgxEngineConfig c; AgxNetDesc d; (void) c; (void) d; return 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), str(src)])
