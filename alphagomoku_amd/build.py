"""Builds the in-tree HIP shared library (libagx.so) for gfx950 with hipcc.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libagx.so")

INCLUDE = os.path.join(HERE, "..", "include")
SOURCES = ["agx_api.hip", "nn_forward.hip", "engine.hip", "tables_host.cpp", "host_util.cpp", "game_buffer.cpp"]
DRIVER = os.path.join(HERE, "agx_selfplay")
AG_LIB = os.path.join(HERE, "libagx_ag.so")               # the reference-named C++ classes (include/alphagomoku_agx/) over the C ABI
BOUNDARY_TEST = os.path.join(HERE, "agx_boundary_test")  # tests/cpp/boundary_main.cpp: the reference's call chain on those classes


BOUNDARY_SRC = os.path.join(HERE, "..", "tests", "cpp", "boundary_main.cpp")
HOST_ONLY = ("ag_classes.cpp", "selfplay_main.cpp")  # plain g++ sources in csrc/ (not part of libagx.so)


BUILD_ID_SRC = os.path.join(CSRC, "build_id.cpp")   # agx_build_hash(): the source hash below, compiled into libagx.so
BUILD_ID_INC = os.path.join(CSRC, "build_id.inc")   # generated: the hash as a string literal (git-ignored)
BUILD_ID_OBJ = os.path.join(CSRC, "build_id.o")


def source_hash():
    """sha256 over the sources libagx.so is built from (csrc/*.hip, *.hpp, *.cpp and include/agx.h).  The library carries the hash of the
    sources it was compiled from (agx_build_hash()); bench.py, the test suite and committed PMC summaries compare against it, so a stale
    prebuilt binary cannot pass for the current sources."""
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".hpp", ".cpp")):
            h.update(name.encode())
            h.update(open(os.path.join(CSRC, name), "rb").read())
    h.update(open(os.path.join(INCLUDE, "agx.h"), "rb").read())
    return h.hexdigest()[:16]


def _recorded_hash():
    if not os.path.exists(BUILD_ID_INC) or not os.path.exists(BUILD_ID_OBJ):
        return None
    return open(BUILD_ID_INC).read().strip().strip('"')


def _mtime(path):
    return os.path.getmtime(path) if os.path.exists(path) else 0.0


def _device_headers():
    """every header a libagx.so object may include: csrc/*.hpp and the C ABI"""
    return [os.path.join(CSRC, n) for n in os.listdir(CSRC) if n.endswith(".hpp")] + [os.path.join(INCLUDE, "agx.h")]


def _stale_objects():
    newest_header = max(_mtime(h) for h in _device_headers())
    stale = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.rsplit(".", 1)[0] + ".o")
        if _mtime(obj) < max(_mtime(os.path.join(CSRC, src)), newest_header):
            stale.append(src)
    return stale


def _boundary_headers():
    d = os.path.join(INCLUDE, "alphagomoku_agx")
    return [os.path.join(d, f) for f in os.listdir(d)] + [os.path.join(INCLUDE, "agx.h"), os.path.join(INCLUDE, "agx.hpp")]


def _stale_host_targets():
    """[(target, newest input)] of the host-side binaries that are older than their inputs"""
    headers = max(_mtime(h) for h in _boundary_headers())
    out = []
    if _mtime(DRIVER) < max(_mtime(os.path.join(CSRC, "selfplay_main.cpp")), headers):
        out.append(DRIVER)
    if _mtime(AG_LIB) < max(_mtime(os.path.join(CSRC, "ag_classes.cpp")), headers):
        out.append(AG_LIB)
    if _mtime(BOUNDARY_TEST) < max(_mtime(BOUNDARY_SRC), headers, _mtime(AG_LIB)) or AG_LIB in out:
        out.append(BOUNDARY_TEST)
    return out


def needs_build():
    objs = [os.path.join(CSRC, src.rsplit(".", 1)[0] + ".o") for src in SOURCES]
    if _stale_objects() or _mtime(LIB) < max(_mtime(o) for o in objs) or _recorded_hash() != source_hash():
        return True
    return bool(_stale_host_targets())


def _compile_cmd(hipcc, src, obj):
    # -ffp-contract=off: the tree kernels must round exactly like the CPU oracle (no fused multiply-add)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]
    if os.environ.get("AGX_SOLVER_PROFILE"):
        cmd += ["-DAGX_SOLVER_PROFILE"]
    if src.endswith(".hip"):
        # the iterative-ILP machine scheduler beats the default on every kernel here (A/B on one box): k_solve 7.99 -> 7.72 ms,
        # 20x20 network kernels +5 % (10x128) / +32 % (2x64), 15x15 network kernels unchanged (they carry an iglp_opt hint)
        cmd += ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
    if src.endswith(".cpp"):
        cmd += ["-x", "hip"]
    return cmd + ["-c", os.path.join(CSRC, src), "-o", obj]


def build(force=False, verbose=True):
    """Incremental: an object is recompiled when its source or any device header is newer, libagx.so is linked when an object is newer,
    the host-side binaries (driver, reference-named classes, boundary test) when their sources / headers are."""
    if os.environ.get("AGX_NO_BUILD"):   # developer A/B runs with prebuilt variant libraries selected by AGX_LIB_PATH (scripts/nn_ab.py)
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cxx = os.environ.get("CXX", "g++")

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    objs = [os.path.join(CSRC, src.rsplit(".", 1)[0] + ".o") for src in SOURCES]
    # build identity: whenever the source hash differs from the one recorded with the last link and no object looks stale by its mtime, the
    # mtimes cannot be trusted (they do not survive every way of copying a tree): everything is compiled again
    current = source_hash()
    stale = list(SOURCES) if force else _stale_objects()
    if not stale and _recorded_hash() != current:
        # (also when NO hash is recorded although objects exist — a tree copied with its .o files but without build_id.inc: the objects' origin is
        #  unknown, stamping them with the current hash would defeat require_current_build)
        stale = list(SOURCES)
    procs = []
    for src in stale:   # in parallel: engine.hip alone takes minutes
        obj = os.path.join(CSRC, src.rsplit(".", 1)[0] + ".o")
        cmd = _compile_cmd(hipcc, src, obj)
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed")
    relink = force or _mtime(LIB) < max(_mtime(o) for o in objs)
    if _recorded_hash() != current:
        with open(BUILD_ID_INC, "w") as f:
            f.write('"%s"\n' % current)
        run([cxx, "-std=c++17", "-O2", "-fPIC", "-c", BUILD_ID_SRC, "-o", BUILD_ID_OBJ])
        relink = True
    if relink:
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + [BUILD_ID_OBJ, "-lz"])
    stale = [DRIVER, AG_LIB, BOUNDARY_TEST] if force else _stale_host_targets()
    if DRIVER in stale:  # native C++ host driver over the C ABI (include/agx.hpp)
        run([cxx, "-std=c++17", "-O2", "-o", DRIVER, os.path.join(CSRC, "selfplay_main.cpp"), "-L" + HERE, "-lagx", "-Wl,-rpath," + HERE, "-lpthread"])
    if AG_LIB in stale:  # the C++ boundary: plain host code (g++), no HIP types — a maintainer of the reference links it like any other library
        run([cxx, "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-o", AG_LIB, os.path.join(CSRC, "ag_classes.cpp"), "-L" + HERE, "-lagx",
             "-Wl,-rpath," + HERE, "-lpthread"])
    if BOUNDARY_TEST in stale:
        run([cxx, "-std=c++17", "-O2", "-Wall", "-o", BOUNDARY_TEST, BOUNDARY_SRC, "-L" + HERE, "-lagx_ag", "-lagx", "-Wl,-rpath," + HERE, "-lpthread"])
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
