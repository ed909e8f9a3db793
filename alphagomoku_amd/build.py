"""Builds the in-tree HIP shared library (libagx.so) for gfx950 with hipcc.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libagx.so")

SOURCES = ["agx_api.hip", "nn_forward.hip", "engine.hip", "tables_host.cpp", "host_util.cpp", "game_buffer.cpp"]
DRIVER = os.path.join(HERE, "agx_selfplay")
AG_LIB = os.path.join(HERE, "libagx_ag.so")               # the reference-named C++ classes (include/alphagomoku_agx/) over the C ABI
BOUNDARY_TEST = os.path.join(HERE, "agx_boundary_test")  # tests/cpp/boundary_main.cpp: the reference's call chain on those classes


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for name in os.listdir(CSRC):
        if os.path.getmtime(os.path.join(CSRC, name)) > t:
            return True
    if os.path.getmtime(os.path.join(HERE, "..", "include", "agx.h")) > t:
        return True
    for extra in (AG_LIB, BOUNDARY_TEST, DRIVER):
        if not os.path.exists(extra):
            return True
    boundary = os.path.join(HERE, "..", "include", "alphagomoku_agx")
    newest = max([os.path.getmtime(os.path.join(boundary, f)) for f in os.listdir(boundary)]
                 + [os.path.getmtime(os.path.join(HERE, "..", "tests", "cpp", "boundary_main.cpp"))])
    if newest > os.path.getmtime(BOUNDARY_TEST):
        return True
    return False


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
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
        cmd += ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    # native C++ host driver over the C ABI (include/agx.hpp)
    cmd = [os.environ.get("CXX", "g++"), "-std=c++17", "-O2", "-o", DRIVER, os.path.join(CSRC, "selfplay_main.cpp"),
           "-L" + HERE, "-lagx", "-Wl,-rpath," + HERE, "-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    # the C++ boundary: plain host code (g++), no HIP types — a maintainer of the reference links it like any other library
    cxx = os.environ.get("CXX", "g++")
    cmd = [cxx, "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-o", AG_LIB, os.path.join(CSRC, "ag_classes.cpp"), "-L" + HERE, "-lagx",
           "-Wl,-rpath," + HERE, "-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    cmd = [cxx, "-std=c++17", "-O2", "-Wall", "-o", BOUNDARY_TEST, os.path.join(HERE, "..", "tests", "cpp", "boundary_main.cpp"), "-L" + HERE, "-lagx_ag",
           "-lagx", "-Wl,-rpath," + HERE, "-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
