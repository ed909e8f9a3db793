"""Registers, scratch, occupancy and LDS of every kernel of the shipped sources -> profiles/<tag>_kernel_resources.txt
(hipcc -Rpass-analysis=kernel-resource-usage with the flags of alphagomoku_amd/build.py; compiles here, no GPU needed).
usage: python scripts/kernel_resources.py r06"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alphagomoku_amd import build  # noqa: E402


def resources(src):
    cmd = build._compile_cmd(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), src, "/dev/null") + ["-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True).stderr
    rows, cur = [], {}
    for line in err.split("\n"):
        m = re.search(r"remark: [^:]*:\d+:\d+: +(Function Name|SGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            m = re.search(r"(Function Name|SGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        key, value = m.group(1), m.group(2)
        if key == "Function Name":
            cur = {"name": value}
            rows.append(cur)
        else:
            cur[key.split(" ")[0]] = value
    return rows


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    out = ["# hipcc -Rpass-analysis=kernel-resource-usage of the final build (%s): registers, scratch, LDS per kernel (scripts/kernel_resources.py)" % build.source_hash()]
    for src in ("nn_forward.hip", "engine.hip"):
        out.append("# " + src)
        for r in resources(src):
            out.append("%-120s sgpr %3s vgpr %3s scratch %4s B/lane occupancy %s waves/SIMD lds %6s B" % (r["name"], r.get("SGPRs", "?"), r.get("VGPRs", "?"),
                                                                                                       r.get("ScratchSize", "?"), r.get("Occupancy", "?"), r.get("LDS", "?")))
    path = os.path.join(ROOT, "profiles", tag + "_kernel_resources.txt")
    open(path, "w").write("\n".join(out) + "\n")
    print(path, len(out), "lines")


if __name__ == "__main__":
    main()
