#!/bin/bash
# usage (on the GPU box, through gpurun): bash scripts/gpujob.sh TAG 'command; command; ...'
# runs the commands from the repository root with their output in gpurun_out/TAG.txt (merged back by gpurun); nothing is copied over the shipped
# library — variant libraries are selected per process with AGX_LIB_PATH (scripts/nn_ab.py, alphagomoku_amd/_lib.py)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
tag="$1"; shift
out="gpurun_out/${tag}.txt"
: > "$out"
bash -c "$*" >> "$out" 2>&1
rc=$?
echo "== exit $rc" >> "$out"
tail -n 60 "$out"
exit $rc
