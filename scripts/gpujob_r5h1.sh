#!/bin/bash
# round 5, final job 1: the full GPU suite of the final build, then the trace + PMC passes (hash-tied summary), the default line, the driver's window
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee gpurun_out/r05_gpu_test_suite_tail.txt
bash scripts/gpujob_pmc.sh 2>&1 | tail -12
