#!/bin/bash
# round 5, job A: parity of the snapshot-undo build (full library) + A/B of the search-launch variants on one box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_engine_gpu.py -x -q -k "pattern_state or solver_matches or speculative_solver or yielding or renju or alpha_beta" > gpurun_out/r5a_parity.log 2>&1
echo "parity: $(tail -1 gpurun_out/r5a_parity.log)"
scripts/ab_variants.sh "--steps 300 --warmup 30 --age-steps 1500" A0 Q0 S1 S2 > gpurun_out/r5a_ab.txt 2>&1
cat gpurun_out/r5a_ab.txt
scripts/ab_engine_run.sh "--steps 300 --warmup 30 --age-steps 1500" - A0 - A0 > gpurun_out/r5a_full_ab.txt 2>&1
cat gpurun_out/r5a_full_ab.txt
scripts/ab_engine_run.sh "--steps 200 --warmup 30 --age-steps 1000 --config C5" - A0 - A0 > gpurun_out/r5a_c5_ab.txt 2>&1
cat gpurun_out/r5a_c5_ab.txt
