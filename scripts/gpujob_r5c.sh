#!/bin/bash
# round 5, job C: search-launch variants (lane-parallel table probes, deferral when the queue has run dry), waves / yield sweeps, the fixed tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
PARITY_K="(speculative_solver_plays_the_same_games and (0-15-8 or 1-15-8)) or yielding_pool" scripts/ab_variants.sh "--steps 300 --warmup 30 --age-steps 1500" S3 S8 S9 SA > gpurun_out/r5c_ab.txt 2>&1
cat gpurun_out/r5c_ab.txt
{
for w in 2304 2560 2816 3072; do echo "waves $w"; scripts/ab_engine_run.sh "--steps 300 --warmup 30 --age-steps 1500 --speculative-waves $w" S9 S8; done
for y in 0.4 0.8 0.95; do echo "yield $y"; scripts/ab_engine_run.sh "--steps 300 --warmup 30 --age-steps 1500 --yield-fraction $y" S9; done
} > gpurun_out/r5c_sweeps.txt 2>&1
cat gpurun_out/r5c_sweeps.txt
AGX_NO_BUILD=1 timeout 1200 python -m pytest tests/test_boundary_gpu.py -x -q -k "sigint or error_behaviour" > gpurun_out/r5c_boundary.log 2>&1; tail -3 gpurun_out/r5c_boundary.log
AGX_NO_BUILD=1 timeout 900 python -m pytest tests/test_nn_gpu.py -x -q -k "fp16_storage" 2>&1 | tail -1
