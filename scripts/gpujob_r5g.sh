#!/bin/bash
# round 5, job G: no FLAT instruction left in the hot kernels — the solver's prefetches and snapshot through a global view of their pointers (SF against SG),
# the tower's pair-balance ticks as LDS instructions (NB against NC) and the in-place kernels' residual scratch through a global view (NB against N0)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
PARITY_K="(speculative_solver_plays_the_same_games and (0-15-8 or 1-15-8)) or yielding_pool" scripts/ab_variants.sh "--steps 300 --warmup 30 --age-steps 1500" SG SF > gpurun_out/r5g_ab.txt 2>&1
cat gpurun_out/r5g_ab.txt
{
scripts/ab_nn15_run.sh NC NB NC NB
scripts/ab_nn_run.sh N0 NB N0 NB
cp alphagomoku_amd/libagx.so /tmp/libagx_keep2.so
cp alphagomoku_amd/libagx_NB.so alphagomoku_amd/libagx.so
AGX_NO_BUILD=1 timeout 900 python -m pytest tests/test_nn_gpu.py -x -q 2>&1 | tail -2
cp /tmp/libagx_keep2.so alphagomoku_amd/libagx.so
scripts/ab_engine_run.sh "--steps 300 --warmup 30 --age-steps 1500" NC NB NC NB
scripts/ab_engine_run.sh "--config C4 --steps 200 --warmup 30 --age-steps 1000" N0 NB N0 NB
} > gpurun_out/r5g_nn.txt 2>&1
cat gpurun_out/r5g_nn.txt
