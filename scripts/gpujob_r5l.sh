#!/bin/bash
# round 5, job L: the 15x15 128-filter tower as twelve waves per workgroup (4 channel groups x 3 position groups of 5 rows, 168 registers: W12) against the final build's tower (NM)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
scripts/ab_nn15_run.sh NM w12 NM w12
cp alphagomoku_amd/libagx.so /tmp/libagx_keep2.so
cp alphagomoku_amd/libagx_w12.so alphagomoku_amd/libagx.so
AGX_NO_BUILD=1 timeout 900 python -m pytest tests/test_nn_gpu.py -x -q 2>&1 | tail -2
cp /tmp/libagx_keep2.so alphagomoku_amd/libagx.so
scripts/ab_engine_run.sh "--steps 300 --warmup 30 --age-steps 1500" NM w12 NM w12
} > gpurun_out/r5l_nn.txt 2>&1
cat gpurun_out/r5l_nn.txt
