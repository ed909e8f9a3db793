#!/bin/bash
# round 5, job F: the 20x20 tower with the residual-scratch addresses made each layer's own (AGX_NN_OPAQUE_SKIP): stand-alone rate, parity, C4 pool
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
scripts/ab_nn_run.sh N0 NA N0 NA
cp alphagomoku_amd/libagx.so /tmp/libagx_keep2.so
cp alphagomoku_amd/libagx_NA.so alphagomoku_amd/libagx.so
AGX_NO_BUILD=1 timeout 900 python -m pytest tests/test_nn_gpu.py -x -q 2>&1 | tail -2
cp /tmp/libagx_keep2.so alphagomoku_amd/libagx.so
scripts/ab_engine_run.sh "--config C4 --steps 200 --warmup 30 --age-steps 1000" N0 NA N0 NA
} > gpurun_out/r5f_nn20.txt 2>&1
cat gpurun_out/r5f_nn20.txt
