#!/bin/bash
# round 5, job K: the 15x15 tower with the accumulators' initial values handed out inside the layer's first stage (NL) against the final build's tower (NM)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
scripts/ab_nn15_run.sh NM NL NM NL
cp alphagomoku_amd/libagx.so /tmp/libagx_keep2.so
cp alphagomoku_amd/libagx_NL.so alphagomoku_amd/libagx.so
AGX_NO_BUILD=1 timeout 900 python -m pytest tests/test_nn_gpu.py -x -q 2>&1 | tail -2
cp /tmp/libagx_keep2.so alphagomoku_amd/libagx.so
scripts/ab_engine_run.sh "--steps 300 --warmup 30 --age-steps 1500" NM NL NM NL
} > gpurun_out/r5k_nn.txt 2>&1
cat gpurun_out/r5k_nn.txt
