#!/bin/bash
# round 5, job U: parking in the renju kernels only — the threshold (AGX_PARK_FRACTION) on C5, C2 with the other kernels untouched; AGX_QUICK builds of the working tree
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
cp alphagomoku_amd/libagx.so /tmp/libagx_keep.so
run() {
  v=$1; shift
  cp alphagomoku_amd/libagx_$v.so alphagomoku_amd/libagx.so
  AGX_NO_BUILD=1 timeout 600 python bench.py --steps 300 --warmup 30 --age-steps 1500 --no-cpu-baseline "$@" 2>/tmp/bench_err.txt | python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
print('$v $*', 'park', os.environ.get('AGX_PARK_FRACTION'), '->', round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, d['speculative_solver'])" || tail -5 /tmp/bench_err.txt
}
{
for f in 0.80 0.86 0.90 0.94 0.97; do AGX_PARK_FRACTION=$f run K1 --config C5; done
for f in 0.86 0.90; do AGX_PARK_FRACTION=$f run K1 --config C5 --yield-fraction 0.6; done
run K0
run K0
cp alphagomoku_amd/libagx_K1.so alphagomoku_amd/libagx.so
AGX_NO_BUILD=1 timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -s -k "yielding_pool and (16-2 or 22-2)" 2>&1 | grep -E "parked|passed|failed|Error|error|assert" | head -20
cp alphagomoku_amd/libagx_K0.so alphagomoku_amd/libagx.so
AGX_NO_BUILD=1 timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -s -k "yielding_pool and (16-0 or 16-1)" 2>&1 | grep -E "parked|passed|failed|Error|error|assert" | head -20
} > gpurun_out/r5u_park.txt 2>&1
cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
cat gpurun_out/r5u_park.txt
