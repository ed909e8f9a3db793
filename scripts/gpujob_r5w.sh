#!/bin/bash
# round 5, job W: a resumed renju solve may be parked again (RP, -DAGX_REPARK=1) against the build's once-only rule (RQ); thresholds; renju parity
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
run RQ --config C5
run RP --config C5
AGX_PARK_FRACTION=0.86 run RP --config C5
AGX_PARK_FRACTION=0.93 run RP --config C5
run RQ --config C5
run RP --config C5
cp alphagomoku_amd/libagx_RP.so alphagomoku_amd/libagx.so
AGX_NO_BUILD=1 timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -s -k "yielding_pool and (16-2 or 22-2)" 2>&1 | grep -E "parked|passed|failed|Error|error|assert" | head -20
AGX_NO_BUILD=1 timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "speculative_solver_plays_the_same_games and 2-15" 2>&1 | tail -1
} > gpurun_out/r5w_repark.txt 2>&1
cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
cat gpurun_out/r5w_repark.txt
