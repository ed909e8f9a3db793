#!/bin/bash
# round 5, job Y: leaves queued as soon as their descent is over (E1) against queued when the game's batch is selected (E0 = the build); E1R renju; AGX_QUICK builds of the working tree
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
cp alphagomoku_amd/libagx.so /tmp/libagx_keep.so
run() {
  v=$1; shift
  cp alphagomoku_amd/libagx_$v.so alphagomoku_amd/libagx.so
  AGX_NO_BUILD=1 timeout 600 python bench.py --steps 400 --warmup 30 --age-steps 1500 --no-cpu-baseline "$@" 2>/tmp/bench_err.txt | python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
print('$v $*', '->', round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, d['speculative_solver'])" || tail -5 /tmp/bench_err.txt
}
{
cp alphagomoku_amd/libagx_E1.so alphagomoku_amd/libagx.so
AGX_NO_BUILD=1 timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "(speculative_solver_plays_the_same_games and (0-15-8 or 1-15-8)) or (yielding_pool and (16-0 or 10-0 or 22-0 or 16-1))" 2>&1 | tail -2
cp alphagomoku_amd/libagx_E1R.so alphagomoku_amd/libagx.so
AGX_NO_BUILD=1 timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "(speculative_solver_plays_the_same_games and 2-15) or (yielding_pool and (16-2 or 22-2))" 2>&1 | tail -2
run E0
run E1
run E0
run E1
run E1 --config C3
run E0 --config C3
run E1R --config C5
} > gpurun_out/r5y_early.txt 2>&1
cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
cat gpurun_out/r5y_early.txt
