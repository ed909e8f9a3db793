#!/bin/bash
# round 5, job T: parking of straggling solves (K0 = 15x15 freestyle / standard, K1 = renju; AGX_QUICK builds of the working tree): parity of yielding pools, C2 / C5 lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
cp alphagomoku_amd/libagx.so /tmp/libagx_keep.so
run() {
  v=$1; shift
  cp alphagomoku_amd/libagx_$v.so alphagomoku_amd/libagx.so
  AGX_NO_BUILD=1 timeout 600 python bench.py --steps 300 --warmup 30 --age-steps 1500 --no-cpu-baseline "$@" 2>/tmp/bench_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v $*', '->', round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, d['speculative_solver'])" || tail -5 /tmp/bench_err.txt
}
{
cp alphagomoku_amd/libagx_K0.so alphagomoku_amd/libagx.so
AGX_NO_BUILD=1 timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -s -k "yielding_pool and (16-0 or 10-0 or 22-0 or 16-1)" 2>&1 | grep -E "parked|passed|failed|Error|error|assert" | head -20
cp alphagomoku_amd/libagx_K1.so alphagomoku_amd/libagx.so
AGX_NO_BUILD=1 timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -s -k "yielding_pool and (16-2 or 22-2)" 2>&1 | grep -E "parked|passed|failed|Error|error|assert" | head -20
run K1 --config C5
run K0
run K1 --config C5
run K0
cp alphagomoku_amd/libagx_K0.so alphagomoku_amd/libagx.so
AGX_NO_BUILD=1 timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "speculative_solver_plays_the_same_games and (0-15-8 or 1-15-8)" 2>&1 | tail -1
} > gpurun_out/r5t_park.txt 2>&1
cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
cat gpurun_out/r5t_park.txt
