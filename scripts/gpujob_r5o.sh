#!/bin/bash
# round 5, job O: the search launch with 16 solver waves per compute unit (10 240 B of LDS state: 448 action-stack entries + 30 frames in LDS; 128 registers,
# 4 waves per SIMD) so that a slice's ~1900 leaves are two rounds of solves instead of 2.47 -> 3 (W4), against the build's 12 per unit (Q3); both AGX_QUICK
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
cp alphagomoku_amd/libagx.so /tmp/libagx_keep.so
run() {
  v=$1; shift
  cp alphagomoku_amd/libagx_$v.so alphagomoku_amd/libagx.so
  AGX_NO_BUILD=1 python bench.py --steps 300 --warmup 30 --age-steps 1500 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v $*', '->', round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, d['speculative_solver'])"
}
{
run Q3
run W4 --speculative-waves 4096
run W4 --speculative-waves 3584
run W4 --speculative-waves 3072
run Q3
run W4 --speculative-waves 4096
cp alphagomoku_amd/libagx_W4.so alphagomoku_amd/libagx.so
AGX_NO_BUILD=1 timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "speculative_solver_plays_the_same_games and (0-15-8 or 1-15-8)" 2>&1 | tail -2
cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
} > gpurun_out/r5o_w4.txt 2>&1
cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
cat gpurun_out/r5o_w4.txt
