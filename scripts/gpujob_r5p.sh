#!/bin/bash
# round 5, job P: 16 solver waves per compute unit (job O) on the other 15x15 workloads — C3 (standard, 10x128, 800 playouts) and C5 (renju: R3 = 12 waves per
# unit as built, R4 = 16) — and together with the defensive-shape preload (T2: W4T2, R4T2); all AGX_QUICK builds from a copy of csrc/
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
parity() {
  cp alphagomoku_amd/libagx_$1.so alphagomoku_amd/libagx.so
  AGX_NO_BUILD=1 timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "$2" 2>&1 | tail -1
}
{
run W4 --speculative-waves 4096
run W4T2 --speculative-waves 4096
run W4 --speculative-waves 4096
run W4T2 --speculative-waves 4096
run Q3 --config C3
run W4 --config C3 --speculative-waves 4096
run W4T2 --config C3 --speculative-waves 4096
run R3 --config C5
run R4 --config C5 --speculative-waves 4096
run R4T2 --config C5 --speculative-waves 4096
run R3 --config C5
run R4 --config C5 --speculative-waves 4096
parity W4T2 "speculative_solver_plays_the_same_games and (0-15-8 or 1-15-8)"
parity R4 "speculative_solver_plays_the_same_games and 2-15"
parity R4T2 "speculative_solver_plays_the_same_games and 2-15"
} > gpurun_out/r5p_w4.txt 2>&1
cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
cat gpurun_out/r5p_w4.txt
