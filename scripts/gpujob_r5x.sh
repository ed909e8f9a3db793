#!/bin/bash
# round 5, job X: the number of chip slices again, with 16 solver waves per compute unit (2 / 4 / 8 slices; C2 and C5)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {
  python bench.py --steps 400 --warmup 30 --age-steps 1500 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$*', '->', round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, 'frac', round(d['roofline']['frac'],3))"
}
{
run --slices 4
run --slices 2
run --slices 8
run --slices 4
run --slices 2
run --config C5 --slices 4
run --config C5 --slices 2
run --config C5 --slices 8
} > gpurun_out/r5x_slices.txt 2>&1
cat gpurun_out/r5x_slices.txt
