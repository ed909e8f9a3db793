#!/bin/bash
# round 5, job Q: the build with 16 solver waves per compute unit on 15x15 boards: full GPU suite, then C2 / C4 lines and the yield fraction again
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee gpurun_out/r5q_suite_tail.txt
run() {
  python bench.py --steps 300 --warmup 30 --age-steps 1500 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$*', '->', round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, d['roofline_solver']['waves_per_launch'], d['speculative_solver'])"
}
{
run
run --config C4
run --config C4 --speculative-waves 3072
run --config C4
run --config C4 --speculative-waves 3072
for y in 0.5 0.6 0.7 0.8; do run --yield-fraction $y; done
run
} > gpurun_out/r5q_lines.txt 2>&1
cat gpurun_out/r5q_lines.txt
