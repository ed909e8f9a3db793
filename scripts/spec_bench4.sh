#!/bin/bash
cd "$(dirname "$0")/.."
run() {
  AGX_NO_BUILD=1 python bench.py --steps ${STEPS:-600} --warmup 20 --age-steps ${AGE:-1500} --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', '->', round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, 'avg chip', round(d['roofline']['time_averaged_whole_chip_frac'],3), d['speculative_solver'], round(d['games_per_sec'],1))"
}
run --slices 4
run --slices 4 --yield-fraction 0.85
run --slices 4 --yield-fraction 0.95
run --slices 4 --speculative-waves 2560
run --slices 4 --speculative-waves 2048
run --slices 2
run --slices 8
run --slices 4 --speculative 0 --yield-fraction 0.75
run --slices 4 --policy-gain 2.5
