#!/bin/bash
cd "$(dirname "$0")/.."
run() {
  AGX_NO_BUILD=1 python bench.py --steps ${STEPS:-400} --warmup 20 --age-steps ${AGE:-0} --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', '->', round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, 'avg chip', round(d['roofline']['time_averaged_whole_chip_frac'],3), d['speculative_solver']['batches_deferred'])"
}
run --slices 4
for t in 120 136 152; do for x in 8 16; do
  run --slices 8 --network-cus $t --tree-cus $x --speculative-waves 8192
done; done
run --slices 4 --network-cus 136 --tree-cus 16 --speculative-waves 8192
run --slices 8 --network-cus 136 --tree-cus 16 --speculative-waves 4096
run --slices 8 --network-cus 136 --tree-cus 16 --speculative-waves 8192 --yield-fraction 0.8
