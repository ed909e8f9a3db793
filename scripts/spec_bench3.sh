#!/bin/bash
# two-partition pool (network CUs / search CUs shared by all slices) against the 4-slice pool, same box
cd "$(dirname "$0")/.."
run() {
  AGX_NO_BUILD=1 python bench.py --steps ${STEPS:-400} --warmup 20 --age-steps 0 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', '->', round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, 'nn frac', round(d['roofline']['frac'],3), 'avg chip', round(d['roofline']['time_averaged_whole_chip_frac'],3), d['speculative_solver']['batches_deferred'])"
}
run --slices 4
for t in 96 128 160; do for sl in 8 16; do for w in 4096 8192; do
  run --slices $sl --network-cus $t --speculative-waves $w
done; done; done
