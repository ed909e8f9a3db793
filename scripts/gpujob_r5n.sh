#!/bin/bash
# round 5, job N: search + network partitions (job M) with compute-unit counts that are multiples of 4 per XCC (whole shader-engine rounds: 128 + 128, 160 + 96)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {
  python bench.py --steps 300 --warmup 30 --age-steps 1500 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$*', '->', round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, 'nn frac', round(d['roofline']['frac'],3))"
}
{
run
run --slices 2 --network-cus 128 --tree-on-network 1 --speculative-waves 3072
run --slices 2 --network-cus 160 --tree-on-network 1 --speculative-waves 2304
run --slices 2 --network-cus 160 --tree-on-network 1 --speculative-waves 1536
run --slices 4 --network-cus 160 --tree-on-network 1 --speculative-waves 3072
run --slices 4 --network-cus 160 --tree-on-network 1 --speculative-waves 2304
run --slices 4 --network-cus 128 --tree-on-network 1 --speculative-waves 3072
run --slices 2 --network-cus 192 --tree-on-network 1 --speculative-waves 1536
run
} > gpurun_out/r5n_partitions.txt 2>&1
cat gpurun_out/r5n_partitions.txt
