#!/bin/bash
# round 5, job M: the chip as a search partition + a network partition shared by the slices, with expand / advance behind the tower on the network partition
# (--tree-on-network) — the search partition then holds only persistent search launches, and the next slice's waves move in as the previous launch's leave
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
run --slices 2 --network-cus 144 --tree-on-network 1 --speculative-waves 2688
run --slices 2 --network-cus 136 --tree-on-network 1 --speculative-waves 2880
run --slices 2 --network-cus 152 --tree-on-network 1 --speculative-waves 2496
run --slices 4 --network-cus 144 --tree-on-network 1 --speculative-waves 5376
run --slices 4 --network-cus 144 --tree-on-network 1 --speculative-waves 3072
run --slices 4 --network-cus 160 --tree-on-network 1 --speculative-waves 4608
run --slices 4 --network-cus 128 --tree-on-network 1 --speculative-waves 6144
run --slices 8 --network-cus 144 --tree-on-network 1 --speculative-waves 5376
run
} > gpurun_out/r5m_partitions.txt 2>&1
cat gpurun_out/r5m_partitions.txt
