#!/bin/bash
# usage: scripts/slices_sweep.sh "<slice counts>" — the default pool stepped as that many CU-masked slices, one box
cd "$(dirname "$0")/.."
for n in $1; do
  python bench.py --steps 1500 --warmup 30 --age-steps 2000 --no-cpu-baseline --slices $n 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('slices $n', round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()})"
done
