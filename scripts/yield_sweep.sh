#!/bin/bash
# usage: scripts/yield_sweep.sh "<fractions>" [extra bench args] — the default pool at several solver_yield_fraction settings on one box
cd "$(dirname "$0")/.."
for y in $1; do
  python bench.py --steps 1500 --warmup 30 --age-steps 2000 --no-cpu-baseline --yield-fraction $y $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('yield $y', round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, d['speculative_solver'])"
done
