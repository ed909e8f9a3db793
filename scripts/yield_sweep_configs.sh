#!/bin/bash
# usage: scripts/yield_sweep_configs.sh "<configs>" "<fractions>" — solver_yield_fraction per BASELINE configuration, one box
cd "$(dirname "$0")/.."
for c in $1; do for y in $2; do
  python bench.py --config $c --steps 1000 --warmup 30 --age-steps 1500 --no-cpu-baseline --yield-fraction $y 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c yield $y', round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()})"
done; done
