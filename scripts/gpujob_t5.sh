#!/bin/bash
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -12
for yf in 0.5 0.6 0.7; do python bench.py --steps 600 --warmup 20 --age-steps 1500 --no-cpu-baseline --yield-fraction $yf 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('yield $yf', round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, d['speculative_solver'])"; done
