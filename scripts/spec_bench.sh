#!/bin/bash
# speculative solver on/off, pool as 1 / 4 slices, same box
cd "$(dirname "$0")/.."
for sl in 4 1; do for sp in 0 1; do
  AGX_NO_BUILD=1 python bench.py --slices $sl --speculative $sp --steps ${STEPS:-150} --warmup 40 --no-cpu-baseline $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('slices', $sl, 'spec', $sp, round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, d['speculative_solver'])"
done; done
