#!/bin/bash
cd "$(dirname "$0")/.."
for w in 2048; do
  AGX_SPEC_TRACE=gpurun_out/trace_$w.txt AGX_NO_BUILD=1 python bench.py --slices 1 --speculative 1 --speculative-waves $w --steps 301 --warmup 0 --no-cpu-baseline $EXTRA 2>gpurun_out/err_$w.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('waves', $w, round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, d['speculative_solver'])"
  grep "k_search_spec profile" gpurun_out/err_$w.log | tail -1
done
