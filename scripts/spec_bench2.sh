#!/bin/bash
cd "$(dirname "$0")/.."
for y in 0.9; do for sl in 4; do
  AGX_SPEC_TRACE=gpurun_out/trace_s$sl.txt AGX_NO_BUILD=1 python bench.py --slices $sl --speculative 1 --yield-fraction $y --steps 1500 --warmup 0 --no-cpu-baseline $EXTRA 2>gpurun_out/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('yield', $y, 'slices', $sl, round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, d['speculative_solver'])"
  grep "k_search_spec profile" gpurun_out/err.log | tail -1
done; done
cp alphagomoku_amd/libagx_B.so alphagomoku_amd/libagx.so
AGX_NO_BUILD=1 timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "yielding_pool or (speculative_solver_plays and (0-15-8 or 1-15-8))" 2>&1 | tail -3
