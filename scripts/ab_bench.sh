#!/bin/bash
# A/B timing of two builds of libagx.so on the SAME box: alphagomoku_amd/libagx_A.so vs libagx_B.so (boxes differ by a few percent)
cd "$(dirname "$0")/.."
cp alphagomoku_amd/libagx.so /tmp/libagx_keep.so
for round in 1 2 3; do
  for v in A B; do
    cp alphagomoku_amd/libagx_$v.so alphagomoku_amd/libagx.so
    AGX_NO_BUILD=1 python bench.py --steps 150 --warmup 40 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()})"
  done
done
cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
