#!/bin/bash
# usage: scripts/ab_engine_run.sh "<bench args>" V1 V2 ... — bench.py with alphagomoku_amd/libagx_<V>.so in place of the library ("-" = the library as built)
cd "$(dirname "$0")/.."
args="$1"; shift
cp alphagomoku_amd/libagx.so /tmp/libagx_keep.so
for v in "$@"; do
  if [ "$v" = "-" ]; then cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so; else cp alphagomoku_amd/libagx_$v.so alphagomoku_amd/libagx.so; fi
  AGX_NO_BUILD=1 python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()})"
done
cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
