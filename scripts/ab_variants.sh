#!/bin/bash
# usage: scripts/ab_variants.sh "<bench args>" V1 V2 ... — for every alphagomoku_amd/libagx_<V>.so (built by scripts/build_engine_variant.sh): the speculative
# solver's parity test against the oracle (15x15 freestyle / standard: what an AGX_QUICK build instantiates; PARITY_K selects other tests, e.g. the renju
# ones for -DAGX_QUICK_RENJU=true builds), then bench.py; one box, so the lines compare
cd "$(dirname "$0")/.."
args="$1"; shift
cp alphagomoku_amd/libagx.so /tmp/libagx_keep.so
for v in "$@"; do
  cp alphagomoku_amd/libagx_$v.so alphagomoku_amd/libagx.so
  AGX_NO_BUILD=1 timeout 600 python -m pytest tests/test_engine_gpu.py -x -q -k "${PARITY_K:-speculative_solver_plays_the_same_games and (0-15-8 or 1-15-8)}" > /tmp/parity_$v.log 2>&1
  par=$(tail -1 /tmp/parity_$v.log)
  case "$par" in *failed*|*error*) grep -E "Error|error|assert|^E " /tmp/parity_$v.log | head -12;; esac
  for rep in 1 2; do
  AGX_NO_BUILD=1 python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, d['speculative_solver'])"
  done
  echo "$v parity: $par"
done
cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
