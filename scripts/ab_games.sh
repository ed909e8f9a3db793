#!/bin/bash
# usage: scripts/ab_games.sh "<games list>" "<extra bench args>" A B [C ...] — one lock-step pool per variant and pool size, same box
cd "$(dirname "$0")/.."
games="$1"; extra="$2"; shift; shift
cp alphagomoku_amd/libagx.so /tmp/libagx_keep.so
for g in $games; do
  for v in "$@"; do
    cp alphagomoku_amd/libagx_$v.so alphagomoku_amd/libagx.so
    AGX_NO_BUILD=1 python bench.py --slices 1 --games $g --steps 120 --warmup 30 --no-cpu-baseline $extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', $g, round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()})"
  done
done
cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
