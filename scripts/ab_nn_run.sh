#!/bin/bash
# usage: scripts/ab_nn_run.sh A B ... — scripts/nn_20x20_bench.py with each variant library (scripts/ab_nn_build.sh) on one box
cd "$(dirname "$0")/.."
cp alphagomoku_amd/libagx.so /tmp/libagx_keep.so
for v in "$@"; do
  cp alphagomoku_amd/libagx_$v.so alphagomoku_amd/libagx.so
  AGX_NO_BUILD=1 AGX_VARIANT=$v python scripts/nn_20x20_bench.py 2>&1 | tail -1
done
cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
