#!/bin/bash
# usage: scripts/ab_nn15_run.sh A B ... — scripts/nn_inloop_check.py (15x15 6x128, in-loop launch sizes) with each variant library
cd "$(dirname "$0")/.."
cp alphagomoku_amd/libagx.so /tmp/libagx_keep.so
for v in "$@"; do
  cp alphagomoku_amd/libagx_$v.so alphagomoku_amd/libagx.so
  echo "variant $v"; AGX_NO_BUILD=1 python scripts/nn_inloop_check.py 2>&1 | head -3
done
cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
