#!/bin/bash
cd "$(dirname "$0")/.."
cp alphagomoku_amd/libagx.so /tmp/libagx_keep.so
rm -f gpurun_out/nn20_*.npz
for v in "$@"; do
  cp alphagomoku_amd/libagx_$v.so alphagomoku_amd/libagx.so
  AGX_NO_BUILD=1 AGX_VARIANT=$v python scripts/nn_20x20_dump.py 2>&1 | tail -1
done
cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
python scripts/nn_20x20_compare.py "$@"
