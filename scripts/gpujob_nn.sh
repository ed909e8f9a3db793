#!/bin/bash
# network-kernel A/B: stamps of the profile variant P (if built), then the in-loop launch sizes with every variant library named in $@ (base = the shipped one)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/${NN_TAG:-r04_nn_ab}.txt
: > $out
if [ -f alphagomoku_amd/libagx_P.so ]; then
  echo "== stamps (AGX_NN_PROFILE), 6x128 15x15" >> $out
  timeout 300 bash scripts/nn_prof_variant.sh P 6 128 15 >> $out 2>&1
fi
cp alphagomoku_amd/libagx.so alphagomoku_amd/libagx_base.so
for round in 1 2; do
  echo "== round $round" >> $out
  timeout 600 bash scripts/ab_nn15_run.sh base "$@" >> $out 2>&1
done
if [ -n "$NN_20" ]; then
  for round in 1 2; do
    for v in base "$@"; do
      cp alphagomoku_amd/libagx.so /tmp/libagx_keep.so; cp alphagomoku_amd/libagx_$v.so alphagomoku_amd/libagx.so
      AGX_VARIANT=$v AGX_NO_BUILD=1 timeout 300 python scripts/nn_20x20_bench.py 2>&1 | tail -1 >> $out
      cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
    done
  done
fi
if [ -n "$NN_CHECK" ]; then
  for v in "$@"; do
    cp alphagomoku_amd/libagx.so /tmp/libagx_keep.so; cp alphagomoku_amd/libagx_$v.so alphagomoku_amd/libagx.so
    echo "== parity $v" >> $out
    AGX_NO_BUILD=1 AGX_ALLOW_STALE=1 timeout 900 python -m pytest tests/test_nn_gpu.py -x -q 2>&1 | tail -3 >> $out
    cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
  done
fi
cat $out
