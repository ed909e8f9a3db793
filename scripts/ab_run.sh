#!/bin/bash
# usage: scripts/ab_run.sh "<command>" A B [C ...]   — runs the command with alphagomoku_amd/libagx_<X>.so swapped in, 2 rounds, same box
cd "$(dirname "$0")/.."
cmd="$1"; shift
cp alphagomoku_amd/libagx.so /tmp/libagx_keep.so
for round in 1 2; do
  for v in "$@"; do
    cp alphagomoku_amd/libagx_$v.so alphagomoku_amd/libagx.so
    echo "== $v"
    bash -c "$cmd"
  done
done
cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
