#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r04_boundary_rate.txt; : > $out
timeout 900 python scripts/boundary_rate.py 1500 2048 >> $out 2>&1
timeout 900 python scripts/boundary_rate.py 3000 2048 >> $out 2>&1
timeout 900 python scripts/boundary_rate.py 3000 1024 >> $out 2>&1
cat $out
