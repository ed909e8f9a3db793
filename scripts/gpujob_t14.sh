#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r04_native_phase.txt; : > $out
TIMEFORMAT='real %R s, user %U s, sys %S s'
for ahead in 2 0 2; do
  echo "agx_selfplay --host-steps-ahead $ahead" >> $out
  { time alphagomoku_amd/agx_selfplay --games 1024 --steps 1500 --drain-every 256 --host-steps-ahead $ahead ; } 2>&1 | sed 's/"ms_per_step.*//' >> $out
done
timeout 900 python scripts/boundary_rate.py 1500 2048 >> $out 2>&1
timeout 900 python scripts/boundary_rate.py 3000 2048 >> $out 2>&1
timeout 900 python -m pytest tests/test_boundary_gpu.py tests/test_engine_gpu.py -x -q -k "boundary or native or generator or restart" 2>&1 | tail -3 >> $out
cat $out
