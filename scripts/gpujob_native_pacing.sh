#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/t12.txt; : > $out
TIMEFORMAT='real %R s, user %U s, sys %S s'
for ahead in 2 0 2 0; do
  echo "agx_selfplay --host-steps-ahead $ahead" >> $out
  { time alphagomoku_amd/agx_selfplay --games 1024 --steps 1500 --drain-every 256 --host-steps-ahead $ahead ; } >> $out 2>&1
done
show='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"]), round(d["ranks"][0]["host_cpu_utilisation"],2), d["ranks"][0].get("host_threads_cpu_seconds"), round(d["ranks"][0]["seconds"],1))'
for hp in ; do python bench.py --steps 1000 --warmup 30 --age-steps 1500 --no-cpu-baseline --host-pacing $hp 2>/dev/null | python -c "$show" "bench host-pacing $hp" >> $out; done
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "selfplay or native" 2>&1 | tail -3 >> $out
cat $out
