#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/t11.txt; : > $out
show='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"]), round(d["ranks"][0]["host_cpu_utilisation"],2), d["ranks"][0].get("host_threads_cpu_seconds"), round(d["ranks"][0]["seconds"],1))'
for hp in 2 0; do python bench.py --steps 600 --warmup 30 --age-steps 1500 --no-cpu-baseline --host-pacing $hp 2>/dev/null | python -c "$show" "host-pacing $hp" >> $out; done
cat $out
