#!/bin/bash
# which environment setting puts the HIP / HSA runtime's helper thread to sleep (one thread stays at 100 % with host pacing on)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/t10.txt; : > $out
show='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"]), round(d["ranks"][0]["host_cpu_utilisation"],2), d["ranks"][0].get("host_threads_cpu_seconds"), round(d["ranks"][0]["seconds"],1))'
run() { label="$1"; shift; env "$@" python bench.py --steps 600 --warmup 30 --age-steps 1500 --no-cpu-baseline 2>/dev/null | python -c "$show" "$label" >> $out; }
run "default" A=1
run "HSA_ENABLE_INTERRUPT=1" HSA_ENABLE_INTERRUPT=1
run "ROC_ACTIVE_WAIT_TIMEOUT=0" ROC_ACTIVE_WAIT_TIMEOUT=0
run "AMD_DIRECT_DISPATCH=0" AMD_DIRECT_DISPATCH=0
run "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
run "ROC_SIGNAL_POOL_SIZE=64" ROC_SIGNAL_POOL_SIZE=64
run "DEBUG_HIP_BLOCK_SYNC" DEBUG_HIP_BLOCK_SYNC=50
run "ROC_CPU_WAIT_FOR_SIGNAL=1" ROC_CPU_WAIT_FOR_SIGNAL=1
run "ROC_CPU_WAIT_FOR_SIGNAL=0" ROC_CPU_WAIT_FOR_SIGNAL=0
timeout 900 python -m pytest tests/test_boundary_gpu.py tests/test_bench_gpu.py -x -q 2>&1 | tail -4 >> $out
cat $out
