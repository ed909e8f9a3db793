#!/bin/bash
# host pacing in the product loops: the boundary programs, the bench tests, agx_selfplay, a line with the per-thread CPU seconds
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1700 python -m pytest tests/test_boundary_gpu.py tests/test_bench_gpu.py tests/test_selfplay_native_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/t9_a.txt
for hp in 2 0; do
  python bench.py --steps 1500 --warmup 30 --age-steps 2000 --no-cpu-baseline --host-pacing $hp 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('host-pacing $hp', round(d['value']), d['ranks'][0]['host_cpu_utilisation'], d['ranks'][0].get('host_threads_cpu_seconds'), d['ranks'][0]['seconds'])" >> gpurun_out/t9_a.txt
done
( /usr/bin/time -v alphagomoku_amd/agx_selfplay --games 1024 --steps 1500 --drain-every 256 ) > gpurun_out/t9_selfplay.txt 2>&1
grep -i "simulations\|Elapsed (wall\|User time\|System time" gpurun_out/t9_selfplay.txt | head -8 >> gpurun_out/t9_a.txt
cat gpurun_out/t9_a.txt
