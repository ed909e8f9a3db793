#!/bin/bash
# host pacing (blocking events, N steps ahead) against the free-running host loop: throughput and host CPU per rank; then the slice split re-measured
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r04_host_pacing.txt
: > $out
show='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],2), "host cpu per rank", [round(r["host_cpu_utilisation"],2) for r in d["ranks"]])'
for hp in 0 2 4 0 2; do
  python bench.py --steps 1500 --warmup 30 --age-steps 2000 --no-cpu-baseline --host-pacing $hp 2>/dev/null | python -c "$show" "1 rank, host-pacing $hp" >> $out
done
for hp in 0 2; do
  AGX_FORCE_DEVICE=0 python bench.py --gpus 8 --config C3 --games 128 --steps 300 --warmup 5 --age-steps 0 --table-entries 65536 --no-cpu-baseline --slices 1 --host-pacing $hp 2>/dev/null | python -c "$show" "8 ranks on one GPU, host-pacing $hp" >> $out
done
echo "== slice split" >> $out
bash scripts/slices_sweep.sh "2 4 8 4" >> $out 2>&1
cat $out
