#!/bin/bash
# round 5, final job 2: the other lines of the round's table, in-kernel stamps of the search launch (C2 and C5), one place / undo by itself
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/gpujob_lines_final.sh 2>&1 | tail -14
bash scripts/solver_profile.sh r05 > gpurun_out/r05_stamps.log 2>&1; tail -12 gpurun_out/r05_stamps.log | cut -c1-300
cp alphagomoku_amd/libagx_P.so /tmp/libagx_P_keep.so; cp alphagomoku_amd/libagx_PR.so alphagomoku_amd/libagx_P.so
bash scripts/solver_profile.sh r05_c5 --config C5 > gpurun_out/r05_c5_stamps.log 2>&1; tail -12 gpurun_out/r05_c5_stamps.log | cut -c1-300
cp /tmp/libagx_P_keep.so alphagomoku_amd/libagx_P.so
bash scripts/gpujob_place.sh 2>&1 | tail -4 | cut -c1-300
{ for y in 0.5 0.6 0.7 0.8; do echo "yield $y"; python bench.py --steps 400 --warmup 30 --age-steps 1500 --no-cpu-baseline --yield-fraction $y 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, d['speculative_solver'])"; done; } > gpurun_out/r05_sweep_yield_fraction.txt 2>&1
cat gpurun_out/r05_sweep_yield_fraction.txt
