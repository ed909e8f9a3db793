#!/bin/bash
# round 5, job D: the wide backup (k_expand) A/B, then the full library: engine parity suite + one line per BASELINE config
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
PARITY_K="(speculative_solver_plays_the_same_games and (0-15-8 or 1-15-8)) or yielding_pool" scripts/ab_variants.sh "--steps 300 --warmup 30 --age-steps 1500" SC SB > gpurun_out/r5d_ab.txt 2>&1
cat gpurun_out/r5d_ab.txt
PARITY_K="speculative_solver_plays_the_same_games and 2-15-8" scripts/ab_variants.sh "--config C5 --steps 200 --warmup 30 --age-steps 1000" R0 R1 > gpurun_out/r5d_c5_ab.txt 2>&1
cat gpurun_out/r5d_c5_ab.txt
timeout 1500 python -m pytest tests/test_engine_gpu.py -x -q > gpurun_out/r5d_engine_tests.log 2>&1; tail -2 gpurun_out/r5d_engine_tests.log
for c in C2 C3 C4 C5; do
  timeout 900 python bench.py --config $c --steps 300 --warmup 30 --age-steps 1500 --no-cpu-baseline > gpurun_out/r5d_line_$c.json 2> gpurun_out/r5d_line_$c.err
  python -c "import json; d=json.load(open('gpurun_out/r5d_line_$c.json')); print('$c', round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, 'frac', round(d['roofline']['frac'],3))"
done
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5d_line_driver_window.json 2>/dev/null
python -c "import json; d=json.load(open('gpurun_out/r5d_line_driver_window.json')); print('driver window', round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, 'frac', round(d['roofline']['frac'],3))"
