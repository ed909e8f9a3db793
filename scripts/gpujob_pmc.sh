#!/bin/bash
# trace + PMC passes of the build (hash tie of profiles/*_pmc_summary.json), then the default line and the driver's window with the summary in place
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/profile_bench.sh r05 300 1000 > gpurun_out/r05_profile.log 2>&1
bash scripts/profile_bench.sh r05_c4 150 600 "--config C4" > gpurun_out/r05_c4_profile.log 2>&1
bash scripts/profile_bench.sh r05_c5 100 400 "--config C5" > gpurun_out/r05_c5_profile.log 2>&1
cp gpurun_out/r05_pmc_summary.json profiles/r05_pmc_summary.json
python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_line.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_line_driver_window_20_steps.json 2>/dev/null
python bench.py --steps 40 --warmup 30 --age-steps 0 --no-cpu-baseline > gpurun_out/r05_pmc_workload_line.json 2>/dev/null
timeout 900 python -m pytest tests/test_nn_gpu.py tests/test_engine_gpu.py -x -q -k 'nn or network or raw or whole_games_bit_exact or full_size_pool' 2>&1 | tail -3
for f in gpurun_out/r05_bench_line.json gpurun_out/r05_bench_line_driver_window_20_steps.json; do python -c "import json; d=json.load(open('$f')); print('$f', round(d['value']), d['roofline']['frac'], d['roofline']['mfma_busy_fraction_pmc'], d['source_hash'])"; done
