#!/bin/bash
# the round's bench lines and profiles (run through gpurun from the repo root); copies go to profiles/<tag>_*
tag=${1:-r05}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
# the profile passes first: the lines behind them then quote the PMC summary of THIS build (bench.py compares source hashes)
bash scripts/profile_bench.sh $tag 300 1000 > gpurun_out/${tag}_profile.log 2>&1
bash scripts/profile_bench.sh ${tag}_c4 150 600 "--config C4" > gpurun_out/${tag}_c4_profile.log 2>&1
bash scripts/profile_bench.sh ${tag}_c5 100 400 "--config C5" > gpurun_out/${tag}_c5_profile.log 2>&1
cp gpurun_out/${tag}_pmc_summary.json profiles/${tag}_pmc_summary.json
python bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench_line.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_line_driver_window_20_steps.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --stagger 0 > gpurun_out/${tag}_bench_line_driver_window_slices_in_phase.json 2>/dev/null
python bench.py --config C3 --steps 3000 --no-cpu-baseline > gpurun_out/${tag}_bench_line_c3_standard_10x128_800.json 2>/dev/null
python bench.py --config C4 --steps 3000 --no-cpu-baseline > gpurun_out/${tag}_bench_line_c4_caro5_20x20.json 2>/dev/null
python bench.py --config C5 --steps 3000 --no-cpu-baseline > gpurun_out/${tag}_bench_line_c5_renju_1600.json 2>/dev/null
python bench.py --policy-gain 2.5 --steps 3000 --no-cpu-baseline > gpurun_out/${tag}_bench_line_trained_like_policy.json 2>/dev/null
python bench.py --speculative 0 --yield-fraction 0.75 --steps 3000 --no-cpu-baseline > gpurun_out/${tag}_bench_line_serial_solver.json 2>/dev/null
for f in gpurun_out/${tag}_bench_line*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', round(d['value']), round(d['ms_per_step'],2), round(d['games_per_sec'],1), round(d['roofline']['frac'],3), round(d['roofline']['time_averaged_whole_chip_frac'],3), d['speculative_solver'])"; done
