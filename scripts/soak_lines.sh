#!/bin/bash
cd "$(dirname "$0")/.."
python bench.py --steps 30000 --no-cpu-baseline > gpurun_out/r06_bench_line_soak_30000_steps.json 2> gpurun_out/r06_soak.err
python scripts/match_bench.py --pairs 1024 --steps 3000 > gpurun_out/r06_match_bench_line.json 2> gpurun_out/r06_match.err
AGX_FORCE_DEVICE=0 python bench.py --gpus 2 --games 512 --steps 1000 --no-cpu-baseline > gpurun_out/r06_bench_line_2_ranks_on_one_gpu.json 2> gpurun_out/r06_2ranks.err
AGX_FORCE_DEVICE=0 python bench.py --gpus 8 --config C3 --games 128 --steps 300 --no-cpu-baseline --slices 1 > gpurun_out/r06_bench_line_8_ranks_on_one_gpu.json 2> gpurun_out/r06_8ranks.err
for f in gpurun_out/r06_bench_line_soak_30000_steps.json gpurun_out/r06_bench_line_2_ranks_on_one_gpu.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', round(d['value']), round(d['ms_per_step'],2), round(d['games_per_sec'],1), d['peak_tree_per_game'], d['speculative_solver'])"; done
tail -2 gpurun_out/r06_match_bench_line.json | cut -c1-600; tail -3 gpurun_out/r06_soak.err gpurun_out/r06_match.err
