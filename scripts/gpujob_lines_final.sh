#!/bin/bash
# the other lines of the round's table from the final build (the default line, the window, the trace and the PMC passes: scripts/gpujob_pmc.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r05
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --stagger 0 > gpurun_out/${tag}_bench_line_driver_window_slices_in_phase.json 2>/dev/null
python bench.py --config C3 --steps 3000 --no-cpu-baseline > gpurun_out/${tag}_bench_line_c3_standard_10x128_800.json 2>/dev/null
python bench.py --config C4 --steps 3000 --no-cpu-baseline > gpurun_out/${tag}_bench_line_c4_caro5_20x20.json 2>/dev/null
python bench.py --config C5 --steps 3000 --no-cpu-baseline > gpurun_out/${tag}_bench_line_c5_renju_1600.json 2>/dev/null
python bench.py --policy-gain 2.5 --steps 3000 --no-cpu-baseline > gpurun_out/${tag}_bench_line_trained_like_policy.json 2>/dev/null
python bench.py --speculative 0 --yield-fraction 0.75 --steps 3000 --no-cpu-baseline > gpurun_out/${tag}_bench_line_serial_solver.json 2>/dev/null
python scripts/match_bench.py --pairs 1024 --steps 3000 > gpurun_out/${tag}_match_bench_line.json 2> gpurun_out/${tag}_match.err
AGX_FORCE_DEVICE=0 python bench.py --gpus 2 --games 512 --steps 1000 --no-cpu-baseline > gpurun_out/${tag}_bench_line_2_ranks_on_one_gpu.json 2> gpurun_out/${tag}_2ranks.err
AGX_FORCE_DEVICE=0 python bench.py --gpus 8 --config C3 --games 128 --steps 300 --no-cpu-baseline --slices 1 > gpurun_out/${tag}_bench_line_8_ranks_on_one_gpu.json 2> gpurun_out/${tag}_8ranks.err
python bench.py --steps 30000 --no-cpu-baseline > gpurun_out/${tag}_bench_line_soak_30000_steps.json 2> gpurun_out/${tag}_soak.err
ls -la gpurun_out/${tag}_bench_line*.json | awk '{print $5, $9}'
