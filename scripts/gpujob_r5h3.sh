#!/bin/bash
# round 5, final job 3: the full GPU suite of the final tree once more (the restart test's statistical bound), the 60 000-step soak
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/r05_gpu_test_suite_tail.txt
python bench.py --steps 60000 --no-cpu-baseline > gpurun_out/r05_bench_line_soak_60000_steps.json 2> gpurun_out/r05_soak60.err
python -c "import json; d=json.load(open('gpurun_out/r05_bench_line_soak_60000_steps.json')); print(round(d['value']), d['ms_per_step'], d['games_per_sec'], d['source_hash'])"
