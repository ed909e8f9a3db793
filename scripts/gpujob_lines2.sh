#!/bin/bash
# the default line and the driver's window once more, now that profiles/r04_pmc_summary.json carries this build's hash (the lines quote its PMC fields)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_line_driver_window_20_steps.json 2>/dev/null
python scripts/design_table.py r04 2>/dev/null | head -0
for f in gpurun_out/r04_bench_line.json gpurun_out/r04_bench_line_driver_window_20_steps.json; do python -c "import json; d=json.load(open('$f')); print('$f', round(d['value']), d['roofline'])"; done
