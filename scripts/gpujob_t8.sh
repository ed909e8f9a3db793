#!/bin/bash
# targeted tests of this batch: network parity, the boundary programs, per-position solver, the thicker full-size property, a short bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1700 python -m pytest tests/test_nn_gpu.py tests/test_boundary_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/t8_a.txt
timeout 1700 python -m pytest tests/test_engine_gpu.py -x -q -k "solver_matches_oracle_per_position or full_size_pool or pattern_state" 2>&1 | tail -8 > gpurun_out/t8_b.txt
timeout 600 python bench.py --steps 600 --warmup 20 --no-cpu-baseline > gpurun_out/t8_line.json 2> gpurun_out/t8_line.err
cat gpurun_out/t8_a.txt gpurun_out/t8_b.txt; head -c 900 gpurun_out/t8_line.json
