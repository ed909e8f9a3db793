#!/bin/bash
# round 4: the full GPU suite, then the round's bench lines, rocprof summaries and in-kernel stamps of the final build
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee gpurun_out/r05_gputest_tail.txt
bash scripts/final_lines.sh r05 2>&1 | tail -14
bash scripts/solver_profile.sh r05 > gpurun_out/r05_stamps.log 2>&1; tail -12 gpurun_out/r05_stamps.log | cut -c1-300
bash scripts/solver_profile.sh r05_c5 --config C5 > gpurun_out/r05_c5_stamps.log 2>&1; tail -12 gpurun_out/r05_c5_stamps.log | cut -c1-300
bash scripts/gpujob_place.sh 2>&1 | tail -4 | cut -c1-300   # per-place instruction counts + an un-profiled line of the PMC passes' own workload
bash scripts/soak_lines.sh 2>&1 | tail -8 | cut -c1-400
