#!/bin/bash
# round 5, job E: lane-derived values recomputed at their uses (AGX_FRESH_LANE): no scratch access left inside the solver's loop
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
PARITY_K="(speculative_solver_plays_the_same_games and (0-15-8 or 1-15-8)) or yielding_pool or solver_matches" scripts/ab_variants.sh "--steps 300 --warmup 30 --age-steps 1500" SE SD > gpurun_out/r5e_ab.txt 2>&1
cat gpurun_out/r5e_ab.txt
PARITY_K="speculative_solver_plays_the_same_games and 2-15-8" scripts/ab_variants.sh "--config C5 --steps 200 --warmup 30 --age-steps 1000" R0 R2 > gpurun_out/r5e_c5_ab.txt 2>&1
cat gpurun_out/r5e_c5_ab.txt
