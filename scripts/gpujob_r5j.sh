#!/bin/bash
# round 5, job J: T2 (line shapes once per generate) + the evaluation weights as immediates (T5) + non-temporal pattern-table gathers (T6)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
PARITY_K="(speculative_solver_plays_the_same_games and (0-15-8 or 1-15-8)) or yielding_pool or pattern_state" scripts/ab_variants.sh "--steps 300 --warmup 30 --age-steps 1500" T0 T2 T5 T6 > gpurun_out/r5j_ab.txt 2>&1
cat gpurun_out/r5j_ab.txt
