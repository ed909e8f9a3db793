#!/bin/bash
# round 5, job I: two more solver variants on the final build: the centre's list edit merged into the first neighbour pass (T1), this lane's line shape of
# getDefensiveMoves requested once per generate() (T2), both (T3)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
PARITY_K="(speculative_solver_plays_the_same_games and (0-15-8 or 1-15-8)) or yielding_pool or pattern_state" scripts/ab_variants.sh "--steps 300 --warmup 30 --age-steps 1500" T0 T1 T2 T3 > gpurun_out/r5i_ab.txt 2>&1
cat gpurun_out/r5i_ab.txt
