#!/bin/bash
PARITY_K="(speculative_solver_plays_the_same_games and 2-15-8) or (whole_games_bit_exact_with_stand_in and 2-4-100) or (whole_games_bit_exact_with_stand_in and 2-8-60)" scripts/ab_variants.sh "--config C5 --steps 300 --warmup 20 --age-steps 1200" V U V > gpurun_out/r04_ab6_renju.txt 2>&1
grep -v "^ \|assert" gpurun_out/r04_ab6_renju.txt | cut -c1-260
