#!/bin/bash
scripts/ab_variants.sh "--steps 400 --warmup 20 --age-steps 1500" X0 E G H I X0 > gpurun_out/r04_ab2.txt 2>&1
cat gpurun_out/r04_ab2.txt
bash scripts/solver_profile.sh r04_c5 --config C5 > gpurun_out/r04_c5_profile.txt 2>&1
tail -12 gpurun_out/r04_c5_profile.txt
