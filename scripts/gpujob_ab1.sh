#!/bin/bash
scripts/ab_variants.sh "--steps 400 --warmup 20 --age-steps 1500" X0 X2 X3 X4 X5 C D X0 > gpurun_out/r04_ab1.txt 2>&1
scripts/ab_variants.sh "--steps 400 --warmup 20 --age-steps 1500 --speculative-waves 2048" B X0 >> gpurun_out/r04_ab1.txt 2>&1
cat gpurun_out/r04_ab1.txt
