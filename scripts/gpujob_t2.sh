#!/bin/bash
scripts/ab_variants.sh "--steps 400 --warmup 20 --age-steps 1500" K J L K > gpurun_out/r04_ab3.txt 2>&1
grep -v "^ \|assert" gpurun_out/r04_ab3.txt | cut -c1-260
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
