#!/bin/bash
scripts/ab_variants.sh "--steps 400 --warmup 20 --age-steps 1500" O M N O > gpurun_out/r04_ab4.txt 2>&1
grep -v "^ \|assert" gpurun_out/r04_ab4.txt | cut -c1-260
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -25
