#!/bin/bash
cd "$(dirname "$0")/.."
for age in 300 1000 3000; do
  echo "== age $age"
  AGX_NO_BUILD=1 timeout -s KILL 120 python bench.py --gpus 1 --games 128 --sims 100 --steps 60 --warmup 5 --age-steps $age --table-entries 65536 --no-cpu-baseline --yield-fraction 0 2>&1 | tail -4 | cut -c1-600
  echo "rc=$?"
done
