#!/bin/bash
# Bench lines of the other BASELINE configurations + the 2-rank flow on one GPU + evaluation matches (run through gpurun from the repo root);
# copies go to profiles/<tag>_bench_line_*.json
tag=${1:-r02}
steps=${2:-3000}
mkdir -p gpurun_out
python bench.py --rules 1 --blocks 10 --sims 800 --steps $steps --no-cpu-baseline > gpurun_out/${tag}_bench_line_c3_standard_10x128_800.json 2> gpurun_out/${tag}_c3.err
python bench.py --board 20 --rules 3 --blocks 10 --steps $steps --no-cpu-baseline > gpurun_out/${tag}_bench_line_c4_caro5_20x20.json 2> gpurun_out/${tag}_c4.err
python bench.py --rules 2 --blocks 10 --sims 1600 --steps $steps --no-cpu-baseline > gpurun_out/${tag}_bench_line_c5_renju_1600.json 2> gpurun_out/${tag}_c5.err
AGX_FORCE_DEVICE=0 python bench.py --gpus 2 --games 512 --steps 1000 --no-cpu-baseline > gpurun_out/${tag}_bench_line_2_ranks_on_one_gpu.json 2> gpurun_out/${tag}_g2.err
python scripts/match_bench.py --pairs 1024 --steps $steps > gpurun_out/${tag}_match_bench_line.json 2> gpurun_out/${tag}_match.err
for f in gpurun_out/${tag}_bench_line_*.json gpurun_out/${tag}_match_bench_line.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], round(d["value"]), round(d["ms_per_step"], 2), d.get("n_gpus"), {k: round(v, 2) for k, v in d.get("kernel_ms_per_step", {}).items()}, round(d.get("games_per_sec", 0), 1), round(d.get("moves_per_sec", 0)))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
