#!/bin/bash
python -m pytest tests/test_engine_gpu.py -x -q -k "restored_games" 2>&1 | grep -E "^E |passed|failed|Error" | head -30
scripts/ab_variants.sh "--steps 400 --warmup 20 --age-steps 1500" N R S T N > gpurun_out/r04_ab5.txt 2>&1
grep -v "^ \|assert" gpurun_out/r04_ab5.txt | cut -c1-200
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stagger', round(d['value']), d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'])"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --stagger 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('in phase', round(d['value']), d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'])"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stagger', round(d['value']), d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'])"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --stagger 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('in phase', round(d['value']), d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'])"
