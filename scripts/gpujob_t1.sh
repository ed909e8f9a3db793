#!/bin/bash
python -m pytest tests/test_engine_gpu.py -x -q -k "speculative_solver_plays_the_same_games and 0-15-8" 2>&1 | grep -E "^E |passed|failed" | head -20
python -m pytest tests/test_nn_gpu.py -x -q -s 2>&1 | grep -E "^E |passed|failed|fp16-storage" | head -30
python -m pytest tests/test_boundary_gpu.py -x -q 2>&1 | tail -15
python -m pytest tests/test_engine_gpu.py -x -q -k "restored or time_limited or set_batch_size or raw_network" 2>&1 | tail -15
AGX_NN_SINGLE_PLANE=1 python scripts/nn_check.py 2>&1 | grep "TFLOP"
python scripts/nn_check.py 2>&1 | grep "TFLOP"
