#!/bin/bash
# round 5, job V: the new default-waves test; wave / game time line of the renju search launch WITH parking (profile build PR)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_engine_gpu.py -x -q -k "default_search_waves" 2>&1 | tail -3
cp alphagomoku_amd/libagx.so /tmp/libagx_keep.so
cp alphagomoku_amd/libagx_PR.so alphagomoku_amd/libagx.so
AGX_SPEC_TRACE=gpurun_out/r5v_trace.txt AGX_NO_BUILD=1 timeout 900 python bench.py --config C5 --steps 203 --warmup 20 --age-steps 1500 --no-cpu-baseline > gpurun_out/r5v_prof_line.json 2> gpurun_out/r5v_prof.err
cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
grep -h "k_search_spec profile" gpurun_out/r5v_prof.err | tail -2 | cut -c1-400
python scripts/spec_waves.py gpurun_out/r5v_trace.txt.waves 1024 > gpurun_out/r5v_waves.txt 2>&1
cat gpurun_out/r5v_waves.txt
python -c "import json; d=json.load(open('gpurun_out/r5v_prof_line.json')); print(round(d['value']), d['kernel_ms_per_step'], d['speculative_solver'])"
