#!/bin/bash
# round 5, job S: wave / game time line of the RENJU search launch (C5, profile build PR: 16 waves per compute unit) — where its 10.5 ms go
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
cp alphagomoku_amd/libagx.so /tmp/libagx_keep.so
cp alphagomoku_amd/libagx_PR.so alphagomoku_amd/libagx.so
AGX_SPEC_TRACE=gpurun_out/r5s_trace.txt AGX_NO_BUILD=1 timeout 900 python bench.py --config C5 --steps 200 --warmup 20 --age-steps 1500 --no-cpu-baseline > gpurun_out/r5s_prof_line.json 2> gpurun_out/r5s_prof.err
cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
grep -h "profile\|frame machine\|generate()\|update_around\|renju" gpurun_out/r5s_prof.err | tail -10 | cut -c1-400
python scripts/spec_waves.py gpurun_out/r5s_trace.txt.waves 1024 > gpurun_out/r5s_waves.txt 2>&1
cat gpurun_out/r5s_waves.txt
python scripts/spec_trace.py gpurun_out/r5s_trace.txt 2>&1 | tail -30
