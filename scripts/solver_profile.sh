#!/bin/bash
# In-kernel stamps of the search launch (run on the GPU box through gpurun): alphagomoku_amd/libagx_P.so = engine.hip built with
# -DAGX_SOLVER_PROFILE -DAGX_SPEC_PROFILE (scripts/build_engine_variant.sh P "..."); the profile lines go to stderr at agx_engine_stats.
# usage: scripts/solver_profile.sh TAG [bench args]
cd "$(dirname "$0")/.."
tag=${1:-r05}; shift
mkdir -p gpurun_out
# (the variant library is selected per process: nothing is copied over the shipped libagx.so)
export AGX_LIB_PATH="$PWD/alphagomoku_amd/libagx_P.so"
AGX_NO_BUILD=1 python bench.py --steps 400 --warmup 20 --age-steps 1500 --no-cpu-baseline "$@" > gpurun_out/${tag}_prof_spec.json 2> gpurun_out/${tag}_prof_spec.err
AGX_NO_BUILD=1 python bench.py --steps 300 --warmup 20 --age-steps 1200 --no-cpu-baseline --speculative 0 --yield-fraction 0.75 "$@" > gpurun_out/${tag}_prof_serial.json 2> gpurun_out/${tag}_prof_serial.err
grep -h "profile\|frame machine\|generate()\|update_around" gpurun_out/${tag}_prof_spec.err | tail -8
echo ---- serial
grep -h "profile\|frame machine\|generate()\|update_around" gpurun_out/${tag}_prof_serial.err | tail -8
