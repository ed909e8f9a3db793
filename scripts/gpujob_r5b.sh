#!/bin/bash
# round 5, job B: wave timeline of the search launch (profile build), the co-resident pairing A/B, the new boundary / bench / network tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
cp alphagomoku_amd/libagx.so /tmp/libagx_keep.so
# 1. profile build: in-kernel stamps + the per-wave trace of the last launches
cp alphagomoku_amd/libagx_P.so alphagomoku_amd/libagx.so
AGX_SPEC_TRACE=gpurun_out/r5b_trace.txt AGX_NO_BUILD=1 timeout 600 python bench.py --steps 300 --warmup 20 --age-steps 1500 --no-cpu-baseline > gpurun_out/r5b_prof_line.json 2> gpurun_out/r5b_prof.err
cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
grep -h "profile\|frame machine\|generate()\|update_around" gpurun_out/r5b_prof.err | tail -8
python scripts/spec_waves.py gpurun_out/r5b_trace.txt.waves 768 > gpurun_out/r5b_waves.txt 2>&1
cat gpurun_out/r5b_waves.txt
python scripts/spec_trace.py gpurun_out/r5b_trace.txt
# 1b. search-launch variants on this box: S1 snapshot undo, S3 + opaque lane in the pattern update, S4 + mbcnt, S5 mbcnt only
scripts/ab_variants.sh "--steps 300 --warmup 30 --age-steps 1500" S1 S3 S4 S5 > gpurun_out/r5b_ab.txt 2>&1
cat gpurun_out/r5b_ab.txt
# 2. the co-resident pairing: (a) as built, 4 slices; (b) capped in-place tower, 4 slices = the cost side; (c) the same tower, 8 half-slices in 4 CU blocks
#    = the pairing; (d) the built tower in that arrangement = time multiplexing only (its workgroups fill a compute unit)
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', round(d['value']), 'ms/step', round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, 'tower frac', round(r['frac'],3), 'chip-wide MFMA', round(r['time_averaged_whole_chip_frac'],3))"; }
B="--steps 300 --warmup 30 --age-steps 1500 --no-cpu-baseline"
{
for rep in 1 2; do
AGX_NO_BUILD=1 python bench.py $B 2>/dev/null | line "a:as-built,4-slices"
cp alphagomoku_amd/libagx_N3.so alphagomoku_amd/libagx.so
AGX_NN_SINGLE_PLANE=1 AGX_NO_BUILD=1 python bench.py $B 2>/dev/null | line "b:capped-inplace-tower,4-slices"
AGX_NN_SINGLE_PLANE=1 AGX_NO_BUILD=1 python bench.py $B --slices 8 --share-cus 2 2>/dev/null | line "c:capped-inplace-tower,8-half-slices-in-4-blocks"
cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
AGX_NO_BUILD=1 python bench.py $B --slices 8 --share-cus 2 2>/dev/null | line "d:as-built,8-half-slices-in-4-blocks"
done
} > gpurun_out/r5b_coresident.txt 2>&1
cat gpurun_out/r5b_coresident.txt
cp alphagomoku_amd/libagx_N3.so alphagomoku_amd/libagx.so
AGX_NN_SINGLE_PLANE=1 AGX_NO_BUILD=1 timeout 600 python -m pytest tests/test_nn_gpu.py -x -q -k "forward_matches_oracle" 2>&1 | tail -1
cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
# 3. the new tests
timeout 1200 python -m pytest tests/test_boundary_gpu.py -x -q -k "sigint or error_behaviour or restart" > gpurun_out/r5b_boundary.log 2>&1; tail -3 gpurun_out/r5b_boundary.log
timeout 900 python -m pytest tests/test_bench_gpu.py -x -q -k "eight_ranks" -s > gpurun_out/r5b_eight.log 2>&1; tail -3 gpurun_out/r5b_eight.log; grep footprint gpurun_out/r5b_eight.log
timeout 900 python -m pytest tests/test_nn_gpu.py -x -q -k "fp16_storage" -s > gpurun_out/r5b_fp16.log 2>&1; tail -2 gpurun_out/r5b_fp16.log; grep "oracle" gpurun_out/r5b_fp16.log
timeout 600 python -m pytest tests/test_engine_gpu.py -x -q -k "time_limited" 2>&1 | tail -1
