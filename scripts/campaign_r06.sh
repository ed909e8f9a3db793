#!/bin/bash
# the round's measurement campaign, ONE gpurun job on the final build: bench lines + trace + PMC passes (final_lines.sh), soak / match / multi-rank
# lines (soak_lines.sh), the 10 000-step runs, the PMC passes' command un-profiled, and — unless CAMPAIGN_SKIP_STAMPS / CAMPAIGN_SKIP_TESTS are set —
# the solver's in-kernel stamps (profile builds libagx_P.so / libagx_Pr.so), the per-place counters and the GPU suite
cd "$(dirname "$0")/.."
bash scripts/final_lines.sh r06 2>&1 | tail -12
bash scripts/soak_lines.sh 2>&1 | tail -8
for c in C3 C4 C5; do lc=$(echo $c | tr A-Z a-z); python bench.py --config $c --steps 10000 --no-cpu-baseline > gpurun_out/r06_bench_line_${lc}_soak_10000_steps.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r06_bench_line_${lc}_soak_10000_steps.json')); print('$c soak', round(d['value']), d['ms_per_step'])"; done
python bench.py --steps 40 --warmup 30 --age-steps 0 --no-cpu-baseline > gpurun_out/r06_pmc_command_line.json 2>/dev/null
python bench.py --config C5 --steps 40 --warmup 30 --age-steps 0 --no-cpu-baseline > gpurun_out/r06_c5_pmc_command_line.json 2>/dev/null
if [ -z "$CAMPAIGN_SKIP_STAMPS" ]; then
bash scripts/solver_profile.sh r06 2>&1 | head -9
AGX_NO_BUILD=1 AGX_LIB_PATH="$PWD/alphagomoku_amd/libagx_Pr.so" python bench.py --config C5 --steps 300 --warmup 20 --age-steps 1500 --no-cpu-baseline > gpurun_out/r06_c5_prof_spec.json 2> gpurun_out/r06_c5_prof_spec.err; grep -h "renju\|solver profile" gpurun_out/r06_c5_prof_spec.err | tail -2
(cd /tmp; export TMPDIR=/tmp; rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES -d $GRAFT_REPO_ROOT/gpurun_out/place_pmc -o p -- python3 $GRAFT_REPO_ROOT/scripts/place_pmc.py run > /dev/null 2>&1)
python3 scripts/place_pmc.py summary gpurun_out/place_pmc/p_results.db gpurun_out/r06_place_pmc.json | tail -7; rm -rf gpurun_out/place_pmc
fi
if [ -z "$CAMPAIGN_SKIP_TESTS" ]; then
python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r06_gpu_test_suite_tail.txt
fi
