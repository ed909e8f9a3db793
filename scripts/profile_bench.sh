#!/bin/bash
# Profiles of the default bench workload on the GPU box (run through gpurun from the repo root):
#   gpurun_out/<tag>_trace   rocprofv3 --kernel-trace --stats          -> profiles/<tag>_bench_kernel_stats.csv (scripts/prof_summary.py)
#   gpurun_out/<tag>_pmc_*   rocprofv3 --pmc, one pass per counter set -> profiles/<tag>_pmc_summary.json       (scripts/pmc_bench_summary.py)
# Counters are collected in their own runs (no tracing flags besides the kernel trace the tool adds itself).
tag=${1:-r05}
steps=${2:-300}
extra="${4:-}"   # e.g. "--config C4": the same passes for another BASELINE configuration (tag r03_c4)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_trace -o t -- python3 bench.py --steps $steps --warmup 30 --age-steps ${3:-1000} --no-cpu-baseline $extra > gpurun_out/${tag}_trace.json 2> gpurun_out/${tag}_trace.err
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${tag}_pmc_fetch -o p -- python3 bench.py --steps 40 --warmup 30 --age-steps 0 --no-cpu-baseline $extra > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${tag}_pmc_write -o p -- python3 bench.py --steps 40 --warmup 30 --age-steps 0 --no-cpu-baseline $extra > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/${tag}_pmc_sq -o p -- python3 bench.py --steps 40 --warmup 30 --age-steps 0 --no-cpu-baseline $extra > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_LDS -d gpurun_out/${tag}_pmc_lds -o p -- python3 bench.py --steps 40 --warmup 30 --age-steps 0 --no-cpu-baseline $extra > /dev/null 2>&1
# the search kernel's instruction mix (scripts/search_phases.py): scalar / memory instruction counts next to SQ_INSTS_VALU / SQ_INSTS_LDS of the pass above
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAVES -d gpurun_out/${tag}_pmc_mix -o p -- python3 bench.py --steps 40 --warmup 30 --age-steps 0 --no-cpu-baseline $extra > /dev/null 2>&1
python3 scripts/prof_summary.py gpurun_out/${tag}_trace/t_results.db gpurun_out/${tag}_bench_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps $steps --warmup 30 --age-steps ${3:-1000} --no-cpu-baseline $extra" "$(cat gpurun_out/${tag}_trace.json | head -c 400)"
python3 scripts/pmc_bench_summary.py gpurun_out/${tag}_pmc_summary.json gpurun_out/${tag}_pmc_fetch/p_results.db gpurun_out/${tag}_pmc_write/p_results.db gpurun_out/${tag}_pmc_sq/p_results.db gpurun_out/${tag}_pmc_lds/p_results.db gpurun_out/${tag}_pmc_mix/p_results.db
# the rocpd databases are hundreds of MB: only the summaries travel back (gpurun merges at most 64 MiB of gpurun_out/)
rm -rf gpurun_out/${tag}_trace gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write gpurun_out/${tag}_pmc_sq gpurun_out/${tag}_pmc_lds gpurun_out/${tag}_pmc_mix
