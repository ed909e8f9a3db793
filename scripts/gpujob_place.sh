#!/bin/bash
# per-place dynamic instruction counts + an un-profiled line of the PMC passes' workload (nodes per launch for the per-node figures)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/place_pmc
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES -d gpurun_out/place_pmc -o p -- python3 scripts/place_pmc.py run > gpurun_out/place_pmc.log 2>&1
echo "place_pmc run: $?"
find gpurun_out/place_pmc -name "*.db" | head
db=$(find gpurun_out/place_pmc -name "*_results.db" | head -1)
python3 scripts/place_pmc.py summary "$db" gpurun_out/r05_place_pmc.json
rm -rf gpurun_out/place_pmc
timeout 600 python3 bench.py --steps 40 --warmup 30 --age-steps 0 --no-cpu-baseline > gpurun_out/r05_pmc_workload_line.json 2> gpurun_out/r05_pmc_workload_line.err
echo "workload line: $?"
head -c 600 gpurun_out/r05_pmc_workload_line.json
