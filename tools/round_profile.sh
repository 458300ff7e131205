#!/bin/bash
# Collects the evidence committed under profiles/: default bench line, rocprofv3
# kernel-trace stats of the same command, PMC passes (counters only).
set -o pipefail
TAG=${1:-r01}
mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/${TAG}_bench_c2.json 2> gpurun_out/${TAG}_bench_c2.err; echo "bench c2 rc=$?"
timeout -k 10 600 python bench.py --workload c3 --no-cpu-baseline > gpurun_out/${TAG}_bench_c3.json 2> gpurun_out/${TAG}_bench_c3.err; echo "bench c3 rc=$?"
# kernel durations: --serial (no overlap of consecutive scans) is the mode the roofline
# figure is quoted in; the default (pipelined) command is traced too
bash tools/prof1.sh ${TAG}_c2 --serial > gpurun_out/${TAG}_prof_c2.log 2>&1
bash tools/prof1.sh ${TAG}_c2_pipelined > gpurun_out/${TAG}_prof_c2_pipelined.log 2>&1
bash tools/prof1.sh ${TAG}_c3 --workload c3 --serial > gpurun_out/${TAG}_prof_c3.log 2>&1
bash tools/pmc.sh ${TAG}_c2 --serial > gpurun_out/${TAG}_pmc_c2.log 2>&1
cut -c1-600 gpurun_out/${TAG}_bench_c2.json; echo; cut -c1-300 gpurun_out/${TAG}_bench_c3.json; echo
head -4 gpurun_out/prof_${TAG}_c2/kernel_stats.csv | cut -c1-200
grep -A28 "== scan" gpurun_out/pmc_${TAG}_c2/summary.txt | head -30
