#!/bin/bash
# The S16-resident PCM part of the round's evidence (see tools/round_profile.sh): kernel traces, a PMC pass (traffic), the
# benches of configs 4 / 5 and the layout sweep with the samples as interleaved int16 in HBM.
set -o pipefail
TAG=${1:-r03}
mkdir -p gpurun_out
bash tools/prof1.sh ${TAG}_c2_s16 --workload c2 --serial --pcm s16 > gpurun_out/${TAG}_prof_c2_s16.log 2>&1
bash tools/prof1.sh ${TAG}_c3_s16 --workload c3 --serial --pcm s16 > gpurun_out/${TAG}_prof_c3_s16.log 2>&1
bash tools/pmc.sh ${TAG}_c2_s16 --workload c2 --serial --pcm s16 > gpurun_out/${TAG}_pmc_c2_s16.log 2>&1
for w in c4 c5; do
  timeout -k 10 600 python bench.py --workload $w --pcm s16 --no-cpu-baseline > gpurun_out/${TAG}_bench_${w}_s16.json 2> gpurun_out/${TAG}_bench_${w}_s16.err; echo "bench $w s16 rc=$?"
done
timeout -k 10 900 python tools/rate_sweep.py --pcm s16 > gpurun_out/${TAG}_rate_sweep_s16.txt 2>/dev/null
for w in c2_s16 c3_s16; do head -5 gpurun_out/prof_${TAG}_$w/kernel_stats.csv | cut -c1-160; done
cat gpurun_out/pmc_${TAG}_c2_s16/summary.txt | grep -A30 "== scan" | grep -E "FETCH|WRITE|INSTS_VALU "
# what goes to profiles/: prof_${TAG}_c{2,3}_s16/kernel_stats.csv -> ${TAG}_c{2,3}_s16_kernel_stats.csv, pmc_${TAG}_c2_s16/summary.txt ->
#   ${TAG}_c2_s16_pmc_summary.txt, ${TAG}_bench_c{4,5}_s16.json, ${TAG}_rate_sweep_s16.txt; then python tools/traffic_from_pmc.py
