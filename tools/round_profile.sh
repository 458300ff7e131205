#!/bin/bash
# Collects the evidence committed under profiles/ (see profiles/README.md): bench lines, rocprofv3
# kernel-trace stats of the same commands (serial launches: the mode kernel durations are quoted in),
# PMC passes (counters only), the rate / layout sweep.
set -o pipefail
TAG=${1:-r03}
mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench default rc=$?"
for w in c3 c4 c5; do
  timeout -k 10 600 python bench.py --workload $w --no-cpu-baseline > gpurun_out/${TAG}_bench_$w.json 2> gpurun_out/${TAG}_bench_$w.err; echo "bench $w rc=$?"
done
for w in c2 c3 c5; do
  bash tools/prof1.sh ${TAG}_$w --workload $w --serial > gpurun_out/${TAG}_prof_$w.log 2>&1
done
bash tools/prof1.sh ${TAG}_c3_adversarial --workload c3 --serial --material adversarial --steps 300 > gpurun_out/${TAG}_prof_c3_adv.log 2>&1
bash tools/prof1.sh ${TAG}_c3_limited --workload c3 --serial --material limited --steps 300 > gpurun_out/${TAG}_prof_c3_lim.log 2>&1
bash tools/prof1.sh ${TAG}_c3_noise --workload c3 --serial --material noise --steps 300 > gpurun_out/${TAG}_prof_c3_noise.log 2>&1
bash tools/prof1.sh ${TAG}_c4 --workload c4 --serial --steps 12 --warmup 2 > gpurun_out/${TAG}_prof_c4.log 2>&1
bash tools/prof1.sh ${TAG}_c2_pipelined --workload c2 > gpurun_out/${TAG}_prof_c2_pipelined.log 2>&1
bash tools/pmc.sh ${TAG}_c2 --workload c2 --serial > gpurun_out/${TAG}_pmc_c2.log 2>&1
bash tools/pmc.sh ${TAG}_c3 --workload c3 --serial > gpurun_out/${TAG}_pmc_c3.log 2>&1
bash tools/pmc.sh ${TAG}_c3_adversarial --workload c3 --serial --material adversarial > gpurun_out/${TAG}_pmc_c3_adv.log 2>&1
timeout -k 10 900 python tools/rate_sweep.py > gpurun_out/${TAG}_rate_sweep.txt 2>/dev/null
timeout -k 10 600 python tools/library_bench.py --albums 40 > gpurun_out/${TAG}_library_bench.json 2> gpurun_out/${TAG}_library_bench.err; echo "library rc=$?"
for f in default c3 c4 c5; do cut -c1-400 gpurun_out/${TAG}_bench_$f.json; echo; done
for w in c2 c3 c4 c5 c3_adversarial c3_limited c3_noise; do head -5 gpurun_out/prof_${TAG}_$w/kernel_stats.csv | cut -c1-160; done
# what goes to profiles/: (cd gpurun_out && for w in c2 c3 c4 c5 c3_adversarial c3_limited c3_noise c2_pipelined; do cp prof_${TAG}_$w/kernel_stats.csv ../profiles/${TAG}_${w}_kernel_stats.csv; done;
#   for w in c2 c3 c3_adversarial; do cp pmc_${TAG}_$w/summary.txt ../profiles/${TAG}_${w}_pmc_summary.txt; done; cp ${TAG}_bench_*.json ${TAG}_rate_sweep.txt ${TAG}_library_bench.json ../profiles/)
