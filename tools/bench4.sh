#!/bin/bash
# C2 / C3 / C3 adversarial / C3 unpruned, serial launches, one line each: tools/bench4.sh outdir [steps]
out=${1:-gpurun_out/bench4}; steps=${2:-300}
mkdir -p $out
timeout -k 10 200 python bench.py --workload c2 --serial --steps $steps --no-cpu-baseline > $out/c2.json 2>$out/c2.err &&
timeout -k 10 200 python bench.py --workload c3 --serial --steps $steps --no-cpu-baseline > $out/c3.json 2>$out/c3.err &&
timeout -k 10 200 python bench.py --workload c3 --serial --steps $steps --no-cpu-baseline --material adversarial > $out/c3_adv.json 2>$out/c3adv.err &&
timeout -k 10 200 python bench.py --workload c3 --serial --steps $steps --no-cpu-baseline --no-tp-prune > $out/c3_noprune.json 2>$out/c3np.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.load(open(f)); print(f, "ms/step", d["ms_per_step"], "kernel", d["roofline"]["kernel_ms_mean"], "min", d["roofline"]["kernel_ms_min"], "frac", d["roofline"]["frac"], "peak", d["result"]["peak"])
    except Exception as e: print(f, "ERR", e)
PY
