#!/bin/bash
# kernel time vs segment length (100 ms sub-blocks per workgroup): tools/seg_sweep.sh outdir
out=${1:-gpurun_out/segsweep}; mkdir -p $out
for wl in c2 c3; do for sb in 36 24 18 12 9; do
  timeout -k 10 120 python bench.py --workload $wl --serial --steps 200 --no-cpu-baseline --seg-subblocks $sb > $out/${wl}_$sb.json 2>$out/${wl}_$sb.err || exit 1
done; done
timeout -k 10 120 python bench.py --workload c3 --serial --steps 200 --no-cpu-baseline --seg-subblocks 18 --material adversarial > $out/c3adv_18.json 2>/dev/null
timeout -k 10 120 python bench.py --workload c3 --serial --steps 200 --no-cpu-baseline --seg-subblocks 9 --material adversarial > $out/c3adv_9.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.load(open(f)); print(f, "kernel", d["roofline"]["kernel_ms_mean"], "min", d["roofline"]["kernel_ms_min"], "frac", d["roofline"]["frac"], "segs", d["config"]["segments"])
    except Exception as e: print(f, "ERR", e)
PY
