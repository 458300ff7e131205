#!/bin/bash
# C2 vs C3 kernel time per programme material (serial launches): tools/mat_sweep.sh outdir
out=${1:-gpurun_out/matsweep}; mkdir -p $out
[ -n "$DBGLIB" ] && export LOUDSCAN_LIB=$DBGLIB
for m in silence noise steps adversarial; do for wl in c2 c3; do
  timeout -k 10 120 python bench.py --workload $wl --serial --steps 200 --no-cpu-baseline --material $m ${DBGLIB:+--debug-counters} > $out/${m}_$wl.json 2>$out/${m}_$wl.err || exit 1
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.load(open(f)); print(f, "kernel", d["roofline"]["kernel_ms_mean"], "min", d["roofline"]["kernel_ms_min"], "frac", d["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
grep -h tp_debug $out/*_c3.err || true
