#!/bin/bash
# effective clock of the scan kernel: GRBM_GUI_ACTIVE / 8 XCDs / kernel time, for compute-only vs full
set -o pipefail
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for d in 0 1 2; do
  OUT=$REPO/gpurun_out/clk_$d; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --debug $d > $OUT/pmc.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --debug $d > $OUT/kt.log 2>&1
  python3 - $OUT $d <<'PY'
import csv, glob, sys
out, d = sys.argv[1], sys.argv[2]
vals = {}
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lgd_scan_kernel" in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
dur = None
for f in glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lgd_scan_kernel" in r["Name"]:
            dur = float(r["AverageNs"])
g = sum(vals["GRBM_GUI_ACTIVE"]) / len(vals["GRBM_GUI_ACTIVE"])
print("debug", d, "kernel avg ns", dur, "GRBM_GUI_ACTIVE", g, "-> clock GHz ~", g / 8 / dur,
      "| VALU active quad-cycles", sum(vals["SQ_ACTIVE_INST_VALU"]) / len(vals["SQ_ACTIVE_INST_VALU"]))
PY
done
