#!/bin/bash
# counters of the mono scan kernel with and without the true-peak kernels between scans
set -o pipefail
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_mono
rm -rf $OUT; mkdir -p $OUT
export PROBE_WARM=6 PROBE_RUN=6 PROBE_CH=${PROBE_CH:-1}
PASSES=(
"SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY"
"TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum"
"SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 $REPO/tools/mono_probe.py TF > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "lgd_scan_kernel" not in k: continue
        acc[k.split("(")[0][-40:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print("==", k)
    for c in sorted(acc[k]):
        v = acc[k][c]; print("  %-28s mean %.6g min %.6g max %.6g (n=%d)" % (c, sum(v)/len(v), min(v), max(v), len(v)))
PY
