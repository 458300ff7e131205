#!/bin/bash
# PMC passes over the bench command (counters only: no --kernel-trace/--stats mix, no sys-trace)
# usage: tools/pmc.sh <tag> [bench args...]
set -o pipefail
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-c2}; shift
OUT=$REPO/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
PASSES=(
"SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
"SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM"
"SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"
"FETCH_SIZE"
"WRITE_SIZE"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-c3 --no-h2d --settle 2 --launches 4 "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 $REPO/tools/pmc_parse.py $OUT | tee $OUT/summary.txt
