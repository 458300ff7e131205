#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench command
set -o pipefail
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_${1:-c2}
rm -rf $OUT; mkdir -p $OUT
shift
# the bench command of ONE workload (python bench.py --workload ...): the CPU baseline leg (no
# kernels), the c3 object and the host-buffer figures of the default line are skipped
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/bench.py --no-cpu-baseline --no-c3 --no-h2d "$@" > $OUT/bench.log 2>&1
echo "rc=$?"
tail -2 $OUT/bench.log | cut -c1-400
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cp {} '$OUT'/kernel_stats.csv; cat {}'
