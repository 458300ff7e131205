#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
rm -f gpurun_out/bench_sweep.log
for c in 25 50 75; do for w in 4 8 12 16; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 3 --chunk $c --waves-per-cu $w --no-cpu-baseline >> gpurun_out/bench_sweep.log 2>&1 || echo "bench fail c=$c w=$w"
done; done
for c in 25 50 75; do for w in 8 16; do
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --workload c3 --chunk $c --waves-per-cu $w --no-cpu-baseline >> gpurun_out/bench_sweep.log 2>&1
done; done
python - <<'PY'
import json
for l in open('gpurun_out/bench_sweep.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['config']['workload'][:3], 'chunk',d['config']['chunk'],'segs',d['config']['segments'],'value',d['value'],'ms/step',d['ms_per_step'],'kern ms',d['roofline']['kernel_ms_mean'],'min',d['roofline']['kernel_ms_min'],'frac',d['roofline']['frac'])
PY
(cd /tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/counters.txt 2>&1)
grep -c . gpurun_out/counters.txt
