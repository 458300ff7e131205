#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
rm -f gpurun_out/bench_sweep.log
run() { timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" >> gpurun_out/bench_sweep.log 2>&1 || echo "bench fail $*"; }
for c in 25 40 48 50 60 75; do for w in 8 12 16; do run --chunk $c --waves-per-cu $w; done; done
run --chunk 75 --waves-per-cu 8 --debug 1
run --chunk 50 --waves-per-cu 12 --debug 1
run --chunk 48 --waves-per-cu 12 --debug 1
for c in 48 50 75; do run --chunk $c --waves-per-cu 8 --workload c3; run --chunk $c --waves-per-cu 12 --workload c3; done
python - <<'PY'
import json
for l in open('gpurun_out/bench_sweep.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['config']['workload'][:3], 'chunk',d['config']['chunk'],'segs',d['config']['segments'],'value',d['value'],'ms/step',d['ms_per_step'],'kern ms',d['roofline']['kernel_ms_mean'],'min',d['roofline']['kernel_ms_min'],'frac',d['roofline']['frac'])
PY
