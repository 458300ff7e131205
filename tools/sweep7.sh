#!/bin/bash
set -o pipefail
rm -f gpurun_out/bench_sweep.log
run() { timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" >> gpurun_out/bench_sweep.log 2>&1 || echo "bench fail $*"; echo "ARGS $*" >> gpurun_out/bench_sweep.log; }
for d in 0 1 5 9 17 29 2; do run --chunk 75 --waves-per-cu 8 --debug $d; done
run --chunk 50 --waves-per-cu 8
run --chunk 50 --waves-per-cu 12
run --chunk 25 --waves-per-cu 16
python - <<'PY'
import json
last=None
for l in open('gpurun_out/bench_sweep.log'):
    if l.startswith('{'):
        last=json.loads(l)
    elif l.startswith('ARGS') and last:
        d=last; print(l.strip()[5:], '| segs',d['config']['segments'],'kern ms',d['roofline']['kernel_ms_mean'],'min',d['roofline']['kernel_ms_min'], 'ms/step', d['ms_per_step']); last=None
PY
