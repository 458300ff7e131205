#!/bin/bash
# first GPU sweep: parity, numerics probe, bench variants
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
timeout -k 10 200 python tools/gpu_err_probe.py 48000 noise > gpurun_out/probe.log 2>&1
timeout -k 10 200 python tools/gpu_err_probe.py 48000 lf >> gpurun_out/probe.log 2>&1
timeout -k 10 200 python tools/gpu_err_probe.py 192000 lf >> gpurun_out/probe.log 2>&1
grep "^fs" gpurun_out/probe.log
for c in 25 50 75; do for w in 4 8 12; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --chunk $c --waves-per-cu $w --no-cpu-baseline >> gpurun_out/bench_sweep.log 2>&1 || echo "bench fail c=$c w=$w"
done; done
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --workload c3 --chunk 25 --no-cpu-baseline >> gpurun_out/bench_sweep.log 2>&1
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --workload c3 --chunk 50 --no-cpu-baseline >> gpurun_out/bench_sweep.log 2>&1
timeout -k 10 400 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_default.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/bench_sweep.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['config']['workload'][:3], 'chunk',d['config']['chunk'],'segs',d['config']['segments'],'value',d['value'],'ms/step',d['ms_per_step'],'kern ms',d['roofline']['kernel_ms_mean'],'frac',d['roofline']['frac'])
PY
tail -3 gpurun_out/bench_default.log
