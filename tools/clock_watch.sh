#!/bin/bash
# shader clock / power while the bench runs pipelined vs serial (rocm-smi samples during a long run)
for mode in "" "--serial"; do
  python bench.py --no-cpu-baseline --steps 12000 --warmup 20 $mode > /tmp/clk_bench.json 2>/dev/null &
  BP=$!
  sleep 2.5
  for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.4; done
  wait $BP
  python -c "
import json; d=json.load(open('/tmp/clk_bench.json')); print('mode [$mode] ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms_mean'])"
done
