#!/bin/bash
# A/B of wave-priority builds (see prio_sweep.sh): pipelined step time and serial kernel time, C2 and C3
for v in "" ${VARIANTS}; do
  export LOUDSCAN_LIB=$PWD/loudgain_amd/csrc/libloudscan_hip$v.so
  for w in c2 c3; do
  python bench.py --no-cpu-baseline --workload $w | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('variant [$v] $w pipelined ms/step', d['ms_per_step'], 'serial kernel', d['roofline']['kernel_ms_mean'])"
  done
done
