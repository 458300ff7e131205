#!/bin/bash
# marginal cost of the kernel's phases: floor-measurement build, serial kernel timing
# debug bits: 1 no loads, 2 no arithmetic, 4 skip phase A, 8 skip the wave scan, 16 skip phase C
export LOUDSCAN_LIB=$PWD/loudgain_amd/csrc/libloudscan_hip_dbg.so
for d in 0 4 8 16 12 20 24 28 1 2 ${EXTRA}; do
  python bench.py --no-cpu-baseline --serial --steps 40 --debug $d "$@" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('debug $d kernel_ms', d['roofline']['kernel_ms_mean'], 'min', d['roofline']['kernel_ms_min'])"
done
