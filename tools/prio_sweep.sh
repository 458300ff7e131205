#!/bin/bash
# wave-priority variants of the scan kernel: build with e.g.
#   hipcc ... -DLGD_PRIO_A=0 -DLGD_PRIO_C=1 -DLGD_PRIO_SCAN=3 -DLGD_PRIO_STAGE=3 -shared -o libloudscan_hip_x.so ...
# and list the suffixes here; serial kernel timing of each
for v in "" ${VARIANTS}; do
  export LOUDSCAN_LIB=$PWD/loudgain_amd/csrc/libloudscan_hip$v.so
  python bench.py --no-cpu-baseline --serial --steps 60 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('variant [$v] kernel_ms', d['roofline']['kernel_ms_mean'], 'min', d['roofline']['kernel_ms_min'])"
done
