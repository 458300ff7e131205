#!/bin/bash
# (the LGD_SKIP_TP / LGD_COND switches this matrix used lived in a throw-away build; with the placement-probe
# build it shows the placement per layout as the product launches it)
export LOUDSCAN_LIB=$PWD/loudgain_amd/csrc/libloudscan_hip_hwid.so
for c in 1 3 6 8; do
  echo "== ch $c"
  PROBE_CH=$c python tools/hwid_probe.py 2>&1 | grep "scan_only\|waves per SIMD"
done
