#!/bin/bash
export LOUDSCAN_LIB=$PWD/loudgain_amd/csrc/variants/lib_HWID.so
for c in 1 3 6 8; do
  for mode in none skip "cond:2048,64" "cond:4096,64" "cond:1024,64" "cond:2048,128"; do
    unset LGD_SKIP_TP LGD_COND
    case $mode in skip) export LGD_SKIP_TP=1;; cond:*) export LGD_COND=${mode#cond:};; esac
    echo "== ch $c mode $mode"
    PROBE_CH=$c python tools/hwid_probe.py 2>&1 | grep "scan_only\|waves per SIMD"
  done
done
