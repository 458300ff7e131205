#!/bin/bash
# tools/variant_probe.sh OUT v1 v2 ... -- tools/tp_probe.py for library variants loudgain_amd/csrc/variants/lib_<v>.so
# (extra arguments for tp_probe.py in $PROBE_ARGS)
out=$1; shift
mkdir -p $(dirname $out); : > $out
for v in "$@"; do
  echo "== $v" >> $out
  LOUDSCAN_LIB=$PWD/loudgain_amd/csrc/variants/lib_$v.so timeout -k 10 300 python tools/tp_probe.py $PROBE_ARGS >> $out 2>&1 || exit 1
done
