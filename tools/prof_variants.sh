#!/bin/bash
# rocprofv3 kernel averages of bench.py for each library variant under loudgain_amd/csrc/variants/
# usage: tools/prof_variants.sh "<bench args>" variant...   ("base" = the product library)
args=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset LOUDSCAN_LIB; else export LOUDSCAN_LIB=$PWD/loudgain_amd/csrc/variants/lib_$v.so; fi
  bash tools/prof1.sh var_$v $args > gpurun_out/var_$v.log 2>&1
  echo "== $v"; grep -E "lgd_scan_kernel|lgd_tp_kernel" gpurun_out/prof_var_$v/kernel_stats.csv | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin): print('   %-50s avg %.1f us min %.1f' % (r[0][:50], float(r[3])/1000, float(r[5])/1000))"
done
