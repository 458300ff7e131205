#!/bin/bash
# rocprofv3 kernel averages of bench.py for each library variant under loudgain_amd/csrc/variants/
# usage: tools/prof_variants.sh "<bench args>" variant...   ("base" = the product library)
args=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset LOUDSCAN_LIB; else export LOUDSCAN_LIB=$PWD/loudgain_amd/csrc/variants/lib_$v.so; fi
  bash tools/prof1.sh var_$v $args > gpurun_out/var_$v.log 2>&1
  echo "== $v"; python3 - <<PY
import csv,glob,statistics
f=sorted(glob.glob('gpurun_out/prof_var_$v/run*/*kernel_trace.csv'))[-1]
d={}
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'lgd_scan_kernel' in n or 'lgd_tp_kernel' in n:
        d.setdefault(n[:48],[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
for n,v in d.items():
    v=v[len(v)//2:]   # second half: clocks and caches settled
    print('   %-48s median %.1f us  min %.1f  mean %.1f  (n=%d)'%(n,statistics.median(v),min(v),statistics.mean(v),len(v)))
PY
done
