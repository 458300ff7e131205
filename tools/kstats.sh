#!/bin/bash
# per-kernel table (calls, median / min / mean us, second half of the launches) of one bench.py run under rocprofv3
# usage: tools/kstats.sh TAG <bench args>
tag=$1; shift
bash tools/prof1.sh $tag "$@" > gpurun_out/kstats_$tag.log 2>&1
python3 - <<PY
import csv,glob,statistics
f=sorted(glob.glob('gpurun_out/prof_$tag/run*/*kernel_trace.csv'))[-1]
d={}
for r in csv.DictReader(open(f)):
    d.setdefault(r['Kernel_Name'][:70],[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
tot=0
for n,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    if not n.startswith(('lgd','void lgd')): continue
    w=v[len(v)//2:]
    print('%-70s n=%5d median %9.1f us  min %9.1f  mean %9.1f'%(n,len(v),statistics.median(w),min(w),statistics.mean(w)))
PY
