"""All kernels between the starts of scan n and scan n+3 from a rocprofv3 kernel trace (times in us)."""
import csv, glob, os, sys
fs = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(fs[-1])), key=lambda r: int(r['Start_Timestamp']))
scan = [r for r in rows if 'lgd_scan_kernel' in r['Kernel_Name']]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
t0 = int(scan[n]['Start_Timestamp']); t1 = int(scan[n + 3]['End_Timestamp'])
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if e >= t0 and s <= t1:
        print("%-42s %8.1f %8.1f  dur %7.1f q%s lds %s wg %s grid %s" % (r['Kernel_Name'][:42], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3,
              r['Queue_Id'], r['LDS_Block_Size'], r['Workgroup_Size_X'], r['Grid_Size_X']))
