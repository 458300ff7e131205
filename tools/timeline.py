"""Print the kernel timeline around two consecutive scan kernels from a rocprofv3 kernel trace."""
import csv, glob, os, statistics as st, sys
fs = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(fs[-1]))]
scan = sorted([r for r in rows if 'lgd_scan_kernel' in r['Kernel_Name']], key=lambda r: int(r['Start_Timestamp']))
gaps = [int(b['Start_Timestamp']) - int(a['End_Timestamp']) for a, b in zip(scan[20:], scan[21:])]
durs = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in scan[20:]]
print("scan kernels", len(scan), "mean dur", st.mean(durs), "mean gap", st.mean(gaps), "min", min(gaps), "max", max(gaps))
others = [r for r in rows if 'lgd_' in r['Kernel_Name'] and 'scan' not in r['Kernel_Name']]
a, b = scan[40], scan[41]
t0 = int(a['Start_Timestamp'])
print("scan40: 0 ..", int(a['End_Timestamp']) - t0, " scan41:", int(b['Start_Timestamp']) - t0, "..", int(b['End_Timestamp']) - t0)
for r in others:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if e > t0 and s < int(b['End_Timestamp']):
        print(r['Kernel_Name'][:24], s - t0, e - t0, "q", r['Queue_Id'], "vgpr", r['VGPR_Count'], "lds", r['LDS_Block_Size'])
