"""Summarise rocprofv3 --pmc csv output: mean counter value per dispatch, per kernel."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            short = ("scan" if "lgd_scan_kernel" in k else "tp" if "lgd_tp_kernel" in k
                     else ("epi:" + k.split("(")[0][:24] if k.startswith("lgd_") else None))
            if short is None:
                continue
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print("==", k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("  %-28s mean %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
