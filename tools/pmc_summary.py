"""Summarise rocprofv3 counter_collection / kernel_trace CSVs under a directory: per kernel name, mean of each counter.
   python tools/pmc_summary.py <dir> [substring-filter]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if flt in k:
            acc[k[:96]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print("KERNEL", k)
    for c, v in sorted(cs.items()):
        print(f"   {c:34s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
dur = defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if flt in k:
            dur[k[:96]].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
for k, v in dur.items():
    v.sort()
    print(f"DURATION {k}: n={len(v)} mean={sum(v)/len(v)/1e3:.1f} us median={v[len(v)//2]/1e3:.1f} us")
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    print("STATS", f)
    for i, line in enumerate(open(f)):
        if i < 15:
            print("  ", line.rstrip()[:200])
