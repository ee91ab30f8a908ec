"""Averages rocprofv3 --pmc counter_collection CSV rows per kernel: python tools/pmc_summary.py file.csv [name-filter]"""
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
flt = sys.argv[2] if len(sys.argv) > 2 else "dpa::"
for r in csv.DictReader(open(sys.argv[1])):
    if flt in r["Kernel_Name"]:
        name = r["Kernel_Name"].split("(")[0].split("::")[-1] if "anonymous" not in r["Kernel_Name"] else r["Kernel_Name"].split("::")[-1].split("(")[0]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {sum(v)/len(v):16.0f}   (n={len(v)})")
