"""Per-kernel means of rocprofv3 --pmc counter_collection CSVs.  Usage: pmc_summary.py <dir> [name filter ...]"""
import csv
import glob
import sys
from collections import defaultdict

d, filt = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if filt and not any(x in n for x in filt):
            continue
        acc[n[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in acc.items():
    print(n)
    for c, v in sorted(cs.items()):
        v = v[len(v) // 2:]  # skip warm-up launches
        print(f"    {c:28s} {sum(v) / len(v):16.1f}  (n={len(v)})")
