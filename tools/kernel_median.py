"""Median duration per (kernel, grid) of a rocprofv3 --kernel-trace csv.  Usage: kernel_median.py <dir> [name filter]"""
import collections, csv, glob, re, sys

flt = sys.argv[2] if len(sys.argv) > 2 else ""
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if flt in n:
            d[((re.search(r"(\w+_kernel(<[^>]*>)?)", n) or re.search(r"(\S+)", n)).group(1)[-60:], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append(
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v = sorted(v)
    print(f"{k[0]:60s} grid {k[1]:>8s} {k[2]:>4s} {k[3]:>3s}  n {len(v):4d}  median {v[len(v) // 2]:8.1f} us  min {v[0]:8.1f}")
