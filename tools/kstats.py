"""Print the per-kernel averages of a rocprofv3 --stats run.  Usage: kstats.py <dir> [name fragment ...]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if len(sys.argv) < 3 or any(k in r["Name"] for k in sys.argv[2:]):
        print("%9.1f us x %5s  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:80]))
