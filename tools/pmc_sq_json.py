"""profiles/rNN_pmc_sq_<config>.json from the SQ counter passes of tools/pmc_kernel.sh (gpurun_out/pmc_<kernel>/sq1.txt ...):
per kernel the mean counters per launch -- bench.py reads SQ_WAIT_ANY / SQ_WAVE_CYCLES (`wave_wait_frac`) and SQ_INSTS_VALU.
Usage: pmc_sq_json.py <out.json> <commit> <config> <pmc_dir> [<pmc_dir> ...]"""
import glob
import json
import re
import sys

out = {"note": "rocprofv3 --pmc SQ_* passes (tools/pmc_kernel.sh: separate passes, kernel-trace only) on the " + sys.argv[3] +
               "-like bench workload, serial single-stream steps, mean per launch after warm-up",
       "commit": sys.argv[2], "kernels": {}}
for d in sys.argv[4:]:
    for f in sorted(glob.glob(d + "/*.txt")):
        name = None
        for line in open(f):
            m = re.match(r"\s+(\S+)\s+([0-9.]+)\s+\(n=", line)
            if m and name:
                out["kernels"].setdefault(name, {})[m.group(1)] = float(m.group(2))
            elif line.strip() and not line.startswith(" "):
                # "void (anonymous namespace)::pair_rows_kernel<16, 1024, ..." -> pair_rows_kernel
                k = re.search(r"::(\w+)", line)
                name = k.group(1) if k else line.strip().split("<")[0]
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, v in out["kernels"].items():
    w, c = v.get("SQ_WAIT_ANY"), v.get("SQ_WAVE_CYCLES")
    print(k, "wave_wait_frac", None if not (w and c) else round(w / c, 4), "VALU insts", v.get("SQ_INSTS_VALU"))
