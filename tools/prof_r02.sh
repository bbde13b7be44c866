#!/bin/bash
# per-kernel timing variants + rocprofv3 kernel trace of the bench (run on the GPU box from the repo root)
mkdir -p gpurun_out
ls -la lpformer_amd/liblpformer_hip.so
for v in ${VARIANTS:-0 4 7}; do LPF_FUSED_DBG=$v timeout 200 python tools/fused_variants.py 2>/dev/null | tail -1; done > gpurun_out/r02_variants.txt
cat gpurun_out/r02_variants.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --no-cpu-baseline --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/r02_bench_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02_bench_rocprof.err
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_r02 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r02_kernel_stats.csv 2>/dev/null
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r02_kernel_stats.csv')))
for r in rows[:14]:
    print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
rm -rf gpurun_out/prof_r02
