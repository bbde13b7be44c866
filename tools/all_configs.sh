#!/bin/bash
# Step time and kernel shares of every synthetic config (GPU box, repo root) -> gpurun_out/<tag>_all_configs.txt
TAG=${1:-r02}
OUT=gpurun_out/${TAG}_all_configs.txt
: > $OUT
for c in collab ddi cora ppa citation2; do
  echo "== $c" >> $OUT
  timeout 900 python3 bench.py --config $c --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print(d["value"], d["ms_per_step"], d.get("ms_per_step_repeats"), d["config"]["workload"])
print({k: v['ms_per_step'] for k, v in list(d.get('kernels', {}).items())[:6]})
bm = d.get("bf16_mode") or {}; print("bf16:", bm.get("value"), bm.get("max_abs_logit_diff_vs_f32"))
" >> $OUT 2>&1
done
cat $OUT
