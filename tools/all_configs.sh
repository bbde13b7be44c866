#!/bin/bash
# Step time and kernel shares of every synthetic config (GPU box, repo root) -> gpurun_out/<tag>_all_configs.txt
TAG=${1:-r02}
OUT=gpurun_out/${TAG}_all_configs.txt
: > $OUT
for c in collab ddi cora ppa citation2; do
  echo "== $c" >> $OUT
  timeout 900 python3 bench.py --config $c --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 | python3 tools/all_configs_fmt.py >> $OUT 2>&1
done
cat $OUT
