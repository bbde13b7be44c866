#!/bin/bash
# driver-style windows (20 steps) against the number of streams the steps rotate over (recorded steps)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for st in 8 10 12 16 20 24; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --launch plan --rows on --weights random --no-bf16 --no-cpu-baseline --no-kernel-timing --streams $st > gpurun_out/streams_bench.log 2>&1
  echo "streams $st: $(tail -1 gpurun_out/streams_bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('ms_per_step_repeats'))")"
done
