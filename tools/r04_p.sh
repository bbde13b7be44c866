#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for args in "--side-stream on --streams 8" "--side-stream off --streams 8" "--side-stream off --streams 12" "--side-stream on --streams 12" "--side-stream off --streams 16" "--side-stream on --streams 4"; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --launch plan --rows on --weights random --no-kernel-timing $args > gpurun_out/r04p_bench.log 2>&1
  echo "$args: $(tail -1 gpurun_out/r04p_bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('ms_per_step_windows'))")"
done
