#!/bin/bash
# the driver's bench command, three times: value, ms per step, window spread, probes, wall time
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for i in 1 2 3; do
  t0=$(date +%s.%N)
  python3 bench.py --gpus 1 --steps 20 --warmup 5 ${BENCH_ARGS:-} > gpurun_out/driver_bench.log 2>&1
  t1=$(date +%s.%N)
  tail -1 gpurun_out/driver_bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']; print(d['value'], d['ms_per_step'], d['ms_per_step_repeats'], c['launch'][:12], c['launch_probe_ms_per_step'], c.get('attention_form_probe_ms_per_step'))"
  python3 -c "print(\"wall %.1f s\" % ($t1 - $t0))"
done
