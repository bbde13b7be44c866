#!/bin/bash
# the driver's bench command and the no-flag default, timed
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
time (python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/driver_bench.log 2>&1)
tail -1 gpurun_out/driver_bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']; print(d['value'], d['ms_per_step'], c['launch'][:20], c['launch_probe_ms_per_step'], c.get('attention_form_probe_ms_per_step'), d['roofline']['kernel'], d['roofline']['frac'], d['cpu_baseline']['value'])"
time (python3 bench.py > gpurun_out/default_bench.log 2>&1)
tail -1 gpurun_out/default_bench.log | cut -c1-160
