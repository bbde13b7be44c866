#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/full_tests.log 2>&1
tail -3 gpurun_out/full_tests.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/full_bench.log 2>&1
tail -1 gpurun_out/full_bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print(d['value'], d['ms_per_step'], c['launch'][:40], c['launch_probe_ms_per_step'], c.get('attention_form_probe_ms_per_step'))"
