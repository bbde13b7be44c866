#!/bin/bash
# Tuning aid: pipelined bench throughput vs number of HIP hardware queues / streams
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo "[$1 streams=$2] $(env $1 timeout 600 python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-bf16 --streams $2 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"; }
run GPU_MAX_HW_QUEUES=4 6
run GPU_MAX_HW_QUEUES=8 6
run GPU_MAX_HW_QUEUES=8 8
run GPU_MAX_HW_QUEUES=2 6
run GPU_MAX_HW_QUEUES=8 12
