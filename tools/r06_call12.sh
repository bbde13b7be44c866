cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_select4.py -x -q 2>&1 | tail -3
BARGS="--gpus 1 --steps 20 --warmup 5 --weights random --no-cpu-baseline --no-bf16 --no-kernel-timing"
run() { python3 bench.py $BARGS "$@" > gpurun_out/ab_bench.log 2>&1; echo "[$*] $(tail -1 gpurun_out/ab_bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('ms_per_step_repeats'), d['config']['launch'][:10])" 2>&1 | tail -1)"; }
for rep in 1 2 3; do run --config collab; run --config collab --select4-threads 512; run --config collab --select4-threads 256; done
for cfg in ppa citation2 ddi cora; do for rep in 1 2; do run --config $cfg; run --config $cfg --select4-threads 512; run --config $cfg --select4-threads 256; run --config $cfg --select4-threads 4608; done; done
