cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
BARGS="--gpus 1 --steps 20 --warmup 5 --weights random --no-cpu-baseline --no-bf16 --rows on --no-kernel-timing"
run() { python3 bench.py $BARGS "$@" > gpurun_out/ab_bench.log 2>&1; echo "[$*] $(tail -1 gpurun_out/ab_bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('ms_per_step_repeats'), d['config']['launch'][:10])" 2>&1 | tail -1)"; }
for rep in 1 2; do
run
run --select4-threads 512
run --select4-threads 4608
run --select4-threads 1024
run --streams 3
run --streams 6
done 2>&1 | tee gpurun_out/r06_ab11.txt
