cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
BARGS="--gpus 1 --steps 20 --warmup 5 --weights random --no-cpu-baseline --no-bf16"
for cfg in ppa citation2; do
  python3 bench.py $BARGS --config $cfg > gpurun_out/ab_bench.log 2>&1
  echo "[$cfg] $(tail -1 gpurun_out/ab_bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
k={a:b['ms_per_step'] for a,b in list(d.get('kernels',{}).items())[:7]}
print(d['value'], d['ms_per_step'], d.get('ms_per_step_repeats'), c['launch'][:10], k)" 2>&1 | tail -1)"
done 2>&1 | tee gpurun_out/r06_ab9.txt
bash tools/select4_stamps.sh "" ppa 2>&1 | tee gpurun_out/r06_select4_stamps_ppa.txt
bash tools/select4_stamps.sh "" citation2 2>&1 | tee gpurun_out/r06_select4_stamps_citation2.txt
