#!/bin/bash
# what the dense tail's weight stream costs the PIPELINED step (timing build -DTC_NOWLOAD, wrong results): nothing --
# 0.1804 against 0.1797 ms per step, the launch alone 52.3 against 55.2 us.  (The selection's ablation builds are for
# tools/select_variants.sh only: they change what is selected, so the kernels behind them do other work -- and the one
# without the look-back never ends.)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() {  # file, flags
  touch lpformer_amd/csrc/$1
  make -C lpformer_amd/csrc EXTRA="$2" > /dev/null 2>&1 || { echo "[$2] build failed"; return; }
  timeout 300 python3 bench.py --gpus 1 --steps 40 --warmup 5 --rows on --launch plan --weights random --no-bf16 --no-cpu-baseline > gpurun_out/pabl_bench.log 2>&1
  echo "[$1 $2] $(tail -1 gpurun_out/pabl_bench.log | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in list(d.get('kernels',{}).items())[:5]})
except Exception as e: print('failed', e)")"
  touch lpformer_amd/csrc/$1
}
run tail_chain.hip ""
run tail_chain.hip "-DTC_NOWLOAD"
run tail_chain.hip ""
make -C lpformer_amd/csrc > /dev/null 2>&1
