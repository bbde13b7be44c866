#!/bin/bash
# Tuning aid: pipelined bench throughput under different launch shapes of the D = 128 flip kernel
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "-DFL_CFG128=32,512,1,1" "-DFL_CFG128=32,512,1,2" "-DFL_CFG128=32,256,1,2" "-DFL_CFG128=32,256,0,4"; do
  touch lpformer_amd/csrc/pair_flip.hip
  make -C lpformer_amd/csrc EXTRA="$cfg" > /dev/null 2>&1
  echo "[$cfg] $(timeout 600 python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-bf16 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["kernels"].items()})')"
done
