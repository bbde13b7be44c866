#!/bin/bash
# Same-lease A/B of pair_rows / tail build variants under the pipelined bench (driver form, no extra legs) + serial kernel
# times.  Usage: tools/r06_ab.sh  (variants from $VARIANTS, one EXTRA string per line; "" = the default build)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
BARGS="--gpus 1 --steps 20 --warmup 5 --weights random --no-cpu-baseline --no-bf16 --rows on ${BENCH_ARGS:-}"
while IFS= read -r ex; do
  for f in ${TOUCH:-pair_rows tail_chain}; do touch lpformer_amd/csrc/$f.hip; done
  make -C lpformer_amd/csrc -j8 EXTRA="$ex" > gpurun_out/ab_build.log 2>&1 || { echo "[$ex] build failed"; tail -5 gpurun_out/ab_build.log; continue; }
  for i in 1 2; do
    python3 bench.py $BARGS > gpurun_out/ab_bench.log 2>&1
    echo "[$ex] $(tail -1 gpurun_out/ab_bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
k={a:b['ms_per_step'] for a,b in list(d.get('kernels',{}).items())[:5]}
print(d['value'], d['ms_per_step'], d.get('ms_per_step_repeats'), c['launch'][:10], k)" 2>&1 | tail -1)"
  done
  [ -n "${SERIAL:-}" ] && echo "[$ex] serial: $(timeout 300 python3 tools/fused_variants.py 2>&1 | tail -1 | cut -c1-330)"
done <<< "${VARIANTS:-}"
for f in ${TOUCH:-pair_rows tail_chain}; do touch lpformer_amd/csrc/$f.hip; done
make -C lpformer_amd/csrc -j8 > /dev/null 2>&1
