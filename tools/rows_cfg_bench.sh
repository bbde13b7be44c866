#!/bin/bash
# pair_rows D = 128 launch shapes: isolated kernel time + pipelined step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
while IFS= read -r ex; do
  touch lpformer_amd/csrc/pair_rows.hip
  make -C lpformer_amd/csrc EXTRA="$ex" > /dev/null 2>&1 || { echo "[$ex] build failed"; continue; }
  for i in 1 2; do
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --rows on > gpurun_out/rowscfg_bench.log 2>&1
    echo "[$ex] $(tail -1 gpurun_out/rowscfg_bench.log | python3 tools/all_configs_fmt.py | head -2 | cut -c1-60 | tr '\n' ' ')"
  done
done <<< "${VARIANTS:-$'-DPR_CFG128=512,1,1\n-DPR_CFG128=512,0,2\n-DPR_CFG128=768,1,1\n-DPR_CFG128=512,1,1 -DPR_GRID2'}"
touch lpformer_amd/csrc/pair_rows.hip; make -C lpformer_amd/csrc > /dev/null 2>&1
