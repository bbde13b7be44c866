#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -m gpu -q -k "pair_rows or sweep" 2>&1 | tail -3
for ex in "" "-DTC_NO_DEAL"; do
  touch lpformer_amd/csrc/tail_chain.hip
  make -C lpformer_amd/csrc EXTRA="$ex" > /dev/null 2>&1 || { echo "[$ex] build failed"; continue; }
  for cfgargs in "" "--config cora"; do
  for i in 1 2; do
    echo "[$ex] $cfgargs"
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --rows on $cfgargs > gpurun_out/r04n_bench.log 2>&1
    tail -1 gpurun_out/r04n_bench.log | python3 tools/all_configs_fmt.py | head -2 | cut -c1-100
  done
  done
done
touch lpformer_amd/csrc/tail_chain.hip; make -C lpformer_amd/csrc > /dev/null 2>&1
