#!/bin/bash
# dense tail: 128 pairs per workgroup (one workgroup of 16 wavefronts per CU) against 64 (two of 8)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for ex in "" "-DLPF_TC_GROUPS=8"; do
  touch lpformer_amd/csrc/tail_chain.hip
  make -C lpformer_amd/csrc EXTRA="$ex" > /dev/null 2>&1 || { echo "[$ex] build failed"; continue; }
  for i in 1 2; do
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --rows on --launch plan --weights random --no-bf16 --no-cpu-baseline > gpurun_out/tailgroups_bench.log 2>&1
    echo "[$ex] $(tail -1 gpurun_out/tailgroups_bench.log | python3 tools/all_configs_fmt.py | head -2 | cut -c1-70 | tr '\n' ' ')"
  done
done
touch lpformer_amd/csrc/tail_chain.hip; make -C lpformer_amd/csrc > /dev/null 2>&1
