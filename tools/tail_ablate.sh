#!/bin/bash
# Tuning aid: tail_chain_kernel without its weight stream (-DTC_NOWLOAD): what the k-group prefetch leaves exposed
cd "$GRAFT_REPO_ROOT" || exit 1
for ex in "" "-DTC_NOWLOAD"; do
  touch lpformer_amd/csrc/tail_chain.hip
  make -C lpformer_amd/csrc EXTRA="$ex" > /dev/null 2>&1 || echo "build failed"
  echo "[$ex] $(LPF_CFG=${1:-collab} timeout 300 python3 tools/fused_variants.py 2>&1 | tail -1 | grep -o '"tail_chain": [0-9.]*')"
done
