#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for ex in "-DFL_OCC=3" "-DFL_OCC=3 -DFL_NOCORR"; do
  touch lpformer_amd/csrc/pair_flip.hip
  make -C lpformer_amd/csrc EXTRA="$ex" > /dev/null 2>&1
  for cfg in collab ddi; do
  echo "$ex $cfg $(LPF_CFG=$cfg timeout 300 python3 tools/fused_variants.py 2>&1 | tail -1 | cut -c1-300)"
  done
done
