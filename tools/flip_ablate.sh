#!/bin/bash
# Tuning aid: ablations of pair_flip_kernel (FL_NOZ: no Z gather, FL_NOQ: no q loads, FL_NOCORR: no corrections)
cd "$GRAFT_REPO_ROOT" || exit 1
for ex in "" "-DFL_NOCORR" "-DFL_NOZ" "-DFL_NOQ" "-DFL_NOZ -DFL_NOQ" "-DFL_NOZ -DFL_NOQ -DFL_NOCORR"; do
  touch lpformer_amd/csrc/pair_flip.hip
  make -C lpformer_amd/csrc EXTRA="$ex" > /dev/null 2>&1
  echo "[$ex] $(LPF_CFG=${1:-collab} timeout 300 python3 tools/fused_variants.py 2>&1 | tail -1 | grep -o '"pair_attention_fused": [0-9.]*')"
done
