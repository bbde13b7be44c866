#!/bin/bash
# Tuning aid: ablations of gcn_fused_kernel (GF_NOMFMA: no product, GF_NOGATHER: no neighbour loads)
cd "$GRAFT_REPO_ROOT" || exit 1
for ex in "" "-DGF_NOMFMA" "-DGF_NOGATHER" "$@"; do
  touch lpformer_amd/csrc/gcn_fused.hip
  make -C lpformer_amd/csrc EXTRA="$ex" > /dev/null 2>&1 || echo "build failed"
  echo "[$ex] $(timeout 300 python3 tools/enc_time.py 2>&1 | tail -1)"
done
