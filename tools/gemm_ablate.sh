#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for ex in "" "-DGR_NOLOAD" "-DGR_NOSTORE" "-DGR_NOMFMA" "-DGR_NOLOAD -DGR_NOSTORE"; do
  touch lpformer_amd/csrc/gemm_f32.hip
  make -C lpformer_amd/csrc EXTRA="$ex" > /dev/null 2>&1 || echo "build failed"
  echo "[$ex] $(LPF_FUSED=0 timeout 300 python3 tools/enc_time.py 2>&1 | tail -1)"
done
