#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
touch lpformer_amd/csrc/pair_fused.hip
make -C lpformer_amd/csrc EXTRA=-DLPF_FUSED_STAMPS > /dev/null 2>&1
echo "== ddi stamps"
LPF_CFG=ddi timeout 600 python3 tools/fused_stamps.py 2>&1 | tail -9
for dbg in 0 64 4; do
  LPF_CFG=ddi LPF_FUSED_DBG=$dbg timeout 600 python3 tools/fused_variants.py 2>&1 | tail -1
done
