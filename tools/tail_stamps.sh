#!/bin/bash
# phase marks of tail_chain_kernel: tools/tail_stamps.sh "<extra -D flags>" [config]
cd "$GRAFT_REPO_ROOT" || exit 1
touch lpformer_amd/csrc/tail_chain.hip
make -C lpformer_amd/csrc EXTRA="-DTC_STAMPS $1" > /dev/null 2>&1 || echo "build failed"
LPF_CFG=${2:-collab} timeout 600 python3 tools/tail_stamps.py 2>&1 | grep -v amdgpu.ids
touch lpformer_amd/csrc/tail_chain.hip; make -C lpformer_amd/csrc > /dev/null 2>&1
