#!/bin/bash
# where a block's time goes in select4_kernel: tools/select4_stamps.sh "<extra -D flags>" [config] [threads]
cd "$GRAFT_REPO_ROOT" || exit 1
touch lpformer_amd/csrc/select4.hip
make -C lpformer_amd/csrc EXTRA="-DS4_STAMPS $1" > /dev/null 2>&1 || echo "build failed"
LPF_CFG=${2:-collab} LPF_SEL4_THREADS=${3:-0} timeout 600 python3 tools/select4_stamps.py 2>&1 | grep -v amdgpu.ids
touch lpformer_amd/csrc/select4.hip; make -C lpformer_amd/csrc > /dev/null 2>&1
