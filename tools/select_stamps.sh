#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
touch lpformer_amd/csrc/select3.hip
make -C lpformer_amd/csrc EXTRA="-DS3_STAMPS $1" > /dev/null 2>&1 || echo "build failed"
LPF_CFG=${2:-collab} timeout 600 python3 tools/select_stamps.py 2>&1 | grep -v amdgpu.ids
touch lpformer_amd/csrc/select3.hip; make -C lpformer_amd/csrc > /dev/null 2>&1
