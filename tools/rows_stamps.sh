#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
touch lpformer_amd/csrc/pair_rows.hip
make -C lpformer_amd/csrc EXTRA="-DPR_STAMPS $1" > /dev/null 2>&1 || echo "build failed"
LPF_CFG=${2:-collab} timeout 600 python3 tools/rows_stamps.py 2>&1 | grep -v amdgpu.ids
touch lpformer_amd/csrc/pair_rows.hip; make -C lpformer_amd/csrc > /dev/null 2>&1
