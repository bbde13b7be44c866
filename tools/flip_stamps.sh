#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
touch lpformer_amd/csrc/pair_flip.hip
make -C lpformer_amd/csrc EXTRA="-DFL_STAMPS $2" > /dev/null 2>&1
LPF_CFG=${1:-collab} timeout 300 python3 tools/flip_stamps.py 2>&1 | tail -2
