#!/bin/bash
# Tuning aid: select3 build variants (EXTRA flags) x configs -> launch times + result digest
cd "$GRAFT_REPO_ROOT" || exit 1
CONFIGS=${CONFIGS:-"collab ppa"}
while IFS= read -r ex; do
  touch lpformer_amd/csrc/select3.hip
  make -C lpformer_amd/csrc EXTRA="$ex" > /dev/null 2>&1 || { echo "[$ex] build failed"; continue; }
  for c in $CONFIGS; do
    echo "[$ex] $(LPF_CFG=$c timeout 600 python3 tools/select_bench.py 2>&1 | grep -v amdgpu.ids | tail -1)"
  done
done <<< "${VARIANTS:-$'\n-DS3_PER_CU=3\n-DS3_PER_CU=4 -DS3_MIN_WAVES=4'}"
touch lpformer_amd/csrc/select3.hip; make -C lpformer_amd/csrc > /dev/null 2>&1
