#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
VARIANTS=$'\n-DLB_BATCH=1\n-DS3_PER_CU=3\n-DS3_PER_CU=4 -DS3_MIN_WAVES=4\n-DS3_PER_CU=3 -DLB_BATCH=16' CONFIGS="collab ppa citation2" bash tools/select_variants.sh 2>&1 | tee gpurun_out/r04c_select_variants.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
bash tools/select_pmc.sh r04c collab
