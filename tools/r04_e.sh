#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
bash tools/select_stamps.sh "" collab 2>&1 | tee gpurun_out/r04e_stamps_collab.txt
VARIANTS=$'\n-DS3_PER_CU=3\n-DS3_PER_CU=4 -DS3_MIN_WAVES=4' CONFIGS="collab ppa citation2" bash tools/select_variants.sh 2>&1 | tee gpurun_out/r04e_select_variants.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_sweep.py -m gpu -q -x 2>&1 | tail -3
bash tools/rows_stamps.sh "" collab 2>&1 | tee gpurun_out/r04e_rows_stamps.txt
