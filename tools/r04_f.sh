#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_sweep.py -m gpu -q -x 2>&1 | tail -3
bash tools/select_stamps.sh "" collab 2>&1 | tee gpurun_out/r04f_stamps_collab.txt
VARIANTS=$'\n-DS3_NO_BLOOM\n-DS3_PER_CU=3' CONFIGS="collab ppa citation2" bash tools/select_variants.sh 2>&1 | tee gpurun_out/r04f_select_variants.txt
bash tools/select_pmc.sh r04f collab
