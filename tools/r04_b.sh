#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_sweep.py -m gpu -q -x > gpurun_out/r04b_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04b_tests.log
tail -5 gpurun_out/r04b_tests.log
VARIANTS=$'\n-DS3_NO_BLOOM\n-DS3_PER_CU=3\n-DS3_PER_CU=4 -DS3_MIN_WAVES=4\n-DS3_PER_CU=1' CONFIGS="collab ppa citation2" bash tools/select_variants.sh 2>&1 | tee gpurun_out/r04b_select_variants.txt
LPF_CFG=cora LPF_GAINS=1,4,8,16,32,64,128,256 timeout 600 python3 tools/flip_breakeven.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04b_breakeven_cora.txt
