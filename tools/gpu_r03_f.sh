#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py -q -x -s > gpurun_out/r03f_train_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03f_train_tests.log
grep -E "passed|failed|FAILED|rc=|Error|error|gradients" gpurun_out/r03f_train_tests.log | tail -20
LPF_TRAIN_PROFILE=1 timeout 600 python3 tools/train_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03f_train_profile.txt
grep "train step" gpurun_out/r03f_train_profile.txt; sed -n '/Name/,+28p' gpurun_out/r03f_train_profile.txt | cut -c1-60,118-200 | head -34
