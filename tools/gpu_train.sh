#!/bin/bash
# Training-step check on the GPU box: gradients vs the reference fixtures, then the step time and its profile.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py -q -x -s 2>&1 | grep -E "passed|failed|gradients|Error" | tail -6
LPF_TRAIN_PROFILE=1 timeout 600 python3 tools/train_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/train_profile.txt
grep "train step" gpurun_out/train_profile.txt; sed -n '/Name/,+24p' gpurun_out/train_profile.txt | cut -c1-60,118-200 | head -28
