#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
make -C lpformer_amd/csrc > /dev/null 2>&1
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04k_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04k_tests.log
grep -E "passed|failed|FAILED|rc=|Error" gpurun_out/r04k_tests.log | tail -12
bash tools/all_configs.sh r04k
bash tools/collect_profiles.sh r04 $(cat tools/.r04_commit 2>/dev/null || echo unknown)
LPF_TRAIN_BS=8192 timeout 600 python3 tools/train_time.py 2>&1 | grep "train step" | tee gpurun_out/r04k_train.txt
LPF_TRAIN_BS=8192 LPF_FUSED_ADAM=1 timeout 600 python3 tools/train_time.py 2>&1 | grep "train step" | tee -a gpurun_out/r04k_train.txt
