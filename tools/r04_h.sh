#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
make -C lpformer_amd/csrc > /dev/null 2>&1
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04h_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04h_tests.log
grep -E "passed|failed|FAILED|rc=|Error" gpurun_out/r04h_tests.log | tail -12
bash tools/all_configs.sh r04h
