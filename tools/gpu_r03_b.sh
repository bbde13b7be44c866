#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r03b_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03b_tests.log
grep -E "passed|failed|FAILED|rc=" gpurun_out/r03b_tests.log | tail -30
bash tools/all_configs.sh r03b > /dev/null 2>&1
cat gpurun_out/r03b_all_configs.txt
