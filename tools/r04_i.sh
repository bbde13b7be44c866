#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r04i_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04i_tests.log
grep -E "passed|failed|FAILED|rc=|Error" gpurun_out/r04i_tests.log | tail -12
bash tools/rows_stamps.sh "" collab 2>&1 | tail -8
bash tools/all_configs.sh r04i
LPF_CFG=collab timeout 600 python3 tools/enc_time.py 2>&1 | tail -12
