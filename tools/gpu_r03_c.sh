#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -k "d256 or ddi or cora or seed3 or seed7 or seed10 or folded or bf16" > gpurun_out/r03c_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03c_tests.log
grep -E "passed|failed|FAILED|rc=|Error" gpurun_out/r03c_tests.log | tail -30
for c in ddi cora; do
  echo "== $c"
  timeout 900 python3 bench.py --config $c --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 | python3 tools/all_configs_fmt.py
done
