#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r03g_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03g_tests.log
grep -E "passed|failed|FAILED|rc=|Error" gpurun_out/r03g_tests.log | tail -12
for c in collab ddi ppa; do
  echo "== $c"
  timeout 900 python3 bench.py --config $c --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 | python3 tools/all_configs_fmt.py
done
