#!/bin/bash
# One GPU-box pass: the whole -m gpu suite, then step time + kernel shares of the synthetic configs.
#   gpurun --timeout 3600 -- 'bash tools/gpu_check.sh [tag] [configs...]'
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-check}; shift
CONFIGS=${@:-collab ddi ppa}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log
grep -E "passed|failed|FAILED|rc=|Error" gpurun_out/${TAG}_tests.log | tail -12
for c in $CONFIGS; do
  echo "== $c"
  timeout 900 python3 bench.py --config $c --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 | python3 tools/all_configs_fmt.py
done
