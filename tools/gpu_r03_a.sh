#!/bin/bash
# round 3, first GPU pass: parity tests, then a bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03a_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03a_tests.log
tail -15 gpurun_out/r03a_tests.log
timeout 600 python bench.py --steps 40 > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err
echo "bench rc=$?"
tail -c 3000 gpurun_out/r03a_bench.json
tail -5 gpurun_out/r03a_bench.err
