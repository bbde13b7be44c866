#!/bin/bash
# Round-4 first GPU pass: the new tests, the bench line, the attention break-even table.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_random_sweep.py tests/test_gpu_configs.py -m gpu -q -x -k "rccl or auto_attention or busy_lane or sweep_with or graphed or overflow" > gpurun_out/r04a_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04a_tests.log
tail -15 gpurun_out/r04a_tests.log
timeout 900 python3 bench.py > gpurun_out/r04a_bench.log 2>&1
echo "bench rc=$?"
tail -c 3000 gpurun_out/r04a_bench.log
LPF_CFG=collab timeout 600 python3 tools/flip_breakeven.py > gpurun_out/r04a_breakeven_collab.txt 2>&1
cat gpurun_out/r04a_breakeven_collab.txt | tail -14
LPF_CFG=ddi timeout 600 python3 tools/flip_breakeven.py > gpurun_out/r04a_breakeven_ddi.txt 2>&1
cat gpurun_out/r04a_breakeven_ddi.txt | tail -14
