#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
bash tools/select_stamps.sh "" collab 2>&1 | tee gpurun_out/r04d_stamps_collab.txt
bash tools/select_stamps.sh "" ppa 2>&1 | tee gpurun_out/r04d_stamps_ppa.txt
make -C lpformer_amd/csrc > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_random_sweep.py -m gpu -q -x > gpurun_out/r04d_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04d_tests.log
tail -25 gpurun_out/r04d_tests.log
timeout 900 python3 bench.py --no-cpu-baseline --weights random > gpurun_out/r04d_bench.log 2>&1
echo "bench rc=$?"
tail -1 gpurun_out/r04d_bench.log | python3 tools/all_configs_fmt.py
