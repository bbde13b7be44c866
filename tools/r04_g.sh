#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_random_sweep.py -m gpu -q -x > gpurun_out/r04g_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04g_tests.log
tail -25 gpurun_out/r04g_tests.log
timeout 900 python3 bench.py --no-cpu-baseline --weights random > gpurun_out/r04g_bench.log 2>&1
echo "bench rc=$?"
tail -1 gpurun_out/r04g_bench.log | python3 tools/all_configs_fmt.py
bash tools/rows_stamps.sh "" collab 2>&1 | tee gpurun_out/r04g_rows_stamps.txt
export LPF_ABLATE=1
VARIANTS=$'-DS3_ABL_NOBUCKET\n-DS3_ABL_NOWALK\n-DS3_ABL_NOFLT\n-DS3_ABL_NOWRITE\n-DS3_ABL_NOLB\n-DS3_ABL_NOBUCKET -DS3_ABL_NOWALK -DS3_ABL_NOFLT\n-DS3_ABL_NOBUCKET -DS3_ABL_NOWALK -DS3_ABL_NOFLT -DS3_ABL_NOWRITE -DS3_ABL_NOLB' CONFIGS="collab" bash tools/select_variants.sh 2>&1 | tee gpurun_out/r04g_select_ablations.txt
