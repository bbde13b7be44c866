#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
make -C lpformer_amd/csrc > /dev/null 2>&1
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04m_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04m_tests.log
grep -E "passed|failed|FAILED|rc=|Error" gpurun_out/r04m_tests.log | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04m_bench20.log 2>&1
tail -1 gpurun_out/r04m_bench20.log | python3 tools/all_configs_fmt.py
LPF_TRAIN_BS=8192 LPF_FUSED_ADAM=1 timeout 600 python3 tools/train_time.py 2>&1 | grep "train step"
