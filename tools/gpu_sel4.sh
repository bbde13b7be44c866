#!/bin/bash
# select4 + rows4: the new tests, the parity suite through the new default path, then serial kernel times A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_select4.py -q -x 2>&1 | tail -15
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_sweep.py -q -x 2>&1 | tail -3
for cfg in ${@:-collab}; do
for blocks in 1 0; do
echo "$cfg blocks=$blocks: $(LPF_CFG=$cfg LPF_SELECT_BLOCKS=$blocks timeout 300 python3 tools/fused_variants.py 2>&1 | tail -1 | cut -c1-400)"
done
echo "$cfg blocks=1 threads=512: $(LPF_CFG=$cfg LPF_SEL4_THREADS=512 timeout 300 python3 tools/fused_variants.py 2>&1 | tail -1 | cut -c1-400)"
done
