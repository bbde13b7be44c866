#!/bin/bash
# quick GPU pass: parity + random sweep tests, then serial kernel times of two configs
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_sweep.py -q -x 2>&1 | tail -2
for cfg in ${@:-collab ddi}; do
echo "$cfg serial: $(LPF_CFG=$cfg timeout 300 python3 tools/fused_variants.py 2>&1 | tail -1 | cut -c1-330)"
done
