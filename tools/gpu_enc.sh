#!/bin/bash
# GPU pass for the encoder work: the fused-layer tests, the full-size encoder-row checks, then encoder timings
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_dist.py -q -x -k "fused_gcn or collab or citation2 or ppa or sharded or dist" 2>&1 | tail -3
for c in ${@:-collab ppa}; do echo "[$c] $(LPF_CFG=$c timeout 600 python3 tools/enc_time.py 2>&1 | tail -1)"; done
