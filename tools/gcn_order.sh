#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_gpu_configs.py -q -x -k "fused_gcn" 2>&1 | tail -2
for o in degree natural shuffle blocks; do echo "[$o] $(LPF_ORDER=$o timeout 300 python3 tools/enc_time.py 2>&1 | tail -1)"; done
echo "[plain] $(LPF_FUSED=0 timeout 300 python3 tools/enc_time.py 2>&1 | tail -1)"
