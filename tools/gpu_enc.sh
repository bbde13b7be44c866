#!/bin/bash
# GPU pass for the encoder work: the fused-layer tests, the full-size encoder-row checks, then encoder timings
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_gpu_configs.py -q -x -k "fused_gcn or collab or citation2 or ppa" 2>&1 | tail -5
timeout 600 python3 - <<'PY'
import os, sys, json, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import lpformer_amd
from lpformer_amd import data as D
from lpformer_amd.profile import KernelTimer
for name in ("collab",):
    cfg = D.CONFIGS[name]; n = cfg["n"]; dev = torch.device("cuda:0")
    ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
    x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
    data = D.build_data(ei, x, n, edge_weight=w, eps=cfg["eps"], ppr_device=dev)
    torch.manual_seed(0)
    model = lpformer_amd.LinkTransformer(D.train_args_for(cfg), data, device=dev).to(dev).eval()
    for fused in (False, True):
        model.encoder_fused = fused
        for _ in range(5): model.propagate()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): model.propagate()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 50
        KernelTimer.reset(); KernelTimer.enabled = True
        for _ in range(5): model.propagate()
        res = {k: round(v[2] * 1e3, 1) for k, v in KernelTimer.summary().items()}
        KernelTimer.enabled = False
        print(name, "fused" if fused else "plain", "encoder_ms", round(ms, 4), json.dumps(res))
PY
