"""Host (OpenMP) vs device PPR producer on the bench graph: time and bit-equality.  Usage: ppr_bench.py [config]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lpformer_amd  # noqa: E402
from lpformer_amd import data as D  # noqa: E402

cfg = D.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "collab"]
n = cfg["n"]
ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
both = np.concatenate([ei, ei[::-1]], axis=1)
t0 = time.perf_counter()
host = lpformer_amd.ppr.calc_ppr(both, n, 0.15, cfg["eps"])
t_host = time.perf_counter() - t0
print(f"host: {t_host:.2f} s on {os.cpu_count()} cores, nnz {host.rowptr[-1]}", flush=True)
for waves in (512, 1024, 2048, 4096, 8192):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tm = {}
    gpu = lpformer_amd.ppr.calc_ppr_gpu(both, n, 0.15, cfg["eps"], n_waves=waves, to_host=False, timings=tm)
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    same = (np.array_equal(gpu[0].cpu().numpy(), host.rowptr) and np.array_equal(gpu[1].cpu().numpy(), host.col)
            and np.array_equal(gpu[2].cpu().numpy().view(np.uint32), host.val.view(np.uint32)))
    print(f"gpu : {t_gpu:.3f} s with {waves} wavefronts (incl. CSR build + upload), bit-identical: {same}  "
          + " ".join(f"{k}={v:.3f}" for k, v in tm.items()), flush=True)
