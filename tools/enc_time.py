import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import lpformer_amd
from lpformer_amd import data as D
from lpformer_amd.profile import KernelTimer
cfg = D.CONFIGS[os.environ.get("LPF_CFG", "collab")]
n = cfg["n"]; dev = torch.device("cuda:0")
ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
data = D.build_data(ei, x, n, edge_weight=w, eps=cfg["eps"], ppr_device=dev)
model = lpformer_amd.LinkTransformer(D.train_args_for(cfg), data, device=dev).to(dev).eval()
for rep in range(3):
    for prec in ("f32", "bf16"):
        model.encoder_precision = prec
        model.propagate(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            model.propagate()
        torch.cuda.synchronize()
        print(prec, round((time.perf_counter() - t0) * 100, 4), "ms")
for prec in ("f32", "bf16"):
    model.encoder_precision = prec
    KernelTimer.reset(); KernelTimer.enabled = True
    for _ in range(5): model.propagate()
    print(prec, {k: round(v[2] * 1e3, 1) for k, v in KernelTimer.summary().items()})
    KernelTimer.enabled = False
