"""Tuning aid: encoder kernel times on the collab-like graph (fused layers)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpformer_amd
from lpformer_amd import data as D
from lpformer_amd.profile import KernelTimer
cfg = D.CONFIGS[os.environ.get("LPF_CFG", "collab")]; n = cfg["n"]; dev = torch.device("cuda:0")
ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
data = D.build_data(ei, x, n, edge_weight=w, eps=1e-2)
from lpformer_amd import graph as G
_orig = G.fused_row_order
def _patched(rowptr, lo, hi, *a, **k):
    if os.environ.get("LPF_HUB_T"):
        k["long_threshold"] = int(os.environ["LPF_HUB_T"])
    order, hubs, parts = _orig(rowptr, lo, hi, *a, **k)
    mode = os.environ.get("LPF_ORDER", "degree")
    nh = 0 if hubs is None else hubs.shape[0]
    body = order[nh:]
    live = body[body >= 0]
    if mode == "natural":
        live = torch.sort(live).values
    elif mode == "shuffle":
        live = live[torch.randperm(live.numel(), device=live.device)]
    elif mode == "blocks":      # degree-sorted inside blocks of 4096 consecutive rows
        live = torch.sort(live).values
        deg = (rowptr[1:] - rowptr[:-1])[live.long()]
        key = (live.long() // 4096) * 100000 + (1000 - deg.clamp(max=999))
        live = live[torch.sort(key, stable=True).indices]
    body = torch.cat([live, body[body < 0]])
    return torch.cat([order[:nh], body]).contiguous(), hubs, parts
G.fused_row_order = _patched
torch.manual_seed(0)
model = lpformer_amd.LinkTransformer(D.train_args_for(cfg), data, device=dev).to(dev).eval()
model.encoder_fused = os.environ.get("LPF_FUSED", "1") == "1"
model.encoder_precision = os.environ.get("LPF_ENC_PREC", "f32")
for _ in range(5): model.propagate()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): model.propagate()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 50
KernelTimer.reset(); KernelTimer.enabled = True
for _ in range(5): model.propagate()
res = {k: round(v[2] * 1e3, 1) for k, v in KernelTimer.summary().items()}
print("encoder_ms", round(ms, 4), json.dumps(res))
