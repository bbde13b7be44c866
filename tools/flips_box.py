"""Tuning aid: how large is the square [0, c]^2 of PPR value pairs inside which NO hidden unit leaves the pattern of (0, 0)
(grid search in float64 on the model's tables), and which share of a batch's entries lies inside it?  (An entry inside
needs no flip detection at all.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpformer_amd
from lpformer_amd import data as D
cfg = D.CONFIGS[os.environ.get("LPF_CFG", "collab")]
n = cfg["n"]; dev = torch.device("cuda:0")
ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
data = D.build_data(ei, x, n, edge_weight=w, eps=cfg["eps"], ppr_device=dev)
def box_radius(tab, st, grid=97, c_hi=1.0):
    """largest c (bisection) with min_k z_k(x, y) > 0 on a grid over [0, c]^2"""
    tab, st = tab.astype(np.float64), st.astype(np.float64)
    def ok(c):
        g = np.linspace(0.0, c, grid)
        xx, yy = np.meshgrid(g, g, indexing="ij")
        xx, yy = xx.ravel(), yy.ravel()
        var = st[0] * xx * xx + st[1] * yy * yy + st[2] + 2 * (st[3] * xx * yy + st[4] * xx + st[5] * yy)
        r = 1.0 / np.sqrt(np.maximum(var, 0) + 1e-5)
        z = r[:, None] * (xx[:, None] * tab[:, 0] + yy[:, None] * tab[:, 1] + tab[:, 2]) + tab[:, 3]
        return z.min() > 0
    lo, hi = 0.0, c_hi
    for _ in range(30):
        mid = 0.5 * (lo + hi)
        lo, hi = (mid, hi) if ok(mid) else (lo, mid)
    return lo
for seed in (0, 1, 2, 3):
    torch.manual_seed(seed)
    model = lpformer_amd.LinkTransformer(D.train_args_for(cfg), data, device=dev).to(dev).eval()
    wts = model._fold()
    batch = torch.from_numpy(D.sample_pairs(ei, n, cfg["batch"], seed=1000)).to(dev)
    sel = model.compute_node_mask(batch)
    out = []
    for t, s in enumerate(sel):
        if s is None or s[1].numel() == 0:
            continue
        c = box_radius(wts["flip_tab"][t].cpu().numpy(), wts["pe_stat"][t].cpu().numpy())
        hi = torch.maximum(s[1], s[2])
        out.append(f"type {t}: c = {c:.4f}, {float((hi <= c).float().mean()) * 100:.1f} % of {hi.numel()} entries inside "
                   f"(p50 / p90 / p99 of max(pa, pb): {float(hi.median()):.4f} / {float(hi.quantile(0.9)):.4f} / {float(hi.quantile(0.99)):.4f})")
    print(f"seed {seed}: " + "; ".join(out))
