"""Tuning aid: how many hidden units would still flip if the entries with a large PPR value (the pair's own endpoints:
two per pair) were evaluated against a SECOND reference pattern, taken at a typical such point?"""
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
for seed in (0, 1, 2):
    torch.manual_seed(seed)
    model = lpformer_amd.LinkTransformer(D.train_args_for(cfg), data, device=dev).to(dev).eval()
    wts = model._fold()
    batch = torch.from_numpy(D.sample_pairs(ei, n, cfg["batch"], seed=1000)).to(dev)
    s = model.compute_node_mask(batch)[1]
    pa, pb = s[1], s[2]
    hi, lo = torch.maximum(pa, pb), torch.minimum(pa, pb)
    t = 1
    tab, st = wts["flip_tab"][t].double(), wts["pe_stat"][t].double()
    def zraw(a, b):        # normalised to the pattern of (0, 0): z < 0 = flipped against it
        a, b = a.double(), b.double()
        var = st[0] * a * a + st[1] * b * b + st[2] + 2.0 * (st[3] * a * b + st[4] * a + st[5] * b)
        r = torch.rsqrt(var.clamp_min(0.0) + 1e-5)
        return r[:, None] * (a[:, None] * tab[:, 0] + b[:, None] * tab[:, 1] + tab[:, 2]) + tab[:, 3]
    for tau in (0.05, 0.1):
        heavy = hi > tau
        xs, ys = hi[heavy], lo[heavy]
        base_cnt = ((zraw(xs, ys) < 0).sum(1) + (zraw(ys, xs) < 0).sum(1)).float()
        # second pattern: the sign of every unit at the median heavy point, per argument order
        xm, ym = xs.median()[None], ys.median()[None]
        p1, p2 = zraw(xm, ym) < 0, zraw(ym, xm) < 0          # units flipped (against the origin's pattern) at that point
        c1 = ((zraw(xs, ys) < 0) != p1).sum(1) + ((zraw(ys, xs) < 0) != p2).sum(1)
        light = ((zraw(hi[~heavy], lo[~heavy]) < 0).sum(1) + (zraw(lo[~heavy], hi[~heavy]) < 0).sum(1)).float()
        print(f"seed {seed} tau {tau}: heavy {int(heavy.sum())} of {heavy.numel()} entries (median x {float(xm):.3f}, y {float(ym):.4f}); "
              f"flips per heavy entry vs origin {base_cnt.mean():.2f}, vs second pattern {c1.float().mean():.2f} "
              f"(p90 {c1.float().quantile(0.9):.0f}); light entries {light.mean():.3f}")
