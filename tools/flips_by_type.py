"""Tuning aid: where do the flipped hidden units of the PE MLPs sit?  Entries and flips per TYPE (cn / 1-hop / >1-hop) on
the bench's batches -- only one type's correction table fits into the attention kernel's LDS at D = 128."""
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
for seed in (0, 1):
    torch.manual_seed(seed)
    model = lpformer_amd.LinkTransformer(D.train_args_for(cfg), data, device=dev).to(dev).eval()
    wts = model._fold()
    tot_e, tot_f = np.zeros(3), np.zeros(3)
    hist = np.zeros(8)
    for i in range(4):
        batch = torch.from_numpy(D.sample_pairs(ei, n, cfg["batch"], seed=1000 + i)).to(dev)
        sel = model.compute_node_mask(batch)
        for t, s in enumerate(sel):
            if s is None:
                continue
            pa, pb = s[1], s[2]
            tab, st = wts["flip_tab"][t], wts["pe_stat"][t]
            f = torch.zeros(pa.numel(), device=dev)
            for a, b in ((pa, pb), (pb, pa)):
                var = st[0] * a * a + st[1] * b * b + st[2] + 2.0 * (st[3] * a * b + st[4] * a + st[5] * b)
                r = torch.rsqrt(var.clamp_min(0.0) + 1e-5)
                z = r[:, None] * (a[:, None] * tab[:, 0] + b[:, None] * tab[:, 1] + tab[:, 2]) + tab[:, 3]
                f += (z < 0).sum(dim=1)
            tot_e[t] += pa.numel(); tot_f[t] += float(f.sum())
            hist += np.bincount(f.clamp(max=7).long().cpu().numpy(), minlength=8)
    print(f"seed {seed}: entries per type {tot_e / 4}, flips per type {tot_f / 4}, flips per entry {tot_f / np.maximum(tot_e, 1)}")
    print(f"         entries by flip count 0..7+: {(hist / hist.sum()).round(3)}")
