"""Tuning aid (library built with -DFL_STAMPS): when do the wavefronts of pair_flip_kernel finish?"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpformer_amd
from lpformer_amd import data as D

cfg = D.CONFIGS[os.environ.get("LPF_CFG", "collab")]
n, bs = cfg["n"], cfg["batch"]
dev = torch.device("cuda:0")
ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
data = D.build_data(ei, x, n, edge_weight=w, eps=cfg["eps"], ppr_device=dev)
torch.manual_seed(0)
model = lpformer_amd.LinkTransformer(D.train_args_for(cfg), data, device=dev).to(dev).eval()
score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, 2).to(dev).eval()
batches = [torch.from_numpy(D.sample_pairs(ei, n, bs, seed=i)).to(dev) for i in range(5)]
h = model.propagate()
for i in range(20):
    model.score_pairs(batches[i % 5], h, score)
torch.cuda.synchronize()
d = model.dim
wpg = {128: 16, 256: 4, 64: 8, 32: 8}[d]
per_cu = {128: 1, 256: 3, 64: 2, 32: 2}[d]
nw = 256 * per_cu * wpg
ws, part, bnd, units_cap = model._fused_attention(batches[4], h, False, None, None)
torch.cuda.synchronize()
end = 3 * units_cap * 2 * (d + 4)      # the stamps sit right below this many floats of the buffer
raw = bnd[:end].view(torch.int64)[-(nw + 1):].cpu().numpy().astype(np.int64)
t0 = raw[-1]
fin = np.sort(raw[:-1] - t0) / 100.0   # 100 MHz constant clock -> us
wg = (raw[:-1] - t0).reshape(-1, wpg).max(axis=1) / 100.0
print(json.dumps({"waves": int(nw), "us_first_wave_done": float(fin[0]), "us_median": float(np.median(fin)),
                  "us_p90": float(fin[int(0.9 * nw)]), "us_last": float(fin[-1]),
                  "wg_done_us_median": float(np.median(wg)), "wg_done_us_min": float(wg.min()), "wg_done_us_max": float(wg.max()),
                  "wg_done_sorted_every16": [float(v) for v in np.sort(wg)[::16]]}))
