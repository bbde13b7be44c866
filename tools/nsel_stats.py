import numpy as np, torch, sys
sys.path.insert(0, '.')
import lpformer_amd
from lpformer_amd import data as D
cfg = D.CONFIGS["collab"]; n = cfg["n"]
ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=10)
x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
data = D.build_data(ei, x, n, edge_weight=w, eps=cfg["eps"])
dev = torch.device("cuda:0")
model = lpformer_amd.LinkTransformer(D.train_args_for(cfg), data, device=dev).to(dev).eval()
b = torch.from_numpy(D.sample_pairs(ei, n, cfg["batch"], seed=0)).to(dev)
sel = model.compute_node_mask(b)
cnt = np.zeros(cfg["batch"], np.int64)
for t in sel:
    np.add.at(cnt, t[0][0].cpu().numpy(), 1)
print("n_sel total", cnt.sum(), "mean", cnt.mean(), "max", cnt.max(), "p50", np.percentile(cnt,50), "p99", np.percentile(cnt,99), "p99.9", np.percentile(cnt,99.9))
print("top10", np.sort(cnt)[-10:])
print("types", [int(t[0].shape[1]) for t in sel])
deg = np.diff(data["adj_mask"].rowptr)
print("rows deg>512:", (deg>512).sum(), "deg p99", np.percentile(deg,99))
