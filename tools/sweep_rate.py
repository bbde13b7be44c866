"""Tuning aid: pairs per second of evaluate.score_edges over a long candidate list (64 batches), recorded steps against
eager launches (LPF_CFG)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpformer_amd
from lpformer_amd import data as D, evaluate as E
name = os.environ.get("LPF_CFG", "collab")
cfg = D.CONFIGS[name]
n = cfg["n"]; dev = torch.device("cuda:0"); bs = cfg["batch"]
ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
data = D.build_data(ei, x, n, edge_weight=w, eps=cfg["eps"], ppr_device=dev)
torch.manual_seed(0)
model = lpformer_amd.LinkTransformer(D.train_args_for(cfg), data, device=dev).to(dev).eval()
score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, 2).to(dev).eval()
edges = torch.from_numpy(np.concatenate([D.sample_pairs(ei, n, bs, seed=2000 + i) for i in range(64)], axis=1)).to(dev)
h = model.propagate()
for plans in (False, True, False, True):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = E.score_edges(model, score, edges, batch_size=bs, h=h, streams=8, plans=plans)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: plans={plans}: {edges.shape[1] / dt / 1e6:.1f} M pairs/s ({dt * 1e3 / 64:.4f} ms per batch of {bs}, second sweep)")
