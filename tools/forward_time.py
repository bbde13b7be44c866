"""Timing aid: the reference-style evaluation step -- h = model(edges); p = score_func(h) -- with the encoder output
cached the way evaluate.score_edges does NOT need (the reference re-runs the encoder per batch, testing.py:87), against
model.score_pairs on the same batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpformer_amd
from lpformer_amd import data as D
from lpformer_amd.profile import KernelTimer
cfg = D.CONFIGS[os.environ.get("LPF_CFG", "collab")]
n, bs = cfg["n"], cfg["batch"]; dev = torch.device("cuda:0")
ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
data = D.build_data(ei, x, n, edge_weight=w, eps=cfg["eps"], ppr_device=dev)
torch.manual_seed(0)
model = lpformer_amd.LinkTransformer(D.train_args_for(cfg), data, device=dev).to(dev).eval()
score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, 2).to(dev).eval()
b = torch.from_numpy(D.sample_pairs(ei, n, bs, seed=0)).to(dev)
h = model.propagate()
def t(fn, k=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
with torch.no_grad():
    print("model(edges) + score_func (encoder re-run per batch, as testing.py:87): %.3f ms" % t(lambda: score(model(b))))
    print("  ... with the predictions fetched per batch as the reference does (.cpu()): %.3f ms" % t(lambda: score(model(b)).cpu()))
    print("score_pairs(...).cpu() per batch: %.3f ms" % t(lambda: model.score_pairs(b, h, score).cpu()))
    print("elementwise_lin + calc_pairwise + score_func on a cached h (testing.py:113-117): %.3f ms" %
          t(lambda: score(torch.cat((model.elementwise_lin(h[b[0]] * h[b[1]]), model.calc_pairwise(b, h)[0]), dim=-1))))
    print("model.score_pairs(edges, h, score_func): %.3f ms" % t(lambda: model.score_pairs(b, h, score)))
    KernelTimer.reset(); KernelTimer.enabled = True
    for _ in range(5): score(torch.cat((model.elementwise_lin(h[b[0]] * h[b[1]]), model.calc_pairwise(b, h)[0]), dim=-1))
    print({k: round(v[2] * 1e3, 1) for k, v in KernelTimer.summary().items()})
