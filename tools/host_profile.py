"""Host-side cost of one pair-stage step (cProfile over the bench step loop; GPU box)."""
import cProfile
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import lpformer_amd  # noqa: E402
from lpformer_amd import data as D  # noqa: E402

import os
cfg = dict(D.CONFIGS[os.environ.get("LPF_CFG", "collab")])
if os.environ.get("LPF_BS"):
    cfg["batch"] = int(os.environ["LPF_BS"])
dev = torch.device("cuda:0")
ei, w = D.chung_lu_graph(cfg["n"], cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
x = np.random.default_rng(1).standard_normal((cfg["n"], cfg["f_in"])).astype(np.float32)
data = D.build_data(ei, x, cfg["n"], edge_weight=w, eps=cfg["eps"], ppr_device=dev)
model = lpformer_amd.LinkTransformer(D.train_args_for(cfg), data, device=dev).to(dev).eval()
score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, 2).to(dev).eval()
batches = [torch.from_numpy(D.sample_pairs(ei, cfg["n"], cfg["batch"], seed=i)).to(dev) for i in range(4)]
h = model.propagate()
lanes = model.lanes(int(sys.argv[1]) if len(sys.argv) > 1 else 6)


def step(i):
    with torch.cuda.stream(lanes[i % len(lanes)]):
        return model.score_pairs(batches[i % 4], h, score)


for i in range(8):
    step(i)
torch.cuda.synchronize()
steps = 200
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    step(i)
pr.disable()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"issue loop {t_issue / steps * 1e3:.3f} ms/step, total {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step")
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
