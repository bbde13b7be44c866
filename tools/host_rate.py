"""Host-side issue cost of one pair-stage step: the same step loop on a batch so small that the GPU is never the limit."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import lpformer_amd  # noqa: E402
from lpformer_amd import data as D  # noqa: E402

cfg = D.CONFIGS["collab"]
dev = torch.device("cuda:0")
ei, w = D.chung_lu_graph(cfg["n"], cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
x = np.random.default_rng(1).standard_normal((cfg["n"], cfg["f_in"])).astype(np.float32)
data = D.build_data(ei, x, cfg["n"], edge_weight=w, eps=cfg["eps"], ppr_device=dev)
model = lpformer_amd.LinkTransformer(D.train_args_for(cfg), data, device=dev).to(dev).eval()
score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, 2).to(dev).eval()
h = model.propagate()
for bs in (256, 32768):
    batches = [torch.from_numpy(D.sample_pairs(ei, cfg["n"], bs, seed=i)).to(dev) for i in range(4)]
    for nl in (1, 4):
        lanes = model.lanes(nl)

        def step(i):
            with torch.cuda.stream(lanes[i % nl]):
                return model.score_pairs(batches[i % 4], h, score)
        for i in range(10):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(300):
            step(i)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"bs={bs:6d} lanes={nl}: issue loop {t_issue / 300 * 1e3:.3f} ms/step, incl. drain {t_all / 300 * 1e3:.3f} ms/step",
              flush=True)
