"""Tuning aid: times the pair-stage kernels of the collab-like bench workload (serial, HIP events)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpformer_amd
from lpformer_amd import data as D
from lpformer_amd.profile import KernelTimer

cfg = D.CONFIGS[os.environ.get("LPF_CFG", "collab")]
n, bs = cfg["n"], cfg["batch"]
dev = torch.device("cuda:0")
ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
data = D.build_data(ei, x, n, edge_weight=w, eps=cfg["eps"], ppr_device=dev)
torch.manual_seed(0)
model = lpformer_amd.LinkTransformer(D.train_args_for(cfg), data, device=dev).to(dev).eval()
score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, 2).to(dev).eval()
model.precision = os.environ.get("LPF_PRECISION", "f32")
model.tail_precision = os.environ.get("LPF_TAIL_PRECISION", "f32")
if os.environ.get("LPF_PT_EXACT_MAX"):
    model.PT_EXACT_MAX = float(os.environ["LPF_PT_EXACT_MAX"])
model.select_blocks = os.environ.get("LPF_SELECT_BLOCKS", "1") == "1"
model.select4_threads = int(os.environ.get("LPF_SEL4_THREADS", "0"))
model.attention_impl = os.environ.get("LPF_ATT", "auto")
batches = [torch.from_numpy(D.sample_pairs(ei, n, bs, seed=i)).to(dev) for i in range(5)]
h = model.propagate()
for i in range(20):
    model.score_pairs(batches[i % 5], h, score)
torch.cuda.synchronize()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.7:
    for i in range(10):
        model.score_pairs(batches[i % 5], h, score)
    torch.cuda.synchronize()
KernelTimer.reset(); KernelTimer.enabled = True
for i in range(40):
    model.score_pairs(batches[i % 5], h, score)
res = {k: round(v[2] * 1e3, 1) for k, v in KernelTimer.summary().items()}
KernelTimer.enabled = False
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(100):
    model.score_pairs(batches[i % 5], h, score)
torch.cuda.synchronize()
res["serial_us_per_step"] = round((time.perf_counter() - t0) * 1e4, 1)
print(os.environ.get("LPF_FUSED_DBG", "0"), json.dumps(res))
