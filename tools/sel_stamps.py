"""Tuning aid: per-item phase timing of select_run (LPF_SEL_DBG=2)."""
import os, sys, ctypes as C
os.environ["LPF_SEL_DBG"] = "2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpformer_amd
from lpformer_amd import data as D, _lib
cfg = D.CONFIGS["collab"]
n, bs = cfg["n"], cfg["batch"]
dev = torch.device("cuda:0")
ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
data = D.build_data(ei, x, n, edge_weight=w, eps=cfg["eps"], ppr_device=dev)
torch.manual_seed(0)
model = lpformer_amd.LinkTransformer(D.train_args_for(cfg), data, device=dev).to(dev).eval()
b = torch.from_numpy(D.sample_pairs(ei, n, bs, seed=0)).to(dev)
for _ in range(5):
    ws = model._select_device(b, False, None)
torch.cuda.synchronize()
items = int(ws.ctl[1].item())
buf = np.zeros(items * 8, np.int64)
lib = _lib.hip()
lib.lpf_select_debug_stamps.argtypes = [C.c_void_p, C.c_int64]
assert lib.lpf_select_debug_stamps(buf.ctypes.data, items) == 0
s = buf.reshape(items, 8)
t0 = s[:, 0].min()
st = (s[:, :6] - t0) / 100.0  # s_memtime ticks at 100 MHz?  report raw ratio too
print("items", items, "kernel span (ticks)", (s[:, 5].max() - t0))
d = np.diff(s[:, :6], axis=1)
names = ["window", "phaseA", "phaseB", "scan", "phaseD"]
for i, nm in enumerate(names):
    print(f"{nm:8s} mean {d[:, i].mean():10.0f}  p50 {np.median(d[:, i]):10.0f}  p99 {np.percentile(d[:, i], 99):10.0f}  max {d[:, i].max():10.0f}")
tot = s[:, 5] - s[:, 0]
print("item total mean", tot.mean(), "max", tot.max())
order = np.argsort(-d[:, 1] - d[:, 2])[:8]
print("slowest A+B items:", [(int(i), int(d[i, 1]), int(d[i, 2]), int(s[i, 6]), int(s[i, 7])) for i in order])
print("start times (first 8, every 256th):", (s[::256, 0] - t0)[:16])
print("end   times (first 8, every 256th):", (s[::256, 5] - t0)[:16])
# slot population: what the slots are and how many survive
err, kept = ws.read_status()
adj, adjx, val, t0, selfp = model._select_graphs(False, None)[:5] if hasattr(model, "_select_graphs") else (None,) * 5
print("slots", int(ws.ctl[0].item()), "kept [cn, 1hop, >1hop]", kept)
try:
    a, c = b[0], b[1]
    rp = adj.rowptr if hasattr(adj, "rowptr") else adj[0]
    deg = (rp[1:] - rp[:-1])
    tl = t0.len.long()
    print("sum dA+dB", int((deg[a] + deg[c]).sum()), "sum shorter T0", int(torch.minimum(tl[a], tl[c]).sum()),
          "mean longer T0", float(torch.maximum(tl[a], tl[c]).float().mean()), "max T0 row", int(tl.max()))
except Exception as e:
    print("population stats failed:", repr(e))
try:
    def runs(nodes):
        lo = rp[nodes]; cnt = deg[nodes]
        base = torch.repeat_interleave(lo - torch.cumsum(cnt, 0) + cnt, cnt)
        return base + torch.arange(int(cnt.sum()), device=dev)
    for nm, nodes in (("N(a)", a), ("N(b)", c)):
        own = selfp[runs(nodes)]
        p1 = ((own + 1.0) - 1.0) >= model.thresh_1hop
        pc = (0.5 * ((own * 2.0 + 2.0) - 2.0)) >= model.thresh_cn
        print(nm, "slots", own.numel(), "own >= th_1hop", int(p1.sum()), "own >= th_cn", int(pc.sum()), "either", int((p1 | pc).sum()))
    print("thresholds", model.thresh_cn, model.thresh_1hop, model.thresh_non1hop)
except Exception as e:
    print("pass-rate stats failed:", repr(e))
