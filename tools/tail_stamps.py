"""Tuning aid: phase marks of tail_chain_kernel (build with EXTRA=-DTC_STAMPS): per wavefront start / stage A read /
stage B k-groups done / LayerNorm B done / stage C r_e k-groups done / r_p k-groups done / end."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpformer_amd
from lpformer_amd import data as D, _lib
name = os.environ.get("LPF_CFG", "collab")
cfg = D.CONFIGS[name]
n = cfg["n"]; dev = torch.device("cuda:0")
ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
data = D.build_data(ei, x, n, edge_weight=w, eps=cfg["eps"], ppr_device=dev)
torch.manual_seed(0)
model = lpformer_amd.LinkTransformer(D.train_args_for(cfg), data, device=dev).to(dev).eval()
score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, 2).to(dev).eval()
model.use_side_stream = False
batches = [torch.from_numpy(D.sample_pairs(ei, n, cfg["batch"], seed=1000 + i)).to(dev) for i in range(3)]
h = model.propagate()
lib = _lib.hip()
fn = lib.lpf_tail_chain_set_stamps
fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
buf = torch.zeros(2048 * 8 * 8, dtype=torch.int64, device=dev)
for b in batches * 3:
    model.score_pairs(b, h, score)
torch.cuda.synchronize()
assert fn(buf.data_ptr()) == 0
names = ["start", "rows read", "stage B loop", "LN B", "C: r_e loop", "C: r_p loop", "end"]
for i, b in enumerate(batches[:2]):
    buf.zero_()
    model.score_pairs(b, h, score)
    torch.cuda.synchronize()
    v = buf.view(-1, 8).cpu().numpy().astype(np.float64)
    v = v[v[:, 0] > 0]
    t0 = v[:, 0].min()
    for kind, tag in ((0, "full workgroups"), (1, "workgroups of pairs without selected nodes")):
        s = v[v[:, 7] == kind]
        if not len(s):
            continue
        rel = (s[:, :7] - t0) / 100.0
        print(f"batch {i}, {tag}: {len(s)} wavefronts; last end {rel[:, 6].max():.1f} us")
        for k in range(7):
            if kind == 1 and 0 < k < 6:
                continue
            print(f"   {names[k]:14s} p10 {np.percentile(rel[:, k], 10):6.1f}  p50 {np.percentile(rel[:, k], 50):6.1f}  "
                  f"p90 {np.percentile(rel[:, k], 90):6.1f}  max {rel[:, k].max():6.1f}")
        if kind == 0:
            d = np.diff(rel, axis=1)
            print("   phase lengths, median:", " ".join(f"{names[k + 1]} {np.median(d[:, k]):.1f}" for k in range(6)))
