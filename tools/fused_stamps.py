"""Tuning aid: per-tile phase timing of pair_fused_kernel, first 8 tiles of 512 wavefronts.  Needs a library built with
the stamps compiled in:  make -C lpformer_amd/csrc clean && make -C lpformer_amd/csrc EXTRA=-DLPF_FUSED_STAMPS"""
import os, sys, ctypes as C
os.environ["LPF_FUSED_DBG"] = "32"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpformer_amd
from lpformer_amd import data as D, _lib
cfg = D.CONFIGS[os.environ.get("LPF_CFG", "collab")]
n, bs = cfg["n"], cfg["batch"]
dev = torch.device("cuda:0")
ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
data = D.build_data(ei, x, n, edge_weight=w, eps=cfg["eps"], ppr_device=dev)
torch.manual_seed(0)
model = lpformer_amd.LinkTransformer(D.train_args_for(cfg), data, device=dev).to(dev).eval()
score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, 2).to(dev).eval()
b = torch.from_numpy(D.sample_pairs(ei, n, bs, seed=0)).to(dev)
h = model.propagate()
for _ in range(20):
    model.score_pairs(b, h, score)
torch.cuda.synchronize()
buf = np.zeros(4096 * 8, np.int64)
lib = _lib.hip()
if not hasattr(lib, "lpf_fused_debug_stamps"):
    sys.exit("this build has no stamps: rebuild with EXTRA=-DLPF_FUSED_STAMPS")
lib.lpf_fused_debug_stamps.argtypes = [C.c_void_p, C.c_int64]
assert lib.lpf_fused_debug_stamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(512, 8, 8)[:, :, :6]          # wave, tile, stamp
ok = (s > 0).all(axis=2)
d = np.diff(s, axis=2)
names = ["setup (records, constants, row request)", "MFMA loop", "score phase", "butterfly", "softmax walk + record stores"]
print("tiles with stamps:", int(ok.sum()), " (s_memtime ticks = shader clocks)")
for i, nm in enumerate(names):
    v = d[:, :, i][ok]
    print(f"{nm:28s} mean {v.mean():9.0f}  p50 {np.median(v):9.0f}  p90 {np.percentile(v, 90):9.0f}")
tot = (s[:, :, 5] - s[:, :, 0])[ok]
print("tile total mean", tot.mean(), " between tiles (end -> next start):",
      np.mean([(s[w_, k + 1, 0] - s[w_, k, 5]) for w_ in range(512) for k in range(7) if ok[w_, k] and ok[w_, k + 1]]))
print("wave 0 tile starts:", (s[0, :, 0] - s[0, 0, 0]).tolist())
print("wave 4 tile starts:", (s[4, :, 0] - s[0, 0, 0]).tolist(), "(D <= 128: same SIMD as wave 0)")
