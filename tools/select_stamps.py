"""Tuning aid: where an item's time goes in select3_run_kernel (build with EXTRA=-DS3_STAMPS; tools/select_stamps.sh).
Thread 0 of every workgroup accumulates wall-clock ticks (100 MHz) between the marks of the item loop."""
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
batches = [torch.from_numpy(D.sample_pairs(ei, n, cfg["batch"], seed=1000 + i)).to(dev) for i in range(4)]
lib = _lib.hip()
buf = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
fn = lib.lpf_select3_set_stamps
fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
for b in batches * 3:
    model._select_device(b, False, None)
torch.cuda.synchronize()
assert fn(buf.data_ptr()) == 0
names = ["-", "top barrier", "barrier behind window/desc requests", "window processed", "desc rest + barrier", "walked entries",
         "filter words", "buckets", "arithmetic+ballots+barrier", "rank scan+publish", "look-back+write-out (parked)",
         "parking", "final look-back+write-out"]
acc = np.zeros((len(batches), 16))
for i, b in enumerate(batches):
    buf.zero_()
    model._select_device(b, False, None)
    torch.cuda.synchronize()
    v = buf.view(-1, 16).cpu().numpy().astype(np.float64)
    live = v[:, 14] > 0
    v = v[live]
    span = (v[:, 14].max() - v[:, 13].min()) / 100.0
    per = v[:, :13].sum(0) / 100.0 / live.sum()
    print(f"batch {i}: {live.sum()} workgroups, kernel span {span:.1f} us, mean busy per workgroup {per.sum():.1f} us; "
          f"workgroup end times (us after the first start) p50 {np.percentile(v[:, 14] - v[:, 13].min(), 50) / 100:.1f} "
          f"max {(v[:, 14].max() - v[:, 13].min()) / 100:.1f}; start spread {(v[:, 13].max() - v[:, 13].min()) / 100:.1f}")
    acc[i, :13] = per
m = acc.mean(0)
for k in range(1, 13):
    print(f"  {names[k]:40s} {m[k]:7.2f} us per workgroup")
