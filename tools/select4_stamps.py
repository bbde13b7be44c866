"""Tuning aid: where a block's time goes in select4_kernel (build with EXTRA=-DS4_STAMPS; tools/select4_stamps.sh).
Thread 0 of every workgroup leaves the wall clock (100 MHz) at the marks of the kernel."""
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
model.select4_threads = int(os.environ.get("LPF_SEL4_THREADS", "0"))
batches = [torch.from_numpy(D.sample_pairs(ei, n, cfg["batch"], seed=1000 + i)).to(dev) for i in range(4)]
lib = _lib.hip()
buf = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
fn = lib.lpf_select4_set_stamps
fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
for b in batches * 3:
    model._select4_device(b, False)
torch.cuda.synchronize()
assert fn(buf.data_ptr()) == 0
names = ["plan issued", "plan barrier", "walked entries", "buckets", "arithmetic+ballots", "typing barrier", "scan+barrier",
         "writes issued (+ further batches)", "end barrier", "table + counters"]
acc = []
for i, b in enumerate(batches):
    buf.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    model._select4_device(b, False)
    e1.record()
    torch.cuda.synchronize()
    v = buf.view(-1, 16).cpu().numpy().astype(np.float64)
    v = v[v[:, 10] > 0]
    t0 = v[:, 0].min()
    dur = (v[:, 10] - v[:, 0]) / 100.0
    start = (v[:, 0] - t0) / 100.0
    print(f"batch {i}: {len(v)} workgroups, event time {e0.elapsed_time(e1) * 1e3:.1f} us, span {(v[:, 10].max() - t0) / 100:.1f} us; "
          f"workgroup life p50 {np.percentile(dur, 50):.1f} p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f}; "
          f"starts p50 {np.percentile(start, 50):.1f} p90 {np.percentile(start, 90):.1f} max {start.max():.1f}; "
          f"slots p50 {np.percentile(v[:, 11], 50):.0f} max {v[:, 11].max():.0f}")
    acc.append(np.diff(v[:, :11], axis=1).mean(0) / 100.0)
m = np.mean(acc, axis=0)
for k, nm in enumerate(names):
    print(f"  {nm:36s} {m[k]:7.2f} us")
