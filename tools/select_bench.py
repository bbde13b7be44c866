"""Tuning aid for the selection kernels (csrc/select3.hip): launch times of select_plan / select_run on the bench's own
batches (serial, HIP events) and a digest of the result (entry regions + segment pointers) to compare builds.
    LPF_CFG=collab|ppa|citation2|ddi python tools/select_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpformer_amd
from lpformer_amd import data as D
from lpformer_amd.profile import KernelTimer
name = os.environ.get("LPF_CFG", "collab")
cfg = D.CONFIGS[name]
n = cfg["n"]; dev = torch.device("cuda:0")
ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
data = D.build_data(ei, x, n, edge_weight=w, eps=cfg["eps"], ppr_device=dev)
torch.manual_seed(0)
model = lpformer_amd.LinkTransformer(D.train_args_for(cfg), data, device=dev).to(dev).eval()
model.select_grid = int(os.environ.get("LPF_SELECT_GRID", "0"))
nb = int(os.environ.get("LPF_BATCHES", "8"))
batches = [torch.from_numpy(D.sample_pairs(ei, n, cfg["batch"], seed=1000 + i)).to(dev) for i in range(nb)]
t0 = time.perf_counter()
ws = model._select_device(batches[0], False, None)
torch.cuda.synchronize()
print(f"# {name}: first selection (index build + sizing) {time.perf_counter() - t0:.2f} s; slots {int(ws.ctl[0])}, "
      f"items {int(ws.ctl[1])}, kept {[int(v) for v in ws.ctl[4:7]]}")
def digest(ws, bs):
    tp = ws.type_ptr.view(3, bs + 1).long()
    tot = tp[:, bs].tolist()
    ent = ws.entries.view(3, ws.ent_cap, 4).long()
    d = int((tp * (torch.arange(tp.numel(), device=dev).view(3, -1) % 8191 + 1)).sum().item())
    for t in range(3):
        e = ent[t, :tot[t]]
        d = (d * 1000003 + int((e * (torch.arange(e.numel(), device=dev).view(-1, 4) % 65521 + 1)).sum().item())) % (1 << 61)
    return d, tot
digs = []
for b in batches:
    ws = model._select_device(b, False, None)
    assert model.check_selection() or os.environ.get("LPF_ABLATE")
    digs.append(digest(ws, b.shape[1])[0])
for _ in range(3):
    for b in batches:
        model._select_device(b, False, None)
torch.cuda.synchronize()
KernelTimer.reset(); KernelTimer.enabled = True
reps = int(os.environ.get("LPF_REPS", "10"))
for _ in range(reps):
    for b in batches:
        model._select_device(b, False, None)
kt = KernelTimer.summary()
KernelTimer.enabled = False
print(f"{name}: select_plan {kt['select_plan'][2] * 1e3:.1f} us  select_run {kt['select_run'][2] * 1e3:.1f} us  "
      f"sum {(kt['select_plan'][2] + kt['select_run'][2]) * 1e3:.1f} us   digest {hash(tuple(digs)) & 0xffffffff:08x}", flush=True)
