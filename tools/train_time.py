"""Timing aid: one training step (forward + backward + Adam) of the collab-like config through lpformer_amd/train.py,
resident graphs (no per-batch masked adjacency), batch = LPF_TRAIN_BS positives + as many negatives."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpformer_amd
from lpformer_amd import data as D
cfg = D.CONFIGS[os.environ.get("LPF_CFG", "collab")]
n = cfg["n"]; dev = torch.device("cuda:0"); bs = int(os.environ.get("LPF_TRAIN_BS", "8192"))
ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
data = D.build_data(ei, x, n, edge_weight=w, eps=cfg["eps"], ppr_device=dev)
targs = dict(D.train_args_for(cfg), att_drop=0.1, dropout=0.1, gnn_drop=0.1, feat_drop=0.1)
torch.manual_seed(0)
model = lpformer_amd.LinkTransformer(targs, data, device=dev).to(dev)
score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, 2, 0.1).to(dev)
# The reference's optimiser as it builds it (train_model.py:98: torch.optim.Adam(params, lr=..), every other argument the
# default -- on device tensors torch then takes its multi-tensor "foreach" path; rounds 4-5 passed fused=False here,
# which ALSO switches foreach off: 7 small kernels per parameter and step, 1.7 ms of a 13.9 ms step).
# LPF_ADAM=fused | single: the one-launch and the per-tensor forms, for comparison.
_adam = {"fused": dict(fused=True), "single": dict(foreach=False)}.get(os.environ.get("LPF_ADAM", ""), {})
opt = torch.optim.Adam(list(model.parameters()) + list(score.parameters()), lr=1e-3, **_adam)
pos = torch.from_numpy(ei[:, ei[0] < ei[1]]).to(dev)
class _ST:  # torch_sparse.SparseTensor look-alike on the device (what the reference's loop builds per batch)
    def __init__(self, r, c, v, n): self._r, self._c, self._v, self._n = r, c, v, n
    def coo(self): return self._r, self._c, self._v
    def sparse_sizes(self): return (self._n, self._n)
# "": no override; "gpu" / "cpu": adj_prop AND adj_mask rebuilt per batch (--mask-input, the pubmed script); "mask": adj_mask
# only (every OGB script: train_model.py:38-46 without --mask-input); "removed": lpformer_amd.RemovedEdges(edges) as
# adj_mask, "removed_both": as adj_mask and adj_prop;
# "mask_raw" / "gpu_raw": the same overrides as unsorted look-alike objects (what this tool passed until round 6)
masked_mode = os.environ.get("LPF_TRAIN_MASKED", "")
if os.environ.get("LPF_MASK_DELTA"):
    model.use_mask_delta = os.environ["LPF_MASK_DELTA"] != "0"
wts = None if w is None else torch.from_numpy(w[ei[0] < ei[1]]).to(dev)
def step(i):
    model.train(); score.train()
    idx = torch.randint(0, pos.shape[1], (bs,), device=dev)
    edges = pos[:, idx]
    if masked_mode == "removed":
        pos_loss = -torch.log(score(model(edges, adj_mask=lpformer_amd.RemovedEdges(edges))) + 1e-6).mean()
    elif masked_mode == "removed_both":       # --mask-input with both differences named
        rm = lpformer_amd.RemovedEdges(edges)
        pos_loss = -torch.log(score(model(edges, adj_prop=rm, adj_mask=rm)) + 1e-6).mean()
    elif masked_mode:
        # train_model.py:40-51: the batch's positive edges removed from the propagation / typing adjacency
        keep = torch.ones(pos.shape[1], dtype=torch.bool, device=dev); keep[idx] = False
        k = pos[:, keep]
        r, c = torch.cat([k[0], k[1]]), torch.cat([k[1], k[0]])
        v = None if wts is None else torch.cat([wts[keep], wts[keep]])
        if masked_mode == "cpu":
            r, c, v = r.cpu(), c.cpu(), None if v is None else v.cpu()
        if masked_mode.endswith("_raw"):      # rounds 2-5 of this tool: look-alikes with their entries in no order
            adjt = _ST(r, c, v, n) if masked_mode != "mask_raw" else None
            adjm = _ST(r, c, None, n)
        else:
            # the objects the reference builds: adj_mask a COALESCED torch sparse COO tensor (:44, sorted by (row, col)),
            # adj_prop a SparseTensor, whose storage is sorted the same way (:51-52) -- the sort is the loop's own cost
            adjm = torch.sparse_coo_tensor(torch.stack([r, c]), torch.ones(r.numel(), dtype=torch.int32, device=r.device),
                                           (n, n)).coalesce()
            adjt = None
            if masked_mode != "mask":
                order = torch.argsort(r * n + c)
                adjt = _ST(r[order], c[order], None if v is None else v[order], n)
        pos_loss = -torch.log(score(model(edges, adj_prop=adjt, adj_mask=adjm)) + 1e-6).mean()
    else:
        pos_loss = -torch.log(score(model(edges)) + 1e-6).mean()
    neg = torch.randint(0, n, (2, bs), device=dev)
    neg_loss = -torch.log(1 - score(model(neg)) + 1e-6).mean()
    loss = pos_loss + neg_loss
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step(); opt.zero_grad()
    return loss
for i in range(3): step(i)
torch.cuda.synchronize()
# five windows of ten steps, the MEDIAN reported (the step is host-paced in parts, and the pool's boxes share their host:
# single windows of the same build differ by +-1 ms)
k, wins = 10, []
for _ in range(5):
    t0 = time.perf_counter()
    for i in range(k): l = step(i)
    torch.cuda.synchronize()
    wins.append((time.perf_counter() - t0) / k)
dt = sorted(wins)[len(wins) // 2]
print(f"train step ({bs} positives + {bs} negatives, two encoder passes): {dt * 1e3:.1f} ms  -> {2 * bs / dt / 1e3:.0f} k pairs/s, "
      f"loss {float(l.detach()):.4f} (windows {' '.join(f'{w * 1e3:.1f}' for w in wins)})")
if os.environ.get("LPF_TRAIN_PROFILE"):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for i in range(3): step(i)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=70, max_name_column_width=90))
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=15, max_name_column_width=60))
