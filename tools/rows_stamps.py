"""Tuning aid: phase marks of pair_rows_kernel (build with EXTRA=-DPR_STAMPS): per wavefront start / set-up done / chunk
sorted / empties / heavy / light end, rounds walked."""
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
model.attention_impl = "flip"
model.select_blocks = os.environ.get("LPF_SELECT_BLOCKS", "1") == "1"
batches = [torch.from_numpy(D.sample_pairs(ei, n, cfg["batch"], seed=1000 + i)).to(dev) for i in range(3)]
h = model.propagate()
lib = _lib.hip()
fn = lib.lpf_pair_rows_set_stamps
fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
buf = torch.zeros(1024 * 16 * 8, dtype=torch.int64, device=dev)
for b in batches * 3:
    model.score_pairs(b, h, score)
torch.cuda.synchronize()
assert fn(buf.data_ptr()) == 0
names = ["start", "tables+range", "chunk staged", "own set-up", "units done", "merge done"]
for i, b in enumerate(batches):
    buf.zero_()
    model.score_pairs(b, h, score)
    torch.cuda.synchronize()
    v = buf.view(-1, 8).cpu().numpy().astype(np.float64)
    v = v[v[:, 0] > 0]
    t0 = v[:, 0].min()
    rel = (v[:, :6] - t0) / 100.0
    print(f"batch {i}: {len(v)} wavefronts; kernel span {rel[:, 5].max():.1f} us")
    nw_ = 16
    own = (v[:, 3] - t0) / 100.0
    wv = np.arange(len(v)) % nw_
    print(f"   before the first barrier, own work done: searching wavefronts (0, 1) p50 {np.median(own[wv < 2]):.1f} max {own[wv < 2].max():.1f}; "
          f"table-filling wavefronts p50 {np.median(own[wv >= 2]):.1f} max {own[wv >= 2].max():.1f}")
    for k in range(6):
        print(f"   {names[k]:14s} p10 {np.percentile(rel[:, k], 10):6.1f}  p50 {np.percentile(rel[:, k], 50):6.1f}  "
              f"p90 {np.percentile(rel[:, k], 90):6.1f}  max {rel[:, k].max():6.1f}")
    if os.environ.get("LPF_STAMP_WG"):   # per workgroup: when its last wavefront finished, tickets drawn, pairs in its range
        nw = len(v) // (v[:, 7].astype(np.int64) & 0xffffffff > -1).sum() if False else 16
        wg = v.reshape(-1, nw, 8)
        end = (wg[:, :, 4].max(1) - t0) / 100.0
        beg = (wg[:, :, 3].min(1) - t0) / 100.0
        tickets = wg[:, :, 6].sum(1)
        pairs = (wg[:, 0, 7].astype(np.int64) & 0xffffffff)
        multi = (wg[:, 0, 7].astype(np.int64) >> 32)
        order = np.argsort(end)
        print("   slowest workgroups (index, units start, units end, tickets, max rounds of a wavefront, pairs, pairs in pieces):")
        for j in order[-8:]:
            print(f"     {j:4d} {beg[j]:6.1f} {end[j]:6.1f} {tickets[j]:4.0f} {wg[j, :, 6].max():3.0f} {pairs[j]:5d} {multi[j]:4d}")
        print("   median ones:")
        for j in order[len(order) // 2 - 3: len(order) // 2 + 3]:
            print(f"     {j:4d} {beg[j]:6.1f} {end[j]:6.1f} {tickets[j]:4.0f} {wg[j, :, 6].max():3.0f} {pairs[j]:5d} {multi[j]:4d}")
        j = order[-1]
        print("   slowest workgroup, per wavefront (units start, units end, rounds):",
              " ".join(f"{(wg[j, w, 3] - t0) / 100:.0f}-{(wg[j, w, 4] - t0) / 100:.0f}/{wg[j, w, 6]:.0f}" for w in range(nw)))
        j = order[len(order) // 2]
        print("   a median workgroup:", " ".join(f"{(wg[j, w, 3] - t0) / 100:.0f}-{(wg[j, w, 4] - t0) / 100:.0f}/{wg[j, w, 6]:.0f}" for w in range(nw)))
        print("   workgroup end times, top 20:", " ".join(f"{e:.0f}" for e in np.sort(end)[-20:]))
        print(f"   corr(end, tickets) {np.corrcoef(end, tickets)[0, 1]:.2f}  corr(end, pairs) {np.corrcoef(end, pairs)[0, 1]:.2f}  tickets p10 {np.percentile(tickets, 10):.0f} p50 {np.percentile(tickets, 50):.0f} p90 {np.percentile(tickets, 90):.0f} max {tickets.max():.0f}")
    r = v[:, 6]
    heavy = (v[:, 7].astype(np.int64) >> 32)
    print(f"   unit rounds per wavefront: mean {r.mean():.2f} p50 {np.percentile(r, 50):.0f} max {r.max():.0f}; per round: "
          f"{np.mean((rel[:, 4] - rel[:, 3])[r > 0] / r[r > 0]):.2f} us; pairs in several pieces per workgroup: mean "
          f"{heavy.mean():.1f} max {heavy.max()}; merge phase mean {np.mean(rel[:, 5] - rel[:, 4]):.1f} us max "
          f"{np.max(rel[:, 5] - rel[:, 4]):.1f}")
