"""Tuning aid: where do the selected entries of a collab-like batch lie relative to the ReLU boundaries of the PE hidden
layers?  For random-init weights and after the bench's 150 training steps, per type:
  * share of entries inside the no-flip square, share without any flipped unit, flips per entry (pattern of (0, 0));
  * distinct activation patterns of the ORDERED points (pa, pb) / (pb, pa) and what the most frequent ones cover;
  * flips per entry against the most frequent pattern of each order instead of the pattern of (0, 0).
Decides whether a per-pattern table or a moved reference pattern can pay."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpformer_amd
from lpformer_amd import data as D
from lpformer_amd import fold as F

cfg = D.CONFIGS[os.environ.get("LPF_CFG", "collab")]
n, bs = cfg["n"], cfg["batch"]
dev = torch.device("cuda:0")
ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
data = D.build_data(ei, x, n, edge_weight=w, eps=cfg["eps"], ppr_device=dev)
torch.manual_seed(0)
model = lpformer_amd.LinkTransformer(D.train_args_for(cfg), data, device=dev).to(dev).eval()
score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, 2).to(dev).eval()
batch = torch.from_numpy(D.sample_pairs(ei, n, bs, seed=0)).to(dev)
sel = model.compute_node_mask(batch)
ent = [(s[1].cpu().numpy().astype(np.float64), s[2].cpu().numpy().astype(np.float64)) for s in sel]
print("entries per type:", [e[0].size for e in ent])
for t, (pa, pb) in enumerate(ent):
    q = [0.1, 0.5, 0.9, 0.99]
    print(f"  type {t}: pa quantiles {np.quantile(pa, q).round(5).tolist()}  pb {np.quantile(pb, q).round(5).tolist()}"
          f"  zeros pa {float((pa == 0).mean()):.3f} pb {float((pb == 0).mean()):.3f}")


def census(tag):
    st = {k: v for k, v in model.state_dict().items()}
    d = model.dim
    tabs, base, s0, wt = F.flip_tables(st, d, 3)
    _, stat = F.pe_tables(st, d, 3)
    out = {}
    for t, (pa, pb) in enumerate(ent):
        tab, s = tabs[t].astype(np.float64), stat[t].astype(np.float64)
        c = F.no_flip_radius(tabs[t], stat[t])
        pats = []
        flips0 = np.zeros(pa.size)
        for xx, yy in ((pa, pb), (pb, pa)):
            var = s[0] * xx * xx + s[1] * yy * yy + s[2] + 2.0 * (s[3] * xx * yy + s[4] * xx + s[5] * yy)
            r = 1.0 / np.sqrt(np.maximum(var, 0.0) + 1e-5)
            z = r[:, None] * (xx[:, None] * tab[:, 0] + yy[:, None] * tab[:, 1] + tab[:, 2]) + tab[:, 3]
            fl = z < 0                       # unit differs from the pattern of (0, 0)
            flips0 += fl.sum(1)
            pats.append(np.packbits(fl, axis=1))
        res = {"entries": int(pa.size), "box_c": round(float(c), 5),
               "in_box": round(float((np.maximum(pa, pb) <= c).mean()), 4),
               "no_flip": round(float((flips0 == 0).mean()), 4), "flips_per_entry": round(float(flips0.mean()), 3)}
        # patterns of the ordered points: distinct, coverage of the top ones, flips against the most frequent one
        for o, p in enumerate(pats):
            u, inv, cnt = np.unique(p, axis=0, return_inverse=True, return_counts=True)
            order = np.argsort(-cnt)
            cov = np.cumsum(cnt[order]) / pa.size
            res[f"order{o}_distinct"] = int(u.shape[0])
            res[f"order{o}_top_cover"] = {k: round(float(cov[min(k, cov.size) - 1]), 4) for k in (1, 2, 4, 8, 16, 32, 64)}
            mode = np.unpackbits(u[order[0]])[: tab.shape[0]].astype(bool)
            mine = np.unpackbits(p, axis=1)[:, : tab.shape[0]].astype(bool)
            res[f"order{o}_flips_vs_mode"] = round(float((mine != mode[None, :]).sum(1).mean()), 3)
            res[f"order{o}_mode_is_origin"] = bool(not mode.any())
        # joint pattern (both orders) of an entry
        both = np.concatenate(pats, axis=1)
        u, cnt = np.unique(both, axis=0, return_counts=True)
        cov = np.cumsum(np.sort(cnt)[::-1]) / pa.size
        res["joint_distinct"] = int(u.shape[0])
        res["joint_top_cover"] = {k: round(float(cov[min(k, cov.size) - 1]), 4) for k in (1, 2, 4, 8, 16, 32, 64, 128)}
        out[f"type{t}"] = res
    print(tag, json.dumps(out, indent=1))


def dump(tag):
    keep = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()
            if k.startswith("ppr_encoder") or "lin_r" in k}
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez_compressed(f"gpurun_out/census_{tag}.npz", **keep,
                        **{f"pa{t}": e[0].astype(np.float32) for t, e in enumerate(ent)},
                        **{f"pb{t}": e[1].astype(np.float32) for t, e in enumerate(ent)})


census("random-init")
if os.environ.get("LPF_DUMP"):
    dump("random")
pos_e = torch.from_numpy(ei[:, ei[0] < ei[1]]).to(dev)
opt = torch.optim.Adam(list(model.parameters()) + list(score.parameters()), lr=float(os.environ.get("LPF_LR", "1e-3")))
gen = torch.Generator(device=dev)
gen.manual_seed(4321)
model.train(); score.train()
tb = 4096
for it in range(int(os.environ.get("LPF_TRAIN_STEPS", "150"))):
    idx = torch.randint(0, pos_e.shape[1], (tb,), device=dev, generator=gen)
    neg = torch.randint(0, n, (2, tb), device=dev, generator=gen)
    loss = (-torch.log(score(model(pos_e[:, idx])) + 1e-6).mean() - torch.log(1 - score(model(neg)) + 1e-6).mean())
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    opt.zero_grad()
model.eval(); score.eval()
print("flips_per_entry (model):", model.flips_per_entry())
census("trained")
if os.environ.get("LPF_DUMP"):
    dump("trained")
