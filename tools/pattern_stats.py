"""Tuning aid: per config, how many of the sample's ordered points the pattern grid tabulates (lpformer_amd/patterns.py)
and the flips per entry that are left for the exact path."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpformer_amd
from lpformer_amd import data as D
for name in sys.argv[1:] or ["collab", "cora", "ddi"]:
    cfg = D.CONFIGS[name]
    n = cfg["n"]; dev = torch.device("cuda:0")
    ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
    x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
    data = D.build_data(ei, x, n, edge_weight=w, eps=cfg["eps"], ppr_device=dev)
    torch.manual_seed(0)
    model = lpformer_amd.LinkTransformer(D.train_args_for(cfg), data, device=dev).to(dev).eval()
    model.attention_impl = "flip"
    wf = model._fold()
    pt = model._pattern_tables(wf)
    smp = model._entry_sample()
    print(name, "D", model.dim, "sample entries", [0 if s is None else int(s[0].numel()) for s in smp])
    for t, st in enumerate(pt["stats"]):
        print("  type", t, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()},
              "box c", round(float(wf["pe_stat"][t, 7]), 5))
        if smp[t] is not None and smp[t][0].numel():
            from lpformer_amd import patterns as P
            ia, ib = P.cell_index(smp[t][0], pt["geo"]), P.cell_index(smp[t][1], pt["geo"])
            g = pt["grid"][t]
            for npk in (8, 16):
                ok = (g[ia, ib] < npk) & (g[ib, ia] < npk)
                print(f"    entries with both orders among the first {npk} patterns: {float(ok.double().mean()):.4f}")
            q = [0.1, 0.5, 0.9, 0.99]
            print("    pa quantiles", np.quantile(smp[t][0].cpu().numpy(), q).round(4).tolist())
    fr, fl = model._flip_stats()
    print("  flips per entry raw", round(fr, 3), "left for the exact path (against the named patterns)", round(fl, 3), "patterns pay:", model._patterns_pay())
