"""Tuning aid: the HOST cost of one eager pair-stage step (every C entry point of the step replaced by a no-op) beside
the real pipelined step -- is the eight-stream loop bound by the device or by the launches?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpformer_amd
from lpformer_amd import data as D, _lib
cfg = D.CONFIGS[os.environ.get("LPF_CFG", "collab")]
n = cfg["n"]; dev = torch.device("cuda:0")
ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
data = D.build_data(ei, x, n, edge_weight=w, eps=cfg["eps"], ppr_device=dev)
torch.manual_seed(0)
model = lpformer_amd.LinkTransformer(D.train_args_for(cfg), data, device=dev).to(dev).eval()
score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, 2).to(dev).eval()
batches = [torch.from_numpy(D.sample_pairs(ei, n, cfg["batch"], seed=1000 + i)).to(dev) for i in range(16)]
h = model.propagate()
lib = _lib.hip()
fns = ["lpf_pair_gather_f32", "lpf_dense_chain_f32", "lpf_dense_chain_side_f32", "lpf_select3_plan", "lpf_select3_run",
       "lpf_select4", "lpf_pair_attention_rows_perm_f32", "lpf_pair_attention_rows4_f32", "lpf_tail_chain_rows_perm_f32"]
with torch.no_grad():
    for nl in (1, 4, 8, 12):
        lanes = model.lanes(nl)
        def sweep(steps):
            for i in range(steps):
                with torch.cuda.stream(lanes[i % nl]):
                    model.score_pairs(batches[i % len(batches)], h, score, logits=True)
        for rep in range(3):
            sweep(4 * nl); torch.cuda.synchronize()
            for s in lanes:
                model.check_selection(s)
        def timed(steps=240):
            sweep(2 * nl); torch.cuda.synchronize()
            ts = []
            for rep in range(5):
                t0 = time.perf_counter(); sweep(steps); t1 = time.perf_counter(); torch.cuda.synchronize()
                ts.append(((time.perf_counter() - t0) / steps * 1e3, (t1 - t0) / steps * 1e3))
            ts.sort()
            return ts[len(ts) // 2]
        real = timed()
        keep = {f: getattr(lib, f) for f in fns}
        for f in fns:
            setattr(lib, f, lambda *a, **k: 0)
        floor = timed()
        for f, v in keep.items():
            setattr(lib, f, v)
        print(f"{nl:2d} streams: step {real[0]:.4f} ms (host issue time {real[1]:.4f}); with no-op launches: {floor[0]:.4f} ms "
              f"(host {floor[1]:.4f})")
