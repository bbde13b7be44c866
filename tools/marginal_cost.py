"""Tuning aid: what each launch of the pair stage costs INSIDE the eight-stream pipeline -- the step time with that
launch skipped (its C entry point replaced by a no-op: results are garbage, only the time is read).  A kernel's serial
duration says little here: a latency-bound kernel hides under the others, an issue-bound one does not."""
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
if os.environ.get("LPF_TAIL_BF16"):
    model.tail_precision = "bf16"
batches = [torch.from_numpy(D.sample_pairs(ei, n, cfg["batch"], seed=1000 + i)).to(dev) for i in range(16)]
h = model.propagate()
nlanes = int(os.environ.get("LPF_STREAMS", "8"))
host = []
lib = _lib.hip()

def measure(steps=240):
    """Eight recorded steps (lpformer_amd.PlannedScorer; LPF_MODE=graph: HIP graphs) replayed round-robin."""
    plan = os.environ.get("LPF_MODE", "plan") == "plan"
    if plan:
        scorers = [lpformer_amd.PlannedScorer(model, score, h, batches[k % len(batches)], logits=True, adopt_input=True)
                   for k in range(nlanes)]
    else:
        scorers = [lpformer_amd.GraphedScorer(model, score, h, batches[k % len(batches)], logits=True, adopt_input=True)
                   for k in range(nlanes)]
    def sweep(k):
        for i in range(k):
            sc = scorers[i % nlanes]
            if plan:
                sc(batches[i % len(batches)], validate=False, ordered=False)
            else:
                with torch.cuda.stream(sc.stream):
                    sc(sc.batch, validate=False)
    sweep(4 * nlanes); torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        t0 = time.perf_counter(); sweep(steps); t1 = time.perf_counter(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / steps * 1e3)
        host.append((t1 - t0) / steps * 1e3)
    return float(np.median(ts))

with torch.no_grad():
    base = measure()
    print(f"all launches: {base:.4f} ms per step; host time per replay {np.median(host):.4f} ms")
    names = {"elementwise branch + q gather": ["lpf_dense_chain_side_f32"],
             "select run": ["lpf_select3_run", "lpf_select4"],
             # (the attention alone cannot be left out: the tail walks the pairs in the order it leaves)
             "dense tail": ["lpf_tail_chain_rows_perm_f32", "lpf_tail_chain_rows_perm_bf16"],
             "attention + tail": ["lpf_pair_attention_rows_perm_f32", "lpf_pair_attention_rows4_f32",
                                  "lpf_tail_chain_rows_perm_f32", "lpf_tail_chain_rows_perm_bf16"]}
    if os.environ.get("LPF_ONLY_BASE"):
        names = {}
    names.pop("select run", None)   # (a selection that does not run leaves garbage counts to size the next workspace from)
    for label, fns in names.items():
        keep = {f: getattr(lib, f) for f in fns}
        for f in fns:
            setattr(lib, f, lambda *a, **k: 0)
        try:
            t = measure()
            print(f"without {label:32s} {t:.4f} ms per step  (marginal cost {1e3 * (base - t):6.1f} us)")
        except Exception as e:  # noqa: BLE001
            print(f"without {label:32s} failed: {type(e).__name__}: {str(e)[:100]}")
        for f, v in keep.items():
            setattr(lib, f, v)
    print(f"all launches again: {measure():.4f} ms per step")
