"""Tuning probe: does a locality-aware node numbering speed up the encoder's aggregation on the bench graph?
Relabels the collab-like (or LPF_CFG) graph by several orders and times propagate() / the SpMM kernel for each."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee, breadth_first_order
import lpformer_amd
from lpformer_amd import data as D, graph
from lpformer_amd.profile import KernelTimer

cfg = D.CONFIGS[os.environ.get("LPF_CFG", "collab")]
n = cfg["n"]; dev = torch.device("cuda:0")
ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
deg = np.bincount(ei[0], minlength=n)
A = sp.csr_matrix((np.ones(ei.shape[1]), (ei[0], ei[1])), shape=(n, n))
orders = {"as generated (random ids)": np.arange(n)}
orders["degree descending"] = np.argsort(-deg, kind="stable")
orders["reverse Cuthill-McKee"] = np.asarray(reverse_cuthill_mckee(A, symmetric_mode=True))
bfs = breadth_first_order(A, int(np.argmax(deg)), directed=False, return_predecessors=False)
rest = np.setdiff1d(np.arange(n), bfs)
orders["BFS from the largest hub"] = np.concatenate([bfs, rest])
ppr = graph.CSR(np.arange(n + 1, dtype=np.int64), np.arange(n, dtype=np.int32), np.full(n, 0.15, np.float32), n)
for name, order in orders.items():
    new_id = np.empty(n, np.int64); new_id[order] = np.arange(n)
    e2 = new_id[ei]
    o = np.argsort(e2[0] * n + e2[1], kind="stable")
    data = D.build_data(e2[:, o], x[order], n, edge_weight=None if w is None else w[o], ppr=ppr)
    torch.manual_seed(0)
    model = lpformer_amd.LinkTransformer(D.train_args_for(cfg), data, device=dev).to(dev).eval()
    res = {}
    for prec in ("f32", "bf16"):
        model.encoder_precision = prec
        for _ in range(3): model.propagate()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): model.propagate()
        torch.cuda.synchronize()
        res[prec] = round((time.perf_counter() - t0) * 100, 4)
        KernelTimer.reset(); KernelTimer.enabled = True
        for _ in range(5): model.propagate()
        res[prec + "_spmm_us"] = round(KernelTimer.summary()["spmm_csr"][2] * 1e3, 1)
        KernelTimer.enabled = False
    print(f"{name:28s}", res, flush=True)
    del model, data
