"""Tuning probe (round 5): what is a locality-aware node numbering worth to the encoder when the graph HAS community
structure?  Degree-corrected planted-partition graph at the collab-like size (power-law expected degrees, a share
`LPF_INTRA` of every node's edges inside its community, the rest Chung-Lu over the whole graph, node ids shuffled) next
to the bench's Chung-Lu graph; the graph is relabelled by several orders and propagate() is timed for each with the
current kernels (one-launch GCN layers).  LPF_GRAPH = sbm | chunglu."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee
import lpformer_amd
from lpformer_amd import data as D, graph
from lpformer_amd.profile import KernelTimer

cfg = D.CONFIGS["collab"]
n, m = cfg["n"], cfg["edges"]
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
kind = os.environ.get("LPF_GRAPH", "sbm")
comm = None
if kind == "chunglu":
    ei, _ = D.chung_lu_graph(n, m, gamma=cfg["gamma"], seed=0, max_weight=0)
else:
    intra = float(os.environ.get("LPF_INTRA", "0.8"))
    csize = int(os.environ.get("LPF_COMM", "120"))
    k = n // csize
    comm = rng.integers(0, k, size=n)                       # community of every node
    order_c = np.argsort(comm, kind="stable")
    start = np.searchsorted(comm[order_c], np.arange(k + 1))
    wts = (np.arange(n, dtype=np.float64) + 10.0) ** (-1.0 / (cfg["gamma"] - 1.0))
    wts = wts[rng.permutation(n)]
    p = wts / wts.sum()
    mm = int(m * 1.2)
    a = rng.choice(n, size=mm, p=p)
    # partner: inside a's community (weighted by the same expected degrees) or anywhere
    inside = rng.random(mm) < intra
    b = rng.choice(n, size=mm, p=p)
    ca = comm[a[inside]]
    lo, hi = start[ca], start[ca + 1]
    b[inside] = order_c[lo + (rng.random(inside.sum()) * (hi - lo)).astype(np.int64)]
    keep = a != b
    lo_, hi_ = np.minimum(a, b)[keep], np.maximum(a, b)[keep]
    key = np.unique(lo_.astype(np.int64) * n + hi_)
    if key.size > m:
        key = np.sort(rng.choice(key, size=m, replace=False))
    lo_, hi_ = key // n, key % n
    src, dst = np.concatenate([lo_, hi_]), np.concatenate([hi_, lo_])
    o = np.argsort(src * n + dst, kind="stable")
    ei = np.stack([src[o], dst[o]]).astype(np.int64)
x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
deg = np.bincount(ei[0], minlength=n)
A = sp.csr_matrix((np.ones(ei.shape[1]), (ei[0], ei[1])), shape=(n, n))
orders = {"as generated (random ids)": np.arange(n), "degree descending": np.argsort(-deg, kind="stable"),
          "reverse Cuthill-McKee": np.asarray(reverse_cuthill_mckee(A, symmetric_mode=True))}
if comm is not None:
    orders["by planted community"] = np.argsort(comm, kind="stable")
    orders["by community, hubs first inside"] = np.lexsort((-deg, comm))
print(f"# {kind}: N = {n}, {ei.shape[1] // 2} undirected edges, max degree {deg.max()}", flush=True)
ppr = graph.CSR(np.arange(n + 1, dtype=np.int64), np.arange(n, dtype=np.int32), np.full(n, 0.15, np.float32), n)
for name, order in orders.items():
    new_id = np.empty(n, np.int64); new_id[order] = np.arange(n)
    e2 = new_id[ei]
    o = np.argsort(e2[0] * n + e2[1], kind="stable")
    data = D.build_data(e2[:, o], x[order], n, edge_weight=None, ppr=ppr)
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
        res[prec + "_encoder_ms"] = round((time.perf_counter() - t0) * 100, 4)
        KernelTimer.reset(); KernelTimer.enabled = True
        for _ in range(5): model.propagate()
        kt = KernelTimer.summary()
        res[prec + "_layer_us"] = {k: round(v[2] * 1e3, 1) for k, v in kt.items() if k in ("gcn_layer_fused", "spmm_row_parts", "spmm_csr")}
        KernelTimer.enabled = False
    print(f"{name:32s}", res, flush=True)
    del model, data
