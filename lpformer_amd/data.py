"""Data-dict builder and synthetic graph generators.

The reference's readers (src/util/read_datasets.py:20-254) need the OGB / HeaRT files, which are not available
offline; this module builds the SAME dictionary schema (keys ``x, adj_t, full_adj_t, adj_mask, full_adj_mask, ppr,
ppr_test, num_nodes`` + split tensors) from an edge list, with graph entries already in CSR form so nothing is
converted on the scoring path, and generates synthetic graphs shaped like the named datasets.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import graph
from .ppr import calc_ppr, calc_ppr_gpu


def to_undirected(edge_index, num_nodes: int) -> np.ndarray:
    """PyG ``to_undirected`` for an unweighted edge list (what the reference applies to the validation edges,
    src/util/read_datasets.py:98-99): both directions of every edge, then COALESCED -- duplicate and reciprocal input
    pairs end up once per direction -- sorted by (row, col)."""
    ei = np.asarray(edge_index, dtype=np.int64).reshape(2, -1)
    row, col = np.concatenate([ei[0], ei[1]]), np.concatenate([ei[1], ei[0]])
    key = np.unique(row * np.int64(num_nodes) + col)
    return np.stack([key // num_nodes, key % num_nodes]).astype(np.int64)


def build_data(edge_index, x, num_nodes: int, *, edge_weight=None, eps: float = 5e-5, alpha: float = 0.15,
               ppr: Optional[graph.CSR] = None, val_edge_index=None, ppr_test: Optional[graph.CSR] = None,
               ppr_threads: int = 0, ppr_device=None) -> dict:
    """Reference-schema data dict (read_data_ogb, src/util/read_datasets.py:20-148) on CSR containers.

    edge_index: [2, E] directed list holding both directions of every undirected edge; edge_weight optional.
    val_edge_index: optional extra (validation) edges used when ``test_set=True`` (``--use-val-in-test``).
    ppr_device: None = host OpenMP producer; a device ("cuda:0") = the GPU producer (bit-identical result)."""
    ei = np.asarray(edge_index, dtype=np.int64)
    n = int(num_nodes)
    data = {"num_nodes": n, "x": torch.as_tensor(x, dtype=torch.float32)}
    w = None if edge_weight is None else np.asarray(edge_weight, np.float32)
    data["adj_t"] = graph.csr_from_coo(ei[0], ei[1], np.ones(ei.shape[1], np.float32) if w is None else w, n)
    data["adj_mask"] = graph.mask_csr(ei, n, symmetric=True)
    def producer(edges):
        if ppr_device is not None:
            return calc_ppr_gpu(edges, n, alpha, eps, device=ppr_device)
        return calc_ppr(edges, n, alpha, eps, ppr_threads)

    data["ppr"] = ppr if ppr is not None else producer(ei)
    if val_edge_index is not None:
        vei = to_undirected(val_edge_index, n)           # mirrored AND coalesced (read_datasets.py:98-99)
        full = np.concatenate([ei, vei], axis=1)
        fw = np.concatenate([np.ones(ei.shape[1], np.float32) if w is None else w, np.ones(vei.shape[1], np.float32)])
        data["full_adj_t"] = graph.csr_from_coo(full[0], full[1], fw, n)
        data["full_adj_mask"] = graph.mask_csr(full, n, symmetric=False)
        data["ppr_test"] = ppr_test if ppr_test is not None else producer(full)
    else:
        data["full_adj_t"], data["full_adj_mask"], data["ppr_test"] = data["adj_t"], data["adj_mask"], data["ppr"]
    return data


def chung_lu_graph(n: int, n_edges: int, gamma: float = 2.5, seed: int = 0, max_weight: int = 0):
    """Power-law (Chung-Lu) simple undirected graph: expected degree of node i ~ (i + i0)^(-1/(gamma-1)).
    Returns (edge_index [2, 2E] both directions sorted, edge_weight or None)."""
    rng = np.random.default_rng(seed)
    expo = 1.0 / (gamma - 1.0)
    wts = (np.arange(n, dtype=np.float64) + 10.0) ** (-expo)
    p = wts / wts.sum()
    cdf = np.cumsum(p)
    m = int(n_edges * 1.15) + 16
    a = np.searchsorted(cdf, rng.random(m)).clip(0, n - 1)
    b = np.searchsorted(cdf, rng.random(m)).clip(0, n - 1)
    perm = rng.permutation(n)  # decouple node id from degree rank
    a, b = perm[a], perm[b]
    keep = a != b
    lo, hi = np.minimum(a, b)[keep], np.maximum(a, b)[keep]
    key = np.unique(lo.astype(np.int64) * n + hi)
    if key.size > n_edges:
        key = np.sort(rng.choice(key, size=n_edges, replace=False))
    lo, hi = key // n, key % n
    src, dst = np.concatenate([lo, hi]), np.concatenate([hi, lo])
    weight = None
    if max_weight > 0:  # Zipf-like integer co-author counts (ogbl-collab carries summed multi-edge weights)
        wv = np.minimum(rng.zipf(2.0, size=lo.size), max_weight).astype(np.float32)
        weight = np.concatenate([wv, wv])
    order = np.argsort(src * n + dst, kind="stable")
    ei = np.stack([src[order], dst[order]]).astype(np.int64)
    return ei, (None if weight is None else weight[order])


def sample_pairs(edge_index, n: int, bs: int, seed: int = 0, frac_edges: float = 0.5) -> np.ndarray:
    """Candidate batch [2, bs]: a fraction of existing edges (positives) and uniform random pairs (negatives)."""
    rng = np.random.default_rng(seed)
    k = int(bs * frac_edges)
    pos = np.asarray(edge_index)[:, rng.integers(0, edge_index.shape[1], size=k)]
    neg = rng.integers(0, n, size=(2, bs - k))
    b = np.concatenate([pos, neg], axis=1)
    return np.ascontiguousarray(b[:, rng.permutation(bs)].astype(np.int64))


# name -> (N, F_in, undirected edges, D, L, residual, thresholds, eps, batch, gamma, max edge weight)
CONFIGS = {
    # SURVEY.md section 8 per-config table; dataset statistics are public OGB numbers, hyper-parameters from
    # scripts/replicate_existing.sh / replicate_heart.sh of the reference.
    "collab": dict(n=235_868, f_in=128, edges=1_180_000, dim=128, gnn_layers=3, residual=False,
                   thresholds=(0.0, 1e-4, 1e-2), eps=5e-5, batch=32_768, gamma=2.5, max_weight=10),
    "ppa": dict(n=576_289, f_in=58, edges=21_200_000, dim=64, gnn_layers=3, residual=True,
                thresholds=(0.0, 1e-4, 1e-2), eps=5e-5, batch=32_768, gamma=2.8, max_weight=0),
    "citation2": dict(n=2_927_963, f_in=128, edges=30_400_000, dim=64, gnn_layers=3, residual=True,
                      thresholds=(0.0, 1e-3, 1e-2), eps=2.5e-3, batch=32_768, gamma=2.6, max_weight=0),
    "ddi": dict(n=4_267, f_in=256, edges=1_070_000, dim=256, gnn_layers=3, residual=False,
                thresholds=(0.0, 1e-2, 1.0), eps=5e-6, batch=8_192, gamma=8.0, max_weight=0),
    "cora": dict(n=2_708, f_in=1_433, edges=4_488, dim=256, gnn_layers=1, residual=False,
                 thresholds=(0.0, 1e-2, 1e-2), eps=1e-4, batch=16_384, gamma=3.0, max_weight=0,
                 layer_norm=False, relu=False),
    "tiny": dict(n=3_000, f_in=32, edges=12_000, dim=64, gnn_layers=2, residual=False,
                 thresholds=(0.0, 1e-3, 1e-2), eps=1e-3, batch=2_048, gamma=2.5, max_weight=4),
}


def train_args_for(cfg: dict) -> dict:
    th = cfg["thresholds"]
    return {"thresh_cn": th[0], "thresh_1hop": th[1], "thresh_non1hop": th[2], "dim": cfg["dim"], "trans_layers": 1,
            "num_heads": 1, "att_drop": 0.0, "dropout": 0.0, "gnn_drop": 0.0, "feat_drop": 0.0, "gcn_cache": True,
            "gnn_layers": cfg["gnn_layers"], "residual": cfg["residual"], "layer_norm": cfg.get("layer_norm", True),
            "relu": cfg.get("relu", True)}
