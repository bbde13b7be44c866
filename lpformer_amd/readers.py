"""Data-dict builders for dataset files on disk: what ``read_data_ogb`` / ``read_data_planetoid``
(src/util/read_datasets.py:20-148, :150-254) return, without ``ogb``, ``torch_sparse`` or PyG.

The dict has the schema the model reads on this package's CSR containers (``lpformer_amd.data``):
``x, num_nodes, adj_t, full_adj_t, adj_mask, full_adj_mask, ppr, ppr_test`` plus the split tensors
``train_pos, train_pos_val, valid_pos, valid_neg, test_pos, test_neg`` the loops use (src/train/train_model.py,
src/train/testing.py).

PARITY: pinned against the reference's own readers for everything the REFERENCE does -- tests/golden/make_reader_golden.py
runs ``read_data_planetoid`` and ``read_data_ogb`` unmodified on tiny datasets (tests/golden/planetoid_tiny/, ogb_tiny/) and
tests/test_readers.py compares every split tensor, the features, both adjacencies, both masks and both PPR matrices (bit
for bit), for collab (year filter, weight-summing symmetrisation, validation edges in the test graph), ppa (HeaRT index
files), ddi (feature-less table, quartered HeaRT validation set) and citation2 (source / target splits, directed PPR).
UNPINNED is the one step the reference delegates to the ``ogb`` package, which is not available offline: raw files ->
graph object.  Its layout is taken from the public OGB documentation --

    <root>/<name with '-' -> '_'>/raw/{edge, num-node-list, node-feat, edge_weight, edge_year}.csv.gz
    <root>/<name ...>/split/<time | throughput | target>/{train, valid, test}.pt      (torch-pickled dicts)

-- with the inverse edges appended for the undirected datasets, and restated once more in the generator's stand-in
dataset class.  The HeaRT text layout (``{train,valid,test}_pos.txt``, ``{valid,test}_neg.txt``, ``gnn_feature``) is the
one the reference itself parses (read_datasets.py:160-215).  What each step does follows the reference line by line and
is cited there.
"""
from __future__ import annotations

import gzip
import os
from typing import Optional

import numpy as np
import torch

from . import data as D
from .ppr import load_or_calc_ppr

# split directory per dataset (OGB master table) and whether the stored edge list needs its inverse edges added
_OGB = {
    "ogbl-collab": ("time", True),
    "ogbl-ppa": ("throughput", True),
    "ogbl-ddi": ("target", True),
    "ogbl-citation2": ("time", False),
}


def _csv(path: str, dtype) -> Optional[np.ndarray]:
    if not os.path.isfile(path):
        return None
    with gzip.open(path, "rt") as f:
        arr = np.loadtxt(f, delimiter=",", dtype=dtype, ndmin=2)
    return arr


def _as_np(v) -> np.ndarray:
    return v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)


def to_undirected_sum(edge: np.ndarray, weight: Optional[np.ndarray], n: int):
    """``torch_geometric.utils.to_undirected(edge_index, weight, reduce='add')``: both directions of every edge,
    duplicates merged with their weights summed, sorted by (row, col).  edge: [2, E]."""
    src = np.concatenate([edge[0], edge[1]]).astype(np.int64)
    dst = np.concatenate([edge[1], edge[0]]).astype(np.int64)
    key = src * n + dst
    w = None if weight is None else np.concatenate([weight, weight]).astype(np.float64)
    uniq, inv = np.unique(key, return_inverse=True)
    out_w = None
    if w is not None:
        out_w = np.zeros(uniq.size, np.float64)
        np.add.at(out_w, inv, w)
        out_w = out_w.astype(np.float32)
    return np.stack([uniq // n, uniq % n]), out_w


def _finish(data: dict, name: str, edge_index: np.ndarray, weight, n: int, x, eps: float, val_edges, *,
            cache_root, ppr_device, ppr_threads, ppr_edges=None) -> dict:
    """adjacency / mask / PPR part shared by both readers (read_datasets.py:79-129).  ``ppr_edges``: the edge list the
    PPR producer runs on when it is not the adjacency's (citation2: the reference hands ``get_ppr`` the DIRECTED
    ``data.edge_index``, :122, while the adjacency is symmetrised, :88-90)."""
    pe = edge_index if ppr_edges is None else ppr_edges
    ppr = load_or_calc_ppr(pe, n, 0.15, eps, cache_root=cache_root, dataset=name, is_val=False,
                           device=ppr_device, num_threads=ppr_threads)
    ppr_test = None
    if val_edges is not None:
        vei = D.to_undirected(val_edges, n)   # coalesced: a validation pair given twice / in both directions counts once
        full = np.concatenate([pe, vei], axis=1)
        ppr_test = load_or_calc_ppr(full, n, 0.15, eps, cache_root=cache_root, dataset=name, is_val=True,
                                    device=ppr_device, num_threads=ppr_threads)
    built = D.build_data(edge_index, x, n, edge_weight=weight, eps=eps, ppr=ppr, val_edge_index=val_edges,
                         ppr_test=ppr_test)
    data.update(built)
    data["edge_index"] = torch.from_numpy(edge_index)
    return data


def read_data_ogb(root: str, data_name: str, *, eps: float = 5e-5, dim: int = 128, use_val_in_test: bool = False,
                  heart_dir: Optional[str] = None, collab_first_year: int = 2007, cache_root: Optional[str] = None,
                  ppr_device=None, ppr_threads: int = 0, seed: Optional[int] = None) -> dict:
    """``read_data_ogb`` (read_datasets.py:20-148) from the raw OGB files.  ``heart_dir``: directory holding
    ``<data_name>/heart_{valid,test}_samples.npy`` (``--heart``); ``collab_first_year``: ``filter_by_year``'s default
    (:259)."""
    if data_name not in _OGB:
        raise ValueError(f"unknown OGB link dataset {data_name!r}")
    split_type, add_inverse = _OGB[data_name]
    base = os.path.join(root, data_name.replace("-", "_"))
    raw = os.path.join(base, "raw")
    edge = _csv(os.path.join(raw, "edge.csv.gz"), np.int64)
    if edge is None:
        raise FileNotFoundError(os.path.join(raw, "edge.csv.gz"))
    edge = edge.T                                                    # [2, E]
    n = int(_csv(os.path.join(raw, "num-node-list.csv.gz"), np.int64).reshape(-1)[0])
    feat = _csv(os.path.join(raw, "node-feat.csv.gz"), np.float32)
    weight = _csv(os.path.join(raw, "edge_weight.csv.gz"), np.float32)
    weight = None if weight is None else weight.reshape(-1)
    split = {s: {k: _as_np(v) for k, v in torch.load(os.path.join(base, "split", split_type, f"{s}.pt"),
                                                      weights_only=False).items()}
             for s in ("train", "valid", "test")}

    if add_inverse:  # the PyG dataset object stores both directions (weights repeated)
        edge_index = np.concatenate([edge, edge[::-1]], axis=1)
        weight = None if weight is None else np.concatenate([weight, weight])
    else:
        edge_index = edge
    if "collab" in data_name:  # filter_by_year (:259-280): training edges from `collab_first_year` on, merged
        keep = split["train"]["year"].reshape(-1) >= collab_first_year
        for k in ("edge", "weight", "year"):
            split["train"][k] = split["train"][k][keep]
        edge_index, weight = to_undirected_sum(split["train"]["edge"].T, split["train"]["weight"].reshape(-1), n)

    data = {"dataset": data_name}
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).long()  # noqa: E731
    if data_name != "ogbl-citation2":  # (:41-46)
        data["train_pos"], data["valid_pos"], data["test_pos"] = (t(split[s]["edge"]) for s in ("train", "valid", "test"))
        data["valid_neg"], data["test_neg"] = t(split["valid"]["edge_neg"]), t(split["test"]["edge_neg"])
    else:                               # (:47-57): (source, target) columns; negatives are K targets per source
        for s, key in (("train", "train_pos"), ("valid", "valid_pos"), ("test", "test_pos")):
            data[key] = t(np.stack([split[s]["source_node"], split[s]["target_node"]], axis=1))
        data["valid_neg"], data["test_neg"] = t(split["valid"]["target_node_neg"]), t(split["test"]["target_node_neg"])
    if heart_dir is not None and "ppa" in data_name:  # (:60-68)
        hd = os.path.join(heart_dir, data_name)
        data["valid_pos"] = data["valid_pos"][torch.load(os.path.join(hd, "valid_samples_index.pt"), weights_only=False)]
        data["test_pos"] = data["test_pos"][torch.load(os.path.join(hd, "test_samples_index.pt"), weights_only=False)]
    g = None if seed is None else torch.Generator().manual_seed(seed)
    idx = torch.randperm(data["train_pos"].size(0), generator=g)[: data["valid_pos"].size(0)]  # (:71-72)
    data["train_pos_val"] = data["train_pos"][idx]

    if feat is not None:
        x = torch.from_numpy(feat)
    else:  # feature-less (ddi): a xavier-initialised table that is never trained (:76-77)
        x = torch.zeros(n, dim)
        torch.nn.init.xavier_uniform_(x, generator=g) if g is not None else torch.nn.init.xavier_uniform_(x)
    ppr_edges = None
    if data_name == "ogbl-citation2":  # directed: symmetrised, reciprocal citations sum (:88-90, to_symmetric)
        ppr_edges = edge_index
        edge_index, weight = to_undirected_sum(edge_index, np.ones(edge_index.shape[1], np.float32), n)
    val_edges = split["valid"]["edge"].T if (use_val_in_test and "edge" in split["valid"]) else None
    _finish(data, data_name, edge_index, weight, n, x, eps, val_edges, cache_root=cache_root, ppr_device=ppr_device,
            ppr_threads=ppr_threads, ppr_edges=ppr_edges)
    if heart_dir is not None:  # (:131-146)
        hd = os.path.join(heart_dir, data_name)
        data["valid_neg"] = torch.from_numpy(np.load(os.path.join(hd, "heart_valid_samples.npy")))
        data["test_neg"] = torch.from_numpy(np.load(os.path.join(hd, "heart_test_samples.npy")))
        if "ddi" in data_name:
            k = data["valid_pos"].size(0) // 4
            sel = torch.randperm(data["valid_pos"].size(0), generator=g)[:k]
            data["valid_pos"], data["valid_neg"] = data["valid_pos"][sel], data["valid_neg"][sel]
            data["train_pos_val"] = data["train_pos_val"][sel]
    return data


def read_data_planetoid(data_dir: str, data_name: str, *, eps: float = 1e-4, heart_dir: Optional[str] = None,
                        cache_root: Optional[str] = None, ppr_device=None, ppr_threads: int = 0,
                        seed: Optional[int] = None) -> dict:
    """``read_data_planetoid`` (read_datasets.py:150-254): the fixed HeaRT splits of cora / citeseer / pubmed as
    tab-separated text files plus the ``gnn_feature`` tensor file.  Self loops are dropped from every split, nodes are
    counted over all positive files, the training edges are used in both directions with weight 1."""
    base = os.path.join(data_dir, data_name)
    pos = {"train": [], "valid": [], "test": []}
    neg = {"valid": [], "test": []}
    nodes = set()
    for s in ("train", "test", "valid"):
        with open(os.path.join(base, f"{s}_pos.txt")) as f:
            for line in f:
                a, b = (int(v) for v in line.strip().split("\t"))
                nodes.add(a)
                nodes.add(b)
                if a != b:
                    pos[s].append((a, b))
    for s in ("test", "valid"):
        with open(os.path.join(base, f"{s}_neg.txt")) as f:
            for line in f:
                a, b = (int(v) for v in line.strip().split("\t"))
                neg[s].append((a, b))
    n = len(nodes)
    train = np.asarray(pos["train"], np.int64).reshape(-1, 2)
    edge_index = np.concatenate([train.T, train.T[::-1]], axis=1)
    feat = torch.load(os.path.join(base, "gnn_feature"), weights_only=False)["entity_embedding"]
    data = {"dataset": data_name}
    t = lambda a: torch.from_numpy(np.asarray(a, np.int64).reshape(-1, 2))  # noqa: E731
    data["train_pos"], data["valid_pos"], data["test_pos"] = t(pos["train"]), t(pos["valid"]), t(pos["test"])
    data["valid_neg"], data["test_neg"] = t(neg["valid"]), t(neg["test"])
    g = None if seed is None else torch.Generator().manual_seed(seed)
    idx = torch.randperm(data["train_pos"].size(0), generator=g)[: data["valid_pos"].size(0)]
    data["train_pos_val"] = data["train_pos"][idx]
    _finish(data, data_name, edge_index, None, n, torch.as_tensor(feat, dtype=torch.float32), eps, None,
            cache_root=cache_root, ppr_device=ppr_device, ppr_threads=ppr_threads)
    if heart_dir is not None:
        hd = os.path.join(heart_dir, data_name)
        data["valid_neg"] = torch.from_numpy(np.load(os.path.join(hd, "heart_valid_samples.npy")))
        data["test_neg"] = torch.from_numpy(np.load(os.path.join(hd, "heart_test_samples.npy")))
    return data
