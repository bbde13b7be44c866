"""Generates tests/golden/planetoid_tiny/ + tests/golden/reader_planetoid.npz (input files in the HeaRT text layout and
what the REFERENCE's own reader makes of them) and tests/golden/ogb_tiny/ + reader_ogb.npz (four tiny datasets in the raw
OGB layout and what the reference's ``read_data_ogb`` makes of the graph objects a documented-behaviour stand-in of the
OGB package builds from them) -- run in the build container, where /root/reference exists:

    python tests/golden/make_reader_golden.py

The reference's ``read_data_planetoid`` (src/util/read_datasets.py:150-254) is imported unmodified (third-party packages
replaced by oracle/ref_shims.py, as in make_golden.py) and pointed at the generated directory; its PPR goes through the
reference's own ``get_ppr_matrix`` / ``create_sparse_ppr_matrix`` without the disk cache (``get_ppr`` writes under the
reference tree, which is read-only).  Only DATA is written: the synthetic input files and the arrays the reference
returned for them.  tests/test_readers.py::test_planetoid_reader_matches_the_reference_reader reads both."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")
from oracle import ref_shims  # noqa: E402

ref_shims.install()

NAME = "tinycora"
OUT_DIR = os.path.join(HERE, "planetoid_tiny")


def write_inputs():
    """A 260-node graph in the layout read_datasets.py:160-215 parses: tab-separated pairs, a self loop in the training
    file (dropped by the reader but its node counted), nodes that appear only in a validation / test positive, the
    feature tensor under 'entity_embedding', the HeaRT negative samples as .npy."""
    rng = np.random.default_rng(7)
    n = 260
    base = os.path.join(OUT_DIR, NAME)
    os.makedirs(base, exist_ok=True)
    os.makedirs(os.path.join(OUT_DIR, "heart", NAME), exist_ok=True)
    deg_w = (np.arange(1, n - 4 + 1) ** -0.8)
    deg_w /= deg_w.sum()
    chain = rng.permutation(n - 4)                      # every one of those ids is named at least once
    pos = [(int(a), int(b)) for a, b in zip(chain[:-1], chain[1:])]
    seen = set(pos)
    rng.shuffle(pos)
    while len(pos) < 1100:
        a, b = rng.choice(n - 4, 2, p=deg_w)          # the last four ids stay out of the training edges
        if a != b and (a, b) not in seen and (b, a) not in seen:
            seen.add((a, b))
            pos.append((int(a), int(b)))
    pos = np.array(pos)
    train, valid, test = pos[:900], pos[900:1000], pos[1000:]
    train = np.concatenate([train[:450], [[17, 17]], train[450:]])            # a self loop: skipped (:169-170)
    valid = np.concatenate([valid, [[n - 4, 3], [n - 3, n - 4]]])             # nodes only the validation file names
    test = np.concatenate([test, [[n - 2, 5], [9, n - 1]]])
    for nm, arr in (("train", train), ("valid", valid), ("test", test)):
        with open(os.path.join(base, f"{nm}_pos.txt"), "w") as f:
            f.writelines(f"{a}\t{b}\n" for a, b in arr)
    for nm, k in (("valid", 102), ("test", 102)):
        with open(os.path.join(base, f"{nm}_neg.txt"), "w") as f:
            f.writelines(f"{a}\t{b}\n" for a, b in rng.integers(0, n, (k, 2)))
    torch.save({"entity_embedding": torch.from_numpy(rng.standard_normal((n, 12)).astype(np.float32))},
               os.path.join(base, "gnn_feature"))
    np.save(os.path.join(OUT_DIR, "heart", NAME, "heart_valid_samples.npy"), rng.integers(0, n, (102, 20, 2)))
    np.save(os.path.join(OUT_DIR, "heart", NAME, "heart_test_samples.npy"), rng.integers(0, n, (102, 20, 2)))
    return n


def main():
    n = write_inputs()
    import util.calc_ppr_scores as cps
    import util.read_datasets as rd
    rd.DATA_DIR = OUT_DIR
    rd.HEART_DIR = os.path.join(OUT_DIR, "heart")

    def get_ppr_no_disk(dataset, edge_index, num_nodes, alpha, eps, is_val):
        nb, w = cps.get_ppr_matrix(edge_index, num_nodes, alpha, eps)
        return cps.create_sparse_ppr_matrix(nb, w).to_torch_sparse_coo_tensor()
    rd.get_ppr = get_ppr_no_disk
    out = {}
    for heart in (False, True):
        args = types.SimpleNamespace(data_name=NAME, eps=1e-4, heart=heart)
        torch.manual_seed(11)
        d = rd.read_data_planetoid(args, "cpu")
        tag = "heart_" if heart else ""
        if not heart:
            assert d["num_nodes"] == n
            out["num_nodes"] = np.int64(d["num_nodes"])
            out["edge_index"] = d["edge_index"].numpy()
            for k in ("train_pos", "train_pos_val", "valid_pos", "test_pos"):
                out[k] = d[k].numpy()
            out["x"] = d["x"].numpy()
            r, c, v = d["adj_t"].coo()
            out["adj_row"], out["adj_col"], out["adj_val"] = r.numpy(), c.numpy(), v.numpy()
            m = d["adj_mask"]
            out["mask_index"], out["mask_val"] = m.indices().numpy(), m.values().numpy()
            out["degree"] = d["degree"].numpy()
            p = d["ppr"].coalesce()
            out["ppr_index"], out["ppr_val"] = p.indices().numpy(), p.values().numpy()
            assert d["full_adj_t"] is not None and d["ppr_test"] is d["ppr"]
        out[tag + "valid_neg"] = d["valid_neg"].numpy()
        out[tag + "test_neg"] = d["test_neg"].numpy()
    np.savez_compressed(os.path.join(HERE, "reader_planetoid.npz"), **out)
    print({k: getattr(v, "shape", v) for k, v in out.items()})


# ------------------------------------------------------------------------------------------------ OGB layouts
OGB_DIR = os.path.join(HERE, "ogb_tiny")
OGB_SPLIT = {"ogbl-collab": ("time", True), "ogbl-ppa": ("throughput", True), "ogbl-ddi": ("target", True),
             "ogbl-citation2": ("time", False)}


def _write_csv(path, arr, fmt):
    import gzip
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with gzip.open(path, "wt") as f:
        np.savetxt(f, arr, delimiter=",", fmt=fmt)


def write_ogb_inputs():
    """Four tiny datasets in the raw OGB layout lpformer_amd/readers.py reads (module header there): collab-like
    (weights, years, duplicate and reciprocal co-authorships), ppa-like (features, HeaRT index files), ddi-like (no
    features), citation2-like (directed, source / target splits, reciprocal citations)."""
    rng = np.random.default_rng(21)
    n = 180
    for name in OGB_SPLIT:
        base = os.path.join(OGB_DIR, name.replace("-", "_"))
        split_type, _ = OGB_SPLIT[name]
        e = rng.integers(0, n, (700, 2))
        e = e[e[:, 0] != e[:, 1]]
        if name == "ogbl-collab":                       # the same pair in several years, in both orientations
            e = np.concatenate([e, e[:60], e[60:120, ::-1]])
        else:
            e = np.unique(e, axis=0)
        if name == "ogbl-citation2":                    # reciprocal citations
            e = np.unique(np.concatenate([e, e[:40, ::-1]]), axis=0)
        m = e.shape[0]
        n_tr = int(0.8 * m)
        perm = rng.permutation(m)
        tr, va, te = e[perm[:n_tr]], e[perm[n_tr:n_tr + (m - n_tr) // 2]], e[perm[n_tr + (m - n_tr) // 2:]]
        _write_csv(os.path.join(base, "raw", "edge.csv.gz"), tr, "%d")          # (the graph = the training edges)
        _write_csv(os.path.join(base, "raw", "num-node-list.csv.gz"), np.array([[n]]), "%d")
        sd = os.path.join(base, "split", split_type)
        os.makedirs(sd, exist_ok=True)
        if name == "ogbl-collab":
            w = rng.integers(1, 4, tr.shape[0])
            year = rng.integers(2000, 2015, tr.shape[0])
            _write_csv(os.path.join(base, "raw", "edge_weight.csv.gz"), w[:, None], "%d")
            _write_csv(os.path.join(base, "raw", "edge_year.csv.gz"), year[:, None], "%d")
            _write_csv(os.path.join(base, "raw", "node-feat.csv.gz"), rng.standard_normal((n, 8)), "%.6f")
            torch.save({"edge": torch.from_numpy(tr), "weight": torch.from_numpy(w), "year": torch.from_numpy(year)},
                       os.path.join(sd, "train.pt"))
            for s, arr in (("valid", va), ("test", te)):
                arr = np.concatenate([arr, arr[:5], arr[5:10, ::-1]])           # validation pairs given twice / reversed
                torch.save({"edge": torch.from_numpy(arr), "weight": torch.ones(arr.shape[0], dtype=torch.int64),
                            "year": torch.full((arr.shape[0],), 2016), "edge_neg": torch.from_numpy(rng.integers(0, n, (90, 2)))},
                           os.path.join(sd, f"{s}.pt"))
        elif name == "ogbl-citation2":
            _write_csv(os.path.join(base, "raw", "node-feat.csv.gz"), rng.standard_normal((n, 6)), "%.6f")
            torch.save({"source_node": torch.from_numpy(tr[:, 0]), "target_node": torch.from_numpy(tr[:, 1])},
                       os.path.join(sd, "train.pt"))
            for s, arr in (("valid", va), ("test", te)):
                torch.save({"source_node": torch.from_numpy(arr[:, 0]), "target_node": torch.from_numpy(arr[:, 1]),
                            "target_node_neg": torch.from_numpy(rng.integers(0, n, (arr.shape[0], 9)))},
                           os.path.join(sd, f"{s}.pt"))
        else:
            if name == "ogbl-ppa":
                _write_csv(os.path.join(base, "raw", "node-feat.csv.gz"), np.eye(5)[rng.integers(0, 5, n)], "%d")
            torch.save({"edge": torch.from_numpy(tr)}, os.path.join(sd, "train.pt"))
            for s, arr in (("valid", va), ("test", te)):
                torch.save({"edge": torch.from_numpy(arr), "edge_neg": torch.from_numpy(rng.integers(0, n, (80, 2)))},
                           os.path.join(sd, f"{s}.pt"))
        hd = os.path.join(OGB_DIR, "heart", name)
        os.makedirs(hd, exist_ok=True)
        np.save(os.path.join(hd, "heart_valid_samples.npy"), rng.integers(0, n, (va.shape[0] + 10, 12, 2)))
        np.save(os.path.join(hd, "heart_test_samples.npy"), rng.integers(0, n, (te.shape[0] + 10, 12, 2)))
        if name == "ogbl-ppa":
            torch.save(torch.from_numpy(np.sort(rng.choice(va.shape[0], 30, replace=False))), os.path.join(hd, "valid_samples_index.pt"))
            torch.save(torch.from_numpy(np.sort(rng.choice(te.shape[0], 30, replace=False))), os.path.join(hd, "test_samples_index.pt"))


class _Data:
    """What the reference touches of a PyG ``Data`` object (read_datasets.py:31-129)."""
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def to(self, device):
        return self

    def __getitem__(self, k):
        return getattr(self, k)

    def __setitem__(self, k, v):
        setattr(self, k, v)


class _OgbDataset:
    """Stand-in for ``ogb.linkproppred.PygLinkPropPredDataset`` over the tiny raw files: the graph object as the OGB
    package is DOCUMENTED to build it -- edge list from raw/edge.csv.gz with the inverse edges appended for the
    undirected datasets (edge features repeated), float node features, [E, 1] edge attributes -- and the split
    dictionaries of split/<type>/*.pt.  (The package itself is not available offline: this part of the pipeline -- files to
    graph object -- stays an assumption of lpformer_amd/readers.py; everything the REFERENCE does with the object is what
    this script records.)"""
    def __init__(self, name):
        import gzip
        self.name = name
        base = os.path.join(OGB_DIR, name.replace("-", "_"))
        split_type, add_inverse = OGB_SPLIT[name]

        def csv(fn, dt):
            path = os.path.join(base, "raw", fn)
            if not os.path.isfile(path):
                return None
            with gzip.open(path, "rt") as f:
                return np.loadtxt(f, delimiter=",", dtype=dt, ndmin=2)
        e = csv("edge.csv.gz", np.int64).T
        n = int(csv("num-node-list.csv.gz", np.int64).reshape(-1)[0])
        feat, w, yr = csv("node-feat.csv.gz", np.float32), csv("edge_weight.csv.gz", np.int64), csv("edge_year.csv.gz", np.int64)
        rep = (lambda a: None if a is None else torch.from_numpy(np.concatenate([a, a]) if add_inverse else a))
        ei = np.concatenate([e, e[::-1]], axis=1) if add_inverse else e
        self._data = _Data(num_nodes=n, edge_index=torch.from_numpy(np.ascontiguousarray(ei)),
                           x=None if feat is None else torch.from_numpy(feat), edge_weight=rep(w), edge_year=rep(yr))
        self._split = {s: torch.load(os.path.join(base, "split", split_type, f"{s}.pt")) for s in ("train", "valid", "test")}

    def __getitem__(self, i):
        return self._data

    def get_edge_split(self):
        return self._split


def main_ogb():
    write_ogb_inputs()
    import util.calc_ppr_scores as cps
    import util.read_datasets as rd
    rd.HEART_DIR = os.path.join(OGB_DIR, "heart")
    rd.PygLinkPropPredDataset = lambda name: _OgbDataset(name)

    def get_ppr_no_disk(dataset, edge_index, num_nodes, alpha, eps, is_val):
        nb, w = cps.get_ppr_matrix(edge_index, num_nodes, alpha, eps)
        return cps.create_sparse_ppr_matrix(nb, w).to_torch_sparse_coo_tensor()
    rd.get_ppr = get_ppr_no_disk
    out = {}
    cases = [("ogbl-collab", True, False), ("ogbl-collab", False, False), ("ogbl-ppa", False, True), ("ogbl-ddi", False, True),
             ("ogbl-ddi", False, False), ("ogbl-citation2", False, False)]
    for name, val_in_test, heart in cases:
        args = types.SimpleNamespace(data_name=name, eps=1e-3, heart=heart, use_val_in_test=val_in_test, dim=16)
        torch.manual_seed(5)
        d = rd.read_data_ogb(args, "cpu")
        tag = f"{name}|{int(val_in_test)}|{int(heart)}|"
        out[tag + "num_nodes"] = np.int64(d["num_nodes"])
        for k in ("train_pos", "train_pos_val", "valid_pos", "valid_neg", "test_pos", "test_neg"):
            out[tag + k] = d[k].numpy()
        out[tag + "x"] = d["x"].detach().numpy()
        for key in ("adj_t", "full_adj_t"):
            r, c, v = d[key].coo()
            out[tag + key] = np.stack([r.numpy().astype(np.float64), c.numpy().astype(np.float64), v.numpy().astype(np.float64)])
        for key in ("adj_mask", "full_adj_mask", "ppr", "ppr_test"):
            t = d[key].coalesce()
            out[tag + key + "_index"], out[tag + key + "_val"] = t.indices().numpy(), t.values().numpy()
    np.savez_compressed(os.path.join(HERE, "reader_ogb.npz"), **out)
    print(len(out), "arrays;", sorted({k.split("|")[0] for k in out}))


if __name__ == "__main__":
    main()
    main_ogb()
