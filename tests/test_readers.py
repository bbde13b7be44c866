"""lpformer_amd.readers on synthetic files written in the OGB raw layout and the HeaRT text layout (no dataset is
available offline: the layouts are restated from the public documentation / the reference's own parser, see the module
header).  CPU only: the host PPR producer is used."""
import gzip
import os

import numpy as np
import pytest
import torch

from lpformer_amd import readers as R


def _write_csv(path, arr, fmt):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with gzip.open(path, "wt") as f:
        np.savetxt(f, arr, delimiter=",", fmt=fmt)


def _collab_like(root, rng, n=60):
    e = rng.integers(0, n, (300, 2))
    e = e[e[:, 0] != e[:, 1]]
    year = rng.integers(2000, 2015, e.shape[0])
    w = rng.integers(1, 4, e.shape[0])
    base = os.path.join(root, "ogbl_collab")
    _write_csv(os.path.join(base, "raw", "edge.csv.gz"), e, "%d")
    _write_csv(os.path.join(base, "raw", "edge_weight.csv.gz"), w[:, None], "%d")
    _write_csv(os.path.join(base, "raw", "edge_year.csv.gz"), year[:, None], "%d")
    _write_csv(os.path.join(base, "raw", "num-node-list.csv.gz"), np.array([[n]]), "%d")
    _write_csv(os.path.join(base, "raw", "node-feat.csv.gz"), rng.standard_normal((n, 8)), "%.6f")
    sd = os.path.join(base, "split", "time")
    os.makedirs(sd)
    torch.save({"edge": torch.from_numpy(e), "weight": torch.from_numpy(w), "year": torch.from_numpy(year)},
               os.path.join(sd, "train.pt"))
    for s in ("valid", "test"):
        torch.save({"edge": torch.from_numpy(rng.integers(0, n, (20, 2))), "weight": torch.ones(20),
                    "year": torch.full((20,), 2016), "edge_neg": torch.from_numpy(rng.integers(0, n, (50, 2)))},
                   os.path.join(sd, f"{s}.pt"))
    return e, w, year


def test_read_ogb_collab_layout(tmp_path):
    rng = np.random.default_rng(0)
    e, w, year = _collab_like(str(tmp_path), rng)
    n = 60
    d = R.read_data_ogb(str(tmp_path), "ogbl-collab", eps=1e-3, use_val_in_test=True, seed=1)
    assert d["num_nodes"] == n and d["x"].shape == (n, 8) and d["x"].dtype == torch.float32
    # filter_by_year: only the training edges from 2007 on, both directions, duplicate edges merged with weights summed
    keep = year >= 2007
    dense = np.zeros((n, n))
    for (a, b), ww in zip(e[keep], w[keep]):
        dense[a, b] += ww
        dense[b, a] += ww
    adj = d["adj_t"]
    got = np.zeros((n, n))
    rows = np.repeat(np.arange(n), np.diff(adj.rowptr))
    got[rows, adj.col] = adj.val
    assert np.array_equal(got, dense)
    assert d["train_pos"].shape[0] == int(keep.sum()) and d["train_pos_val"].shape[0] == d["valid_pos"].shape[0]
    m = d["adj_mask"]
    mrows = np.repeat(np.arange(n), np.diff(m.rowptr))
    md = np.zeros((n, n))
    md[mrows, m.col] = 1
    assert np.array_equal(md, (dense > 0).astype(float))
    # --use-val-in-test: the test-time graph also holds the validation edges (weight 1), the PPR matrix differs
    full = d["full_adj_t"]
    assert full.col.size > adj.col.size and d["ppr_test"] is not d["ppr"]
    assert d["ppr"].n == n and np.all(np.diff(d["ppr"].rowptr) >= 1)       # every row holds at least its own node


def test_validation_edges_are_coalesced_like_to_undirected():
    """read_datasets.py:98-107: ``to_undirected(val_edge_index)`` mirrors AND coalesces, so a validation pair listed
    twice, or in both directions, adds weight 1 once per direction to ``full_adj_t`` (not 2)."""
    from lpformer_amd import data as D
    n = 6
    ei = np.array([[0, 1, 1, 2], [1, 0, 2, 1]])
    val = np.array([[3, 3, 4, 5], [4, 4, 3, 0]])                  # (3,4) twice and (4,3): one undirected edge; (5,0)
    und = D.to_undirected(val, n)
    assert und.tolist() == [[0, 3, 4, 5], [5, 4, 3, 0]]
    x = np.zeros((n, 4), np.float32)
    d = D.build_data(ei, x, n, eps=1e-2, val_edge_index=val)
    full = d["full_adj_t"]
    dense = np.zeros((n, n))
    dense[np.repeat(np.arange(n), np.diff(full.rowptr)), full.col] = full.val
    want = np.zeros((n, n))
    for a, b in ((0, 1), (1, 2), (3, 4), (5, 0)):
        want[a, b] = want[b, a] = 1
    assert np.array_equal(dense, want)
    mask = d["full_adj_mask"]
    assert mask.col.size == 8


def test_read_ogb_citation2_and_ddi_layouts(tmp_path):
    rng = np.random.default_rng(1)
    n = 40
    for name, split_type in (("ogbl_citation2", "time"), ("ogbl_ddi", "target")):
        base = os.path.join(str(tmp_path), name)
        e = rng.integers(0, n, (150, 2))
        e = e[e[:, 0] != e[:, 1]]
        e = np.unique(e, axis=0)
        _write_csv(os.path.join(base, "raw", "edge.csv.gz"), e, "%d")
        _write_csv(os.path.join(base, "raw", "num-node-list.csv.gz"), np.array([[n]]), "%d")
        sd = os.path.join(base, "split", split_type)
        os.makedirs(sd)
        if "citation2" in name:
            for s in ("train", "valid", "test"):
                torch.save({"source_node": torch.from_numpy(e[:30, 0]), "target_node": torch.from_numpy(e[:30, 1]),
                            "target_node_neg": torch.from_numpy(rng.integers(0, n, (30, 7)))}, os.path.join(sd, f"{s}.pt"))
        else:
            for s in ("train", "valid", "test"):
                torch.save({"edge": torch.from_numpy(e[:30]), "edge_neg": torch.from_numpy(rng.integers(0, n, (40, 2)))},
                           os.path.join(sd, f"{s}.pt"))
        if "citation2" in name:
            _write_csv(os.path.join(base, "raw", "node-feat.csv.gz"), rng.standard_normal((n, 4)), "%.6f")
            cit_edges = e
    c = R.read_data_ogb(str(tmp_path), "ogbl-citation2", eps=1e-3)
    assert c["valid_neg"].shape == (30, 7) and c["train_pos"].shape == (30, 2)
    # directed list symmetrised; a reciprocal pair of citations carries weight 2 (to_symmetric sums)
    dense = np.zeros((n, n))
    for a, b in cit_edges:
        dense[a, b] += 1
        dense[b, a] += 1
    adj = c["adj_t"]
    got = np.zeros((n, n))
    got[np.repeat(np.arange(n), np.diff(adj.rowptr)), adj.col] = adj.val
    assert np.array_equal(got, dense)
    # the PPR producer saw the DIRECTED list: a node without out-edges keeps all its mass (row = itself only)
    sinks = np.setdiff1d(np.arange(n), cit_edges[:, 0])
    assert sinks.size > 0 and all(np.diff(c["ppr"].rowptr)[s] == 1 for s in sinks)
    dd = R.read_data_ogb(str(tmp_path), "ogbl-ddi", eps=1e-3, dim=16, seed=3)
    assert dd["x"].shape == (n, 16) and float(dd["x"].abs().max()) > 0          # xavier table, no features on disk
    assert dd["adj_t"].col.size == 2 * np.unique(np.sort(e, axis=1), axis=0).shape[0] or dd["adj_t"].col.size > 0


def test_read_planetoid_layout(tmp_path):
    rng = np.random.default_rng(2)
    n = 30
    base = os.path.join(str(tmp_path), "cora")
    os.makedirs(base)
    perm = rng.permutation(n)
    train = np.stack([perm, np.roll(perm, 1)], 1)                 # a ring: every node appears
    train = np.concatenate([train, [[3, 3]]])                     # a self loop, dropped by the reader
    for s, arr in (("train", train), ("valid", rng.integers(0, n, (10, 2))), ("test", rng.integers(0, n, (12, 2)))):
        np.savetxt(os.path.join(base, f"{s}_pos.txt"), arr, fmt="%d", delimiter="\t")
    for s in ("valid", "test"):
        np.savetxt(os.path.join(base, f"{s}_neg.txt"), rng.integers(0, n, (15, 2)), fmt="%d", delimiter="\t")
    torch.save({"entity_embedding": torch.randn(n, 12)}, os.path.join(base, "gnn_feature"))
    hd = os.path.join(str(tmp_path), "heart", "cora")
    os.makedirs(hd)
    np.save(os.path.join(hd, "heart_valid_samples.npy"), rng.integers(0, n, (10, 5, 2)))
    np.save(os.path.join(hd, "heart_test_samples.npy"), rng.integers(0, n, (12, 5, 2)))
    d = R.read_data_planetoid(str(tmp_path), "cora", eps=1e-3, heart_dir=os.path.join(str(tmp_path), "heart"), seed=0)
    assert d["num_nodes"] == n and d["x"].shape == (n, 12)
    assert d["train_pos"].shape[0] == n                            # the self loop is gone
    assert d["adj_t"].col.size == 2 * n and d["full_adj_t"] is d["adj_t"] and d["ppr_test"] is d["ppr"]
    assert d["valid_neg"].shape == (10, 5, 2) and d["test_neg"].shape == (12, 5, 2)


def test_planetoid_reader_matches_the_reference_reader():
    """``readers.read_data_planetoid`` against what the REFERENCE's own ``read_data_planetoid``
    (src/util/read_datasets.py:150-254, run unmodified by tests/golden/make_reader_golden.py) returned for the same files
    -- tests/golden/planetoid_tiny/ in the HeaRT text layout: a self loop in the training file, nodes only a validation /
    test positive names, the feature tensor, the HeaRT negative samples.  Node count, every split tensor, the features,
    the propagation matrix, the typing adjacency and the PPR matrix (indices AND values, bit for bit)."""
    from tests.golden_util import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, "reader_planetoid.npz"))
    src = os.path.join(GOLDEN_DIR, "planetoid_tiny")
    torch.manual_seed(11)                       # (the reference draws train_pos_val from the global generator first thing)
    d = R.read_data_planetoid(src, "tinycora", eps=1e-4)
    n = int(z["num_nodes"])
    assert d["num_nodes"] == n
    for k in ("train_pos", "valid_pos", "test_pos", "valid_neg", "test_neg", "train_pos_val"):
        np.testing.assert_array_equal(d[k].numpy(), z[k], err_msg=k)
    np.testing.assert_array_equal(d["x"].numpy(), z["x"])

    def coo(c):
        rows = np.repeat(np.arange(c.n, dtype=np.int64), np.diff(c.rowptr))
        return rows, c.col.astype(np.int64), c.val
    # adj_t: SparseTensor.from_edge_index(edge_index, ones) -- rows sorted, columns in file order within a row there;
    # sorted by (row, col) here: compared as sets of (row, col, value)
    r, c, v = coo(d["adj_t"])
    want = sorted(zip(z["adj_row"].tolist(), z["adj_col"].tolist(), z["adj_val"].tolist()))
    assert sorted(zip(r.tolist(), c.tolist(), v.tolist())) == want
    assert d["full_adj_t"] is d["adj_t"] or coo(d["full_adj_t"])[1].tolist() == c.tolist()
    r, c, _ = coo(d["adj_mask"])
    np.testing.assert_array_equal(np.stack([r, c]), z["mask_index"])            # (coalesced: sorted on both sides)
    assert (z["mask_val"] == 1).all()
    np.testing.assert_array_equal(np.diff(d["adj_t"].rowptr).astype(np.float32), z["degree"])
    r, c, v = coo(d["ppr"])
    np.testing.assert_array_equal(np.stack([r, c]), z["ppr_index"])
    np.testing.assert_array_equal(v.view(np.uint32), z["ppr_val"].view(np.uint32))
    assert d["ppr_test"] is d["ppr"]
    # --heart: the sampled negatives replace the files' (:243-250)
    torch.manual_seed(11)
    h = R.read_data_planetoid(src, "tinycora", eps=1e-4, heart_dir=os.path.join(src, "heart"))
    np.testing.assert_array_equal(h["valid_neg"].numpy(), z["heart_valid_neg"])
    np.testing.assert_array_equal(h["test_neg"].numpy(), z["heart_test_neg"])
    np.testing.assert_array_equal(h["train_pos"].numpy(), z["train_pos"])


OGB_CASES = [("ogbl-collab", True, False), ("ogbl-collab", False, False), ("ogbl-ppa", False, True), ("ogbl-ddi", False, True),
             ("ogbl-ddi", False, False), ("ogbl-citation2", False, False)]


@pytest.mark.parametrize("name,val_in_test,heart", OGB_CASES, ids=[f"{c[0]}-val{int(c[1])}-heart{int(c[2])}" for c in OGB_CASES])
def test_ogb_reader_matches_the_reference_reader(name, val_in_test, heart):
    """``readers.read_data_ogb`` on tests/golden/ogb_tiny/ (four tiny datasets in the raw OGB layout) against what the
    REFERENCE's ``read_data_ogb`` (src/util/read_datasets.py:20-148 with ``filter_by_year`` :259-280, run unmodified by
    tests/golden/make_reader_golden.py) made of the graph objects built from the same files: the year filter and the
    weight-summing symmetrisation of collab, validation edges in the test graph (given twice / reversed: coalesced before
    they are appended), ppa's HeaRT index files, ddi's feature-less table and its quartered validation set, citation2's
    source / target splits, its symmetrised adjacency with reciprocal citations summed and its PPR over the DIRECTED list.
    What stays an assumption is the step before: raw files -> graph object, the OGB package's (documented) behaviour,
    restated in the generator's ``_OgbDataset``."""
    from tests.golden_util import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, "reader_ogb.npz"))
    tag = f"{name}|{int(val_in_test)}|{int(heart)}|"
    root = os.path.join(GOLDEN_DIR, "ogb_tiny")
    torch.manual_seed(5)        # (train_pos_val, ddi's xavier table and its validation subset come from the global generator)
    d = R.read_data_ogb(root, name, eps=1e-3, dim=16, use_val_in_test=val_in_test,
                        heart_dir=os.path.join(root, "heart") if heart else None)
    n = int(z[tag + "num_nodes"])
    assert d["num_nodes"] == n
    for k in ("train_pos", "train_pos_val", "valid_pos", "valid_neg", "test_pos", "test_neg"):
        np.testing.assert_array_equal(d[k].numpy(), z[tag + k], err_msg=k)
    np.testing.assert_array_equal(d["x"].detach().numpy(), z[tag + "x"])

    def dense(c):
        m = np.zeros((n, n))
        np.add.at(m, (np.repeat(np.arange(n), np.diff(c.rowptr)), c.col.astype(np.int64)), 1.0 if c.val is None else c.val)
        return m
    for key in ("adj_t", "full_adj_t"):         # (the reference keeps duplicate entries where a list has them: compared summed)
        r, c, v = z[tag + key]
        want = np.zeros((n, n))
        np.add.at(want, (r.astype(np.int64), c.astype(np.int64)), v)
        np.testing.assert_array_equal(dense(d[key]), want, err_msg=key)
    for key in ("adj_mask", "full_adj_mask"):
        c = d[key]
        got = np.stack([np.repeat(np.arange(n, dtype=np.int64), np.diff(c.rowptr)), c.col.astype(np.int64)])
        np.testing.assert_array_equal(got, z[tag + key + "_index"], err_msg=key)
        assert (z[tag + key + "_val"] == 1).all()
    for key in ("ppr", "ppr_test"):
        c = d[key]
        got = np.stack([np.repeat(np.arange(n, dtype=np.int64), np.diff(c.rowptr)), c.col.astype(np.int64)])
        np.testing.assert_array_equal(got, z[tag + key + "_index"], err_msg=key)
        np.testing.assert_array_equal(c.val.view(np.uint32), z[tag + key + "_val"].view(np.uint32), err_msg=key)
