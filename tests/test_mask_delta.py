"""CPU tests of lpformer_amd/mask_delta.py: the training loop's masked adjacency (reference src/train/train_model.py:38-46)
taken as a DIFFERENCE to the resident adjacency -- the selection of the masked graph must be, bit for bit, the selection
of the resident graph patched over the entries that touch a removed edge (src/models/link_transformer.py:229-250,
290-291,316-317,438-443).  Checked against the oracle's ``select_nodes`` run on the masked adjacency itself."""
import numpy as np
import pytest
import torch

from lpformer_amd import data as D
from lpformer_amd import mask_delta
from lpformer_amd.ppr import calc_ppr
from oracle import lpformer_oracle as O


def _type_major(sel, tags, bs):
    """Oracle selection -> lpf_select_export's layout (type-major arrays + per-type segment pointers)."""
    pair, node, pa, pb, tp = [], [], [], [], np.zeros((3, bs + 1), np.int64)
    for t, tag in enumerate(tags):
        if tag not in sel:
            continue
        ix, a, b = sel[tag]
        pair.append(ix[0]); node.append(ix[1]); pa.append(a); pb.append(b)
        tp[t, 1:] = np.cumsum(np.bincount(ix[0], minlength=bs))
    cat = lambda xs, dt: torch.from_numpy(np.concatenate(xs).astype(dt)) if xs else torch.zeros(0, dtype=torch.float32)
    return {"sel_pair": cat(pair, np.int32), "sel_node": cat(node, np.int32), "sel_pa": cat(pa, np.float32),
            "sel_pb": cat(pb, np.float32), "type_ptr": torch.from_numpy(tp.reshape(-1))}


def _lookup_for(ppr):
    rp, col, val = ppr

    def lookup(rows, cols):
        out = np.zeros(rows.numel(), np.float32)
        for i, (r, c) in enumerate(zip(rows.tolist(), cols.tolist())):
            seg = col[rp[r]:rp[r + 1]]
            j = np.searchsorted(seg, c)
            if j < seg.size and seg[j] == c:
                out[i] = val[rp[r] + j]
        return torch.from_numpy(out)
    return lookup


@pytest.mark.parametrize("thresholds,seed", [((0.0, 1e-3, 1e-2), 0), ((0.0, 1e-2, 1.0), 1), ((0.0, 1.0, 1.0), 2),
                                             ((0.0, 0.0, 1e-3), 3), ((0.0, 1e-4, 1e-2), 4)])
def test_patched_selection_is_the_selection_of_the_masked_graph(thresholds, seed):
    n = 400
    ei, _ = D.chung_lu_graph(n, 2600, gamma=2.3, seed=seed)
    mask = O.symmetric_mask_csr(ei, n)
    p = calc_ppr(ei, n, 0.15, 2e-4)
    ppr = (p.rowptr, p.col.astype(np.int64), p.val)
    rng = np.random.default_rng(seed)
    und = ei[:, ei[0] < ei[1]]
    # the batch: positives whose edges are removed (hub endpoints repeat: several removed edges per node, and removed
    # neighbours that are common neighbours of other positives of the same batch), some pairs twice, a few non-edges
    hub = np.argsort(np.bincount(ei[0], minlength=n))[-6:]
    at_hub = und[:, np.isin(und[0], hub) | np.isin(und[1], hub)]
    pos = np.concatenate([und[:, rng.integers(0, und.shape[1], 90)], at_hub[:, rng.integers(0, at_hub.shape[1], 60)]], axis=1)
    batch = np.concatenate([pos, pos[::-1, :10], rng.integers(0, n, (2, 20))], axis=1).astype(np.int64)
    removed = pos
    rm_keys = set((removed[0] * n + removed[1]).tolist()) | set((removed[1] * n + removed[0]).tolist())
    keep = ei[:, [k not in rm_keys for k in (ei[0] * n + ei[1]).tolist()]]
    masked = O.symmetric_mask_csr(keep, n)
    mode = "cn" if thresholds[1] == 1 and thresholds[2] == 1 else ("1-hop" if thresholds[2] == 1 else "all")
    tags = ("cn", "onehop", "non1hop")
    bs = batch.shape[1]
    want = O.select_nodes(batch, masked, ppr, thresholds, n=n, adj_unmasked=mask)
    plain = O.select_nodes(batch, mask, ppr, thresholds, n=n)
    assert any(plain[t][0].shape != want[t][0].shape for t in want), "the removed edges must matter"
    own = mask_delta.edge_keys(torch.from_numpy(mask[0]), torch.from_numpy(mask[1].astype(np.int32)), n)
    # the override as the reference passes it (a coalesced COO of the kept edges) and as an explicit list
    mrow = np.repeat(np.arange(n), np.diff(masked[0]))
    rk = mask_delta.removed_from_coo(own, torch.from_numpy(mrow), torch.from_numpy(masked[1]), n, limit=1 << 20)
    rk2 = mask_delta.removed_from_edges(own, torch.from_numpy(removed), n)
    shuffled = rng.permutation(mrow.size)
    rk3 = mask_delta.removed_from_coo(own, torch.from_numpy(mrow[shuffled]), torch.from_numpy(masked[1][shuffled]), n,
                                      limit=1 << 20)
    assert torch.equal(rk, rk2) and torch.equal(rk, rk3) and rk.numel() == len(rm_keys)
    pair, node, pa, pb, tp, counts = mask_delta.patch_selection(_type_major(plain, tags, bs), torch.from_numpy(batch), rk, n,
                                                                mode, thresholds[1], _lookup_for(ppr))
    tp = tp.view(3, bs + 1).numpy()
    base = 0
    for t, tag in enumerate(tags):
        cnt = int(tp[t, bs])
        if tag not in want:
            assert cnt == 0
            continue
        ix, a, b = want[tag]
        sl = slice(base, base + cnt)
        np.testing.assert_array_equal(np.stack([pair[sl].numpy(), node[sl].numpy()]), ix)
        np.testing.assert_array_equal(pa[sl].numpy().view(np.uint32), a.view(np.uint32))
        np.testing.assert_array_equal(pb[sl].numpy().view(np.uint32), b.view(np.uint32))
        np.testing.assert_array_equal(counts[t].numpy(), np.bincount(ix[0], minlength=bs))
        base += cnt
    if mode != "cn":      # a common neighbour that lost one edge was re-typed with the one-hop round trip
        demoted = np.setdiff1d(want["onehop"][0][0] * n + want["onehop"][0][1], plain["onehop"][0][0] * n + plain["onehop"][0][1])
        assert demoted.size > 0 or thresholds[1] >= 1e-2


def test_an_override_that_is_not_a_subset_is_a_graph_of_its_own():
    n = 50
    ei, _ = D.chung_lu_graph(n, 200, seed=1)
    mask = O.symmetric_mask_csr(ei, n)
    own = mask_delta.edge_keys(torch.from_numpy(mask[0]), torch.from_numpy(mask[1].astype(np.int32)), n)
    row = np.repeat(np.arange(n), np.diff(mask[0]))
    col = mask[1]
    # one foreign edge: not a difference
    absent = next((i, j) for i in range(n) for j in range(n) if i != j and i * n + j not in set(own.tolist()))
    r2, c2 = np.append(row, absent[0]), np.append(col, absent[1])
    assert mask_delta.removed_from_coo(own, torch.from_numpy(r2), torch.from_numpy(c2), n, limit=1 << 20) is None
    # more edges removed than the limit allows
    assert mask_delta.removed_from_coo(own, torch.from_numpy(row[:10]), torch.from_numpy(col[:10]), n, limit=8) is None
    # the identity override removes nothing; an empty one removes everything
    assert mask_delta.removed_from_coo(own, torch.from_numpy(row), torch.from_numpy(col), n, limit=8).numel() == 0
    assert mask_delta.removed_from_coo(own, torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64), n,
                                       limit=1 << 20).numel() == own.numel()
    # edges the adjacency does not hold and ids out of range are ignored by the explicit form
    e = torch.tensor([[absent[0], int(row[0]), -1, n], [absent[1], int(col[0]), 3, 2]])
    rk = mask_delta.removed_from_edges(own, e, n)
    assert rk.tolist() == sorted({int(row[0]) * n + int(col[0]), int(col[0]) * n + int(row[0])})


def test_propagation_override_lands_on_the_resident_structure():
    """``--mask-input`` (src/train/train_model.py:47-56): the propagation matrix of a batch is the resident graph minus
    the batch's positives.  Its raw weights laid out on the resident GCN structure: kept edges their weight, removed
    edges 0, the diagonal its placeholder; anything else (a foreign edge, another weight, most edges gone) is rejected."""
    from lpformer_amd import graph as G
    n = 120
    ei, w = D.chung_lu_graph(n, 500, seed=3, max_weight=4)
    st = G.gcn_structure_csr(ei, w, n)
    rows = np.repeat(np.arange(n), np.diff(st.rowptr))
    keys = torch.from_numpy(rows * n + st.col.astype(np.int64))
    raw = torch.from_numpy(st.val)
    rng = np.random.default_rng(0)
    und = np.flatnonzero(ei[0] < ei[1])
    gone = set(rng.choice(und, 40, replace=False).tolist())
    gone_keys = {int(ei[0, i] * n + ei[1, i]) for i in gone} | {int(ei[1, i] * n + ei[0, i]) for i in gone}
    keep = np.array([int(a * n + b) not in gone_keys for a, b in zip(ei[0], ei[1])])
    perm = rng.permutation(int(keep.sum()))           # (any order: a SparseTensor-like object need not be sorted)
    r, c, v = (torch.from_numpy(x[keep][perm]) for x in (ei[0], ei[1], w))
    new_w = mask_delta.prop_weights_from_coo(keys, raw, r, c, v, n)
    want = st.val.copy()
    want[np.isin(rows * n + st.col, list(gone_keys))] = 0.0
    np.testing.assert_array_equal(new_w.numpy(), want)
    assert int((new_w == 0).sum()) == len(gone_keys)
    # unweighted override of a weighted graph, a foreign edge, most edges gone: graphs of their own
    assert mask_delta.prop_weights_from_coo(keys, raw, r, c, None, n) is None
    absent = next((i, j) for i in range(n) for j in range(n) if i != j and i * n + j not in set(keys.tolist()))
    r2, c2 = torch.cat([r, torch.tensor([absent[0]])]), torch.cat([c, torch.tensor([absent[1]])])
    assert mask_delta.prop_weights_from_coo(keys, raw, r2, c2, torch.cat([v, torch.ones(1)]), n) is None
    assert mask_delta.prop_weights_from_coo(keys, raw, r[:50], c[:50], v[:50], n) is None
    # self-loops of the override are ignored (the normalisation sets every diagonal entry itself)
    r3, c3 = torch.cat([r, torch.tensor([5])]), torch.cat([c, torch.tensor([5])])
    np.testing.assert_array_equal(mask_delta.prop_weights_from_coo(keys, raw, r3, c3, torch.cat([v, torch.full((1,), 7.0)]), n).numpy(), want)
    # the same difference NAMED (adj_prop=RemovedEdges(edges)): identical weights, with edges the structure does not hold,
    # self-loops, ids out of range and both orientations ignored / merged
    gone_e = np.array([[ei[0, i], ei[1, i]] for i in gone], dtype=np.int64).T
    extra = np.array([[absent[0], 7, -1, n, gone_e[1, 0]], [absent[1], 7, 3, 2, gone_e[0, 0]]], dtype=np.int64)
    named = mask_delta.prop_weights_minus_edges(keys, raw, torch.from_numpy(np.concatenate([gone_e, extra], axis=1)), n)
    np.testing.assert_array_equal(named.numpy(), want)
    np.testing.assert_array_equal(mask_delta.prop_weights_minus_edges(keys, raw, torch.zeros(2, 0, dtype=torch.int64), n).numpy(),
                                  st.val)
