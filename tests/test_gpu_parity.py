"""Parity of the HIP path (through the C ABI) against the golden vectors recorded from the reference and against
the CPU oracle.  Run on the GPU box: python -m pytest tests -m gpu."""
import numpy as np
import pytest
import torch

import lpformer_amd
from lpformer_amd import graph
from oracle import lpformer_oracle as O
from tests.golden_util import LP_CASES, Fixture

pytestmark = pytest.mark.gpu
TOL = 1e-4  # north_star: link logits within 1e-4 in fp32


def _build(fx, sparse_inputs=True):
    """Model + score head on cuda:0 from a fixture; graph entries given as torch sparse COO tensors (the types the
    reference's data dict holds) or as CSR containers."""
    n, dev = fx.n, torch.device("cuda:0")
    data = {"x": torch.from_numpy(fx["x"]).to(dev), "num_nodes": n}

    def pack(prefix, ei_key, w_key, ppr_prefix):
        ei = fx[ei_key].astype(np.int64)
        adj_t = graph.csr_from_coo(ei[0], ei[1], fx[w_key], n)
        mask = graph.mask_csr(ei, n, symmetric=True)
        ppr = graph.csr_from_coo(fx[ppr_prefix + "row"], fx[ppr_prefix + "col"], fx[ppr_prefix + "val"], n)
        if sparse_inputs:
            return adj_t.to_torch_sparse_coo().to(dev), mask.to_torch_sparse_coo().to(dev).int(), \
                ppr.to_torch_sparse_coo().to(dev)
        return adj_t, mask, ppr

    data["adj_t"], data["adj_mask"], data["ppr"] = pack("", "edge_index", "edge_weight", "ppr_")
    if fx.test_set:
        data["full_adj_t"], data["full_adj_mask"], data["ppr_test"] = pack("full_", "full_edge_index",
                                                                            "full_edge_weight", "ppr_test_")
    else:
        data["full_adj_t"], data["full_adj_mask"], data["ppr_test"] = data["adj_t"], data["adj_mask"], data["ppr"]
    cfg = {k: fx.cfg[k] for k in ("thresh_cn", "thresh_1hop", "thresh_non1hop", "dim", "trans_layers", "num_heads",
                                  "att_drop", "dropout", "gnn_drop", "feat_drop", "gcn_cache", "gnn_layers",
                                  "residual", "layer_norm", "relu")}
    model = lpformer_amd.LinkTransformer(cfg, data, device=dev).to(dev)
    score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, fx.cfg["pred_layers"]).to(dev)
    m_sd, s_sd = fx.state_dicts()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in m_sd.items()}, strict=True)
    score.load_state_dict({k: torch.from_numpy(v) for k, v in s_sd.items()}, strict=True)
    return model.eval(), score.eval()


def _err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


@pytest.mark.parametrize("case", LP_CASES)
@pytest.mark.parametrize("indexed", [True, False])  # indexed fast path and general PPR-streaming kernel
def test_selection_bit_exact_vs_reference(case, indexed):
    fx = Fixture(case)
    model, _ = _build(fx)
    model.use_select_index = indexed
    infos = model.compute_node_mask(torch.from_numpy(fx["batch"]), test_set=fx.test_set)
    for tag, info in zip(("cn", "onehop", "non1hop"), infos):
        if f"sel_{tag}_ix" not in fx:
            assert info is None
            continue
        ix, pa, pb = (t.cpu().numpy() for t in info)
        np.testing.assert_array_equal(ix, fx[f"sel_{tag}_ix"])
        np.testing.assert_array_equal(pa.view(np.uint32), fx[f"sel_{tag}_pa"].view(np.uint32))
        np.testing.assert_array_equal(pb.view(np.uint32), fx[f"sel_{tag}_pb"].view(np.uint32))


@pytest.mark.parametrize("case", LP_CASES)
@pytest.mark.parametrize("sparse_inputs", [True, False])
def test_forward_vs_reference(case, sparse_inputs):
    fx = Fixture(case)
    model, score = _build(fx, sparse_inputs)
    batch = torch.from_numpy(fx["batch"]).cuda()
    x_node = model.propagate(test_set=fx.test_set)
    assert _err(x_node.cpu(), fx["x_node"]) <= TOL
    # the call pattern of test_heart_negatives (src/train/testing.py:105-117)
    ew = model.elementwise_lin(x_node[batch[0]] * x_node[batch[1]])
    pw, _ = model.calc_pairwise(batch, x_node, test_set=fx.test_set)
    assert _err(ew.cpu(), fx["elementwise_feats"]) <= TOL
    assert _err(model._last_att.cpu(), fx["att_post_ln"]) <= TOL
    assert _err(pw.cpu(), fx["pairwise_feats"]) <= TOL
    prob = score(torch.cat((ew, pw), dim=-1))
    assert _err(prob.cpu(), fx["prob"]) <= TOL
    # the call pattern of test_edge (src/train/testing.py:86-88)
    feats, attw = model(batch, test_set=fx.test_set, return_weights=True)
    assert _err(feats.cpu(), fx["combined_feats"]) <= TOL
    assert _err(score.logits(feats).cpu(), fx["logit"]) <= TOL
    assert _err(score(feats).cpu(), fx["prob"]) <= TOL
    np.testing.assert_array_equal(attw[0].cpu().numpy(), fx["att_weights"][0])
    assert _err(attw[1].cpu(), fx["att_weights"][1]) <= TOL


def test_vs_oracle_on_fresh_inputs():
    """Seeded inputs that are not in any fixture: HIP path vs the CPU oracle (same weights)."""
    from lpformer_amd import data as D
    rng = np.random.default_rng(11)
    n, dim = 700, 128
    ei, w = D.chung_lu_graph(n, 2600, seed=5, max_weight=5)
    x = rng.standard_normal((n, 48)).astype(np.float32)
    ppr = lpformer_amd.calc_ppr(ei, n, 0.15, 2e-4)
    d = D.build_data(ei, x, n, edge_weight=w, ppr=ppr)
    cfg = D.train_args_for(dict(thresholds=(0.0, 1e-3, 5e-3), dim=dim, gnn_layers=2, residual=False))
    dev = torch.device("cuda:0")
    model = lpformer_amd.LinkTransformer(cfg, d, device=dev).to(dev).eval()
    score = lpformer_amd.mlp_score(2 * dim, 2 * dim, 1, 2).to(dev).eval()
    with torch.no_grad():
        for p in list(model.parameters()) + list(score.parameters()):
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    P = {f"model.{k}": v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    P.update({f"score.{k}": v.detach().cpu().numpy() for k, v in score.state_dict().items()})
    batch = D.sample_pairs(ei, n, 1500, seed=2)
    batch[:, :4] = np.array([[0, 5, 7, 7], [0, 5, 9, 9]])  # a == b and duplicate pairs
    ocfg = dict(cfg, pred_layers=2)
    ref = O.forward(batch, x, O.gcn_norm(ei, w, n), O.symmetric_mask_csr(ei, n),
                    (ppr.rowptr, ppr.col.astype(np.int64), ppr.val), P, ocfg)
    feats = model(torch.from_numpy(batch))
    assert _err(feats.cpu(), ref["combined_feats"]) <= TOL
    assert _err(score.logits(feats).cpu(), ref["logit"]) <= TOL
    for indexed in (True, False):
        model.use_select_index = indexed
        infos = model.compute_node_mask(torch.from_numpy(batch))
        for tag, info in zip(("cn", "onehop", "non1hop"), infos):
            np.testing.assert_array_equal(info[0].cpu().numpy(), ref["sel"][tag][0])
            np.testing.assert_array_equal(info[1].cpu().numpy().view(np.uint32), ref["sel"][tag][1].view(np.uint32))
            np.testing.assert_array_equal(info[2].cpu().numpy().view(np.uint32), ref["sel"][tag][2].view(np.uint32))
    # a caller-supplied adjacency override (the training loop's masked adjacency) takes the general kernel
    infos = model.compute_node_mask(torch.from_numpy(batch), adj=d["adj_mask"].to_torch_sparse_coo())
    np.testing.assert_array_equal(infos[1][0].cpu().numpy(), ref["sel"]["onehop"][0])
    # size-independent properties: permuting the batch permutes the scores; swapping (a,b) leaves them unchanged
    perm = torch.randperm(batch.shape[1])
    s0 = score.logits(feats)
    s1 = score.logits(model(torch.from_numpy(batch)[:, perm]))
    assert _err(s0[perm.cuda()].cpu(), s1.cpu()) <= 1e-5
    s2 = score.logits(model(torch.from_numpy(batch[::-1].copy())))
    assert _err(s0.cpu(), s2.cpu()) <= 2e-5


def test_empty_and_tiny_batches():
    fx = Fixture("lp_all_d64")
    model, score = _build(fx)
    one = torch.from_numpy(fx["batch"][:, :1])
    f1 = model(one, test_set=fx.test_set)
    assert f1.shape == (1, 128)
    assert _err(f1.cpu(), fx["combined_feats"][:1]) <= TOL
    iso = torch.tensor([[fx.n - 1], [fx.n - 2]])  # two isolated nodes: no selected entries -> attention = bias
    fi = model(iso, test_set=fx.test_set)
    assert torch.isfinite(fi).all()


def _oracle_params(fx):
    return fx.params


def test_alternating_test_set_on_one_model():
    """The reference's ``test()`` (src/train/testing.py:141-156) calls the SAME model with test_set=False and
    test_set=True in turn; with --use-val-in-test the two graphs differ (full_adj_t, ppr_test).  Each call must see
    its own encoder output and node-level projections -- never the previous call's (cached) ones."""
    fx = Fixture("lp_all_d64_residual_valtest")
    model, score = _build(fx)
    n = fx.n
    batch = fx["batch"]
    tb = torch.from_numpy(batch).cuda()
    # expected outputs for the TRAIN graph (test_set=False) from the oracle; for the FULL graph from the fixture
    ei = fx["edge_index"].astype(np.int64)
    ppr_tr = O.csr_from_coo(fx["ppr_row"].astype(np.int64), fx["ppr_col"].astype(np.int64), fx["ppr_val"], n)
    ref_tr = O.forward(batch, fx["x"], O.gcn_norm(ei, fx["edge_weight"], n), O.symmetric_mask_csr(ei, n), ppr_tr,
                       fx.params, fx.cfg)
    assert np.abs(ref_tr["logit"] - fx["logit"]).max() > 1e-3  # the two graphs do give different scores
    for rep in range(3):
        for test_set, want_feats, want_logit in ((False, ref_tr["combined_feats"], ref_tr["logit"]),
                                                 (True, fx["combined_feats"], fx["logit"])):
            feats = model(tb, test_set=test_set)                     # test_edge pattern: encoder inside forward
            assert _err(feats.cpu(), want_feats) <= TOL, (rep, test_set)
            assert _err(score.logits(feats).cpu(), want_logit) <= TOL
            h = model.propagate(test_set=test_set)                   # HeaRT pattern: h computed inside a function
            got = model.score_pairs(tb, h, score, test_set=test_set, logits=True)
            del h
            assert _err(got.cpu(), want_logit) <= TOL, (rep, test_set)
    # changed node features are seen too (same tensor object, bumped version; then a new tensor at a recycled address)
    with torch.no_grad():
        model.data["x"].mul_(0.5)
    f_half = model(tb, test_set=True)
    assert _err(f_half.cpu(), fx["combined_feats"]) > 1e-3
    model.data["x"] = torch.from_numpy(fx["x"]).cuda()
    assert _err(model(tb, test_set=True).cpu(), fx["combined_feats"]) <= TOL


def test_masked_adjacency_override_vs_reference():
    """The training loop's overrides (src/train/train_model.py:40-59), pinned by reference output: CN / 1-hop typing
    from the adjacency WITHOUT the batch's positive edges, >1-hop exclusion from the unmasked adjacency
    (link_transformer.py:438-443), optional propagation over the masked adjacency.  Takes the general selection
    kernel (same_adj = 0).  The overrides are passed as the reference passes them: a torch sparse COO int tensor
    and a torch_sparse.SparseTensor (here: the stand-in class with the same ``coo()`` / ``sparse_sizes()`` surface)."""
    from oracle.ref_shims import SparseTensor
    fx = Fixture("lp_all_d64_maskedadj")
    model, score = _build(fx)
    n = fx.n
    keep = torch.from_numpy(fx["masked_keep_edges"].astype(np.int64))
    masked_adjt = SparseTensor.from_edge_index(keep, sparse_sizes=(n, n)).to_symmetric()
    masked_adj = masked_adjt.to_torch_sparse_coo_tensor().coalesce().bool().int().cuda()
    mb = torch.from_numpy(fx["masked_batch"])
    infos = model.compute_node_mask(mb, False, masked_adj)
    for tag, info in zip(("cn", "onehop", "non1hop"), infos):
        ix, pa, pb = (t.cpu().numpy() for t in info)
        np.testing.assert_array_equal(ix, fx[f"masked_sel_{tag}_ix"])
        np.testing.assert_array_equal(pa.view(np.uint32), fx[f"masked_sel_{tag}_pa"].view(np.uint32))
        np.testing.assert_array_equal(pb.view(np.uint32), fx[f"masked_sel_{tag}_pb"].view(np.uint32))
    # ... which took the override as a DIFFERENCE to the resident adjacency (selection over the walk indexes + the patch of
    # lpformer_amd/mask_delta.py); the same through the general kernels over a CSR built from the override, and with the
    # removed edges named explicitly (lpformer_amd.RemovedEdges) -- all three bit-exact
    assert model._delta_cache[0] is masked_adj and model._delta_cache[2] is not None and model._delta_cache[2].numel() > 0
    ei = fx["edge_index"].astype(np.int64)
    kept = set((keep[0] * n + keep[1]).tolist()) | set((keep[1] * n + keep[0]).tolist())
    removed = ei[:, [k not in kept for k in (ei[0] * n + ei[1]).tolist()]]
    assert removed.shape[1] > 0
    import lpformer_amd
    for form in ("general", "removed"):
        model.use_mask_delta = form != "general"
        model._delta_cache = None
        ov = masked_adj if form == "general" else lpformer_amd.RemovedEdges(torch.from_numpy(removed))
        infos2 = model.compute_node_mask(mb, False, ov)
        assert (model._delta_cache is None) == (form == "general")
        for tag, info in zip(("cn", "onehop", "non1hop"), infos2):
            ix, pa, pb = (t.cpu().numpy() for t in info)
            np.testing.assert_array_equal(ix, fx[f"masked_sel_{tag}_ix"])
            np.testing.assert_array_equal(pa.view(np.uint32), fx[f"masked_sel_{tag}_pa"].view(np.uint32))
            np.testing.assert_array_equal(pb.view(np.uint32), fx[f"masked_sel_{tag}_pb"].view(np.uint32))
        feats = model(mb, adj_mask=ov)            # (eval mode: the fused kernels behind a CSR of the override)
        assert _err(feats.cpu(), fx["masked_combined_feats"]) <= TOL
    model.use_mask_delta = True
    feats = model(mb, adj_mask=masked_adj)
    assert _err(feats.cpu(), fx["masked_combined_feats"]) <= TOL
    assert _err(score.logits(feats).cpu(), fx["masked_logit"]) <= TOL
    h = model.propagate(masked_adjt)
    assert _err(h.cpu(), fx["masked_prop_x_node"]) <= TOL
    # (an UNWEIGHTED override of a weighted graph, as the reference builds it: a graph of its own.  The same edges WITH the
    #  resident weights share the resident structure -- raw weights with the removed edges at 0, re-normalised by
    #  lpf_gcn_norm_csr -- and give what a graph built from them gives)
    own_prop = model._device_graph("prop", model.data["adj_t"])
    assert model._override["prop"][1].rowptr is not own_prop.rowptr

    class _Weighted:
        def __init__(self, r, c, v):
            self._r, self._c, self._v = r, c, v

        def coo(self):
            return self._r, self._c, self._v

        def sparse_sizes(self):
            return (n, n)
    km = np.array([k in kept for k in (ei[0] * n + ei[1]).tolist()])
    wov = [torch.from_numpy(ei[0][km]).cuda(), torch.from_numpy(ei[1][km]).cuda(),
           torch.from_numpy(fx["edge_weight"][km].astype(np.float32)).cuda()]
    h_shared = model.propagate(_Weighted(*wov))
    assert model._override["prop"][1].rowptr is own_prop.rowptr
    model.use_mask_delta = False
    model._override.clear()
    h_own = model.propagate(_Weighted(*wov))
    assert model._override["prop"][1].rowptr is not own_prop.rowptr
    assert _err(h_shared.cpu(), h_own.cpu().numpy()) <= 1e-5 and _err(h_own.cpu(), h.cpu().numpy()) > 1e-4
    model.use_mask_delta = True
    model._override.clear()
    feats = model(mb, adj_prop=masked_adjt, adj_mask=masked_adj)
    assert _err(feats.cpu(), fx["masked_prop_combined_feats"]) <= TOL
    assert _err(score.logits(feats).cpu(), fx["masked_prop_logit"]) <= TOL
    # the same overrides as objects that already live on the GPU (what the reference's loop hands over when its data
    # sits on the device): converted there, no host round trip
    class _GpuSparseTensor:
        def __init__(self, st):
            r, c, v = st.coo()
            self._r, self._c, self._v, self._n = r.cuda(), c.cuda(), None if v is None else v.cuda(), st.sparse_sizes()

        def coo(self):
            return self._r, self._c, self._v

        def sparse_sizes(self):
            return self._n
    h_dev = model.propagate(_GpuSparseTensor(masked_adjt))
    assert _err(h_dev.cpu(), fx["masked_prop_x_node"]) <= TOL
    feats = model(mb, adj_prop=_GpuSparseTensor(masked_adjt), adj_mask=_GpuSparseTensor(masked_adjt))
    assert _err(feats.cpu(), fx["masked_prop_combined_feats"]) <= TOL
    # without the override the same batch gives the unmasked result again (no stale override state)
    plain = model.compute_node_mask(mb, False, None)
    assert plain[1][0].shape[1] != fx["masked_sel_onehop_ix"].shape[1]
    # a fresh override per batch (what the training loop does) must not accumulate device copies
    before = len(model._graphs)
    for i in range(4):
        k2 = keep[:, torch.randperm(keep.shape[1])[: keep.shape[1] - 10 * (i + 1)]]
        adj_i = SparseTensor.from_edge_index(k2, sparse_sizes=(n, n)).to_symmetric()
        model(mb, adj_prop=adj_i, adj_mask=adj_i.to_torch_sparse_coo_tensor().coalesce().bool().int())
    assert len(model._graphs) == before and len(model._override) <= 2


def test_invalid_inputs_are_rejected_or_harmless():
    fx = Fixture("lp_all_d64")
    model, score = _build(fx)
    empty = torch.zeros(2, 0, dtype=torch.int64)
    assert model(empty, test_set=fx.test_set).shape == (0, 128)
    sel = model.compute_node_mask(empty, test_set=fx.test_set)
    assert all(t[0].shape == (2, 0) for t in sel if t is not None)
    with pytest.raises(Exception):
        model(torch.tensor([[0, fx.n], [1, 2]]), test_set=fx.test_set)       # node id out of range
    with pytest.raises(Exception):
        model(torch.tensor([[0, -1], [1, 2]]), test_set=fx.test_set)


def test_forward_reuses_encoder_output_only_while_nothing_changed():
    """forward() in eval mode keeps the encoder output across calls (the reference re-runs the encoder per batch on the
    same inputs): the result must equal a fresh propagate after every kind of change -- encoder weights updated in
    place, a loaded state dict, features changed in place, the other graph of test_set -- and the switch must turn
    the reuse off."""
    fx = Fixture("lp_all_d64_residual_valtest")
    model, score = _build(fx)
    batch = torch.from_numpy(fx["batch"])

    def fresh(test_set):
        return model.pair_features(batch, model.propagate(None, test_set), test_set=test_set)

    for ts in (False, True, False):
        assert torch.equal(model(batch, test_set=ts), fresh(ts))
    first = model(batch, test_set=False)
    calls = {"n": 0}
    orig = model.propagate
    model.propagate = lambda *a, **k: (calls.__setitem__("n", calls["n"] + 1), orig(*a, **k))[1]
    assert torch.equal(model(batch, test_set=False), first) and calls["n"] == 0          # reused
    with torch.no_grad():
        model.node_encoder.gnn_encoder.convs[0].lin.weight.mul_(1.01)                     # optimiser-style update
    out = model(batch, test_set=False)
    assert calls["n"] == 1 and not torch.equal(out, first)
    model.propagate = orig
    assert torch.equal(out, fresh(False))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    sd["gnn_norm.weight"] = sd["gnn_norm.weight"] * 0.5
    model.load_state_dict(sd)
    assert torch.equal(model(batch, test_set=False), fresh(False))
    with torch.no_grad():
        model.data["x"].add_(0.25)                                                        # features changed in place
    assert torch.equal(model(batch, test_set=False), fresh(False))
    model.reuse_encoder_output = False
    assert torch.equal(model(batch, test_set=True), fresh(True)) and model._enc_cache is not None


@pytest.mark.parametrize("case", ["lp_all_d64_heads2", "lp_all_d64_layers2"])
def test_two_heads_and_two_layers_vs_reference(case):
    """num_heads = 2 (src/modules/layers.py:129-135,180-224) and trans_layers = 2 (src/models/link_transformer.py:55-62,
    150-152,167-168), recorded from the reference.  Heads: every head attends with its block of lin_l / lin_r and its
    row of att over the shared selection, the blocks are concatenated, post_att_norm and pairwise_lin run over 2 D
    (+ counts) features.  Layers: the first is 2 D wide, the halves of its output are the second one's "edge" input, both
    attend over the same selection and positional encodings.  Through forward, calc_pairwise, pair_features and
    score_pairs, which take the layer-by-layer, head-by-head path of lpformer_amd/train.py in evaluation mode; the
    one-layer one-head entry points (recorded plans, return_weights) refuse loudly."""
    import lpformer_amd
    fx = Fixture(case)
    assert (fx.cfg["num_heads"], fx.cfg["trans_layers"]) in ((2, 1), (1, 2))
    model, score = _build(fx)
    assert model._multi_head and model.att_layers[0].att.lin_r.weight.shape == (128, 128)
    assert len(model.att_layers) == fx.cfg["trans_layers"]
    batch = torch.from_numpy(fx["batch"]).cuda()
    x_node = model.propagate(test_set=fx.test_set)
    assert _err(x_node.cpu(), fx["x_node"]) <= TOL
    infos = model.compute_node_mask(batch, fx.test_set)
    for tag, info in zip(("cn", "onehop", "non1hop"), infos):
        np.testing.assert_array_equal(info[0].cpu().numpy(), fx[f"sel_{tag}_ix"])
    pw, attw = model.calc_pairwise(batch, x_node, test_set=fx.test_set)
    assert attw is None and _err(pw.cpu(), fx["pairwise_feats"]) <= TOL
    feats = model(batch, test_set=fx.test_set)
    assert feats.shape == (batch.shape[1], 128) and _err(feats.cpu(), fx["combined_feats"]) <= TOL
    assert _err(model.pair_features(batch, x_node, test_set=fx.test_set).cpu(), fx["combined_feats"]) <= TOL
    assert _err(score.logits(feats).cpu(), fx["logit"]) <= TOL
    assert _err(model.score_pairs(batch, x_node, score, test_set=fx.test_set).cpu(), fx["prob"]) <= TOL
    assert _err(model.score_pairs(batch, x_node, score, test_set=fx.test_set, logits=True).cpu(), fx["logit"]) <= TOL
    with pytest.raises(NotImplementedError):
        model(batch, test_set=fx.test_set, return_weights=True)
    with pytest.raises(NotImplementedError):
        lpformer_amd.PlannedScorer(model, score, x_node, batch)
    # the evaluation sweep (lpformer_amd/evaluate.py) takes the eager path for such a model
    from lpformer_amd import evaluate
    sweep = evaluate.score_edges(model, score, batch.t(), batch_size=64, h=x_node, test_set=fx.test_set, logits=True)
    assert _err(sweep.cpu(), fx["logit"]) <= TOL
