"""The other BASELINE.json configurations as parity cases (GPU box): full-size or stated reduced-size synthetic
look-alikes, checked through (i) the CPU oracle on a sample of the pairs and (ii) size-independent properties."""
import numpy as np
import pytest
import torch

import lpformer_amd
from lpformer_amd import data as D
from oracle import lpformer_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(name, scale=1.0, seed=0, bs=None, ppr_device=None):
    cfg = dict(D.CONFIGS[name])
    n = int(cfg["n"] * scale)
    edges = int(cfg["edges"] * scale)
    ei, w = D.chung_lu_graph(n, edges, gamma=cfg["gamma"], seed=seed, max_weight=cfg["max_weight"])
    x = np.random.default_rng(seed + 1).standard_normal((n, cfg["f_in"])).astype(np.float32)
    data = D.build_data(ei, x, n, edge_weight=w, eps=cfg["eps"], ppr_device=ppr_device)
    args = D.train_args_for(cfg)
    torch.manual_seed(seed)
    model = lpformer_amd.LinkTransformer(args, data, device=DEV).to(DEV).eval()
    score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, 2).to(DEV).eval()
    batch = D.sample_pairs(ei, n, bs or cfg["batch"], seed=seed + 2)
    return cfg, n, ei, w, x, data, args, model, score, batch


def _oracle_sample(model, score, data, args, batch, x_node, k=192):
    P = {f"model.{a}": v.detach().cpu().numpy() for a, v in model.state_dict().items()}
    P.update({f"score.{a}": v.detach().cpu().numpy() for a, v in score.state_dict().items()})
    mask, ppr = data["adj_mask"], data["ppr"]
    sample = batch[:, :k]
    return sample, O.forward(sample, None, None, (mask.rowptr, mask.col.astype(np.int64)),
                             (ppr.rowptr, ppr.col.astype(np.int64), ppr.val), P, dict(args, pred_layers=2),
                             x_node=x_node.cpu().numpy())


def _check_encoder_rows(model, data, args, x, n_rows=96, seed=7):
    """Every encoder layer at full size against the oracle on a sample of rows (hub rows included): the layer's
    INPUT is taken from the device, the oracle computes A_hat[rows, :] (X W^T) + b -> LN -> ReLU -> residual for the
    sampled rows only (the dense transform restricted to the rows' neighbours), and the final ``gnn_norm``."""
    layers = []
    h = model.propagate(_layers_out=layers)
    adj = data["adj_t"]
    n = adj.n
    row = np.repeat(np.arange(n, dtype=np.int64), np.diff(adj.rowptr))
    a_rp, a_col, a_val = O.gcn_norm(np.stack([row, adj.col.astype(np.int64)]), adj.val, n)
    deg = np.diff(a_rp)
    rng = np.random.default_rng(seed)
    rows = np.unique(np.concatenate([rng.integers(0, n, n_rows), np.argsort(deg)[-8:], np.argsort(deg)[:4]]))
    P = {f"model.{k}": v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    pre = "model.node_encoder.gnn_encoder"
    worst = 0.0
    for i in range(args["gnn_layers"]):
        xin = layers[i].cpu().numpy()
        want = np.zeros((rows.size, args["dim"]), np.float32)
        wt = P[f"{pre}.convs.{i}.lin.weight"]
        for j, r in enumerate(rows):
            nb, wv = a_col[a_rp[r]:a_rp[r + 1]], a_val[a_rp[r]:a_rp[r + 1]]
            want[j] = (O.linear(xin[nb], wt) * wv[:, None]).sum(axis=0, dtype=np.float32)
        want = want + P[f"{pre}.convs.{i}.bias"]
        if args["layer_norm"]:
            want = O.layer_norm(want, P[f"{pre}.lns.{i}.weight"], P[f"{pre}.lns.{i}.bias"])
        if args["relu"]:
            want = np.maximum(want, 0)
        if args["residual"] and xin.shape[1] == want.shape[1]:
            want = xin[rows] + want
        if i == args["gnn_layers"] - 1:
            want = O.layer_norm(want, P["model.gnn_norm.weight"], P["model.gnn_norm.bias"])
        got = layers[i + 1][torch.from_numpy(rows).to(DEV)].cpu().numpy()
        worst = max(worst, float(np.abs(got - want).max()))
    assert worst <= 1e-4, f"encoder rows differ from the oracle by {worst}"
    return h


def _check(name, scale, bs=None, k=192, ppr_device=None):
    cfg, n, ei, w, x, data, args, model, score, batch = _setup(name, scale, bs=bs, ppr_device=ppr_device)
    tb = torch.from_numpy(batch).to(DEV)
    h = _check_encoder_rows(model, data, args, x)
    assert torch.isfinite(h).all()
    logits = score.logits(model.pair_features(tb, h))
    assert torch.isfinite(logits).all()
    # (i) oracle on a sample of the same pairs: index sets bit-exact, logits within 1e-4
    sample, ref = _oracle_sample(model, score, data, args, batch, h, k)
    sel = model.compute_node_mask(torch.from_numpy(sample))
    for tag, info in zip(("cn", "onehop", "non1hop"), sel):
        if info is None:
            assert tag not in ref["sel"]
            continue
        np.testing.assert_array_equal(info[0].cpu().numpy(), ref["sel"][tag][0])
        np.testing.assert_array_equal(info[1].cpu().numpy().view(np.uint32), ref["sel"][tag][1].view(np.uint32))
    got = score.logits(model.pair_features(torch.from_numpy(sample).to(DEV), h)).cpu().numpy()
    assert np.abs(got - ref["logit"]).max() <= 1e-4
    # (ii) properties at the full batch size: sub-batch consistency, endpoint swap, permutation
    assert (logits[:k].cpu().numpy() == got).all()  # scoring a prefix alone gives bitwise the same scores
    swapped = score.logits(model.pair_features(tb.flip(0).contiguous(), h))
    assert (logits - swapped).abs().max().item() <= 5e-5
    perm = torch.randperm(tb.shape[1], device=DEV)
    permuted = score.logits(model.pair_features(tb[:, perm].contiguous(), h))
    assert (logits[perm] - permuted).abs().max().item() <= 1e-5
    return cfg, model


def test_collab_full_size():
    _check("collab", 1.0)


def test_ddi_like_dense_neighbourhoods():
    """N=4267, ~500 neighbours per node, mode "1-hop", D=256: rows longer than the LDS staging caps."""
    _check("ddi", 1.0, k=96)


def test_cora_like_no_layernorm():
    """D=256, L=1, --no-layer-norm --no-relu, F=1433 (not a multiple of 4), eps 1e-4."""
    _check("cora", 1.0, bs=4096)


def test_citation2_like_quarter_size():
    """N=732k (1/4 of ogbl-citation2), residual GCN, D=64, theta=(0,1e-3,1e-2), eps 2.5e-3."""
    _check("citation2", 0.25)


def test_ppa_like_eighth_size():
    """N=72k, mean degree ~74 (1/8 of ogbl-ppa's nodes and edges), F=58 (not a multiple of 4), D=64, residual."""
    _check("ppa", 0.125)


def test_citation2_like_full_size():
    """BASELINE config 4 at its stated size: N=2,927,963, 30.4 M undirected edges, D=64, L=3 residual, full-graph
    encoder resident in HBM; PPR (eps 2.5e-3) from the device producer.  Oracle on a pair sample + an encoder row
    sample, properties at the full 32,768-pair batch."""
    _check("citation2", 1.0, ppr_device=DEV)


def test_ppa_like_full_size():
    """BASELINE config 3's per-GPU share at its stated size: N=576,289, 21.2 M undirected edges (mean degree ~74,
    hub pairs cut into slices), F=58, D=64, residual, 32,768 pairs per GPU."""
    _check("ppa", 1.0, ppr_device=DEV)


def test_pyg_style_facade_matches_core():
    cfg, n, ei, w, x, data, args, model, score, batch = _setup("tiny")
    ppr = data["ppr"]
    m = lpformer_amd.LPFormer(cfg["f_in"], cfg["dim"], num_gnn_layers=cfg["gnn_layers"],
                              ppr_thresholds=list(cfg["thresholds"]), device=DEV).to(DEV).eval()
    m.core.load_state_dict(model.state_dict())
    m.score.load_state_dict(score.state_dict())
    # the facade takes an unweighted edge_index: compare with a core built on the same unweighted graph
    data2 = D.build_data(ei, x, n, ppr=ppr)
    core2 = lpformer_amd.LinkTransformer(args, data2, device=DEV).to(DEV).eval()
    core2.load_state_dict(model.state_dict())
    tb = torch.from_numpy(batch).to(DEV)
    want = score.logits(core2(tb))
    got = m(tb, torch.from_numpy(x).to(DEV), torch.from_numpy(ei).to(DEV), ppr)
    assert (want - got).abs().max().item() <= 1e-6
    sp = lpformer_amd.LPFormer.calc_sparse_ppr(torch.from_numpy(ei), n, 0.15, cfg["eps"])
    assert sp._nnz() == ppr.nnz
    # training mode: logits with an autograd graph (all dropouts 0 here: the same numbers), gradients = the core's
    tx, te = torch.from_numpy(x).to(DEV), torch.from_numpy(ei).to(DEV)
    for mod in (m, core2, score):
        mod.train()
    for mod in (m.core, core2):
        mod.att_drop = 0.0
        mod.node_encoder.feat_drop = 0.0
        mod.node_encoder.gnn_encoder.dropout = 0.0
        mod.att_layers[0].dropout = 0.0
        mod.elementwise_lin.dropout = mod.pairwise_lin.dropout = 0.0
    m.score.dropout = score.dropout = 0.0
    lg = m(tb, tx, te, ppr)
    assert lg.requires_grad and (lg.detach() - got).abs().max().item() <= 1e-4
    torch.nn.functional.binary_cross_entropy_with_logits(lg, torch.ones_like(lg)).backward()
    from lpformer_amd import train as lpf_train
    ref = lpf_train.score_train(score, core2(tb), logits=True)
    torch.nn.functional.binary_cross_entropy_with_logits(ref, torch.ones_like(ref)).backward()
    for (k, p), (_, q) in zip(m.core.named_parameters(), core2.named_parameters()):
        if q.grad is not None:
            scale = max(float(q.grad.abs().max()), 1e-6)
            assert p.grad is not None and float((p.grad - q.grad).abs().max()) / scale <= 1e-4, k


def test_batches_pipelined_on_two_streams_match_serial():
    """Consecutive batches issued on alternating HIP streams (per-stream workspaces, side streams for the
    elementwise / q branches) give bit-identical scores to the strictly serial single-stream run."""
    cfg, n, ei, w, x, data, args, model, score, _ = _setup("cora", bs=2048)
    h = model.propagate()
    batches = [torch.from_numpy(D.sample_pairs(ei, n, 2048 + 100 * i, seed=50 + i)).to(DEV) for i in range(6)]
    model.use_side_stream = False
    serial = [score.logits(model.pair_features(b, h)).clone() for b in batches]
    torch.cuda.synchronize()
    model.use_side_stream = True
    lanes = [torch.cuda.Stream(DEV) for _ in range(2)]
    outs = []
    for rep in range(3):  # repeated so that workspaces are reused while the other lane is still running
        outs = []
        for i, b in enumerate(batches):
            with torch.cuda.stream(lanes[i % 2]):
                outs.append(score.logits(model.pair_features(b, h)))
    torch.cuda.synchronize()
    for a, b in zip(serial, outs):
        assert torch.equal(a, b)


def test_evaluation_sweep_matches_per_batch_loop():
    """lpformer_amd.evaluate.score_edges (encoder once, batches pipelined over streams, scores kept on the device)
    equals the reference-style loop score_func(model(edge)) batch by batch; HeaRT negatives keep their [P, K] shape."""
    from lpformer_amd import evaluate as E
    cfg, n, ei, w, x, data, args, model, score, _ = _setup("cora", bs=1024)
    rng = np.random.default_rng(4)
    edges = torch.from_numpy(rng.integers(0, n, size=(5000, 2)))            # the reference's [P, 2] split layout
    loop = torch.cat([score(model(edges[i:i + 1024].t())) for i in range(0, 5000, 1024)])
    sweep = E.score_edges(model, score, edges, batch_size=1024, streams=3)
    bad = torch.nonzero((loop - sweep).abs() > 1e-6).flatten().tolist()   # folded score head: re-associated Linears
    assert sweep.is_cuda and not bad, (bad[:16], sweep[bad[:16]].tolist())
    # a long sweep replays recorded steps (one PlannedScorer per stream) for its full batches: the same bits
    planned = E.score_edges(model, score, edges, batch_size=256, streams=3, plans=True)
    assert torch.equal(planned, E.score_edges(model, score, edges, batch_size=256, streams=3, plans=False))
    assert (planned - loop).abs().max().item() <= 1e-6
    neg = torch.from_numpy(rng.integers(0, n, size=(40, 25, 2)))
    sn = E.score_negatives(model, score, neg, batch_size=300)
    assert sn.shape == (40, 25)
    # (another split of the same pairs: a pair without selected nodes is scored by the short form of the head or, in a
    # workgroup it shares with pairs that have some, by the full one -- equal up to rounding, not bit for bit)
    other = E.score_edges(model, score, neg.reshape(-1, 2), batch_size=1000, streams=1)
    assert (sn.reshape(-1) - other).abs().max().item() <= 5e-7
    model.tail_skip_empty = False
    assert torch.equal(E.score_negatives(model, score, neg, batch_size=300).reshape(-1),
                       E.score_edges(model, score, neg.reshape(-1, 2), batch_size=1000, streams=1))
    model.tail_skip_empty = True
    m = E.ranking_metrics(sweep[:40], sn)
    assert 0.0 < m["MRR"] <= 1.0 and 0.0 <= E.hits_at_k(sweep[:40], sn, 20) <= 1.0


def test_folded_score_path_matches_modules_and_oracle():
    """model.score_pairs (three Linears around the module boundary folded into one) against
    score_func(model.pair_features(...)) and against the CPU oracle, for D = 64 (counts 4) and the 1-hop mode
    (counts 3: padded K)."""
    for name, bs in (("ppa", 3000), ("ddi", 1500), ("collab", 4000)):
        scale = {"ppa": 0.02, "ddi": 1.0, "collab": 0.05}[name]
        cfg, n, ei, w, x, data, args, model, score, batch = _setup(name, scale=scale, bs=bs)
        h = model.propagate()
        b = torch.from_numpy(batch).to(DEV)
        ref_p = score(model.pair_features(b, h))
        ref_l = score.logits(model.pair_features(b, h))
        assert (model.score_pairs(b, h, score) - ref_p).abs().max().item() <= 2e-6
        model.use_tail_chain = False  # per-layer chains instead of lpf_tail_chain_f32 (the only path for D = 256)
        assert (model.score_pairs(b, h, score) - ref_p).abs().max().item() <= 2e-6
        model.use_tail_chain = True
        got_l = model.score_pairs(b, h, score, logits=True)
        assert (got_l - ref_l).abs().max().item() <= 2e-5 * max(1.0, ref_l.abs().max().item())
        sample, ref = _oracle_sample(model, score, data, args, batch, h, k=160)
        assert np.abs(got_l[:160].cpu().numpy() - ref["logit"]).max() <= 1e-4 * max(1.0, float(np.abs(ref["logit"]).max()))


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_step_replays_from_a_captured_graph(mode):
    """A whole scoring step (selection without read-back, one-pass attention, merged tail, side-stream branches) is
    captured in a HIP graph once; replays on other batches give bitwise the scores of the eager path -- in fp32 and with
    the bf16 switches of the pair stage on."""
    cfg, n, ei, w, x, data, args, model, score, batch = _setup("collab", scale=0.1, bs=4096)
    model.precision = model.tail_precision = mode
    h = model.propagate()
    batches = [torch.from_numpy(D.sample_pairs(ei, n, 4096, seed=70 + i)).to(DEV) for i in range(4)]
    scorer = lpformer_amd.GraphedScorer(model, score, h, batches[0], logits=True)
    for b in batches + batches[:2]:
        got = scorer(b).clone()
        torch.cuda.synchronize()
        want = model.score_pairs(b, h, score, logits=True)
        assert torch.equal(got, want)
    assert model.check_selection(scorer.stream)
    # a scorer that adopts its input reads the caller's tensor in place: no copy when called with that very tensor,
    # an in-place refill is seen by the next replay, any other tensor is copied in as before
    own = batches[1].clone()
    adopted = lpformer_amd.GraphedScorer(model, score, h, own, logits=True, adopt_input=True)
    assert adopted.batch is own
    assert torch.equal(adopted(own), model.score_pairs(batches[1], h, score, logits=True))
    own.copy_(batches[2])
    assert torch.equal(adopted(own), model.score_pairs(batches[2], h, score, logits=True))
    assert torch.equal(adopted(batches[3]), model.score_pairs(batches[3], h, score, logits=True))
    assert torch.equal(own, batches[3]) and adopted.check()


def _eager_checked(model, b, h, score):
    """Eager logits under the path's contract: a batch that outgrows the lane's selection workspace (sized by the
    batches the lane saw before) comes back as NaN with the sticky bit raised; ``check_selection()`` sizes it again."""
    out = model.score_pairs(b, h, score, logits=True)
    if not model.check_selection():
        out = model.score_pairs(b, h, score, logits=True)
        assert model.check_selection()
    return out


@pytest.mark.parametrize("name,scale,mode", [("collab", 0.1, "f32"), ("collab", 0.1, "bf16"), ("ppa", 0.02, "f32"),
                                             ("cora", 1.0, "f32")])
def test_step_replays_from_a_recorded_plan(name, scale, mode):
    """lpformer_amd.PlannedScorer: the C-ABI launches of one step, recorded and replayed as plain launches -- bitwise the
    eager scores for other batches (ids pointer replaced in the recorded arguments), for a batch in another memory
    layout (copied into the plan's own ids), across a parameter update (re-recorded) and an overflow (flagged, NaN,
    re-recorded with a larger workspace); D = 128 (activation-pattern attention, rows), D = 64 (matrix-core attention,
    records), D = 256."""
    cfg, n, ei, w, x, data, args, model, score, batch = _setup(name, scale=scale, bs=4096)
    model.precision = model.tail_precision = mode
    h = model.propagate()
    batches = [torch.from_numpy(D.sample_pairs(ei, n, 4096, seed=90 + i)).to(DEV) for i in range(4)]
    # without the side stream the recording is launches only; with it (the default) the two hand-overs are in it too
    model.use_side_stream = False
    serial = lpformer_amd.PlannedScorer(model, score, h, batches[0], logits=True)
    assert all(c[1] is not None for c in serial._plan)
    model.use_side_stream = True
    plan = lpformer_amd.PlannedScorer(model, score, h, batches[0], logits=True)
    assert sum(1 for c in plan._plan if c[1] is None) == 2
    assert torch.equal(serial(batches[1]).clone(), plan(batches[1]))
    n_launch = sum(1 for c in plan._plan if c[1] is not None)
    assert 4 <= n_launch <= 8, [c[0] for c in plan._plan]
    for b in batches + batches[:2]:
        got = plan(b).clone()
        torch.cuda.synchronize()
        assert torch.equal(got, model.score_pairs(b, h, score, logits=True))
    assert plan.check()
    strided = torch.zeros(2, 8192, dtype=torch.int64, device=DEV)[:, ::2]          # another layout: copied in
    strided.copy_(batches[2])
    got = plan(strided).clone()
    torch.cuda.synchronize()
    assert torch.equal(got, model.score_pairs(batches[2], h, score, logits=True))
    caps = plan.captures
    with torch.no_grad():
        score.lins[1].bias.add_(0.25)
    want = model.score_pairs(batches[1], h, score, logits=True).clone()
    got = plan(batches[1]).clone()
    torch.cuda.synchronize()
    assert plan.captures == caps + 1 and torch.equal(got, want)
    if name == "collab":
        deg = np.diff(data["adj_mask"].rowptr)
        hubs = np.argsort(deg)[-64:]
        rng = np.random.default_rng(5)
        bd = torch.from_numpy(np.stack([rng.choice(hubs, 4096), rng.choice(hubs, 4096)])).to(DEV)
        sparse = torch.from_numpy(rng.integers(0, n, size=(2, 4096))).to(DEV)
        p2 = lpformer_amd.PlannedScorer(model, score, h, sparse, logits=True)
        bad = p2(bd)
        torch.cuda.synchronize()
        all_nan = bool(torch.isnan(bad).all().item())
        assert not p2.check() and all_nan
        good = p2(bd).clone()
        assert p2.check()
        assert torch.equal(good, _eager_checked(model, bd, h, score))


@pytest.mark.parametrize("name,scale", [("collab", 0.1), ("ppa", 0.02), ("cora", 1.0)])
def test_query_from_table_and_gemm_agree(name, scale):
    """q = lin_l(x_a) + lin_l(x_b) from the per-node table Y (gathered inside the elementwise branch's launch) and as one
    [BS, D] x [D, D] product on x_a + x_b per batch (``query_from``): the same scores to rounding, through the scoring
    path and through the module-by-module one; D = 128 (rows form), 64 (matrix-core attention), 256."""
    cfg, n, ei, w, x, data, args, model, score, batch = _setup(name, scale=scale, bs=3000)
    tb = torch.from_numpy(batch).to(DEV)
    h = model.propagate()
    outs = {}
    for mode in ("table", "gemm"):
        model.query_from = mode
        outs[mode] = (model.score_pairs(tb, h, score, logits=True).clone(), model.calc_pairwise(tb, h)[0].clone())
        assert model.check_selection()
    for a, b in zip(outs["table"], outs["gemm"]):
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, float(a.abs().max()))
    sample, ref = _oracle_sample(model, score, data, args, batch, h, k=96)
    assert np.abs(outs["gemm"][0][:96].cpu().numpy() - ref["logit"]).max() <= 1e-4


def test_selection_overflow_is_flagged_and_recovered():
    """A batch with many more selected entries than the workspace was sized for: the scores come back as NaN (never
    silently wrong), check_selection() reports it once, and the re-scored batch is right."""
    cfg, n, ei, w, x, data, args, model, score, _ = _setup("collab", scale=0.05, bs=2048)
    h = model.propagate()
    rng = np.random.default_rng(0)
    sparse = torch.from_numpy(rng.integers(0, n, size=(2, 2048))).to(DEV)            # random pairs: few entries
    deg = np.diff(data["adj_mask"].rowptr)
    hubs = np.argsort(deg)[-64:]
    dense = torch.from_numpy(np.stack([rng.choice(hubs, 2048), rng.choice(hubs, 2048)])).to(DEV)  # hub pairs
    model.score_pairs(sparse, h, score)                     # sizes the workspace for the sparse batch
    assert model.check_selection()
    bad = model.score_pairs(dense, h, score, logits=True)
    torch.cuda.synchronize()
    assert torch.isnan(bad).all()
    assert not model.check_selection()                      # reported once, workspace marked for re-sizing
    good = model.score_pairs(dense, h, score, logits=True)
    assert model.check_selection() and torch.isfinite(good).all()
    ref = score.logits(model.pair_features(dense, h))
    assert (good - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    # the module API (what the reference's loops call) recovers by itself: same overflow, finite features, and they
    # equal the ones computed on a workspace that was large enough from the start
    model2 = lpformer_amd.LinkTransformer(args, data, device=DEV).to(DEV).eval()
    model2.load_state_dict(model.state_dict())
    model2.calc_pairwise(sparse, h)                          # sizes the workspace for the sparse batch
    out, _ = model2.calc_pairwise(dense, h)                  # overflows inside, is run again
    want, _ = model.calc_pairwise(dense, h)
    assert torch.isfinite(out).all() and torch.equal(out, want)


def test_check_selection_orders_its_read_behind_a_busy_lane():
    """check_selection(lane) called from ANOTHER stream while the lane still has the overflowing batch queued behind
    a long kernel: the status read must wait for the lane (never "ok" for a batch that has not run yet), and the
    clear must not race with the kernel that raises the bit."""
    cfg, n, ei, w, x, data, args, model, score, _ = _setup("collab", scale=0.05, bs=2048)
    h = model.propagate()
    rng = np.random.default_rng(1)
    sparse, dense = _sparse_then_hub_edges(n, data, rng, 2048, 2048)
    sparse, dense = (torch.from_numpy(e.T.copy()).to(DEV) for e in (sparse, dense))
    lane = model.lanes(1)[0]
    big = torch.randn(8192, 8192, device=DEV)
    torch.cuda.synchronize()
    with torch.cuda.stream(lane):
        model.score_pairs(sparse, h, score)               # sizes the lane's workspace for the sparse batch
    assert model.check_selection(lane)
    for _ in range(3):
        with torch.cuda.stream(lane):
            for _ in range(6):
                big = (big @ big).mul_(1e-4)              # tens of ms queued in front of the selection
            bad = model.score_pairs(dense, h, score, logits=True)
        assert not lane.query()                           # the lane is still busy when the check is made ...
        assert not model.check_selection(lane)            # ... from the main stream: must report the overflow
        assert torch.isnan(bad).all()
        with torch.cuda.stream(lane):
            good = model.score_pairs(dense, h, score, logits=True)
        assert model.check_selection(lane) and torch.isfinite(good).all()
        with torch.cuda.stream(lane):
            model._ws.clear()
            model.score_pairs(sparse, h, score)
        assert model.check_selection(lane)


def _sparse_then_hub_edges(n, data, rng, n_sparse, n_dense, n_hubs=64):
    deg = np.diff(data["adj_mask"].rowptr)
    hubs = np.argsort(deg)[-n_hubs:]
    sparse = rng.integers(0, n, size=(n_sparse, 2))
    dense = np.stack([rng.choice(hubs, n_dense), rng.choice(hubs, n_dense)], axis=1)
    return sparse, dense


def test_sweep_with_late_hub_batches_is_finite_and_right():
    """score_edges / score_negatives size a lane's selection workspace from its FIRST batch; later hub-heavy batches
    overflow it.  The sweep must notice (sticky status per lane), score those batches again and return finite scores
    equal to the per-batch loop -- and the ranking metrics computed from them must be the ones of the loop
    (a NaN score would compare False in `neg >= pos` and silently improve every rank)."""
    from lpformer_amd import evaluate as E
    cfg, n, ei, w, x, data, args, model, score, _ = _setup("collab", scale=0.05, bs=1024)
    h = model.propagate()
    rng = np.random.default_rng(5)
    sparse, dense = _sparse_then_hub_edges(n, data, rng, 4096, 6144)
    edges = torch.from_numpy(np.concatenate([sparse, dense]))            # sparse batches first, hub batches after
    for streams in (1, 3):
        model._ws.clear()                                                 # fresh workspaces: sized from the sparse start
        sweep = E.score_edges(model, score, edges, batch_size=1024, h=h, streams=streams)
        assert torch.isfinite(sweep).all()
        loop = torch.cat([score(model.pair_features(edges[i:i + 1024].t().contiguous().to(DEV), h))
                          for i in range(0, edges.shape[0], 1024)])
        assert (sweep - loop).abs().max().item() <= 2e-6
        for lane in model.lanes(streams):
            assert model.check_selection(lane)                            # nothing left pending
    # the same through recorded steps: a plan's workspace is sized from the batch it was recorded with (a sparse one)
    planned = E.score_edges(model, score, edges, batch_size=512, h=h, streams=2, plans=True)
    assert torch.isfinite(planned).all() and (planned - loop).abs().max().item() <= 2e-6
    # HeaRT layout: positives with K negatives each, the hub-heavy negatives at the end of the flattened list
    model._ws.clear()
    pos = torch.from_numpy(sparse[:64])
    neg = torch.from_numpy(np.concatenate([sparse[64:64 + 64 * 20], dense[:64 * 44]]).reshape(64, 64, 2))
    sp = E.score_edges(model, score, pos, batch_size=64, h=h, streams=1)
    sn = E.score_negatives(model, score, neg, batch_size=512, h=h, streams=2)
    assert torch.isfinite(sn).all()
    ref_n = torch.cat([score(model.pair_features(neg.reshape(-1, 2)[i:i + 512].t().contiguous().to(DEV), h))
                       for i in range(0, 64 * 64, 512)]).view(64, 64)
    got, want = E.ranking_metrics(sp, sn), E.ranking_metrics(sp, ref_n)
    assert got == want or all(abs(got[k] - want[k]) <= 1e-6 for k in got)


def test_graphed_scorer_survives_cache_replacement_staleness_and_overflow():
    """The captured graph reads Z / Y / folded tables / workspaces through raw pointers: (a) scoring ANOTHER encoder
    output through the model (second scorer, eager calls) must not disturb replays of the first; (b) a parameter
    update is noticed and the graph captured again; (c) a batch that overflows the captured workspace is reported by
    check() and right after the re-capture."""
    cfg, n, ei, w, x, data, args, model, score, _ = _setup("collab", scale=0.05, bs=2048)
    rng = np.random.default_rng(9)
    sparse, dense = _sparse_then_hub_edges(n, data, rng, 2048, 2048)
    b0 = torch.from_numpy(sparse.T.copy()).to(DEV)
    b1 = torch.from_numpy(D.sample_pairs(ei, n, 2048, seed=3)).to(DEV)
    bd = torch.from_numpy(dense.T.copy()).to(DEV)
    h = model.propagate()
    s1 = lpformer_amd.GraphedScorer(model, score, h, b1, logits=True)
    want1 = model.score_pairs(b1, h, score, logits=True).clone()
    # (a) another encoder output goes through the model's caches (Z / Y / bf16 copy replaced, workspaces reused)
    x2 = data["x"].clone()
    data["x"] = x2 * 1.5
    model._x_cache = None
    h2 = model.propagate()
    s2 = lpformer_amd.GraphedScorer(model, score, h2, b1, logits=True)
    for _ in range(3):
        model.score_pairs(b0, h2, score)
        torch.empty(64 << 20, device=DEV).fill_(1.0)        # churn the allocator: freed blocks get reused
    torch.cuda.synchronize()
    assert torch.equal(s1(b1), want1)
    assert torch.equal(s2(b1), model.score_pairs(b1, h2, score, logits=True))
    # (b) parameter update -> re-capture, new scores
    caps = s1.captures
    with torch.no_grad():
        score.lins[1].bias.add_(0.25)
    got = s1(b1).clone()
    assert s1.captures == caps + 1
    assert (got - (want1 + 0.25)).abs().max().item() <= 1e-5
    # (c) overflow: workspace of s3 is sized from a sparse batch, the hub batch does not fit
    s3 = lpformer_amd.GraphedScorer(model, score, h, b0, logits=True)
    bad = s3(bd)
    all_nan = bool(torch.isnan(bad).all().item())    # (read before check(): a re-capture recycles the result tensor)
    assert not s3.check() and all_nan
    good = s3(bd).clone()
    assert s3.check() and torch.isfinite(good).all()
    assert torch.equal(good, _eager_checked(model, bd, h, score))


def test_pyg_facade_recovers_from_selection_overflow():
    cfg, n, ei, w, x, data, args, model, score, _ = _setup("collab", scale=0.05, bs=1024)
    rng = np.random.default_rng(2)
    sparse, dense = _sparse_then_hub_edges(n, data, rng, 1024, 1024)
    m = lpformer_amd.LPFormer(cfg["f_in"], cfg["dim"], num_gnn_layers=cfg["gnn_layers"],
                              ppr_thresholds=list(cfg["thresholds"]), device=DEV).to(DEV).eval()
    xt, et = torch.from_numpy(x).to(DEV), torch.from_numpy(ei).to(DEV)
    a = m(torch.from_numpy(sparse.T.copy()).to(DEV), xt, et, data["ppr"])
    b = m(torch.from_numpy(dense.T.copy()).to(DEV), xt, et, data["ppr"])       # overflows the workspace sized by `a`
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    m.core._ws.clear()
    assert torch.equal(b, m(torch.from_numpy(dense.T.copy()).to(DEV), xt, et, data["ppr"]))


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_encoder_matches_unsharded(world):
    """The row-sharded encoder of every rank, emulated on one GPU (ragged N: not a multiple of the world size): per
    layer each rank produces its own row block -- a fused layer from the stitched layer input, an unfused one by
    transforming its rows and aggregating the stitched transformed rows --, the blocks are stitched where the RCCL
    all-gather would assemble them; the result is bitwise the unsharded encoder output.  Residual GCN (the shard must
    add its own rows of the layer input) with hub rows in every block; ppa-like: F = 58, D = 64, so layer 0 takes the
    two-launch path and layers 1, 2 the fused one."""
    from lpformer_amd import dist as LD
    cfg, n, ei, w, x, data, args, model, score, batch = _setup("ppa", scale=0.03)
    want = model.propagate()
    a_hat = model._device_graph("prop", model._data_obj("adj", False))
    n_layers = args["gnn_layers"]
    spans = [LD.row_range(n, world, r) for r in range(world)]
    x_full = model._features()
    xs = [x_full[lo:hi] for lo, hi in spans]
    kinds = []
    for i in range(n_layers):
        kinds.append(model._fusable(i, xs[0].shape[1]))
        if kinds[-1]:
            if x_full is None:
                x_full = torch.cat(xs)                                           # <- all-gather of the layer input
            xs = [model._layer_fused(i, a_hat, x_full, lo, hi) for lo, hi in spans]
        else:
            t_full = torch.cat([model._layer_transform(i, xr) for xr in xs])    # <- all-gather of the transformed rows
            xs = [model._layer_aggregate(i, a_hat, t_full, lo, hi, xr) for (lo, hi), xr in zip(spans, xs)]
        x_full = None
    assert kinds == [False, True, True]
    got = torch.cat(xs)                                                          # <- all-gather of node embeddings
    assert torch.equal(got, want)
    # pairs are split by index, scores need no exchange: scoring the shards separately = scoring the batch
    tb = torch.from_numpy(batch).to(DEV)
    whole = model.score_pairs(tb, want, score, logits=True)
    parts = torch.cat([model.score_pairs(LD.shard_pairs(tb, world, r).contiguous(), want, score, logits=True)
                       for r in range(world)])
    assert (whole - parts).abs().max().item() <= 1e-5


@pytest.mark.parametrize("name,scale,bs", [("collab", 0.1, 4096), ("ppa", 0.02, 3000), ("citation2", 0.01, 3000)])
def test_bf16_throughput_mode(name, scale, bs):
    """precision = "bf16" (bf16 storage of the node table Z + bf16 matrix cores for Wfold h, fp32 accumulation and
    fp32 everything else): the selected index sets are untouched (selection never sees bf16) and the logits stay
    within the stated tolerance of the fp32 path: 5e-3 absolute (observed 3e-4 ... 1.2e-3 on logits of magnitude ~1;
    bf16 keeps 8 significant bits of Z and of Wfold, the accumulation and everything downstream are fp32)."""
    cfg, n, ei, w, x, data, args, model, score, batch = _setup(name, scale=scale, bs=bs)
    h = model.propagate()
    b = torch.from_numpy(batch).to(DEV)
    ref = model.score_pairs(b, h, score, logits=True).clone()
    sel32 = model.compute_node_mask(b)
    model.precision = "bf16"
    got = model.score_pairs(b, h, score, logits=True)
    sel16 = model.compute_node_mask(b)
    model.precision = "f32"
    for a, c in zip(sel32, sel16):
        if a is None:
            continue
        assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1]) and torch.equal(a[2], c[2])
    err = (got - ref).abs().max().item()
    tol = 5e-3
    print(f"bf16 vs f32 logits on {name}: max abs diff {err:.3e} (tolerance {tol:.3e})")
    assert torch.isfinite(got).all() and err <= tol
    assert err > 0.0  # it really is a different arithmetic


@pytest.mark.gpu
def test_spmm_and_gemm_bf16_entry_points():
    """lpf_gemm_f32_out_bf16 = the fp32 product rounded to nearest even once; lpf_spmm_csr_bf16 = the fp32 aggregation
    of a bf16 table (sums, epilogue and output fp32): compared with plain torch on the same bf16-rounded table."""
    from lpformer_amd import _lib
    from lpformer_amd._lib import ptr, check
    torch.manual_seed(3)
    m, k, n = 1000, 96, 128
    a = torch.randn(m, k, device=DEV)
    w = torch.randn(n, k, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    out16 = torch.empty(m, n, dtype=torch.bfloat16, device=DEV)
    check(_lib.hip().lpf_gemm_f32_out_bf16(m, n, k, ptr(a), k, ptr(w), k, None, None, 0, ptr(out16), n, 0, st), "gemm")
    ref = a.double() @ w.double().t()
    rel = ((out16.double() - ref).abs() / ref.abs().clamp_min(1.0)).max().item()
    assert rel <= 2.0 ** -8                       # one bf16 rounding of an fp32-accurate product
    assert (out16.float() - ref.float().to(torch.bfloat16).float()).ne(0).float().mean().item() < 0.01

    # aggregation: random CSR with a hub row, bias + LayerNorm + ReLU epilogue
    rows, d = 700, 128
    deg = torch.randint(0, 12, (rows,))
    deg[5] = 400
    rp = torch.zeros(rows + 1, dtype=torch.int64)
    rp[1:] = torch.cumsum(deg, 0)
    nnz = int(rp[-1])
    col = torch.randint(0, m, (nnz,), dtype=torch.int32)
    val = torch.rand(nnz)
    bias, g, b = torch.randn(d, device=DEV), torch.rand(d, device=DEV) + 0.5, torch.randn(d, device=DEV)
    rp_d, col_d, val_d = rp.to(DEV), col.to(DEV), val.to(DEV)
    long_rows = torch.nonzero(deg > 128).flatten().to(torch.int32).to(DEV)
    got = torch.empty(rows, d, device=DEV)
    check(_lib.hip().lpf_spmm_csr_bf16(rows, d, ptr(rp_d), ptr(col_d), ptr(val_d), ptr(out16), n, ptr(got), d,
                                       ptr(bias), ptr(g), ptr(b), None, 0, None, None, 1, ptr(long_rows),
                                       long_rows.numel(), st), "spmm")
    table = out16.double().cpu()
    acc = torch.zeros(rows, d, dtype=torch.float64)
    r_of = torch.repeat_interleave(torch.arange(rows), deg)
    acc.index_add_(0, r_of, table[col.long()] * val.double()[:, None])
    y = acc + bias.double().cpu()
    y = torch.nn.functional.layer_norm(y, (d,), g.double().cpu(), b.double().cpu(), 1e-5).clamp_min(0)
    assert (got.double().cpu() - y).abs().max().item() <= 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("name,scale,bs", [("collab", 0.1, 4096), ("ppa", 0.02, 3000), ("citation2", 0.01, 3000),
                                           ("ddi", 1.0, 1024), ("cora", 1.0, 2048)])
def test_bf16_encoder_mode(name, scale, bs):
    """encoder_precision = "bf16": the per-layer table X W^T is gathered from bf16 storage (GEMM, neighbour sums and
    epilogue fp32).  Stated tolerance: node embeddings (LayerNorm output, magnitude ~1) within 3e-2 absolute of the
    fp32 encoder, logits of the full bf16 mode (encoder + attention) within 5e-3 (observed <= 1.2e-3); the selected index sets do not
    depend on the encoder at all."""
    cfg, n, ei, w, x, data, args, model, score, batch = _setup(name, scale=scale, bs=bs)
    h32 = model.propagate()
    b = torch.from_numpy(batch).to(DEV)
    ref = model.score_pairs(b, h32, score, logits=True).clone()
    model.encoder_precision = "bf16"
    h16 = model.propagate()
    model.precision = "bf16"     # (D = 256: the bf16 node table under the activation-pattern kernel; the tail stays fp32)
    got = model.score_pairs(b, h16, score, logits=True)
    model.encoder_precision = model.precision = "f32"
    e_h = (h16 - h32).abs().max().item()
    e_l = (got - ref).abs().max().item()
    print(f"bf16 encoder on {name}: max |dX| {e_h:.3e}, max |dlogit| (full bf16 mode) {e_l:.3e}")
    assert torch.isfinite(h16).all() and 0.0 < e_h <= 3e-2
    assert torch.isfinite(got).all() and e_l <= 5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("name,scale,bs", [("collab", 0.1, 4096), ("ppa", 0.02, 3000), ("citation2", 0.01, 3000)])
def test_bf16_tail_mode(name, scale, bs):
    """tail_precision = "bf16": the two GEMMs of the dense tail on the bf16 matrix cores (bf16 weights, activations
    rounded to bf16 as they enter a GEMM, fp32 accumulate; merge, LayerNorms, dot and sigmoid fp32).  Stated tolerance:
    logits within 5e-3 absolute of the fp32 path (observed <= 1e-3 on logits of magnitude ~0.2: two chained K = 144 /
    256 products with 8-bit significands)."""
    cfg, n, ei, w, x, data, args, model, score, batch = _setup(name, scale=scale, bs=bs)
    h = model.propagate()
    b = torch.from_numpy(batch).to(DEV)
    ref = model.score_pairs(b, h, score, logits=True).clone()
    model.tail_precision = "bf16"
    got = model.score_pairs(b, h, score, logits=True)
    model.tail_precision = "f32"
    err = (got - ref).abs().max().item()
    print(f"bf16 tail on {name}: max |dlogit| {err:.3e}, logit range {ref.abs().max().item():.2f}")
    assert torch.isfinite(got).all() and 0.0 < err <= 5e-3


@pytest.mark.parametrize("dim,f_in", [(128, 128), (64, 64), (32, 32), (64, 40)])
def test_fused_gcn_layer_matches_two_launches(dim, f_in):
    """lpf_gcn_layer_fused_f32 (aggregate, then transform in registers) against lpf_gemm_f32 + lpf_spmm_csr_f32 on a
    heavy-tailed graph with hub rows (> 128 entries), isolated nodes and a row count that is no multiple of 16; whole
    encoder and a row block (the gather_once layout's last layer).  f_in != dim: layer 0 cannot fuse, the others do."""
    cfg = dict(D.CONFIGS["collab"], n=20011, edges=90000, f_in=f_in)
    n = cfg["n"]
    ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=2.1, seed=3, max_weight=cfg["max_weight"])
    keep = (ei[0] < n - 7) & (ei[1] < n - 7)           # the last seven nodes keep only their self loops
    ei, w = ei[:, keep], (None if w is None else w[keep])
    x = np.random.default_rng(4).standard_normal((n, f_in)).astype(np.float32)
    data = D.build_data(ei, x, n, edge_weight=w, eps=1e-3)
    args = dict(D.train_args_for(cfg), dim=dim)
    torch.manual_seed(5)
    model = lpformer_amd.LinkTransformer(args, data, device=DEV).to(DEV).eval()
    with torch.no_grad():   # biases / LayerNorm affine parameters away from their initial 0 / 1
        for p in model.node_encoder.parameters():
            if p.dim() == 1:
                p.add_(0.3 * torch.randn_like(p))
    a_hat = model._device_graph("prop", model._data_obj("adj", False))
    deg = (a_hat.rowptr[1:] - a_hat.rowptr[:-1])
    assert int(deg.max()) > 128 and int(deg.min()) == 1
    model.encoder_fused = True
    fused_layers = []
    h1 = model.propagate(_layers_out=fused_layers)
    model.encoder_fused = False
    plain_layers = []
    h0 = model.propagate(_layers_out=plain_layers)
    assert torch.isfinite(h1).all()
    for a, b in zip(fused_layers, plain_layers):
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))
    # a row block of the last layer, input = the plain path's input of that layer
    last = len(model.node_encoder.gnn_encoder.convs) - 1
    lo, hi = 5003, 17011
    model.encoder_fused = True
    blk = model._layer(last, a_hat, plain_layers[last], lo, hi)
    assert float((blk - h0[lo:hi]).abs().max()) <= 2e-5 * max(1.0, float(h0.abs().max()))
    # launch-to-launch determinism
    again = model._layer(last, a_hat, plain_layers[last], lo, hi)
    assert torch.equal(blk, again)


@pytest.mark.parametrize("name,scale", [("collab", 0.1), ("cora", 1.0)])
def test_query_table_follows_an_in_place_update_of_lin_l(name, scale):
    """The per-node query table Y = X W_l^T + b_l (``query_from = "table"``) is parameter-derived: an in-place change of
    ``att.lin_l`` with the SAME encoder output reused (a frozen encoder with a trained head, an in-place
    ``load_state_dict``) must rebuild it -- eager and through a recorded plan, D = 128 and D = 256 -- or the queries are
    silently those of the old weights."""
    cfg, n, ei, w, x, data, args, model, score, batch = _setup(name, scale=scale, bs=2048)
    tb = torch.from_numpy(batch).to(DEV)
    h = model.propagate()
    plan = lpformer_amd.PlannedScorer(model, score, h, tb, logits=True)
    before = model.score_pairs(tb, h, score, logits=True).clone()
    assert model.check_selection() and torch.equal(before, plan(tb))
    with torch.no_grad():
        lin_l = model.att_layers[0].att.lin_l
        lin_l.weight.add_(0.05 * torch.randn_like(lin_l.weight))
        lin_l.bias.add_(0.1)
    after = model.score_pairs(tb, h, score, logits=True).clone()
    assert model.check_selection()
    assert (after - before).abs().max().item() > 1e-3, "the update must move the scores"
    sample, ref = _oracle_sample(model, score, data, args, batch, h)
    k = sample.shape[1]
    assert np.abs(after[:k].cpu().numpy() - ref["logit"]).max() <= 1e-4 * max(1.0, float(np.abs(ref["logit"]).max()))
    got = plan(tb).clone()                      # re-records: a parameter's version changed
    torch.cuda.synchronize()
    assert torch.equal(got, after)



def test_models_the_pattern_table_covers_badly_keep_the_type_major_path():
    """``_patterns_pay``: above ``PT_EXACT_MAX`` flipped units per entry left for the exact path (open by default since
    round 6: the table form measured ahead almost everywhere) the hot path goes back to the type-major attention kernel
    behind lpf_select4 + lpf_select4_regions -- the same scores (within the bar of two kernel forms), checked on the same
    model by moving the threshold."""
    cfg, n, ei, w, x, data, args, model, score, batch = _setup("collab", scale=0.1, bs=3000)
    model.attention_impl = "flip"
    tb = torch.from_numpy(batch).to(DEV)
    h = model.propagate()
    assert model._patterns_pay() and model._uses_select4()
    a = model.score_pairs(tb, h, score, logits=True).clone()
    assert model.check_selection()
    model.PT_EXACT_MAX = -1.0
    model._pt_choice = None
    assert not model._patterns_pay() and not model._uses_select4()
    b = model.score_pairs(tb, h, score, logits=True).clone()
    assert model.check_selection()
    assert (a - b).abs().max().item() <= 2e-5 * max(1.0, float(a.abs().max()))
    raw, left = model._flip_stats()
    assert 0.0 <= left <= raw


@pytest.mark.parametrize("name,scale,dim", [("collab", 0.1, 128), ("cora", 1.0, 256), ("ppa", 0.02, 64), ("tiny", 1.0, 32)])
def test_rows_tail_with_and_without_the_order_matches_oracle(name, scale, dim):
    """The pair-major form at every width: the dense tail behind the attention's order (pairs with selected nodes first,
    workgroups of pairs without any take the short form of the score head) and without it -- logits within 2e-5 of each
    other and of the oracle's (stated bar 1e-4: five times the margin)."""
    cfg, n, ei, w, x, data, args, model, score, batch = _setup(name, scale=scale, bs=3000)
    if model.dim != dim:
        args = dict(args, dim=dim)
        torch.manual_seed(3)
        model = lpformer_amd.LinkTransformer(args, data, device=DEV).to(DEV).eval()
        score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, 2).to(DEV).eval()
    model.attention_impl = "flip"           # the rows form at every D
    tb = torch.from_numpy(batch).to(DEV)
    h = model.propagate()
    outs = {}
    for skip in (True, False):
        model.tail_skip_empty = skip
        outs[skip] = model.score_pairs(tb, h, score, logits=True).clone()
        assert model.check_selection()
    scale_l = max(1.0, float(outs[True].abs().max()))
    assert (outs[True] - outs[False]).abs().max().item() <= 2e-5 * scale_l
    sample, ref = _oracle_sample(model, score, data, args, batch, h)
    k = sample.shape[1]
    for skip in (True, False):
        err = np.abs(outs[skip][:k].cpu().numpy() - ref["logit"]).max()
        assert err <= 2e-5 * max(1.0, float(np.abs(ref["logit"]).max())), err
