"""The training step: loss and the gradient of every parameter against fixtures recorded from the
reference's own training-step code (tests/golden/make_golden.py::build_train_case: model.train() with all dropout
probabilities 0, positives with their edges masked out of the typing / propagation adjacency, random negatives,
log-loss, loss.backward()).  Run on the GPU box: python -m pytest tests -m gpu."""
import json
import os

import numpy as np
import pytest
import torch

import lpformer_amd
from lpformer_amd import graph
from oracle.fixture_weights import make_state
from oracle.ref_shims import SparseTensor
from tests.golden_util import GOLDEN_DIR

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TRAIN_CASES = ["train_step_d64", "train_step_d64_residual", "train_step_d64_heads2", "train_step_d32_layers2"]
# (heads2: num_heads = 2; layers2: trans_layers = 2, the first layer 2 dim wide)


def _load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    cfg = json.loads(str(z["config_json"]))
    return z, cfg


def _build(z, cfg):
    n = cfg["n"]
    ei = z["edge_index"].astype(np.int64)
    adj_t = graph.csr_from_coo(ei[0], ei[1], z["edge_weight"], n)
    mask = graph.mask_csr(ei, n, symmetric=True)
    ppr = graph.csr_from_coo(z["ppr_row"], z["ppr_col"], z["ppr_val"], n)
    data = {"x": torch.from_numpy(z["x"]).to(DEV), "num_nodes": n, "adj_t": adj_t, "adj_mask": mask, "ppr": ppr,
            "full_adj_t": adj_t, "full_adj_mask": mask, "ppr_test": ppr}
    args = {k: cfg[k] for k in ("thresh_cn", "thresh_1hop", "thresh_non1hop", "dim", "trans_layers", "num_heads",
                                "att_drop", "dropout", "gnn_drop", "feat_drop", "gcn_cache", "gnn_layers",
                                "residual", "layer_norm", "relu")}
    model = lpformer_amd.LinkTransformer(args, data, device=DEV).to(DEV)
    score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, cfg["pred_layers"], 0.0).to(DEV)
    params = make_state(cfg["param_shapes"], cfg["seed"])
    have = set(model.state_dict())     # (mask modes "1-hop" / "cn" own fewer positional encoders and count columns)
    model.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in params.items()
                           if k.startswith("model.") and k[6:] in have
                           and tuple(v.shape) == tuple(model.state_dict()[k[6:]].shape)}, strict=False)
    score.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in params.items() if k.startswith("score.")})
    return model, score


def _step(model, score, z, cfg):
    n = cfg["n"]
    keep = torch.from_numpy(z["keep_edges"].astype(np.int64))
    masked = SparseTensor.from_edge_index(keep, sparse_sizes=(n, n)).to_symmetric()
    masked_adj = masked.to_torch_sparse_coo_tensor().coalesce().bool().int()
    masked_adjt = masked if cfg["mask_input"] else None
    edges = torch.from_numpy(z["pos_edges"]).to(DEV)
    neg = torch.from_numpy(z["neg_edges"]).to(DEV)
    h = model(edges, adj_prop=masked_adjt, adj_mask=masked_adj)            # train_model.py:59
    pos_out = score(h)
    pos_loss = -torch.log(pos_out + 1e-6).mean()
    hn = model(neg)                                                        # :66
    neg_out = score(hn)
    neg_loss = -torch.log(1 - neg_out + 1e-6).mean()
    loss = pos_loss + neg_loss
    loss.backward()
    return loss, pos_out, neg_out, h


@pytest.mark.parametrize("case", TRAIN_CASES)
def test_training_step_matches_reference_gradients(case):
    z, cfg = _load(case)
    model, score = _build(z, cfg)
    model.train()
    score.train()
    loss, pos_out, neg_out, h = _step(model, score, z, cfg)
    assert abs(loss.item() - float(z["loss"])) <= 1e-5
    assert np.abs(pos_out.detach().cpu().numpy() - z["pos_out"]).max() <= 1e-5
    assert np.abs(neg_out.detach().cpu().numpy() - z["neg_out"]).max() <= 1e-5
    assert np.abs(h.detach().cpu().numpy() - z["pos_feats"]).max() <= 1e-4
    worst = 0.0
    checked = 0
    for tag, mod in (("model", model), ("score", score)):
        for name, p in mod.named_parameters():
            key = f"grad.{tag}.{name}"
            if key not in z.files:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, name   # unused parameter
                continue
            want = z[key]
            got = p.grad.detach().cpu().numpy()
            scale = max(float(np.abs(want).max()), 1e-6)
            err = float(np.abs(got - want).max()) / scale
            worst = max(worst, err)
            checked += 1
            assert err <= 1e-4, f"{key}: relative error {err:.3e}"
    assert checked >= 50
    print(f"{case}: {checked} gradients, worst relative error {worst:.2e}")


def test_train_epoch_shaped_loop_runs_and_learns():
    """The reference's ``train_epoch`` body (src/train/train_model.py:22-81) with the drop-in modules, dropouts ON
    (feat / gnn / attention / pairwise drop / pred): a few optimiser steps with gradient clipping reduce the loss on a
    fixed set of edges; afterwards the same model evaluates through the inference kernels."""
    z, cfg = _load("train_step_d64")
    cfg = dict(cfg, att_drop=0.1, dropout=0.1, gnn_drop=0.1, feat_drop=0.1)
    model, score = _build(z, cfg)
    score.dropout = 0.1
    opt = torch.optim.Adam(list(model.parameters()) + list(score.parameters()), lr=5e-3)
    n = cfg["n"]
    ei = z["edge_index"].astype(np.int64)
    train_pos = torch.from_numpy(ei[:, ei[0] < ei[1]].T.copy()).to(DEV)
    torch.manual_seed(0)
    losses = []
    for epoch in range(6):
        model.train()
        score.train()
        perm = torch.randperm(train_pos.shape[0], device=DEV)[:128]
        adjmask = torch.ones(train_pos.shape[0], dtype=torch.bool, device=DEV)
        adjmask[perm] = False
        keep = train_pos[adjmask].t().cpu()
        masked = SparseTensor.from_edge_index(keep, sparse_sizes=(n, n)).to_symmetric()
        masked_adj = masked.to_torch_sparse_coo_tensor().coalesce().bool().int()
        edges = train_pos[perm].t()
        h = model(edges, adj_prop=masked, adj_mask=masked_adj)
        pos_loss = -torch.log(score(h) + 1e-6).mean()
        neg_edges = torch.randint(0, n, (2, edges.shape[1]), device=DEV)
        neg_loss = -torch.log(1 - score(model(neg_edges)) + 1e-6).mean()
        loss = pos_loss + neg_loss
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        torch.nn.utils.clip_grad_norm_(score.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and min(losses[-2:]) < losses[0]
    model.eval()
    score.eval()
    with torch.no_grad():
        p = score(model(train_pos[:64].t()))
    assert torch.isfinite(p).all() and p.shape == (64,)


def test_cn_mode_training_step_with_masked_adjacency():
    """Mask mode "cn" (thresh_1hop = thresh_non1hop = 1; the reference's HeaRT citeseer / pubmed runs,
    scripts/replicate_heart.sh:7,10 -- both TRAINING commands) through the training step: the batch's positive edges are
    masked out of the typing adjacency (train_model.py:40-46), so the selection runs on the general path
    (csrc/select2.hip, mode_cn).  The reference itself crashes in this mode on torch >= 2.1, so the check is
    against the oracle's restatement: train-mode features (dropouts 0) == oracle == eval-mode inference kernels; the
    gradient agrees with a central finite difference of the loss along two parameter directions; a short training loop
    with the dropouts on reduces the loss."""
    from oracle import lpformer_oracle as O
    z, cfg = _load("train_step_d64")
    cfg = dict(cfg, thresh_cn=0.0, thresh_1hop=1.0, thresh_non1hop=1.0)
    model, score = _build(z, cfg)
    assert model.mask == "cn" and model.count_dim == 1
    n = cfg["n"]
    ei = z["edge_index"].astype(np.int64)
    keep = torch.from_numpy(z["keep_edges"].astype(np.int64))
    masked = SparseTensor.from_edge_index(keep, sparse_sizes=(n, n)).to_symmetric()
    masked_adj = masked.to_torch_sparse_coo_tensor().coalesce().bool().int()
    edges = torch.from_numpy(z["pos_edges"]).to(DEV)
    neg = torch.from_numpy(z["neg_edges"]).to(DEV)

    def loss_of():
        h = model(edges, adj_mask=masked_adj)
        hn = model(neg)
        return (-torch.log(score(h) + 1e-6).mean() - torch.log(1 - score(hn) + 1e-6).mean()), h

    model.train(); score.train()
    loss, h_train = loss_of()
    loss.backward()
    # oracle: the same features with the masked typing adjacency (the >1-hop exclusion plays no role in this mode)
    P = {f"model.{k}": v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    P.update({f"score.{k}": v.detach().cpu().numpy() for k, v in score.state_dict().items()})
    kp = keep.numpy()
    both = np.concatenate([kp, kp[::-1]], axis=1)
    ppr = (np.asarray(model.data["ppr"].rowptr), np.asarray(model.data["ppr"].col).astype(np.int64),
           np.asarray(model.data["ppr"].val))
    ref = O.forward(z["pos_edges"], z["x"], O.gcn_norm(ei, z["edge_weight"], n), O.symmetric_mask_csr(both, n), ppr, P,
                    dict(cfg, pred_layers=cfg["pred_layers"]))
    assert set(ref["sel"]) == {"cn"} and ref["sel"]["cn"][0].shape[1] >= 5, ref["sel"]["cn"][0].shape
    scale = max(1.0, float(np.abs(ref["combined_feats"]).max()))
    assert np.abs(h_train.detach().cpu().numpy() - ref["combined_feats"]).max() <= 1e-4 * scale
    model.eval(); score.eval()
    with torch.no_grad():
        h_eval = model(edges, adj_mask=masked_adj)
    assert (h_eval - h_train.detach()).abs().max().item() <= 1e-4 * scale
    # finite differences along two parameter directions
    model.train(); score.train()
    for p in (model.ppr_encoder_cn.linears[1].weight, model.pairwise_lin.linears[0].weight):
        g = p.grad.detach().clone()
        d = torch.randn_like(p)
        d /= d.norm()
        eps = 2e-2
        with torch.no_grad():
            p.add_(eps * d)
            lp, _ = loss_of()
            p.sub_(2 * eps * d)
            lm, _ = loss_of()
            p.add_(eps * d)
        fd = (lp.item() - lm.item()) / (2 * eps)
        an = float((g * d).sum())
        assert abs(fd - an) <= 5e-2 * max(abs(an), 1e-3) + 2e-4, (fd, an)
    # a short loop with the dropouts on
    cfg2 = dict(cfg, att_drop=0.1, dropout=0.1, gnn_drop=0.1, feat_drop=0.1)
    model2, score2 = _build(z, cfg2)
    opt = torch.optim.Adam(list(model2.parameters()) + list(score2.parameters()), lr=5e-3)
    torch.manual_seed(0)
    losses = []
    for _ in range(6):
        model2.train(); score2.train()
        l2 = (-torch.log(score2(model2(edges, adj_mask=masked_adj)) + 1e-6).mean()
              - torch.log(1 - score2(model2(neg)) + 1e-6).mean())
        l2.backward()
        torch.nn.utils.clip_grad_norm_(model2.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
        losses.append(l2.item())
    assert all(np.isfinite(losses)) and min(losses[-2:]) < losses[0]


def test_overrides_taken_as_differences_train_like_graphs_of_their_own():
    """The training loop's per-batch overrides (src/train/train_model.py:38-56) as DIFFERENCES to the resident graphs --
    typing adjacency: selection over the walk indexes + the patch of lpformer_amd/mask_delta.py, also with the removed
    edges named explicitly (``lpformer_amd.RemovedEdges``); propagation matrix (same edges, resident weights): the
    resident structure with the removed edges' weights at 0, re-normalised, its transposed structure shared in the
    backward pass -- against the same step with both overrides turned into graphs of their own (rounds 2-5): loss and
    every gradient within 1e-5 relative."""
    z, cfg = _load("train_step_d64")
    n = cfg["n"]
    ei = z["edge_index"].astype(np.int64)
    w = z["edge_weight"].astype(np.float32)
    edges = torch.from_numpy(z["pos_edges"]).to(DEV)
    neg = torch.from_numpy(z["neg_edges"]).to(DEV)
    gone = set((z["pos_edges"][0] * n + z["pos_edges"][1]).tolist()) | set((z["pos_edges"][1] * n + z["pos_edges"][0]).tolist())
    km = np.array([k not in gone for k in (ei[0] * n + ei[1]).tolist()])
    assert 0 < km.sum() < km.size

    class _ST:
        def __init__(self, r, c, v):
            self._r, self._c, self._v = r, c, v

        def coo(self):
            return self._r, self._c, self._v

        def sparse_sizes(self):
            return (n, n)
    r, c = torch.from_numpy(ei[0][km]).to(DEV), torch.from_numpy(ei[1][km]).to(DEV)
    v = torch.from_numpy(w[km]).to(DEV)

    def run(delta, removed):
        model, score = _build(z, cfg)
        model.train(); score.train()
        model.use_mask_delta = delta
        ov_mask = lpformer_amd.RemovedEdges(edges) if removed else _ST(r, c, None)
        # ("both": the propagation matrix named as a difference too -- no tensor of the kept edges at all)
        ov_prop = lpformer_amd.RemovedEdges(edges) if removed == "both" else _ST(r, c, v)
        h = model(edges, adj_prop=ov_prop, adj_mask=ov_mask)
        loss = (-torch.log(score(h) + 1e-6).mean() - torch.log(1 - score(model(neg)) + 1e-6).mean())
        loss.backward()
        shared = model._override["prop"][1].rowptr is model._device_graph("prop", model.data["adj_t"]).rowptr
        grads = {k: p.grad.detach().clone() for k, p in list(model.named_parameters()) + list(score.named_parameters())
                 if p.grad is not None}
        return loss.item(), grads, shared, model._delta_cache

    l0, g0, shared0, dc0 = run(False, False)
    assert not shared0 and dc0 is None
    for removed in (False, True, "both"):
        l1, g1, shared1, dc1 = run(True, removed)
        assert shared1 and dc1 is not None and dc1[2] is not None and dc1[2].numel() > 0
        assert abs(l1 - l0) <= 1e-5 * max(1.0, abs(l0))
        assert set(g1) == set(g0)
        for k in g0:
            scale = max(float(g0[k].abs().max()), 1e-6)
            assert float((g1[k] - g0[k]).abs().max()) / scale <= 1e-5, k


@pytest.mark.parametrize("dim,drop_p,skip", [(128, 0.0, False), (64, 0.0, True), (32, 0.0, False), (128, 0.25, True),
                                             (64, 0.1, False), (32, 0.5, True)])
def test_fused_layer_backward_matches_fp64_autograd(dim, drop_p, skip):
    """``train.GcnFusedFn`` (round 6: forward writes the aggregated rows and applies dropout + the skip connection in the
    launch; backward = dropout/ReLU/LayerNorm backward, dW = du^T h, dx = ONE more launch of the fused layer over the
    transposed graph with the transposed weight image and the skip gradient as residual) against torch autograd in fp64
    of y = [x +] mask * ReLU(LN(A (x W^T) + b)) / (1 - p) -- on a DIRECTED weighted graph (A^T != A) with hub rows (> 64
    entries: the slice-sum path, both directions), isolated rows and a row count that is no multiple of 16.  The mask
    of the fp64 side is ``train.drop_keep_mask``, the torch restatement of the kernels' hash: an element the kernels
    and the restatement disagree on would show as an O(1) error."""
    from lpformer_amd import data as D
    from lpformer_amd import train
    n = 5003
    ei, _ = D.chung_lu_graph(n, 40000, gamma=2.1, seed=2)
    rng = np.random.default_rng(dim)
    keep = rng.random(ei.shape[1]) < 0.6                     # drop 40 % of the directed entries: no symmetry left
    r, c = ei[0][keep], ei[1][keep]
    v = (rng.random(r.size) + 0.1).astype(np.float32) / 8
    a = graph.csr_from_coo(r, c, v, n).to_device(DEV)
    at = train._transpose_csr(a)
    deg, deg_t = a.rowptr[1:] - a.rowptr[:-1], at.rowptr[1:] - at.rowptr[:-1]
    assert int(deg.max()) > 64 and int(deg_t.max()) > 64 and int(deg.min()) == 0 and int(deg_t.min()) == 0
    torch.manual_seed(dim)
    x = torch.randn(n, dim, device=DEV, requires_grad=True)
    w = (torch.randn(dim, dim, device=DEV) / dim ** 0.5).requires_grad_()
    b, g, be = (t.requires_grad_() for t in (0.1 * torch.randn(dim, device=DEV), 1 + 0.1 * torch.randn(dim, device=DEV),
                                             0.1 * torch.randn(dim, device=DEV)))
    up = torch.randn(n, dim, device=DEV)

    class _M:      # what GcnFusedFn reads of the model: the two weight-image caches
        from lpformer_amd.link_transformer import _PackedSquare
        _conv_packs, _conv_packs_t = [_PackedSquare()], [_PackedSquare()]
    torch.manual_seed(1000 + dim)
    seed = train._next_drop_seed(torch.device(DEV))
    torch.manual_seed(1000 + dim)                            # re-seeded: the launch below draws the same seed
    out = train.GcnFusedFn.apply(_M, 0, x, w, a, b, g, be, drop_p, skip)
    (out * up).sum().backward()
    got = [t.grad.clone() for t in (x, w, b, g, be)]
    # fp64 reference
    mask = train.drop_keep_mask(seed, drop_p, n, dim, DEV)
    if drop_p > 0:
        frac = float(mask.double().mean())
        assert abs(frac - (1 - drop_p)) <= 4 * (drop_p * (1 - drop_p) / mask.numel()) ** 0.5 + 1e-9
        # no row or column pattern: every row / column keeps about its share
        assert float((mask.double().mean(0) - (1 - drop_p)).abs().max()) <= 6 * (drop_p * (1 - drop_p) / n) ** 0.5
        assert float((mask.double().mean(1) - (1 - drop_p)).abs().max()) <= 6 * (drop_p * (1 - drop_p) / dim) ** 0.5
    else:
        assert bool(mask.all())
    dense = torch.zeros(n, n, dtype=torch.float64, device=DEV)
    rows = torch.repeat_interleave(torch.arange(n, device=DEV), deg)
    dense.index_put_((rows, a.col.long()), a.val.double(), accumulate=True)
    xs = [t.detach().double().requires_grad_() for t in (x, w, b, g, be)]
    ref = torch.relu(torch.nn.functional.layer_norm(dense @ (xs[0] @ xs[1].t()) + xs[2], (dim,), xs[3], xs[4]))
    ref = ref * mask.double() / (1 - drop_p)
    if skip:
        ref = ref + xs[0]
    assert float((out.detach().double() - ref.detach()).abs().max()) <= 2e-5 * max(1.0, float(ref.detach().abs().max()))
    (ref * up.double()).sum().backward()
    for name, gt, want in zip(("dx", "dW", "dbias", "dgamma", "dbeta"), got, (t.grad for t in xs)):
        err, scale = float((gt.double() - want).abs().max()), max(1.0, float(want.abs().max()))
        assert err <= 1e-4 * scale, (name, err, scale)
    # the first layer's input takes no gradient: no launch over the transposed graph, the other gradients unchanged
    x0 = x.detach()
    for t in (w, b, g, be):
        t.grad = None
    torch.manual_seed(1000 + dim)
    (train.GcnFusedFn.apply(_M, 0, x0, w, a, b, g, be, drop_p, skip) * up).sum().backward()
    assert torch.equal(w.grad, got[1]) and torch.equal(b.grad, got[2])
    # another draw is another mask
    if drop_p > 0:
        assert not torch.equal(train.drop_keep_mask(train._next_drop_seed(torch.device(DEV)), drop_p, n, dim, DEV), mask)


@pytest.mark.parametrize("dim", [32, 64, 128, 256])
def test_attention_stage_training_kernels_match_fp64_autograd(dim):
    """``train.PairAttentionFn`` (csrc/pair_train.hip: PE hidden layer, folded key projection, per-pair leaky-ReLU
    attention with PyG's segment softmax -- layers.py:193-224, link_transformer.py:182-211) forward and every gradient
    against torch autograd in fp64 of the same algebra, at all four lane groupings (D = 32 / 64 / 128 / 256), on ragged
    pairs: empty pairs, pairs of 1 .. 3 entries (less than one trip of four), hub pairs of 37 and 130 entries, three
    types with different sizes, nodes shared by many entries (the dZ atomics)."""
    from lpformer_amd import train
    torch.manual_seed(dim)
    rng = np.random.default_rng(dim)
    bs, n_nodes, n_t = 301, 500, 3
    counts = rng.integers(0, 7, (n_t, bs))
    counts[:, rng.random(bs) < 0.3] = 0                          # pairs without any entry
    counts[0, 5], counts[1, 17], counts[2, 200] = 37, 130, 41    # hub pairs
    seg = np.zeros((n_t, bs + 1), np.int64)
    base = 0
    tbase = [0]
    for t in range(n_t):
        seg[t, 0] = base
        seg[t, 1:] = base + np.cumsum(counts[t])
        base = int(seg[t, bs])
        tbase.append(base)
    n = base
    e_node = torch.from_numpy(rng.integers(0, n_nodes, n).astype(np.int32)).to(DEV)
    e_pa, e_pb = (torch.from_numpy(rng.random(n).astype(np.float32) * 0.05).to(DEV) for _ in range(2))
    mk = lambda *shape, s=1.0: (s * torch.randn(*shape, device=DEV)).requires_grad_()
    z, q = mk(n_nodes, dim), mk(bs, dim)
    att, bias = mk(dim, s=0.3), mk(dim, s=0.1)
    wfold, bfold = mk(n_t, dim, dim, s=dim ** -0.5), mk(n_t, dim, s=0.1)
    w1s, b1s = mk(n_t, dim, 2, s=3.0), mk(n_t, dim, s=0.2)
    gams, bets = (1 + 0.1 * torch.randn(n_t, dim, device=DEV)).requires_grad_(), mk(n_t, dim, s=0.1)
    params = (z, q, att, bias, wfold, bfold, w1s, b1s, gams, bets)
    up = torch.randn(bs, dim, device=DEV)
    out = train.PairAttentionFn.apply(*params, e_node, e_pa, e_pb, torch.from_numpy(seg.reshape(-1)).to(DEV), tbase)
    (out * up).sum().backward()
    got = [p.grad.clone() for p in params]
    # fp64 restatement
    P = [p.detach().double().requires_grad_() for p in params]
    z_, q_, att_, bias_, wf_, bf_, w1_, b1_, g_, be_ = P
    pair_of = np.zeros(n, np.int64)
    type_of = np.zeros(n, np.int64)
    for t in range(n_t):
        for p in range(bs):
            pair_of[seg[t, p]:seg[t, p + 1]] = p
        type_of[tbase[t]:tbase[t + 1]] = t
    pair_t, type_t = torch.from_numpy(pair_of).to(DEV), torch.from_numpy(type_of).to(DEV)
    pa, pb = e_pa.double(), e_pb.double()

    def hidden(x, y):
        u = x[:, None] * w1_[type_t][:, :, 0] + y[:, None] * w1_[type_t][:, :, 1] + b1_[type_t]
        mu, var = u.mean(1, keepdim=True), u.var(1, unbiased=False, keepdim=True)
        return torch.relu((u - mu) / torch.sqrt(var + 1e-5) * g_[type_t] + be_[type_t])
    h = hidden(pa, pb) + hidden(pb, pa)
    kp = torch.einsum("ed,eod->eo", h, wf_[type_t]) + bf_[type_t]
    k = z_[e_node.long()] + kp
    s = (torch.nn.functional.leaky_relu(k * q_[pair_t], 0.2) * att_).sum(1)
    smax = torch.full((bs,), -float("inf"), dtype=torch.float64, device=DEV).scatter_reduce(0, pair_t, s.detach(), "amax")
    w = torch.exp(s - smax[pair_t])
    den = torch.zeros(bs, dtype=torch.float64, device=DEV).index_add(0, pair_t, w) + 1e-16
    ref = torch.zeros(bs, dim, dtype=torch.float64, device=DEV).index_add(0, pair_t, (w / den[pair_t])[:, None] * k) + bias_
    err = float((out.detach().double() - ref.detach()).abs().max())
    assert err <= 2e-5 * max(1.0, float(ref.detach().abs().max())), err
    (ref * up.double()).sum().backward()
    names = ("dz", "dq", "datt", "dbias", "dwfold", "dbfold", "dw1", "db1", "dgamma", "dbeta")
    for name, g, want in zip(names, got, (p.grad for p in P)):
        e, scale = float((g.double() - want).abs().max()), max(1.0, float(want.abs().max()))
        assert e <= 2e-4 * scale, (name, e, scale)


def test_training_step_with_no_selected_node_at_all():
    """Thresholds nothing passes: every pair's attention output is the bias, the entry arrays are empty, the node table of
    the attention stage has no row to hold -- forward, backward and an optimiser step run, every gradient is finite, and
    the parameters the empty stage cannot reach (PE encoders, att, lin_r) receive zeros or None, not garbage."""
    from lpformer_amd import data as D
    n = 600
    ei, _ = D.chung_lu_graph(n, 2400, gamma=2.4, seed=4)
    x = np.random.default_rng(2).standard_normal((n, 64)).astype(np.float32)
    data = D.build_data(ei, x, n, eps=1e-3)
    cfg = D.train_args_for(dict(thresholds=(0.99, 0.99, 0.99), dim=64, gnn_layers=2, residual=True))
    cfg.update(att_drop=0.1, dropout=0.1, gnn_drop=0.1, feat_drop=0.1)
    torch.manual_seed(0)
    model = lpformer_amd.LinkTransformer(cfg, data, device=DEV).to(DEV).train()
    score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, 2, 0.1).to(DEV).train()
    opt = torch.optim.Adam(list(model.parameters()) + list(score.parameters()), lr=1e-3)
    batch = torch.from_numpy(D.sample_pairs(ei, n, 256, seed=9)).to(DEV)
    assert all(info is None or info[0].shape[1] == 0 for info in model.eval().compute_node_mask(batch))
    model.train()
    loss = -torch.log(score(model(batch)) + 1e-6).mean()
    loss.backward()
    assert torch.isfinite(loss)
    for name, p in list(model.named_parameters()) + list(score.named_parameters()):
        assert p.grad is None or torch.isfinite(p.grad).all(), name
    for name, p in model.named_parameters():
        if name.startswith("ppr_encoder") or name.startswith("att_layers.0.att.lin_r"):
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
    opt.step()


def test_a_seeded_training_step_gives_the_same_bits_twice():
    """Every reduction of the training step has a fixed order -- the weight-gradient partials, the LayerNorm / column-sum
    partials, the by-node sums of the attention stage's and the endpoint gathers' gradients (no float atomics since round
    6), the in-kernel dropout masks and torch's own generators under ``torch.manual_seed``: two runs of the same seeded
    step (all dropouts and the random attention drop ON, the batch's positives removed from the typing adjacency) give
    bitwise the same loss and the same gradient for every parameter."""
    z, cfg = _load("train_step_d64_residual")
    edges = torch.from_numpy(z["pos_edges"]).to(DEV)
    neg = torch.from_numpy(z["neg_edges"]).to(DEV)

    def run():
        model, score = _build(z, cfg)
        model.att_drop, model.node_encoder.feat_drop = 0.1, 0.1
        model.node_encoder.gnn_encoder.dropout = 0.1
        model.att_layers[0].dropout = 0.1
        model.elementwise_lin.dropout = model.pairwise_lin.dropout = score.dropout = 0.1
        model.train(); score.train()
        torch.manual_seed(123)
        loss = (-torch.log(score(model(edges, adj_mask=lpformer_amd.RemovedEdges(edges))) + 1e-6).mean()
                - torch.log(1 - score(model(neg)) + 1e-6).mean())
        loss.backward()
        return loss.detach().clone(), {k: p.grad.detach().clone() for k, p in
                                       list(model.named_parameters()) + list(score.named_parameters()) if p.grad is not None}
    l0, g0 = run()
    l1, g1 = run()
    assert torch.equal(l0, l1) and set(g0) == set(g1) and len(g0) > 40
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    assert float(l0) > 0 and any(float(v.abs().max()) > 0 for v in g0.values())
