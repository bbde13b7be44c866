"""Seeded random sweep: HIP path vs the CPU oracle over model widths, mask modes ("all", "1-hop", "cn"), thresholds, hub-heavy graphs
(degrees above the 512-candidate item size, PPR rows above the 256-entry LDS cap), weighted / residual encoders and
batches with a == b, duplicate and isolated pairs.  Selection must be bit-exact through both selection kernels."""
import numpy as np
import pytest
import torch

import lpformer_amd
from lpformer_amd import data as D
from oracle import lpformer_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4

HUB_SEEDS = (1, 7)
CASES = [
    # seed, n, undirected edges, gamma, dim, layers, residual, thresholds (cn, 1hop, >1hop), eps, weighted
    (0, 400, 1500, 2.5, 32, 1, False, (0.0, 1e-3, 1e-2), 1e-3, False),
    (1, 900, 9000, 2.05, 64, 2, True, (0.0, 1e-4, 1e-2), 1e-4, True),     # hubs with degree > 512
    (2, 1500, 6000, 2.2, 128, 3, False, (1e-3, 1e-3, 5e-3), 2e-4, True),
    (3, 600, 20000, 3.0, 256, 2, False, (0.0, 1e-2, 1.0), 1e-4, False),   # dense, "1-hop" mode (ddi-like)
    (4, 2500, 7000, 2.1, 64, 3, True, (0.0, 0.0, 1e-2), 1e-4, False),     # theta_1hop = 0: P1 is the whole matrix
    (5, 1200, 15000, 2.05, 128, 1, False, (0.0, 1e-5, 1e-3), 5e-5, True),  # long PPR rows, many >1-hop nodes
    (6, 300, 600, 2.5, 32, 2, True, (5e-2, 5e-2, 5e-2), 1e-3, False),     # high thresholds: mostly empty pairs
    (7, 800, 12000, 2.02, 256, 2, True, (0.0, 1e-4, 1e-2), 1e-4, True),
    # mask mode "cn" (thresh_1hop == thresh_non1hop == 1, replicate_heart.sh:7,10): common neighbours only, t = 1 round
    # trip, one count feature.  The reference crashes here on torch >= 2.1: HIP vs the oracle's restatement only.
    (8, 700, 9000, 2.3, 64, 2, False, (0.0, 1.0, 1.0), 1e-4, False),
    (9, 500, 12000, 2.6, 128, 1, False, (2e-3, 1.0, 1.0), 1e-4, True),
    (10, 400, 8000, 3.0, 256, 1, False, (0.0, 1.0, 1.0), 1e-3, False),
]


@pytest.mark.parametrize("case", CASES, ids=[f"seed{c[0]}_d{c[4]}" for c in CASES])
def test_random_config_matches_oracle(case):
    seed, n, edges, gamma, dim, layers, residual, th, eps, weighted = case
    rng = np.random.default_rng(100 + seed)
    ei, w = D.chung_lu_graph(n, edges, gamma=gamma, seed=seed, max_weight=6 if weighted else 0)
    if seed in HUB_SEEDS:  # two explicit hubs: their pair is cut into slices of N(a) and of N(b)
        star = np.concatenate([np.stack([np.zeros(640, np.int64), rng.choice(np.arange(2, n), 640, replace=False)]),
                               np.stack([np.ones(560, np.int64), rng.choice(np.arange(2, n), 560, replace=False)])], 1)
        allp = np.concatenate([ei, star, star[::-1]], axis=1)
        allw = None if w is None else np.concatenate([w, np.ones(2 * star.shape[1], np.float32)])
        _, keep = np.unique(allp[0] * n + allp[1], return_index=True)
        ei, w = allp[:, keep], (None if allw is None else allw[keep])
    x = rng.standard_normal((n, 40)).astype(np.float32)
    ppr = lpformer_amd.calc_ppr(ei, n, 0.15, eps)
    d = D.build_data(ei, x, n, edge_weight=w, ppr=ppr)
    cfg = D.train_args_for(dict(thresholds=th, dim=dim, gnn_layers=layers, residual=residual))
    torch.manual_seed(seed)
    model = lpformer_amd.LinkTransformer(cfg, d, device=DEV).to(DEV).eval()
    score = lpformer_amd.mlp_score(2 * dim, 2 * dim, 1, 2).to(DEV).eval()
    with torch.no_grad():
        for p in list(model.parameters()) + list(score.parameters()):
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    P = {f"model.{k}": v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    P.update({f"score.{k}": v.detach().cpu().numpy() for k, v in score.state_dict().items()})
    batch = D.sample_pairs(ei, n, 700, seed=seed + 50)
    deg = np.bincount(ei[0], minlength=n)
    hub = int(np.argmax(deg))
    batch[:, :6] = np.array([[0, 5, 7, 7, hub, hub], [0, 5, 9, 9, (hub + 1) % n, hub]])  # a == b, duplicates, hub pairs
    batch[:, 7:9] = np.array([[0, 1], [1, 0]])
    iso = np.flatnonzero(deg == 0)
    if iso.size >= 2:
        batch[:, 6] = iso[:2]
    ref = O.forward(batch, x, O.gcn_norm(ei, w, n), O.symmetric_mask_csr(ei, n),
                    (ppr.rowptr, ppr.col.astype(np.int64), ppr.val), P, dict(cfg, pred_layers=2))
    tags = {"all": ("cn", "onehop", "non1hop"), "1-hop": ("cn", "onehop"), "cn": ("cn",)}[model.mask]
    assert all(i is None for i in model.compute_node_mask(torch.from_numpy(batch))[len(tags):])
    assert sum(ref["sel"][t][0].shape[1] for t in tags) > 100
    for indexed in (True, False):   # walk indexes (select3.hip) and the general path over the raw PPR rows (select2.hip)
        model.use_select_index = indexed
        infos = model.compute_node_mask(torch.from_numpy(batch))
        for tag, info in zip(tags, infos):
            np.testing.assert_array_equal(info[0].cpu().numpy(), ref["sel"][tag][0])
            np.testing.assert_array_equal(info[1].cpu().numpy().view(np.uint32), ref["sel"][tag][1].view(np.uint32))
            np.testing.assert_array_equal(info[2].cpu().numpy().view(np.uint32), ref["sel"][tag][2].view(np.uint32))
        for impl in ("flip", "mfma"):   # activation-pattern kernel and matrix-core kernel: same records, same features
            model.attention_impl = impl
            feats = model(torch.from_numpy(batch))
            scale = max(1.0, float(np.abs(ref["combined_feats"]).max()))
            assert np.abs(feats.cpu().numpy() - ref["combined_feats"]).max() <= TOL * scale, impl
            assert np.abs(score.logits(feats).cpu().numpy() - ref["logit"]).max() <= \
                TOL * max(1.0, float(np.abs(ref["logit"]).max())), impl
    assert deg.max() > 512 or seed not in HUB_SEEDS, "hub cases must exercise sliced pairs"


@pytest.mark.parametrize("dim,gain", [(32, 40.0), (64, 25.0), (128, 60.0), (256, 15.0)])
def test_flip_attention_with_weights_that_flip_most_units(dim, gain):
    """The activation-pattern kernel is exact for every input, only its cost depends on the weights: first PE layers
    scaled up (and shifted LayerNorm offsets) until a large share of the hidden units leaves the pattern of (0, 0) on a
    typical entry -- the correction path carries most of the product -- still match the oracle and the matrix-core
    kernel."""
    seed = 21
    rng = np.random.default_rng(seed)
    n = 500
    ei, w = D.chung_lu_graph(n, 6000, gamma=2.3, seed=seed, max_weight=0)
    x = rng.standard_normal((n, 24)).astype(np.float32)
    ppr = lpformer_amd.calc_ppr(ei, n, 0.15, 1e-4)
    d = D.build_data(ei, x, n, edge_weight=w, ppr=ppr)
    cfg = D.train_args_for(dict(thresholds=(0.0, 1e-3, 1e-2), dim=dim, gnn_layers=1, residual=False))
    torch.manual_seed(seed)
    model = lpformer_amd.LinkTransformer(cfg, d, device=DEV).to(DEV).eval()
    score = lpformer_amd.mlp_score(2 * dim, 2 * dim, 1, 2).to(DEV).eval()
    with torch.no_grad():
        for enc in (model.ppr_encoder_cn, model.ppr_encoder_onehop, model.ppr_encoder_non1hop):
            enc.linears[0].weight.mul_(gain)
            enc.norm.bias.add_(0.3 * torch.randn_like(enc.norm.bias))
    P = {f"model.{k}": v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    P.update({f"score.{k}": v.detach().cpu().numpy() for k, v in score.state_dict().items()})
    batch = D.sample_pairs(ei, n, 600, seed=seed + 1)
    ref = O.forward(batch, x, O.gcn_norm(ei, w, n), O.symmetric_mask_csr(ei, n),
                    (ppr.rowptr, ppr.col.astype(np.int64), ppr.val), P, dict(cfg, pred_layers=2))
    # how many units flip?  (restated from lpformer_amd.fold.flip_tables: z < 0 <=> the unit left the pattern of (0, 0))
    from lpformer_amd import fold
    tabs, _, _, _ = fold.flip_tables({k: v for k, v in model.state_dict().items()}, dim, 3)
    _, stat = fold.pe_tables({k: v for k, v in model.state_dict().items()}, dim, 3)
    pa, pb = ref["sel"]["onehop"][1].astype(np.float64), ref["sel"]["onehop"][2].astype(np.float64)
    var = stat[1, 0] * pa * pa + stat[1, 1] * pb * pb + stat[1, 2] + 2 * (stat[1, 3] * pa * pb + stat[1, 4] * pa + stat[1, 5] * pb)
    z = (tabs[1, :, 0][None] * pa[:, None] + tabs[1, :, 1][None] * pb[:, None] + tabs[1, :, 2][None]) \
        / np.sqrt(var + 1e-5)[:, None] + tabs[1, :, 3][None]
    assert (z < 0).mean() > 0.05, "the case must flip a sizeable share of the units"
    outs = {}
    for impl in ("flip", "mfma"):
        model.attention_impl = impl
        feats = model(torch.from_numpy(batch))
        outs[impl] = feats
        scale = max(1.0, float(np.abs(ref["combined_feats"]).max()))
        assert np.abs(feats.cpu().numpy() - ref["combined_feats"]).max() <= TOL * scale, impl
        lg = model.score_pairs(torch.from_numpy(batch).to(DEV), model.propagate(), score, logits=True)
        assert model.check_selection()
        assert np.abs(lg.cpu().numpy() - ref["logit"]).max() <= TOL * max(1.0, float(np.abs(ref["logit"]).max())), impl
    assert (outs["flip"] - outs["mfma"]).abs().max().item() <= 1e-5 * max(1.0, outs["mfma"].abs().max().item())


@pytest.mark.parametrize("dim", [128, 256])
def test_auto_attention_choice_follows_the_weights(dim):
    """``attention_impl = "auto"`` estimates the flipped units per entry on a sample of the model's own PPR matrix
    (LinkTransformer.flips_per_entry) and takes the activation-pattern kernel below the measured break-even, the
    matrix-core kernel above it.  Random-init weights: few flips -> "flip"; first PE layers scaled up: many -> "mfma";
    the estimate follows a parameter update without being asked; both choices within 1e-4 of the oracle."""
    seed = 33
    rng = np.random.default_rng(seed)
    n = 500
    ei, w = D.chung_lu_graph(n, 6000, gamma=2.3, seed=seed, max_weight=0)
    x = rng.standard_normal((n, 24)).astype(np.float32)
    ppr = lpformer_amd.calc_ppr(ei, n, 0.15, 1e-4)
    d = D.build_data(ei, x, n, edge_weight=w, ppr=ppr)
    cfg = D.train_args_for(dict(thresholds=(0.0, 1e-3, 1e-2), dim=dim, gnn_layers=1, residual=False))
    torch.manual_seed(seed)
    model = lpformer_amd.LinkTransformer(cfg, d, device=DEV).to(DEV).eval()
    score = lpformer_amd.mlp_score(2 * dim, 2 * dim, 1, 2).to(DEV).eval()
    batch = D.sample_pairs(ei, n, 600, seed=seed + 1)
    assert model.attention_impl == "auto"
    model.flip_recheck_every = 1      # look at the weights at every parameter version (the default amortises: below)
    chosen = []
    for gain in (1.0, 200.0):
        with torch.no_grad():
            for enc in (model.ppr_encoder_cn, model.ppr_encoder_onehop, model.ppr_encoder_non1hop):
                enc.linears[0].weight.mul_(gain)
        P = {f"model.{k}": v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        P.update({f"score.{k}": v.detach().cpu().numpy() for k, v in score.state_dict().items()})
        ref = O.forward(batch, x, O.gcn_norm(ei, w, n), O.symmetric_mask_csr(ei, n),
                        (ppr.rowptr, ppr.col.astype(np.int64), ppr.val), P, dict(cfg, pred_layers=2))
        flips = model.flips_per_entry()
        chosen.append((model.attention_kernel(), flips))
        assert (flips <= model.flip_break_even()) == (chosen[-1][0] == "flip")
        lg = model.score_pairs(torch.from_numpy(batch).to(DEV), model.propagate(), score, logits=True)
        assert model.check_selection()
        assert np.abs(lg.cpu().numpy() - ref["logit"]).max() <= TOL * max(1.0, float(np.abs(ref["logit"]).max()))
    assert [c[0] for c in chosen] == ["flip", "mfma"], chosen
    assert chosen[0][1] < model.flip_break_even() < chosen[1][1], chosen
    # a loop that alternates optimiser steps with scoring does not pay the estimate (a selection + host reads) per step:
    # with the default spacing the last choice stands until enough parameter versions have gone by, and training never asks
    model.flip_recheck_every = 16
    with torch.no_grad():
        model.ppr_encoder_onehop.linears[0].weight.mul_(1.0 / 200.0)
    est = model._flip_est
    assert model.attention_kernel() == "mfma" and model._flip_est is est       # (no new estimate yet)
    model.train()
    assert model.attention_kernel() == "mfma" and model._flip_est is est
    model.eval()


@pytest.mark.parametrize("case", [c for c in CASES if c[7][0] <= 0.0], ids=[f"seed{c[0]}_d{c[4]}" for c in CASES if c[7][0] <= 0.0])
def test_removed_edges_selection_is_the_selection_of_the_masked_graph(case):
    """The training loop's masked typing adjacency (src/train/train_model.py:38-46; link_transformer.py:229-250,438-443)
    on random graphs, through all three forms of the override -- the resident-index selection patched over the entries
    that touch a removed edge (the override as a coalesced tensor: the difference found on the device; and
    ``RemovedEdges``: the difference named) against the general kernels over a CSR built from the override --
    bit-exact, and against the ORACLE's selection of the masked graph.  Batches: positives whose edges are removed
    (hub endpoints repeat), their reverses, duplicates, non-edges."""
    seed, n, edges, gamma, dim, layers, residual, th, eps, weighted = case
    rng = np.random.default_rng(300 + seed)
    ei, w = D.chung_lu_graph(n, edges, gamma=gamma, seed=seed, max_weight=6 if weighted else 0)
    x = rng.standard_normal((n, 40)).astype(np.float32)
    ppr = lpformer_amd.calc_ppr(ei, n, 0.15, eps)
    d = D.build_data(ei, x, n, edge_weight=w, ppr=ppr)
    cfg = D.train_args_for(dict(thresholds=th, dim=dim, gnn_layers=layers, residual=residual))
    torch.manual_seed(seed)
    model = lpformer_amd.LinkTransformer(cfg, d, device=DEV).to(DEV).eval()
    und = ei[:, ei[0] < ei[1]]
    hub = np.argsort(np.bincount(ei[0], minlength=n))[-4:]
    at_hub = und[:, np.isin(und[0], hub) | np.isin(und[1], hub)]
    pos = np.concatenate([und[:, rng.integers(0, und.shape[1], 300)], at_hub[:, rng.integers(0, at_hub.shape[1], 100)]], axis=1)
    batch = np.concatenate([pos, pos[::-1, :40], pos[:, :25], rng.integers(0, n, (2, 60))], axis=1).astype(np.int64)
    gone = set((pos[0] * n + pos[1]).tolist()) | set((pos[1] * n + pos[0]).tolist())
    keep = np.array([k not in gone for k in (ei[0] * n + ei[1]).tolist()])
    kr, kc = torch.from_numpy(ei[0][keep]).to(DEV), torch.from_numpy(ei[1][keep]).to(DEV)
    masked_t = torch.sparse_coo_tensor(torch.stack([kr, kc]), torch.ones(kr.numel(), dtype=torch.int32, device=DEV),
                                       (n, n)).coalesce()
    tb = torch.from_numpy(batch)

    def select(ov, delta):
        model.use_mask_delta = delta
        model._delta_cache = None
        model._override.clear()
        out = [tuple(t.cpu().numpy() for t in info) for info in model.compute_node_mask(tb, False, ov) if info is not None]
        assert (model._delta_cache is not None) == delta
        return out
    general = select(masked_t, False)
    for ov in (masked_t, lpformer_amd.RemovedEdges(torch.from_numpy(pos))):
        got = select(ov, True)
        assert len(got) == len(general)
        for a, b in zip(got, general):
            np.testing.assert_array_equal(a[0], b[0])
            np.testing.assert_array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
            np.testing.assert_array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    # ... and the oracle on the masked graph (>1-hop exclusion from the UNMASKED adjacency)
    mask_full = O.symmetric_mask_csr(ei, n)
    masked = O.symmetric_mask_csr(ei[:, keep], n)
    want = O.select_nodes(batch, masked, (ppr.rowptr, ppr.col.astype(np.int64), ppr.val), th, n=n, adj_unmasked=mask_full)
    tags = ("cn", "onehop", "non1hop")[:len(general)]
    for tag, g in zip(tags, general):
        ix, a, b = want[tag] if tag in want else (np.zeros((2, 0), np.int64), np.zeros(0, np.float32), np.zeros(0, np.float32))
        np.testing.assert_array_equal(g[0], ix)
        np.testing.assert_array_equal(g[1].view(np.uint32), a.view(np.uint32))
        np.testing.assert_array_equal(g[2].view(np.uint32), b.view(np.uint32))
    plain = [tuple(t.cpu().numpy() for t in info) for info in model.compute_node_mask(tb, False, None) if info is not None]
    changed = any(p[0].shape != g[0].shape for p, g in zip(plain, general))
    assert changed, "the removed edges must matter"
