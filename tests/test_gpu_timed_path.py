"""The launches ``bench.py`` times, pinned against the oracle at the bench's own sizes (GPU box).

``bench.py``'s step is ``model.score_pairs(batch, h, score)``: elementwise layer + q gather (side launch) -> selection ->
pair-major attention with the tail's order -> short-form dense tail, issued eagerly, from a recorded plan or from four
lanes.  ``tests/test_gpu_configs.py`` pins ``pair_features`` (the module-by-module form) at full size; here the step
itself is compared with ``oracle/lpformer_oracle.py`` (reference src/train/testing.py:105-117 ->
src/models/link_transformer.py:82-178, src/models/other_models.py:142-179) on a 2,048-pair sample of the full batch of
the full-size collab-like / ddi-like / cora-like problems, in every launch form; the 48 weight / threshold / width
configurations of ``tools/stress_patterns.py`` (activation-pattern table coverage from ~100 % down to a few per cent)
as a parametrised test; and one model after 150 steps of the repo's own training step (trained PE weights, table coverage
below 100 %).  Tolerance: logits within 1e-4 (relative to max(1, |logit|_max)) -- BASELINE.json's fp32 bound."""
import itertools

import numpy as np
import pytest
import torch

import lpformer_amd
from lpformer_amd import data as D
from oracle import lpformer_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


def _params(model, score):
    P = {f"model.{a}": v.detach().cpu().numpy() for a, v in model.state_dict().items()}
    P.update({f"score.{a}": v.detach().cpu().numpy() for a, v in score.state_dict().items()})
    return P


def _oracle_logits(model, score, data, args, pairs, h):
    mask, ppr = data["adj_mask"], data["ppr"]
    ref = O.forward(pairs, None, None, (mask.rowptr, mask.col.astype(np.int64)),
                    (ppr.rowptr, ppr.col.astype(np.int64), ppr.val), _params(model, score), dict(args, pred_layers=2),
                    x_node=h.cpu().numpy())
    return ref["logit"]


def _rel(got, want):
    return float(np.abs(got - want).max()) / max(1.0, float(np.abs(want).max()))


@pytest.mark.parametrize("name", ["collab", "ddi", "cora"])
def test_bench_step_at_full_size_matches_oracle_in_every_launch_form(name):
    """Full-size problem exactly as ``bench.py`` builds it (graph seed 0, features seed 1, PPR from the device
    producer, the config's own batch size); ``score_pairs(..., logits=True)`` on the full batch -- eager, from a
    recorded plan (``PlannedScorer``), from a captured HIP graph and from four lanes -- against the oracle on 2,048 of
    its pairs; the forms bitwise equal to each other."""
    cfg = dict(D.CONFIGS[name])
    n, bs, k = cfg["n"], cfg["batch"], 2048
    ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
    x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
    data = D.build_data(ei, x, n, edge_weight=w, eps=cfg["eps"], ppr_device=DEV)
    args = D.train_args_for(cfg)
    torch.manual_seed(0)
    model = lpformer_amd.LinkTransformer(args, data, device=DEV).to(DEV).eval()
    score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, 2).to(DEV).eval()
    batches = [torch.from_numpy(D.sample_pairs(ei, n, bs, seed=100 + i)).to(DEV) for i in range(4)]
    h = model.propagate()
    assert model._uses_rows(), "the bench's step at D >= 128 is the pair-major form"
    # eager
    for _attempt in range(3):
        eager = [model.score_pairs(b, h, score, logits=True).clone() for b in batches]
        if model.check_selection():
            break
    else:
        raise AssertionError("the selection workspace could not be sized")
    rng = np.random.default_rng(5)
    pick = np.sort(rng.choice(bs, size=k, replace=False))
    want = _oracle_logits(model, score, data, args, batches[0][:, torch.from_numpy(pick).to(DEV)].cpu().numpy(), h)
    got = eager[0].cpu().numpy()[pick]
    assert np.isfinite(got).all()
    assert _rel(got, want) <= TOL, f"{name}: eager step differs from the oracle by {_rel(got, want):.2e}"
    # the same pairs scored ALONE (a 2,048-pair batch: other ranges, other cuts): still the oracle's logits
    for _attempt in range(3):
        alone = model.score_pairs(batches[0][:, torch.from_numpy(pick).to(DEV)].contiguous(), h, score, logits=True)
        if model.check_selection():
            break
    assert _rel(alone.cpu().numpy(), want) <= TOL
    # recorded plan and captured graph: bitwise the eager step, for the recorded batch and for another one
    plan = lpformer_amd.PlannedScorer(model, score, h, batches[0], logits=True)
    graph = lpformer_amd.GraphedScorer(model, score, h, batches[0], logits=True)
    for i in (0, 1, 2):
        out_p = plan(batches[i]).clone()
        out_g = graph(batches[i]).clone()
        torch.cuda.synchronize()
        assert plan.check() and graph.check()
        assert torch.equal(out_p, eager[i]), f"{name}: recorded plan != eager (batch {i})"
        assert torch.equal(out_g, eager[i]), f"{name}: captured graph != eager (batch {i})"
    # four lanes (the bench's rotation): every lane's result bitwise the serial one
    lanes = model.lanes(4)
    for _attempt in range(3):
        outs = []
        for i, lane in enumerate(lanes):
            lane.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(lane):
                outs.append(model.score_pairs(batches[i], h, score, logits=True))
        torch.cuda.synchronize()
        if all(model.check_selection(lane) for lane in lanes):
            break
    for i in range(4):
        assert torch.equal(outs[i], eager[i]), f"{name}: lane {i} != serial"
    # probabilities (what the bench's step returns) are the sigmoid of those logits
    prob = model.score_pairs(batches[0], h, score)
    assert model.check_selection()
    assert (prob - torch.sigmoid(eager[0])).abs().max().item() <= 1e-6


STRESS = list(itertools.product((128, 256), (1.0, 6.0, 30.0, 120.0),
                                ((0.0, 1e-3, 1e-2), (0.0, 1e-2, 1.0), (1e-3, 1.0, 1.0)), (0, 1)))


def stress_case(k, dim, gain, th, seed):
    """One configuration of the pattern-table stress sweep: random graph, PE weights scaled by ``gain`` (coverage of the
    activation-pattern table from ~100 % to a few per cent), trained-like LayerNorm offsets; select4 -> rows4 -> tail
    against the oracle.  Returns (relative logit error, flips raw, flips left, coverage per type, mask mode, entries)."""
    rng = np.random.default_rng(1000 + k)
    n = int(rng.integers(300, 1500))
    ei, w = D.chung_lu_graph(n, int(n * rng.uniform(3, 14)), gamma=float(rng.uniform(2.05, 2.8)), seed=k,
                             max_weight=int(rng.integers(0, 2)) * 5)
    x = rng.standard_normal((n, 24)).astype(np.float32)
    ppr = lpformer_amd.calc_ppr(ei, n, 0.15, 10.0 ** rng.uniform(-4.3, -3))
    d = D.build_data(ei, x, n, edge_weight=w, ppr=ppr)
    cfg = D.train_args_for(dict(thresholds=th, dim=dim, gnn_layers=1, residual=False))
    torch.manual_seed(seed + 10 * k)
    model = lpformer_amd.LinkTransformer(cfg, d, device=DEV).to(DEV).eval()
    score = lpformer_amd.mlp_score(2 * dim, 2 * dim, 1, 2).to(DEV).eval()
    with torch.no_grad():
        for nm in ("ppr_encoder_cn", "ppr_encoder_onehop", "ppr_encoder_non1hop"):
            enc = getattr(model, nm, None)
            if enc is None:
                continue
            enc.linears[0].weight.mul_(gain)
            enc.linears[0].bias.mul_(float(rng.uniform(0.2, 2.0)))
            enc.norm.bias.add_(float(rng.uniform(0, 0.5)) * torch.randn_like(enc.norm.bias))
            enc.norm.weight.mul_(1.0 + 0.3 * torch.randn_like(enc.norm.weight))
    model.attention_impl = "flip"
    model.PT_EXACT_MAX = float("inf")
    batch = D.sample_pairs(ei, n, int(rng.integers(200, 900)), seed=k)
    ref = O.forward(batch, x, O.gcn_norm(ei, w, n), O.symmetric_mask_csr(ei, n),
                    (ppr.rowptr, ppr.col.astype(np.int64), ppr.val), _params(model, score), dict(cfg, pred_layers=2))
    assert model._uses_select4() and model._uses_rows()
    lg = model.score_pairs(torch.from_numpy(batch).to(DEV), model.propagate(), score, logits=True)
    assert model.check_selection()
    raw, left = model._flip_stats()
    cov = [s["covered"] for s in model._pattern_tables(model._fold())["stats"]]
    n_sel = sum(v[0].shape[1] for v in ref["sel"].values() if v is not None and v[0] is not None)
    return _rel(lg.cpu().numpy(), ref["logit"]), raw, left, cov, model.mask, n_sel


@pytest.mark.parametrize("k", range(len(STRESS)))
def test_pattern_table_stress(k):
    dim, gain, th, seed = STRESS[k]
    err, _, _, _, _, _ = stress_case(k + 1, dim, gain, th, seed)
    assert err <= TOL, f"case {k + 1} (D={dim}, gain={gain}, thresholds={th}): relative logit error {err:.2e}"


def test_step_on_trained_weights_matches_oracle():
    """150 steps of the repo's own training step (the bench's ``trained_weights`` leg: positives = existing edges,
    negatives = uniform pairs, Adam) on a collab-like graph at 1/10 size move the PE weights the way training does -- the
    no-flip square collapses, several flipped units per entry, the pattern table no longer covers everything -- then the
    bench's step against the oracle."""
    cfg = dict(D.CONFIGS["collab"])
    n = int(cfg["n"] * 0.1)
    ei, w = D.chung_lu_graph(n, int(cfg["edges"] * 0.1), gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
    x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
    data = D.build_data(ei, x, n, edge_weight=w, eps=cfg["eps"], ppr_device=DEV)
    args = D.train_args_for(cfg)
    torch.manual_seed(0)
    model = lpformer_amd.LinkTransformer(args, data, device=DEV).to(DEV)
    score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, 2).to(DEV)
    pos_e = torch.from_numpy(ei[:, ei[0] < ei[1]]).to(DEV)
    opt = torch.optim.Adam(list(model.parameters()) + list(score.parameters()), lr=1e-3)
    gen = torch.Generator(device=DEV)
    gen.manual_seed(4321)
    model.train(); score.train()
    tb = 1024
    for _ in range(150):
        idx = torch.randint(0, pos_e.shape[1], (tb,), device=DEV, generator=gen)
        neg = torch.randint(0, n, (2, tb), device=DEV, generator=gen)
        loss = (-torch.log(score(model(pos_e[:, idx])) + 1e-6).mean() - torch.log(1 - score(model(neg)) + 1e-6).mean())
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
    model.eval(); score.eval()
    raw0 = model.flips_per_entry(raw=True)
    assert raw0 > 0.5, f"training did not move the PE weights ({raw0} flipped units per entry)"
    batch = D.sample_pairs(ei, n, 4096, seed=9)
    tb_ = torch.from_numpy(batch).to(DEV)
    h = model.propagate()
    model.attention_impl = "flip"           # (the bench's `auto` keeps this kernel on trained weights: flips left < 5)
    assert model._uses_select4() and model._uses_rows()
    for _attempt in range(3):
        lg = model.score_pairs(tb_, h, score, logits=True)
        if model.check_selection():
            break
    cov = [s["covered"] for s in model._pattern_tables(model._fold())["stats"] if s["covered"] is not None]
    assert min(cov) < 1.0, f"the pattern table still covers every entry ({cov}): the exact path was not exercised"
    want = _oracle_logits(model, score, data, args, batch[:, :2048], h)
    err = _rel(lg.cpu().numpy()[:2048], want)
    assert err <= TOL, f"trained weights: the step differs from the oracle by {err:.2e}"
    # ... and from a recorded plan, bitwise
    plan = lpformer_amd.PlannedScorer(model, score, h, tb_, logits=True)
    out = plan(tb_).clone()
    torch.cuda.synchronize()
    assert plan.check() and torch.equal(out, lg)
