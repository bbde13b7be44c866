"""The one-launch, pair-major selection (csrc/select4.hip, ``lpf_select4``) and the pair-major attention that reads it
(``lpf_pair_attention_rows4_*``): index sets and PPR values BIT-EXACT against the reference fixtures and the oracle
(src/models/link_transformer.py:214-319, 434-481), scores within 1e-4 of the oracle and within 2e-6 of the two-launch
type-major path; blocks that span several 4,096-slot batches (hub pairs), ragged last blocks, every mask mode, a
workspace that is too small (NaN + sticky bit, then a correct re-score)."""
import numpy as np
import pytest
import torch

import lpformer_amd
from lpformer_amd import data as D
from oracle import lpformer_oracle as O
from tests.golden_util import LP_CASES, Fixture
from tests.test_gpu_parity import _build

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


def sel4_sets(model, batch, test_set=False, threads=0):
    """Runs lpf_select4 through the model's own workspace plumbing and converts what it leaves (pair-major entries, a
    table entry per pair) into the reference's layout: per type (ix int64 [2, n], pa, pb) sorted by (pair, node)."""
    model.select4_threads = threads
    batch = model._prep_batch(batch)
    with torch.cuda.device(model.device):
        ws = model._select4_device(batch, test_set)
        assert model.check_selection()
    bs = batch.shape[1]
    tab = ws.pair_tab[:4 * bs].view(bs, 4).cpu().numpy().astype(np.int64)
    ent = ws.entries.view(-1, 4).cpu().numpy()
    blk = ws.blk_cnt.cpu().numpy().reshape(-1, 2)
    cnt = tab[:, 1:].sum(1)
    # the table is consistent with itself and with the block counts {entries, pairs with entries}
    for b in range((bs + 63) // 64):
        assert blk[b, 0] == cnt[64 * b: 64 * b + 64].sum() and blk[b, 1] == (cnt[64 * b: 64 * b + 64] > 0).sum()
    assert (tab[:, 0] >= 0).all() and (tab[:, 0] + cnt <= ws.ent_cap).all()
    order = np.argsort(tab[:, 0], kind="stable")     # no two pairs' entries overlap
    nz = order[cnt[order] > 0]
    assert (tab[nz[1:], 0] >= tab[nz[:-1], 0] + cnt[nz[:-1]]).all()
    idx = np.concatenate([np.arange(s, s + c) for s, c in zip(tab[:, 0], cnt)]) if cnt.sum() else np.zeros(0, np.int64)
    rec = ent[idx]
    word = rec[:, 0].view(np.uint32) if rec.size else np.zeros(0, np.uint32)
    pair, typ = (word & 0x1FFFFFFF).astype(np.int64), ((word >> 29) & 3).astype(np.int64)
    # every entry sits in its own pair's range, the per-type counts are the table's
    assert np.array_equal(pair, np.repeat(np.arange(bs), cnt))
    for t in (1, 2, 3):
        assert np.array_equal(np.bincount(pair[typ == t], minlength=bs), tab[:, t])
    out = {}
    for t, tag in ((1, "cn"), (2, "onehop"), (3, "non1hop")):
        m = typ == t
        p, v = pair[m], rec[m, 1].astype(np.int64)
        order = np.lexsort((v, p))
        out[tag] = (np.stack([p[order], v[order]]), rec[m, 2][order].view(np.float32), rec[m, 3][order].view(np.float32))
    return out, ws


def _assert_sets_equal(got, want, tags):
    for tag in tags:
        np.testing.assert_array_equal(got[tag][0], want[tag][0])
        np.testing.assert_array_equal(got[tag][1].view(np.uint32), np.asarray(want[tag][1]).view(np.uint32))
        np.testing.assert_array_equal(got[tag][2].view(np.uint32), np.asarray(want[tag][2]).view(np.uint32))
    for tag in ("cn", "onehop", "non1hop"):
        if tag not in tags:
            assert got[tag][0].shape[1] == 0


# launch shapes: workgroup size + 4096 * (blocks of 64 pairs per workgroup - 1); 0 = the default (1,024 threads; two
# blocks per workgroup for batches that still give every CU a workgroup then, one otherwise)
SHAPES = [0, 1024, 4096 + 1024, 512, 4096 + 512, 3 * 4096 + 1024, 256]


@pytest.mark.parametrize("case", LP_CASES)
@pytest.mark.parametrize("threads", SHAPES)
def test_select4_bit_exact_vs_reference_fixtures(case, threads):
    fx = Fixture(case)
    model, _ = _build(fx)
    got, _ = sel4_sets(model, torch.from_numpy(fx["batch"]), test_set=fx.test_set, threads=threads)
    want = {t: (fx[f"sel_{t}_ix"], fx[f"sel_{t}_pa"], fx[f"sel_{t}_pb"]) for t in fx.sel_tags()}
    _assert_sets_equal(got, want, fx.sel_tags())


SWEEP = [
    # seed, n, undirected edges, gamma, dim, thresholds (cn, 1hop, >1hop), eps, weighted, hubs, batch size
    (1, 900, 9000, 2.05, 64, (0.0, 1e-4, 1e-2), 1e-4, True, True, 700),      # hub pairs: blocks of several batches
    (2, 1500, 6000, 2.2, 128, (1e-3, 1e-3, 5e-3), 2e-4, True, False, 1000),
    (3, 600, 20000, 3.0, 256, (0.0, 1e-2, 1.0), 1e-4, False, False, 333),    # dense, "1-hop" mode; ragged last block
    (4, 2500, 7000, 2.1, 64, (0.0, 0.0, 1e-2), 1e-4, False, False, 700),     # theta_1hop = 0: px rows cannot stand in
    (5, 1200, 15000, 2.05, 128, (0.0, 1e-5, 1e-3), 5e-5, True, False, 65),   # long PPR rows, many >1-hop nodes
    (7, 800, 12000, 2.02, 256, (0.0, 1e-4, 1e-2), 1e-4, True, True, 700),
    (9, 500, 12000, 2.6, 128, (2e-3, 1.0, 1.0), 1e-4, True, False, 640),     # mask mode "cn"
]


def _sweep_case(case):
    seed, n, edges, gamma, dim, th, eps, weighted, hubs, bs = case
    rng = np.random.default_rng(300 + seed)
    ei, w = D.chung_lu_graph(n, edges, gamma=gamma, seed=seed, max_weight=6 if weighted else 0)
    if hubs:
        star = np.concatenate([np.stack([np.zeros(640, np.int64), rng.choice(np.arange(2, n), 640, replace=False)]),
                               np.stack([np.ones(560, np.int64), rng.choice(np.arange(2, n), 560, replace=False)])], 1)
        allp = np.concatenate([ei, star, star[::-1]], axis=1)
        allw = None if w is None else np.concatenate([w, np.ones(2 * star.shape[1], np.float32)])
        _, keep = np.unique(allp[0] * n + allp[1], return_index=True)
        ei, w = allp[:, keep], (None if allw is None else allw[keep])
    x = rng.standard_normal((n, 24)).astype(np.float32)
    ppr = lpformer_amd.calc_ppr(ei, n, 0.15, eps)
    d = D.build_data(ei, x, n, edge_weight=w, ppr=ppr)
    cfg = D.train_args_for(dict(thresholds=th, dim=dim, gnn_layers=1, residual=False))
    torch.manual_seed(seed)
    model = lpformer_amd.LinkTransformer(cfg, d, device=DEV).to(DEV).eval()
    score = lpformer_amd.mlp_score(2 * dim, 2 * dim, 1, 2).to(DEV).eval()
    batch = D.sample_pairs(ei, n, bs, seed=seed + 50)
    deg = np.bincount(ei[0], minlength=n)
    hub = int(np.argmax(deg))
    batch[:, :6] = np.array([[0, 5, 7, 7, hub, hub], [0, 5, 9, 9, (hub + 1) % n, hub]])  # a == b, duplicates, hub pairs
    batch[:, 7:9] = np.array([[0, 1], [1, 0]])
    if hubs:   # a whole block of hub pairs: its slots do not fit one 4,096-slot batch of the kernel
        batch[:, 64:128] = np.array([[0, 1] * 32, [1, 0] * 32])
    iso = np.flatnonzero(deg == 0)
    if iso.size >= 2:
        batch[:, 6] = iso[:2]
    return model, score, batch, (ei, w, x, ppr, cfg, n)


@pytest.mark.parametrize("case", SWEEP, ids=[f"seed{c[0]}_d{c[4]}" for c in SWEEP])
def test_select4_matches_oracle_and_the_type_major_path(case):
    model, score, batch, (ei, w, x, ppr, cfg, n) = _sweep_case(case)
    tags = {"all": ("cn", "onehop", "non1hop"), "1-hop": ("cn", "onehop"), "cn": ("cn",)}[model.mask]
    bt = torch.from_numpy(batch)
    # the oracle's sets (CPU restatement of the reference) ...
    ref = O.select_nodes(batch, O.symmetric_mask_csr(ei, n), (ppr.rowptr, ppr.col.astype(np.int64), ppr.val),
                         (cfg["thresh_cn"], cfg["thresh_1hop"], cfg["thresh_non1hop"]), n=n)
    for threads in SHAPES:
        got, ws = sel4_sets(model, bt, threads=threads)
        _assert_sets_equal(got, ref, tags)
    # ... and the two-launch type-major path (select3.hip + lpf_select_export) say the same
    infos = model.compute_node_mask(bt)
    _assert_sets_equal(got, {t: tuple(v.cpu().numpy() for v in i) for t, i in zip(tags, infos)}, tags)
    assert sum(got[t][0].shape[1] for t in tags) > 100
    # scores: oracle within 1e-4; the two selection forms within 2e-6 of each other (a pair's entries are summed in
    # candidate-slot order instead of type by type)
    if cfg["dim"] >= 128:
        model.attention_impl = "flip"
        h = model.propagate()
        P = {f"model.{k}": v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        P.update({f"score.{k}": v.detach().cpu().numpy() for k, v in score.state_dict().items()})
        full = O.forward(batch, x, O.gcn_norm(ei, w, n), O.symmetric_mask_csr(ei, n),
                         (ppr.rowptr, ppr.col.astype(np.int64), ppr.val), P, dict(cfg, pred_layers=2))
        outs = {}
        for blocks in (True, False):
            model.select_blocks = blocks
            lg = model.score_pairs(bt.to(DEV), h, score, logits=True)
            assert model.check_selection()
            outs[blocks] = lg.cpu().numpy()
            assert np.abs(outs[blocks] - full["logit"]).max() <= TOL * max(1.0, float(np.abs(full["logit"]).max())), blocks
        assert np.abs(outs[True] - outs[False]).max() <= 2e-6 * max(1.0, float(np.abs(outs[False]).max()))
        # module-by-module API through the same kernels
        model.select_blocks = True
        feats = model(bt)
        assert np.abs(feats.cpu().numpy() - full["combined_feats"]).max() <= \
            TOL * max(1.0, float(np.abs(full["combined_feats"]).max()))


def test_select4_overflow_raises_the_sticky_bit_and_recovers():
    """A workspace sized for a sparse batch, then a batch that needs more room: NaN scores + the sticky bit, and
    ``check_selection()`` makes the next call size the buffer again -- nothing is written or read outside it."""
    model, score, batch, _ = _sweep_case(SWEEP[1])
    model.attention_impl = "flip"
    h = model.propagate()
    n = model.num_nodes
    sparse = torch.from_numpy(np.stack([np.arange(1000) % n, (np.arange(1000) * 7 + 3) % n]).astype(np.int64)).to(DEV)
    dense = torch.from_numpy(batch).to(DEV)
    lg0 = model.score_pairs(sparse, h, score, logits=True)
    assert model.check_selection() and torch.isfinite(lg0).all()
    ws = next(v for k, v in model._ws.items() if isinstance(k, tuple) and k[0] == "sel4")
    ws.ensure(ent_cap=64, shrink=True)       # (force the condition whatever the two batches need)
    lg1 = model.score_pairs(dense, h, score, logits=True)
    assert torch.isnan(lg1).any()
    assert not model.check_selection()
    lg2 = model.score_pairs(dense, h, score, logits=True)
    assert model.check_selection() and torch.isfinite(lg2).all()
    model.select_blocks = False
    lg3 = model.score_pairs(dense, h, score, logits=True)
    assert model.check_selection()
    assert (lg2 - lg3).abs().max().item() <= 2e-6 * max(1.0, lg3.abs().max().item())


def test_select4_replays_bitwise_and_ignores_where_blocks_land():
    """Scores do not depend on where the allocation atomic put a block: the same batch scored repeatedly, alone and
    pipelined over streams, gives the same bits."""
    model, score, batch, _ = _sweep_case(SWEEP[4])
    model.attention_impl = "flip"
    h = model.propagate()
    bt = torch.from_numpy(batch).to(DEV)
    first = model.score_pairs(bt, h, score, logits=True).clone()
    assert model.check_selection()
    for _ in range(5):
        again = model.score_pairs(bt, h, score, logits=True)
        assert torch.equal(first, again)
    outs = []
    for st in model.lanes(4):
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            outs.append(model.score_pairs(bt, h, score, logits=True))
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(first, o)


def _regions(model, batch, test_set, form):
    """type_ptr [3, bs + 1] and the three regions' records as lpf_select3_run (form False) / lpf_select4 +
    lpf_select4_regions (form True) leave them."""
    from lpformer_amd.link_transformer import _Select4RegionsWorkspace
    model.select4_regions = form
    with torch.cuda.device(model.device):
        for _attempt in range(3):
            ws = model._select_device(batch, test_set)
            if model.check_selection():
                break
        else:
            raise AssertionError("the selection workspace could not be sized")
    assert isinstance(ws, _Select4RegionsWorkspace) == form
    bs = batch.shape[1]
    tp = ws.type_ptr[:3 * (bs + 1)].view(3, bs + 1).cpu().numpy()
    ent = ws.entries.view(3, ws.ent_cap, 4).cpu().numpy()
    return tp, [ent[t, :tp[t, bs]].copy() for t in range(3)]


@pytest.mark.parametrize("case", LP_CASES)
def test_select4_regions_are_select3s_regions(case):
    """``lpf_select4`` + ``lpf_select4_regions`` leave what ``lpf_select3_plan`` / ``_run`` leave -- the same segment
    pointers, the same records in the same order (pair, candidate slot) -- so every consumer of the type-major form (the
    matrix-core attention, the merging tail, ``lpf_select_export`` with its merge of the one-hop runs) reads either."""
    fx = Fixture(case)
    model, _ = _build(fx)
    batch = model._prep_batch(torch.from_numpy(fx["batch"]))
    tp4, ent4 = _regions(model, batch, fx.test_set, True)
    tp3, ent3 = _regions(model, batch, fx.test_set, False)
    np.testing.assert_array_equal(tp4, tp3)
    for t in range(3):
        np.testing.assert_array_equal(ent4[t], ent3[t])
    # ... and the reference layout out of the export, bit-exact against the fixture
    model.select4_regions = True
    infos = model.compute_node_mask(torch.from_numpy(fx["batch"]), test_set=fx.test_set)
    for tag, info in zip(("cn", "onehop", "non1hop"), infos):
        if info is None:
            assert tag not in fx.sel_tags()
            continue
        np.testing.assert_array_equal(info[0].cpu().numpy(), fx[f"sel_{tag}_ix"])
        np.testing.assert_array_equal(info[1].cpu().numpy().view(np.uint32), fx[f"sel_{tag}_pa"].view(np.uint32))
        np.testing.assert_array_equal(info[2].cpu().numpy().view(np.uint32), fx[f"sel_{tag}_pb"].view(np.uint32))


@pytest.mark.parametrize("name,scale,bs,seed", [("ppa", 0.02, 5000, 0), ("citation2", 0.01, 3001, 1), ("collab", 0.05, 777, 2),
                                                ("ddi", 0.25, 300, 3), ("tiny", 1.0, 64, 4), ("tiny", 1.0, 1, 5)])
def test_select4_regions_on_hub_heavy_and_ragged_batches(name, scale, bs, seed):
    """Hub pairs whose walks span several slot batches, ragged last blocks, a one-pair batch, dense rows: regions and
    pointers equal select3's; scores through the matrix-core attention behind either selection bitwise equal; a
    workspace that is too small raises the sticky bit and the re-scored batch is right."""
    cfg = dict(D.CONFIGS[name])
    n = max(64, int(cfg["n"] * scale))
    ei, w = D.chung_lu_graph(n, int(cfg["edges"] * scale), gamma=cfg["gamma"], seed=seed, max_weight=cfg["max_weight"])
    x = np.random.default_rng(seed).standard_normal((n, 16)).astype(np.float32)
    data = D.build_data(ei, x, n, edge_weight=w, eps=max(cfg["eps"], 1e-4))
    args = D.train_args_for(dict(cfg, dim=64, gnn_layers=1))
    torch.manual_seed(seed)
    model = lpformer_amd.LinkTransformer(args, data, device=DEV).to(DEV).eval()
    score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, 2).to(DEV).eval()
    batch = model._prep_batch(torch.from_numpy(D.sample_pairs(ei, n, bs, seed=seed + 7)))
    tp4, ent4 = _regions(model, batch, False, True)
    tp3, ent3 = _regions(model, batch, False, False)
    np.testing.assert_array_equal(tp4, tp3)
    for t in range(3):
        np.testing.assert_array_equal(ent4[t], ent3[t])
    h = model.propagate()
    out = {}
    for form in (True, False):
        model.select4_regions = form
        for _attempt in range(3):
            lg = model.score_pairs(batch, h, score, logits=True).clone()
            if model.check_selection():
                break
        out[form] = lg
    assert torch.isfinite(out[True]).all() and torch.equal(out[True], out[False])
    # a region that is too small: sticky bit + NaN, then the right scores
    model.select4_regions = True
    ws = model._select_device(batch, False)
    assert model.check_selection()
    if int(tp4[:, -1].max()) > 16:
        ws.ensure(ent_cap=8, shrink=True)
        bad = model.score_pairs(batch, h, score, logits=True)
        assert not model.check_selection() and torch.isnan(bad).any()
        for _attempt in range(3):
            again = model.score_pairs(batch, h, score, logits=True)
            if model.check_selection():
                break
        assert torch.equal(again, out[True])
