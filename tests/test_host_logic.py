"""CPU-only checks: C-ABI libraries load and export every declared symbol, host PPR producer is bit-exact against
the reference's golden vectors and the oracle, module surface/state_dict keys match the reference."""
import os
import re

import numpy as np
import pytest
import torch

import lpformer_amd
from lpformer_amd import _lib, fold, graph
from oracle import lpformer_oracle as O
from tests.golden_util import GOLDEN_DIR, LP_CASES, PPR_CASES, Fixture

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_header_symbols_exported():
    """Every function declared in include/lpformer_hip.h is exported by one of the two libraries and bound."""
    hdr = open(os.path.join(ROOT, "include", "lpformer_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(lpf_[a-z0-9_]+)\s*\(", hdr))
    bound = set(_lib.HIP_PROTOTYPES) | set(_lib.HOST_PROTOTYPES)
    assert declared == bound, (declared ^ bound)
    hip, host = _lib.hip(), _lib.host()  # loading must work without a GPU (no compute calls here)
    for name in _lib.HIP_PROTOTYPES:
        assert hasattr(hip, name)
    for name in _lib.HOST_PROTOTYPES:
        assert hasattr(host, name)
    assert hip.lpf_abi_version() == _lib.ABI_VERSION == host.lpf_host_abi_version()
    assert b"invalid" in hip.lpf_strerror(-1)


@pytest.mark.parametrize("case", PPR_CASES)
def test_host_ppr_push_matches_reference(case):
    z = np.load(f"{GOLDEN_DIR}/{case}.npz")
    n = int(z["n"])
    for eps in z["eps_list"]:
        tag = f"{eps:g}".replace("-", "m").replace(".", "p")
        for threads in (1, 3):
            csr = lpformer_amd.ppr.calc_ppr(z["edge_index"], n, 0.15, float(eps), num_threads=threads)
            rows = np.repeat(np.arange(n), np.diff(csr.rowptr))
            np.testing.assert_array_equal(rows, z[f"row_{tag}"])
            np.testing.assert_array_equal(csr.col, z[f"col_{tag}"])
            np.testing.assert_array_equal(csr.val.view(np.uint32), z[f"val_{tag}"].view(np.uint32))


def test_host_ppr_push_matches_oracle_on_directed_graph():
    rng = np.random.default_rng(3)
    n = 90
    ei = rng.integers(0, n, size=(2, 400))
    ei = np.concatenate([ei, np.stack([np.arange(5), np.arange(5)])], axis=1)  # self loops, dangling nodes exist
    rowptr, col = O.edge_csr(ei, n)
    r, c, v = O.ppr_push(rowptr, col, 0.15, 1e-3)
    csr = lpformer_amd.ppr.calc_ppr(ei, n, 0.15, 1e-3, num_threads=2)
    np.testing.assert_array_equal(np.repeat(np.arange(n), np.diff(csr.rowptr)), r)
    np.testing.assert_array_equal(csr.col, c)
    np.testing.assert_array_equal(csr.val.view(np.uint32), v.view(np.uint32))


def _model_for(fx, device="cpu"):
    n = fx.n
    data = {"x": torch.from_numpy(fx["x"]), "num_nodes": n}
    cfg = {k: fx.cfg[k] for k in ("thresh_cn", "thresh_1hop", "thresh_non1hop", "dim", "trans_layers", "num_heads",
                                  "att_drop", "dropout", "gnn_drop", "feat_drop", "gcn_cache", "gnn_layers",
                                  "residual", "layer_norm", "relu")}
    return lpformer_amd.LinkTransformer(cfg, data, device=device)


@pytest.mark.parametrize("case", LP_CASES)
def test_state_dict_keys_match_reference(case):
    fx = Fixture(case)
    model = _model_for(fx)
    score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, fx.cfg["pred_layers"])
    want = fx.cfg["param_shapes"]
    got = {f"model.{k}": list(v.shape) for k, v in model.state_dict().items()}
    got.update({f"score.{k}": list(v.shape) for k, v in score.state_dict().items()})
    assert got == want
    m_sd, s_sd = fx.state_dicts()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in m_sd.items()}, strict=True)
    score.load_state_dict({k: torch.from_numpy(v) for k, v in s_sd.items()}, strict=True)


def test_no_cpu_fallback():
    fx = Fixture("lp_all_d64")
    model = _model_for(fx).eval()
    with pytest.raises(Exception):
        model.elementwise_lin(torch.zeros(4, 64))
    with pytest.raises(NotImplementedError):
        model.train()
        model.propagate()


def test_graph_builders_match_oracle():
    fx = Fixture("lp_all_d128_weighted")
    n = fx.n
    m_o = O.symmetric_mask_csr(fx.edge_index, n)
    m = graph.mask_csr(fx.edge_index, n)
    np.testing.assert_array_equal(m.rowptr, m_o[0])
    np.testing.assert_array_equal(m.col, m_o[1])
    s = graph.gcn_structure_csr(fx.edge_index, fx.edge_weight, n)
    a_o = O.gcn_norm(fx.edge_index, fx.edge_weight, n)
    np.testing.assert_array_equal(s.rowptr, a_o[0])
    np.testing.assert_array_equal(s.col, a_o[1])
    r, c, v = fx.ppr_coo
    p = graph.csr_from_coo(r, c, v, n)
    pf = graph.prefilter_nonhop(p, fx.cfg["thresh_non1hop"])
    assert 0 < pf.nnz < p.nnz
    coo = p.to_torch_sparse_coo()
    r2, c2, v2, n2 = graph.as_coo_numpy(coo)
    np.testing.assert_array_equal(r2, r)
    np.testing.assert_array_equal(c2, c)
    np.testing.assert_array_equal(v2, v)


def test_fold_algebra_matches_oracle():
    """k_e = Z[v] + Wfold h_e + bfold and the closed-form LayerNorm statistics reproduce the reference's
    lin_r([X[v] ; pe_e]) (float64 check of the algebra the kernels rely on)."""
    fx = Fixture("lp_all_d64")
    P = fx.params
    m_sd, _ = fx.state_dicts()
    sd = {k: torch.from_numpy(v) for k, v in m_sd.items()}
    d = fx.cfg["dim"]
    w = fold.fold_attention(sd, d, 3)
    tab, stat = fold.pe_tables(sd, d, 3)
    rng = np.random.default_rng(0)
    pa, pb = rng.random(50).astype(np.float32) * 0.1, rng.random(50).astype(np.float32) * 0.1
    xv = rng.standard_normal((50, d)).astype(np.float32)
    for t, tag in enumerate(("cn", "onehop", "non1hop")):
        sel = {tag: (np.zeros((2, 50), np.int64), pa, pb)}
        pes = O.pos_encodings(sel, P)
        k_ref = O.linear(np.concatenate([xv, pes], axis=1), P["model.att_layers.0.att.lin_r.weight"],
                         P["model.att_layers.0.att.lin_r.bias"])

        def hidden(x, y):
            var = (stat[t, 0] * x * x + stat[t, 1] * y * y + stat[t, 2]
                   + 2 * (stat[t, 3] * x * y + stat[t, 4] * x + stat[t, 5] * y))
            rstd = 1.0 / np.sqrt(var + 1e-5)
            u = tab[t, :, 0][None] * x[:, None] + tab[t, :, 1][None] * y[:, None] + tab[t, :, 2][None]
            return np.maximum(rstd[:, None] * u + tab[t, :, 3][None], 0)

        h = hidden(pa, pb) + hidden(pb, pa)
        k = xv @ w["w_rx"].T + w["b_r"] + h @ w["wfold"][t].T + w["bfold"][t]
        assert np.abs(k - k_ref).max() < 2e-5
    # packed image is a permutation of wfold
    pk = w["wfold_packed"]
    c, sq, lane, u = 1, 3, 45, 2
    assert pk[2, c, sq, lane, u] == w["wfold"][2, 32 * c + (lane & 31), (lane >> 5) * (d // 2) + 4 * sq + u]


def test_flip_tables_reproduce_the_key_projection():
    """The activation-pattern form of the PE key projection (csrc/pair_flip.hip): with the four base vectors of the units
    active at (0, 0) and a correction Wfold[:, k] |y_k| per unit whose ReLU state differs from that pattern,
    Wfold h_e + bfold comes out as the reference's lin_r([. ; pe_e]) -- for inputs near 0 (no unit flips), for typical PPR
    values and for large ones (many flips), in float64 to 1e-9 of the direct product."""
    fx = Fixture("lp_all_d64")
    m_sd, _ = fx.state_dicts()
    sd = {k: torch.from_numpy(v) for k, v in m_sd.items()}
    d = fx.cfg["dim"]
    w = fold.fold_attention(sd, d, 3)
    tab, stat = (a.astype(np.float64) for a in fold.pe_tables(sd, d, 3))
    tabs, base, s0, wt = fold.flip_tables(sd, d, 3)
    tabs, base, wt = tabs.astype(np.float64), base.astype(np.float64), wt.astype(np.float64)
    rng = np.random.default_rng(1)
    flips_seen = 0
    for scale in (1e-4, 2e-2, 0.5, 5.0):
        pa, pb = rng.random(64) * scale, rng.random(64) * scale
        for t in range(3):
            def rstd(x, y):
                var = (stat[t, 0] * x * x + stat[t, 1] * y * y + stat[t, 2]
                       + 2 * (stat[t, 3] * x * y + stat[t, 4] * x + stat[t, 5] * y))
                return 1.0 / np.sqrt(var + 1e-5)

            def pre(tb, x, y, r):
                return r[:, None] * (tb[t, :, 0][None] * x[:, None] + tb[t, :, 1][None] * y[:, None] + tb[t, :, 2][None]) \
                    + tb[t, :, 3][None]
            r1, r2 = rstd(pa, pb), rstd(pb, pa)
            h = np.maximum(pre(tab, pa, pb, r1), 0) + np.maximum(pre(tab, pb, pa, r2), 0)
            want = h @ w["wfold"][t].astype(np.float64).T + w["bfold"][t]
            # the kernel's evaluation
            z1, z2 = pre(tabs, pa, pb, r1), pre(tabs, pb, pa, r2)        # negative <=> the unit left the pattern of (0, 0)
            got = (base[t, 0][None] * (r1 * pa + r2 * pb)[:, None] + base[t, 1][None] * (r1 * pb + r2 * pa)[:, None]
                   + base[t, 2][None] * (r1 + r2)[:, None] + base[t, 3][None])
            got = got + np.maximum(-z1, 0) @ wt[t] + np.maximum(-z2, 0) @ wt[t]
            flips_seen += int((z1 < 0).sum() + (z2 < 0).sum())
            assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max())   # (fp32-stored tables)
    assert flips_seen > 100           # the large inputs did exercise the correction path
    assert set(np.unique(s0)) <= {0, 1}


def test_ppr_cache_round_trip(tmp_path):
    """load_or_calc_ppr: first call computes and stores under the reference's directory / file naming, the second
    call reads the same matrix back; a cache for another graph size, or one that was computed on a different edge set
    under the same name (edge fingerprint in the file), is refused."""
    rng = np.random.default_rng(0)
    n = 120
    ei = rng.integers(0, n, size=(2, 500))
    a = lpformer_amd.ppr.load_or_calc_ppr(ei, n, 0.15, 1e-3, cache_root=str(tmp_path), dataset="toy", is_val=True)
    path = lpformer_amd.ppr.ppr_cache_path(str(tmp_path), "toy", 0.15, 1e-3, True)
    assert path.endswith(os.path.join("node_subsets", "ppr", "toy", "sparse_adj-015_eps-0001_val.lpf.npz"))
    assert os.path.isfile(path)
    b = lpformer_amd.ppr.load_or_calc_ppr(ei[:, ::-1].copy(), n, 0.15, 1e-3, cache_root=str(tmp_path),
                                          dataset="toy", is_val=True)  # same edge set (any order): served from the cache
    with pytest.raises(_lib.LpfError):                                 # another edge set under the same name
        lpformer_amd.ppr.load_or_calc_ppr(ei[:, :400], n, 0.15, 1e-3, cache_root=str(tmp_path), dataset="toy", is_val=True)
    np.testing.assert_array_equal(a.rowptr, b.rowptr)
    np.testing.assert_array_equal(a.col, b.col)
    np.testing.assert_array_equal(a.val.view(np.uint32), b.val.view(np.uint32))
    with pytest.raises(_lib.LpfError):
        lpformer_amd.ppr.load_or_calc_ppr(ei, n + 1, 0.15, 1e-3, cache_root=str(tmp_path), dataset="toy", is_val=True)


def test_ranking_metrics_match_reference_formulas():
    """hits@K and the optimistic/pessimistic-rank MRR against plain numpy restatements of the reference's formulas
    (src/train/evaluation.py:23-50 and the OGB hits@K rule), with ties."""
    from lpformer_amd import evaluate as E
    rng = np.random.default_rng(0)
    pos = np.round(rng.random(300), 2).astype(np.float32)  # rounding creates ties
    neg = np.round(rng.random((300, 40)), 2).astype(np.float32)
    for k in (1, 10, 50, 100, 20000):
        flat = np.sort(neg.reshape(-1))[::-1]
        want = 1.0 if flat.size < k else float((pos > flat[k - 1]).mean())
        assert abs(E.hits_at_k(torch.from_numpy(pos), torch.from_numpy(neg), k) - want) < 1e-7
    opt = (neg >= pos[:, None]).sum(1)
    pes = (neg > pos[:, None]).sum(1)
    rank = 0.5 * (opt + pes) + 1
    got = E.ranking_metrics(torch.from_numpy(pos), torch.from_numpy(neg))
    assert abs(got["MRR"] - float((1.0 / rank).mean())) < 1e-6
    for k in (10, 50, 100):
        assert abs(got[f"Hits@{k}"] - float((rank <= k).mean())) < 1e-7


def test_encoder_plan_cost_model():
    from lpformer_amd import dist as LD
    # collab-like: the encoder is cheaper than four 121 MB all-gathers -> replicate
    p = LD.encoder_plan(1.1, 235_868, 128, 3, 8, allgather_gbps=300.0)
    assert p["mode"] == "replicated" and p["sharded_ms"] > p["replicated_ms"]
    # citation2-like: 10 ms of encoder against 4 x 750 MB -> shard when the exchange is fast enough
    assert LD.encoder_plan(10.0, 2_927_963, 64, 3, 8, allgather_gbps=1000.0)["mode"] == "sharded"
    assert LD.encoder_plan(10.0, 2_927_963, 64, 3, 8, allgather_gbps=200.0)["mode"] == "replicated"
    assert LD.encoder_plan(5.0, 1000, 64, 3, 1, allgather_gbps=1.0)["mode"] == "replicated"
    # the single all-gather of [X | Z] (2 d floats per node against (L + 1) x d) wins where the LAST layer's
    # aggregation and the per-node projections are most of the work -- a one-layer encoder with costly projections --
    # because it shards the projections too; with three layers and a fast exchange plain row sharding is cheaper
    p = LD.encoder_plan(10.0, 2_927_963, 64, 1, 8, allgather_gbps=500.0, last_agg_ms=9.5, node_keys_ms=4.0)
    assert p["mode"] == "gather_once" and p["gather_once_ms"] < min(p["replicated_ms"], p["sharded_ms"])
    assert LD.encoder_plan(10.0, 2_927_963, 64, 3, 8, allgather_gbps=1500.0, last_agg_ms=3.0,
                           node_keys_ms=1.0)["mode"] == "sharded"
    p = LD.encoder_plan(1.1, 235_868, 128, 3, 8, allgather_gbps=300.0, last_agg_ms=0.25, node_keys_ms=0.1)
    assert p["mode"] == "replicated" and set(p) >= {"replicated_ms", "sharded_ms", "gather_once_ms", "allgather_ms"}


def test_hashed_index_layout():
    """graph.hash_index_device: every entry sits in the bucket its hash names, no bucket holds more than 16, the
    compact view gives the input back, no bucket holds more than HASH_BUCKET (the builder is plain torch, so it runs on the CPU too)."""
    import torch
    from lpformer_amd import graph
    rng = np.random.default_rng(5)
    n = 300
    lens = rng.integers(0, 90, n)
    lens[7] = 0
    lens[11] = 2000                     # a long row
    cols, rp = [], np.zeros(n + 1, np.int64)
    for i in range(n):
        # (row 12: columns that all hash into few buckets at first -- forces the bucket count of the row to grow)
        c = np.sort(rng.choice(200_000, lens[i], replace=False)) if i != 12 else np.arange(60) * 4096
        lens[i] = c.size
        cols.append(c)
        rp[i + 1] = rp[i] + c.size
    col = np.concatenate(cols).astype(np.int32)
    val = rng.random(col.size).astype(np.float32)
    p = graph.DeviceCSR(torch.from_numpy(rp), torch.from_numpy(col), torch.from_numpy(val), n)
    h = graph.hash_index_device(p)
    back = h.to_host_compact()
    assert np.array_equal(back.rowptr, rp) and np.array_equal(back.col, col) and np.array_equal(back.val, val)
    cv, hrp, nbk = h.cv.numpy(), h.rowptr.numpy(), h.len.numpy().astype(np.int64)
    hb = graph.HASH_BUCKET
    assert np.array_equal(np.diff(hrp), hb * nbk) and nbk[7] == 0
    for i in (0, 11, 12, 299):
        for c in cols[i][:50]:
            b = ((int(c) * graph.HASH_MUL & 0xFFFFFFFF) * int(nbk[i])) >> 32
            line = cv[hrp[i] + hb * b: hrp[i] + hb * b + hb, 0]
            assert c in line


def test_get_ppr_has_the_reference_signature_and_cache_name(tmp_path):
    """``get_ppr(dataset, edge_index, num_nodes, alpha, eps, is_val)`` (src/util/calc_ppr_scores.py:245-270): the six
    positional arguments, the cache directory and file stem of the reference, a coalesced sparse COO tensor out; the
    second call loads the cache."""
    import os
    import lpformer_amd
    from lpformer_amd import ppr as P
    rng = np.random.default_rng(0)
    n = 40
    e = rng.integers(0, n, (2, 120))
    ei = np.concatenate([e, e[::-1]], axis=1)
    a = lpformer_amd.get_ppr("toy", torch.from_numpy(ei), n, 0.15, 1e-3, True, root_dir=str(tmp_path))
    assert a.layout == torch.sparse_coo and a.is_coalesced() and tuple(a.shape) == (n, n)
    ref = P.ppr_reference_cache_path(str(tmp_path), "toy", 0.15, 1e-3, True)
    assert ref.endswith(os.path.join("node_subsets", "ppr", "toy", "sparse_adj-015_eps-0001_val.pt"))
    assert os.path.isfile(ref[:-3] + ".lpf.npz")               # (no torch_sparse here: the CSR sibling)
    b = lpformer_amd.get_ppr("toy", torch.from_numpy(ei), n, 0.15, 1e-3, True, root_dir=str(tmp_path))
    assert torch.equal(a.indices(), b.indices()) and torch.equal(a.values(), b.values())
    want = lpformer_amd.calc_ppr(ei, n, 0.15, 1e-3)
    assert a._nnz() == want.nnz and torch.equal(a.values(), torch.from_numpy(want.val))


def test_fused_row_order_covers_rows_once_and_slices_hubs():
    """graph.fused_row_order (work list of csrc/gcn_fused.hip): every row of the block exactly once, hub rows as codes
    <= -2 in front, the rest by falling degree, padding -1; the hubs' slices tile their entry ranges."""
    import torch
    from lpformer_amd import graph as G
    rng = np.random.default_rng(0)
    deg = rng.integers(0, 40, size=1000)
    deg[[3, 500, 777]] = [129, 1000, 257]
    deg[10] = 128
    rowptr = torch.from_numpy(np.concatenate([[0], np.cumsum(deg)]).astype(np.int64))
    for lo, hi in ((0, 1000), (2, 779), (4, 500)):
        order, hubs, parts = G.fused_row_order(rowptr, lo, hi, long_threshold=128)
        o = order.numpy()
        assert o.size % 16 == 0
        hub_rows = [r for r in (3, 500, 777) if lo <= r < hi]
        if not hub_rows:
            assert hubs is None and parts is None
            n_h = 0
        else:
            h, p = hubs.numpy(), parts.numpy()
            n_h = len(hub_rows)
            assert h[:, 0].tolist() == hub_rows and (o[:n_h] == -2 - np.arange(n_h)).all()
            for k, r in enumerate(hub_rows):
                sl = p[h[k, 1]:h[k, 1] + h[k, 2]]
                assert sl[0, 0] == rowptr[r] and sl[-1, 1] == rowptr[r + 1]
                assert (sl[1:, 0] == sl[:-1, 1]).all() and ((sl[:, 1] - sl[:, 0]) <= 256).all()
                assert h[k, 2] == -(-deg[r] // 256)
            assert h[:, 1].tolist() == np.concatenate([[0], np.cumsum(h[:-1, 2])]).tolist()
        body = o[n_h:]
        live = body[body >= 0]
        assert (body[live.size:] == -1).all()
        assert sorted(live.tolist() + hub_rows) == list(range(lo, hi))
        d = deg[live]
        assert (d[:-1] >= d[1:]).all() and d.max() <= 128


def test_packed_square_weight_matches_pack_dense():
    """The device-side packing of a GCN layer's weight (link_transformer._PackedSquare, torch operators) is the image
    fold.pack_dense(w, 1) builds on the host."""
    import torch
    from lpformer_amd import fold
    from lpformer_amd.link_transformer import _PackedSquare
    for d in (32, 64, 128):
        w = torch.randn(d, d)
        assert np.array_equal(_PackedSquare().get(w).numpy(), fold.pack_dense(w.numpy(), 1))


def test_recording_hooks_report_launches_pointers_and_hand_overs():
    """_lib.recording: what lpformer_amd.PlannedScorer records a scoring step through -- entry points fetched from the
    library handle, tensors whose address is taken, stream hand-overs; nothing is reported outside the block, and a
    second recording inside one is refused."""
    import torch
    from lpformer_amd import _lib

    class Rec:
        def __init__(self):
            self.names, self.kept, self.waits = [], [], []

        def launch(self, name, fn):
            self.names.append(name)
            return fn

        def keep(self, t):
            self.kept.append(t)

        def wait(self, a, b):
            self.waits.append((a, b))

    class FakeStream:
        def __init__(self):
            self.waited = []

        def wait_stream(self, other):
            self.waited.append(other)

    t = torch.zeros(4)
    a, b = FakeStream(), FakeStream()
    with _lib.recording(Rec()) as rec:
        assert _lib.hip().lpf_abi_version() == _lib.ABI_VERSION
        assert _lib.ptr(t) == t.data_ptr() and _lib.ptr(None) is None
        _lib.stream_wait(a, b)
        with pytest.raises(_lib.LpfError):
            with _lib.recording(Rec()):
                pass
    assert rec.names == ["lpf_abi_version"] and rec.kept == [t] and rec.waits == [(a, b)] and a.waited == [b]
    _lib.hip().lpf_abi_version()
    _lib.ptr(t)
    _lib.stream_wait(a, b)
    assert rec.names == ["lpf_abi_version"] and len(rec.kept) == 1 and len(rec.waits) == 1 and a.waited == [b, b]


def test_short_form_of_the_head_for_pairs_without_selected_nodes():
    """fold.empty_pair_head_bias (the short form the dense tail runs for pairs that select nothing) against
    the oracle's module arithmetic: attention over an empty set = bias, post_att_norm, zero counts, pairwise_lin, the
    concatenation with the elementwise branch and the score head -- logit = w_dot . ReLU(A_e r_e + bC_empty) + b_dot for
    any elementwise hidden activation r_e."""
    fx = Fixture("lp_all_d64")
    P = fx.params
    d = fx.cfg["dim"]
    rng = np.random.default_rng(3)
    n = 40
    # the reference's way: pairwise branch of a pair whose three selected sets are empty
    pre = np.zeros((n, d), np.float32) + P["model.att_layers.0.att.bias"]
    post = O.layer_norm(pre, P["model.att_layers.0.post_att_norm.weight"], P["model.att_layers.0.post_att_norm.bias"])
    pw = O.mlp2(np.concatenate([post, np.zeros((n, 4), np.float32)], axis=1), P, "model.pairwise_lin")
    x_e = rng.standard_normal((n, d)).astype(np.float32)               # any elementwise product x_a * x_b
    h = O.linear(x_e, P["model.elementwise_lin.linears.0.weight"], P["model.elementwise_lin.linears.0.bias"])
    r_e = np.maximum(O.layer_norm(h, P["model.elementwise_lin.norm.weight"], P["model.elementwise_lin.norm.bias"]), 0)
    ew = O.linear(r_e, P["model.elementwise_lin.linears.1.weight"], P["model.elementwise_lin.linears.1.bias"])
    _, want = O.mlp_score(np.concatenate([ew, pw], axis=1).astype(np.float32), P, 2)
    # the folded way (LinkTransformer._score_fold in float64, then the short form)
    f64 = lambda k: P[k].astype(np.float64)   # noqa: E731
    ws0, bs0 = f64("score.lins.0.weight"), f64("score.lins.0.bias")
    a_e = ws0[:, :d] @ f64("model.elementwise_lin.linears.1.weight")
    a_p = ws0[:, d:] @ f64("model.pairwise_lin.linears.1.weight")
    c = bs0 + ws0[:, :d] @ f64("model.elementwise_lin.linears.1.bias") + ws0[:, d:] @ f64("model.pairwise_lin.linears.1.bias")
    a_fold = np.concatenate([a_e, a_p], axis=1)
    bc_empty = fold.empty_pair_head_bias(P["model.att_layers.0.att.bias"], P["model.att_layers.0.post_att_norm.weight"],
                                         P["model.att_layers.0.post_att_norm.bias"],
                                         P["model.pairwise_lin.linears.0.weight"], P["model.pairwise_lin.linears.0.bias"],
                                         P["model.pairwise_lin.norm.weight"], P["model.pairwise_lin.norm.bias"],
                                         a_fold, c, d)
    assert bc_empty.shape == (2 * d,) and bc_empty.dtype == np.float32
    hid = np.maximum(r_e.astype(np.float64) @ a_e.T + bc_empty, 0)
    got = hid @ f64("score.lins.1.weight").reshape(-1) + f64("score.lins.1.bias")[0]
    assert np.abs(got - want).max() < 2e-5


def test_no_flip_radius_is_a_square_without_flips():
    """fold.no_flip_radius: inside [0, c]^2 no hidden unit of the PE MLP leaves the pattern of (0, 0) (checked on random
    points of the square, float64, both argument orders by symmetry of the square); just outside it some unit does, or
    the search stopped at its cap; a unit sitting at zero at the origin gives 0."""
    fx = Fixture("lp_all_d64")
    m_sd, _ = fx.state_dicts()
    sd = {k: torch.from_numpy(v) for k, v in m_sd.items()}
    d = fx.cfg["dim"]
    tabs, _, _, _ = fold.flip_tables(sd, d, 3)
    _, stat = fold.pe_tables(sd, d, 3)
    rng = np.random.default_rng(0)

    def zmin(t, xy):
        st, tb = stat[t].astype(np.float64), tabs[t].astype(np.float64)
        x, y = xy[:, 0], xy[:, 1]
        var = st[0] * x * x + st[1] * y * y + st[2] + 2 * (st[3] * x * y + st[4] * x + st[5] * y)
        r = 1 / np.sqrt(np.maximum(var, 0) + 1e-5)
        return (r[:, None] * (x[:, None] * tb[:, 0] + y[:, None] * tb[:, 1] + tb[:, 2]) + tb[:, 3]).min()
    for t in range(3):
        c = fold.no_flip_radius(tabs[t], stat[t])
        assert 0.0 < c <= 0.5
        assert zmin(t, rng.uniform(0, c, (100_000, 2))) > 0
        if c < 0.45:
            edge = np.concatenate([np.stack([np.full(2000, 1.25 * c), np.linspace(0, 1.25 * c, 2000)], 1),
                                   np.stack([np.linspace(0, 1.25 * c, 2000), np.full(2000, 1.25 * c)], 1)])
            assert zmin(t, edge) < 0
    bad = tabs[0].copy()
    bad[5, 2:] = 0.0           # unit 5: exactly zero at the origin
    assert fold.no_flip_radius(bad, stat[0]) == 0.0


def test_recorded_plan_moves_every_pointer_into_the_ids():
    """PlannedScorer._replay: an argument that points INTO the ids tensor the step was recorded with (the tensor itself,
    its second row) follows the ids of the replay; everything else -- workspaces, sizes, a value that merely lies near --
    stays; stream hand-overs are replayed in place; a non-zero status raises."""
    from lpformer_amd.graphed import PlannedScorer

    ids = torch.zeros(2, 100, dtype=torch.int64)[:, 10:42]          # a [2, 32] window of a longer id list: rows 800 B apart
    base = ids.data_ptr()
    seen = []

    class FakeStream:
        def wait_stream(self, other):
            seen.append(("wait", self, other))
    a, b = FakeStream(), FakeStream()
    ok = lambda *args: (seen.append(args), 0)[1]     # noqa: E731
    bad = lambda *args: 3                              # noqa: E731
    extent = (1 * 100 + 31 * 1 + 1) * 8
    plan = PlannedScorer.__new__(PlannedScorer)
    plan.batch = ids
    calls = [("k1", ok, (32, base, 100, base + 800, 12345)), (None, None, (a, b)),
             ("k2", ok, (base + extent, base - 8, None))]
    plan._plan = [(n, f, args, tuple(i for i, v in enumerate(args)
                                     if f is not None and isinstance(v, int) and base <= v < base + extent))
                  for n, f, args in calls]
    assert plan._plan[0][3] == (1, 3) and plan._plan[2][3] == ()
    plan._replay(base)
    assert seen == [(32, base, 100, base + 800, 12345), ("wait", a, b), (base + extent, base - 8, None)]
    del seen[:]
    plan._replay(base + 4096)
    assert seen[0] == (32, base + 4096, 100, base + 4896, 12345) and seen[2] == (base + extent, base - 8, None)
    plan._plan.append(("k3", bad, (1,), ()))
    with pytest.raises(_lib.LpfError):
        plan._replay(base)


def test_transposed_weight_cache_follows_the_parameter_object_and_its_version():
    """train._transposed keeps W^T across the two passes of a step: valid for the SAME parameter object at the SAME
    version only (a storage address is not an identity: the next model's parameter can live there at the same version --
    the failure this test's rule replaced showed as wrong gradients in whichever model was built second)."""
    import gc
    from lpformer_amd import train
    p = torch.nn.Parameter(torch.randn(3, 5))
    t1 = train._transposed(p)
    assert torch.equal(t1, p.detach().t()) and train._transposed(p) is t1
    with torch.no_grad():
        p.add_(1.0)                                   # an optimiser step: same object, next version
    t2 = train._transposed(p)
    assert t2 is not t1 and torch.equal(t2, p.detach().t())
    # another parameter object at the same address and version count is another parameter
    key, ver = id(p), p._version
    entry = train._wt_cache[key]
    del p
    gc.collect()
    assert entry[0]() is None                         # (the cache does not keep parameters alive)
    q = torch.nn.Parameter(torch.randn(3, 5))
    with torch.no_grad():
        q.add_(0.0)
    train._wt_cache[id(q)] = (entry[0], q._version, t2)      # what a recycled id would find
    assert torch.equal(train._transposed(q), q.detach().t())
    # temporaries are never cached
    n = len(train._wt_cache)
    v = torch.randn(4, 2)
    assert torch.equal(train._transposed(v), v.t()) and len(train._wt_cache) == n
