"""Pin the CPU oracle against the golden vectors recorded from the reference itself (CPU only)."""
import numpy as np
import pytest

from oracle import lpformer_oracle as O
from tests.golden_util import LP_CASES, MASKED_CASES, PPR_CASES, Fixture, GOLDEN_DIR

FLOAT_TOL = 2e-5  # abs, fp32 re-association only; observed ~1e-6


def _graph(fx):
    n = fx.n
    adj_norm = O.gcn_norm(fx.edge_index, fx.edge_weight, n)
    mask = O.symmetric_mask_csr(fx.edge_index, n)
    r, c, v = fx.ppr_coo
    ppr = O.csr_from_coo(r, c, v, n)
    return adj_norm, mask, ppr


@pytest.mark.parametrize("case", LP_CASES)
def test_selection_bit_exact(case):
    fx = Fixture(case)
    _, mask, ppr = _graph(fx)
    th = (fx.cfg["thresh_cn"], fx.cfg["thresh_1hop"], fx.cfg["thresh_non1hop"])
    sel = O.select_nodes(fx["batch"], mask, ppr, th, n=fx.n)
    assert sorted(sel) == sorted(fx.sel_tags())
    for tag in fx.sel_tags():
        ix, pa, pb = sel[tag]
        np.testing.assert_array_equal(ix, fx[f"sel_{tag}_ix"])
        # emitted PPR values feed the positional MLP: bit-exact too
        np.testing.assert_array_equal(pa.view(np.uint32), fx[f"sel_{tag}_pa"].view(np.uint32))
        np.testing.assert_array_equal(pb.view(np.uint32), fx[f"sel_{tag}_pb"].view(np.uint32))


@pytest.mark.parametrize("case", LP_CASES)
def test_forward_matches_reference(case):
    fx = Fixture(case)
    adj_norm, mask, ppr = _graph(fx)
    res = O.forward(fx["batch"], fx["x"], adj_norm, mask, ppr, fx.params, fx.cfg, want_parts=True)
    for key in ("x_node", "elementwise_feats", "att_pre_ln", "att_post_ln", "pairwise_feats", "combined_feats",
                "logit", "prob"):
        err = np.abs(res[key] - fx[key]).max()
        assert err <= FLOAT_TOL, f"{case}:{key} max abs err {err}"
    # attention weights (return_weights=True path, src/modules/layers.py:73-75): row 0 = pair position, row 1 = alpha
    aw = fx["att_weights"]
    np.testing.assert_array_equal(aw[0].astype(np.int64), res["ix"][0])
    assert np.abs(aw[1] - res["alpha"]).max() <= FLOAT_TOL


@pytest.mark.parametrize("case", MASKED_CASES)
def test_masked_adjacency_override_matches_reference(case):
    """The training loop's call pattern (src/train/train_model.py:40-59): CN / 1-hop typing from the adjacency with
    the batch's positive edges removed, >1-hop exclusion from the UNMASKED adjacency (link_transformer.py:438-443);
    optional propagation over the masked (unweighted) adjacency."""
    fx = Fixture(case)
    n = fx.n
    adj_norm, mask, ppr = _graph(fx)
    keep = fx["masked_keep_edges"].astype(np.int64)
    masked = O.symmetric_mask_csr(keep, n)
    th = (fx.cfg["thresh_cn"], fx.cfg["thresh_1hop"], fx.cfg["thresh_non1hop"])
    mb = fx["masked_batch"]
    sel = O.select_nodes(mb, masked, ppr, th, n=n, adj_unmasked=mask)
    for tag in ("cn", "onehop", "non1hop"):
        ix, pa, pb = sel[tag]
        np.testing.assert_array_equal(ix, fx[f"masked_sel_{tag}_ix"])
        np.testing.assert_array_equal(pa.view(np.uint32), fx[f"masked_sel_{tag}_pa"].view(np.uint32))
        np.testing.assert_array_equal(pb.view(np.uint32), fx[f"masked_sel_{tag}_pb"].view(np.uint32))
    # the override must matter: the same batch typed with the unmasked adjacency selects a different set
    plain = O.select_nodes(mb, mask, ppr, th, n=n)
    assert plain["onehop"][0].shape != sel["onehop"][0].shape or not np.array_equal(plain["onehop"][0], sel["onehop"][0])
    res = O.forward(mb, fx["x"], adj_norm, masked, ppr, fx.params, fx.cfg, adj_unmasked=mask)
    assert np.abs(res["combined_feats"] - fx["masked_combined_feats"]).max() <= FLOAT_TOL
    assert np.abs(res["logit"] - fx["masked_logit"]).max() <= FLOAT_TOL
    # --mask-input: propagation over the symmetrised, unweighted kept edges
    both = np.concatenate([keep, keep[::-1]], axis=1)
    prop = O.gcn_norm(both, None, n)
    res = O.forward(mb, fx["x"], prop, masked, ppr, fx.params, fx.cfg, adj_unmasked=mask)
    assert np.abs(res["x_node"] - fx["masked_prop_x_node"]).max() <= FLOAT_TOL
    assert np.abs(res["logit"] - fx["masked_prop_logit"]).max() <= FLOAT_TOL


def test_naive_threshold_would_differ():
    """The fp32 `+t-t` round trip is not optional: a plain P >= theta test selects a different 1-hop set."""
    fx = Fixture("lp_all_d64")
    _, mask, ppr = _graph(fx)
    b = fx["batch"].astype(np.int64)
    rp, col, val = ppr
    th = np.float32(fx.cfg["thresh_1hop"])
    n_naive = 0
    got = fx["sel_onehop_ix"]
    for k in range(b.shape[1]):
        na = set(mask[1][mask[0][b[0, k]]:mask[0][b[0, k] + 1]].tolist())
        nb = set(mask[1][mask[0][b[1, k]]:mask[0][b[1, k] + 1]].tolist())
        pa = dict(zip(col[rp[b[0, k]]:rp[b[0, k] + 1]].tolist(), val[rp[b[0, k]]:rp[b[0, k] + 1]].tolist()))
        pb = dict(zip(col[rp[b[1, k]]:rp[b[1, k] + 1]].tolist(), val[rp[b[1, k]]:rp[b[1, k] + 1]].tolist()))
        for v in na ^ nb:
            if np.float32(pa.get(v, 0.0)) >= th and np.float32(pb.get(v, 0.0)) >= th:
                n_naive += 1
    assert n_naive != got.shape[1]


@pytest.mark.parametrize("case", PPR_CASES)
def test_ppr_push_bit_exact(case):
    z = np.load(f"{GOLDEN_DIR}/{case}.npz")
    n = int(z["n"])
    rowptr, col = O.edge_csr(z["edge_index"], n)
    for eps in z["eps_list"]:
        if eps < 5e-4 and n > 250:
            continue
        tag = f"{eps:g}".replace("-", "m").replace(".", "p")
        r, c, v = O.ppr_push(rowptr, col, 0.15, float(eps))
        np.testing.assert_array_equal(r, z[f"row_{tag}"])
        np.testing.assert_array_equal(c, z[f"col_{tag}"])
        np.testing.assert_array_equal(v.view(np.uint32), z[f"val_{tag}"].view(np.uint32))
