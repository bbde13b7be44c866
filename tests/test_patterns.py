"""CPU tests of lpformer_amd/patterns.py: the table of activation patterns the pair-major attention kernel reads instead of
looking at an entry's 2 D hidden units (reference: the PE MLPs, src/models/link_transformer.py:67-76,182-211, entering the
key through lin_r, src/modules/layers.py:193-224).  What must hold whatever the weights are:
  * the host's cell of a value is the kernel's (fp32 bit arithmetic) and the cells tile [0, 1];
  * a cell marked clean holds ONE activation pattern -- checked by brute force on points of the cell;
  * the four vectors of a tabulated pattern reproduce the hidden layer's contribution to the key on every point of its cells;
  * the last cell of either axis (values > 1, NaN) is never tabulated, id 0 is the pattern of (0, 0)."""
import numpy as np
import pytest
import torch

from lpformer_amd import fold, patterns


def _state(dim, seed, gain=1.0, beta=0.0):
    g = torch.Generator().manual_seed(seed)
    st = {"att_layers.0.att.lin_r.weight": torch.randn(dim, 2 * dim, generator=g) * 0.2}
    for k in fold.PE_KEYS:
        st[f"{k}.linears.0.weight"] = (torch.rand(dim, 2, generator=g) - 0.5) * gain
        st[f"{k}.linears.0.bias"] = torch.rand(dim, generator=g) - 0.5
        st[f"{k}.norm.weight"] = 1.0 + 0.1 * torch.randn(dim, generator=g)
        st[f"{k}.norm.bias"] = beta * torch.randn(dim, generator=g)
        st[f"{k}.linears.1.weight"] = torch.randn(dim, dim, generator=g) * 0.2
        st[f"{k}.linears.1.bias"] = torch.randn(dim, generator=g) * 0.1
    return st


def _hidden(st, t, x, y):
    """y_k(x, y) of PE MLP t in float64: LayerNorm(W1 [x, y] + b1) -> [n, D] (before the ReLU)."""
    k = fold.PE_KEYS[t]
    w1, b1 = st[f"{k}.linears.0.weight"].double(), st[f"{k}.linears.0.bias"].double()
    g, be = st[f"{k}.norm.weight"].double(), st[f"{k}.norm.bias"].double()
    h = torch.stack([x, y], dim=1) @ w1.T + b1
    h = (h - h.mean(1, keepdim=True)) / torch.sqrt(h.var(1, unbiased=False, keepdim=True) + 1e-5)
    return h * g + be


def _words(bits):
    return patterns._pack(bits)


def test_cells_tile_the_unit_interval_and_match_the_bit_arithmetic():
    for m, oe in ((6, -12), (3, -8), (4, -12)):
        geo = patterns.grid_geometry(m, oe)
        e, n = geo["edges"], geo["n"]
        assert e[0] == 0.0 and np.all(np.diff(e) > 0) and e[n - 2] <= 1.0 < e[n - 1]
        rng = np.random.default_rng(m)
        v = np.concatenate([rng.random(20000), 10.0 ** rng.uniform(-7, 0, 20000), [0.0, 1.0]]).astype(np.float32)
        c = patterns.cell_index(torch.from_numpy(v), geo).numpy()
        assert c.min() == 0 and c.max() == n - 2 and c[-2] == 0 and c[-1] == n - 2
        # v lies in its cell up to the rounding of v + ofs (2^-24 relative to the sum)
        tol = (v.astype(np.float64) + geo["ofs"]) * 2.0 ** -23
        assert np.all(v >= e[c] - tol) and np.all(v < e[c + 1] + tol)
        # out of range: the last cell
        bad = torch.tensor([1.5, float("nan"), -1.0, float("inf")])
        assert patterns.cell_index(bad, geo).tolist() == [n - 1] * 4


@pytest.mark.parametrize("dim,seed,gain,beta", [(32, 1, 1.0, 0.0), (32, 2, 8.0, 0.3), (64, 3, 40.0, 1.0)])
def test_a_clean_cell_holds_one_pattern_and_its_vectors_give_the_key(dim, seed, gain, beta):
    st = _state(dim, seed, gain, beta)
    sample = [(torch.rand(500) * 0.2, torch.rand(500) * 0.2) for _ in range(3)]
    out = patterns.build(st, dim, 3, sample, m=3, ofs_exp=-8)
    geo, n = out["geo"], out["geo"]["n"]
    e = torch.from_numpy(geo["edges"])
    rng = torch.Generator().manual_seed(seed)
    w_rp = st["att_layers.0.att.lin_r.weight"].double()[:, dim:]
    for t in range(3):
        grid = out["grid"][t]
        assert (grid[n - 1, :] == patterns.AMBIGUOUS).all() and (grid[:, n - 1] == patterns.AMBIGUOUS).all()
        # flagged cells name a tabulated pattern too (the one the exact path starts from)
        assert int((grid & 0x7f).max()) < patterns.NPAT
        assert int((grid < patterns.AMBIGUOUS).sum()) > 0
        # id 0 is the pattern of the point (0, 0)
        zero = torch.zeros(1, dtype=torch.float64)
        p0 = _words(_hidden(st, t, zero, zero) > 0)
        ii, jj = torch.nonzero(grid < patterns.AMBIGUOUS, as_tuple=True)
        ids = grid[ii, jj].long()
        # points of every tabulated cell: its corners (pulled inside by a hair), its centre, random interior points
        reps = 6
        u = torch.rand((reps, ii.numel()), generator=rng, dtype=torch.float64)
        w = torch.rand((reps, ii.numel()), generator=rng, dtype=torch.float64)
        u[0], w[0], u[1], w[1], u[2], w[2] = 0.0, 0.0, 1.0, 1.0, 0.0, 1.0
        eps = 1e-9
        x = (e[ii] + (e[ii + 1] - e[ii]) * u.clamp(eps, 1 - eps)).reshape(-1)
        y = (e[jj] + (e[jj + 1] - e[jj]) * w.clamp(eps, 1 - eps)).reshape(-1)
        pid = ids.repeat(reps)
        hid = _hidden(st, t, x, y)
        got = _words(hid > 0)
        # (a) one pattern per cell: the pattern of every point equals the pattern of its cell's first point
        first = got[: ii.numel()].repeat(reps, 1)
        assert (got == first).all()
        if (ids == 0).any():
            k0 = int(torch.nonzero(ids == 0)[0, 0])
            assert (got[k0] == p0[0]).all()
        # (b) the table's vectors reproduce  W_rp W2 ReLU(hidden) + W_rp b2  for the order (x, y) on these points
        k = fold.PE_KEYS[t]
        w2, b2 = st[f"{k}.linears.1.weight"].double(), st[f"{k}.linears.1.bias"].double()
        direct = torch.relu(hid) @ (w_rp @ w2).T + w_rp @ b2
        h_lin = torch.stack([x, y], dim=1) @ st[f"{k}.linears.0.weight"].double().T + st[f"{k}.linears.0.bias"].double()
        r = 1.0 / torch.sqrt(h_lin.var(1, unbiased=False) + 1e-5)
        b = out["base"][t].double()[pid]          # [points, 4, D]
        table = b[:, 0] * (r * x)[:, None] + b[:, 1] * (r * y)[:, None] + b[:, 2] * r[:, None] + b[:, 3]
        scale = max(1.0, float(direct.abs().max()))
        assert float((direct - table).abs().max()) <= 2e-6 * scale
        # (c) flagged cells (a boundary may cross them / pattern not tabulated): the kernel's exact path starts from the
        #     tabulated pattern the cell names and adds Wfold[:, k] |y_k| for every unit whose state differs from it
        fi, fj = torch.nonzero(grid[: n - 1, : n - 1] >= patterns.AMBIGUOUS, as_tuple=True)
        if fi.numel():
            pick = torch.randperm(fi.numel(), generator=rng)[:400]
            fi, fj = fi[pick], fj[pick]
            fx = e[fi] + (e[fi + 1] - e[fi]) * torch.rand(fi.numel(), generator=rng, dtype=torch.float64)
            fy = e[fj] + (e[fj + 1] - e[fj]) * torch.rand(fj.numel(), generator=rng, dtype=torch.float64)
            fid = (grid[fi, fj] & 0x7f).long()
            hid_f = _hidden(st, t, fx, fy)
            sgn = out["sign"][t].long() & 0xffffffff                      # [NPAT, D / 32]
            sh = torch.arange(32)
            differs0 = ((sgn[fid][:, :, None] >> sh) & 1).reshape(fid.numel(), -1)[:, :dim].bool()
            on_ref = (_hidden(st, t, zero, zero) > 0) ^ differs0          # the named pattern's active units
            flipped = (hid_f > 0) ^ on_ref
            hl = torch.stack([fx, fy], dim=1) @ st[f"{k}.linears.0.weight"].double().T + st[f"{k}.linears.0.bias"].double()
            rf = 1.0 / torch.sqrt(hl.var(1, unbiased=False) + 1e-5)
            bf = out["base"][t].double()[fid]
            tab_f = bf[:, 0] * (rf * fx)[:, None] + bf[:, 1] * (rf * fy)[:, None] + bf[:, 2] * rf[:, None] + bf[:, 3]
            corr = (hid_f.abs() * flipped) @ (w_rp @ w2).T
            direct_f = torch.relu(hid_f) @ (w_rp @ w2).T + w_rp @ b2
            assert float((direct_f - (tab_f + corr)).abs().max()) <= 2e-6 * max(1.0, float(direct_f.abs().max()))
            # ... and the named pattern is a near one: fewer differing units than against the pattern of (0, 0) on average
            d_ref = flipped.sum(1).double().mean()
            d_0 = ((hid_f > 0) ^ (_hidden(st, t, zero, zero) > 0)).sum(1).double().mean()
            assert d_ref <= d_0 + 1e-9


def test_the_sample_decides_which_patterns_are_tabulated():
    """Results never depend on the sample -- but the patterns the sample's points see must be among the tabulated ones
    (here: points far from the origin, whose pattern a prior that favours the origin would not pick)."""
    dim = 32
    st = _state(dim, 5, gain=6.0, beta=0.5)
    far = [(0.5 + 0.01 * torch.rand(400), 0.02 + 0.001 * torch.rand(400)) for _ in range(3)]
    a = patterns.build(st, dim, 3, far, m=3, ofs_exp=-8, npat=4)
    b = patterns.build(st, dim, 3, None, m=3, ofs_exp=-8, npat=4)
    geo = a["geo"]
    for t in range(3):
        ia, ib = patterns.cell_index(far[t][0], geo), patterns.cell_index(far[t][1], geo)
        cov_a = float((a["grid"][t][ia, ib] < patterns.AMBIGUOUS).double().mean())
        cov_b = float((b["grid"][t][ia, ib] < patterns.AMBIGUOUS).double().mean())
        assert cov_a >= cov_b
        assert a["stats"][t]["covered"] is not None and a["stats"][t]["sample_points"] == 800
        # with or without the sample a cell never holds a pattern that is not its own
        both = (a["grid"][t] < patterns.AMBIGUOUS) & (b["grid"][t] < patterns.AMBIGUOUS)
        ii, jj = torch.nonzero(both, as_tuple=True)
        if ii.numel():
            va = a["base"][t][a["grid"][t][ii, jj].long()]
            vb = b["base"][t][b["grid"][t][ii, jj].long()]
            assert torch.equal(va, vb)


def test_nearest_pattern_is_the_hamming_nearest():
    g = torch.Generator().manual_seed(7)
    words = torch.randint(0, 1 << 32, (500, 4), generator=g, dtype=torch.int64)
    chosen = torch.randint(0, 1 << 32, (8, 4), generator=g, dtype=torch.int64)
    got = patterns.nearest_pattern(words, chosen, chunk=128)
    bits = lambda x: ((x[..., None] >> torch.arange(32)) & 1).reshape(*x.shape[:-1], -1)
    dist = (bits(words)[:, None, :] != bits(chosen)[None, :, :]).sum(-1)
    assert torch.equal(got, dist.argmin(dim=1))
    assert int(patterns._popcount32(torch.tensor([0, 1, 0xffffffff, 0x80000001])).sum()) == 0 + 1 + 32 + 2


def test_a_unit_that_is_zero_everywhere_does_not_make_every_cell_ambiguous():
    """A pruned hidden unit (LayerNorm gain and bias exactly 0) is inactive at every point: it may not fail the
    clean-cell test in every cell (coverage 0 %); and values just below zero land in the kernel's LAST cell."""
    dim = 32
    st = _state(dim, 5)
    n_clean = []
    for prune in (False, True):
        if prune:
            for k in fold.PE_KEYS:
                st[f"{k}.norm.weight"][3] = 0.0
                st[f"{k}.norm.bias"][3] = 0.0
        pt = patterns.build(st, dim, 3, sample=None, device="cpu", m=3, ofs_exp=-8)
        n_clean.append([s["clean"] for s in pt["stats"]])
        if prune:
            assert all(int(w) & (1 << 3) == 0 for w in pt["sign"][:, :, 0].reshape(-1).tolist())
    assert min(n_clean[1]) > 0.0, "a dead unit flagged the whole grid"
    assert all(b >= a - 1e-12 for a, b in zip(n_clean[0], n_clean[1]))   # (one unit fewer to cross a cell)
    geo = patterns.grid_geometry()
    below = torch.tensor([-2.0 ** -13, -2.0 ** -20])
    assert patterns.cell_index(below, geo).tolist() == [geo["n"] - 1] * 2
