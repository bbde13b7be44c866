"""Activation-pattern tables of the pair-major attention kernel (csrc/pair_rows.hip, PT form).

Reference: the positional encoding of a selected entry, ``pe = g_t([pa, pb]) + g_t([pb, pa])`` with
``g_t(x) = W2 ReLU(LayerNorm(W1 x + b1)) + b2`` (src/models/link_transformer.py:67-76,182-211), enters the key as
``W_rp pe`` (src/modules/layers.py:193-224).  With the hidden layer in closed form (``fold.flip_tables``)
    y_k(x, y) = r(x, y) (ta_k x + tc_k y + td_k) + beta_k,   r = 1 / s,   s(x, y) = sqrt(var(x, y) + 1e-5),
the set S(x, y) = {k : y_k > 0} of active units is piecewise constant on the plane of PPR value pairs and on a fixed
set S the sum over k in S of Wfold[:, k] y_k is  P_S (r x) + Q_S (r y) + R_S r + B_S.  ``fold.flip_tables`` keeps
these four vectors for ONE pattern -- the one of (0, 0) -- and the kernel pays per entry for finding and correcting the
units that left it: 0.85 units per entry on random-init weights, 4 after 150 training steps (no entry is inside the
no-flip square then).  But the selected entries of a batch only ever see a few dozen patterns -- 16 of them cover
99.97 % of the ordered points on both sets of weights (tools/pattern_census.py) -- so this module tabulates

* ``grid``  uint8 [3, n, n]: the plane cut into n x n cells, logarithmic in both coordinates (cell of a value v: the
  upper bits of the fp32 number v + 2^-12, ``cell_index``); a cell holds the id of the pattern that is PROVABLY the
  pattern of every point of the cell, or -- flag ``AMBIGUOUS`` set: a boundary may cross it / its pattern is not among
  the tabulated ones -- the id of the tabulated pattern NEAREST to the pattern of its centre (fewest differing units):
  such entries take the kernel's exact path, which finds and corrects the units that differ from THAT pattern (one or
  two instead of the many that differ from the pattern of (0, 0) far from the origin);
* ``sign``  int32 [3, NPAT, D / 32]: bit k of pattern s = unit k's state differs from its state in pattern 0 (the exact
  path's hidden-layer table is signed for pattern 0);
* ``base``  float32 [3, NPAT, 4, D]: (P_S, Q_S, R_S, B_S + bfold / 2) of the tabulated patterns, id 0 = the pattern
  of (0, 0).  An entry in clean cells adds the four vectors of the pattern of (pa, pb) and those of (pb, pa).

The proof per cell and unit uses  sign(y_k) = sign(f_k),  f_k = ta_k x + tc_k y + td_k + beta_k s(x, y),  and that s is
CONVEX (the Euclidean norm of an affine map of (x, y)): for beta_k >= 0 f_k is convex -- its maximum over a rectangle
is attained at a corner, its minimum is bounded below by the tangent plane at the centre --, for beta_k < 0 concave,
the other way round.  A cell is clean when every unit has min f > 0 or max f < 0 by these bounds (float64).

Which patterns are tabulated is decided by a sample of (pa, pb) values of the model's own graph (``LinkTransformer.
_entry_sample``: the selection run on 4,096 sample pairs); results never depend on the choice, only the share of
entries that take the slower path does."""
import numpy as np
import torch

from . import fold

GRID_M = 6            # cells per octave = 2^GRID_M
GRID_OFS_EXP = -12    # v + 2^GRID_OFS_EXP: values below ~2^-12 share the first cells (linear there, logarithmic above)
NPAT = 16             # pattern slots per type in ``base`` / ``sign``; D >= 256: the first half is used (the kernel's LDS)
AMBIGUOUS = 0x80


def grid_geometry(m: int = GRID_M, ofs_exp: int = GRID_OFS_EXP) -> dict:
    """Cells of one axis.  cell(v) = clamp((bits(fp32(v + ofs)) >> shift) - base, 0, n - 1); cell n - 2 holds v = 1, cell
    n - 1 everything above (and NaNs, negative numbers): always AMBIGUOUS.  ``edges`` float64 [n + 1]: cell j covers
    [edges[j], edges[j + 1]) up to the rounding of v + ofs (``classify`` widens every cell by more than that)."""
    ofs = np.float32(2.0 ** ofs_exp)
    shift = 23 - m
    base = int(ofs.view(np.uint32)) >> shift
    n = (int((np.float32(1.0) + ofs).view(np.uint32)) >> shift) - base + 2
    bits = ((np.arange(n + 1, dtype=np.uint64) + np.uint64(base)) << np.uint64(shift)).astype(np.uint32)
    edges = bits.view(np.float32).astype(np.float64) - float(ofs)
    edges[0] = 0.0
    return {"m": m, "ofs": float(ofs), "shift": shift, "base": base, "n": n, "edges": edges}


def cell_index(v: torch.Tensor, geo: dict) -> torch.Tensor:
    """The kernel's cell of every value (fp32 arithmetic, bit for bit: csrc/pair_rows.hip ``grid_cell`` -- a value below
    the grid, negative or NaN lands in the LAST cell, which is never tabulated)."""
    u = (v.to(torch.float32) + geo["ofs"]).view(torch.int32).to(torch.int64) & 0xffffffff
    c = (u >> geo["shift"]) - geo["base"]
    last = geo["n"] - 1
    return torch.where(c < 0, torch.full_like(c, last), c.clamp_max(last))


def _f_and_grad(x, y, ta, tc, td, be, st):
    """f_k and its gradient at the points (x, y): [C] -> three [C, D]."""
    var = st[0] * x * x + st[1] * y * y + st[2] + 2.0 * (st[3] * x * y + st[4] * x + st[5] * y)
    s = torch.sqrt(var.clamp_min(0.0) + 1e-5)
    sx, sy = (st[0] * x + st[3] * y + st[4]) / s, (st[1] * y + st[3] * x + st[5]) / s
    f = x[:, None] * ta + y[:, None] * tc + td + be * s[:, None]
    return f, ta + be * sx[:, None], tc + be * sy[:, None]


def _pack(bits: torch.Tensor) -> torch.Tensor:
    """bool [C, D] -> int64 [C, ceil(D / 32)] (32 units per word)."""
    c, d = bits.shape
    w = (d + 31) // 32
    if w * 32 != d:
        bits = torch.cat([bits, torch.zeros(c, w * 32 - d, dtype=torch.bool, device=bits.device)], dim=1)
    sh = torch.arange(32, device=bits.device, dtype=torch.int64)
    return (bits.view(c, w, 32).to(torch.int64) << sh).sum(-1)


def _unique_rows(rows: torch.Tensor):
    """``torch.unique(rows, dim=0, return_inverse=True)`` through a 64-bit hash of every row (the row-wise unique sorts
    with a lexicographic comparator: ~100 s for 6e5 rows of 8 words on the device, the hashed one milliseconds); the hash
    is checked -- every row must equal the representative of its key -- and the slow form is the fallback."""
    mult = torch.tensor([0x9E3779B97F4A7C15 - (1 << 64), 0xC2B2AE3D27D4EB4F - (1 << 64), 0x165667B19E3779F9,
                         0x27D4EB2F165667C5, 0x85EBCA77C2B2AE63 - (1 << 64), 0x2545F4914F6CDD1D, 0x5851F42D4C957F2D,
                         0x14057B7EF767814F], dtype=torch.int64, device=rows.device)
    w = rows.shape[1]
    mult = mult.repeat((w + 7) // 8)[:w] + 2 * torch.arange(w, device=rows.device, dtype=torch.int64)
    key = ((rows ^ (rows >> 29)) * mult).sum(dim=1)
    ukey, inv = torch.unique(key, return_inverse=True)
    first = torch.full((ukey.numel(),), rows.shape[0], dtype=torch.int64, device=rows.device)
    first.scatter_reduce_(0, inv, torch.arange(rows.shape[0], device=rows.device), reduce="amin")
    uniq = rows[first]
    if not bool((uniq[inv] == rows).all()):      # (a collision of the hash)
        return torch.unique(rows, dim=0, return_inverse=True)
    return uniq, inv


def _popcount32(x: torch.Tensor) -> torch.Tensor:
    """Set bits of the low 32 bits of every int64 element."""
    x = x - ((x >> 1) & 0x55555555)
    x = (x & 0x33333333) + ((x >> 2) & 0x33333333)
    x = (x + (x >> 4)) & 0x0F0F0F0F
    return ((x * 0x01010101) >> 24) & 0xFF


def nearest_pattern(words: torch.Tensor, chosen: torch.Tensor, chunk: int = 1 << 16) -> torch.Tensor:
    """For every row of ``words`` [C, W] the index of the row of ``chosen`` [S, W] with the fewest differing bits (ties:
    the lowest index) -- int64 [C]."""
    out = torch.empty(words.shape[0], dtype=torch.int64, device=words.device)
    for a in range(0, words.shape[0], chunk):
        w = words[a:a + chunk]
        dist = _popcount32(w[:, None, :] ^ chosen[None, :, :]).sum(-1)
        out[a:a + chunk] = torch.argmin(dist, dim=1)
    return out


def classify(tab: torch.Tensor, st: torch.Tensor, geo: dict, chunk: int = 1 << 15):
    """tab float64 [D, 4] = (ta, tc, td, beta) of one PE MLP (unsigned), st float64 [6] (``fold.pe_tables``).  Returns
    (words int64 [n * n, W]: the pattern at every cell's centre, clean bool [n * n]); cell (i, j) = x in cell i, y in
    cell j, index i * n + j."""
    dev, n = tab.device, geo["n"]
    e = torch.from_numpy(geo["edges"]).to(dev)
    # every cell widened by 2^-21 of its upper edge (+ ofs): more than the rounding of v + ofs in fp32 can move a value
    pad = (e[1:] + geo["ofs"]) * 2.0 ** -21
    lo, hi = (e[:-1] - pad).clamp_min(0.0), e[1:] + pad
    ta, tc, td, be = tab[:, 0], tab[:, 1], tab[:, 2], tab[:, 3]
    convex = be >= 0
    # a unit that is zero identically (LayerNorm gain and bias exactly 0: a pruned unit) is inactive everywhere and owes
    # no correction in any cell: it must not make every cell of the grid ambiguous
    dead = (ta.abs() + tc.abs() + td.abs() + be.abs()) == 0
    words = torch.empty((n * n, (tab.shape[0] + 31) // 32), dtype=torch.int64, device=dev)
    clean = torch.empty(n * n, dtype=torch.bool, device=dev)
    for a in range(0, n * n, chunk):
        idx = torch.arange(a, min(a + chunk, n * n), device=dev)
        i, j = idx // n, idx % n
        x0, x1, y0, y1 = lo[i], hi[i], lo[j], hi[j]
        cmin = cmax = None
        for xx, yy in ((x0, y0), (x0, y1), (x1, y0), (x1, y1)):
            f = _f_and_grad(xx, yy, ta, tc, td, be, st)[0]
            cmin = f if cmin is None else torch.minimum(cmin, f)
            cmax = f if cmax is None else torch.maximum(cmax, f)
        fc, gx, gy = _f_and_grad(0.5 * (x0 + x1), 0.5 * (y0 + y1), ta, tc, td, be, st)
        spread = gx.abs() * (0.5 * (x1 - x0))[:, None] + gy.abs() * (0.5 * (y1 - y0))[:, None]
        fmin = torch.where(convex, fc - spread, cmin)
        fmax = torch.where(convex, cmax, fc + spread)
        clean[idx] = ((fmin > 0) | (fmax < 0) | dead).all(dim=1)
        words[idx] = _pack(fc > 0)
    # the open-ended last cell of either axis is never clean
    g = clean.view(n, n)
    g[n - 1, :] = False
    g[:, n - 1] = False
    return words, clean


def build(state: dict, dim: int, n_types: int, sample=None, device=None, m: int = GRID_M, ofs_exp: int = GRID_OFS_EXP,
          npat: int = NPAT) -> dict:
    """``grid`` uint8 [3, n, n], ``base`` float32 [3, npat, 4, D], ``geo`` and per-type statistics (``n_patterns``: the
    distinct patterns of clean cells, ``clean``: share of clean cells, ``covered``: share of the sample's ordered points
    that lie in a tabulated cell).  ``sample``: per type (pa, pb) tensors or None."""
    device = torch.device(device if device is not None else "cpu")
    geo = grid_geometry(m, ofs_exp)
    n = geo["n"]
    grid = torch.full((3, n * n), AMBIGUOUS, dtype=torch.uint8, device=device)
    npat_k = npat // 2 if dim >= 256 else npat       # what the kernel's LDS holds (csrc/pair_rows.hip pr_patterns)
    base = np.zeros((3, npat, 4, dim), np.float64)
    sign = np.zeros((3, npat, (dim + 31) // 32), np.int64)
    stats = []
    # parameter-only algebra in float64 numpy on the host (as fold.py; small), the per-cell work on ``device``
    f64 = lambda name: state[name].detach().cpu().double().numpy()
    w_rp = f64("att_layers.0.att.lin_r.weight")[:, dim:]
    ii = torch.arange(n, device=device, dtype=torch.float64)
    prior = (1e-6 / ((ii[:, None] + 1.0) * (ii[None, :] + 1.0))).reshape(-1)   # (no sample: cells near the origin first)
    for t in range(n_types):
        k = fold.PE_KEYS[t]
        w1, b1 = f64(f"{k}.linears.0.weight"), f64(f"{k}.linears.0.bias")
        g, be = f64(f"{k}.norm.weight"), f64(f"{k}.norm.bias")
        w2, b2 = f64(f"{k}.linears.1.weight"), f64(f"{k}.linears.1.bias")
        wfold, bfold = w_rp @ w2, w_rp @ (2.0 * b2)
        w0c, w1c, bc = w1[:, 0] - w1[:, 0].mean(), w1[:, 1] - w1[:, 1].mean(), b1 - b1.mean()
        st_h = np.array([(w0c * w0c).mean(), (w1c * w1c).mean(), (bc * bc).mean(), (w0c * w1c).mean(),
                         (w0c * bc).mean(), (w1c * bc).mean()])
        tab_h = np.stack([g * w0c, g * w1c, g * bc, be], axis=1)
        tab, st = torch.from_numpy(tab_h).to(device), torch.from_numpy(st_h).to(device)
        words, clean = classify(tab, st, geo)
        weight = prior.clone()
        n_sample, cells = 0, None
        if sample is not None and sample[t] is not None and sample[t][0].numel() > 0:
            ia, ib = cell_index(sample[t][0].to(device), geo), cell_index(sample[t][1].to(device), geo)
            cells = torch.cat([ia * n + ib, ib * n + ia])
            weight += torch.bincount(cells, minlength=n * n).to(torch.float64)
            n_sample = int(cells.numel())
        # pattern 0: the one of the point (0, 0) itself (``fold.flip_tables``: r0 td + beta > 0)
        r0 = 1.0 / np.sqrt((bc * bc).mean() + 1e-5)
        s0 = _pack(torch.from_numpy((tab_h[:, 2] * r0 + be) > 0)[None, :].to(device))
        uniq, inv = _unique_rows(torch.cat([s0, words[clean]]))
        # (fp32 sums: they only rank the patterns, and an fp64 index_add_ is a compare-and-swap loop per element on the
        #  device -- most cells share a few patterns: ~30 s per type at D = 256)
        wsum = torch.zeros(uniq.shape[0], dtype=torch.float32, device=device)
        wsum.index_add_(0, inv[1:], weight[clean].to(torch.float32))
        wsum[inv[0]] = float("inf")                       # id 0
        order = torch.argsort(wsum, descending=True)[:npat_k]
        ident = torch.full((uniq.shape[0],), AMBIGUOUS, dtype=torch.int64, device=device)
        ident[order] = torch.arange(order.numel(), device=device)
        # every cell: the nearest tabulated pattern, flagged -- then the clean cells of tabulated patterns: their own, unflagged
        gt = grid[t]
        gt[:] = (nearest_pattern(words, uniq[order]) | AMBIGUOUS).to(torch.uint8)
        own = ident[inv[1:]]
        cidx = torch.nonzero(clean)[:, 0]
        keep = own < AMBIGUOUS
        gt[cidx[keep]] = own[keep].to(torch.uint8)
        g2 = gt.view(n, n)                                 # (out of range on either axis: the exact path against pattern 0)
        g2[n - 1, :] = AMBIGUOUS
        g2[:, n - 1] = AMBIGUOUS
        chosen = uniq[order].cpu().numpy()                 # [<= npat, W] int64
        sign[t, :chosen.shape[0]] = (chosen ^ chosen[0:1]).astype(np.int64)
        for s in range(chosen.shape[0]):
            on = ((chosen[s][:, None] >> np.arange(32, dtype=np.int64)) & 1).reshape(-1)[:dim].astype(np.float64)
            ws = wfold * on[None, :]
            base[t, s, 0], base[t, s, 1], base[t, s, 2] = ws @ tab_h[:, 0], ws @ tab_h[:, 1], ws @ tab_h[:, 2]
            base[t, s, 3] = ws @ be + 0.5 * bfold
        covered = float((gt[cells] < AMBIGUOUS).double().mean()) if n_sample else None
        stats.append({"n_patterns": int(uniq.shape[0]), "clean": float(clean.double().mean()), "covered": covered,
                      "sample_points": n_sample})
    base = torch.from_numpy(base.astype(np.float32)).to(device)
    sign_t = torch.from_numpy((sign & 0xffffffff).astype(np.uint32).view(np.int32)).to(device)
    return {"grid": grid.view(3, n, n).contiguous(), "base": base.contiguous(), "sign": sign_t.contiguous(), "geo": geo,
            "stats": stats}
