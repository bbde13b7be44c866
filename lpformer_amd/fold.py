"""Parameter-only precomputation for the fused attention kernels (float64 on the host, stored as fp32).

The reference evaluates, per selected node e of type t (src/models/link_transformer.py:182-211,
src/modules/layers.py:206-218):

    pe_e = g_t([pa,pb]) + g_t([pb,pa]),   g_t(x) = W2_t ReLU(LN_t(W1_t x + b1_t)) + b2_t
    k_e  = W_r [X[v_e] ; pe_e] + b_r

Writing W_r = [W_rx | W_rp] this is   k_e = Z[v_e] + Wfold_t h_e + bfold_t   with

    Z      = X W_rx^T + b_r                         (once per encoder run, a GEMM)
    h_e    = ReLU(LN_t(W1_t [pa,pb] + b1_t)) + ReLU(LN_t(W1_t [pb,pa] + b1_t))
    Wfold_t = W_rp W2_t ,  bfold_t = W_rp (2 b2_t)

and LN_t of the 2-input first layer has closed-form statistics: with u_k = w0_k x + w1_k y + b_k,
mean_k u = m0 x + m1 y + mb and var_k u is the quadratic form of the centred second moments of (w0, w1, b).
The tables below carry exactly those quantities (layouts documented in include/lpformer_hip.h).
"""
from __future__ import annotations

import numpy as np

PE_KEYS = ("ppr_encoder_cn", "ppr_encoder_onehop", "ppr_encoder_non1hop")


def _f64(t):
    return t.detach().cpu().double().numpy()


def pe_tables(state: dict, dim: int, n_types: int):
    """pe_tab float32[3, D, 4], pe_stat float32[3, 8] from the ``ppr_encoder_*`` MLP parameters."""
    tab = np.zeros((3, dim, 4), np.float64)
    stat = np.zeros((3, 8), np.float64)
    for t in range(n_types):
        k = PE_KEYS[t]
        w1, b1 = _f64(state[f"{k}.linears.0.weight"]), _f64(state[f"{k}.linears.0.bias"])  # [D,2], [D]
        g, be = _f64(state[f"{k}.norm.weight"]), _f64(state[f"{k}.norm.bias"])
        w0c, w1c, bc = w1[:, 0] - w1[:, 0].mean(), w1[:, 1] - w1[:, 1].mean(), b1 - b1.mean()
        tab[t, :, 0], tab[t, :, 1], tab[t, :, 2], tab[t, :, 3] = g * w0c, g * w1c, g * bc, be
        stat[t, :6] = [(w0c * w0c).mean(), (w1c * w1c).mean(), (bc * bc).mean(), (w0c * w1c).mean(),
                       (w0c * bc).mean(), (w1c * bc).mean()]
    return tab.astype(np.float32), stat.astype(np.float32)


def flip_tables(state: dict, dim: int, n_types: int, prefix="att_layers.0.att"):
    """Tables of the activation-pattern attention kernel (csrc/pair_flip.hip).  The hidden layer of a PE MLP is
    y_k(x, y) = r(x, y) (ta_k x + tc_k y + td_k) + beta_k followed by ReLU (``pe_tables``), so for the set S0 of units
    that are active at (x, y) = (0, 0)
        sum_{k in S0} Wfold[:, k] y_k = P0 (r x) + Q0 (r y) + R0 r + B0,
    and summed over both argument orders, with bfold:  P0 (r1 pa + r2 pb) + Q0 (r1 pb + r2 pa) + R0 (r1 + r2) + C0,
    C0 = 2 B0 + bfold.  Returns  tab_signed float32[3, D, 4] = (ta, tc, td, beta) of every unit times +1 (unit in S0)
    or -1 (not in S0) -- the kernel evaluates z_k = (+-) y_k and a unit has left the pattern of (0, 0) exactly when
    z_k < 0, its correction being Wfold[:, k] |y_k| = -Wfold[:, k] z_k --,  base float32[3, 4, D] = (P0, Q0, R0, C0),
    s0 uint32[3, D] (1 = unit in S0; tests),  wfold_t float32[3, D, D] with wfold_t[t, k, c] = Wfold_t[c, k].
    Everything in float64 on the host, stored as fp32."""
    w_r = _f64(state[f"{prefix}.lin_r.weight"])
    w_rp = w_r[:, dim:]
    base = np.zeros((3, 4, dim))
    tabs = np.zeros((3, dim, 4))
    s0 = np.zeros((3, dim), np.uint32)
    wt = np.zeros((3, dim, dim))
    for t in range(n_types):
        k = PE_KEYS[t]
        w1, b1 = _f64(state[f"{k}.linears.0.weight"]), _f64(state[f"{k}.linears.0.bias"])
        g, be = _f64(state[f"{k}.norm.weight"]), _f64(state[f"{k}.norm.bias"])
        w2, b2 = _f64(state[f"{k}.linears.1.weight"]), _f64(state[f"{k}.linears.1.bias"])
        wfold, bfold = w_rp @ w2, w_rp @ (2.0 * b2)
        ta, tc = g * (w1[:, 0] - w1[:, 0].mean()), g * (w1[:, 1] - w1[:, 1].mean())
        bc = b1 - b1.mean()
        td = g * bc
        r0 = 1.0 / np.sqrt((bc * bc).mean() + 1e-5)          # LayerNorm of the hidden pre-activations at (0, 0)
        on = (td * r0 + be) > 0
        s0[t] = on
        tabs[t] = np.stack([ta, tc, td, be], axis=1) * np.where(on, 1.0, -1.0)[:, None]
        ws = wfold[:, on]
        base[t, 0], base[t, 1], base[t, 2] = ws @ ta[on], ws @ tc[on], ws @ td[on]
        base[t, 3] = 2.0 * (ws @ be[on]) + bfold
        wt[t] = wfold.T
    return tabs.astype(np.float32), base.astype(np.float32), s0, np.ascontiguousarray(wt.astype(np.float32))


def no_flip_radius(tab_signed: np.ndarray, stat: np.ndarray, grid: int = 41, c_max: float = 0.5) -> float:
    """Largest c (bisection to 1e-4, then shrunk by 5 %) such that on [0, c]^2 NO hidden unit of one PE MLP leaves the
    pattern of (0, 0): z_k(x, y) = r(x, y) (ta_k x + tc_k y + td_k) + beta_k > 0 for every unit k of ``tab_signed`` [D, 4]
    (``flip_tables``: signed for that pattern) on a ``grid`` x ``grid`` lattice of the square, float64.  The square is
    symmetric, so both argument orders of an entry (pa, pb) with max(pa, pb) <= c are inside it: the pair-major
    attention kernel (csrc/pair_rows.hip) then takes the base vectors as they are and never looks at the entry's units.
    (Between lattice points z moves by O((c / grid)^2) of its curvature -- a unit that dips below zero there does so by
    ~1e-7 and owes a correction of that size, far below the fp32 noise of the sums it would join.)  0 when some unit is
    already at or below zero at the origin."""
    tab, st = np.asarray(tab_signed, np.float64), np.asarray(stat, np.float64)

    def ok(c):
        g = np.linspace(0.0, c, grid)
        xx, yy = (a.ravel() for a in np.meshgrid(g, g, indexing="ij"))
        var = st[0] * xx * xx + st[1] * yy * yy + st[2] + 2.0 * (st[3] * xx * yy + st[4] * xx + st[5] * yy)
        r = 1.0 / np.sqrt(np.maximum(var, 0.0) + 1e-5)
        z = r[:, None] * (xx[:, None] * tab[:, 0] + yy[:, None] * tab[:, 1] + tab[:, 2]) + tab[:, 3]
        return bool(z.min() > 0.0)
    if not ok(0.0):
        return 0.0
    lo, hi = 0.0, float(c_max)
    if ok(hi):
        return 0.95 * hi
    while hi - lo > 1e-4:
        mid = 0.5 * (lo + hi)
        lo, hi = (mid, hi) if ok(mid) else (lo, mid)
    return 0.95 * lo


def fold_attention(state: dict, dim: int, n_types: int, prefix="att_layers.0.att"):
    """Returns dict of fp32 arrays: w_rx [D,D], b_r [D], wfold [3,D,D], bfold [3,D], wfold_packed (MFMA A-operand
    order), wcat [D, 3D+4] = [Wfold_0 | Wfold_1 | Wfold_2 | bfold_0 bfold_1 bfold_2 | att bias], w_l, b_l2 = 2 b_l."""
    w_r, b_r = _f64(state[f"{prefix}.lin_r.weight"]), _f64(state[f"{prefix}.lin_r.bias"])
    w_l, b_l = _f64(state[f"{prefix}.lin_l.weight"]), _f64(state[f"{prefix}.lin_l.bias"])
    att, bias = _f64(state[f"{prefix}.att"]).reshape(-1), _f64(state[f"{prefix}.bias"])
    assert w_r.shape == (dim, 2 * dim) and att.size == dim, "HIP path supports num_heads=1, one attention layer"
    w_rx, w_rp = w_r[:, :dim], w_r[:, dim:]
    wfold = np.zeros((3, dim, dim))
    bfold = np.zeros((3, dim))
    for t in range(n_types):
        k = PE_KEYS[t]
        w2, b2 = _f64(state[f"{k}.linears.1.weight"]), _f64(state[f"{k}.linears.1.bias"])
        wfold[t] = w_rp @ w2
        bfold[t] = w_rp @ (2.0 * b2)
    wcat = np.concatenate([wfold[0], wfold[1], wfold[2], bfold.T, bias[:, None]], axis=1)  # [D, 3D+4]
    return {
        "w_rx": w_rx.astype(np.float32), "b_r": b_r.astype(np.float32),
        "w_l": w_l.astype(np.float32), "b_l": b_l.astype(np.float32), "b_l2": (2.0 * b_l).astype(np.float32),
        # both node-level projections as ONE [2D, D] product (Z | Y side by side: the node table is read once)
        "w_zy": np.concatenate([w_rx, w_l]).astype(np.float32), "b_zy": np.concatenate([b_r, b_l]).astype(np.float32),
        "att": att.astype(np.float32), "wfold": wfold.astype(np.float32), "bfold": bfold.astype(np.float32),
        "wfold_packed": pack_wfold(wfold.astype(np.float32)),
        "wfold_packed_bf16": pack_wfold_bf16(wfold.astype(np.float32)).view(np.int16), "wcat": np.ascontiguousarray(wcat.astype(np.float32)),
    }


def pack_wfold(wfold: np.ndarray) -> np.ndarray:
    """[3, D, D] (out, in) -> [3, D/32, D/8, 64, 4]: element (t, c, sq, lane, u) =
    wfold[t, 32c + (lane & 31), (lane >> 5) * (D/2) + 4 sq + u]   (A operand of v_mfma_f32_32x32x2_f32, lane
    (row, half) consuming k = half*D/2 + step)."""
    _, d, _ = wfold.shape
    nt, nsq = d // 32, d // 8
    w = wfold.reshape(3, nt, 32, 2, nsq, 4)          # t, c, row, half, sq, u
    w = w.transpose(0, 1, 4, 3, 2, 5)                 # t, c, sq, half, row, u
    return np.ascontiguousarray(w.reshape(3, nt, nsq, 64, 4))


def to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 bit patterns (uint16), round to nearest even."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def pack_wfold_bf16(wfold: np.ndarray) -> np.ndarray:
    """[3, D, D] (out, in) -> uint16 [3, D/32, D/16, 64, 8]: element (t, c, s, lane, j) =
    bf16(wfold[t, 32c + (lane & 31), 16s + 8 (lane >> 5) + j])   (B operand of v_mfma_f32_32x32x16_bf16)."""
    _, d, _ = wfold.shape
    nt, ns = d // 32, d // 16
    w = wfold.reshape(3, nt, 32, ns, 2, 8)            # t, c, col, s, half, j
    w = w.transpose(0, 1, 3, 4, 2, 5)                  # t, c, s, half, col, j
    return to_bf16_bits(np.ascontiguousarray(w.reshape(3, nt, ns, 64, 8)))


def _pad_to(a: np.ndarray, n: int) -> np.ndarray:
    out = np.zeros(n, np.float32)
    out[:a.size] = a.reshape(-1)
    return out


def dense_stage_groups(nt1: int, nt2: int) -> int:
    """k-groups (16 input features each) per pipeline stage of lpf_dense_chain_f32 for a (nt1, nt2)-tile chain
    (mirrors dc_groups in csrc/dense_chain.hip: one, see there)."""
    return 1


def pack_dense(w: np.ndarray, g: int, k_groups: int = 0) -> np.ndarray:
    """[N, K] -> lpf_dense_chain_f32 weight image (either layer).  Output tiles are padded to an even count; k-group
    ks: float4 (c, lane = 16q + i) = W[16c + i][16ks + 4q + 0..3]; a stage = ``g`` consecutive k-groups, zero padded
    to a multiple of 512 float4.  ``k_groups``: number of k-groups to lay out when larger than ceil(K / 16) (layer 2
    runs over the padded hidden tiles of layer 1)."""
    n, k = w.shape
    nt = ((n + 15) // 16 + 1) & ~1
    ng = max((k + 15) // 16, k_groups)
    ns = (ng + g - 1) // g
    wp = np.zeros((nt * 16, ns * g * 16), np.float32)
    wp[:n, :k] = w
    r = wp.reshape(nt, 16, ns, g, 4, 4)           # c, i, stage, sq, q, u
    r = np.ascontiguousarray(r.transpose(2, 3, 0, 4, 1, 5)).reshape(ns, g * nt * 64 * 4)
    per_stage = -(-(g * nt * 64) // 512) * 512 * 4
    out = np.zeros((ns, per_stage), np.float32)
    out[:, :r.shape[1]] = r
    return out.reshape(-1)


def dense_chain_tables(w1, b1, ln_g=None, ln_b=None, w2=None, b2=None) -> dict:
    """fp32 arrays for one lpf_dense_chain_f32 call (vectors zero padded to an even number of 16-feature tiles)."""
    w1 = np.asarray(w1, np.float32)
    n1 = w1.shape[0]
    nt1 = (n1 + 15) // 16
    p1 = ((nt1 + 1) & ~1) * 16
    dot = w2 is not None and np.asarray(w2).shape[0] == 1
    nt2 = 0 if (w2 is None or dot) else (np.asarray(w2).shape[0] + 15) // 16
    g = dense_stage_groups(nt1, nt2)
    out = {"w1p": pack_dense(w1, g), "b1": _pad_to(np.asarray(b1, np.float32), p1)}
    if ln_g is not None:
        out["ln_g"], out["ln_b"] = _pad_to(np.asarray(ln_g, np.float32), p1), _pad_to(np.asarray(ln_b, np.float32), p1)
    if w2 is not None:
        w2 = np.asarray(w2, np.float32)
        if dot:
            out["w2p"], out["b2"] = _pad_to(w2, p1), np.asarray(b2, np.float32).reshape(-1)[:1].copy()
        else:
            out["w2p"] = pack_dense(w2, g, k_groups=p1 // 16)
            out["b2"] = _pad_to(np.asarray(b2, np.float32), ((nt2 + 1) & ~1) * 16)
    return out


def empty_pair_head_bias(att_bias, lnA_g, lnA_b, w_p0, b_p0, lnB_g, lnB_b, a_fold, c_fold, dim: int) -> np.ndarray:
    """Bias of the folded score head for a pair WITHOUT selected nodes, float64 on the host: its attention output is
    post_att_norm(0 / (0 + 1e-16) + bias) (layers.py:78,220), its count features are zero
    (link_transformer.py:340-356), so its whole pairwise branch is a constant vector
        r_p0 = ReLU(LN_B(W_p0 [LN_A(att_bias) | 0] + b_p0))
    and the head sees  A_e r_e + (c + A_p r_p0).  Returns c + A_p r_p0 (fp32, [2D])."""
    f64 = lambda t: np.asarray(t, np.float64)  # noqa: E731

    def ln(x, g, b):
        mu = x.mean()
        return (x - mu) / np.sqrt(((x - mu) ** 2).mean() + 1e-5) * f64(g)[:x.size] + f64(b)[:x.size]

    w_p0 = f64(w_p0)
    pd = w_p0.shape[0]
    feats = np.zeros(w_p0.shape[1])
    feats[:dim] = ln(f64(att_bias), lnA_g, lnA_b)
    r_p0 = np.maximum(ln(w_p0 @ feats + f64(b_p0), lnB_g, lnB_b), 0.0)
    return (f64(c_fold) + f64(a_fold)[:, dim:dim + pd] @ r_p0).astype(np.float32)


def empty_pair_row(att_bias, lnA_g, lnA_b, dim: int) -> np.ndarray:
    """The attention branch's output row of a pair WITHOUT selected nodes, post_att_norm(0 / (0 + 1e-16) + bias)
    (layers.py:78,220), float64 on the host -> fp32 [D]: with an order for the tail the pair-major attention kernel does
    not write such a pair's row at all, the tail takes this one."""
    x = np.asarray(att_bias, np.float64)[:dim]
    mu = x.mean()
    y = (x - mu) / np.sqrt(((x - mu) ** 2).mean() + 1e-5) * np.asarray(lnA_g, np.float64)[:dim] + \
        np.asarray(lnA_b, np.float64)[:dim]
    return y.astype(np.float32)


def tail_chain_tables(wcat, lnA_g, lnA_b, w_p0, b_p0, lnB_g, lnB_b, a_fold, c_fold, w_dot, b_dot, dim: int,
                      att_bias=None) -> dict:
    """fp32 arrays of one lpf_tail_chain_f32 call.  ``a_fold`` [2D, D + pd] = [A_e | A_p] (LinkTransformer._score_fold);
    its r_p columns are moved behind the D r_e columns on an even-tile boundary, as the kernel walks them.
    ``att_bias``: adds ``bC_empty`` (``empty_pair_head_bias``)."""
    wcat, w_p0, a_fold = (np.asarray(t, np.float32) for t in (wcat, w_p0, a_fold))
    pd = w_p0.shape[0]
    ntpb = ((pd + 15) // 16 + 1) & ~1
    wc = np.zeros((a_fold.shape[0], dim + 16 * ntpb), np.float32)
    wc[:, :dim] = a_fold[:, :dim]
    wc[:, dim:dim + pd] = a_fold[:, dim:dim + pd]
    return {
        "wA": pack_dense(wcat, 1), "lnA_g": _pad_to(np.asarray(lnA_g, np.float32), dim),
        "lnA_b": _pad_to(np.asarray(lnA_b, np.float32), dim),
        "wB": pack_dense(w_p0, 1, k_groups=dim // 16 + 1), "bB": _pad_to(np.asarray(b_p0, np.float32), 16 * ntpb),
        "lnB_g": _pad_to(np.asarray(lnB_g, np.float32), 16 * ntpb),
        "lnB_b": _pad_to(np.asarray(lnB_b, np.float32), 16 * ntpb),
        "wC": pack_dense(wc, 1), "bC": np.asarray(c_fold, np.float32).reshape(-1).copy(),
        "w_dot": np.asarray(w_dot, np.float32).reshape(-1).copy(),
        "b_dot": np.asarray(b_dot, np.float32).reshape(-1)[:1].copy(),
        **({} if att_bias is None else {"bC_empty": empty_pair_head_bias(att_bias, lnA_g, lnA_b, w_p0, b_p0, lnB_g, lnB_b,
                                                                       a_fold, c_fold, dim),
                                        "row_empty": empty_pair_row(att_bias, lnA_g, lnA_b, dim)}),
    }
