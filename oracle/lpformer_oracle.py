"""CPU oracle for LPFormer's link-scoring forward pass.  TEST INFRASTRUCTURE ONLY.

A numpy (fp32) restatement of the reference algorithm, function by function, each citing the
reference file:line it follows (paths relative to /root/reference/).  It is the checker for the
HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import it.  The product (``lpformer_amd``) never does.

Pinning: the reference ships no tests or golden vectors, so this oracle is pinned against outputs
of the reference ITSELF, produced in the build container by ``tests/golden/make_golden.py``
(reference Python unmodified; third-party packages absent from the image replaced by
``oracle/ref_shims.py``) and committed under ``tests/golden/*.npz``.  ``tests/test_oracle_golden.py``
checks: selected index sets + emitted PPR values bit-exact, floats <= 2e-5 abs (observed ~1e-6).
At the third-party boundary (torch_geometric 2.2.0 / torch_sparse / torch_scatter semantics) parity
is unpinned by the reference; rows a4/a11 of SURVEY.md section 8 are the spec.

Graph containers here are plain CSR triples (rowptr int64, col int64, val) with sorted columns.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
LN_EPS = F32(1e-5)


# =============================================================================== containers
def csr_from_coo(row, col, val, n_rows, *, sum_duplicates=True):
    """Sorted, duplicate-free CSR from COO triplets (duplicates summed, like torch ``coalesce``)."""
    row = np.asarray(row, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    n_cols = int(n_rows)
    key = row * n_cols + col
    if val is None:
        ukey = np.unique(key)
        uval = None
    else:
        val = np.asarray(val)
        ukey, inv = np.unique(key, return_inverse=True)
        if sum_duplicates:
            uval = np.zeros(ukey.size, dtype=val.dtype)
            np.add.at(uval, inv, val)
        else:
            uval = np.zeros(ukey.size, dtype=val.dtype)
            uval[inv] = val
    urow = ukey // n_cols
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.add.at(rowptr, urow + 1, 1)
    np.cumsum(rowptr, out=rowptr)
    return rowptr, (ukey % n_cols).astype(np.int64), uval


def symmetric_mask_csr(edge_index, n):
    """0/1 symmetric adjacency pattern: ``adj_t.to_symmetric()...coalesce().bool().int()``
    (src/util/read_datasets.py:88-95).  Returns (rowptr, col)."""
    ei = np.asarray(edge_index, dtype=np.int64)
    row = np.concatenate([ei[0], ei[1]])
    col = np.concatenate([ei[1], ei[0]])
    rowptr, c, _ = csr_from_coo(row, col, None, n)
    return rowptr, c


def pattern_csr(edge_index, n):
    """0/1 pattern of a (possibly multi-)edge list WITHOUT symmetrising (read_datasets.py:109-110,234-235)."""
    ei = np.asarray(edge_index, dtype=np.int64)
    rowptr, c, _ = csr_from_coo(ei[0], ei[1], None, n)
    return rowptr, c


# =============================================================================== a4: GCN
def gcn_norm(edge_index, edge_weight, n):
    """torch_geometric 2.2.0 ``gcn_norm`` on a SparseTensor as called by ``GCNConv`` from
    src/models/other_models.py:35,66 (normalize=True, add_self_loops=True): missing values -> 1; the diagonal is
    SET to 1 (existing self-loops replaced); deg_i = sum_j w_ij; w_ij <- d_i^-1/2 * w_ij * d_j^-1/2, inf -> 0.
    The SparseTensor keeps duplicate edges as separate entries (no coalesce), which sums in deg and in the SpMM,
    so summing duplicates up front is equivalent.  Returns CSR (rowptr, col, val f32)."""
    ei = np.asarray(edge_index, dtype=np.int64)
    w = np.ones(ei.shape[1], F32) if edge_weight is None else np.asarray(edge_weight, F32)
    off = ei[0] != ei[1]
    d = np.arange(n, dtype=np.int64)
    row = np.concatenate([ei[0][off], d])
    col = np.concatenate([ei[1][off], d])
    val = np.concatenate([w[off], np.ones(n, F32)])
    rowptr, col, val = csr_from_coo(row, col, val, n)
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    deg = np.zeros(n, F32)
    np.add.at(deg, rows, val)
    with np.errstate(divide="ignore"):
        dis = np.power(deg, F32(-0.5), dtype=F32)
    dis[np.isinf(dis)] = 0
    val = (val * dis[rows]).astype(F32)
    val = (val * dis[col]).astype(F32)
    return rowptr, col, val


def spmm(rowptr, col, val, x):
    """out[i] = sum_j A[i,j] x[j]  (GCNConv.propagate -> torch_sparse.matmul(adj_t, x, 'add'))."""
    n = rowptr.size - 1
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    out = np.zeros((n, x.shape[1]), F32)
    np.add.at(out, rows, x[col] * val[:, None])
    return out


def layer_norm(x, gamma, beta):
    """torch.nn.LayerNorm over the last dim (biased variance, eps 1e-5)."""
    x = x.astype(F32)
    mu = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    return (xc / np.sqrt(var + LN_EPS) * gamma + beta).astype(F32)


def linear(x, w, b=None):
    y = x.astype(F32) @ w.astype(F32).T
    return y if b is None else (y + b).astype(F32)


def gcn_encoder(x, adj_norm, P, prefix, n_layers, *, residual, use_ln, use_relu):
    """``GCN.forward`` (src/models/other_models.py:61-76) in eval mode (dropout is identity): per layer
    xi = A_hat (x W^T) + b -> LN (if layer_norm) -> ReLU (if relu) -> x = x + xi when residual and shapes match."""
    rowptr, col, val = adj_norm
    for i in range(n_layers):
        xi = spmm(rowptr, col, val, linear(x, P[f"{prefix}.convs.{i}.lin.weight"])) + P[f"{prefix}.convs.{i}.bias"]
        if use_ln:
            xi = layer_norm(xi, P[f"{prefix}.lns.{i}.weight"], P[f"{prefix}.lns.{i}.bias"])
        if use_relu:
            xi = np.maximum(xi, 0)
        x = (x + xi).astype(F32) if (residual and x.shape[-1] == xi.shape[-1]) else xi.astype(F32)
    return x


def propagate(x, adj_norm, P, cfg):
    """``LinkTransformer.propagate`` (src/models/link_transformer.py:110-129): NodeEncoder (feature dropout is
    identity in eval; src/modules/node_encoder.py:35-44) then the extra ``gnn_norm`` LayerNorm (:127)."""
    h = gcn_encoder(x.astype(F32), adj_norm, P, "model.node_encoder.gnn_encoder", cfg["gnn_layers"],
                    residual=cfg["residual"], use_ln=cfg["layer_norm"], use_relu=cfg["relu"])
    return layer_norm(h, P["model.gnn_norm.weight"], P["model.gnn_norm.bias"])


# =============================================================================== a6-a8: node selection
def _expand_rows(rowptr, nodes):
    """COO (pair position, flat index into the CSR arrays) for the rows `nodes` (one row per pair)."""
    start = rowptr[nodes]
    cnt = rowptr[nodes + 1] - start
    pos = np.repeat(np.arange(nodes.size, dtype=np.int64), cnt)
    base = np.repeat(start - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt)
    flat = np.arange(int(cnt.sum()), dtype=np.int64) + base
    return pos, flat


def _lookup(sorted_keys, vals, query):
    """vals[i] where sorted_keys[i]==query else 0 (absent sparse entries read as 0)."""
    out = np.zeros(query.size, F32)
    if sorted_keys.size == 0 or query.size == 0:
        return out, np.zeros(query.size, bool)
    idx = np.searchsorted(sorted_keys, query)
    idx[idx >= sorted_keys.size] = sorted_keys.size - 1
    hit = sorted_keys[idx] == query
    out[hit] = vals[idx[hit]]
    return out, hit


def select_nodes(batch, adj, ppr, thresholds, *, n, adj_unmasked=None):
    """``compute_node_mask`` + ``get_ppr_vals`` + ``get_non_1hop_ppr`` (src/models/link_transformer.py:214-319,
    434-481), eval mode (no ``drop_pairwise``).

    adj / adj_unmasked: (rowptr, col) 0/1 symmetric pattern; ppr: (rowptr, col, val f32).
    For pair position k=(a,b) and node v with t = A[a,v]+A[b,v] (:237):
      t in {1,2}: pa = fl32((fl32(fl32(P[a,v]*t)+t)-t)/t) (:290-291,316-317; P reads 0 where not stored),
                  kept iff pa >= theta and pb >= theta with theta = thresh_cn (t=2) / thresh_1hop (t=1) (:241-250)
      t == 0 w.r.t. the UNMASKED adjacency (:443), P[a,v] and P[b,v] both stored:
                  sa = fl32(fl32(P[a,v]+1)-1) (:464-476), kept iff sa,sb >= thresh_non1hop (:478), mode "all" only.
    Returns dict tag -> (ix int64 [2,n], pa f32, pb f32), each sorted by (pair position, node) like the
    coalesced COO the reference reads them from.  Modes follow :39-44."""
    th_cn, th_1, th_n = thresholds
    mode = "cn" if (th_n == 1 and th_1 == 1) else ("1-hop" if (th_n == 1 and th_1 < 1) else "all")
    batch = np.asarray(batch, dtype=np.int64)
    a, b = batch[0], batch[1]
    N = np.int64(n)
    a_rp, a_col = adj
    p_rp, p_col, p_val = ppr
    p_val = np.asarray(p_val, dtype=F32)   # (no copy when it already is fp32: callers pass whole-graph arrays)
    if mode == "cn":
        return {"cn": _select_cn_only(a, b, N, a_rp, a_col, p_rp, p_col, p_val, th_cn)}

    # union of the two adjacency rows with multiplicity t (sparse add, :237)
    pos_a, fl_a = _expand_rows(a_rp, a)
    pos_b, fl_b = _expand_rows(a_rp, b)
    keys = np.concatenate([pos_a * N + a_col[fl_a], pos_b * N + a_col[fl_b]])
    ukeys, t = np.unique(keys, return_counts=True)
    tf = t.astype(F32)

    # PPR rows of a and b keyed the same way (index_select on the sparse PPR, :290-291)
    ppos_a, pfl_a = _expand_rows(p_rp, a)
    ppos_b, pfl_b = _expand_rows(p_rp, b)
    pk_a, pv_a = ppos_a * N + p_col[pfl_a], p_val[pfl_a]
    pk_b, pv_b = ppos_b * N + p_col[pfl_b], p_val[pfl_b]

    raw_a, _ = _lookup(pk_a, pv_a, ukeys)
    raw_b, _ = _lookup(pk_b, pv_b, ukeys)
    pa = ((raw_a * tf + tf) - tf) / tf
    pb = ((raw_b * tf + tf) - tf) / tf
    keep_cn = (pa >= F32(th_cn)) & (pb >= F32(th_cn))
    keep_1 = (pa >= F32(th_1)) & (pb >= F32(th_1))
    keep = np.where(t == 1, keep_1, keep_cn)

    def pack(sel):
        k = ukeys[sel]
        return np.stack([k // N, k % N]).astype(np.int64), pa[sel].astype(F32), pb[sel].astype(F32)

    out = {"cn": pack(keep & (t == 2)), "onehop": pack(keep & (t == 1))}
    if mode != "all":
        return out

    # >1-hop: both PPR entries stored, not adjacent to either endpoint in the UNMASKED graph
    if adj_unmasked is None:
        excl = ukeys
    else:
        u_rp, u_col = adj_unmasked
        qa, fa = _expand_rows(u_rp, a)
        qb, fb = _expand_rows(u_rp, b)
        excl = np.unique(np.concatenate([qa * N + u_col[fa], qb * N + u_col[fb]]))
    common, ia, ib = np.intersect1d(pk_a, pk_b, assume_unique=True, return_indices=True)
    far = ~np.isin(common, excl, assume_unique=True)
    va, vb = pv_a[ia][far], pv_b[ib][far]
    both = (va > 0) & (vb > 0)  # sign() term (:464-465); stored PPR values are > 0
    sa = ((va + F32(1)) - F32(1)).astype(F32)
    sb = ((vb + F32(1)) - F32(1)).astype(F32)
    ok = both & (sa >= F32(th_n)) & (sb >= F32(th_n))
    k = common[far][ok]
    out["non1hop"] = (np.stack([k // N, k % N]).astype(np.int64), sa[ok], sb[ok])
    return out


def _select_cn_only(a, b, N, a_rp, a_col, p_rp, p_col, p_val, th_cn):
    """Mask mode "cn" (thresh_1hop == thresh_non1hop == 1, src/models/link_transformer.py:39-44).  PARITY UNPINNED:
    the reference crashes in this mode under torch >= 2.1 (sparse * sparse now keeps explicit zeros and the
    work-around at :304-313 indexes with a mask of the wrong length), so no reference output can be recorded here;
    this restates what the code computes on the torch versions it was written for.
      pair_adj = src_adj * tgt_adj (:232-234): value 1 exactly at the common neighbours, nothing elsewhere;
      get_ppr_vals with t = 1 everywhere (:290-291,316-317): pa = fl32(fl32(fl32(P[a,v]*1)+1)-1)/1, likewise pb
        (P reads 0 where not stored);
      node_type == 1 for every entry, so the filter is cn_filt_cond (:241-247): pa >= thresh_cn and pb >= thresh_cn;
      returned as (pair_ix, src_ppr, tgt_ppr), None, None (:271-272), sorted by (pair position, node)."""
    pos_a, fl_a = _expand_rows(a_rp, a)
    pos_b, fl_b = _expand_rows(a_rp, b)
    ka, kb = pos_a * N + a_col[fl_a], pos_b * N + a_col[fl_b]
    ukeys = np.intersect1d(ka, kb)          # rows are duplicate-free: the intersection is the common neighbours
    ppos_a, pfl_a = _expand_rows(p_rp, a)
    ppos_b, pfl_b = _expand_rows(p_rp, b)
    raw_a, _ = _lookup(ppos_a * N + p_col[pfl_a], p_val[pfl_a], ukeys)
    raw_b, _ = _lookup(ppos_b * N + p_col[pfl_b], p_val[pfl_b], ukeys)
    one = F32(1)
    pa = (((raw_a * one + one) - one) / one).astype(F32)
    pb = (((raw_b * one + one) - one) / one).astype(F32)
    keep = (pa >= F32(th_cn)) & (pb >= F32(th_cn))
    k = ukeys[keep]
    return np.stack([k // N, k % N]).astype(np.int64), pa[keep], pb[keep]


# =============================================================================== a9, a13: MLPs
def mlp2(x, P, prefix):
    """2-layer ``MLP`` (src/models/other_models.py:125-138): Linear -> LayerNorm -> ReLU -> (dropout 0) -> Linear."""
    h = linear(x, P[f"{prefix}.linears.0.weight"], P[f"{prefix}.linears.0.bias"])
    h = np.maximum(layer_norm(h, P[f"{prefix}.norm.weight"], P[f"{prefix}.norm.bias"]), 0)
    return linear(h, P[f"{prefix}.linears.1.weight"], P[f"{prefix}.linears.1.bias"])


def pos_encodings(sel, P):
    """``get_pos_encodings`` (src/models/link_transformer.py:182-211): per type pe = g([pa,pb]) + g([pb,pa]),
    concatenated in (cn, onehop, non1hop) order."""
    enc = {"cn": "model.ppr_encoder_cn", "onehop": "model.ppr_encoder_onehop", "non1hop": "model.ppr_encoder_non1hop"}
    parts = []
    for tag in ("cn", "onehop", "non1hop"):
        if tag not in sel:
            continue
        _, pa, pb = sel[tag]
        ab = np.stack([pa, pb], axis=1).astype(F32)
        ba = np.stack([pb, pa], axis=1).astype(F32)
        parts.append(mlp2(ab, P, enc[tag]) + mlp2(ba, P, enc[tag]))
    return np.concatenate(parts, axis=0).astype(F32)


# =============================================================================== a10, a11: attention
def link_attention(ix, x_node, batch, pes, P, bs, prefix="model.att_layers.0"):
    """``LinkAttention.forward/message`` (src/modules/layers.py:161-224) with PyG's target_to_source lifting,
    segment softmax (max-shifted, denominator + 1e-16) and scatter-sum; H = 1.  Returns (pre-LN out, alpha)."""
    pair, node = ix[0], ix[1]
    w_l, b_l = P[f"{prefix}.att.lin_l.weight"], P[f"{prefix}.att.lin_l.bias"]
    w_r, b_r = P[f"{prefix}.att.lin_r.weight"], P[f"{prefix}.att.lin_r.bias"]
    att = P[f"{prefix}.att.att"].reshape(-1)
    k = linear(np.concatenate([x_node[node], pes], axis=1), w_r, b_r)                 # :206-208
    q = linear(x_node[batch[0]], w_l, b_l) + linear(x_node[batch[1]], w_l, b_l)        # :212-214
    s = k * q[pair]
    s = np.where(s > 0, s, F32(0.2) * s).astype(F32)                                   # :217
    score = (s * att).sum(axis=1, dtype=F32)                                            # :218
    smax = np.full(bs, -np.inf, F32)
    np.maximum.at(smax, pair, score)
    e = np.exp(score - smax[pair]).astype(F32)
    den = np.zeros(bs, F32)
    np.add.at(den, pair, e)
    alpha = (e / (den + F32(1e-16))[pair]).astype(F32)                                 # :220
    out = np.zeros((bs, k.shape[1]), F32)
    np.add.at(out, pair, k * alpha[:, None])                                            # :224 + aggregate
    return (out + P[f"{prefix}.att.bias"]).astype(F32), alpha                           # :184-185


def structure_counts(sel, bs):
    """``get_structure_cnts`` (src/models/link_transformer.py:340-386): counts of the SELECTED sets;
    num_neighbors = n_cn + n_1hop."""
    def cnt(tag):
        c = np.zeros(bs, F32)
        if tag in sel:
            np.add.at(c, sel[tag][0][0], F32(1))
        return c
    n_cn, n_1 = cnt("cn"), cnt("onehop")
    if "onehop" not in sel:  # mode "cn": the single count of get_count (:155)
        return n_cn[:, None].astype(F32)
    cols = [n_cn, n_1]
    if "non1hop" in sel:
        cols.append(cnt("non1hop"))
    cols.append(n_cn + n_1)
    return np.stack(cols, axis=1).astype(F32)


def calc_pairwise(batch, x_node, sel, P, *, want_parts=False):
    """``calc_pairwise`` (src/models/link_transformer.py:132-178), one attention layer, then
    ``LinkTransformerLayer``'s post-attention LayerNorm (src/modules/layers.py:78) and ``pairwise_lin``."""
    bs = batch.shape[1]
    order = [t for t in ("cn", "onehop", "non1hop") if t in sel]
    ix = np.concatenate([sel[t][0] for t in order], axis=1)
    pes = pos_encodings(sel, P)
    pre, alpha = link_attention(ix, x_node, batch, pes, P, bs)
    post = layer_norm(pre, P["model.att_layers.0.post_att_norm.weight"], P["model.att_layers.0.post_att_norm.bias"])
    feats = np.concatenate([post, structure_counts(sel, bs)], axis=1)
    out = mlp2(feats, P, "model.pairwise_lin")
    if want_parts:
        return out, {"att_pre_ln": pre, "att_post_ln": post, "alpha": alpha, "ix": ix, "pes": pes}
    return out


def mlp_score(h, P, n_layers=2):
    """``mlp_score.forward`` (src/models/other_models.py:173-179), eval mode: (Linear, ReLU)* Linear, sigmoid.
    Returns (probability, pre-sigmoid logit)."""
    for i in range(n_layers - 1):
        h = np.maximum(linear(h, P[f"score.lins.{i}.weight"], P[f"score.lins.{i}.bias"]), 0)
    logit = linear(h, P[f"score.lins.{n_layers - 1}.weight"], P[f"score.lins.{n_layers - 1}.bias"]).reshape(-1)
    return (F32(1) / (F32(1) + np.exp(-logit))).astype(F32), logit.astype(F32)


def forward(batch, x, adj_norm, adj_mask, ppr, P, cfg, *, x_node=None, want_parts=False, adj_unmasked=None):
    """``LinkTransformer.forward`` (src/models/link_transformer.py:82-107) + the caller's ``score_func``
    (src/train/testing.py:87-88).  Returns a dict of every stage's output."""
    batch = np.asarray(batch, dtype=np.int64)
    if x_node is None:
        x_node = propagate(x, adj_norm, P, cfg)
    th = (cfg["thresh_cn"], cfg["thresh_1hop"], cfg["thresh_non1hop"])
    sel = select_nodes(batch, adj_mask, ppr, th, n=x_node.shape[0], adj_unmasked=adj_unmasked)
    ew = mlp2((x_node[batch[0]] * x_node[batch[1]]).astype(F32), P, "model.elementwise_lin")   # :101-102
    pw, parts = calc_pairwise(batch, x_node, sel, P, want_parts=True)
    comb = np.concatenate([ew, pw], axis=1).astype(F32)                                          # :105
    prob, logit = mlp_score(comb, P, cfg.get("pred_layers", 2))
    res = {"x_node": x_node, "sel": sel, "elementwise_feats": ew, "pairwise_feats": pw,
           "combined_feats": comb, "prob": prob, "logit": logit}
    if want_parts:
        res.update(parts)
    return res


# =============================================================================== a18: PPR push
def ppr_push(indptr, indices, alpha, eps):
    """``calc_ppr`` (src/util/calc_ppr_scores.py:136-192) as plain Python: Andersen push per source, LIFO stack,
    float64 arithmetic, p/r as insertion-ordered dicts; then ``create_sparse_ppr_matrix`` (:221-241): values
    rounded to float32, rows sorted by column.  Small graphs only.  Returns COO (row, col, val f32)."""
    n = len(indptr) - 1
    deg = np.diff(indptr)
    alpha_eps = alpha * eps
    rows, cols, vals = [], [], []
    for src in range(n):
        p = {src: 0.0}
        r = {src: alpha}
        q = [src]
        while q:
            u = q.pop()
            res = r.get(u, 0)
            p[u] = p.get(u, 0.0) + res
            r[u] = 0
            for v in indices[indptr[u]:indptr[u + 1]]:
                v = int(v)
                push = (1 - alpha) * res / deg[u]
                r[v] = r.get(v, 0) + push
                if r[v] >= alpha_eps * deg[v] and v not in q:
                    q.append(v)
        ks = np.fromiter(p.keys(), dtype=np.int64, count=len(p))
        vs = np.fromiter(p.values(), dtype=np.float64, count=len(p)).astype(F32)
        o = np.argsort(ks, kind="stable")
        rows.append(np.full(ks.size, src, np.int64))
        cols.append(ks[o])
        vals.append(vs[o])
    return np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)


def edge_csr(edge_index, n):
    """``get_ppr_matrix`` graph prep (src/util/calc_ppr_scores.py:111-117): coalesce (sort + dedup) then CSR."""
    ei = np.asarray(edge_index, dtype=np.int64)
    rowptr, col, _ = csr_from_coo(ei[0], ei[1], None, n)
    return rowptr, col
