"""An adjacency override as a DIFFERENCE to the resident adjacency (the training step).

The reference's training loop hands ``model(edges, adj_mask=masked_adj)`` a fresh sparse tensor per batch: every training
edge except the batch's own positives (src/train/train_model.py:38-46).  ``compute_node_mask`` types a candidate node v
of pair (a, b) by t = A'[a, v] + A'[b, v] over that masked adjacency A' (src/models/link_transformer.py:229-237) while the
>1-hop set keeps excluding the neighbours of the UNMASKED adjacency A (:438-443).  Round 2-5 turned every such tensor into
a CSR of its own (a device sort + unique over all 2 E edges, 4 ms per batch on the collab-like graph) and ran the general
selection kernels over it.  But A' = A minus a handful of edges R, and only entries (k, v) with (a_k, v) or (b_k, v) in R
can differ between the two selections:

  D_a = {v : (a, v) in R} is a subset of N(a); for v in D_a or D_b the type drops by one per removed side,
    t = 2 (common neighbour), one side removed  -> t' = 1: a one-hop node, values ((p * 1 + 1) - 1) / 1 of the RAW PPR
                                                   values (:290-291,316-317 -- not the t = 2 round trip the unmasked
                                                   selection stored), kept iff both >= thresh_1hop (:241-250)
    t = 2, both sides removed                   -> t' = 0: gone (it is adjacent in A, so it is no >1-hop node either)
    t = 1 (one-hop), its one side removed       -> t' = 0: gone
    t = 0 (>1-hop)                               -> never in D_a or D_b: unchanged

so the masked selection is the resident-index selection (``lpf_select3_*`` over the walk indexes, no graph rebuilt)
followed by this patch over the few entries that touch R.  Requires thresh_cn <= 0 (every shipped script): then every
common neighbour is IN the unmasked result and can be re-typed; with thresh_cn > 0 a common neighbour that failed its
own test could still pass the one-hop test after losing an edge -- such models keep the general path.

Everything here is torch index arithmetic on whatever device the tensors live on (the CPU tests run it against the oracle);
the one data-sized look-up -- raw PPR values of the re-typed entries -- is passed in (``lpf_csr_lookup_f32`` on the GPU).
"""
from __future__ import annotations

import torch


def edge_keys(rowptr: torch.Tensor, col: torch.Tensor, n: int) -> torch.Tensor:
    """Sorted int64 keys row * n + col of a CSR pattern with sorted columns."""
    rows = torch.repeat_interleave(torch.arange(n, device=rowptr.device), rowptr[1:] - rowptr[:-1])
    return rows * n + col.long()


def _member(sorted_keys: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    if sorted_keys.numel() == 0:
        return torch.zeros_like(q, dtype=torch.bool)
    pos = torch.searchsorted(sorted_keys, q).clamp_(max=sorted_keys.numel() - 1)
    return sorted_keys[pos] == q


def removed_from_edges(own: torch.Tensor, edges: torch.Tensor, n: int) -> torch.Tensor:
    """Directed keys of the undirected ``edges`` [2, K] that the adjacency ``own`` (sorted keys) holds: sorted, unique."""
    e = edges.long().reshape(2, -1)
    ok = (e[0] >= 0) & (e[0] < n) & (e[1] >= 0) & (e[1] < n)
    ok = torch.cat([ok, ok])
    q = torch.where(ok, torch.cat([e[0] * n + e[1], e[1] * n + e[0]]), torch.zeros((), dtype=torch.int64, device=e.device))
    # keys that are out of range or that the adjacency does not hold become -1; one more -1 is appended so that the sorted
    # unique keys ALWAYS start with it: ONE host read-back (unique's size) instead of one per filter
    q = torch.where(ok & _member(own, q), q, torch.full((), -1, dtype=torch.int64, device=e.device))
    return torch.unique(torch.cat([q, torch.full((1,), -1, dtype=torch.int64, device=e.device)]))[1:]


def removed_from_coo(own: torch.Tensor, row: torch.Tensor, col: torch.Tensor, n: int, limit: int):
    """The override (row, col) as a difference to ``own``: sorted directed keys own holds and the override does not -- or
    None when the override is not a subset of ``own`` or differs by more than ``limit`` directed edges (it is then a
    graph of its own).  One host synchronisation."""
    mk = row.long() * n + col.long()
    if mk.numel() == 0:
        return own if own.numel() <= limit else None
    inc = (mk[1:] > mk[:-1]).all() if mk.numel() > 1 else torch.ones((), dtype=torch.bool, device=mk.device)
    hit = _member(mk, own)                     # (meaningful only when mk is sorted and duplicate-free: checked below)
    n_hit, is_sorted = (int(v) for v in torch.stack([hit.sum(), inc.long()]).tolist())
    if not is_sorted:                          # (a SparseTensor-like object may hand its entries over in any order)
        mk = torch.unique(mk)
        hit = _member(mk, own)
        n_hit = int(hit.sum())
    if n_hit != mk.numel() or own.numel() - n_hit > limit:
        return None
    return own[~hit]


def round_trip1(v: torch.Tensor) -> torch.Tensor:
    """((v * 1 + 1) - 1) / 1 in separately rounded fp32 operations (link_transformer.py:290-291,316-317 with t = 1)."""
    one = torch.ones((), dtype=torch.float32, device=v.device)
    return (v * one + one) - one


def _positions(mask: torch.Tensor, count: int) -> torch.Tensor:
    """Indices of the ``count`` set elements of ``mask`` (count known on the host: no read-back where torch offers the
    fixed-size form on this device)."""
    try:
        return torch.nonzero_static(mask, size=count).flatten()
    except (RuntimeError, NotImplementedError, AttributeError):
        return torch.nonzero(mask).flatten()


def patch_selection(sel: dict, batch: torch.Tensor, rk: torch.Tensor, n: int, mode: str, th_1hop: float, lookup):
    """The selection of the MASKED adjacency from the selection ``sel`` of the resident one.

    sel: type-major arrays ``sel_pair`` / ``sel_node`` (int32), ``sel_pa`` / ``sel_pb`` (fp32) -- CN, then one-hop, then
    >1-hop entries, each run sorted by (pair, node) -- and ``type_ptr`` int64 [3 * (bs + 1)] of per-type segment pointers
    (``lpf_select_export``'s layout).  rk: sorted directed keys of the removed edges.  lookup(rows, cols) -> raw PPR
    values P[rows, cols] (fp32, 0 where nothing is stored).  Returns (pair, node, pa, pb, type_ptr, counts int64 [3, bs])
    in the same layout, new tensors."""
    bs = batch.shape[1]
    dev = batch.device
    tp = sel["type_ptr"][:3 * (bs + 1)].view(3, bs + 1)
    n0, n1, n2 = sel["tot"] if "tot" in sel else (int(v) for v in tp[:, bs].tolist())
    m, tot = n0 + n1, n0 + n1 + n2
    pair_i, node_i = sel["sel_pair"][:tot], sel["sel_node"][:tot]
    pa, pb = sel["sel_pa"][:tot], sel["sel_pb"][:tot]
    pair, node = pair_i[:m].long(), node_i[:m].long()
    a, b = batch[0][pair], batch[1][pair]
    in_a, in_b = _member(rk, a * n + node), _member(rk, b * n + node)
    # entries that touch no removed edge stay what they are, in their order (ONE compaction for both regions: the kept
    # common neighbours still precede the kept one-hop nodes)
    keep_m = ~(in_a | in_b)
    demote_m = (in_a ^ in_b)[:n0]
    n_kept, n_di = (int(v) for v in torch.stack([keep_m.sum(), demote_m.sum()]).tolist())     # (one read-back for both)
    kept = _positions(keep_m, n_kept)
    kp, kn, ka, kb = pair_i[kept], node_i[kept], pa[kept], pb[kept]
    types = (kept >= n0).long()                # 0: common neighbour, 1: one-hop (mode "cn": all 0)
    if mode != "cn":
        # common neighbours that lost ONE of their two edges: one-hop candidates now (pair_adj = 1, :237)
        di = _positions(demote_m, n_di)
        if n_di > 0:
            dn = node[di]
            raw = lookup(torch.cat([a[di], b[di]]), torch.cat([dn, dn]))
            va, vb = round_trip1(raw[:di.numel()]), round_trip1(raw[di.numel():])
            ok = torch.nonzero((va >= th_1hop) & (vb >= th_1hop)).flatten()
            if ok.numel() > 0:
                dk = di[ok]
                kp, kn = torch.cat([kp, pair_i[dk]]), torch.cat([kn, node_i[dk]])
                ka, kb = torch.cat([ka, va[ok]]), torch.cat([kb, vb[ok]])
                types = torch.cat([types, torch.ones_like(dk)])
                # back into (type, pair, node) order
                order = torch.argsort((types * bs + kp.long()) * n + kn.long())
                kp, kn, ka, kb, types = kp[order], kn[order], ka[order], kb[order], types[order]
    counts = torch.zeros(3, bs, dtype=torch.int64, device=dev)
    counts[:2] = torch.zeros(2 * bs, dtype=torch.int64, device=dev).index_add_(
        0, types * bs + kp.long(), torch.ones(kp.numel(), dtype=torch.int64, device=dev)).view(2, bs)   # (no host read-back)
    counts[2] = tp[2, 1:] - tp[2, :-1]
    new_tp = torch.zeros(3, bs + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, dim=1, out=new_tp[:, 1:])
    if n2:
        kp, kn = torch.cat([kp, pair_i[m:tot]]), torch.cat([kn, node_i[m:tot]])
        ka, kb = torch.cat([ka, pa[m:tot]]), torch.cat([kb, pb[m:tot]])
    return kp, kn, ka, kb, new_tp.reshape(-1), counts


def prop_weights_from_coo(own_keys: torch.Tensor, own_w: torch.Tensor, row: torch.Tensor, col: torch.Tensor, val, n: int,
                          min_kept: float = 0.5):
    """A propagation-matrix override as a difference to the resident structure (the ``--mask-input`` loop,
    src/train/train_model.py:47-56: the same edges minus the batch's positives, through GCNConv's ``gcn_norm``).

    own_keys / own_w: sorted keys row * n + col and RAW weights of the resident GCN structure (every diagonal entry
    included).  Returns the raw weights of the override laid out on THAT structure -- removed edges weigh 0, the
    diagonal keeps its placeholder -- or None when the override is not a subset of the structure, weighs a kept edge
    differently or keeps less than ``min_kept`` of the edges (it is then normalised as a graph of its own).  The
    normalisation kernel (``lpf_gcn_norm_csr``) takes it from there: degrees from the weights that are left.  One host
    synchronisation."""
    row, col = row.long(), col.long()
    off = row != col
    mk = row * n + col
    w = torch.ones(mk.numel(), dtype=torch.float32, device=mk.device) if val is None else val.float().reshape(-1)
    pos = torch.searchsorted(own_keys, mk).clamp_(max=max(own_keys.numel() - 1, 0))
    found = (own_keys[pos] == mk) | ~off
    same = (own_w[pos] == w) | ~off
    new_w = torch.zeros_like(own_w)
    new_w[pos[off]] = w[off]                 # (duplicate entries of a non-coalesced override: the last one wins -- they are
    #                                           equal to the resident weight or the override is rejected below)
    r = torch.div(own_keys, n, rounding_mode="floor")
    diag = own_keys - r * n == r
    new_w[diag] = own_w[diag]
    n_off_own = own_keys.numel() - n
    ok, n_kept = (int(v) for v in torch.stack([(found & same).all().long(), (new_w != 0).sum() - diag.sum()]).tolist())
    if not ok or n_kept < min_kept * max(n_off_own, 1):
        return None
    return new_w


def prop_weights_minus_edges(own_keys: torch.Tensor, own_w: torch.Tensor, edges: torch.Tensor, n: int) -> torch.Tensor:
    """The raw weights of the resident GCN structure with the undirected ``edges`` [2, K] removed (both directions at 0;
    the diagonal keeps its placeholder, edges the structure does not hold and ids out of range are ignored) --
    ``adj_prop=RemovedEdges(edges)``: what ``prop_weights_from_coo`` derives from a tensor of all the kept edges, without
    the tensor.  No host synchronisation."""
    e = edges.long().reshape(2, -1)
    ok = (e[0] >= 0) & (e[0] < n) & (e[1] >= 0) & (e[1] < n) & (e[0] != e[1])
    q = torch.cat([e[0] * n + e[1], e[1] * n + e[0]])
    ok2 = torch.cat([ok, ok])
    new_w = own_w.clone()
    if own_keys.numel() == 0 or q.numel() == 0:
        return new_w
    pos = torch.searchsorted(own_keys, q).clamp_(max=own_keys.numel() - 1)
    hit = (own_keys[pos] == q) & ok2
    new_w[pos[hit]] = 0.0
    return new_w
