"""Training-mode forward of ``LinkTransformer`` / ``mlp_score`` with autograd (rows a15, f2).

The reference's training step (src/train/train_model.py:35-81) calls ``model(edges, adj_prop, adj_mask)`` in train
mode, ``score_func(h)``, and back-propagates a log-loss.  This module builds that forward as an autograd graph on the
device:

* the heavy, data-sized operators are ``torch.autograd.Function`` wrappers around the C-ABI kernels, forward AND
  backward: every Linear (``lpf_gemm_f32``: y = x W^T, dx = dy W; ``lpf_gemm_tn_f32``: dW = dy^T x), every LayerNorm
  (``lpf_layernorm_f32`` / ``lpf_layernorm_bwd_f32``), the GCN aggregation
  (``lpf_spmm_csr_f32`` with the propagation matrix, its transpose for the gradient) and the node selection
  (``lpf_select_plan`` / ``_run`` / ``_export``; integer work, no gradient);
* ReLU, dropout, the leaky-ReLU score, the per-pair segment softmax and the index gathers / scatter-adds
  around them are ordinary differentiable torch operators on the device in this version (elementwise and
  index-bound; they are the next candidates for dedicated kernels -- DESIGN.md section 8).

Evaluation never comes through here: ``model.eval()`` takes the fused inference kernels.  Gradients are pinned by
reference-generated fixtures (tests/golden/train_step_*.npz, tests/test_gpu_train.py).

``drop_pairwise`` (link_transformer.py:322-337): keeps ceil(n (1 - p)) entries chosen by a random permutation,
applied to (CN + 1-hop) together and to (>1-hop) separately (:257-260); the permutation is drawn on the device here
(the reference draws it on the CPU), so the kept set is equal in distribution, not bitwise.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import _lib, graph
from ._lib import check, ptr


def _stream(t: torch.Tensor):
    return torch._C._cuda_getCurrentRawStream(t.device.index if t.device.index is not None
                                              else torch.cuda.current_device())


def _rows4(x: torch.Tensor) -> torch.Tensor:
    """fp32, 2-D, inner stride 1, row stride a multiple of 4 floats and 16-byte aligned (what lpf_gemm_f32 wants);
    zero-padded copy when needed."""
    x = x.detach()
    if x.dtype != torch.float32:
        x = x.float()
    if x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16 or x.stride(0) < x.shape[1]:
        k = x.shape[1]
        buf = torch.zeros(x.shape[0], (k + 3) & ~3, dtype=torch.float32, device=x.device)
        buf[:, :k] = x
        x = buf[:, :k]
    return x


def _gemm(a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """a [M, K] @ w[N, K]^T -> [M, N] through lpf_gemm_f32."""
    a, w = _rows4(a), _rows4(w)
    m, k = a.shape
    n = w.shape[0]
    out = torch.empty(m, (n + 3) & ~3, dtype=torch.float32, device=a.device)[:, :n]
    if m == 0 or n == 0:
        return out
    if k == 0:
        return out.zero_()
    check(_lib.hip().lpf_gemm_f32(m, n, k, ptr(a), a.stride(0), ptr(w), w.stride(0), None, None, 0, ptr(out),
                                  out.stride(0), 0, _stream(a)), "lpf_gemm_f32")
    return out


def _gemm_tn(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a[M, N]^T @ b[M, K] -> [N, K] through lpf_gemm_tn_f32 (the weight gradient of a Linear)."""
    a, b = a.contiguous(), b.contiguous()
    m, n = a.shape
    k = b.shape[1]
    out = torch.empty(n, k, dtype=torch.float32, device=a.device)
    lib = _lib.hip()
    ws = torch.empty(max(int(lib.lpf_gemm_tn_workspace_floats(m, n, k)), 1), dtype=torch.float32, device=a.device)
    check(lib.lpf_gemm_tn_f32(m, n, k, ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(out), out.stride(0), ptr(ws),
                              _stream(a)), "lpf_gemm_tn_f32")
    return out


class LinearFn(torch.autograd.Function):
    """y = x W^T (+ b); forward and both gradients on the fp32 matrix cores (lpf_gemm_f32)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        y = _gemm(x, weight)
        return y + bias if bias is not None else y.clone()

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _gemm(dy, weight.t().contiguous()).contiguous()        # [M, N] @ [K, N]^T
        if ctx.needs_input_grad[1]:
            dw = _gemm_tn(dy, x)                                          # dY^T X, rows split over the GPU
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(dim=0)
        return dx, dw, db


def linear(x, weight, bias=None):
    return LinearFn.apply(x, weight, bias)


def _spmm_plain(a: graph.DeviceCSR, h: torch.Tensor) -> torch.Tensor:
    """out = A h (no epilogue) through lpf_spmm_csr_f32."""
    h = _rows4(h)
    n, d = a.n, h.shape[1]
    out = torch.empty(n, d, dtype=torch.float32, device=h.device)
    deg = a.rowptr[1:] - a.rowptr[:-1]
    cache = a.__dict__.setdefault("_long_rows_full", None)
    if cache is None:
        rows = torch.nonzero(deg > 128).flatten().to(torch.int32)
        cache = a.__dict__["_long_rows_full"] = (rows if rows.numel() else False)
    long_rows = cache if cache is not False else None
    check(_lib.hip().lpf_spmm_csr_f32(n, d, ptr(a.rowptr), ptr(a.col), ptr(a.val), ptr(h), h.stride(0), ptr(out),
                                      out.stride(0), None, None, None, None, 0, None, None, 0, ptr(long_rows),
                                      0 if long_rows is None else long_rows.numel(), _stream(h)), "lpf_spmm_csr_f32")
    return out


def _transpose_csr(a: graph.DeviceCSR) -> graph.DeviceCSR:
    """A^T as a device CSR (cached on ``a``): needed for the gradient of the aggregation."""
    hit = a.__dict__.get("_transposed")
    if hit is None:
        n = a.n
        rows = torch.repeat_interleave(torch.arange(n, device=a.col.device), a.rowptr[1:] - a.rowptr[:-1])
        key = a.col.long() * n + rows                      # sort by (column, row)
        order = torch.argsort(key)
        counts = torch.bincount(a.col.long(), minlength=n)
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=a.col.device)
        torch.cumsum(counts, 0, out=rowptr[1:])
        hit = a.__dict__["_transposed"] = graph.DeviceCSR(rowptr, rows[order].to(torch.int32).contiguous(),
                                                          a.val[order].contiguous(), n, None)
    return hit


class SpmmFn(torch.autograd.Function):
    """out = A_hat t; gradient dt = A_hat^T dout (both lpf_spmm_csr_f32)."""

    @staticmethod
    def forward(ctx, t, a_hat):
        ctx.a_hat = a_hat
        return _spmm_plain(a_hat, t)

    @staticmethod
    def backward(ctx, dout):
        return _spmm_plain(_transpose_csr(ctx.a_hat), dout.contiguous()), None


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm (eps 1e-5) through lpf_layernorm_f32 / lpf_layernorm_bwd_f32."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        y = torch.empty_like(x2)
        check(_lib.hip().lpf_layernorm_f32(x2.shape[0], x2.shape[1], ptr(x2), x2.stride(0), ptr(weight), ptr(bias),
                                           ptr(y), y.stride(0), 0, _stream(x2)), "lpf_layernorm_f32")
        ctx.save_for_backward(x2, weight)
        ctx.shape = x.shape
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        d = x2.shape[1]
        dy2 = dy.reshape(-1, d).contiguous()
        dx = torch.empty_like(x2)
        dg, db = torch.empty(d, dtype=torch.float32, device=x2.device), torch.empty(d, dtype=torch.float32, device=x2.device)
        lib = _lib.hip()
        ws = torch.empty(int(lib.lpf_layernorm_bwd_workspace_floats(d)), dtype=torch.float32, device=x2.device)
        check(lib.lpf_layernorm_bwd_f32(x2.shape[0], d, ptr(x2), x2.stride(0), ptr(dy2), dy2.stride(0), ptr(weight),
                                        ptr(dx), dx.stride(0), ptr(dg), ptr(db), ptr(ws), _stream(x2)),
              "lpf_layernorm_bwd_f32")
        return dx.reshape(ctx.shape), dg, db


def layer_norm(x, weight, bias):
    """LayerNorm over the last dimension; the C-ABI kernels when the width allows (D % 4 == 0, D <= 256)."""
    d = x.shape[-1]
    if d % 4 or d > 256 or x.numel() == 0:
        return F.layer_norm(x, (d,), weight, bias)
    return LayerNormFn.apply(x, weight, bias)


def _mlp(mod, x):
    """The reference's MLP (other_models.py:125-138): (Linear -> LayerNorm -> ReLU -> dropout)* -> Linear."""
    for lin in mod.linears[:-1]:
        x = linear(x, lin.weight, lin.bias)
        if mod.norm is not None:
            x = layer_norm(x, mod.norm.weight, mod.norm.bias)
        x = F.relu(x)
        x = F.dropout(x, p=mod.dropout, training=True)
    last = mod.linears[-1]
    return linear(x, last.weight, last.bias)


def _pe_mlp(mod, pa, pb):
    """ppr_encoder_*: g([pa, pb]) + g([pb, pa]) (link_transformer.py:182-211); the 2 -> D first layer is two
    broadcast multiply-adds."""
    l0, l1 = mod.linears
    w, b = l0.weight, l0.bias

    def g(x, y):
        hdn = x[:, None] * w[:, 0][None, :] + y[:, None] * w[:, 1][None, :] + b[None, :]
        hdn = F.relu(layer_norm(hdn, mod.norm.weight, mod.norm.bias))
        return linear(hdn, l1.weight, l1.bias)

    return g(pa, pb) + g(pb, pa)


def drop_pairwise(n: int, p: float, device) -> torch.Tensor:
    """Indices kept by the reference's ``drop_pairwise`` (link_transformer.py:322-337): ceil(n (1 - p)) entries of a
    random permutation, in permuted order."""
    keep = math.ceil(n * (1 - p))
    return torch.randperm(n, device=device)[:keep]


def forward_train(model, batch, adj_prop=None, adj_mask=None, test_set=False):
    """``LinkTransformer.forward`` in training mode (link_transformer.py:82-178 with the dropouts of
    node_encoder.py:40, other_models.py:69, layers.py:80 and the random attention drop of :257-260) -> [BS, 2D] with
    an autograd graph."""
    dev = model.device
    batch = model._prep_batch(batch)
    bs, d = batch.shape[1], model.dim
    enc = model.node_encoder.gnn_encoder
    a_hat = model._device_graph("prop", model._data_obj("adj", test_set) if adj_prop is None else adj_prop)
    # ---- encoder (node_encoder.py:35-44, other_models.py:61-76, link_transformer.py:127)
    x = model._features()
    x = F.dropout(x, p=model.node_encoder.feat_drop, training=True)
    for i, conv in enumerate(enc.convs):
        xi = SpmmFn.apply(linear(x, conv.lin.weight, None), a_hat) + conv.bias
        if enc.lns is not None:
            xi = layer_norm(xi, enc.lns[i].weight, enc.lns[i].bias)
        xi = F.dropout(xi, p=enc.dropout, training=True)
        if enc.relu:
            xi = F.relu(xi)
        x = x + xi if (enc.residual and x.shape[-1] == xi.shape[-1]) else xi
    x_node = layer_norm(x, model.gnn_norm.weight, model.gnn_norm.bias)
    xa, xb = x_node[batch[0]], x_node[batch[1]]
    ew = _mlp(model.elementwise_lin, xa * xb)
    # ---- selection (integer work, no gradient) in the reference's layout, then the random attention drop
    with torch.no_grad():
        s = model._select(batch, test_set, adj_mask)
        tp = s["type_ptr"][:3 * (bs + 1)].view(3, bs + 1)
        tot = [int(v) for v in tp[:, bs].tolist()]
        n_types = 3 if model.mask == "all" else 2
        segs, base = [], 0
        for t in range(3):
            sl = slice(base, base + tot[t])
            segs.append((s["sel_pair"][sl].long(), s["sel_node"][sl].long(), s["sel_pa"][sl].clone(),
                         s["sel_pb"][sl].clone()))
            base += tot[t]
        if model.att_drop > 0:
            # CN and 1-hop are dropped together, >1-hop separately (link_transformer.py:257-260)
            n01 = tot[0] + tot[1]
            keep = drop_pairwise(n01, model.att_drop, dev)
            k0, k1 = keep[keep < tot[0]], keep[keep >= tot[0]] - tot[0]
            segs[0] = tuple(v[k0] for v in segs[0])
            segs[1] = tuple(v[k1] for v in segs[1])
            if n_types == 3:
                k2 = drop_pairwise(tot[2], model.att_drop, dev)
                segs[2] = tuple(v[k2] for v in segs[2])
        counts = [torch.bincount(sg[0], minlength=bs).float() for sg in segs]
    # ---- positional encodings + attention (link_transformer.py:182-211, layers.py:161-224)
    encoders = [model.ppr_encoder_cn, model.ppr_encoder_onehop] + ([model.ppr_encoder_non1hop] if n_types == 3 else [])
    pair = torch.cat([segs[t][0] for t in range(n_types)])
    node = torch.cat([segs[t][1] for t in range(n_types)])
    pes = torch.cat([_pe_mlp(encoders[t], segs[t][2], segs[t][3]) for t in range(n_types)])
    att = model.att_layers[0].att
    k = linear(torch.cat([x_node[node], pes], dim=1), att.lin_r.weight, att.lin_r.bias)
    q = linear(xa, att.lin_l.weight, att.lin_l.bias) + linear(xb, att.lin_l.weight, att.lin_l.bias)
    score = (F.leaky_relu(k * q[pair], 0.2) * att.att.reshape(1, -1)).sum(dim=-1)
    # PyG softmax over the entries of a pair: shift by the segment max, denominator + 1e-16
    smax = torch.full((bs,), float("-inf"), device=dev).scatter_reduce(0, pair, score.detach(), "amax",
                                                                      include_self=True)
    e = torch.exp(score - smax[pair])
    den = torch.zeros(bs, device=dev).index_add(0, pair, e) + 1e-16
    alpha = e / den[pair]
    out = torch.zeros(bs, d, device=dev).index_add(0, pair, k * alpha[:, None]) + att.bias
    layer = model.att_layers[0]
    out = layer_norm(out, layer.post_att_norm.weight, layer.post_att_norm.bias)
    out = F.dropout(out, p=layer.dropout, training=True)
    # ---- count features + pairwise_lin (link_transformer.py:170-177, 340-356)
    if n_types == 3:
        cf = torch.stack([counts[0], counts[1], counts[2], counts[0] + counts[1]], dim=1)
    else:
        cf = torch.stack([counts[0], counts[1], counts[0] + counts[1]], dim=1)
    pw = _mlp(model.pairwise_lin, torch.cat([out, cf], dim=1))
    return torch.cat([ew, pw], dim=-1)


def score_train(score_func, x):
    """``mlp_score.forward`` in training mode (other_models.py:173-179): (Linear, ReLU, dropout)* Linear, sigmoid."""
    for lin in score_func.lins[:-1]:
        x = F.relu(linear(x, lin.weight, lin.bias))
        x = F.dropout(x, p=score_func.dropout, training=True)
    last = score_func.lins[-1]
    return torch.sigmoid(linear(x, last.weight, last.bias)).squeeze(-1)
