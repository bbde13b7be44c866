"""Training-mode forward of ``LinkTransformer`` / ``mlp_score`` with autograd (rows a15, f2).

The reference's training step (src/train/train_model.py:35-81) calls ``model(edges, adj_prop, adj_mask)`` in train
mode, ``score_func(h)``, and back-propagates a log-loss.  This module builds that forward as an autograd graph on the
device:

* the heavy, data-sized operators are ``torch.autograd.Function`` wrappers around the C-ABI kernels, forward AND
  backward: every Linear (``lpf_gemm_f32``: y = x W^T, dx = dy W; ``lpf_gemm_tn_f32``: dW = dy^T x), every LayerNorm
  (``lpf_layernorm_f32`` / ``lpf_layernorm_bwd_f32``), the GCN aggregation
  (``lpf_spmm_csr_f32`` with the propagation matrix, its transpose for the gradient) and the node selection
  (``lpf_select_plan`` / ``_run`` / ``_export``; integer work, no gradient);
* the pair stage has kernels of its own (csrc/pair_train.hip): the PE hidden layer, the per-pair leaky-ReLU attention
  with PyG's segment softmax (forward saving the raw scores and the per-pair softmax state, backward producing dK, dZ,
  dq, datt, dbias), the gradients of the PE hidden layer, the endpoint gathers and their scatter-add, column sums;
* ReLU, dropout masks, bias adds, the parameter-sized folds (W_rp W2) and the integer bookkeeping of the random
  attention drop are ordinary torch operators on the device.

Evaluation never comes through here: ``model.eval()`` takes the fused inference kernels.  Gradients are pinned by
reference-generated fixtures (tests/golden/train_step_*.npz, tests/test_gpu_train.py).

``drop_pairwise`` (link_transformer.py:322-337): keeps ceil(n (1 - p)) entries chosen by a random permutation,
applied to (CN + 1-hop) together and to (>1-hop) separately (:257-260); the permutation is drawn on the device here
(the reference draws it on the CPU), so the kept set is equal in distribution, not bitwise.
"""
from __future__ import annotations

import math
import weakref

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


def _gemm(a: torch.Tensor, w: torch.Tensor, bias=None) -> torch.Tensor:
    """a [M, K] @ w[N, K]^T (+ bias) -> [M, N] through lpf_gemm_f32 (the bias rides in the kernel's epilogue)."""
    a, w = _rows4(a), _rows4(w)
    m, k = a.shape
    n = w.shape[0]
    out = torch.empty(m, (n + 3) & ~3, dtype=torch.float32, device=a.device)[:, :n]
    if m == 0 or n == 0:
        return out
    if k == 0:
        return out.zero_() if bias is None else out.copy_(bias.detach().expand(m, n))
    if bias is not None:
        bias = bias.detach().float().contiguous()
    check(_lib.hip().lpf_gemm_f32(m, n, k, ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), None, 0, ptr(out),
                                  out.stride(0), 0, _stream(a)), "lpf_gemm_f32")
    return out


def _gemm_tn(a: torch.Tensor, b: torch.Tensor, colsum: bool = False):
    """a[M, N]^T @ b[M, K] -> [N, K] through lpf_gemm_tn_f32 (the weight gradient of a Linear); ``colsum``: also the
    column sums of ``a`` (the bias gradient) from the same pass -> (dW, db)."""
    a, b = a.contiguous(), b.contiguous()
    m, n = a.shape
    k = b.shape[1]
    out = torch.empty(n, k, dtype=torch.float32, device=a.device)
    lib = _lib.hip()
    ws = torch.empty(max(int(lib.lpf_gemm_tn_workspace_floats(m, n, k)), 1), dtype=torch.float32, device=a.device)
    if colsum and n > 0 and k > 0:
        cs = torch.empty(n, dtype=torch.float32, device=a.device)
        check(lib.lpf_gemm_tn_colsum_f32(m, n, k, ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(out), out.stride(0),
                                         ptr(cs), ptr(ws), _stream(a)), "lpf_gemm_tn_colsum_f32")
        return out, cs
    check(lib.lpf_gemm_tn_f32(m, n, k, ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(out), out.stride(0), ptr(ws),
                              _stream(a)), "lpf_gemm_tn_f32")
    return (out, _colsum(a)) if colsum else out


def _colsum(x: torch.Tensor) -> torch.Tensor:
    """Column sums of a [M, D] matrix (bias gradients) through lpf_colsum_f32 (deterministic); torch for odd widths."""
    m, d = x.shape
    if d % 4 or d > 256 or x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16:
        return x.sum(dim=0)
    out = torch.empty(d, dtype=torch.float32, device=x.device)
    lib = _lib.hip()
    ws = torch.empty(int(lib.lpf_train_partial_blocks(m)) * d, dtype=torch.float32, device=x.device)
    check(lib.lpf_colsum_f32(m, d, ptr(x), x.stride(0), ptr(out), ptr(ws), _stream(x)), "lpf_colsum_f32")
    return out


_wt_cache: dict = {}


def _transposed(weight: torch.Tensor) -> torch.Tensor:
    """``weight.t().contiguous()`` kept per PARAMETER OBJECT and version: a step runs every Linear's backward twice
    (positives and negatives, src/train/train_model.py:59,66) on the same parameter version.  The entry holds a weak
    reference to the parameter and is valid only for that very object (an address or a storage pointer alone can be a
    later model's parameter at the same version); temporaries are never cached."""
    if not isinstance(weight, torch.nn.Parameter):
        return weight.t().contiguous()
    hit = _wt_cache.get(id(weight))
    if hit is not None and hit[0]() is weight and hit[1] == weight._version:
        return hit[2]
    if len(_wt_cache) >= 256:
        for k in [k for k, v in _wt_cache.items() if v[0]() is None]:
            del _wt_cache[k]
        if len(_wt_cache) >= 256:
            _wt_cache.clear()
    wt = weight.detach().t().contiguous()
    _wt_cache[id(weight)] = (weakref.ref(weight), weight._version, wt)
    return wt


class LinearFn(torch.autograd.Function):
    """y = x W^T (+ b); forward and both gradients on the fp32 matrix cores (lpf_gemm_f32)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return _gemm(x, weight, bias)      # a fresh tensor: nothing else holds it

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _gemm(dy, _transposed(weight)).contiguous()            # [M, N] @ [K, N]^T
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] and want_db:
            dw, db = _gemm_tn(dy, x, colsum=True)                         # dY^T X and dY^T 1 in one pass over dY
        elif ctx.needs_input_grad[1]:
            dw = _gemm_tn(dy, x)                                          # dY^T X, rows split over the GPU
        elif want_db:
            db = _colsum(dy)
        return dx, dw, db


def linear(x, weight, bias=None):
    return LinearFn.apply(x, weight, bias)


def _spmm_plain(a: graph.DeviceCSR, h: torch.Tensor, bias=None) -> torch.Tensor:
    """out = A h (+ bias) through lpf_spmm_csr_f32 (no other epilogue)."""
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
                                      out.stride(0), ptr(bias), None, None, None, 0, None, None, 0, ptr(long_rows),
                                      0 if long_rows is None else long_rows.numel(), _stream(h)), "lpf_spmm_csr_f32")
    return out


def _transpose_csr(a: graph.DeviceCSR) -> graph.DeviceCSR:
    """A^T as a device CSR (cached on ``a``): needed for the gradient of the aggregation."""
    hit = a.__dict__.get("_transposed")
    src = a.__dict__.get("_structure_of")
    if hit is None and src is not None:
        # an override that shares the resident graph's structure (LinkTransformer._prop_delta): the transposed structure
        # and the permutation are the resident graph's, only the values are gathered again
        t = _transpose_csr(src)
        hit = a.__dict__["_transposed"] = graph.DeviceCSR(t.rowptr, t.col, a.val[src.__dict__["_transposed_order"]].contiguous(),
                                                          a.n, None)
        # functions of the structure alone -- the one-launch layer's degree-ordered tiles and hub slices, the long-row
        # list of lpf_spmm_csr_f32 -- are the resident transposed graph's (recomputing them was a sort over the rows
        # and a host synchronisation per batch: 1.7 of the 15 ms of a step that overrides the propagation matrix)
        hit.__dict__["_fused_order"] = t.__dict__.setdefault("_fused_order", {})
        if t.__dict__.get("_long_rows_full") is None:
            deg = t.rowptr[1:] - t.rowptr[:-1]
            rows = torch.nonzero(deg > 128).flatten().to(torch.int32)
            t.__dict__["_long_rows_full"] = rows if rows.numel() else False
        hit.__dict__["_long_rows_full"] = t.__dict__["_long_rows_full"]
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
        a.__dict__["_transposed_order"] = order
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


class LnReluFn(torch.autograd.Function):
    """ReLU(LayerNorm(x)) as one forward kernel (lpf_layernorm_f32 with the ReLU flag) and one backward kernel
    (lpf_layernorm_relu_bwd_f32: the ReLU mask is recomputed from x, nothing but x is saved)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        y = torch.empty_like(x2)
        check(_lib.hip().lpf_layernorm_f32(x2.shape[0], x2.shape[1], ptr(x2), x2.stride(0), ptr(weight), ptr(bias),
                                           ptr(y), y.stride(0), _lib.FLAG_RELU, _stream(x2)), "lpf_layernorm_f32")
        ctx.save_for_backward(x2, weight, bias)
        ctx.shape = x.shape
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, bias = ctx.saved_tensors
        dx, dg, db, _ = _ln_relu_bwd(x2, dy.reshape(-1, x2.shape[1]).contiguous(), weight, bias)
        return dx.reshape(ctx.shape), dg, db


def _ln_relu_bwd(x2, dy2, weight, bias, drop_p: float = 0.0, drop_seed: int = 0):
    """(dx, dgamma, dbeta, column sums of dx) of y = ReLU(LN(x)) -- or, ``drop_p`` > 0, of y = dropout(ReLU(LN(x))) with
    the in-kernel mask of the fused GCN layer -- through lpf_layernorm_relu_drop_bwd_f32."""
    d = x2.shape[1]
    dev = x2.device
    dx = torch.empty_like(x2)
    dg = torch.empty(d, dtype=torch.float32, device=dev)
    db = torch.empty(d, dtype=torch.float32, device=dev)
    dxs = torch.empty(d, dtype=torch.float32, device=dev)
    lib = _lib.hip()
    ws = torch.empty(int(lib.lpf_layernorm_bwd_workspace_floats(d)), dtype=torch.float32, device=dev)
    check(lib.lpf_layernorm_relu_drop_bwd_f32(x2.shape[0], d, ptr(x2), x2.stride(0), ptr(dy2), dy2.stride(0), ptr(weight),
                                              ptr(bias), drop_p, drop_seed, ptr(dx), dx.stride(0), ptr(dg), ptr(db),
                                              ptr(dxs), ptr(ws), _stream(x2)), "lpf_layernorm_relu_drop_bwd_f32")
    return dx, dg, db, dxs


class GcnLayerFn(torch.autograd.Function):
    """One GCN layer behind its Linear: r = ReLU(LN(A_hat t + b)) (other_models.py:66-69 with the dropout moved behind
    the ReLU, which it commutes with).  Forward: the aggregation with the bias added in its epilogue
    (lpf_spmm_csr_f32), then LayerNorm + ReLU in one kernel; only the pre-norm tensor is saved.  Backward: one fused
    LayerNorm/ReLU backward that also yields the bias gradient, then the aggregation with A_hat^T."""

    @staticmethod
    def forward(ctx, t, a_hat, conv_bias, ln_w, ln_b):
        u = _spmm_plain(a_hat, t, bias=conv_bias)
        r = torch.empty_like(u)
        check(_lib.hip().lpf_layernorm_f32(u.shape[0], u.shape[1], ptr(u), u.stride(0), ptr(ln_w), ptr(ln_b), ptr(r),
                                           r.stride(0), _lib.FLAG_RELU, _stream(u)), "lpf_layernorm_f32")
        ctx.save_for_backward(u, ln_w, ln_b)
        ctx.a_hat = a_hat
        return r

    @staticmethod
    def backward(ctx, dr):
        u, ln_w, ln_b = ctx.saved_tensors
        du, dg, db, dbias = _ln_relu_bwd(u, dr.contiguous(), ln_w, ln_b)
        dt = _spmm_plain(_transpose_csr(ctx.a_hat), du)
        return dt, None, dbias, dg, db


def _fused_layer(a: graph.DeviceCSR, x2: torch.Tensor, wp: torch.Tensor, bias=None, ln_w=None, ln_b=None, flags: int = 0,
                 pre: bool = False, agg: bool = False, residual=None, drop_p: float = 0.0, drop_seed: int = 0):
    """One launch of ``lpf_gcn_layer_fused_train_f32`` over the whole graph ``a``:
    out = residual + dropout(epilogue((a x2) Wp^T)) -> (out, pre-norm rows or None, aggregated rows or None).  Hub rows go
    through their slice sums first (csrc/gcn_fused.hip)."""
    n, d = a.n, x2.shape[1]
    lib, st = _lib.hip(), _stream(x2)
    cache = a.__dict__.setdefault("_fused_order", {})
    if (0, n) not in cache:
        cache[(0, n)] = graph.fused_row_order(a.rowptr, 0, n)
    order, hubs, parts = cache[(0, n)]
    t_parts = None
    if hubs is not None:
        t_parts = torch.empty(parts.shape[0], d, dtype=torch.float32, device=x2.device)
        check(lib.lpf_spmm_row_parts_f32(d, ptr(parts), parts.shape[0], ptr(a.col), ptr(a.val), ptr(x2), x2.stride(0),
                                         ptr(t_parts), st), "lpf_spmm_row_parts_f32")
    out = torch.empty(n, d, dtype=torch.float32, device=x2.device)
    u = torch.empty(n, d, dtype=torch.float32, device=x2.device) if pre else None
    h = torch.empty(n, d, dtype=torch.float32, device=x2.device) if agg else None
    check(lib.lpf_gcn_layer_fused_train_f32(
        d, order.numel() // 16, ptr(order), 0, ptr(a.rowptr), ptr(a.col), ptr(a.val), ptr(x2), x2.stride(0), ptr(wp),
        ptr(out), d, ptr(bias), ptr(ln_w), ptr(ln_b), ptr(residual), 0 if residual is None else residual.stride(0), flags,
        ptr(hubs), ptr(t_parts), ptr(u), d, ptr(h), d, drop_p, drop_seed, st), "lpf_gcn_layer_fused_train_f32")
    return out, u, h


_seed_state = [None, 0]


def _next_drop_seed(device=None) -> int:
    """A fresh 64-bit seed for an in-kernel dropout mask, drawn the way torch's own dropout kernels draw their Philox
    counters: from the device's default generator -- its seed and its current offset, which is then advanced -- so that
    ``torch.manual_seed`` (re-)starts the sequence exactly as it does for torch's random operators.  Host-side state
    only: no device generator launch, no synchronisation.  (Without a device -- CPU tests -- a counter under
    ``torch.initial_seed()``.)"""
    if device is not None and torch.device(device).type == "cuda":
        dev = torch.device(device)
        gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
        base, off = gen.initial_seed(), gen.get_offset()
        gen.set_offset(off + 4)
    else:
        base = torch.initial_seed()
        if _seed_state[0] != base:
            _seed_state[0], _seed_state[1] = base, 0
        _seed_state[1] += 1
        off = _seed_state[1]
    z = (base * 0x9E3779B97F4A7C15 + off * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF    # splitmix64 finaliser
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def drop_keep_mask(seed: int, p: float, n: int, d: int, device) -> torch.Tensor:
    """The keep mask [n, d] (bool) of the in-kernel dropout for (seed, p): csrc/lpf_common.h ``lpf_drop_bits`` restated
    with torch integer arithmetic (test infrastructure and debugging; the kernels never read a mask)."""
    m32 = 0xFFFFFFFF

    def mix(h):
        h = h ^ (h >> 16); h = (h * 0x7feb352d) & m32
        h = h ^ (h >> 15); h = (h * 0x846ca68b) & m32
        return h ^ (h >> 16)
    rows = torch.arange(n, dtype=torch.int64, device=device)
    cols = torch.arange(d, dtype=torch.int64, device=device)
    rk = mix(((rows & m32) * 0x9E3779B1 + (seed & m32)) & m32) ^ (rows >> 32)
    bits = mix((rk[:, None] + ((cols * 0x85EBCA77) & m32)[None, :] + (seed >> 32)) & m32)
    t = p * 4294967296.0
    thresh = 0 if t <= 0 else (4294967295 if t >= 4294967295.0 else int(t))
    return bits >= thresh


class GcnFusedFn(torch.autograd.Function):
    """A square GCN layer WITH its Linear, dropout and skip connection,
        y = [x +] dropout(ReLU(LN(A_hat (x W^T) + b)), p)            (other_models.py:66-75)
    forward in ONE launch (lpf_gcn_layer_fused_train_f32, csrc/gcn_fused.hip: aggregate, then transform; the pre-norm
    rows u and the aggregated rows h = A_hat x are written beside the result for the backward; the dropout mask is a hash
    of (seed, row, feature), never stored).  Backward: fused dropout/ReLU/LayerNorm backward (+ bias gradient) -> du;
    dW = du^T h (no aggregation: h is the forward's); dx = [dy +] (A_hat^T du) W -- the SAME launch over the transposed
    graph with the transposed weight image, the skip connection's gradient as its residual, and no launch at all for
    the first layer, whose input takes no gradient.  (Rounds 4-5: A_hat^T du by lpf_spmm_csr_f32, two GEMMs, dropout,
    its backward and both additions as torch kernels, in every layer.)"""

    @staticmethod
    def forward(ctx, model, i, x, weight, a_hat, conv_bias, ln_w, ln_b, drop_p=0.0, skip=False):
        x2 = _rows4(x)
        seed = _next_drop_seed(x2.device) if drop_p > 0 else 0
        y, u, h = _fused_layer(a_hat, x2, model._conv_packs[i].get(weight), conv_bias, ln_w, ln_b, _lib.FLAG_RELU,
                               pre=True, agg=True, residual=x2 if skip else None, drop_p=drop_p, drop_seed=seed)
        ctx.save_for_backward(h, weight, u, ln_w, ln_b)
        ctx.a_hat, ctx.pack_t, ctx.drop, ctx.skip = a_hat, model._conv_packs_t[i], (drop_p, seed), skip
        return y

    @staticmethod
    def backward(ctx, dy):
        h, weight, u, ln_w, ln_b = ctx.saved_tensors
        dy = _rows4(dy)
        du, dg, db, dbias = _ln_relu_bwd(u, dy, ln_w, ln_b, *ctx.drop)
        dw = _gemm_tn(du, h)
        dx = None
        if ctx.needs_input_grad[2]:
            dx = _fused_layer(_transpose_csr(ctx.a_hat), du, ctx.pack_t.get(weight, transposed=True),
                              residual=dy if ctx.skip else None)[0]
        return None, None, dx, dw, None, dbias, dg, db, None, None


def layer_norm(x, weight, bias):
    """LayerNorm over the last dimension; the C-ABI kernels when the width allows (D % 4 == 0, D <= 256)."""
    d = x.shape[-1]
    if d % 4 or d > 256 or x.numel() == 0:
        return F.layer_norm(x, (d,), weight, bias)
    return LayerNormFn.apply(x, weight, bias)


def _mlp(mod, x, training: bool = True):
    """The reference's MLP (other_models.py:125-138): (Linear -> LayerNorm -> ReLU -> dropout)* -> Linear."""
    for lin in mod.linears[:-1]:
        x = linear(x, lin.weight, lin.bias)
        if mod.norm is not None and x.shape[-1] % 4 == 0 and x.shape[-1] <= 256 and x.numel() > 0:
            x = LnReluFn.apply(x, mod.norm.weight, mod.norm.bias)
        else:
            if mod.norm is not None:
                x = layer_norm(x, mod.norm.weight, mod.norm.bias)
            x = F.relu(x)
        x = F.dropout(x, p=mod.dropout, training=training)
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
    batch = model._prep_batch(batch)
    x_node = encoder_train(model, adj_prop, test_set)
    ew, pw = pair_stage(model, x_node, batch, adj_mask, test_set, training=True)
    return torch.cat([ew, pw], dim=-1)


def encoder_train(model, adj_prop=None, test_set=False):
    """The node encoder in training mode (node_encoder.py:35-44, other_models.py:61-76, link_transformer.py:127) ->
    x_node [N, D] with an autograd graph."""
    enc = model.node_encoder.gnn_encoder
    a_hat = model._device_graph("prop", model._data_obj("adj", test_set) if adj_prop is None else adj_prop)
    # ---- encoder (node_encoder.py:35-44, other_models.py:61-76, link_transformer.py:127)
    x = model._features()
    x = F.dropout(x, p=model.node_encoder.feat_drop, training=True)
    for i, conv in enumerate(enc.convs):
        if enc.lns is not None and enc.relu and model._fusable(i, x.shape[1]) and 0.0 <= float(enc.dropout) < 1.0:
            # a square layer: Linear + aggregation + bias + LayerNorm + ReLU forward in one launch
            # (a square layer is as wide as its input: the skip connection of other_models.py:72-75 applies whenever
            # ``residual`` is set)
            x = GcnFusedFn.apply(model, i, x, conv.lin.weight, a_hat, conv.bias, enc.lns[i].weight, enc.lns[i].bias,
                                 float(enc.dropout), bool(enc.residual))
            continue
        t = linear(x, conv.lin.weight, None)
        if enc.lns is not None and enc.relu and t.shape[1] % 4 == 0 and t.shape[1] <= 256:
            # aggregation + bias, LayerNorm + ReLU: two forward and two backward kernels; the dropout (a non-negative
            # mask) is applied behind the ReLU, with which it commutes
            xi = GcnLayerFn.apply(t, a_hat, conv.bias, enc.lns[i].weight, enc.lns[i].bias)
            xi = F.dropout(xi, p=enc.dropout, training=True)
        else:
            xi = SpmmFn.apply(t, a_hat) + conv.bias
            if enc.lns is not None:
                xi = layer_norm(xi, enc.lns[i].weight, enc.lns[i].bias)
            xi = F.dropout(xi, p=enc.dropout, training=True)
            if enc.relu:
                xi = F.relu(xi)
        x = x + xi if (enc.residual and x.shape[-1] == xi.shape[-1]) else xi
    x_node = layer_norm(x, model.gnn_norm.weight, model.gnn_norm.bias)
    return x_node.contiguous()


def pair_stage(model, x_node, batch, adj_mask=None, test_set=False, training: bool = True):
    """Everything behind the encoder (link_transformer.py:100-107,132-178): the elementwise branch, the selection, the
    positional encodings, the attention layer -- EVERY HEAD of it (layers.py:193-224: ``lin_l`` / ``lin_r`` produce H
    blocks of D features, head h attends with its block, its ``att`` row and its slice of the bias; the blocks are
    concatenated, :180-183) --, the count features and ``pairwise_lin`` -> (elementwise [BS, D], pairwise [BS, D]).
    ``training``: dropouts and the random attention drop on, an autograd graph is built when gradients are enabled;
    off: the evaluation-mode forward of a model the one-launch inference kernels do not cover (num_heads > 1)."""
    dev = model.device
    bs, d = batch.shape[1], model.dim
    end_sort = None
    if torch.is_grad_enabled() and x_node.requires_grad and bs > 0:
        # the endpoints sorted once: both endpoint gathers (x_a * x_b here, x_a + x_b for the queries) sum their gradients
        # node by node in this order -- with the by-node sum of the attention stage and the deterministic reductions of
        # every other kernel, a seeded training step gives the same bits every time it is run
        k, o = torch.sort(batch.reshape(-1), stable=True)
        end_sort = (k.to(torch.int32), o)
    ew = _mlp(model.elementwise_lin, PairGatherFn.apply(x_node, batch, True, end_sort), training)  # x_a * x_b (:101-102)
    # ---- selection (integer work, no gradient) in the reference's layout, then the random attention drop
    n_types = {"all": 3, "1-hop": 2, "cn": 1}[model.mask]
    with torch.no_grad():
        s = model._select(batch, test_set, adj_mask)
        tp = s["type_ptr"][:3 * (bs + 1)].view(3, bs + 1)
        tot = list(s["tot"]) if "tot" in s else [int(v) for v in tp[:, bs].tolist()]   # (read with the selection's status)
        n_all = sum(tot)
        e_pair, e_node = s["sel_pair"][:n_all], s["sel_node"][:n_all]
        e_pa, e_pb = s["sel_pa"][:n_all], s["sel_pb"][:n_all]
        if training and model.att_drop > 0 and n_all > 0:
            # drop_pairwise (link_transformer.py:322-337): ceil(n (1 - p)) entries of a random permutation survive, CN
            # and 1-hop together, >1-hop separately (:257-260).  The survivors are kept in (type, pair) order -- the
            # reference leaves them in permuted order, which only the summation order of its scatters can see.
            # (the survivors' positions sorted: sizes known on the host, no boolean mask -- five masked selections were five
            #  compactions with a host synchronisation each)
            n01 = tot[0] + tot[1]
            parts = [drop_pairwise(n01, model.att_drop, dev)]
            if tot[2] > 0:
                parts.append(n01 + drop_pairwise(tot[2], model.att_drop, dev))
            idx = torch.sort(torch.cat(parts)).values
            tid = (idx >= tot[0]).long() + (idx >= n01).long()
            e_pair, e_node, e_pa, e_pb = e_pair[idx], e_node[idx], e_pa[idx], e_pb[idx]
            # (index_add_, not bincount: torch's bincount reads the largest key back to the host first)
            cnt = torch.zeros(3 * bs, dtype=torch.int64, device=dev).index_add_(
                0, tid * bs + e_pair.long(), torch.ones(idx.numel(), dtype=torch.int64, device=dev)).view(3, bs)
        else:
            cnt = (tp[:, 1:] - tp[:, :-1]).long()
        # The DISTINCT nodes of the batch's entries (a few tens of thousands of the graph's nodes; a handful for random
        # negatives): the node half of lin_r, its two gradient products and the gradient of its row gather run over those
        # rows only, and the entries' gradients meet per node in the sorted order found here (no atomics).  Their number
        # comes back to the host with the per-type totals below: one read-back for both.
        n_kept = int(e_node.numel())
        if n_kept > 0:
            keys, order = torch.sort(e_node, stable=True)
            head = torch.ones(n_kept, dtype=torch.bool, device=dev)
            head[1:] = keys[1:] != keys[:-1]
            uid = torch.cumsum(head, 0) - 1                       # rank of the node of each sorted position
            back = torch.cat([cnt.sum(dim=1), uid[-1:] + 1]).tolist()
        else:
            back = cnt.sum(dim=1).tolist() + [0]
        tot, n_uniq = [int(v) for v in back[:3]], int(back[3])
        tbase = [0, tot[0], tot[0] + tot[1], tot[0] + tot[1] + tot[2]]
        seg = torch.zeros(3, bs + 1, dtype=torch.int64, device=dev)
        torch.cumsum(cnt, dim=1, out=seg[:, 1:])
        seg += torch.tensor(tbase[:3], dtype=torch.int64, device=dev)[:, None]
        # (copies: the export arrays live in per-stream workspaces that the next forward overwrites, and the backward
        #  pass of THIS forward still needs them)
        if n_kept > 0:
            nodes_u = torch.zeros(n_uniq, dtype=torch.int64, device=dev).scatter_(0, uid, keys.long())   # (a run writes one value)
            e_node = torch.empty(n_kept, dtype=torch.int32, device=dev)
            e_node[order] = uid.to(torch.int32)                   # entries -> rows of the compact node table
            node_sort = (uid.to(torch.int32), order)
        else:
            nodes_u = torch.zeros(1, dtype=torch.int64, device=dev)          # (a valid row for the kernels' pointers)
            e_node = e_node.to(torch.int32).clone()
            node_sort = None
        e_pa, e_pb = e_pa.clone(), e_pb.clone()
        counts = cnt.float()
    # ---- positional encodings + attention (link_transformer.py:182-211, layers.py:161-224): dedicated kernels
    encoders = [model.ppr_encoder_cn, getattr(model, "ppr_encoder_onehop", None),
                getattr(model, "ppr_encoder_non1hop", None)][:n_types]
    heads = int(model.train_args["num_heads"])
    x_u = x_node.index_select(0, nodes_u)                           # rows of the batch's DISTINCT nodes
    w1s = torch.stack([e.linears[0].weight for e in encoders])
    b1s = torch.stack([e.linears[0].bias for e in encoders])
    gams = torch.stack([e.norm.weight for e in encoders])
    bets = torch.stack([e.norm.bias for e in encoders])
    out = None
    for li, layer in enumerate(model.att_layers):
        # Every layer attends over the SAME selection and positional encodings (link_transformer.py:150-152,167-168); its
        # "edge" input is cat(x_a, x_b) for the first layer and the previous layer's output after it, halved either way
        # (layers.py:209-214): q = lin_l(e1) + lin_l(e2) = lin_l.weight (e1 + e2) + 2 lin_l.bias -- one [BS, .] product.
        att = layer.att
        c = att.att.shape[-1]                                       # out_channels: dim, or 2 dim in the first of two layers
        e_sum = PairGatherFn.apply(x_node, batch, False, end_sort) if li == 0 else out[:, :d] + out[:, d:]
        att_rows = att.att.reshape(heads, c)
        outs = []
        for h in range(heads):
            sl = slice(h * c, (h + 1) * c)
            w_r = att.lin_r.weight[sl]
            w_rx, w_rp = w_r[:, :d], w_r[:, d:]
            z = linear(x_u, w_rx, att.lin_r.bias[sl])               # node half of lin_r, once per DISTINCT node
            q = linear(e_sum, att.lin_l.weight[sl], 2.0 * att.lin_l.bias[sl])
            # the second PE Linear folded into the PE half of lin_r: k_e = Z[v] + (W_rp W2_t) h_e + W_rp (2 b2_t)
            wfold = torch.stack([linear(w_rp, e.linears[1].weight.t()) for e in encoders])
            bfold = torch.stack([linear(2.0 * e.linears[1].bias[None, :], w_rp)[0] for e in encoders])
            outs.append(PairAttentionFn.apply(z, q, att_rows[h], att.bias[sl], wfold, bfold, w1s, b1s, gams, bets, e_node,
                                              e_pa, e_pb, seg, tuple(tbase), node_sort))
        out = outs[0] if heads == 1 else torch.cat(outs, dim=1)
        out = layer_norm(out, layer.post_att_norm.weight, layer.post_att_norm.bias)
        out = F.dropout(out, p=layer.dropout, training=training)
    # ---- count features + pairwise_lin (link_transformer.py:170-177, 340-356)
    if n_types == 3:
        cf = torch.stack([counts[0], counts[1], counts[2], counts[0] + counts[1]], dim=1)
    elif n_types == 2:
        cf = torch.stack([counts[0], counts[1], counts[0] + counts[1]], dim=1)
    else:
        cf = counts[0][:, None]
    pw = _mlp(model.pairwise_lin, torch.cat([out, cf], dim=1), training)
    return ew, pw


class PairGatherFn(torch.autograd.Function):
    """x_a * x_b (``product``) or x_a + x_b of the pairs' endpoint rows (link_transformer.py:101-102, layers.py:212-215)
    through lpf_pair_gather_f32; the gradient is scattered back with float atomics (lpf_pair_scatter_add_f32)."""

    @staticmethod
    def forward(ctx, x, batch, product, end_sort=None):
        """``end_sort`` = (sorted endpoint ids cat(a, b) as int32, the permutation that sorts them): the backward then sums
        the endpoint gradients node by node in that order (``lpf_segment_rows_sum_f32``: no atomics, the same bits from run
        to run); without it they are added with float atomics (``lpf_pair_scatter_add_f32``)."""
        x = x.contiguous()
        bs, d = batch.shape[1], x.shape[1]
        out = torch.empty(bs, d, dtype=torch.float32, device=x.device)
        args = (ptr(out), d, None, 0) if product else (None, 0, ptr(out), d)
        check(_lib.hip().lpf_pair_gather_f32(bs, d, ptr(batch), batch.stride(0), x.shape[0], ptr(x), x.stride(0), *args,
                                             _stream(x)), "lpf_pair_gather_f32")
        ctx.save_for_backward(x, batch)
        ctx.product, ctx.end_sort = product, end_sort
        return out

    @staticmethod
    def backward(ctx, dout):
        x, batch = ctx.saved_tensors
        dout = dout.contiguous()
        bs, d = batch.shape[1], x.shape[1]
        dx = torch.zeros_like(x)
        if ctx.end_sort is not None and bs > 0 and d in (32, 64, 128, 256):
            keys, order = ctx.end_sort
            if ctx.product:       # d(x_a * x_b): dX[a] += dout * x_b, dX[b] += dout * x_a
                src = torch.cat([dout * x.index_select(0, batch[1]), dout * x.index_select(0, batch[0])])
            else:                 # d(x_a + x_b)
                src = torch.cat([dout, dout])
            check(_lib.hip().lpf_segment_rows_sum_f32(2 * bs, d, ptr(keys), ptr(order), ptr(src), d, ptr(dx), dx.stride(0),
                                                      _stream(x)), "lpf_segment_rows_sum_f32")
            return dx, None, None, None
        dm, ds = (dout, None) if ctx.product else (None, dout)
        check(_lib.hip().lpf_pair_scatter_add_f32(bs, d, ptr(batch), batch.stride(0), x.shape[0], ptr(x), x.stride(0),
                                                  ptr(dm), d, ptr(ds), d, ptr(dx), dx.stride(0), _stream(x)),
              "lpf_pair_scatter_add_f32")
        return dx, None, None, None


def _partial_ws(d: int, k: int, device) -> torch.Tensor:
    return torch.empty(int(_lib.hip().lpf_train_partial_blocks(0)) * k * d, dtype=torch.float32, device=device)


class PairAttentionFn(torch.autograd.Function):
    """PE hidden layer -> folded key projection -> per-pair leaky-ReLU attention with PyG's segment softmax
    (layers.py:193-224, link_transformer.py:182-211), forward and backward on the kernels of csrc/pair_train.hip plus
    lpf_gemm_f32 / lpf_gemm_tn_f32 for the entry-sized products (dH = dK Wfold, dWfold = dK^T H)."""

    @staticmethod
    def forward(ctx, z, q, att, bias, wfold, bfold, w1s, b1s, gams, bets, e_node, e_pa, e_pb, seg, tbase, node_sort=None):
        """z: rows the entries' ``e_node`` index (the node table, or -- forward_train -- its rows for the batch's distinct
        nodes); ``node_sort`` = (sorted e_node, the permutation that sorts it) when the caller has them."""
        lib, st = _lib.hip(), _stream(z)
        z, q = z.contiguous(), q.contiguous()
        att, bias = att.contiguous(), bias.contiguous()
        wfold, bfold = wfold.contiguous(), bfold.contiguous()
        w1s, b1s, gams, bets = w1s.contiguous(), b1s.contiguous(), gams.contiguous(), bets.contiguous()
        # d: width of the keys, the queries and the output (the layer's out_channels); dh: width of the PE hidden layer
        # (the model's dim) -- equal except in the first of two attention layers, whose out_channels is 2 dim
        # (link_transformer.py:55-58)
        bs, d, n, dh = q.shape[0], q.shape[1], int(e_node.numel()), w1s.shape[1]
        dev = z.device
        h = torch.empty(max(n, 1), dh, dtype=torch.float32, device=dev)
        kp = torch.empty(max(n, 1), d, dtype=torch.float32, device=dev)
        for t in range(wfold.shape[0]):
            lo, hi = tbase[t], tbase[t + 1]
            if hi <= lo:
                continue
            check(lib.lpf_pe_hidden_fwd_f32(hi - lo, dh, ptr(w1s[t]), ptr(b1s[t]), ptr(gams[t]), ptr(bets[t]),
                                            e_pa.data_ptr() + 4 * lo, e_pb.data_ptr() + 4 * lo, ptr(h[lo:]), dh, st),
                  "lpf_pe_hidden_fwd_f32")
            check(lib.lpf_gemm_f32(hi - lo, d, dh, ptr(h[lo:]), dh, ptr(wfold[t]), dh, ptr(bfold[t]), None, 0, ptr(kp[lo:]),
                                   d, 0, st), "lpf_gemm_f32")
        out = torch.empty(bs, d, dtype=torch.float32, device=dev)
        score = torch.empty(max(n, 1), dtype=torch.float32, device=dev)
        pmax = torch.empty(bs, dtype=torch.float32, device=dev)
        pinv = torch.empty(bs, dtype=torch.float32, device=dev)
        check(lib.lpf_pair_attention_train_fwd_f32(bs, n, d, ptr(seg), ptr(e_node), ptr(z), z.stride(0), ptr(kp), d, ptr(q),
                                                   d, ptr(att), ptr(bias), ptr(out), d, ptr(score), ptr(pmax), ptr(pinv),
                                                   st), "lpf_pair_attention_train_fwd_f32")
        ctx.save_for_backward(z, q, att, bias, wfold, w1s, b1s, gams, bets, e_node, e_pa, e_pb, seg, h, kp, out, score,
                              pmax, pinv)
        ctx.tbase, ctx.node_sort = tbase, node_sort
        return out

    @staticmethod
    def backward(ctx, dout):
        (z, q, att, bias, wfold, w1s, b1s, gams, bets, e_node, e_pa, e_pb, seg, h, kp, out, score, pmax,
         pinv) = ctx.saved_tensors
        lib, st, tbase = _lib.hip(), _stream(z), ctx.tbase
        dout = dout.contiguous()
        bs, d, n, dh_w = q.shape[0], q.shape[1], int(e_node.numel()), w1s.shape[1]
        dev = z.device
        dk = torch.empty(max(n, 1), d, dtype=torch.float32, device=dev)
        dz = torch.zeros_like(z)
        dq = torch.empty_like(q)
        dab = torch.empty(2, d, dtype=torch.float32, device=dev)
        check(lib.lpf_pair_attention_train_bwd_f32(
            bs, n, d, ptr(seg), ptr(e_node), ptr(z), z.stride(0), ptr(kp), d, ptr(q), d, ptr(att), ptr(bias), ptr(out), d,
            ptr(score), ptr(pmax), ptr(pinv), ptr(dout), d, ptr(dk), d, None, dz.stride(0), ptr(dq), d, ptr(dab),
            ptr(_partial_ws(d, 2, dev)), st), "lpf_pair_attention_train_bwd_f32")
        if n > 0:
            # dZ[v] = sum of dK over the entries of node v, run by run of the sorted node list (no atomics: 4 D of them per
            # entry were more than half of the kernel above; and dZ is now the same bits from run to run)
            keys, order = ctx.node_sort if ctx.node_sort is not None else torch.sort(e_node, stable=True)
            check(lib.lpf_segment_rows_sum_f32(n, d, ptr(keys), ptr(order), ptr(dk), d, ptr(dz), dz.stride(0), st),
                  "lpf_segment_rows_sum_f32")
        n_t = wfold.shape[0]
        dwfold = torch.zeros_like(wfold)
        dbfold = torch.zeros(n_t, d, dtype=torch.float32, device=dev)
        g5 = torch.zeros(n_t, 5, dh_w, dtype=torch.float32, device=dev)
        dh = torch.empty(max(n, 1), dh_w, dtype=torch.float32, device=dev)
        for t in range(n_t):
            lo, hi = tbase[t], tbase[t + 1]
            if hi <= lo:
                continue
            m = hi - lo
            wt = wfold[t].t().contiguous()                                  # dH = dK Wfold: [m, d] x [d, dh]
            check(lib.lpf_gemm_f32(m, dh_w, d, ptr(dk[lo:]), d, ptr(wt), d, None, None, 0, ptr(dh[lo:]), dh_w, 0, st),
                  "lpf_gemm_f32")
            ws = torch.empty(max(int(lib.lpf_gemm_tn_workspace_floats(m, d, dh_w)), 1), dtype=torch.float32, device=dev)
            check(lib.lpf_gemm_tn_colsum_f32(m, d, dh_w, ptr(dk[lo:]), d, ptr(h[lo:]), dh_w, ptr(dwfold[t]), dh_w,
                                             ptr(dbfold[t]), ptr(ws), st), "lpf_gemm_tn_colsum_f32")   # dWfold = dK^T H, dbfold = dK^T 1
            check(lib.lpf_pe_hidden_bwd_f32(m, dh_w, ptr(w1s[t]), ptr(b1s[t]), ptr(gams[t]), ptr(bets[t]),
                                            e_pa.data_ptr() + 4 * lo, e_pb.data_ptr() + 4 * lo, ptr(dh[lo:]), dh_w,
                                            ptr(g5[t]), ptr(_partial_ws(dh_w, 5, dev)), st), "lpf_pe_hidden_bwd_f32")
        dw1s = torch.stack([g5[:, 0], g5[:, 1]], dim=2)                    # [T, D, 2]
        return (dz, dq, dab[0], dab[1], dwfold, dbfold, dw1s, g5[:, 2], g5[:, 3], g5[:, 4], None, None, None, None,
                None, None)


def score_train(score_func, x, logits: bool = False):
    """``mlp_score.forward`` in training mode (other_models.py:173-179): (Linear, ReLU, dropout)* Linear, sigmoid
    (``logits``: without the sigmoid -- the PyG-style facade returns those)."""
    for lin in score_func.lins[:-1]:
        x = F.relu(linear(x, lin.weight, lin.bias))
        x = F.dropout(x, p=score_func.dropout, training=True)
    last = score_func.lins[-1]
    y = linear(x, last.weight, last.bias).squeeze(-1)
    return y if logits else torch.sigmoid(y)
