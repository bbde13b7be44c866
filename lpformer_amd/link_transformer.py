"""MI355X-native LPFormer link-scoring modules with the reference's module surface.

Drop-in for the reference's ``models.link_transformer.LinkTransformer`` and ``models.other_models.mlp_score``
(constructor arguments, method names, ``state_dict`` keys and return shapes; reference
src/models/link_transformer.py:16-178, src/models/other_models.py:142-179), so the reference's evaluation loops
(src/train/testing.py) run unchanged.  All arithmetic goes through the C-ABI kernels in ``liblpformer_hip.so``;
torch supplies parameter containers, device memory and streams only.  There is no CPU or eager fallback: without a
GPU and the built library every forward raises.

Scope of this round: inference (``model.eval()``; dropout is the identity).  The training step (autograd through
the fused kernels, random attention drop) is a later row of the plan and raises ``NotImplementedError``.
"""
from __future__ import annotations

import functools
import math
import weakref
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib, fold, graph, mask_delta, patterns
from . import dist as lpf_dist
from ._lib import FLAG_RELU, check, ptr
from .profile import KernelTimer


def _stream(device):
    """Raw hipStream_t of torch's current stream on ``device`` (the stream every C-ABI launch goes to)."""
    idx = device.index if isinstance(device, torch.device) and device.index is not None else torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(idx)


def _on_device(fn):
    """Run a public entry point with the model's device current: the C-ABI launches go to that device's stream, so a
    model on cuda:1 must not launch while cuda:0 is current."""
    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        idx = self.device.index if self.device.type == "cuda" else None
        if idx is None or not torch.cuda.is_available() or idx == torch.cuda.current_device():
            return fn(self, *args, **kwargs)
        with torch.cuda.device(idx):
            return fn(self, *args, **kwargs)
    return wrapper


def _glorot(t: torch.Tensor):
    bound = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-bound, bound)


def _require_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise _lib.LpfError(f"{what}: tensors must live on an MI355X (got {t.device}); "
                            "lpformer_amd has no CPU fallback")


def _pad4(k: int) -> int:
    return (k + 3) & ~3


def _as_f32_rows(x: torch.Tensor) -> torch.Tensor:
    """fp32, 2-D, unit inner stride, 16-byte aligned rows (ld % 4 == 0); copies only when needed."""
    if x.dtype != torch.float32:
        x = x.float()
    if x.dim() != 2:
        x = x.reshape(-1, x.shape[-1])
    if x.stride(1) != 1 or x.stride(0) % 4 != 0 or x.data_ptr() % 16 != 0 or x.stride(0) < x.shape[1]:
        k = x.shape[1]
        buf = torch.zeros(x.shape[0], _pad4(k), dtype=torch.float32, device=x.device)
        buf[:, :k] = x
        x = buf[:, :k]
    return x


# ------------------------------------------------------------------------------------------ kernels wrappers
def gemm(a: torch.Tensor, w: torch.Tensor, bias=None, addend=None, relu=False, out=None,
         tag="gemm") -> torch.Tensor:
    """out = a @ w.T (+bias) (+addend) (ReLU) through ``lpf_gemm_f32``.  ``a``/``w`` rows must be 16-byte aligned."""
    _require_gpu(a, "gemm")
    m, k = a.shape
    n = w.shape[0]
    assert w.shape[1] == k, (a.shape, w.shape)
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=a.device)
    with KernelTimer.span(tag):
        check(_lib.hip().lpf_gemm_f32(m, n, k, ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(addend),
                                      0 if addend is None else addend.stride(0), ptr(out), out.stride(0),
                                      FLAG_RELU if relu else 0, _stream(a.device)), "lpf_gemm_f32")
    return out


def layernorm_(x: torch.Tensor, g, b, relu=False, out=None) -> torch.Tensor:
    out = x if out is None else out
    with KernelTimer.span("layernorm"):
        check(_lib.hip().lpf_layernorm_f32(x.shape[0], x.shape[1], ptr(x), x.stride(0), ptr(g), ptr(b), ptr(out),
                                           out.stride(0), FLAG_RELU if relu else 0, _stream(x.device)),
              "lpf_layernorm_f32")
    return out


class _PaddedLinear:
    """fp32 weight with rows padded to a multiple of 4 floats (what lpf_gemm_f32 wants), refreshed when the
    parameter changes."""

    def __init__(self):
        self._key = None
        self._w = None

    def get(self, weight: torch.Tensor) -> torch.Tensor:
        key = (weight.data_ptr(), weight._version, weight.device)
        if key != self._key:
            w = weight.detach()
            if w.dtype != torch.float32 or w.shape[1] % 4 or not w.is_contiguous() or w.data_ptr() % 16:
                buf = torch.zeros(w.shape[0], _pad4(w.shape[1]), dtype=torch.float32, device=w.device)
                buf[:, :w.shape[1]] = w
                w = buf[:, :weight.shape[1]]
            self._w, self._key = w, key
        return self._w


class _PackedSquare:
    """``fold.pack_dense(W, 1)`` image of a square fp32 weight on its device (csrc/gcn_fused.hip), refreshed when the
    parameter changes."""

    def __init__(self):
        self._key = None
        self._w = None

    def get(self, weight: torch.Tensor, transposed: bool = False) -> torch.Tensor:
        """The image of ``weight`` -- or (``transposed``, the training backward: dx = (A^T du) W) of its transpose."""
        key = (weight.data_ptr(), weight._version, weight.device, transposed)
        if key != self._key:
            # fold.pack_dense(w, 1) with torch operators on the weight's own device: in training the weight changes
            # every step, and a host round trip per layer and step would cost more than the launch it feeds
            w = weight.detach().to(torch.float32)
            if transposed:
                w = w.t()
            n, k = w.shape
            nt, ng = ((n + 15) // 16 + 1) & ~1, (k + 15) // 16
            wp = torch.zeros(nt * 16, ng * 16, dtype=torch.float32, device=w.device)
            wp[:n, :k] = w
            r = wp.view(nt, 16, ng, 4, 4).permute(2, 0, 3, 1, 4).reshape(ng, nt * 256)     # stage, c, q, i, u
            per_stage = -(-(nt * 64) // 512) * 512 * 4
            img = torch.zeros(ng, per_stage, dtype=torch.float32, device=w.device)
            img[:, :nt * 256] = r
            self._w, self._key = img.reshape(-1), key
        return self._w


class DenseChain:
    """Cached device tables + launcher for ``lpf_dense_chain_f32``: Linear (+addend) (+LayerNorm) (+ReLU) (+Linear),
    refreshed when a parameter changes.  ``run`` returns None when the shape has no fused instantiation (the caller
    then takes the unfused HIP kernels)."""

    BUILT = {(2, 0), (3, 0), (4, 0), (5, 0), (8, 0), (9, 0), (16, 0), (17, 0), (32, 0), (2, 2), (4, 4), (8, 8),
             (16, 16), (3, 2), (5, 4), (9, 8), (17, 16)}
    BUILT_GATHER = {1: {(2, 0), (4, 0), (8, 0), (16, 0), (2, 2), (4, 4), (8, 8), (16, 16)},
                    2: {(2, 0), (4, 0), (8, 0), (16, 0)}}

    def __init__(self, tag: str):
        self.tag = tag
        self._key = None
        self._t = None

    @staticmethod
    def _vkey(t):
        return None if t is None else (t.data_ptr(), t._version, str(t.device))

    def tables(self, w1, b1, ln_g=None, ln_b=None, w2=None, b2=None):
        key = tuple(self._vkey(t) for t in (w1, b1, ln_g, ln_b, w2, b2))
        if key != self._key:
            npy = lambda t: None if t is None else t.detach().float().cpu().numpy()  # noqa: E731
            n1 = w1.shape[0]
            tabs = fold.dense_chain_tables(npy(w1), npy(b1) if b1 is not None else np.zeros(n1, np.float32),
                                           npy(ln_g), npy(ln_b), npy(w2),
                                           None if w2 is None else (npy(b2) if b2 is not None
                                                                    else np.zeros(w2.shape[0], np.float32)))
            dev = w1.device
            self._t = {k: torch.from_numpy(v).to(dev) for k, v in tabs.items()}
            self._t["shape"] = (w1.shape[1], n1, 0 if w2 is None else w2.shape[0])
            self._key = key
        return self._t

    def run(self, t: dict, x: torch.Tensor, relu: bool, batch=None, in_mode=0, addend=None, out=None, prob=None,
            want_logit=True, side=None):
        """x: [M, K1] rows (in_mode 0) or the node-feature matrix gathered through ``batch`` ([2, M] int64).
        ``side = (S, out)`` (gather modes): the launch also leaves ``out[m] = S[a_m] + S[b_m]`` for a second per-node
        table S (``lpf_dense_chain_side_f32``)."""
        k1, n1, n2 = t["shape"]
        nt1, dot = (n1 + 15) // 16, n2 == 1
        nt2 = 0 if (n2 == 0 or dot) else (n2 + 15) // 16
        built = self.BUILT if in_mode == 0 else self.BUILT_GATHER[in_mode]
        if (nt1, nt2) not in built or k1 % 4 or x.stride(0) % 4 or x.data_ptr() % 16 or x.stride(1) != 1:
            return None
        m = x.shape[0] if in_mode == 0 else batch.shape[1]
        dev = x.device
        if dot:
            res = torch.empty(m, dtype=torch.float32, device=dev)
            o, pr, ldo = (ptr(res), None, 0) if want_logit else (None, ptr(res), 0)
        else:
            width = n2 if n2 else n1
            if width % 4:
                return None
            res = out if out is not None else torch.empty(m, width, dtype=torch.float32, device=dev)
            o, pr, ldo = ptr(res), None, res.stride(0)
        extra, name = (), "lpf_dense_chain_f32"
        if side is not None:
            s_tab, s_out = side
            if in_mode == 0 or s_tab.shape[0] != x.shape[0] or s_tab.shape[1] % 4 or s_tab.stride(0) % 4:
                raise ValueError("side table: gather modes only, one row per row of x, a multiple of 4 wide")
            extra = (ptr(s_tab), s_tab.stride(0), s_tab.shape[1], ptr(s_out), s_out.stride(0))
            name = "lpf_dense_chain_side_f32"
        with KernelTimer.span(self.tag):
            check(getattr(_lib.hip(), name)(
                m, in_mode, ptr(x), x.stride(0), ptr(batch), 0 if batch is None else batch.stride(0),
                x.shape[0] if in_mode else 0, k1,
                ptr(t["w1p"]), n1, ptr(t["b1"]), ptr(addend), 0 if addend is None else addend.stride(0),
                ptr(t.get("ln_g")), ptr(t.get("ln_b")), FLAG_RELU if relu else 0, ptr(t.get("w2p")), n2,
                ptr(t.get("b2")), o, ldo, pr, *extra, _stream(dev)), name)
        return res


# ------------------------------------------------------------------------------------------ parameter modules
class MLP(nn.Module):
    """Same parameters and semantics as the reference's ``MLP`` (src/models/other_models.py:80-138):
    (Linear -> LayerNorm -> ReLU -> dropout)* -> Linear.  Eval-mode forward on the HIP kernels."""

    def __init__(self, num_layers, in_channels, hid_channels, out_channels, drop=0, norm="layer", sigmoid=False,
                 bias=True):
        super().__init__()
        if norm not in ("layer", None):
            raise NotImplementedError("only LayerNorm MLPs are used by LPFormer")
        self.dropout, self.sigmoid = drop, sigmoid
        self.norm = nn.LayerNorm(hid_channels) if norm == "layer" else None
        self.linears = nn.ModuleList()
        if num_layers == 1:
            self.linears.append(nn.Linear(in_channels, out_channels, bias=bias))
        else:
            self.linears.append(nn.Linear(in_channels, hid_channels, bias=bias))
            for _ in range(num_layers - 2):
                self.linears.append(nn.Linear(hid_channels, hid_channels, bias=bias))
            self.linears.append(nn.Linear(hid_channels, out_channels, bias=bias))
        self._pads = [_PaddedLinear() for _ in self.linears]
        self._chain = DenseChain("dense_chain_mlp")
        self._chain1 = DenseChain("dense_chain_mlp_hidden")  # first layer alone (LinkTransformer.score_pairs)

    def run(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, batch=None, in_mode=0) -> torch.Tensor:
        """x: [M, K] fp32 device rows (16-byte aligned).  Optionally writes the result into ``out`` (a strided view).
        ``in_mode`` 1/2 with ``batch`` [2, M]: the input rows are x[a]*x[b] / x[a]+x[b], gathered inside the kernel."""
        if self.training and self.dropout > 0:
            raise NotImplementedError("training-mode dropout is not part of the HIP inference path")
        if len(self.linears) == 2 and self.norm is not None:  # one launch: Linear -> LayerNorm -> ReLU -> Linear
            l1, l2 = self.linears
            t = self._chain.tables(l1.weight, l1.bias, self.norm.weight, self.norm.bias, l2.weight, l2.bias)
            y = self._chain.run(t, x, relu=True, batch=batch, in_mode=in_mode, out=out)
            if y is not None:
                return y
        if in_mode:
            d = x.shape[1]
            gathered = torch.empty(batch.shape[1], d, dtype=torch.float32, device=x.device)
            args = (ptr(gathered), d, None, 0) if in_mode == 1 else (None, 0, ptr(gathered), d)
            with KernelTimer.span("pair_gather"):
                check(_lib.hip().lpf_pair_gather_f32(batch.shape[1], d, ptr(batch), batch.stride(0), x.shape[0], ptr(x),
                                                     x.stride(0), *args, _stream(x.device)), "lpf_pair_gather_f32")
            x = gathered
        h = x
        for i, lin in enumerate(self.linears[:-1]):
            w = self._pads[i].get(lin.weight)
            hid = torch.empty(h.shape[0], _pad4(w.shape[0]), dtype=torch.float32, device=h.device)[:, :w.shape[0]]
            gemm(h, w, lin.bias, out=hid)
            if self.norm is not None:
                layernorm_(hid, self.norm.weight, self.norm.bias, relu=True)
            else:
                layernorm_(hid, None, None, relu=True)
            h = hid
        last = self.linears[-1]
        y = gemm(h, self._pads[-1].get(last.weight), last.bias, out=out)
        return y

    def forward(self, x):
        _require_gpu(x, "MLP.forward")
        with torch.no_grad():
            lead = x.shape[:-1]
            y = self.run(_as_f32_rows(x))
            y = y.reshape(*lead, y.shape[-1]).squeeze(-1)
            return torch.sigmoid(y) if self.sigmoid else y


class mlp_score(nn.Module):  # noqa: N801  (name kept for drop-in compatibility)
    """Score head with the reference's parameters (src/models/other_models.py:142-179): (Linear, ReLU, dropout)*
    Linear, sigmoid; returns PROBABILITIES of shape [M] like the reference.  ``logits()`` exposes the pre-sigmoid
    values."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout=0):
        super().__init__()
        self.lins = nn.ModuleList()
        if num_layers == 1:
            self.lins.append(nn.Linear(in_channels, out_channels))
        else:
            self.lins.append(nn.Linear(in_channels, hidden_channels))
            for _ in range(num_layers - 2):
                self.lins.append(nn.Linear(hidden_channels, hidden_channels))
            self.lins.append(nn.Linear(hidden_channels, out_channels))
        self.dropout = dropout
        self._pads = [_PaddedLinear() for _ in self.lins]
        self._chain = DenseChain("dense_chain_score")
        self._chain_fold = DenseChain("dense_chain_score")  # folded first layer (LinkTransformer.score_pairs)

    def _run(self, x: torch.Tensor, want_prob: bool) -> torch.Tensor:
        _require_gpu(x, "mlp_score.forward")
        if self.training and self.dropout > 0:
            raise NotImplementedError("training-mode dropout without autograd: use model.eval() for inference")
        with torch.no_grad():
            h = _as_f32_rows(x)
            if len(self.lins) == 2 and self.lins[1].out_features == 1:  # Linear -> ReLU -> dot -> sigmoid, one launch
                l1, l2 = self.lins
                t = self._chain.tables(l1.weight, l1.bias, None, None, l2.weight, l2.bias)
                res = self._chain.run(t, h, relu=True, want_logit=not want_prob)
                if res is not None:
                    return res
            for i, lin in enumerate(self.lins[:-1]):
                h = gemm(h, self._pads[i].get(lin.weight), lin.bias, relu=True)
            return self._tail(h, want_prob)

    def _tail(self, h: torch.Tensor, want_prob: bool) -> torch.Tensor:
        """Last Linear (+ sigmoid) on the hidden activations ``h`` (unfused kernels)."""
        last = self.lins[-1]
        if last.out_features == 1:
            res = torch.empty(h.shape[0], dtype=torch.float32, device=h.device)
            w = last.weight.detach().reshape(-1).contiguous()
            b = 0.0
            if last.bias is not None:  # scalar kernel argument: read it back only when the parameter changes
                key = (last.bias.data_ptr(), last.bias._version)
                if getattr(self, "_bias_key", None) != key:
                    self._bias_key, self._bias_val = key, float(last.bias.detach().item())
                b = self._bias_val
            check(_lib.hip().lpf_rowdot_sigmoid_f32(h.shape[0], h.shape[1], ptr(h), h.stride(0), ptr(w), b,
                                                    None if want_prob else ptr(res),
                                                    ptr(res) if want_prob else None, _stream(h.device)),
                  "lpf_rowdot_sigmoid_f32")
            return res
        y = gemm(h, self._pads[-1].get(last.weight), last.bias)
        return torch.sigmoid(y).squeeze(-1) if want_prob else y.squeeze(-1)

    def forward(self, x):
        if self.training and torch.is_grad_enabled():
            from . import train as lpf_train
            return lpf_train.score_train(self, x)
        return self._run(x, True)

    def logits(self, x):
        return self._run(x, False)


class _GCNConvParams(nn.Module):
    """Parameter container named like PyG's GCNConv: ``lin.weight`` (no bias, glorot) and ``bias`` (zeros)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        _glorot(self.lin.weight)


class GCN(nn.Module):
    """Parameters of the reference's ``GCN`` (src/models/other_models.py:10-76); evaluated by LinkTransformer.propagate."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout, residual=False, cached=False,
                 normalize=True, layer_norm=True, relu=True):
        super().__init__()
        if not normalize:
            raise NotImplementedError("GCN without normalisation is not used by LPFormer")
        if not layer_norm and num_layers >= 3:
            # the reference constructor crashes here (self.lns is None, other_models.py:41,46); keep it an error
            raise ValueError("layer_norm=False requires gnn_layers <= 2 (as in the reference)")
        if num_layers == 1:
            hidden_channels = out_channels
        self.relu, self.dropout, self.residual = relu, dropout, residual
        self.convs = nn.ModuleList([_GCNConvParams(in_channels, hidden_channels)])
        self.lns = nn.ModuleList([nn.LayerNorm(hidden_channels)]) if layer_norm else None
        if num_layers > 1:
            for _ in range(num_layers - 2):
                self.convs.append(_GCNConvParams(hidden_channels, hidden_channels))
                self.lns.append(nn.LayerNorm(hidden_channels))
            self.convs.append(_GCNConvParams(hidden_channels, out_channels))
            if layer_norm:
                self.lns.append(nn.LayerNorm(hidden_channels))


class NodeEncoder(nn.Module):
    """Parameters of the reference's ``NodeEncoder`` (src/modules/node_encoder.py:8-44), including the unused
    ``feat_transform`` Linear that lives in its state_dict."""

    def __init__(self, data, train_args, device="cuda"):
        super().__init__()
        self.dim = train_args["dim"]
        init_dim = self.dim if "emb" in data else data["x"].size(1)
        self.feat_drop = train_args.get("feat_drop", 0)
        self.feat_transform = nn.Linear(init_dim, self.dim)
        self.gnn_encoder = GCN(init_dim, self.dim, self.dim, train_args["gnn_layers"], train_args.get("gnn_drop", 0),
                               cached=train_args.get("gcn_cache"), residual=train_args["residual"],
                               layer_norm=train_args["layer_norm"], relu=train_args["relu"])


class LinkAttention(nn.Module):
    """Parameters of the reference's ``LinkAttention`` (src/modules/layers.py:88-157): lin_l [H*C, D], lin_r
    [H*C, 2D], att [1, H, C], bias [H*C]."""

    def __init__(self, in_channels, out_channels, train_args, node_dim=None):
        super().__init__()
        self.heads = train_args["num_heads"]
        nd = (in_channels if node_dim is None else node_dim) * 2
        self.lin_l = nn.Linear(in_channels, self.heads * out_channels)
        self.lin_r = nn.Linear(nd, self.heads * out_channels)
        self.att = nn.Parameter(torch.empty(1, self.heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(self.heads * out_channels))
        _glorot(self.lin_l.weight)
        _glorot(self.lin_r.weight)
        _glorot(self.att)


class LinkTransformerLayer(nn.Module):
    """Parameters of the reference's ``LinkTransformerLayer`` (src/modules/layers.py:17-82)."""

    def __init__(self, dim, train_args, out_dim=None, node_dim=None):
        super().__init__()
        self.dropout = train_args.get("dropout", 0)
        out_dim = dim if out_dim is None else out_dim
        self.att = LinkAttention(dim, out_dim, train_args, node_dim=node_dim)
        self.post_att_norm = nn.LayerNorm(out_dim * train_args["num_heads"])


class _SelectWorkspace:
    """Per (stream, batch size) buffers of the selection kernels (include/lpformer_hip.h, "Selection, second
    generation").  ``ctl`` and the chained-scan words persist across launches (epoch-tagged, never cleared)."""

    def __init__(self, device, bs: int):
        self.device, self.bs = device, bs
        self.plan_blocks = int(_lib.hip().lpf_select_plan_blocks(bs))
        self.ctl = torch.zeros(16, dtype=torch.int64, device=device)  # LPF_SELECT_CTL_WORDS
        self.desc = torch.empty(16 * max(bs, 1), dtype=torch.int64, device=device)
        self.offs = torch.empty(bs + 1, dtype=torch.int64, device=device)
        self.plan_lb = torch.zeros(self.plan_blocks + 1, dtype=torch.int64, device=device)
        self.type_ptr = torch.zeros(3 * (bs + 1), dtype=torch.int32, device=device)
        self.item_cap = self.ent_cap = 0
        self.item_pair = self.run_lb = self.entries = None
        self.calibrated = False

    def ensure(self, item_cap: int, ent_cap: int, shrink: bool = False):
        if item_cap > self.item_cap:
            self.item_cap = int(item_cap)
            self.item_pair = torch.empty(self.item_cap, dtype=torch.int32, device=self.device)
            self.run_lb = torch.zeros(3 * self.item_cap, dtype=torch.int64, device=self.device)
        if ent_cap > self.ent_cap or (shrink and ent_cap < self.ent_cap // 2):
            self.ent_cap = int(ent_cap)
            self.entries = None  # release before allocating the new size
            self.entries = torch.empty(3 * self.ent_cap * 4, dtype=torch.int32, device=self.device)

    def read_status(self, stream=None):
        """(error bits, [n_cn, n_1hop, n_non1hop]) of the last batch.  The read is ordered after everything queued on
        ``stream`` (a ``torch.cuda.Stream``; default: the current one): the copy is issued ON that stream and waited
        for, so a selection kernel still in flight there cannot be read as "ok"."""
        if stream is None:
            v = self.ctl.tolist()
        else:
            with torch.cuda.stream(stream):
                v = self.ctl.tolist()
        return int(v[3]), [int(v[4]), int(v[5]), int(v[6])]

    def clear_errors(self, stream=None):
        """Clears the sticky bits on ``stream`` (default: the current one) -- the stream whose kernels raise them, so
        the write cannot race with a kernel's atomicOr."""
        if stream is None:
            self.ctl[3] = 0
        else:
            with torch.cuda.stream(stream):
                self.ctl[3] = 0


class _Select4Workspace:
    """Per (stream, batch size) buffers of the one-launch, pair-major selection (``lpf_select4``,
    include/lpformer_hip.h): ``pair_tab`` int32 [bs, 4] = {first entry, n_cn, n_1hop, n_non1hop}, ``blk_cnt`` int32
    [ceil(bs / 64)], ``entries`` 16-byte records.  ``ctl`` persists across launches (word 3: sticky error bits; word 0:
    the room the last batch needed)."""

    def __init__(self, device, bs: int):
        self.device, self.bs = device, bs
        self.ctl = torch.zeros(32, dtype=torch.int64, device=device)  # LPF_SELECT4_CTL_WORDS
        self.pair_tab = torch.zeros(4 * max(bs, 1), dtype=torch.int32, device=device)
        self.blk_cnt = torch.zeros(2 * ((bs + _lib.SELECT4_BLOCK - 1) // _lib.SELECT4_BLOCK) + 2, dtype=torch.int32,
                                   device=device)   # {entries, pairs with entries} per block
        self.ent_cap = 0
        self.kept_cap = 0       # selected entries the attention's scratch is sized for (its units of 16)
        self.entries = None
        self.calibrated = False

    def ensure(self, ent_cap: int, shrink: bool = False):
        if ent_cap > self.ent_cap or (shrink and ent_cap < self.ent_cap // 2):
            self.ent_cap = int(ent_cap)
            self.entries = None  # release before allocating the new size
            self.entries = torch.empty(self.ent_cap * 4, dtype=torch.int32, device=self.device)

    def read_status(self, stream=None):
        """(error bits, [entries the last batch needed room for]); the read is ordered behind ``stream`` (see
        ``_SelectWorkspace.read_status``)."""
        if stream is None:
            v = self.ctl.tolist()
        else:
            with torch.cuda.stream(stream):
                v = self.ctl.tolist()
        return int(v[3]), [int(v[0])]

    def clear_errors(self, stream=None):
        if stream is None:
            self.ctl[3] = 0
        else:
            with torch.cuda.stream(stream):
                self.ctl[3] = 0

    def kept(self) -> int:
        """Selected entries of the last batch (synchronises)."""
        return int(self.blk_cnt[:-2:2].sum().item())


class _Select4RegionsWorkspace(_SelectWorkspace):
    """``lpf_select4`` + ``lpf_select4_regions``: the one-launch selection for consumers of the TYPE-MAJOR form (the
    matrix-core attention, the record-merging tail, ``lpf_select_export``).  Looks like a ``_SelectWorkspace`` to them
    (``type_ptr``, ``entries``, ``ent_cap``, ``ctl`` with the sticky bits in word 3 and the totals per type in words
    4-6); the pair-major half (``pair_tab``, ``blk_cnt``, ``blk_types``, ``entries4``) stays inside."""

    def __init__(self, device, bs: int):
        self.device, self.bs = device, bs
        nblk = (bs + _lib.SELECT4_BLOCK - 1) // _lib.SELECT4_BLOCK
        self.ctl = torch.zeros(32, dtype=torch.int64, device=device)            # LPF_SELECT4_CTL_WORDS
        self.pair_tab = torch.zeros(4 * max(bs, 1), dtype=torch.int32, device=device)
        self.blk_cnt = torch.zeros(2 * nblk + 2, dtype=torch.int32, device=device)
        self.blk_types = torch.zeros(4 * nblk + 4, dtype=torch.int32, device=device)
        self.type_ptr = torch.zeros(3 * (bs + 1), dtype=torch.int32, device=device)
        self.ent_cap = self.ent_cap4 = self.item_cap = 0
        self.entries = self.entries4 = self.item_pair = self.run_lb = None
        self.calibrated = False

    def ensure4(self, ent_cap4: int, shrink: bool = False):
        if ent_cap4 > self.ent_cap4 or (shrink and ent_cap4 < self.ent_cap4 // 2):
            self.ent_cap4 = int(ent_cap4)
            self.entries4 = None
            self.entries4 = torch.empty(self.ent_cap4 * 4, dtype=torch.int32, device=self.device)

    def ensure(self, item_cap: int = 0, ent_cap: int = 0, shrink: bool = False):
        if ent_cap > self.ent_cap or (shrink and ent_cap < self.ent_cap // 2):
            self.ent_cap = int(ent_cap)
            self.entries = None
            self.entries = torch.empty(3 * self.ent_cap * 4, dtype=torch.int32, device=self.device)


# ------------------------------------------------------------------------------------------ the model
class LinkTransformer(nn.Module):
    """LPFormer link-representation model on MI355X.

    ``LinkTransformer(train_args, data, device)`` with the reference's ``train_args`` keys (thresh_cn, thresh_1hop,
    thresh_non1hop, dim, trans_layers, num_heads, att_drop, dropout, gnn_drop, feat_drop, gcn_cache, gnn_layers,
    residual, layer_norm, relu) and ``data`` dict (x, adj_t, full_adj_t, adj_mask, full_adj_mask, ppr, ppr_test).
    Graph entries may be torch sparse tensors, torch_sparse.SparseTensor-like objects, scipy matrices or
    ``lpformer_amd.graph.CSR``; they are converted to device CSR once and cached.

    Methods mirror the reference: ``forward`` (:82-107), ``propagate`` (:110-129), ``calc_pairwise`` (:132-178),
    ``compute_node_mask`` (:214-276), attributes ``out_dim`` and ``elementwise_lin``.
    """

    def __init__(self, train_args, data, device="cuda"):
        super().__init__()
        self.train_args, self.data, self.device = train_args, data, torch.device(device)
        self.thresh_cn = train_args["thresh_cn"]
        self.thresh_1hop = train_args["thresh_1hop"]
        self.thresh_non1hop = train_args["thresh_non1hop"]
        if self.thresh_non1hop == 1 and self.thresh_1hop == 1:
            self.mask = "cn"
        elif self.thresh_non1hop == 1 and self.thresh_1hop < 1:
            self.mask = "1-hop"
        else:
            self.mask = "all"
        self.dim = train_args["dim"]
        self.att_drop = train_args.get("att_drop", 0)
        self.num_layers = train_args["trans_layers"]
        self.num_nodes = data["x"].shape[0]
        self.out_dim = self.dim * 2

        self.gnn_norm = nn.LayerNorm(self.dim)
        self.node_encoder = NodeEncoder(data, train_args, device=device)
        self.att_layers = nn.ModuleList()
        inner = self.dim * 2 if self.num_layers > 1 else self.dim
        self.att_layers.append(LinkTransformerLayer(self.dim, train_args, out_dim=inner))
        for _ in range(self.num_layers - 2):
            self.att_layers.append(LinkTransformerLayer(self.dim, train_args, node_dim=self.dim))
        if self.num_layers > 1:
            self.att_layers.append(LinkTransformerLayer(self.dim, train_args, out_dim=self.dim, node_dim=self.dim))
        self.elementwise_lin = MLP(2, self.dim, self.dim, self.dim)
        self.ppr_encoder_cn = MLP(2, 2, self.dim, self.dim)
        if self.mask == "cn":
            count_dim = 1
        elif self.mask == "1-hop":
            self.ppr_encoder_onehop = MLP(2, 2, self.dim, self.dim)
            count_dim = 3
        else:
            count_dim = 4
            self.ppr_encoder_onehop = MLP(2, 2, self.dim, self.dim)
            self.ppr_encoder_non1hop = MLP(2, 2, self.dim, self.dim)
        self.count_dim = count_dim
        pairwise_dim = self.dim * train_args["num_heads"] + count_dim
        self.pairwise_lin = MLP(2, pairwise_dim, pairwise_dim, self.dim)

        # runtime state (not parameters)
        self._graphs = {}      # (kind, id(obj)) -> (obj, DeviceCSR) for the graphs held in self.data
        self._override = {}    # kind -> (obj, DeviceCSR): the LAST caller-supplied override only
        self._folded = None    # (param version key, dict of device tensors)
        self._z_cache = None   # (key, Z)
        self._y_cache = None   # (key, Y): the per-node query table, built lazily (query_from = "table")
        self._x_cache = None   # (key, padded features)
        self._ws = {}          # named workspaces
        self._param_list = None  # cached list(self.parameters()) for the fold key
        self._fold_memo = None   # the folded tables for the duration of one score_pairs call
        self._chain_att = DenseChain("dense_chain_attn_out")   # attention output projection + post_att_norm
        self._chain_q = DenseChain("pair_q")                   # q = lin_l(x_a) + lin_l(x_b)
        self._conv_pads = [_PaddedLinear() for _ in self.node_encoder.gnn_encoder.convs]
        self._conv_packs = [_PackedSquare() for _ in self.node_encoder.gnn_encoder.convs]
        self._conv_packs_t = [_PackedSquare() for _ in self.node_encoder.gnn_encoder.convs]   # (W^T: train.py)
        self.query_from = "table"              # "table" or "gemm": see _pair_q
        # square GCN layers (in = out = D <= 128) in one launch, aggregate-then-transform (csrc/gcn_fused.hip); False:
        # always lpf_gemm_f32 + lpf_spmm_csr_f32
        self.encoder_fused = True
        self.last_stats = {}
        self._shard = (0, 1)   # (rank, world)
        self.encoder_mode = "sharded"  # with world > 1: "sharded" (rows + all-gather per layer) or "replicated"
        self.use_select_index = True  # False: always run the general (PPR-streaming) selection kernel
        # an adjacency override that is the model's own typing adjacency minus a few edges (the training loop's masked
        # adjacency, src/train/train_model.py:38-46) is taken as that DIFFERENCE: selection over the resident walk indexes,
        # then a patch over the entries that touch a removed edge (lpformer_amd/mask_delta.py) -- no CSR is built from the
        # override.  False: every override becomes a graph of its own on the general selection path (rounds 2-5)
        self.use_mask_delta = True
        self.mask_delta_limit = 1 << 20     # removed directed edges above which an override is a graph of its own
        self._delta_cache = None            # (override object, test_set, removed keys or None): the LAST override only
        self.select_grid = 0           # workgroups of lpf_select3_run (0 = as many as are resident at once)
        # the selection in front of the pair-major attention: True = ONE launch leaving pair-major entries + a table entry
        # per pair (csrc/select4.hip: blocks of 64 pairs, no plan launch, no chained scan); False = lpf_select3_plan / _run
        # (type-major regions, which lpf_select_export and the record-writing attention kernels keep using)
        self.select_blocks = True
        self.select4_threads = 0       # threads per workgroup of lpf_select4 (0 = default)
        # consumers of the TYPE-MAJOR form (D < 128: the matrix-core attention; every model: compute_node_mask, the
        # training step, the attention weights) behind lpf_select4 + lpf_select4_regions instead of lpf_select3_plan /
        # _run: a block of 64 pairs occupies ONE workgroup for as long as its own walks take instead of the whole chip
        # for as long as the batch's do -- about half the CU-time on hub-heavy batches (ppa-like, citation2-like)
        self.select4_regions = True
        # the elementwise branch and the q projection only need X and the batch: they run on a second HIP stream
        # underneath the (latency/issue-bound) selection kernels.  False: everything on the caller's stream.
        self.use_side_stream = True
        self._side = None
        self.use_tail_chain = True    # score_pairs: lpf_tail_chain_f32 instead of three dense-chain launches
        self.use_fused_attention = True  # one-pass attention on the selection regions
        # fp32 one-pass attention: "flip" = the key projection of the positional encoding evaluated through the ReLU
        # activation pattern of its hidden layer (csrc/pair_flip.hip: O(D) per entry + O(D) per unit that left the
        # pattern of (0, 0); gather-bound), "mfma" = the D x D product per entry on the fp32 matrix cores
        # (csrc/pair_fused.hip; its cost does not depend on the weights).  Same records, same consumers.
        self.attention_impl = "auto"   # "auto": "flip" for D >= 128, "mfma" below (at D = 64 the matrix-core kernel's
        #                                2 D^2 FLOP per entry take 22 us on the ppa-like batch, the flip kernel 33)
        # the "flip" arithmetic PAIR-MAJOR (csrc/pair_rows.hip): a group of lanes walks all entries of a pair and leaves
        # its finished row [post_att_norm(attention output) | counts] -- no records for the tail to chase and merge.
        # False: the unit-major kernel (csrc/pair_flip.hip) + record merge, as the matrix-core kernel always does.
        self.attention_rows = True
        self.flip_recheck_every = 16   # "auto": parameter versions between two looks at the weights (attention_kernel)
        self._refolds = 0
        # behind the pair-major kernel: hand the dense tail the pairs with selected nodes first and let the workgroups that
        # see only pairs without any (their attention branch is a constant) run the elementwise half of the head alone
        self.tail_skip_empty = True
        # "f32" (parity mode, logits within 1e-4 of the reference) or "bf16" (throughput mode of score_pairs: the node
        # table Z is stored in bf16 and Wfold h runs on the bf16 matrix cores; selection and everything else as in f32)
        self.precision = "f32"
        # "f32" or "bf16": in bf16 the per-layer table X W^T that the aggregation gathers (and that a row-sharded
        # encoder all-gathers) is stored in bf16; the GEMM, the sums over neighbours and the epilogue stay fp32
        self.encoder_precision = "f32"
        # forward() in eval mode keeps the encoder output while none of its inputs has changed (_propagate_reusing)
        self.reuse_encoder_output = True
        self._enc_cache = None
        # "f32" or "bf16": in bf16 the two GEMMs of the dense tail (first layer of pairwise_lin, folded score head) run
        # on the bf16 matrix cores with bf16 weights and activations rounded to bf16 (fp32 accumulate; record merge,
        # LayerNorms, dot product and sigmoid stay fp32); logits within 5e-3 of fp32 (observed <= 1e-3).
        self.tail_precision = "f32"

    # ---------------------------------------------------------------------------------- support checks
    @property
    def _multi_head(self) -> bool:
        """num_heads > 1 (layers.py:129-135,180-224): the head-by-head path of lpformer_amd/train.py ``pair_stage`` -- the
        training step's kernels, one pass per head over the shared selection -- serves evaluation too; the one-launch
        inference kernels (select4 / pair_rows / tail_chain, the folded score head, recorded plans) are single-head."""
        return int(self.train_args["num_heads"]) > 1 or self.num_layers > 1      # (two layers take the same path)

    def _check_supported(self, train_ok: bool = False, heads_ok: bool = False):
        if self.num_layers > 2 or (self.num_layers == 2 and (int(self.train_args["num_heads"]) != 1 or self.dim > 128)):
            raise NotImplementedError("trans_layers = 1, or 2 with num_heads = 1 and dim <= 128: the reference's layer "
                                      "stack is shape-consistent for nothing else (link_transformer.py:55-62: the first "
                                      "of two layers is 2 dim wide, its halves feed the second)")
        if self._multi_head and not heads_ok:
            raise NotImplementedError("num_heads > 1 / trans_layers = 2 run through forward / calc_pairwise / pair_features "
                                      "/ score_pairs (layer by layer, head by head); this entry point is built on the "
                                      "one-layer, one-head inference kernels")
        if self.dim not in (32, 64, 128, 256):
            raise NotImplementedError("dim must be one of 32, 64, 128, 256 for the gfx950 kernels")
        if self.training and not train_ok:
            raise NotImplementedError("in training mode only forward() is available (the reference's training loop calls "
                                      "nothing else, src/train/train_model.py:59-66); the inference kernels behind "
                                      "propagate / calc_pairwise / score_pairs need model.eval()")
        if "emb" in self.data:
            raise NotImplementedError("data['emb'] is never set by the reference's readers and is not supported")

    # ---------------------------------------------------------------------------------- graph state
    def _data_obj(self, kind: str, test_set: bool):
        if kind == "ppr":
            return self.data["ppr_test"] if (test_set and "ppr_test" in self.data) else self.data["ppr"]
        suffix = "mask" if kind == "mask" else "t"
        return self.data[f"full_adj_{suffix}"] if test_set else self.data[f"adj_{suffix}"]

    def _in_data(self, obj) -> bool:
        return any(obj is v for v in self.data.values())

    def _device_graph(self, kind: str, obj) -> graph.DeviceCSR:
        """kind in {'prop' (GCN-normalised), 'mask', 'ppr', 't0' (blocked >1-hop index of the general selection path)}.
        Graphs held in ``self.data`` are converted once and stay resident.  A caller-supplied override (the training
        loop passes a fresh ``adj_mask`` / ``adj_prop`` per batch, src/train/train_model.py:40-59) only occupies ONE
        slot per kind: the previous override's device copy is released when the next one arrives."""
        persistent = self._in_data(obj)
        if persistent:
            key = (kind, id(obj))
            hit = self._graphs.get(key)
        else:
            key = None
            hit = self._override.get(kind)
        if hit is not None and hit[0] is obj:
            return hit[1]
        dev = self.device
        if isinstance(obj, graph.RemovedEdges) and kind != "prop":
            raise TypeError("RemovedEdges describes a difference to the model's own typing adjacency (adj_mask) or "
                            "propagation matrix (adj_prop), nothing else")
        if kind == "prop" and not persistent:
            g = self._prop_delta(obj)
            if g is not None:
                self._override[kind] = (obj, g)
                return g
            if isinstance(obj, graph.RemovedEdges):
                raise ValueError("adj_prop=RemovedEdges(...) needs the model's own data['adj_t'] resident with its raw "
                                 "weights (a graph this model normalised itself)")
        if kind == "t0":    # per-model indexes: filtered on the device from the resident PPR matrix
            g = graph.ppr_filter_device_blocked(self._device_graph("ppr", obj), 0, self.thresh_non1hop)
        elif isinstance(obj, graph.CSR) and kind in ("mask", "ppr"):
            if kind == "ppr" and obj.val is None:
                raise ValueError("the PPR matrix needs values")
            # this package's own containers are sorted and coalesced already: upload as they are
            g = (obj if kind == "ppr" else graph.CSR(obj.rowptr, obj.col, None, obj.n)).to_device(dev)
        elif kind in ("prop", "mask") and graph.as_coo_device(obj, dev) is not None:
            # a graph that already lives on this GPU (the training loop's per-batch masked adjacency): stay there
            row, col, val, n = graph.as_coo_device(obj, dev)
            if kind == "prop":
                g = graph.gcn_norm_device(graph.gcn_structure_csr_device(row, col, val, n))
            else:
                g = graph.csr_from_coo_device(row, col, None, n)
        else:
            row, col, val, n = graph.as_coo_numpy(obj)
            if kind == "prop":
                g = graph.gcn_norm_device(graph.gcn_structure_csr(np.stack([row, col]), val, n).to_device(dev))
            elif kind == "mask":
                g = graph.csr_from_coo(row, col, None, n).to_device(dev)
            else:
                if val is None:
                    raise ValueError("the PPR matrix needs values")
                g = graph.csr_from_coo(row, col, val, n).to_device(dev)
        if g.n != self.num_nodes:
            raise ValueError(f"graph has {g.n} nodes, the model {self.num_nodes}")
        if persistent:
            self._graphs[key] = (obj, g)
        else:
            self._override[kind] = (obj, g)
        return g

    def _features(self) -> torch.Tensor:
        x = self.data["x"]
        hit = self._x_cache
        if hit is None or hit[0] is not x or hit[1] != x._version:  # identity, not address (see _node_keys)
            hit = self._x_cache = (x, x._version, _as_f32_rows(x.detach().to(self.device)))
        return hit[2]

    def _workspace(self, name: str, numel: int, dtype, st=None) -> torch.Tensor:
        """Named scratch buffer of the CURRENT stream (callers may pipeline batches on several streams; each stream
        owns its own set, allocated under that stream so the caching allocator orders its reuse correctly)."""
        key = (name, st if st is not None else _stream(self.device))
        t = self._ws.get(key)
        if t is None or t.numel() < numel or t.dtype != dtype:
            grow = int(numel * 1.25) + 64
            t = torch.empty(grow, dtype=dtype, device=self.device)
            self._ws[key] = t
        return t

    def _zero_workspace(self, name: str, numel: int, st=None) -> torch.Tensor:
        """``_workspace`` whose memory is zero when it is (re)allocated: for buffers with padding columns that the
        kernels never write and the consumers read."""
        key = (name, st if st is not None else _stream(self.device))
        t = self._ws.get(key)
        if t is None or t.numel() < numel:
            t = self._ws[key] = torch.zeros(int(numel * 1.25) + 64, dtype=torch.float32, device=self.device)
        return t[:numel]

    # ---------------------------------------------------------------------------------- folded weights
    def _fold(self):
        if self._fold_memo is not None:   # (inside one score_pairs call: the ~40 version counters were just walked)
            return self._fold_memo
        if self._param_list is None:
            self._param_list = list(self.parameters())
        key = tuple((p.data_ptr(), p._version) for p in self._param_list)
        if self._folded is not None and self._folded[0] == key:
            return self._folded[1]
        sd = {k: v for k, v in self.state_dict().items()}
        n_types = {"all": 3, "1-hop": 2, "cn": 1}[self.mask]
        out = fold.fold_attention(sd, self.dim, n_types)
        out["pe_tab"], out["pe_stat"] = fold.pe_tables(sd, self.dim, n_types)
        out["flip_tab"], out["flip_base"], _, out["wfold_t"] = fold.flip_tables(sd, self.dim, n_types)
        # pe_stat[t][7]: the square of PPR value pairs inside which no unit flips (the pair-major kernel skips the look at
        # an entry's units there; the unit-major kernels never read the slot)
        out["pe_stat"] = out["pe_stat"].copy()
        for t in range(n_types):
            out["pe_stat"][t, 7] = fold.no_flip_radius(out["flip_tab"][t], out["pe_stat"][t])
        dev = {k: torch.from_numpy(np.ascontiguousarray(v)).to(self.device) for k, v in out.items()}
        self._folded = (key, dev)
        self._refolds = getattr(self, "_refolds", 0) + 1
        self._z_cache = None
        self._y_cache = None      # (Y = X W_l^T + b_l is parameter-derived too: a stale Y would give stale queries)
        self._zb_cache = None
        self._chain_att._key = None
        return dev

    # ---------------------------------------------------------------------------------- encoder
    @_on_device
    def propagate(self, adj=None, test_set=False, _layers_out=None):
        """GCN encoder + ``gnn_norm`` -> [N, D] (reference :110-129).  L x (MFMA GEMM, fused CSR SpMM).
        With more than one rank (``set_row_shard``) the encoder is either REPLICATED (every rank runs all of it, no
        exchange) or ROW-SHARDED: a rank transforms and aggregates only its block of node rows and the transformed
        rows are all-gathered once per layer (RCCL over xGMI), plus one all-gather of the output.
        ``_layers_out`` (tests): a list that receives every layer's input and the final output."""
        self._check_supported(heads_ok=True)
        with torch.no_grad():
            a_hat = self._device_graph("prop", self._data_obj("adj", test_set) if adj is None else adj)
            n_layers = len(self.node_encoder.gnn_encoder.convs)
            rank, world = self._shard
            if (world == 1 and not lpf_dist.FORCE_COLLECTIVES) or self.encoder_mode == "replicated":
                x = self._features()
                for i in range(n_layers):
                    if _layers_out is not None:
                        _layers_out.append(x)
                    x = self._layer(i, a_hat, x, 0, self.num_nodes)
            elif self.encoder_mode == "gather_once":
                # BASELINE.json's literal layout -- "a single RCCL all-gather of node embeddings after the encoder":
                # layers 1..L-1 run on every rank (no exchange), the LAST layer's aggregation + epilogue and the two
                # per-node projection of the attention (Z: _node_keys) runs on the rank's row block only, and ONE
                # all-gather of [X_node | Z] rows (2 D floats per node) hands every rank both tables.  A fused
                # last layer (csrc/gcn_fused.hip) is row-sharded whole; an unfused one keeps its X W^T replicated (a
                # row's aggregation reads the transformed rows of all its neighbours).  Every element stays bitwise
                # equal to the unsharded encoder.
                x = self._features()
                for i in range(n_layers - 1):
                    x = self._layer(i, a_hat, x, 0, self.num_nodes)
                lo, hi = lpf_dist.row_range(self.num_nodes, world, rank)
                w = self._fold()
                d = self.dim
                pack = torch.empty(hi - lo, 2 * d, dtype=torch.float32, device=self.device)
                x_rows = self._layer(n_layers - 1, a_hat, x, lo, hi)
                pack[:, :d] = x_rows
                gemm(x_rows, w["w_rx"], w["b_r"], out=pack[:, d:], tag="gemm_node_keys")  # Z of the rank's rows
                full = lpf_dist.allgather_rows(pack, self.num_nodes)      # the all-gather of node embeddings
                x = full[:, :d]
                torch.cuda.current_stream(self.device).synchronize()      # other streams read Z (as _node_keys)
                self._z_cache = (weakref.ref(x), x._version, full[:, d:])
            else:
                # row-sharded: a rank produces its block of rows of every layer.  A fused layer (aggregate, then
                # transform) reads the layer INPUT of all nodes -- the features for layer 0 (replicated: no exchange),
                # an all-gather of the previous layer's rows after that; an unfused one transforms the rank's rows
                # and all-gathers the transformed rows.  Either way one all-gather per layer, and one of the output.
                lo, hi = lpf_dist.row_range(self.num_nodes, world, rank)
                x_full = self._features()
                x = x_full[lo:hi]                                # the rank's rows of the layer input
                for i in range(n_layers):
                    if self._fusable(i, x.shape[1]):
                        if x_full is None:
                            x_full = lpf_dist.allgather_rows(x, self.num_nodes)
                        x = self._layer_fused(i, a_hat, _as_f32_rows(x_full), lo, hi)
                    else:
                        t = lpf_dist.allgather_rows(self._layer_transform(i, x), self.num_nodes)   # [N, D] everywhere
                        x = self._layer_aggregate(i, a_hat, t, lo, hi, x)
                    x_full = None
                x = lpf_dist.allgather_rows(x, self.num_nodes)  # the all-gather of node embeddings
            if _layers_out is not None:
                _layers_out.append(x)
            return x

    def _layer(self, i: int, a_hat: graph.DeviceCSR, x: torch.Tensor, lo: int, hi: int) -> torch.Tensor:
        """Rows [lo, hi) of layer i's output from the layer input ``x`` of ALL nodes: one launch when the layer is
        square and small enough for the fused kernel, transform + aggregate otherwise."""
        x = _as_f32_rows(x)
        if self._fusable(i, x.shape[1]):
            return self._layer_fused(i, a_hat, x, lo, hi)
        return self._layer_aggregate(i, a_hat, self._layer_transform(i, x), lo, hi, x[lo:hi])

    def _fusable(self, i: int, in_dim: int) -> bool:
        """Layer i as ONE launch (csrc/gcn_fused.hip)?  Square weight, D in {32, 64, 128} (bf16 table: 64, 128)."""
        d_out, d_in = self.node_encoder.gnn_encoder.convs[i].lin.weight.shape
        ok = (32, 64, 128) if self.encoder_precision != "bf16" else (64, 128)   # (bf16 table: a row is >= one line)
        return self.encoder_fused and d_in == d_out and d_out in ok and in_dim == d_in

    def _layer_fused(self, i: int, a_hat: graph.DeviceCSR, x: torch.Tensor, lo: int, hi: int) -> torch.Tensor:
        """``lpf_gcn_layer_fused_f32``: out[r] = epilogue((sum_e w_e x[col_e]) W^T) for r in [lo, hi) -- the same layer
        as ``_layer_transform`` + ``_layer_aggregate`` with the sum taken before the product.  With
        ``encoder_precision == "bf16"`` the rows are gathered from a bf16 image of ``x`` (``lpf_gcn_layer_fused_bf16``;
        sums, product and epilogue fp32) and a whole-graph layer also emits that image of its OUTPUT for the next one."""
        enc = self.node_encoder.gnn_encoder
        conv = enc.convs[i]
        d = x.shape[1]
        last = i == len(enc.convs) - 1
        b16 = self.encoder_precision == "bf16"
        lib, st = _lib.hip(), _stream(self.device)
        cache = a_hat.__dict__.setdefault("_fused_order", {})
        key = (lo, hi, b16)
        if key not in cache:
            cache[key] = graph.fused_row_order(a_hat.rowptr, lo, hi, pad_hubs=b16)
        order, hubs, parts = cache[key]
        ln = enc.lns[i] if enc.lns is not None else None
        res = x[lo:hi] if enc.residual else None
        out = torch.empty(hi - lo, d, dtype=torch.float32, device=self.device)
        xb = self._bf16p(x) if b16 else None
        t_parts = None
        if hubs is not None:   # hub rows: their slices are summed first, the layer kernel reads the sums
            t_parts = self._workspace("gcn_t_parts", parts.shape[0] * d, torch.float32, st)
            with KernelTimer.span("spmm_row_parts"):
                if b16:
                    check(lib.lpf_spmm_row_parts_bf16p(d, ptr(parts), parts.shape[0], ptr(a_hat.col), ptr(a_hat.val),
                                                       ptr(xb), xb.stride(0), ptr(t_parts), st), "lpf_spmm_row_parts_bf16p")
                else:
                    check(lib.lpf_spmm_row_parts_f32(d, ptr(parts), parts.shape[0], ptr(a_hat.col), ptr(a_hat.val),
                                                     ptr(x), x.stride(0), ptr(t_parts), st), "lpf_spmm_row_parts_f32")
        common = (ptr(self._conv_packs[i].get(conv.lin.weight)), ptr(out), out.stride(0), ptr(conv.bias),
                  ptr(ln.weight) if ln is not None else None, ptr(ln.bias) if ln is not None else None,
                  ptr(res), 0 if res is None else res.stride(0),
                  ptr(self.gnn_norm.weight) if last else None, ptr(self.gnn_norm.bias) if last else None,
                  FLAG_RELU if enc.relu else 0, ptr(hubs), ptr(t_parts))
        with KernelTimer.span("gcn_layer_fused"):
            if b16:
                whole = lo == 0 and hi == self.num_nodes and not last
                out_b = torch.empty(hi - lo, d, dtype=torch.bfloat16, device=self.device) if whole else None
                check(lib.lpf_gcn_layer_fused_bf16(
                    d, order.numel() // 16, ptr(order), lo, ptr(a_hat.rowptr), ptr(a_hat.col), ptr(a_hat.val), ptr(xb),
                    xb.stride(0), *common, ptr(out_b), d, st), "lpf_gcn_layer_fused_bf16")
                if whole:
                    self._xb_cache = (weakref.ref(out), out_b)
            else:
                check(lib.lpf_gcn_layer_fused_f32(
                    d, order.numel() // 16, ptr(order), lo, ptr(a_hat.rowptr), ptr(a_hat.col), ptr(a_hat.val), ptr(x),
                    x.stride(0), *common, None, 0, st), "lpf_gcn_layer_fused_f32")
        return out

    def _bf16p(self, x: torch.Tensor) -> torch.Tensor:
        """The bf16 image of a [N, D] fp32 table in the order ``lpf_gcn_layer_fused_bf16`` gathers from (element
        32 i + 8 q + 4 h + u = feature 16 (2 i + h) + 4 q + u): the one the previous fused layer wrote beside ``x``, or a
        torch permute + cast (the feature table: once per model; an all-gathered layer input: once per layer)."""
        hit = getattr(self, "_xb_cache", None)
        if hit is not None and hit[0]() is x:
            return hit[1]
        feat = getattr(self, "_xb_feat", None)
        if feat is not None and feat[0]() is x and feat[1] == x._version:
            return feat[2]
        n, d = x.shape
        xb = x.reshape(n, d // 32, 2, 4, 4).permute(0, 1, 3, 2, 4).reshape(n, d).to(torch.bfloat16).contiguous()
        if x is self._features():
            self._xb_feat = (weakref.ref(x), x._version, xb)
        return xb

    def _layer_transform(self, i: int, x_rows: torch.Tensor) -> torch.Tensor:
        """GCNConv.lin of layer i on the given rows (other_models.py:66 -> PyG GCNConv: x W^T, no bias).  With
        ``encoder_precision == "bf16"`` the product (fp32 MFMA) is stored as bf16: it is the table the aggregation
        gathers -- and, row-sharded, the tensor that crosses xGMI."""
        conv = self.node_encoder.gnn_encoder.convs[i]
        w = self._conv_pads[i].get(conv.lin.weight)
        a = _as_f32_rows(x_rows)
        if self.encoder_precision != "bf16":
            return gemm(a, w, tag="gemm_encoder")
        m, k = a.shape
        n = w.shape[0]
        ld = (n + 7) // 8 * 8
        out = torch.empty(m, ld, dtype=torch.bfloat16, device=a.device)[:, :n]
        with KernelTimer.span("gemm_encoder"):
            check(_lib.hip().lpf_gemm_f32_out_bf16(m, n, k, ptr(a), a.stride(0), ptr(w), w.stride(0), None, None, 0,
                                                   ptr(out), out.stride(0), 0, _stream(a.device)),
                  "lpf_gemm_f32_out_bf16")
        return out

    def _layer_aggregate(self, i: int, a_hat: graph.DeviceCSR, t: torch.Tensor, lo: int, hi: int,
                         x_rows: torch.Tensor) -> torch.Tensor:
        """Rows [lo, hi) of layer i's output from the transformed features ``t`` of ALL nodes: A_hat t + bias ->
        LayerNorm -> ReLU -> (+ residual: ``x_rows`` = the same rows of the layer input) and, for the last layer, the
        model's ``gnn_norm`` (other_models.py:66-74, link_transformer.py:127), all in the SpMM epilogue."""
        enc = self.node_encoder.gnn_encoder
        conv = enc.convs[i]
        d = t.shape[1]
        last = i == len(enc.convs) - 1
        res = x_rows if (enc.residual and x_rows.shape[1] == d) else None
        ln = enc.lns[i] if enc.lns is not None else None
        long_rows = self._long_rows(a_hat, lo, hi)
        out = torch.empty(hi - lo, d, dtype=torch.float32, device=self.device)
        bf16 = t.dtype == torch.bfloat16
        fn = _lib.hip().lpf_spmm_csr_bf16 if bf16 else _lib.hip().lpf_spmm_csr_f32
        with KernelTimer.span("spmm_csr"):
            check(fn(
                hi - lo, d, a_hat.rowptr.data_ptr() + 8 * lo, ptr(a_hat.col), ptr(a_hat.val), ptr(t),
                t.stride(0), ptr(out), out.stride(0), ptr(conv.bias),
                ptr(ln.weight) if ln is not None else None, ptr(ln.bias) if ln is not None else None,
                ptr(res), 0 if res is None else res.stride(0),
                ptr(self.gnn_norm.weight) if last else None, ptr(self.gnn_norm.bias) if last else None,
                FLAG_RELU if enc.relu else 0, ptr(long_rows), 0 if long_rows is None else long_rows.numel(),
                _stream(self.device)), "lpf_spmm_csr_bf16" if bf16 else "lpf_spmm_csr_f32")
        return out

    def _long_rows(self, a_hat: graph.DeviceCSR, lo: int, hi: int):
        """Hub rows (> LPF_SPMM_LONG_ROW entries) of the local row block, as row ids relative to `lo`; cached on the
        device graph itself, so the list lives exactly as long as the graph it describes."""
        cache = a_hat.__dict__.setdefault("_long_rows", {})
        if (lo, hi) not in cache:
            deg = (a_hat.rowptr[lo + 1:hi + 1] - a_hat.rowptr[lo:hi])
            rows = torch.nonzero(deg > 128).flatten().to(torch.int32)
            cache[(lo, hi)] = rows if rows.numel() else None
        return cache[(lo, hi)]

    def set_row_shard(self, rank: int, world: int, mode: str = "sharded"):
        """This process is rank ``rank`` of ``world`` (default process group, lpformer_amd/dist.py).  ``mode``:
        "sharded" = row-sharded encoder with an all-gather per layer (L + 1 collectives), "replicated" = every rank runs
        the whole encoder (no exchange), "gather_once" = layers 1..L-1 replicated, the last layer's aggregation and the
        per-node attention projections row-sharded, ONE all-gather of [X_node | Z]
        (``lpformer_amd.dist.encoder_plan`` prices the three)."""
        if not (0 <= rank < world):
            raise ValueError("need 0 <= rank < world")
        if mode not in ("sharded", "replicated", "gather_once"):
            raise ValueError("mode must be 'sharded', 'replicated' or 'gather_once'")
        self._shard = (rank, world)
        self.encoder_mode = mode

    def _node_keys(self, x_node: torch.Tensor, w):
        """Per encoder output (cached on the tensor's identity and version), the node-level projection that the
        reference recomputes per selected node:  Z = X_node W_rx^T + b_r, the node half of lin_r (k_e = Z[v] + ...).
        With ``query_from = "table"`` the query table Y = X_node W_l^T + b_l (``_node_y``) comes out of the same product:
        Z and Y are then the two halves of the rows of one [N, 2D] table (row stride 2D for both)."""
        # Keyed on the IDENTITY of the encoder output (weak reference) + its version: the encoder writes its output
        # through raw pointers (no version bump) and the caching allocator hands the same address to the next
        # propagate(), so neither data_ptr nor _version alone can tell two encoder outputs apart; a tensor object can.
        hit = self._z_cache
        if hit is None or hit[0]() is not x_node or hit[1] != x_node._version:
            xr = _as_f32_rows(x_node)
            if self.query_from == "table" and 2 * self.dim <= 256:
                # the query table Y will be asked for next: ONE [N, D] x [D, 2D] product leaves both, Z and Y being the
                # two halves of its rows (one launch and one pass over X; the time is the matrix pipe's either way:
                # 0.20 ms on collab-like against 0.09 + 0.10)
                zy = gemm(xr, w["w_zy"], w["b_zy"], tag="gemm_node_keys")
                z, y = zy[:, :self.dim], zy[:, self.dim:]
            else:
                z, y = gemm(xr, w["w_rx"], w["b_r"], tag="gemm_node_keys"), None    # [N, D]
            torch.cuda.current_stream(self.device).synchronize()  # once per encoder output: other streams read Z
            hit = self._z_cache = (weakref.ref(x_node), x_node._version, z)
            if y is not None:
                self._y_cache = (weakref.ref(x_node), x_node._version, y)
        return hit[2]

    def _node_y(self, x_node: torch.Tensor, w) -> torch.Tensor:
        """Y = X_node W_l^T + b_l, lin_l per node (cached like Z): with it q_pair = Y[a] + Y[b] -- literally the
        reference's expression (layers.py:212-215) -- is a gather-add per batch instead of a product."""
        hit = getattr(self, "_y_cache", None)
        if hit is None or hit[0]() is not x_node or hit[1] != x_node._version:
            y = gemm(_as_f32_rows(x_node), w["w_l"], w["b_l"], tag="gemm_node_query")
            torch.cuda.current_stream(self.device).synchronize()  # once per encoder output: other streams read Y
            hit = self._y_cache = (weakref.ref(x_node), x_node._version, y)
        return hit[2]

    def _pair_q(self, batch, x_node, w) -> torch.Tensor:
        """q_pair = lin_l(x_a) + lin_l(x_b)  (layers.py:212-215) on the current stream.  ``query_from``:
        "table" -- Y[a] + Y[b] from the per-node table ``_node_y`` (one N x D x D product per ENCODER OUTPUT, then 12 us
        of gather per batch: the evaluation pattern, one encoder pass for thousands of batches);
        "gemm"  -- W_l (x_a + x_b) + 2 b_l per batch, the endpoint rows gathered and added inside the first stage of
        ``lpf_dense_chain_f32`` (in_mode 2; 19 us per collab-like batch, nothing per encoder output: the
        ``test_edge`` pattern, one encoder pass per batch)."""
        if self.query_from == "table":
            y = self._node_y(x_node, w)
            bs, d = batch.shape[1], self.dim
            q = torch.empty(bs, d, dtype=torch.float32, device=self.device)
            with KernelTimer.span("pair_gather_q"):
                check(_lib.hip().lpf_pair_gather_f32(bs, d, ptr(batch), batch.stride(0), y.shape[0], ptr(y), y.stride(0),
                                                     None, 0, ptr(q), d, _stream(self.device)), "lpf_pair_gather_f32")
            return q
        lin_l = self.att_layers[0].att.lin_l
        t = self._chain_q.tables(lin_l.weight, w["b_l2"])
        q = self._chain_q.run(t, x_node, relu=False, batch=batch, in_mode=2)
        if q is None:   # (a shape without a fused instantiation: gather-add, then the plain product)
            bs, d = batch.shape[1], x_node.shape[1]
            xs = torch.empty(bs, d, dtype=torch.float32, device=self.device)
            check(_lib.hip().lpf_pair_gather_f32(bs, d, ptr(batch), batch.stride(0), x_node.shape[0], ptr(x_node),
                                                 x_node.stride(0), None, 0, ptr(xs), d, _stream(self.device)),
                  "lpf_pair_gather_f32")
            q = gemm(xs, lin_l.weight, w["b_l2"], tag="pair_q")
        return q

    def _z_bf16(self, z: torch.Tensor) -> torch.Tensor:
        """bf16 copy of the node table Z (once per encoder output; the storage format of the bf16 throughput mode)."""
        hit = getattr(self, "_zb_cache", None)
        if hit is None or hit[0] is not z:
            zb = z.to(torch.bfloat16).contiguous()
            torch.cuda.current_stream(self.device).synchronize()  # other streams read it
            hit = self._zb_cache = (z, zb)
        return hit[1]

    # ---------------------------------------------------------------------------------- selection
    def _sel_ws(self, st, bs: int, regions: Optional[bool] = None) -> "_SelectWorkspace":
        """The type-major selection workspace of (stream, batch size); ``regions``: it must be (True) / must not be
        (False) the one behind ``lpf_select4`` -- a workspace of the other kind is replaced; None: whatever is there."""
        key = ("sel2", st, bs)
        ws = self._ws.get(key)
        if ws is None or (regions is not None and isinstance(ws, _Select4RegionsWorkspace) != regions):
            ws = self._ws[key] = (_Select4RegionsWorkspace if regions else _SelectWorkspace)(self.device, bs)
        return ws

    def _regions_launch(self, ws, batch, wi):
        """``lpf_select4`` + ``lpf_select4_regions`` on the current stream; no host sync."""
        self._select4_launch(ws, batch, wi, regions=True)
        with KernelTimer.span("select_regions"):
            check(_lib.hip().lpf_select4_regions(batch.shape[1], ptr(ws.pair_tab), ptr(ws.blk_types), ptr(ws.entries4),
                                                 ws.ent_cap4, ptr(ws.type_ptr), ptr(ws.entries), ws.ent_cap, ptr(ws.ctl),
                                                 _stream(self.device)), "lpf_select4_regions")

    def _uses_select4_regions(self, adj_mask=None) -> bool:
        """True when the type-major consumers sit behind lpf_select4 + lpf_select4_regions."""
        return self.select4_regions and self.select_blocks and adj_mask is None and self.use_select_index

    def _select_regions_device(self, batch: torch.Tensor, test_set: bool) -> "_Select4RegionsWorkspace":
        """As ``_select_device`` for the walk-index path, through the one-launch selection: sized from earlier batches
        (a block reserves its candidate slots in the pair-major buffer, the regions hold the kept entries), a batch that
        does not fit raises the sticky bits (NaN scores, ``check_selection()`` sizes again), the first batch of a
        (stream, batch size) is sized exactly with synchronisations."""
        st = _stream(self.device)
        bs = batch.shape[1]
        ws = self._sel_ws(st, bs, regions=True)
        wi = self._select_graphs(test_set, None)
        if not ws.calibrated:
            ws.ensure4(16)
            ws.ensure(ent_cap=16)
            self._select4_launch(ws, batch, wi, regions=True)    # (token buffer: the kernel still reports the room it needs)
            err, _ = ws.read_status()
            need = int(ws.ctl[0].item())
            ws.shape = self._select4_shape(bs, int(ws.ctl[1].item()))
            ws.clear_errors()
            if err & _lib.SELECT_ERR_NODE_RANGE:
                raise IndexError(f"batch holds node ids outside [0, {self.num_nodes})")
            ws.ensure4(3 * need + 65536, shrink=True)
            self._regions_launch(ws, batch, wi)                  # (token regions: the totals per type come back)
            _, tot = ws.read_status()
            ws.clear_errors()
            ws.ensure(ent_cap=2 * max(tot) + 4096, shrink=True)
            ws.calibrated = True
        self._regions_launch(ws, batch, wi)
        return ws

    def _select_launch(self, ws, batch, graphs):
        """The two selection launches on the current stream; no host sync.  ``graphs`` is a ``graph.WalkIndex`` (the
        evaluation path, lpf_select3_*) or the tuple of the general path (adjacency override: lpf_select_plan / _run
        over the raw PPR rows)."""
        lib, st = _lib.hip(), _stream(self.device)
        bs = batch.shape[1]
        if isinstance(graphs, graph.WalkIndex):
            wi, cn = graphs, 1 if self.mask == "cn" else 0
            with KernelTimer.span("select_plan"):
                check(lib.lpf_select3_plan(bs, ptr(batch), batch.stride(0), self.num_nodes, ptr(wi.rec), ptr(wi.adj_cv),
                                           ptr(wi.a1_cv), ptr(wi.px_cv), ptr(wi.t0_cv), cn, 1 if wi.use_px else 0,
                                           ptr(ws.desc), ptr(ws.offs), ptr(ws.item_pair), ws.item_cap, ptr(ws.ctl),
                                           ptr(ws.plan_lb), st), "lpf_select3_plan")
            with KernelTimer.span("select_run"):
                check(lib.lpf_select3_run(bs, ptr(ws.desc), ptr(ws.offs), ptr(ws.item_pair), ws.item_cap, ptr(ws.ctl),
                                          ptr(ws.run_lb), ptr(wi.u.cv), ptr(wi.mini), float(self.thresh_cn),
                                          float(self.thresh_1hop),
                                          float(self.thresh_non1hop), cn, ptr(ws.type_ptr), ptr(ws.entries),
                                          ws.ent_cap, self.select_grid, st), "lpf_select3_run")
            return
        adj, adjx, val, t0 = graphs
        with KernelTimer.span("select_plan"):
            check(lib.lpf_select_plan(bs, ptr(batch), batch.stride(0), self.num_nodes, ptr(adj.rowptr),
                                      ptr(val.rowptr), ptr(t0.rowptr) if t0 is not None else None,
                                      ptr(adjx.rowptr) if adjx is not adj else None,
                                      ptr(getattr(val, "len", None)), ptr(getattr(t0, "len", None)),
                                      ptr(ws.desc), ptr(ws.offs),
                                      ptr(ws.item_pair), ws.item_cap, ptr(ws.ctl), ptr(ws.plan_lb), st),
                  "lpf_select_plan")
        with KernelTimer.span("select_run"):
            check(lib.lpf_select_run(bs, ptr(ws.desc), ptr(ws.offs), ptr(ws.item_pair), ws.item_cap, ptr(ws.ctl),
                                     ptr(ws.run_lb), ptr(adj.col), None,
                                     ptr(adjx.col) if adjx is not adj else None, ptr(val.col), ptr(val.val), None,
                                     ptr(t0.cv) if t0 is not None else None,
                                     ptr(t0.skip) if t0 is not None else None, float(self.thresh_cn),
                                     float(self.thresh_1hop), float(self.thresh_non1hop), 1 if self.mask == "cn" else 0,
                                     ptr(ws.type_ptr), ptr(ws.entries), ws.ent_cap, 0, st), "lpf_select_run")

    def _walk_index(self, mask_obj, ppr_obj) -> graph.WalkIndex:
        """The per-model indexes of the walk-plan selection for this (adjacency, PPR matrix) pair of ``self.data``,
        built on the device on first use and kept with the graphs."""
        key = ("walk", id(mask_obj), id(ppr_obj))
        hit = self._graphs.get(key)
        if hit is None or hit[0] is not mask_obj or hit[1] is not ppr_obj:
            wi = graph.build_walk_index(self._device_graph("mask", mask_obj), self._device_graph("ppr", ppr_obj),
                                        self.thresh_1hop, self.thresh_non1hop, want_t0=self.mask == "all")
            hit = self._graphs[key] = (mask_obj, ppr_obj, wi)
        return hit[2]

    def _select_graphs(self, test_set: bool, adj_mask=None):
        """What the selection launches read.  Evaluation (the typing adjacency is the model's own): the walk indexes
        (``graph.WalkIndex``).  A caller-supplied adjacency override (training) or ``use_select_index = False``: the
        tuple (typing adjacency, unmasked adjacency, raw PPR rows, T0 rows or None) of the general path."""
        ppr_obj, mask_obj = self._data_obj("ppr", test_set), self._data_obj("mask", test_set)
        if adj_mask is None and self.use_select_index:
            return self._walk_index(mask_obj, ppr_obj)
        ppr = self._device_graph("ppr", ppr_obj)
        adjx = self._device_graph("mask", mask_obj)
        if isinstance(adj_mask, graph.RemovedEdges):
            # a consumer on the general path (the fused evaluation kernels behind an override): the own adjacency without
            # the removed edges as a CSR of its own (one slot, like every override)
            hit = self._override.get("mask")
            if hit is None or hit[0] is not adj_mask:
                own = self._own_mask_keys(test_set)
                rk = mask_delta.removed_from_edges(own, torch.as_tensor(adj_mask.edges).to(self.device), self.num_nodes)
                keep = own[~mask_delta._member(rk, own)]
                r = torch.div(keep, self.num_nodes, rounding_mode="floor")
                hit = self._override["mask"] = (adj_mask, graph.csr_from_coo_device(r, keep - r * self.num_nodes, None,
                                                                                    self.num_nodes))
            adj = hit[1]
        else:
            adj = adjx if adj_mask is None else self._device_graph("mask", adj_mask)
        t0 = self._device_graph("t0", ppr_obj) if self.mask == "all" else None
        return adj, adjx, ppr, t0

    def _select_device(self, batch: torch.Tensor, test_set: bool, adj_mask=None) -> "_SelectWorkspace":
        """Selection for the hot path: the result stays in the stream's workspace (per-type entry regions + int32
        segment pointers), sized from earlier batches; nothing is read back.  A batch that does not fit raises the
        sticky error bits in ``ws.ctl`` -- consumers clamp, the scores of such a batch come out as NaN, and
        ``check_selection()`` (called by the evaluation sweep and by every API that synchronises anyway) grows the
        workspace.  The first batch of a (stream, batch size) is sized exactly, with one synchronisation."""
        if self._uses_select4_regions(adj_mask):
            return self._select_regions_device(batch, test_set)
        st = _stream(self.device)
        bs = batch.shape[1]
        ws = self._sel_ws(st, bs, regions=False)
        graphs = self._select_graphs(test_set, adj_mask)
        if not ws.calibrated:
            # exact sizing, once (synchronises): the kernels count every selected entry even when the entry regions
            # are too small to hold them, so one pass with token regions gives the totals
            ws.ensure(item_cap=max(bs, 4096) + 16, ent_cap=16)
            for _attempt in range(3):
                self._select_launch(ws, batch, graphs)
                items = int(ws.ctl[1].item())
                err, tot = ws.read_status()
                ws.clear_errors()
                if err & _lib.SELECT_ERR_NODE_RANGE:
                    raise IndexError(f"batch holds node ids outside [0, {self.num_nodes})")
                if not (err & _lib.SELECT_ERR_ITEM_CAP):
                    break
                ws.ensure(item_cap=2 * items + 16, ent_cap=16)
            else:
                raise _lib.LpfError("selection workspace could not be sized")
            ws.ensure(item_cap=2 * items + 16, ent_cap=2 * max(tot) + 4096, shrink=True)
            ws.calibrated = True
        self._select_launch(ws, batch, graphs)
        return ws

    def _uses_select4(self, adj_mask=None) -> bool:
        """True when the hot path's selection is the one-launch, pair-major kernel (csrc/select4.hip): the model's own
        typing adjacency through the walk indexes, feeding the pair-major attention."""
        return self.select_blocks and adj_mask is None and self.use_select_index and self._patterns_pay()

    # Flipped units per entry LEFT for the exact path (entries in flagged cells, counted against the tabulated patterns
    # their cells name) above which the pair-major kernel behind select4 would be taken to lose to the type-major form, whose
    # exact path is the rule rather than the exception.  Round 6 measured the two forms over a sweep of PE weights
    # (tools/pt_breakeven.py, profiles/r06_pt_breakeven_d128.txt / _d256.txt; both now behind lpf_select4): at D = 128 the
    # table form is ahead at 0.0 / 0.27 / 0.62 / 5.4 / 11.6 / 12.1 units left (110 / 151 / 193 / 496 / 728 / 777 us for
    # selection + attention against 141 / 246 / 348 / 653 / 1,002 / 826) and 9 % behind at ONE point (8.1 left: 568 against
    # 517); at D = 256 it is ahead at every point (3.0 ... 43 left).  Far up there both lose to the matrix-core kernel
    # anyway (`attention_kernel`).  So the gate is open by default; the mechanism stays for callers and tests.
    PT_EXACT_MAX = float("inf")

    def _patterns_pay(self) -> bool:
        """False when the activation-pattern table covers too little of this model's entries (``PT_EXACT_MAX``): the hot
        path then keeps select3 + the type-major kernels.  Looked at once per parameter version in evaluation mode (a
        selection on a sample + the table build, 0.2-0.3 s), with the same spacing as ``attention_kernel``'s estimate."""
        if self.dim < 128 or not (self.attention_rows and self.use_fused_attention):
            return True
        self._fold()
        last = getattr(self, "_pt_choice", None)        # (parameter key, choice, refolds at the time)
        if last is not None and (last[0] is self._folded[0] or self.training or
                                 self._refolds - last[2] < self.flip_recheck_every):
            return last[1]
        if self.training:
            return True
        ok = self._flip_stats()[1] <= self.PT_EXACT_MAX
        self._pt_choice = (self._folded[0], ok, self._refolds)
        return ok

    # Launch shape of lpf_select4 from what the first batch of a (stream, batch size) showed (round 6, same-lease interleaved
    # A/B under the pipelined bench, profiles/r06_select4_shapes_pipelined.txt): where a batch gives every CU two
    # workgroups even at one block of 64 pairs each and a block holds a few thousand candidate slots -- collab-like 3,000,
    # ppa-like 4,500 -- 512-thread workgroups of ONE block (two to a CU, their phases out of step; they also fit beside
    # another kernel's workgroup) take the step down 1-3.5 % (collab-like 0.1434-0.1456 -> 0.1381-0.1423, 0.1365 -> 0.1350
    # on a faster lease; ppa-like 0.0833 -> 0.0805-0.0829); blocks of ~1,400 slots (citation2-like: +4 %) or a batch of a
    # few hundred blocks (ddi-like, cora-like: equal) keep the library's default (1,024 threads, two blocks).
    S4_SHAPE_SLOTS = (2000, 6000)

    def _select4_shape(self, bs: int, slots: int) -> int:
        nblk = (bs + _lib.SELECT4_BLOCK - 1) // _lib.SELECT4_BLOCK
        n_cu = torch.cuda.get_device_properties(self.device).multi_processor_count
        lo, hi = self.S4_SHAPE_SLOTS
        return 512 if (nblk >= 2 * n_cu and lo <= slots / max(nblk, 1) <= hi) else 0

    def _select4_launch(self, ws, batch, wi, regions: bool = False):
        """``lpf_select4`` into ``ws`` (a ``_Select4Workspace``, or -- ``regions`` -- the pair-major half of a
        ``_Select4RegionsWorkspace``)."""
        lib, st = _lib.hip(), _stream(self.device)
        cn = 1 if self.mask == "cn" else 0
        with KernelTimer.span("select_run"):
            check(lib.lpf_select4(batch.shape[1], ptr(batch), batch.stride(0), self.num_nodes, ptr(wi.rec), ptr(wi.adj_cv),
                                  ptr(wi.a1_cv), ptr(wi.px_cv), ptr(wi.t0_cv), ptr(wi.u.cv), ptr(wi.mini), cn,
                                  1 if wi.use_px else 0, float(self.thresh_cn), float(self.thresh_1hop),
                                  float(self.thresh_non1hop), ptr(ws.ctl), ptr(ws.pair_tab), ptr(ws.blk_cnt),
                                  ptr(getattr(ws, "blk_types", None)), ptr(ws.entries4 if regions else ws.entries),
                                  ws.ent_cap4 if regions else ws.ent_cap,
                                  self.select4_threads or getattr(ws, "shape", 0), st), "lpf_select4")

    def _select4_device(self, batch: torch.Tensor, test_set: bool) -> "_Select4Workspace":
        """The one-launch selection for the hot path (``lpf_select4``): pair-major entries, a table entry per pair,
        nothing read back.  The entry buffer is sized from earlier batches (a batch needs room for its candidate
        slots, which the kernel reports in ``ctl[0]``); a batch that does not fit raises the sticky error bits, its
        scores come out as NaN and ``check_selection()`` sizes the buffer again.  The first batch of a (stream, batch
        size) is sized exactly, with one synchronisation."""
        st = _stream(self.device)
        bs = batch.shape[1]
        key = ("sel4", st, bs)
        ws = self._ws.get(key)
        if ws is None:
            ws = self._ws[key] = _Select4Workspace(self.device, bs)
        wi = self._select_graphs(test_set, None)
        if not ws.calibrated:
            ws.ensure(ent_cap=16)
            self._select4_launch(ws, batch, wi)      # (token buffer: the kernel still reports the room it needs)
            err, need = ws.read_status()
            ws.shape = self._select4_shape(bs, int(ws.ctl[1].item()))
            ws.clear_errors()
            if err & _lib.SELECT_ERR_NODE_RANGE:
                raise IndexError(f"batch holds node ids outside [0, {self.num_nodes})")
            # a block's place is reserved for its candidate SLOTS (an upper bound of what it keeps): three times the
            # first batch's, hub-heavy batches need several times the room of sparse ones
            ws.ensure(ent_cap=3 * need[0] + 65536, shrink=True)
            self._select4_launch(ws, batch, wi)
            ws.kept_cap = 2 * ws.kept() + 65536
            ws.calibrated = True
        self._select4_launch(ws, batch, wi)
        return ws

    def check_selection(self, stream=None) -> bool:
        """Synchronising check of the sticky selection status of ``stream`` (default: the current one).  Returns True
        when every batch since the last check fitted its workspace; otherwise clears the status, marks the workspace
        for re-sizing and returns False (the caller re-scores those batches).  Node ids out of range raise
        IndexError.  The status is read (and cleared) ON ``stream`` itself, behind everything queued there, whatever
        the caller's current stream is: a lane that has not run yet cannot be read as "ok"."""
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        st = stream.cuda_stream
        ok = True
        for key, ws in list(self._ws.items()):
            if not (isinstance(key, tuple) and key[0] in ("sel2", "sel4") and key[1] == st):
                continue
            err, _ = ws.read_status(stream)
            if err == 0:
                continue
            ok = False
            ws.clear_errors(stream)
            if err & _lib.SELECT_ERR_NODE_RANGE:
                raise IndexError(f"batch holds node ids outside [0, {self.num_nodes})")
            ws.calibrated = False   # re-size on the next batch
        return ok

    def _prop_delta(self, obj):
        """A propagation-matrix override (the ``--mask-input`` loop, src/train/train_model.py:47-56) that is the resident
        graph minus some edges, as a graph SHARING the resident structure: the raw weights with the removed edges at 0,
        re-normalised by ``lpf_gcn_norm_csr`` (degrees from what is left).  Nothing structural is rebuilt -- no sort of the
        edge list, no degree-ordered tiles for the one-launch layer, no transposed structure for the backward pass
        (4 ms per batch on the collab-like graph).  None: the override is a graph of its own (lpformer_amd/mask_delta.py)."""
        own_obj = self.data.get("adj_t")
        named = isinstance(obj, graph.RemovedEdges)
        if own_obj is None or obj is own_obj or not (self.use_mask_delta or named):
            return None
        own = self._device_graph("prop", own_obj)
        raw = own.__dict__.get("_struct_val")
        if raw is None:
            return None
        keys = own.__dict__.get("_edge_keys")
        if keys is None:
            keys = own.__dict__["_edge_keys"] = mask_delta.edge_keys(own.rowptr, own.col, own.n)
        if isinstance(obj, graph.RemovedEdges):
            # the difference named (no tensor of all the kept edges was built): the removed edges' raw weights to 0
            e = obj.edges if isinstance(obj.edges, torch.Tensor) else torch.as_tensor(np.asarray(obj.edges))
            new_w = mask_delta.prop_weights_minus_edges(keys, raw, e.to(self.device), own.n)
            return self._share_structure(own, new_w)
        coo = graph.as_coo_device(obj, self.device)
        if coo is None:
            try:
                row, col, val, nn = graph.as_coo_numpy(obj)
            except TypeError:
                return None
            coo = (torch.from_numpy(np.ascontiguousarray(row)).to(self.device),
                   torch.from_numpy(np.ascontiguousarray(col)).to(self.device),
                   None if val is None else torch.from_numpy(np.ascontiguousarray(val)).to(self.device), nn)
        if coo[3] != own.n:
            return None
        new_w = mask_delta.prop_weights_from_coo(keys, raw, coo[0], coo[1], coo[2], own.n)
        if new_w is None:
            return None
        return self._share_structure(own, new_w)

    @staticmethod
    def _share_structure(own: graph.DeviceCSR, new_w: torch.Tensor) -> graph.DeviceCSR:
        """The propagation matrix of raw weights ``new_w`` laid out on ``own``'s structure: normalised, sharing rowptr /
        col, the one-launch layer's tiles and (train.py) the transposed structure."""
        g = graph.gcn_norm_device(graph.DeviceCSR(own.rowptr, own.col, new_w, own.n, None))
        for k in ("_fused_order", "_long_rows"):      # (functions of the structure alone: one dict for both graphs)
            g.__dict__[k] = own.__dict__.setdefault(k, {})
        for k in ("_long_rows_full", "_edge_keys"):
            if k in own.__dict__:
                g.__dict__[k] = own.__dict__[k]
        g.__dict__["_structure_of"] = own
        return g

    def _own_mask_keys(self, test_set: bool) -> torch.Tensor:
        """Sorted int64 keys row * N + col of the model's own typing adjacency (kept with its device graph)."""
        g = self._device_graph("mask", self._data_obj("mask", test_set))
        k = g.__dict__.get("_edge_keys")
        if k is None:
            k = g.__dict__["_edge_keys"] = mask_delta.edge_keys(g.rowptr, g.col, g.n)
        return k

    def _mask_delta(self, adj_mask, test_set: bool):
        """The override as a difference to the model's own typing adjacency: sorted directed keys of the removed edges, or
        None (not a subset of the own adjacency / too many edges removed / a model whose common-neighbour threshold is
        positive, lpformer_amd/mask_delta.py).  The last override's answer is kept (the training loop passes a fresh
        object per batch; ``compute_node_mask`` and ``forward`` of one batch share theirs)."""
        if not (self.use_mask_delta and self.use_select_index) or self.thresh_cn > 0:
            return None
        hit = self._delta_cache
        if hit is not None and hit[0] is adj_mask and hit[1] == test_set:
            return hit[2]
        n, dev = self.num_nodes, self.device
        own = self._own_mask_keys(test_set)
        if isinstance(adj_mask, graph.RemovedEdges):
            rk = mask_delta.removed_from_edges(own, torch.as_tensor(adj_mask.edges).to(dev), n)
        else:
            coo = graph.as_coo_device(adj_mask, dev)
            if coo is None:     # (a host-side container: its indices go to the device once, nothing is sorted there)
                row, col, _, nn = graph.as_coo_numpy(adj_mask)
                coo = (torch.from_numpy(np.ascontiguousarray(row)).to(dev), torch.from_numpy(np.ascontiguousarray(col)).to(dev),
                       None, nn)
            if coo[3] != n:
                raise ValueError(f"graph has {coo[3]} nodes, the model {n}")
            rk = mask_delta.removed_from_coo(own, coo[0], coo[1], n, self.mask_delta_limit)
        self._delta_cache = (adj_mask, test_set, rk)
        return rk

    def _ppr_lookup(self, test_set: bool):
        """rows, cols -> raw PPR values (0 where nothing is stored) through ``lpf_csr_lookup_f32``."""
        ppr = self._device_graph("ppr", self._data_obj("ppr", test_set))

        def lookup(rows, cols):
            out = torch.empty(rows.numel(), dtype=torch.float32, device=self.device)
            rows, cols = rows.contiguous(), cols.contiguous()
            check(_lib.hip().lpf_csr_lookup_f32(rows.numel(), ppr.n, ptr(rows), ptr(cols), ptr(ppr.rowptr), ptr(ppr.col),
                                                ptr(ppr.val), ptr(out), _stream(self.device)), "lpf_csr_lookup_f32")
            return out
        return lookup

    def _select_patched(self, batch: torch.Tensor, test_set: bool, rk: torch.Tensor):
        """``_select`` for an override given as removed edges: the resident-index selection, then the patch."""
        s = self._select(batch, test_set, None)
        bs = s["bs"]
        if bs == 0 or rk.numel() == 0:
            return s
        pair, node, pa, pb, tp, counts = mask_delta.patch_selection(s, batch, rk, self.num_nodes, self.mask,
                                                                    self.thresh_1hop, self._ppr_lookup(test_set))
        feats, d = s["feats"], self.dim
        c = counts.to(torch.float32)
        if self.count_dim == 4:
            feats[:, d:d + 4] = torch.stack([c[0], c[1], c[2], c[0] + c[1]], dim=1)
        elif self.count_dim == 3:
            feats[:, d:d + 3] = torch.stack([c[0], c[1], c[0] + c[1]], dim=1)
        else:
            feats[:, d] = c[0]
        return {"bs": bs, "cap": max(int(pair.numel()), 1), "type_ptr": tp, "sel_pair": pair, "sel_node": node,
                "sel_pa": pa, "sel_pb": pb, "feats": feats, "ldf": s["ldf"]}

    def _select(self, batch: torch.Tensor, test_set: bool, adj_mask=None):
        """Selection in the REFERENCE layout (module-by-module path, compute_node_mask, attention weights): the two
        selection launches, a status check (this path synchronises), then lpf_select_export.  Returns a dict of device
        arrays (type-major entries sorted by (pair, node), int64 segment pointers, float count features)."""
        if adj_mask is not None:
            rk = self._mask_delta(adj_mask, test_set)
            if rk is not None:
                return self._select_patched(batch, test_set, rk)
        lib, st = _lib.hip(), _stream(self.device)
        bs = batch.shape[1]
        ldf = _pad4(self.dim + self.count_dim)
        feats = torch.empty(bs, ldf, dtype=torch.float32, device=self.device)  # [att out | counts | pad]
        if ldf > self.dim + self.count_dim:
            feats[:, self.dim + self.count_dim:].zero_()  # the attention output and the counts are written below
        type_ptr = self._workspace("type_ptr", 3 * (bs + 1), torch.int64, st)
        if bs == 0:
            type_ptr[:3].zero_()
            empty_i = torch.empty(0, dtype=torch.int32, device=self.device)
            empty_f = torch.empty(0, dtype=torch.float32, device=self.device)
            return {"bs": 0, "cap": 0, "type_ptr": type_ptr, "sel_pair": empty_i, "sel_node": empty_i,
                    "sel_pa": empty_f, "sel_pb": empty_f, "feats": feats, "ldf": ldf}
        for _attempt in range(3):
            ws = self._select_device(batch, test_set, adj_mask)
            err, tot = ws.read_status()
            if err == 0:
                break
            ws.clear_errors()
            if err & _lib.SELECT_ERR_NODE_RANGE:
                raise IndexError(f"batch holds node ids outside [0, {self.num_nodes})")
            ws.calibrated = False  # did not fit: size again for this batch
        else:
            raise _lib.LpfError("selection workspace could not be sized")
        cap = sum(tot)
        sel_pair = self._workspace("sel_pair", cap, torch.int32, st)
        sel_node = self._workspace("sel_node", cap, torch.int32, st)
        sel_pa = self._workspace("sel_pa", cap, torch.float32, st)
        sel_pb = self._workspace("sel_pb", cap, torch.float32, st)
        with KernelTimer.span("select_export"):
            check(lib.lpf_select_export(bs, ptr(ws.type_ptr), ptr(ws.entries), ws.ent_cap, ptr(type_ptr),
                                        feats.data_ptr() + 4 * self.dim, ldf, self.count_dim, ptr(sel_pair),
                                        ptr(sel_node), ptr(sel_pa), ptr(sel_pb), st), "lpf_select_export")
        return {"bs": bs, "cap": max(cap, 1), "type_ptr": type_ptr, "sel_pair": sel_pair, "sel_node": sel_node,
                "sel_pa": sel_pa, "sel_pb": sel_pb, "feats": feats, "ldf": ldf, "tot": tuple(int(v) for v in tot)}

    def _prep_batch(self, batch) -> torch.Tensor:
        batch = torch.as_tensor(batch).to(self.device)
        if batch.dtype != torch.int64:
            batch = batch.long()
        if batch.dim() != 2 or batch.shape[0] != 2:
            raise ValueError("batch must be a 2 x BS tensor of node ids")
        if batch.stride(1) != 1:
            batch = batch.contiguous()
        return batch

    @_on_device
    def compute_node_mask(self, batch, test_set=False, adj=None):
        """Reference-format selection result (:214-276): three tuples (ix int64 [2,n], ppr_src, ppr_tgt) for CN,
        1-hop and >1-hop nodes (None for >1-hop in "1-hop" mode, for both in "cn" mode), each sorted by (pair
        position, node)."""
        self._check_supported(heads_ok=True)
        with torch.no_grad():
            batch = self._prep_batch(batch)
            s = self._select(batch, test_set, adj)
            bs = s["bs"]
            tp = s["type_ptr"][:3 * (bs + 1)].view(3, bs + 1)
            tot = tp[:, bs].tolist()
            out, base = [], 0
            for t in range(3):
                if (t == 2 and self.mask != "all") or (t == 1 and self.mask == "cn"):
                    out.append(None)   # (mode "1-hop": no >1-hop tuple; mode "cn": common neighbours only, :271-274)
                    continue
                sl = slice(base, base + tot[t])
                ix = torch.stack([s["sel_pair"][sl].long(), s["sel_node"][sl].long()])
                out.append((ix, s["sel_pa"][sl].clone(), s["sel_pb"][sl].clone()))
                base += tot[t]
            return tuple(out)

    # ---------------------------------------------------------------------------------- pair stage
    def lanes(self, k: int):
        """``k`` persistent HIP streams for pipelining independent batches (workspaces are kept per stream, so callers
        should reuse these instead of creating streams per sweep)."""
        if not hasattr(self, "_lanes"):
            self._lanes = []
        while len(self._lanes) < k:
            self._lanes.append(torch.cuda.Stream(self.device))
        return self._lanes[:k]

    def _fork(self):
        """Side stream ordered after everything already queued on the caller's stream (None when disabled)."""
        if not self.use_side_stream or KernelTimer.enabled:  # per-kernel timing wants serial launches
            return None
        if self._side is None:
            self._side = {}
        main = torch.cuda.current_stream(self.device)
        side = self._side.get(main.cuda_stream)
        if side is None:
            side = self._side[main.cuda_stream] = torch.cuda.Stream(self.device)
        _lib.stream_wait(side, main)
        return side

    # Measured break-even of the two fp32 one-pass attention kernels, in mean flipped hidden units per selected entry
    # (tools/flip_breakeven.py, profiles/r04_flip_breakeven.txt): the cost of csrc/pair_flip.hip grows with the number
    # of units of the PE hidden layer that leave the activation pattern of (0, 0), csrc/pair_fused.hip does the whole
    # D x D product whatever the weights are.
    FLIP_BREAK_EVEN = {128: 5.0, 256: 24.0}
    # ... and of the table form of the activation-pattern kernel behind select4, in units per entry LEFT for its exact path
    # (tools/pt_breakeven.py, profiles/r06_pt_breakeven_*.txt): at D = 128 its launch takes 68 / 110 / 151 / 454 us at 0 /
    # 0.27 / 0.62 / 5.4 units left against 165 us for the matrix-core kernel (+ 16 us for the regions its selection then
    # needs); at D = 256 96 / 132 / 263 us at 3.0 / 6.1 / 21.6 left against 93 (+ 8): the two lines cross at ~0.9 and ~3.4
    PT_BREAK_EVEN = {128: 0.8, 256: 3.4}

    def flip_break_even(self) -> float:
        """The threshold `attention_kernel` holds `flips_per_entry()` against: units left for the exact path of the table
        form where the model runs it, all flipped units otherwise."""
        table = self.PT_BREAK_EVEN if self._uses_select4() else self.FLIP_BREAK_EVEN
        return table.get(self.dim, 6.0)

    @_on_device
    def _entry_sample(self, n_pairs: int = 4096, seed: int = 0):
        """Per type the (pa, pb) values of the entries a sample batch selects -- half existing edges, half uniform
        pairs of the model's OWN graph and PPR matrix, the selection kernels run on it.  A property of the graph, the
        PPR matrix and the thresholds, not of the weights: drawn once per model.  It decides which activation patterns
        are tabulated (``patterns.build``) and feeds ``flips_per_entry``; results never depend on it."""
        mask_obj, ppr_obj = self._data_obj("mask", False), self._data_obj("ppr", False)
        hit = getattr(self, "_sample_cache", None)
        if hit is not None and hit[0] is mask_obj and hit[1] is ppr_obj:
            return hit[2]
        if hit is not None:     # another graph / PPR matrix was bound: everything derived from the old sample goes with it
            self._drop_sample_state()
        with torch.no_grad():
            mask = self._device_graph("mask", mask_obj)
            gen = torch.Generator(device=self.device)
            gen.manual_seed(seed)
            k = n_pairs // 2 if mask.nnz > 0 else 0
            parts = []
            if k:
                e = torch.randint(0, mask.nnz, (k,), device=self.device, generator=gen)
                rows = torch.searchsorted(mask.rowptr, e, right=True) - 1
                parts.append(torch.stack([rows, mask.col[e].long()]))
            parts.append(torch.randint(0, self.num_nodes, (2, n_pairs - k), device=self.device, generator=gen))
            batch = torch.cat(parts, dim=1).contiguous()
            sel = self._select(batch, False, None)
            bs = sel["bs"]
            tot = sel["type_ptr"][:3 * (bs + 1)].view(3, bs + 1)[:, bs].tolist()
            out, base = [None, None, None], 0
            for t in range({"all": 3, "1-hop": 2, "cn": 1}[self.mask]):
                out[t] = (sel["sel_pa"][base:base + tot[t]].clone(), sel["sel_pb"][base:base + tot[t]].clone())
                base += tot[t]
        self._sample_cache = (mask_obj, ppr_obj, out)
        return out

    def _drop_sample_state(self):
        """Forget the entry sample and what was decided from it (the select4 / select3 gate, the `auto` attention choice,
        the tabulated patterns): called when the graph or the PPR matrix the sample was drawn from is replaced."""
        self._sample_cache = None
        self._flip_est = self._pt_choice = self._auto_choice = self._pt_cache = None
        if self._folded is not None:
            self._folded[1].pop("patterns", None)

    def _pattern_tables(self, w: dict) -> dict:
        """Activation-pattern tables of the pair-major attention behind select4 (``lpformer_amd/patterns.py``), built on
        the device once per parameter version and kept with the folded weights (one lifetime: recorded plans and
        captured graphs hold pointers into both)."""
        pt = w.get("patterns")
        if pt is None:
            # keyed on the versions of the tensors the tables are built from (the PE MLPs and lin_r), not on the whole
            # fold: an optimiser step that leaves them alone (frozen encoders, a score-head-only fine-tune) keeps them
            sample = self._entry_sample()
            key = self._pattern_key()
            hit = getattr(self, "_pt_cache", None)
            if hit is not None and hit[0] == key:
                pt = hit[1]
            else:
                n_types = {"all": 3, "1-hop": 2, "cn": 1}[self.mask]
                sd = {k: v for k, v in self.state_dict().items()}
                pt = patterns.build(sd, self.dim, n_types, sample, device=self.device)
                self._pt_cache = (key, pt)
            w["patterns"] = pt
        return pt

    def _pattern_key(self):
        ps = [self.att_layers[0].att.lin_r.weight]
        for name in ("ppr_encoder_cn", "ppr_encoder_onehop", "ppr_encoder_non1hop"):
            mod = getattr(self, name, None)
            if mod is not None:
                ps += list(mod.parameters())
        return tuple((p.data_ptr(), p._version) for p in ps)

    def _prepare_pair_major(self, adj_mask=None):
        """Everything the pair-major attention decides or builds on the host -- the select4 / select3 gate (a 4,096-pair
        selection + the table build, once per parameter version) and the activation-pattern tables -- BEFORE a side
        stream is forked: none of it may run between the fork and the join of a step."""
        if self.dim >= 128 and self.use_fused_attention and self._uses_rows() and self._uses_select4(adj_mask):
            self._pattern_tables(self._fold())

    @_on_device
    def _flip_stats(self):
        """(raw, left): mean number of hidden units of the PE MLPs (both argument orders, ``2 D`` per entry) whose ReLU
        state differs from the one at (0, 0), over the entries of ``_entry_sample()`` -- all of them / only those of
        entries whose patterns ``_pattern_tables`` does not hold (what the exact path of the kernel behind select4 has to
        find and correct one by one).  Once per parameter version."""
        w = self._fold()
        hit = getattr(self, "_flip_est", None)
        if hit is not None and hit[0] is self._folded[0]:
            return hit[1]
        with torch.no_grad():
            sample = self._entry_sample()
            pt = self._pattern_tables(w) if self.dim >= 128 else None
            raw, left, total = 0.0, 0.0, 0
            for t in range({"all": 3, "1-hop": 2, "cn": 1}[self.mask]):
                pa, pb = sample[t]
                total += pa.numel()
                if pa.numel() == 0:
                    continue
                tab, st = w["flip_tab"][t], w["pe_stat"][t]     # rows (ta, tc, td, beta) times the unit's sign at (0, 0)
                slow, named = None, (None, None)
                if pt is not None:
                    # entries outside the no-flip square whose cells are flagged: the exact path, which starts from the
                    # tabulated patterns the two cells name
                    ia, ib = patterns.cell_index(pa, pt["geo"]), patterns.cell_index(pb, pt["geo"])
                    g = pt["grid"][t]
                    g1, g2 = g[ia, ib].long(), g[ib, ia].long()
                    slow = (((g1 | g2) & patterns.AMBIGUOUS) != 0) & (torch.maximum(pa, pb) > st[7])
                    sh = torch.arange(32, device=self.device)
                    sgn = pt["sign"][t].long() & 0xffffffff        # [NPAT, D / 32]: units that differ from pattern 0
                    bits = lambda ids: ((sgn[ids & 31][:, :, None] >> sh) & 1).reshape(ids.numel(), -1)[:, :self.dim].bool()
                    named = (bits(g1), bits(g2))
                for (x, y), ref in zip(((pa, pb), (pb, pa)), named):
                    var = st[0] * x * x + st[1] * y * y + st[2] + 2.0 * (st[3] * x * y + st[4] * x + st[5] * y)
                    r = torch.rsqrt(var.clamp_min(0.0) + 1e-5)
                    z = r[:, None] * (x[:, None] * tab[:, 0] + y[:, None] * tab[:, 1] + tab[:, 2]) + tab[:, 3]
                    raw += float((z < 0).sum().item())
                    if slow is None:
                        left += float((z < 0).sum().item())
                    else:
                        left += float((((z < 0) ^ ref).sum(dim=1) * slow).sum().item())
            est = (raw / max(1, total), left / max(1, total))
        self._flip_est = (self._folded[0], est)
        return est

    def flips_per_entry(self, raw: bool = False) -> float:
        """Flipped hidden units per selected entry that the attention kernel has to find and correct one by one: behind
        select4 only those of entries whose patterns are not tabulated, otherwise (or with ``raw``) all of them.  This
        is what the cost of the activation-pattern kernels depends on; a property of the
        ``ppr_encoder_*`` weights (reference link_transformer.py:67-76) and of the PPR values."""
        r, left = self._flip_stats()
        return r if raw or not self._uses_select4() else left

    def attention_kernel(self) -> str:
        """The fp32 one-pass attention kernel this model runs: ``attention_impl`` with "auto" resolved.  Below D = 128
        the matrix-core kernel always wins; from D = 128 on "auto" looks at the weights: the
        activation-pattern kernel while ``flips_per_entry()`` is below the measured break-even, the matrix-core kernel
        (whose cost does not depend on the weights) above it."""
        if self.attention_impl == "auto":
            if self.dim < 128:
                return "mfma"
            # The estimate runs a 4,096-pair selection and reads counts back (host synchronisations): it is made once per
            # parameter version in evaluation mode -- but not inside a loop that alternates optimiser steps with scoring
            # (at most once per `flip_recheck_every` parameter versions), and never while training.
            self._fold()
            last = getattr(self, "_auto_choice", None)      # (parameter key, choice, refolds at the time)
            if last is not None and (last[0] is self._folded[0] or self.training or
                                     self._refolds - last[2] < self.flip_recheck_every):
                return last[1]
            if self.training:
                return "flip"
            choice = "flip" if self.flips_per_entry() <= self.flip_break_even() else "mfma"
            self._auto_choice = (self._folded[0], choice, self._refolds)
            return choice
        return self.attention_impl

    def _uses_rows(self) -> bool:
        """True when the one-pass attention runs pair-major and hands over finished rows (csrc/pair_rows.hip)."""
        return self.attention_rows and self.attention_kernel() == "flip"

    def _attention_rows(self, batch, x_node, test_set, adj_mask, side, out, n_counts, order=False, q=None):
        """q gather -> selection -> pair-major one-pass attention writing ``out[p] = [post_att_norm(attention output) |
        n_counts count features]`` (``out``: [BS, ld] fp32, ld % 4 == 0).  Returns the selection workspace; with
        ``order`` also (perm int32[BS], n_nonempty int64[1]): the pairs with selected nodes first, for
        ``lpf_tail_chain_rows_perm_*``."""
        lib, st, d = _lib.hip(), _stream(self.device), self.dim
        bs = batch.shape[1]
        w = self._fold()
        z = self._node_keys(x_node, w)
        if q is None:   # (score_pairs has it gathered by the elementwise branch's launch)
            with torch.cuda.stream(side if side is not None else torch.cuda.current_stream(self.device)):
                q = self._pair_q(batch, x_node, w)
        four = self._uses_select4(adj_mask)
        ws = self._select4_device(batch, test_set) if four else self._select_device(batch, test_set, adj_mask)
        if side is not None:
            _lib.stream_wait(torch.cuda.current_stream(self.device), side)
        layer = self.att_layers[0]
        units_cap = (ws.kept_cap + 15) // 16 + 1 if four else (3 * ws.ent_cap + 15) // 16 + 1
        pieces = self._workspace("att_pieces", units_cap * 2 * int(lib.lpf_pair_rows_piece_floats(d)), torch.float32, st)
        extra = ()
        if order:
            perm = self._workspace("att_perm", bs, torch.int32, st)
            nfull = self._workspace("att_nfull", 1, torch.int64, st)
            extra = (ptr(perm), ptr(nfull)) if four else \
                (ptr(perm), ptr(self._zero_workspace("att_perm_lb", 2 * _lib.ROWS_PERM_LB_WORDS, st)), ptr(nfull))
        zt = self._z_bf16(z) if self.precision == "bf16" else z
        if four:    # (the base vectors of the tabulated activation patterns + the grid that finds an entry's two)
            pt = self._pattern_tables(w)
            geo = pt["geo"]
            bases = (ptr(pt["base"]), ptr(pt["grid"]), ptr(pt["sign"]), geo["n"], geo["shift"], geo["base"], geo["ofs"])
        else:       # (the one pattern of (0, 0))
            bases = (ptr(w["flip_base"]),)
        tabs = (ptr(zt), zt.stride(0), ptr(q), q.stride(0), ptr(w["flip_tab"]), ptr(w["pe_stat"]), *bases,
                ptr(w["wfold_t"]), ptr(w["att"]), ptr(layer.att.bias), ptr(layer.post_att_norm.weight),
                ptr(layer.post_att_norm.bias), n_counts, ptr(ws.ctl), ptr(pieces), units_cap, ptr(out), out.stride(0))
        with KernelTimer.span("pair_attention_rows"):
            if four:
                name = "lpf_pair_attention_rows4" + ("_zbf16" if self.precision == "bf16" else "_f32")
                check(getattr(lib, name)(d, bs, ptr(ws.pair_tab), ptr(ws.blk_cnt), ptr(ws.entries), ws.ent_cap, *tabs,
                                         *(extra or (None, None)), st), name)
            else:
                name = ("lpf_pair_attention_rows" + ("_perm" if order else "") +
                        ("_zbf16" if self.precision == "bf16" else "_f32"))
                check(getattr(lib, name)(d, bs, ptr(ws.type_ptr), ptr(ws.entries), ws.ent_cap, *tabs, *extra, st), name)
        return (ws, perm, nfull) if order else ws

    def _fused_attention(self, batch, x_node, test_set, adj_mask, side, q=None):
        """q gather (side stream) -> selection (two launches, nothing read back) -> one-pass attention.  Returns the
        selection workspace and the record buffers (part, bnd, units_cap) for ``lpf_tail_chain_merge_*`` /
        ``lpf_pair_attention_merge_f32``."""
        lib, st, d = _lib.hip(), _stream(self.device), self.dim
        bs = batch.shape[1]
        w = self._fold()
        z = self._node_keys(x_node, w)
        if q is None:
            with torch.cuda.stream(side if side is not None else torch.cuda.current_stream(self.device)):
                q = self._pair_q(batch, x_node, w)
        ws = self._select_device(batch, test_set, adj_mask)
        if side is not None:
            _lib.stream_wait(torch.cuda.current_stream(self.device), side)
        rs = d + 4
        units_cap = (ws.ent_cap + 15) // 16 + 1
        part = self._workspace("att_part", 3 * bs * rs, torch.float32, st)
        bnd = self._workspace("att_bnd", 3 * units_cap * 2 * rs, torch.float32, st)
        with KernelTimer.span("pair_attention_fused"):
            if self.precision == "bf16" and self.attention_kernel() == "flip":
                zb = self._z_bf16(z)      # bf16 node table, fp32 arithmetic (no D x D product to run in bf16)
                check(lib.lpf_pair_attention_flip_zbf16(
                    d, bs, ptr(ws.type_ptr), ptr(ws.entries), ws.ent_cap, ptr(zb), zb.stride(0), ptr(q), q.stride(0),
                    ptr(w["flip_tab"]), ptr(w["pe_stat"]), ptr(w["flip_base"]), ptr(w["wfold_t"]),
                    ptr(w["att"]), ptr(part), ptr(bnd), units_cap, st), "lpf_pair_attention_flip_zbf16")
            elif self.precision == "bf16":
                zb = self._z_bf16(z)
                check(lib.lpf_pair_attention_fused_bf16(
                    d, bs, ptr(ws.type_ptr), ptr(ws.entries), ws.ent_cap, ptr(zb), zb.stride(0), ptr(q),
                    q.stride(0), ptr(w["pe_tab"]), ptr(w["pe_stat"]), ptr(w["wfold_packed_bf16"]),
                    ptr(w["bfold"]), ptr(w["att"]), ptr(part), ptr(bnd), units_cap, st),
                    "lpf_pair_attention_fused_bf16")
            elif self.attention_kernel() == "flip":
                check(lib.lpf_pair_attention_flip_f32(
                    d, bs, ptr(ws.type_ptr), ptr(ws.entries), ws.ent_cap, ptr(z), z.stride(0), ptr(q), q.stride(0),
                    ptr(w["flip_tab"]), ptr(w["pe_stat"]), ptr(w["flip_base"]), ptr(w["wfold_t"]),
                    ptr(w["att"]), ptr(part), ptr(bnd), units_cap, st), "lpf_pair_attention_flip_f32")
            else:
                check(lib.lpf_pair_attention_fused_f32(
                    d, bs, ptr(ws.type_ptr), ptr(ws.entries), ws.ent_cap, ptr(z), z.stride(0), ptr(q),
                    q.stride(0), ptr(w["pe_tab"]), ptr(w["pe_stat"]), ptr(w["wfold_packed"]),
                    ptr(w["bfold"]), ptr(w["att"]), ptr(part), ptr(bnd), units_cap, st),
                    "lpf_pair_attention_fused_f32")
        return ws, part, bnd, units_cap

    def _pair_attention(self, batch, x_node, test_set, adj_mask, return_weights, stop_after_gather=False):
        """Selection -> PE + attention (+ post-norm) -> count features.  Returns (feats [BS, ld] = [attention output |
        counts | pad], att_weights or None, unchecked); the caller applies ``pairwise_lin`` (or its folded first
        layer).  ``unchecked``: the selection ran without a read-back, so the caller owes a ``check_selection()``.
        ``stop_after_gather``: return (G [BS, 4D+4], feats, False) right after the softmax-gather instead (the
        attention output projection then belongs to ``lpf_tail_chain_f32``)."""
        with torch.no_grad():
            lib, st, d = _lib.hip(), _stream(self.device), self.dim
            bs = batch.shape[1]
            if (d in (32, 64, 128, 256) and self.use_fused_attention and not return_weights and not stop_after_gather
                    and bs > 0):
                # one-pass attention on the selection regions, then the records merged straight into the feature
                # rows [post_att_norm(attention output) | counts] -- no reference-layout export, nothing read back
                # (a batch that overflows the selection workspace comes back as NaN: check_selection())
                ld = (d + self.count_dim + 3) // 4 * 4
                feats = torch.empty(bs, ld, dtype=torch.float32, device=self.device)
                if ld > d + self.count_dim:
                    feats[:, d + self.count_dim:].zero_()
                layer = self.att_layers[0]
                self._prepare_pair_major(adj_mask)
                side = self._fork()
                if self._uses_rows():
                    self._attention_rows(batch, x_node, test_set, adj_mask, side, feats, self.count_dim)
                    self._last_att = feats[:, :d]
                    return feats, None, True
                ws, part, bnd, units_cap = self._fused_attention(batch, x_node, test_set, adj_mask, side)
                with KernelTimer.span("pair_attention_merge"):
                    check(lib.lpf_pair_attention_merge_f32(
                        bs, d, self.count_dim, ptr(part), ptr(bnd), units_cap, ptr(ws.type_ptr),
                        ptr(layer.att.bias), ptr(layer.post_att_norm.weight), ptr(layer.post_att_norm.bias),
                        ptr(ws.ctl), ptr(feats), ld, st), "lpf_pair_attention_merge_f32")
                self._last_att = feats[:, :d]
                return feats, None, True   # (unchecked: calc_pairwise reads the status back after queueing its own work)
            w = self._fold()
            z = self._node_keys(x_node, w)
            side = self._fork()
            with torch.cuda.stream(side if side is not None else torch.cuda.current_stream(self.device)):
                q = self._pair_q(batch, x_node, w)

            s = self._select(batch, test_set, adj_mask)
            if side is not None:
                _lib.stream_wait(torch.cuda.current_stream(self.device), side)  # q (and the elementwise branch) are done
            score = self._workspace("score", s["cap"], torch.float32, st)
            with KernelTimer.span("pair_scores"):
                check(lib.lpf_pair_scores_f32(d, ptr(s["type_ptr"]), bs, ptr(s["sel_pair"]), ptr(s["sel_node"]),
                                              ptr(s["sel_pa"]), ptr(s["sel_pb"]), ptr(z), z.stride(0), ptr(q),
                                              q.stride(0), ptr(w["pe_tab"]), ptr(w["pe_stat"]), ptr(w["wfold_packed"]),
                                              ptr(w["bfold"]), ptr(w["att"]), ptr(score), s["cap"], st),
                      "lpf_pair_scores_f32")
            ldg = 4 * d + 4
            g = torch.empty(bs, ldg, dtype=torch.float32, device=self.device)
            alpha = torch.empty(s["cap"], dtype=torch.float32, device=self.device) if return_weights else None
            with KernelTimer.span("pair_softmax_gather"):
                check(lib.lpf_pair_softmax_gather_f32(d, bs, ptr(s["type_ptr"]), ptr(s["sel_node"]), ptr(s["sel_pa"]),
                                                      ptr(s["sel_pb"]), ptr(score), ptr(z), z.stride(0),
                                                      ptr(w["pe_tab"]), ptr(w["pe_stat"]), ptr(g), ldg, ptr(alpha),
                                                      ptr(self._workspace("sg_heavy", bs + 1, torch.int32, st)), st),
                      "lpf_pair_softmax_gather_f32")
            feats = s["feats"]
            if stop_after_gather:
                return g, feats, False
            att_view = feats[:, :d]
            layer = self.att_layers[0]
            # sum_e alpha_e k_e + bias, then post_att_norm: one launch (or GEMM + LayerNorm for unbuilt shapes)
            t = self._chain_att.tables(w["wcat"], None, layer.post_att_norm.weight, layer.post_att_norm.bias)
            if self._chain_att.run(t, g[:, d:], relu=False, addend=g[:, :d], out=att_view) is None:
                gemm(g[:, d:], w["wcat"], None, addend=g[:, :d], out=att_view, tag="gemm_attn_out")
                layernorm_(att_view, layer.post_att_norm.weight, layer.post_att_norm.bias)
            self._last_att = att_view
            att_weights = None
            if return_weights:
                tp = s["type_ptr"][:3 * (bs + 1)].view(3, bs + 1)
                total = int(tp[:, bs].sum().item())
                att_weights = torch.stack((s["sel_pair"][:total].float(), alpha[:total]))
            return feats, att_weights, False   # (_select read the status back already)

    @_on_device
    def calc_pairwise(self, batch, X_node, test_set=False, adj_mask=None, return_weights=False, _out=None):
        """Pairwise branch (:132-178): selection -> PE + attention -> counts -> ``pairwise_lin``.
        Returns ([BS, D], att_weights or None)."""
        self._check_supported(heads_ok=True)
        _require_gpu(X_node, "calc_pairwise")
        if self._multi_head:
            pw = self._pair_stage_heads(batch, X_node, test_set, adj_mask, return_weights)[1]
            if _out is not None:
                _out.copy_(pw)
            return (pw if _out is None else _out), None
        with torch.no_grad():
            batch = self._prep_batch(batch)
            for _attempt in range(3):
                feats, att_weights, unchecked = self._pair_attention(batch, _as_f32_rows(X_node), test_set, adj_mask,
                                                                     return_weights)
                out = self.pairwise_lin.run(feats[:, :self.dim + self.count_dim], out=_out)
                # The callers of this API are the reference's loops: they fetch the predictions of every batch right
                # away and never heard of check_selection().  So the status of the read-back-free selection is read
                # here, AFTER this call's own launches are queued (the wait is the one the caller's .cpu() would pay),
                # and a batch that overflowed its workspace is simply run again on the re-sized one.
                # (node ids out of range raise IndexError from check_selection; `_out` then holds NaN rows)
                if not unchecked or self.check_selection():
                    break
            else:
                raise _lib.LpfError("calc_pairwise: the selection workspace could not be sized")
            return out, att_weights

    # ---------------------------------------------------------------------------------- folded score path
    def _score_fold(self, score_func):
        """Parameter-only fold across the module boundary: there is no non-linearity between the last Linear of
        ``elementwise_lin`` / ``pairwise_lin`` and the first Linear of the score head, so
            lins0([ew | pw]) = A_e r_e + A_p r_p + c,   A_e = W_s0[:, :D] W_e1,  A_p = W_s0[:, D:] W_p1,
            c = b_s0 + W_s0[:, :D] b_e1 + W_s0[:, D:] b_p1
        with r_e / r_p the hidden activations (after LayerNorm + ReLU) of the two MLPs.  Folded in float64."""
        d, pd = self.dim, self.dim + self.count_dim
        ps = [self.elementwise_lin.linears[1].weight, self.elementwise_lin.linears[1].bias,
              self.pairwise_lin.linears[1].weight, self.pairwise_lin.linears[1].bias,
              score_func.lins[0].weight, score_func.lins[0].bias]
        key = tuple((p.data_ptr(), p._version) for p in ps) + (id(score_func),)
        hit = getattr(self, "_score_fold_cache", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        we1, be1, wp1, bp1, ws0, bs0 = (p.detach().double().cpu() for p in ps)
        kpad = _pad4(d + pd)
        a = torch.zeros(ws0.shape[0], kpad, dtype=torch.float64)
        a[:, :d] = ws0[:, :d] @ we1
        a[:, d:d + pd] = ws0[:, d:] @ wp1
        c = bs0 + ws0[:, :d] @ be1 + ws0[:, d:] @ bp1
        out = (a.float().to(self.device), c.float().to(self.device), kpad)
        self._score_fold_cache = (key, out)
        return out

    def _tail_tables(self, score_func, a, c):
        """Device tables of ``lpf_tail_chain_f32`` (refreshed with the folds they are built from)."""
        layer, pw = self.att_layers[0], self.pairwise_lin
        ps = [layer.post_att_norm.weight, layer.post_att_norm.bias, pw.linears[0].weight, pw.linears[0].bias,
              pw.norm.weight, pw.norm.bias, score_func.lins[1].weight, score_func.lins[1].bias, layer.att.bias]
        self._fold()
        key = (tuple((p.data_ptr(), p._version) for p in ps), a.data_ptr(), self._folded[0])
        hit = getattr(self, "_tail_cache", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        w = self._fold()
        npy = lambda t: t.detach().float().cpu().numpy()  # noqa: E731
        tabs = fold.tail_chain_tables(npy(w["wcat"]), npy(ps[0]), npy(ps[1]), npy(ps[2]), npy(ps[3]), npy(ps[4]),
                                      npy(ps[5]), npy(a)[:, :self.dim + self.dim + self.count_dim], npy(c), npy(ps[6]),
                                      npy(ps[7]), self.dim, att_bias=npy(ps[8]))
        dev = {k: torch.from_numpy(v).to(self.device) for k, v in tabs.items()}
        # bf16 images of the two GEMM weights (same element order: a lane's four fp32 become its four bf16)
        for k in ("wB", "wC"):
            dev[k + "_bf16"] = torch.from_numpy(fold.to_bf16_bits(tabs[k]).view(np.int16)).to(self.device)
        self._tail_cache = (key, dev)
        return dev

    @_on_device
    def score_pairs(self, batch, X_node, score_func, test_set=False, adj_mask=None, logits=False):
        """``score_func(cat(elementwise_lin(x_a * x_b), calc_pairwise(...)[0]))`` -- the reference's scoring
        expression (src/train/testing.py:29-31,113-117) -- with the three Linear layers around the module boundary
        folded into one (``_score_fold``): probabilities (or logits) of shape [BS].  Falls back to the unfolded
        modules when the score head is not the two-layer MLP every script uses."""
        self._check_supported(heads_ok=True)
        _require_gpu(X_node, "score_pairs")
        two_layer = (len(score_func.lins) == 2 and score_func.lins[1].out_features == 1 and
                     len(self.elementwise_lin.linears) == 2 and len(self.pairwise_lin.linears) == 2 and
                     not (score_func.training and score_func.dropout > 0))
        if not two_layer or self._multi_head:
            feats = self.pair_features(batch, X_node, test_set=test_set, adj_mask=adj_mask)
            return score_func.logits(feats) if logits else score_func(feats)
        self._fold_memo = None
        self._fold_memo = self._fold()   # one walk over the parameters' version counters per call, not one per helper
        try:
            return self._score_pairs_folded(batch, X_node, score_func, test_set, adj_mask, logits)
        finally:
            self._fold_memo = None

    def _score_pairs_folded(self, batch, X_node, score_func, test_set, adj_mask, logits):
        with torch.no_grad():
            d, pd = self.dim, self.dim + self.count_dim
            batch = self._prep_batch(batch)
            bs = batch.shape[1]
            x_node = _as_f32_rows(X_node)
            a, c, kpad = self._score_fold(score_func)
            r = torch.empty(bs, kpad, dtype=torch.float32, device=self.device)  # [r_e | r_p | pad]
            ew, pw = self.elementwise_lin, self.pairwise_lin
            one_pass = d in (32, 64, 128, 256) and self.use_tail_chain and self.use_fused_attention and bs > 0
            # the attention's query q = Y[a] + Y[b] is gathered by the launch of the elementwise branch (same ids, a
            # second table): one launch less per step -- 19 us of a 193 us pipelined step as a launch of its own
            q_side = None
            if one_pass and self.query_from == "table" and (d < 256 or self._uses_rows()):
                q_side = (self._node_y(x_node, self._fold()), torch.empty(bs, d, dtype=torch.float32, device=self.device))
            if one_pass:
                self._prepare_pair_major(adj_mask)
            side = self._fork()
            with torch.cuda.stream(side if side is not None else torch.cuda.current_stream(self.device)):
                t = ew._chain1.tables(ew.linears[0].weight, ew.linears[0].bias, ew.norm.weight, ew.norm.bias)
                if ew._chain1.run(t, x_node, relu=True, batch=batch, in_mode=1, out=r[:, :d], side=q_side) is None:
                    q_side = None
                    prod = torch.empty(bs, d, dtype=torch.float32, device=self.device)
                    with KernelTimer.span("pair_gather"):
                        check(_lib.hip().lpf_pair_gather_f32(bs, d, ptr(batch), batch.stride(0), x_node.shape[0], ptr(x_node),
                                                             x_node.stride(0), ptr(prod), d, None, 0,
                                                             _stream(self.device)), "lpf_pair_gather_f32")
                    gemm(prod, ew._pads[0].get(ew.linears[0].weight), ew.linears[0].bias, out=r[:, :d])
                    layernorm_(r[:, :d], ew.norm.weight, ew.norm.bias, relu=True)
            if (d in (32, 64, 128, 256) and self.use_tail_chain and self.use_fused_attention and bs > 0 and
                    (self._uses_rows() or d == 256)):
                # hot path: 2 selection launches (nothing read back) -> attention leaving finished rows -> the dense
                # tail that is left: pairwise_lin's first layer, folded score head, sigmoid.  The rows come from the
                # pair-major kernel or (D = 256 without it: the record-merging tail has no instantiation that wide) from
                # the unit-major kernel + lpf_pair_attention_merge_f32
                lib, st = _lib.hip(), _stream(self.device)
                order = None
                if self._uses_rows():
                    rows = self._zero_workspace("att_rows", bs * (d + 4), st).view(bs, d + 4)   # (pad columns stay zero)
                    if self.tail_skip_empty:
                        ws, *order = self._attention_rows(batch, x_node, test_set, adj_mask, side, rows, self.count_dim,
                                                          order=True, q=q_side and q_side[1])
                    else:
                        ws = self._attention_rows(batch, x_node, test_set, adj_mask, side, rows, self.count_dim,
                                                  q=q_side and q_side[1])
                else:
                    rows, _, _ = self._pair_attention(batch, x_node, test_set, adj_mask, False)   # [BS, D + 4]
                    ws = self._sel_ws(st, bs)
                tt = self._tail_tables(score_func, a, c)
                res = torch.empty(bs, dtype=torch.float32, device=self.device)
                with KernelTimer.span("tail_chain"):
                    b16 = self.tail_precision == "bf16"
                    # (behind select4 the attention kernel leaves no rows for pairs without selected nodes: the tail
                    #  takes the constant row itself; behind select3 the rows are there and row_empty stays NULL)
                    row0 = ptr(tt["row_empty"]) if (order and self._uses_select4(adj_mask)) else None
                    extra = (ptr(order[0]), ptr(order[1]), ptr(tt["bC_empty"]), row0) if order else ()
                    name = "lpf_tail_chain_rows" + ("_perm" if order else "") + ("_bf16" if b16 else "_f32")
                    sfx = "_bf16" if b16 else ""
                    check(getattr(lib, name)(
                        bs, d, self.count_dim, ptr(rows), rows.stride(0), ptr(tt["wB" + sfx]),
                        ptr(tt["bB"]), ptr(tt["lnB_g"]), ptr(tt["lnB_b"]), ptr(r), r.stride(0),
                        ptr(tt["wC" + sfx]), ptr(tt["bC"]), ptr(tt["w_dot"]), ptr(tt["b_dot"]),
                        ptr(ws.ctl), *extra, ptr(res) if logits else None, None if logits else ptr(res), st), name)
                return res
            if d in (32, 64, 128) and self.use_tail_chain and self.use_fused_attention and bs > 0:
                # 2 selection launches (nothing read back) -> one-pass attention (records) -> merged dense tail
                lib, st = _lib.hip(), _stream(self.device)
                ws, part, bnd, units_cap = self._fused_attention(batch, x_node, test_set, adj_mask, side,
                                                                 q=q_side and q_side[1])
                tt = self._tail_tables(score_func, a, c)
                res = torch.empty(bs, dtype=torch.float32, device=self.device)
                with KernelTimer.span("tail_chain"):
                    b16 = self.tail_precision == "bf16"
                    fn = lib.lpf_tail_chain_merge_bf16 if b16 else lib.lpf_tail_chain_merge_f32
                    check(fn(
                        bs, d, self.count_dim, ptr(part), ptr(bnd), units_cap, ptr(ws.type_ptr),
                        ptr(self.att_layers[0].att.bias),
                        ptr(tt["lnA_g"]), ptr(tt["lnA_b"]), ptr(tt["wB_bf16" if b16 else "wB"]), ptr(tt["bB"]),
                        ptr(tt["lnB_g"]), ptr(tt["lnB_b"]), ptr(r), r.stride(0), ptr(tt["wC_bf16" if b16 else "wC"]),
                        ptr(tt["bC"]), ptr(tt["w_dot"]),
                        ptr(tt["b_dot"]), ptr(ws.ctl), ptr(res) if logits else None, None if logits else ptr(res),
                        st), "lpf_tail_chain_merge")
                return res
            if d in (32, 64, 128) and self.use_tail_chain:  # attention output + pairwise hidden + head: one launch
                g, feats, _ = self._pair_attention(batch, x_node, test_set, adj_mask, False, stop_after_gather=True)
                tt = self._tail_tables(score_func, a, c)
                res = torch.empty(bs, dtype=torch.float32, device=self.device)
                with KernelTimer.span("tail_chain"):
                    check(_lib.hip().lpf_tail_chain_f32(
                        bs, d, self.count_dim, ptr(g), g.stride(0), ptr(tt["wA"]), ptr(tt["lnA_g"]), ptr(tt["lnA_b"]),
                        feats.data_ptr() + 4 * d, feats.stride(0), ptr(tt["wB"]), ptr(tt["bB"]), ptr(tt["lnB_g"]),
                        ptr(tt["lnB_b"]), ptr(r), r.stride(0), ptr(tt["wC"]), ptr(tt["bC"]), ptr(tt["w_dot"]),
                        ptr(tt["b_dot"]), ptr(res) if logits else None, None if logits else ptr(res),
                        _stream(self.device)), "lpf_tail_chain_f32")
                return res
            feats, _, _ = self._pair_attention(batch, x_node, test_set, adj_mask, False)  # joins the side stream
            if kpad > d + pd:
                r[:, d + pd:].zero_()
            xin = feats[:, :pd]
            t = pw._chain1.tables(pw.linears[0].weight, pw.linears[0].bias, pw.norm.weight, pw.norm.bias)
            if pw._chain1.run(t, xin, relu=True, out=r[:, d:d + pd]) is None:
                gemm(xin, pw._pads[0].get(pw.linears[0].weight), pw.linears[0].bias, out=r[:, d:d + pd])
                layernorm_(r[:, d:d + pd], pw.norm.weight, pw.norm.bias, relu=True)
            l2 = score_func.lins[1]
            t = score_func._chain_fold.tables(a, c, None, None, l2.weight, l2.bias)
            res = score_func._chain_fold.run(t, r, relu=True, want_logit=logits)
            if res is None:
                hid = gemm(r, a, c, relu=True)
                res = score_func._tail(hid, not logits)
            return res

    @_on_device
    def forward(self, batch, adj_prop=None, adj_mask=None, test_set=False, return_weights=False):
        """Link representations [BS, 2D] = [elementwise branch | pairwise branch] (reference :82-107).  Like the
        reference, every call re-runs the encoder; evaluation loops that propagate once should call ``propagate``
        + ``elementwise_lin`` + ``calc_pairwise`` (src/train/testing.py:96-121)."""
        if self.training:
            # autograd graph on the device: GEMMs, aggregation and selection through the C ABI (lpformer_amd/train.py)
            self._check_supported(train_ok=True, heads_ok=True)
            if return_weights:
                raise NotImplementedError("return_weights is an evaluation-time debugging aid (layers.py:69-75)")
            from . import train as lpf_train
            return lpf_train.forward_train(self, batch, adj_prop, adj_mask, test_set)
        self._check_supported(heads_ok=True)
        with torch.no_grad():
            batch = self._prep_batch(batch)
            x_node = self._propagate_reusing(adj_prop, test_set)
            out = self.pair_features(batch, x_node, test_set=test_set, adj_mask=adj_mask,
                                     return_weights=return_weights)
            return out

    def _pair_stage_heads(self, batch, X_node, test_set, adj_mask, return_weights=False):
        """Evaluation-mode pair stage of a multi-head model: lpformer_amd/train.py ``pair_stage`` with every dropout and the
        random attention drop off, no autograd graph -> (elementwise [BS, D], pairwise [BS, D])."""
        if return_weights:
            raise NotImplementedError("return_weights (layers.py:69-75: the heads' weights averaged, a debugging aid) is "
                                      "available for num_heads = 1")
        from . import train as lpf_train
        with torch.no_grad():
            return lpf_train.pair_stage(self, _as_f32_rows(X_node), self._prep_batch(batch), adj_mask, test_set,
                                        training=False)

    def _propagate_reusing(self, adj_prop, test_set):
        """``propagate`` for ``forward`` in eval mode.  The reference's evaluation loop calls ``model(edges)`` per batch
        and so re-runs the whole encoder per batch (src/train/testing.py:87 -> link_transformer.py:100) on unchanged
        inputs; here the output is kept while NOTHING it depends on has changed: the graph object of this
        ``test_set``, the feature tensor (identity and version), every encoder parameter (storage and version: optimiser
        steps, ``load_state_dict`` and ``.to()`` all change one of them), the precision switch and the shard layout.
        The objects in the key are held, so an address cannot be recycled under it.  ``reuse_encoder_output = False``
        turns this off; an ``adj_prop`` override is never cached."""
        if adj_prop is not None or not self.reuse_encoder_output:
            return self.propagate(adj_prop, test_set)
        obj, feats = self._data_obj("adj", test_set), self.data["x"]
        params = list(self.node_encoder.parameters()) + list(self.gnn_norm.parameters())
        key = (id(obj), id(feats), getattr(feats, "_version", 0),
               tuple((q.data_ptr(), q._version) for q in params), self.encoder_precision, self._shard,
               self.encoder_mode)
        hit = self._enc_cache
        if hit is not None and hit[0] == key and hit[1] is obj and hit[2] is feats:
            return hit[3]
        x_node = self.propagate(None, test_set)
        self._enc_cache = (key, obj, feats, x_node)
        return x_node

    @_on_device
    def pair_features(self, batch, X_node, test_set=False, adj_mask=None, return_weights=False):
        """[elementwise_lin(X[a]*X[b]) | calc_pairwise(...)] written straight into one [BS, 2D] buffer."""
        self._check_supported(heads_ok=True)
        if self._multi_head:
            ew, pw = self._pair_stage_heads(batch, X_node, test_set, adj_mask, return_weights)
            comb = torch.cat([ew, pw], dim=-1)
            return (comb, None) if return_weights else comb
        with torch.no_grad():
            d = self.dim
            batch = self._prep_batch(batch)
            bs = batch.shape[1]
            x_node = _as_f32_rows(X_node)
            comb = torch.empty(bs, 2 * d, dtype=torch.float32, device=self.device)
            side = self._fork()
            with torch.cuda.stream(side if side is not None else torch.cuda.current_stream(self.device)):
                self.elementwise_lin.run(x_node, out=comb[:, :d], batch=batch, in_mode=1)
            _, attw = self.calc_pairwise(batch, x_node, test_set, adj_mask, return_weights, _out=comb[:, d:])
            return (comb, attw) if return_weights else comb
