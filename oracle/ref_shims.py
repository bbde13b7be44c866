"""Stand-ins for the third-party packages the reference imports but that are absent here.

TEST INFRASTRUCTURE ONLY.  Used by ``tests/golden/make_golden.py`` (run in the build
container, where ``/root/reference`` is mounted) so that the reference's own Python
(``src/models/link_transformer.py``, ``src/modules/layers.py``, ``src/modules/node_encoder.py``,
``src/models/other_models.py``, ``src/util/calc_ppr_scores.py``) executes UNMODIFIED on CPU and
produces the golden vectors committed under ``tests/golden/``.

Nothing here is reference code.  It restates, from their published behaviour, only the
third-party names the reference imports:

* ``torch_geometric==2.2.0`` (reference ``requirements.txt:2``):
    ``nn.GCNConv`` + ``gcn_norm`` on a SparseTensor (call sites ``src/models/other_models.py:35,45,48,66``),
    ``nn.conv.MessagePassing`` lift/aggregate (``src/modules/layers.py:88,109,173``),
    ``utils.softmax`` (``layers.py:220``), ``nn.dense.linear.Linear`` and ``nn.inits.glorot/zeros``
    (``layers.py:130-131,153-157``), ``typing.OptTensor``, ``utils.coalesce/to_undirected``
    (``src/util/calc_ppr_scores.py:14``).
* ``torch_scatter`` (un-pinned, ``requirements.txt:6``): ``scatter`` (``src/models/link_transformer.py:4,371,383``).
* ``torch_sparse`` (un-pinned, ``requirements.txt:5``): ``SparseTensor`` (``other_models.py:5``,
  ``calc_ppr_scores.py:11``), ``matmul.spmm_*`` (imported, never called).
* ``numba`` (``calc_ppr_scores.py:134``): ``jit`` -> identity, ``prange`` -> ``range``, ``int64`` -> ``int``.
  Python dict insertion order and ``list.pop()`` match numba's typed containers.
* ``ogb.linkproppred`` (module-level import only).

The reference publishes no tests for this boundary, so these semantics are the spec
("parity unpinned" at the third-party boundary).
"""
import inspect
import math
import sys
import types
from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor


# --------------------------------------------------------------------------- torch_scatter
def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
    assert out is None
    index = index.long()
    if dim < 0:
        dim += src.dim()
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    shape = list(src.shape)
    shape[dim] = dim_size
    idx = index
    if src.dim() > 1:
        view = [1] * src.dim()
        view[dim] = -1
        idx = index.view(view).expand_as(src)
    if reduce in ("sum", "add"):
        return torch.zeros(shape, dtype=src.dtype).scatter_add_(dim, idx, src)
    if reduce == "max":
        res = torch.zeros(shape, dtype=src.dtype)
        return res.scatter_reduce_(dim, idx, src, reduce="amax", include_self=False)
    raise NotImplementedError(reduce)


# --------------------------------------------------------------------------- torch_sparse
class SparseTensor:
    """Minimal COO container with torch_sparse's sort-on-construct behaviour (no dedup)."""

    def __init__(self, row, col, value=None, sparse_sizes=None, is_sorted=False):
        row, col = row.long(), col.long()
        M, N = sparse_sizes
        if not is_sorted:
            perm = (row * N + col).argsort(stable=True)
            row, col = row[perm], col[perm]
            value = value[perm] if value is not None else None
        self.row, self.col, self.value, self._sizes = row, col, value, (int(M), int(N))

    @classmethod
    def from_edge_index(cls, edge_index, edge_attr=None, sparse_sizes=None, is_sorted=False):
        return cls(edge_index[0], edge_index[1], edge_attr, sparse_sizes, is_sorted)

    def sparse_sizes(self):
        return self._sizes

    def size(self, d):
        return self._sizes[d]

    def has_value(self):
        return self.value is not None

    def fill_value(self, v):
        return SparseTensor(self.row, self.col, torch.full((self.row.numel(),), float(v)), self._sizes, True)

    def coo(self):
        return self.row, self.col, self.value

    def to_symmetric(self, reduce="sum"):
        N = max(self._sizes)
        row = torch.cat([self.row, self.col])
        col = torch.cat([self.col, self.row])
        key = row * N + col
        uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
        value = None
        if self.value is not None:
            v2 = torch.cat([self.value, self.value])
            value = torch.zeros(uniq.numel(), dtype=v2.dtype).scatter_add_(0, inv, v2)
        return SparseTensor(uniq // N, uniq % N, value, (N, N), True)

    def coalesce(self, reduce="sum"):
        N = self._sizes[1]
        key = self.row * N + self.col
        uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
        value = None
        if self.value is not None:
            value = torch.zeros(uniq.numel(), dtype=self.value.dtype).scatter_add_(0, inv, self.value)
        return SparseTensor(uniq // N, uniq % N, value, self._sizes, True)

    def to_torch_sparse_coo_tensor(self):
        v = self.value if self.value is not None else torch.ones(self.row.numel())
        return torch.sparse_coo_tensor(torch.stack([self.row, self.col]), v, self._sizes)

    def to(self, *a, **k):
        return self

    to_device = to


def _fill_diag(adj: SparseTensor, fill: float) -> SparseTensor:
    """torch_sparse.fill_diag: existing diagonal entries are REPLACED, missing ones inserted."""
    M, N = adj.sparse_sizes()
    keep = adj.row != adj.col
    d = torch.arange(min(M, N))
    row = torch.cat([adj.row[keep], d])
    col = torch.cat([adj.col[keep], d])
    val = torch.cat([adj.value[keep], torch.full((d.numel(),), float(fill), dtype=adj.value.dtype)])
    return SparseTensor(row, col, val, (M, N), False)


def _gcn_norm(adj_t: SparseTensor) -> SparseTensor:
    """torch_geometric 2.2.0 ``gcn_norm`` SparseTensor branch, add_self_loops=True, improved=False."""
    if not adj_t.has_value():
        adj_t = adj_t.fill_value(1.0)
    adj_t = _fill_diag(adj_t, 1.0)
    M, N = adj_t.sparse_sizes()
    deg = torch.zeros(M, dtype=adj_t.value.dtype).scatter_add_(0, adj_t.row, adj_t.value)
    dis = deg.pow(-0.5)
    dis.masked_fill_(dis == float("inf"), 0.0)
    val = adj_t.value * dis[adj_t.row]
    val = val * dis[adj_t.col]
    return SparseTensor(adj_t.row, adj_t.col, val, (M, N), True)


def _spmm_add(adj: SparseTensor, x: Tensor) -> Tensor:
    out = torch.zeros(adj.size(0), x.size(1), dtype=x.dtype)
    return out.index_add_(0, adj.row, x[adj.col] * adj.value.unsqueeze(-1))


# --------------------------------------------------------------------------- torch_geometric
def glorot(t):
    if t is not None:
        stdv = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
        t.data.uniform_(-stdv, stdv)


def zeros(t):
    if t is not None:
        t.data.fill_(0)


class Linear(nn.Module):
    def __init__(self, in_channels, out_channels, bias=True, weight_initializer=None, bias_initializer=None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.weight)
        if self.bias is not None:
            bound = 1.0 / math.sqrt(self.in_channels)
            self.bias.data.uniform_(-bound, bound)

    def forward(self, x):
        return nn.functional.linear(x, self.weight, self.bias)


class GCNConv(nn.Module):
    def __init__(self, in_channels, out_channels, cached=False, normalize=True, **kw):
        super().__init__()
        self.cached, self.normalize = cached, normalize
        self._cached_adj_t = None
        self.lin = Linear(in_channels, out_channels, bias=False, weight_initializer="glorot")
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def reset_parameters(self):
        self.lin.reset_parameters()
        zeros(self.bias)
        self._cached_adj_t = None

    def forward(self, x, adj_t):
        if self.normalize:
            cache = self._cached_adj_t
            if cache is None:
                adj_t = _gcn_norm(adj_t)
                if self.cached:
                    self._cached_adj_t = adj_t
            else:
                adj_t = cache
        x = self.lin(x)
        out = _spmm_add(adj_t, x)
        return out + self.bias


def softmax(src, index, ptr=None, num_nodes=None, dim=0):
    N = int(index.max()) + 1 if num_nodes is None else num_nodes
    src_max = scatter(src.detach(), index, dim, dim_size=N, reduce="max")
    out = (src - src_max.index_select(dim, index)).exp()
    out_sum = scatter(out, index, dim, dim_size=N, reduce="sum") + 1e-16
    return out / out_sum.index_select(dim, index)


class MessagePassing(nn.Module):
    """Just enough of PyG's propagate(): lift `_i`/`_j` args, message, scatter-add to size[i]."""

    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2, **kw):
        super().__init__()
        assert aggr == "add"
        self.flow, self.node_dim = flow, node_dim

    def propagate(self, edge_index, size=None, **kwargs):
        i, j = (1, 0) if self.flow == "source_to_target" else (0, 1)
        sizes = [None, None]
        args = {}
        for name in inspect.signature(self.message).parameters:
            if name[-2:] in ("_i", "_j"):
                dim = j if name.endswith("_j") else i
                data = kwargs.get(name[:-2])
                if isinstance(data, (tuple, list)):
                    if isinstance(data[1 - dim], Tensor) and sizes[1 - dim] is None:
                        sizes[1 - dim] = data[1 - dim].size(self.node_dim)
                    data = data[dim]
                if isinstance(data, Tensor):
                    if sizes[dim] is None:
                        sizes[dim] = data.size(self.node_dim)
                    data = data.index_select(self.node_dim, edge_index[dim].long())
                args[name] = data
            elif name == "index":
                args[name] = edge_index[i].long()
            elif name == "ptr":
                args[name] = None
            elif name == "size_i":
                args[name] = None  # filled below
            else:
                args[name] = kwargs.get(name)
        if "size_i" in args:
            args["size_i"] = sizes[i]
        out = self.message(**args)
        return scatter(out, edge_index[i].long(), dim=self.node_dim, dim_size=sizes[i], reduce="sum")


def coalesce(edge_index, edge_attr=None, num_nodes=None, reduce="add"):
    N = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    key = torch.unique(edge_index[0].long() * N + edge_index[1].long(), sorted=True)
    return torch.stack([key // N, key % N])


def to_undirected(edge_index, edge_attr=None, num_nodes=None, reduce="add"):
    """torch_geometric.utils.to_undirected (2.2.0): both directions of every edge, duplicates merged, sorted by
    (row, col); with ``edge_attr`` the attributes of merged entries are reduced ("add": summed) and the pair
    (edge_index, edge_attr) is returned (src/util/read_datasets.py:96,276)."""
    if isinstance(edge_attr, int):      # (the positional num_nodes of the old two-argument call)
        edge_attr, num_nodes = None, edge_attr
    both = torch.cat([edge_index, edge_index.flip(0)], dim=1)
    if edge_attr is None:
        return coalesce(both, num_nodes=num_nodes)
    if reduce not in ("add", "sum"):
        raise NotImplementedError(reduce)
    N = int(both.max()) + 1 if num_nodes is None else num_nodes
    key = both[0].long() * N + both[1].long()
    uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
    attr = torch.cat([edge_attr, edge_attr], dim=0)
    out = torch.zeros((uniq.numel(),) + tuple(attr.shape[1:]), dtype=attr.dtype).index_add_(0, inv, attr)
    return torch.stack([uniq // N, uniq % N]), out


# --------------------------------------------------------------------------- registration
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def degree(index, num_nodes=None, dtype=None):
    """torch_geometric.utils.degree: occurrences of every node id in ``index`` (src/util/read_datasets.py:116,240)."""
    n = int(index.max()) + 1 if num_nodes is None else int(num_nodes)
    out = torch.zeros(n, dtype=dtype or torch.float32, device=index.device)
    return out.scatter_add_(0, index.long(), torch.ones(index.numel(), dtype=out.dtype, device=index.device))


def install():
    """Register the stand-in modules in ``sys.modules`` (idempotent)."""
    if "torch_geometric" in sys.modules and getattr(sys.modules["torch_geometric"], "_lpf_shim", False):
        return
    _mod("torch_scatter", scatter=scatter)
    ts_mm = _mod("torch_sparse.matmul", spmm_max=None, spmm_mean=None, spmm_add=_spmm_add)
    _mod("torch_sparse", SparseTensor=SparseTensor, matmul=ts_mm)
    tg = _mod("torch_geometric", _lpf_shim=True)
    tg.nn = _mod("torch_geometric.nn", GCNConv=GCNConv)
    tg.nn.conv = _mod("torch_geometric.nn.conv", MessagePassing=MessagePassing)
    tg.nn.dense = _mod("torch_geometric.nn.dense")
    tg.nn.dense.linear = _mod("torch_geometric.nn.dense.linear", Linear=Linear)
    tg.nn.inits = _mod("torch_geometric.nn.inits", glorot=glorot, zeros=zeros)
    tg.utils = _mod("torch_geometric.utils", softmax=softmax, coalesce=coalesce, to_undirected=to_undirected, degree=degree)
    tg.typing = _mod("torch_geometric.typing", OptTensor=Optional[Tensor])
    tg.transforms = _mod("torch_geometric.transforms")

    def _jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    _mod("numba", jit=_jit, njit=_jit, prange=range, int64=int)
    ogb = _mod("ogb")
    ogb.linkproppred = _mod("ogb.linkproppred", PygLinkPropPredDataset=None, Evaluator=None)
    try:
        import joblib  # noqa: F401  (installed here; only shim when missing)
    except ImportError:
        _mod("joblib")
