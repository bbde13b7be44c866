"""CSR graph containers and builders (host side).

The reference keeps its graph state as ``torch_sparse.SparseTensor`` / torch sparse COO tensors and materialises
BS x N COO temporaries per batch (src/util/read_datasets.py:85-129, src/models/link_transformer.py:229-237).
The MI355X path keeps three CSR structures with sorted int32 columns resident in HBM instead:

* the GCN-normalised propagation matrix (diagonal forced to 1, ``gcn_norm`` semantics),
* the 0/1 symmetric adjacency mask,
* the PPR matrix (fp32 values), plus an optional per-threshold prefiltered copy for the >1-hop candidates.

Everything here is one-time data preparation (the reference's ``cached=True`` / dataset-loading analogue).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import scipy.sparse as sp
import torch

from . import _lib


@dataclass
class CSR:
    """Host CSR: rowptr int64[n+1], col int32[nnz] ascending inside rows without duplicates (the kernels binary-search
    the rows and the model uploads these containers as they are), optional fp32 values."""
    rowptr: np.ndarray
    col: np.ndarray
    val: Optional[np.ndarray]
    n: int

    @property
    def nnz(self) -> int:
        return int(self.col.size)

    def to_device(self, device) -> "DeviceCSR":
        return DeviceCSR(torch.from_numpy(self.rowptr).to(device), torch.from_numpy(self.col).to(device),
                         None if self.val is None else torch.from_numpy(self.val).to(device), self.n, self)

    def to_torch_sparse_coo(self) -> torch.Tensor:
        """Same object type the reference stores in data['ppr'] / data['adj_mask']."""
        rows = np.repeat(np.arange(self.n, dtype=np.int64), np.diff(self.rowptr))
        idx = torch.from_numpy(np.stack([rows, self.col.astype(np.int64)]))
        v = torch.ones(self.nnz) if self.val is None else torch.from_numpy(self.val)
        return torch.sparse_coo_tensor(idx, v, (self.n, self.n)).coalesce()


@dataclass
class DeviceCSR:
    rowptr: torch.Tensor  # int64
    col: torch.Tensor     # int32
    val: Optional[torch.Tensor]
    n: int
    host: Optional[CSR] = None

    @property
    def nnz(self) -> int:
        return int(self.col.numel())

    def to_host(self) -> CSR:
        """Host copy (cached): the arrays of indexes that were built on the device are downloaded on first use."""
        if self.host is None:
            self.host = CSR(self.rowptr.cpu().numpy(), self.col.cpu().numpy(),
                            None if self.val is None else self.val.cpu().numpy(), self.n)
        return self.host


def fused_row_order(rowptr: torch.Tensor, lo: int, hi: int, long_threshold: int = 64, part: int = 256,
                    pad_hubs: bool = False):
    """Work order of rows [lo, hi) for ``lpf_gcn_layer_fused_f32`` (csrc/gcn_fused.hip).  Returns
    ``(order, hubs, parts)``: ``order`` int32 codes, 16 per tile -- hub rows first (more than ``long_threshold`` stored
    entries -- measured per layer, slice kernel + layer kernel, ppa-like graph (mean degree 74): 64: 843 + 740 us,
    128: 476 + 1,168, 256: 266 + 1,420, 512: 150 + 1,575; collab-like (mean 11): 32: 53 + 186, 64: 38 + 197,
    128: 27 + 217: the slice kernel moves an entry faster than a 16-row tile does --; code -2 - k for the k-th one),
    then the other rows by falling degree (stable), padded with -1;
    ``hubs`` int32 [n_hub, 3] = (global row id, first slice, number of slices) and ``parts`` int64 [n_slices, 2] =
    the slices' entry ranges (``part`` entries each, ``lpf_spmm_row_parts_f32``), both None without hub rows.
    ``pad_hubs``: pad the hub codes to a multiple of 16 so that no tile mixes hub and ordinary rows.
    The 16 rows of a tile walk their entry lists in lockstep: sorted, they are equally long."""
    dev = rowptr.device
    deg = rowptr[lo + 1:hi + 1] - rowptr[lo:hi]
    rows = torch.arange(lo, hi, dtype=torch.int64, device=dev)
    is_long = deg > long_threshold
    long_rows = rows[is_long]
    short = rows[~is_long]
    _, perm = torch.sort(deg[~is_long], descending=True, stable=True)
    hub_codes = -2 - torch.arange(long_rows.numel(), dtype=torch.int64, device=dev)
    if pad_hubs and hub_codes.numel() % 16:      # hub rows in tiles of their own (the bf16-table kernel wants that)
        hub_codes = torch.cat([hub_codes, torch.full(((-hub_codes.numel()) % 16,), -1, dtype=torch.int64, device=dev)])
    codes = torch.cat([hub_codes, short[perm]])
    pad = (-codes.numel()) % 16
    if pad:
        codes = torch.cat([codes, torch.full((pad,), -1, dtype=torch.int64, device=dev)])
    order = codes.to(torch.int32).contiguous()
    if long_rows.numel() == 0:
        return order, None, None
    ldeg = deg[is_long]
    n_sl = (ldeg + part - 1) // part
    first = torch.cumsum(n_sl, 0) - n_sl
    hubs = torch.stack([long_rows, first, n_sl], dim=1).to(torch.int32).contiguous()
    owner = torch.repeat_interleave(torch.arange(long_rows.numel(), device=dev), n_sl)
    within = torch.arange(int(n_sl.sum()), device=dev) - first[owner]
    e0 = rowptr[long_rows][owner] + within * part
    e1 = torch.minimum(e0 + part, rowptr[long_rows + 1][owner])
    return order, hubs, torch.stack([e0, e1], dim=1).to(torch.int64).contiguous()


def _from_scipy(m: sp.spmatrix, n: int, keep_val: bool) -> CSR:
    m = m.tocsr()
    m.sum_duplicates()
    m.sort_indices()
    return CSR(m.indptr.astype(np.int64), m.indices.astype(np.int32),
               m.data.astype(np.float32) if keep_val else None, n)


def csr_from_coo(row, col, val, n: int, *, keep_val=True) -> CSR:
    """Sorted CSR from COO triplets; duplicate entries are summed (torch ``coalesce`` semantics)."""
    row = np.asarray(row, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    data = np.ones(row.size, np.float32) if val is None else np.asarray(val, dtype=np.float32)
    return _from_scipy(sp.coo_matrix((data, (row, col)), shape=(n, n)), n, keep_val and val is not None)


def mask_csr(edge_index, n: int, *, symmetric=True) -> CSR:
    """0/1 adjacency pattern.  symmetric=True mirrors ``adj_t.to_symmetric()...coalesce().bool().int()``
    (src/util/read_datasets.py:88-95); False mirrors the un-symmetrised variants (:109-110, :234-235)."""
    ei = np.asarray(edge_index, dtype=np.int64)
    r, c = ei[0], ei[1]
    if symmetric:
        r, c = np.concatenate([r, c]), np.concatenate([c, r])
    return csr_from_coo(r, c, None, n)


def gcn_structure_csr(edge_index, edge_weight, n: int) -> CSR:
    """Structure the GCN normalisation works on: off-diagonal entries (duplicates summed) plus EVERY diagonal
    entry (``fill_diag`` replaces existing self-loops).  Values are the raw weights; the diagonal value is a
    placeholder that ``lpf_gcn_norm_csr`` overrides with 1."""
    ei = np.asarray(edge_index, dtype=np.int64)
    w = np.ones(ei.shape[1], np.float32) if edge_weight is None else np.asarray(edge_weight, np.float32).reshape(-1)
    off = ei[0] != ei[1]
    d = np.arange(n, dtype=np.int64)
    return csr_from_coo(np.concatenate([ei[0][off], d]), np.concatenate([ei[1][off], d]),
                        np.concatenate([w[off], np.ones(n, np.float32)]), n)


def prefilter_nonhop(ppr: CSR, thresh_non1hop: float) -> CSR:
    """Entries that can pass the >1-hop test: p > 0 and fl32(fl32(p+1)-1) >= f32(theta_n)
    (src/models/link_transformer.py:464-478).  A per-model index over the PPR matrix; selection results are
    identical with or without it."""
    v = ppr.val.astype(np.float32)
    keep = (v > 0) & (((v + np.float32(1)) - np.float32(1)) >= np.float32(thresh_non1hop))
    rows = np.repeat(np.arange(ppr.n, dtype=np.int64), np.diff(ppr.rowptr))[keep]
    rowptr = np.zeros(ppr.n + 1, np.int64)
    np.add.at(rowptr, rows + 1, 1)
    np.cumsum(rowptr, out=rowptr)
    return CSR(rowptr, ppr.col[keep].copy(), v[keep].copy(), ppr.n)


def prefilter_onehop(ppr: CSR, thresh_1hop: float) -> CSR:
    """The "P1" index: PPR entries that can pass the one-hop test, fl32(fl32(p+1)-1) >= f32(theta_1)
    (src/models/link_transformer.py:241-250 with the round trip of :290-317 for t = 1).  For theta_1 <= 0 this is the
    whole matrix.  Selection results are identical with or without it."""
    v = ppr.val.astype(np.float32)
    keep = ((v + np.float32(1)) - np.float32(1)) >= np.float32(thresh_1hop)
    rows = np.repeat(np.arange(ppr.n, dtype=np.int64), np.diff(ppr.rowptr))[keep]
    rowptr = np.zeros(ppr.n + 1, np.int64)
    np.add.at(rowptr, rows + 1, 1)
    np.cumsum(rowptr, out=rowptr)
    return CSR(rowptr, ppr.col[keep].copy(), v[keep].copy(), ppr.n)


def self_ppr(adj: CSR, ppr: CSR) -> np.ndarray:
    """selfp[e] = P[i, j] for every adjacency entry e = (i, j), 0 where the PPR matrix stores nothing: the PPR of a
    node to its own neighbours, aligned with the adjacency CSR."""
    n = adj.n
    a_rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(adj.rowptr))
    p_rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(ppr.rowptr))
    a_key = a_rows * n + adj.col.astype(np.int64)
    p_key = p_rows * n + ppr.col.astype(np.int64)  # sorted: rows ascending, columns ascending inside rows
    out = np.zeros(adj.nnz, np.float32)
    if p_key.size:
        idx = np.searchsorted(p_key, a_key)
        idx[idx >= p_key.size] = p_key.size - 1
        hit = p_key[idx] == a_key
        out[hit] = ppr.val[idx[hit]]
    return out


class RemovedEdges:
    """An adjacency override stated as a DIFFERENCE: the model's own typing adjacency (``adj_mask=``) -- or propagation
    matrix (``adj_prop=``, the ``--mask-input`` loop, src/train/train_model.py:47-56) -- minus these undirected edges.

    The reference's training loop rebuilds the masked adjacency of every batch from all training edges
    (src/train/train_model.py:38-46: ``adjmask[perm] = 0; SparseTensor.from_edge_index(train_pos[adjmask])...``).
    ``model(edges, adj_mask=masked_adj)`` accepts that tensor as it is -- the difference to the resident adjacency is then
    found on the device --; a loop that knows which edges it removed can say so directly,
    ``model(edges, adj_mask=lpformer_amd.RemovedEdges(edges))``, and skip building the tensor.
    ``edges``: [2, K] node ids (either direction; edges the adjacency does not hold are ignored)."""

    def __init__(self, edges):
        self.edges = edges


def as_coo_numpy(obj):
    """(row, col, val|None, n) from the graph objects the reference's data dict may hold:
    torch sparse COO/CSR tensors, scipy matrices, torch_sparse.SparseTensor-like objects (``.coo()``), or CSR."""
    if isinstance(obj, CSR):
        rows = np.repeat(np.arange(obj.n, dtype=np.int64), np.diff(obj.rowptr))
        return rows, obj.col.astype(np.int64), obj.val, obj.n
    if isinstance(obj, torch.Tensor):
        t = obj.detach().cpu()
        if t.layout == torch.sparse_csr:
            t = t.to_sparse_coo()
        if t.layout != torch.sparse_coo:
            raise TypeError("dense tensors are not accepted as graphs")
        t = t.coalesce()
        ix = t.indices().numpy()
        return ix[0], ix[1], t.values().to(torch.float32).numpy(), int(t.shape[0])
    if sp.issparse(obj):
        m = obj.tocoo()
        return m.row.astype(np.int64), m.col.astype(np.int64), m.data.astype(np.float32), int(m.shape[0])
    if hasattr(obj, "coo") and hasattr(obj, "sparse_sizes"):  # torch_sparse.SparseTensor duck type
        row, col, val = obj.coo()
        n = int(obj.sparse_sizes()[0])
        return (row.cpu().numpy(), col.cpu().numpy(),
                None if val is None else val.detach().cpu().to(torch.float32).numpy(), n)
    raise TypeError(f"unsupported graph container {type(obj)!r}")


def as_coo_device(obj, device):
    """(row, col, val|None, n) as tensors on ``device`` when ``obj`` already lives there (a torch sparse COO tensor or a
    torch_sparse.SparseTensor-like object with CUDA tensors), else None.  The training loop builds a fresh masked
    adjacency per batch on the GPU (src/train/train_model.py:40-51): this path keeps it there."""
    device = torch.device(device)
    if isinstance(obj, torch.Tensor) and obj.layout == torch.sparse_coo and obj.device == device:
        t = obj.detach().coalesce()
        ix = t.indices()
        return ix[0], ix[1], t.values().to(torch.float32), int(t.shape[0])
    if hasattr(obj, "coo") and hasattr(obj, "sparse_sizes"):
        row, col, val = obj.coo()
        if isinstance(row, torch.Tensor) and row.device == device:
            return row, col, None if val is None else val.detach().to(torch.float32), int(obj.sparse_sizes()[0])
    return None


def csr_from_coo_device(row: torch.Tensor, col: torch.Tensor, val, n: int, keep_val: bool = True) -> DeviceCSR:
    """Device twin of ``csr_from_coo``: sorted CSR, duplicate entries summed (coalesce semantics)."""
    key = row.long() * n + col.long()
    uniq, inv = torch.unique(key, return_inverse=True)           # sorted
    out_val = None
    if keep_val and val is not None:
        out_val = torch.zeros(uniq.numel(), dtype=torch.float32, device=key.device).index_add_(0, inv, val.float())
    r = torch.div(uniq, n, rounding_mode="floor")
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=key.device)
    rowptr[1:] = torch.cumsum(torch.bincount(r, minlength=n), 0)
    return DeviceCSR(rowptr, (uniq - r * n).to(torch.int32), out_val, n, None)


def gcn_structure_csr_device(row: torch.Tensor, col: torch.Tensor, val, n: int) -> DeviceCSR:
    """Device twin of ``gcn_structure_csr``: off-diagonal entries (duplicates summed) plus every diagonal entry."""
    w = torch.ones(row.numel(), dtype=torch.float32, device=row.device) if val is None else val.float().reshape(-1)
    off = row != col
    d = torch.arange(n, device=row.device, dtype=row.dtype)
    return csr_from_coo_device(torch.cat([row[off], d]), torch.cat([col[off], d]),
                               torch.cat([w[off], torch.ones(n, dtype=torch.float32, device=row.device)]), n)


def gcn_norm_device(struct: DeviceCSR, stream=None) -> DeviceCSR:
    """Run ``lpf_gcn_norm_csr`` on a structure from ``gcn_structure_csr``; returns a CSR sharing rowptr/col."""
    dev = struct.rowptr.device
    w_out = torch.empty(struct.nnz, dtype=torch.float32, device=dev)
    dis = torch.empty(struct.n, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
    _lib.check(_lib.hip().lpf_gcn_norm_csr(struct.n, _lib.ptr(struct.rowptr), _lib.ptr(struct.col),
                                            _lib.ptr(struct.val), _lib.ptr(w_out), _lib.ptr(dis), st),
               "lpf_gcn_norm_csr")
    out = DeviceCSR(struct.rowptr, struct.col, w_out, struct.n, None)
    out.__dict__["_struct_val"] = struct.val     # (the raw weights: an override that only drops edges re-normalises them)
    return out


def ppr_filter_device(ppr: DeviceCSR, mode: int, theta: float) -> DeviceCSR:
    """T0 (mode 0) / P1 (mode 1) index of a device-resident PPR matrix through ``lpf_ppr_filter_count`` / ``_fill``
    (device twin of ``prefilter_nonhop`` / ``prefilter_onehop``)."""
    from . import _lib
    lib, dev, n = _lib.hip(), ppr.rowptr.device, ppr.n
    st = torch.cuda.current_stream(dev).cuda_stream
    lens = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    _lib.check(lib.lpf_ppr_filter_count(n, _lib.ptr(ppr.rowptr), _lib.ptr(ppr.val), mode, float(theta),
                                        _lib.ptr(lens), st), "lpf_ppr_filter_count")
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens[:n], 0, out=rowptr[1:])
    nnz = int(rowptr[-1].item())
    col = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
    val = torch.empty(max(nnz, 1), dtype=torch.float32, device=dev)
    _lib.check(lib.lpf_ppr_filter_fill(n, _lib.ptr(ppr.rowptr), _lib.ptr(ppr.col), _lib.ptr(ppr.val), mode,
                                       float(theta), _lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(val), st),
               "lpf_ppr_filter_fill")
    return DeviceCSR(rowptr, col[:nnz], val[:nnz], n, None)


BLOCK = 16  # entries per block of a blocked index (one 64-byte line of columns)


@dataclass
class BlockedIndex:
    """A filtered PPR index (T0 / P1) laid out for two-step lookups (csrc/select2.hip, s2_find_blocked): every row is
    padded to a multiple of 16 entries (padding columns = INT32_MAX, values 0), so each 16-entry block is one aligned
    64-byte line, and ``skip[b]`` = last column of block b: a lookup reads the row's skip entries (one or two lines),
    then ONE block, instead of walking a binary search through memory.  ``len`` = real entries per row."""
    rowptr: torch.Tensor   # int64 [n+1], padded units (multiples of 16)
    col: torch.Tensor      # int32
    val: torch.Tensor      # float32
    cv: torch.Tensor       # int32 [padded entries, 2]: {column, value bits} interleaved (what the kernels read)
    len: torch.Tensor      # int32 [n]
    skip: torch.Tensor     # int32 [rowptr[n] / 16]
    n: int

    def to_host_compact(self) -> CSR:
        """The index as a plain CSR without padding (tests, statistics)."""
        ln = self.len.cpu().numpy().astype(np.int64)
        rp = self.rowptr.cpu().numpy()
        col, val = self.col.cpu().numpy(), self.val.cpu().numpy()
        keep = np.zeros(col.size, bool)
        starts = np.repeat(rp[:-1], ln)
        within = np.arange(int(ln.sum())) - np.repeat(np.concatenate([[0], np.cumsum(ln)[:-1]]), ln)
        keep[starts + within] = True
        rowptr = np.zeros(self.n + 1, np.int64)
        np.cumsum(ln, out=rowptr[1:])
        return CSR(rowptr, col[keep].copy(), val[keep].copy(), self.n)


def ppr_filter_device_blocked(ppr: DeviceCSR, mode: int, theta: float) -> BlockedIndex:
    """T0 (mode 0) / P1 (mode 1) index of a device-resident PPR matrix in the blocked layout (same kept entries, same
    order as ``ppr_filter_device``; ``lpf_ppr_filter_count`` / ``_fill`` with padded row starts)."""
    lib, dev, n = _lib.hip(), ppr.rowptr.device, ppr.n
    st = torch.cuda.current_stream(dev).cuda_stream
    lens = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    _lib.check(lib.lpf_ppr_filter_count(n, _lib.ptr(ppr.rowptr), _lib.ptr(ppr.val), mode, float(theta),
                                        _lib.ptr(lens), st), "lpf_ppr_filter_count")
    padded = (lens[:n] + (BLOCK - 1)) // BLOCK * BLOCK
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(padded, 0, out=rowptr[1:])
    total = int(rowptr[-1].item())
    col = torch.full((max(total, BLOCK),), 2**31 - 1, dtype=torch.int32, device=dev)
    val = torch.zeros(max(total, BLOCK), dtype=torch.float32, device=dev)
    _lib.check(lib.lpf_ppr_filter_fill(n, _lib.ptr(ppr.rowptr), _lib.ptr(ppr.col), _lib.ptr(ppr.val), mode,
                                       float(theta), _lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(val), st),
               "lpf_ppr_filter_fill")
    skip = col.view(-1, BLOCK)[:, BLOCK - 1].contiguous()
    # spare entries: a lookup reads 32 skip entries from the aligned group its window starts in
    skip = torch.cat([skip, torch.full((40,), 2**31 - 1, dtype=torch.int32, device=dev)])
    cv = torch.stack([col, val.view(torch.int32)], dim=1).contiguous()
    return BlockedIndex(rowptr, col, val, cv, lens[:n].to(torch.int32), skip, n)


HASH_MUL = 2654435761  # Fibonacci hashing constant (2^32 / golden ratio)
HASH_BUCKET = 8        # entries per bucket of a hashed index (one 64-byte half line)
BLOOM_MUL1, BLOOM_MUL2 = 0x85EBCA6B, 0xC2B2AE35   # constants of ``bloom_hash`` (csrc/select3.hip::s3_bloom_hash)
MINI_WORDS = 32          # 32-bit words of a node's MINI filter (1,024 bits = 128 bytes, one cache line)
MINI_SALT = 0x9E3779B9   # the mini filter hashes key ^ MINI_SALT: independent of the row filter's bits


def _pack_bits(n_words: int, bitpos: torch.Tensor, out: torch.Tensor, chunk: int = 1 << 22) -> None:
    """OR the bits ``bitpos`` (int64 positions, word * 32 + bit) into the int32 words ``out`` (flat view, n_words long)
    without atomics: mark them in chunks of words, pack 32 flags per word."""
    dev = bitpos.device
    w2 = torch.tensor([1 << b for b in range(31)] + [-(1 << 31)], dtype=torch.int32, device=dev)
    word_of = torch.div(bitpos, 32, rounding_mode="floor")
    for lo in range(0, n_words, chunk):
        hi = min(lo + chunk, n_words)
        m = (word_of >= lo) & (word_of < hi)
        if not bool(m.any().item()):
            continue
        flags = torch.zeros((hi - lo) * 32, dtype=torch.bool, device=dev)
        flags[bitpos[m] - lo * 32] = True
        packed = (flags.view(-1, 32).to(torch.int32) * w2[None, :]).sum(dim=1, dtype=torch.int32)
        nz = packed != 0
        out[lo:hi][nz] |= packed[nz]


def mini_filters(rowptr: torch.Tensor, col: torch.Tensor, n: int) -> torch.Tensor:
    """int32 [n, MINI_WORDS]: a fixed 1,024-bit absence filter per row of a CSR (rows = nodes, ``col`` = the keys of the
    row's hashed union): key c sets two bits of ONE word -- h = bloom_hash(c ^ MINI_SALT), word h >> 27, bits h & 31
    and (h >> 5) & 31.  csrc/select3.hip stages the filters of an item's endpoints in LDS (one 128-byte line each, read
    coalesced) and tests every candidate there: a candidate that fails is not in the row and costs no random read at
    all.  Rows of a few hundred keys pass 10-20 % of the absent candidates, hub rows almost everything (they fall
    through to the row's own filter in front of its buckets)."""
    dev = rowptr.device
    nnz = int(rowptr[-1].item())
    out = torch.zeros(n * MINI_WORDS, dtype=torch.int32, device=dev)
    if nnz:
        row = torch.repeat_interleave(torch.arange(n, device=dev), rowptr[1:] - rowptr[:-1])
        h = bloom_hash(col[:nnz].long() ^ MINI_SALT)
        word = row * MINI_WORDS + (h >> 27)
        _pack_bits(n * MINI_WORDS, torch.cat([word * 32 + (h & 31), word * 32 + ((h >> 5) & 31)]), out)
    return out.view(n, MINI_WORDS)


def bloom_hash(key: torch.Tensor) -> torch.Tensor:
    """The filter's 32-bit mix of a node id (int64 tensor in, int64 values < 2^32 out); select3.hip::s3_bloom_hash."""
    h = (key * BLOOM_MUL1) & 0xFFFFFFFF
    h = h ^ (h >> 15)
    h = (h * BLOOM_MUL2) & 0xFFFFFFFF
    return h ^ (h >> 13)


@dataclass
class HashedIndex:
    """A filtered PPR index (P1) laid out for ONE-step lookups (csrc/select2.hip, s2_lookup_hashed): row i owns
    ``len[i]`` buckets of HASH_BUCKET = 8 {column, value bits} entries (64 aligned bytes each, empty entries carry
    column INT32_MAX), entry (i, c) lives in bucket ``((c * HASH_MUL mod 2^32) * len[i]) >> 32`` of the row.  Bucket
    counts start at one per four entries and grow for the rows where some bucket would hold more than 8, so a lookup
    never has to look further than its bucket.  ``rowptr`` counts ENTRIES (8 per bucket)."""
    rowptr: torch.Tensor   # int64 [n+1]
    cv: torch.Tensor       # int32 [HASH_BUCKET * buckets, 2] = 8 entries per bucket
    len: torch.Tensor      # int32 [n]: buckets per row
    n: int

    def to_host_compact(self) -> CSR:
        """The index as a plain sorted CSR (tests, statistics)."""
        cv = self.cv.cpu().numpy()
        rp = self.rowptr.cpu().numpy()
        row = np.repeat(np.arange(self.n), np.diff(rp))
        live = cv[: row.size, 0] != 2**31 - 1
        row, col, val = row[live], cv[: row.size][live, 0], cv[: row.size][live, 1].copy().view(np.float32)
        order = np.lexsort((col, row))
        rowptr = np.zeros(self.n + 1, np.int64)
        np.cumsum(np.bincount(row, minlength=self.n), out=rowptr[1:])
        return CSR(rowptr, col[order].copy(), val[order].copy(), self.n)


def hash_index_device(p: DeviceCSR) -> HashedIndex:
    """Bucketised layout of a device-resident sorted CSR (see HashedIndex).  One-time, a few torch passes."""
    dev, n = p.rowptr.device, p.n
    ln = p.rowptr[1:] - p.rowptr[:-1]
    nnz = int(p.rowptr[-1].item())
    col = p.col[:nnz].long()
    row = torch.repeat_interleave(torch.arange(n, device=dev), ln)
    h = (col * HASH_MUL) & 0xFFFFFFFF
    nbk = (ln + HASH_BUCKET // 2 - 1) // (HASH_BUCKET // 2)
    for _ in range(64):
        base = torch.cumsum(nbk, 0) - nbk
        total = int(nbk.sum().item())
        key = base[row] + ((h * nbk[row]) >> 32)
        cnt = torch.bincount(key, minlength=max(total, 1))
        over = cnt > HASH_BUCKET
        if nnz == 0 or not bool(over.any().item()):
            break
        bucket_row = torch.repeat_interleave(torch.arange(n, device=dev), nbk)
        bad = torch.unique(bucket_row[over[:total]])
        nbk[bad] = nbk[bad] * 3 // 2 + 1
    else:
        raise RuntimeError("hash_index_device: bucket sizes did not settle")
    cv = torch.zeros((max(total, 1) * HASH_BUCKET, 2), dtype=torch.int32, device=dev)
    cv[:, 0] = 2**31 - 1
    if nnz:
        order = torch.argsort(key, stable=True)
        key_s = key[order]
        start = torch.cumsum(cnt, 0) - cnt
        pos = key_s * HASH_BUCKET + (torch.arange(nnz, device=dev) - start[key_s])
        cv[pos, 0] = p.col[:nnz][order]
        cv[pos, 1] = p.val[:nnz].view(torch.int32)[order]
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(nbk * HASH_BUCKET, 0, out=rowptr[1:])
    return HashedIndex(rowptr, cv.contiguous(), nbk.to(torch.int32), n)


def self_ppr_device(adj: DeviceCSR, ppr: DeviceCSR) -> torch.Tensor:
    """``self_ppr`` on the device (``lpf_self_ppr``)."""
    from . import _lib
    dev = adj.rowptr.device
    out = torch.empty(max(adj.nnz, 1), dtype=torch.float32, device=dev)
    _lib.check(_lib.hip().lpf_self_ppr(adj.n, _lib.ptr(adj.rowptr), _lib.ptr(adj.col), _lib.ptr(ppr.rowptr),
                                       _lib.ptr(ppr.col), _lib.ptr(ppr.val), _lib.ptr(out),
                                       torch.cuda.current_stream(dev).cuda_stream), "lpf_self_ppr")
    return out[:adj.nnz]


# ------------------------------------------------------------------------------------------------ walk indexes
@dataclass
class WalkIndex:
    """Per-model indexes of the walk-plan selection (csrc/select3.hip, include/lpformer_hip.h ``lpf_select3_*``), built
    once per (adjacency, PPR matrix, thresholds).  All "cv" arrays are int32 [entries, 2] = {node, fp32 value bits}.

    * ``adj_cv``  adjacency rows with the PPR of the row's node to that neighbour (0 where nothing is stored);
    * ``a1_cv``   the adjacency entries whose own value passes the one-hop test ``fl32(fl32(p+1)-1) >= theta_1``;
    * ``px_cv``   PPR entries (i, v) with v NOT adjacent to i that pass the weaker of the one-hop / >1-hop tests;
    * ``t0_cv``   the px entries that pass the >1-hop test (``p > 0`` and round trip ``>= theta_n``), None without them;
    * ``u``       hashed union of adjacency row and px row (``HashedIndex``), the sign bit of a value = "adjacent";
    * ``rec``     int32 [n, 16]: one 64-byte record per node with the five row starts (int64) and the five lengths;
    * ``mini``    int32 [n, 32]: a fixed 1,024-bit absence filter of every union row (``mini_filters``).
    """
    rec: torch.Tensor
    adj_cv: torch.Tensor
    a1_cv: torch.Tensor
    px_cv: torch.Tensor
    t0_cv: Optional[torch.Tensor]
    u: "HashedIndex"
    n: int
    use_px: bool
    mini: Optional[torch.Tensor] = None   # int32 [n, MINI_WORDS]: ``mini_filters`` of the union rows

    def lengths(self):
        """(deg, n_a1, n_px, n_t0, u_buckets) as int32 [n] views of ``rec`` (statistics, tests)."""
        return tuple(self.rec[:, 10 + i] for i in range(5))


def _csr_rows(rowptr: torch.Tensor, n: int) -> torch.Tensor:
    return torch.repeat_interleave(torch.arange(n, device=rowptr.device), rowptr[1:] - rowptr[:-1])


def _round_trip1(v: torch.Tensor) -> torch.Tensor:
    """fl32(fl32(v + 1) - 1): the reference's round trip for t = 1 (src/models/link_transformer.py:290-291,316-317 and
    :464-476), two separately rounded fp32 operations."""
    one = torch.ones((), dtype=torch.float32, device=v.device)
    return (v + one) - one


def _cv(col: torch.Tensor, bits: torch.Tensor) -> torch.Tensor:
    out = torch.stack([col.to(torch.int32), bits.view(torch.int32) if bits.dtype == torch.float32 else bits], dim=1)
    if out.shape[0] == 0:  # (the kernels want non-null pointers)
        out = torch.zeros((1, 2), dtype=torch.int32, device=col.device)
    return out.contiguous()


def build_walk_index(adj, ppr, th_1hop: float, th_non1hop: float, want_t0: bool) -> WalkIndex:
    """Builds the ``WalkIndex`` of an adjacency (``rowptr``, ``col``, ``n``; sorted, duplicate-free) and a PPR matrix
    (``rowptr``, ``col``, ``val``) that live on one device -- a few sorts / searches in torch, one-time.  Works on CPU
    tensors too (tests)."""
    dev, n = adj.rowptr.device, adj.n
    f32 = torch.float32
    deg = adj.rowptr[1:] - adj.rowptr[:-1]
    nnz_a, nnz_p = int(adj.rowptr[-1].item()), int(ppr.rowptr[-1].item())
    akey = _csr_rows(adj.rowptr, n) * n + adj.col[:nnz_a].long()          # ascending: rows, columns inside rows
    prow = _csr_rows(ppr.rowptr, n)
    pkey = prow * n + ppr.col[:nnz_p].long()
    pval = ppr.val[:nnz_p].to(f32)

    def member(sorted_keys, query):
        """(hit mask, position) of ``query`` in ``sorted_keys``."""
        if sorted_keys.numel() == 0 or query.numel() == 0:
            z = torch.zeros(query.numel(), dtype=torch.long, device=dev)
            return torch.zeros(query.numel(), dtype=torch.bool, device=dev), z
        pos = torch.searchsorted(sorted_keys, query).clamp_(max=sorted_keys.numel() - 1)
        return sorted_keys[pos] == query, pos

    # self PPR of every adjacency entry
    selfp = torch.zeros(nnz_a, dtype=f32, device=dev)
    hit, pos = member(pkey, akey)
    selfp[hit] = pval[pos[hit]]
    t1 = torch.tensor(float(th_1hop), dtype=f32, device=dev)
    tn = torch.tensor(float(th_non1hop), dtype=f32, device=dev)
    strong = _round_trip1(selfp) >= t1
    # px: PPR entries passing the weaker test, not adjacent
    rt = _round_trip1(pval)
    keep = rt >= (torch.minimum(t1, tn) if want_t0 else t1)
    pk, pv = pkey[keep], pval[keep]
    is_adj, _ = member(akey, pk)
    px_key, px_val = pk[~is_adj], pv[~is_adj]
    px_row = torch.div(px_key, n, rounding_mode="floor")
    px_col = px_key - px_row * n
    far = (px_val > 0) & (_round_trip1(px_val) >= tn) if want_t0 else None

    def starts(lens):
        return torch.cumsum(lens, 0) - lens

    arow = _csr_rows(adj.rowptr, n)
    n_a1 = torch.bincount(arow[strong], minlength=n)
    n_px = torch.bincount(px_row, minlength=n)
    n_t0 = torch.bincount(px_row[far], minlength=n) if want_t0 else torch.zeros(n, dtype=torch.long, device=dev)
    # hashed union: adjacency entries (sign bit set) + px entries, merged in key order
    sign = torch.tensor(-2**31, dtype=torch.int32, device=dev)
    ukey = torch.cat([akey, px_key])
    ubits = torch.cat([selfp.view(torch.int32) | sign, px_val.view(torch.int32)])
    order = torch.argsort(ukey)
    ukey, ubits = ukey[order], ubits[order]
    urow = torch.div(ukey, n, rounding_mode="floor")
    u_rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.bincount(urow, minlength=n), 0, out=u_rowptr[1:])
    ucol = (ukey - urow * n).to(torch.int32)
    u = hash_index_device(DeviceCSR(u_rowptr, ucol, ubits.view(f32), n, None))
    mini = mini_filters(u_rowptr, ucol, n)

    rec64 = torch.zeros((n, 8), dtype=torch.int64, device=dev)
    rec64[:, 0] = adj.rowptr[:-1]
    rec64[:, 1] = starts(n_a1)
    rec64[:, 2] = starts(n_px)
    rec64[:, 3] = starts(n_t0)
    rec64[:, 4] = u.rowptr[:-1]
    rec = rec64.view(torch.int32)                      # [n, 16]; little endian: int64 field f = columns 2f, 2f+1
    for i, lens in enumerate((deg, n_a1, n_px, n_t0, u.len)):
        rec[:, 10 + i] = lens.to(torch.int32)
    return WalkIndex(rec=rec.contiguous(), adj_cv=_cv(adj.col[:nnz_a], selfp),
                     a1_cv=_cv(adj.col[:nnz_a][strong], selfp[strong]), px_cv=_cv(px_col, px_val),
                     t0_cv=_cv(px_col[far], px_val[far]) if want_t0 else None, u=u, n=n,
                     use_px=float(th_1hop) > 0.0, mini=mini.contiguous())
