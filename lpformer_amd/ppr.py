"""PPR producer: Andersen push behind ``lpf_ppr_push_cpu`` (host C++/OpenMP, liblpformer_host.so) and
``lpf_ppr_push_f64`` + ``lpf_ppr_pack_csr`` (MI355X, liblpformer_hip.so).

Both replace the reference's numba ``calc_ppr`` + Python list packing (src/util/calc_ppr_scores.py:103-241) with the
same push order and float64 arithmetic, so the resulting sparse matrix -- index sets and fp32 values -- is
bit-identical; it is what makes the PPR neighbour sets of the scoring path reproducible.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .graph import CSR, csr_from_coo


def calc_ppr(edge_index, num_nodes: int, alpha: float = 0.15, eps: float = 5e-5, num_threads: int = 0) -> CSR:
    """PPR matrix as a host CSR (rows sorted by column, fp32 values).

    edge_index: [2, E] directed edge list (both directions present for an undirected graph); it is coalesced
    (sorted, duplicates dropped) first, like ``get_ppr_matrix`` (calc_ppr_scores.py:111-117)."""
    ei = edge_index.detach().cpu().numpy() if isinstance(edge_index, torch.Tensor) else np.asarray(edge_index)
    g = csr_from_coo(ei[0], ei[1], None, num_nodes)
    indptr = np.ascontiguousarray(g.rowptr, dtype=np.int64)
    indices = np.ascontiguousarray(g.col, dtype=np.int32)
    rowptr = np.zeros(num_nodes + 1, dtype=np.int64)
    col_p, val_p = C.c_void_p(), C.c_void_p()
    lib = _lib.host()
    rc = lib.lpf_ppr_push_cpu(num_nodes, indptr.ctypes.data, indices.ctypes.data, float(alpha), float(eps),
                              rowptr.ctypes.data, C.byref(col_p), C.byref(val_p), int(num_threads))
    if rc != 0:
        raise _lib.LpfError(f"lpf_ppr_push_cpu failed with code {rc}")
    nnz = int(rowptr[-1])
    try:
        col = np.ctypeslib.as_array(C.cast(col_p, C.POINTER(C.c_int32)), shape=(max(nnz, 1),))[:nnz].copy()
        val = np.ctypeslib.as_array(C.cast(val_p, C.POINTER(C.c_float)), shape=(max(nnz, 1),))[:nnz].copy()
    finally:
        lib.lpf_host_free(col_p)
        lib.lpf_host_free(val_p)
    return CSR(rowptr, col, val, num_nodes)


def calc_ppr_gpu(edge_index, num_nodes: int, alpha: float = 0.15, eps: float = 5e-5, device="cuda", n_waves: int = 0,
                 state_budget_bytes: int = 8 << 30, to_host: bool = True, pool_capacity: int = 0,
                 timings: dict = None):
    """Same result as ``calc_ppr`` computed on the GPU: one wavefront per source, dense per-wavefront push state.

    ``n_waves`` (0 = as many as ``state_budget_bytes`` of HBM allow, at most 8192) bounds the concurrent sources
    (the push itself keeps getting faster up to 8192 wavefronts, but a first-time allocation of the state costs
    about 30 ms per GB, so the default stops at 8 GB);
    ``pool_capacity`` (0 = 384 entries per node) is the first guess of the result size (a too small guess costs one
    exact-size rerun).  ``timings`` (optional dict) receives the seconds spent per phase (synchronising).
    Returns a host ``CSR`` (``to_host=True``) or the device tensors ``(rowptr int64, col int32, val fp32)``."""
    import time
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.LpfError("calc_ppr_gpu needs an MI355X device; use calc_ppr for the host producer")

    def mark(name, t0):
        if timings is not None:
            torch.cuda.synchronize(dev)
            timings[name] = timings.get(name, 0.0) + time.perf_counter() - t0
        return time.perf_counter()

    t = time.perf_counter()
    ei = edge_index.detach().cpu().numpy() if isinstance(edge_index, torch.Tensor) else np.asarray(edge_index)
    g = csr_from_coo(ei[0], ei[1], None, num_nodes)
    t = mark("host_csr_s", t)
    n = int(num_nodes)
    lib = _lib.hip()
    st = torch.cuda.current_stream(dev).cuda_stream
    rowptr = torch.from_numpy(np.ascontiguousarray(g.rowptr, dtype=np.int64)).to(dev)
    col = torch.from_numpy(np.ascontiguousarray(g.col, dtype=np.int32)).to(dev)
    if n == 0:
        z = torch.zeros(1, dtype=torch.int64, device=dev)
        return CSR(np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float32), 0) if to_host else \
            (z, col[:0], torch.zeros(0, device=dev))
    if n_waves <= 0:
        per_wave = max(1, lib.lpf_ppr_push_workspace_bytes(n, 4, float(alpha), float(eps)) // 4)
        n_waves = int(min(8192, max(4, state_budget_bytes // per_wave), 4 * ((n + 3) // 4)))
    n_waves = max(4, n_waves - n_waves % 4)
    ws_bytes = lib.lpf_ppr_push_workspace_bytes(n, n_waves, float(alpha), float(eps))
    if ws_bytes <= 0:
        raise _lib.LpfError("lpf_ppr_push_workspace_bytes: invalid arguments")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    row_off = torch.empty(n, dtype=torch.int64, device=dev)
    row_len = torch.empty(n, dtype=torch.int32, device=dev)
    counters = torch.zeros(4, dtype=torch.int64, device=dev)
    t = mark("alloc_upload_s", t)
    cap = int(pool_capacity) if pool_capacity > 0 else max(1 << 16, 384 * n)
    while True:
        pool_col = torch.empty(cap, dtype=torch.int32, device=dev)
        pool_val = torch.empty(cap, dtype=torch.float32, device=dev)
        _lib.check(lib.lpf_ppr_push_f64(n, _lib.ptr(rowptr), _lib.ptr(col), float(alpha), float(eps), n_waves,
                                        _lib.ptr(ws), ws_bytes, _lib.ptr(pool_col), _lib.ptr(pool_val), cap,
                                        _lib.ptr(row_off), _lib.ptr(row_len), _lib.ptr(counters), st),
                   "lpf_ppr_push_f64")
        _, nnz, bad, _ = (int(v) for v in counters.tolist())
        if bad:
            raise _lib.LpfError(f"lpf_ppr_push_f64: {bad} rows exceeded the 1/(alpha*eps) list bound")
        if nnz <= cap:
            break
        cap = nnz  # deterministic: the second run needs exactly this many slots
    t = mark("push_s", t)
    del ws
    pk_bytes = lib.lpf_ppr_pack_workspace_bytes(n, nnz)
    pk = torch.empty(max(pk_bytes, 256), dtype=torch.uint8, device=dev)
    out_rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    out_col = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
    out_val = torch.empty(max(nnz, 1), dtype=torch.float32, device=dev)
    _lib.check(lib.lpf_ppr_pack_csr(n, _lib.ptr(row_off), _lib.ptr(row_len), _lib.ptr(pool_col), _lib.ptr(pool_val),
                                    nnz, _lib.ptr(out_rowptr), _lib.ptr(out_col), _lib.ptr(out_val), _lib.ptr(pk),
                                    pk.numel(), st), "lpf_ppr_pack_csr")
    out_col, out_val = out_col[:nnz], out_val[:nnz]
    t = mark("sort_pack_s", t)
    if not to_host:
        return out_rowptr, out_col, out_val
    return CSR(out_rowptr.cpu().numpy(), out_col.cpu().numpy(), out_val.cpu().numpy(), n)


def ppr_cache_path(root: str, dataset: str, alpha: float, eps: float, is_val: bool = False) -> str:
    """The reference's cache location and name (calc_ppr_scores.py:249-257) with this repo's container suffix:
    ``<root>/node_subsets/ppr/<dataset>/sparse_adj-015_eps-5e-05[_val].lpf.npz``.  The reference pickles a
    ``torch_sparse.SparseTensor`` there, which only that package can read back; the CSR triplet below carries the same
    matrix (rows sorted by column, fp32 values) without the dependency."""
    import os
    val_suf = "_val" if is_val else ""
    name = f"sparse_adj-{str(alpha).replace('.', '')}_eps-{str(eps).replace('.', '')}{val_suf}.lpf.npz"
    return os.path.join(root, "node_subsets", "ppr", dataset, name)


def _edge_fingerprint(edge_index, num_nodes: int) -> np.ndarray:
    """(edge count, order-independent 64-bit checksum of the (row, col) pairs): cheap identity of an edge set."""
    ei = np.asarray(edge_index.cpu() if hasattr(edge_index, "cpu") else edge_index, dtype=np.int64).reshape(2, -1)
    key = ei[0].astype(np.uint64) * np.uint64(num_nodes) + ei[1].astype(np.uint64)
    mixed = (key * np.uint64(0x9E3779B97F4A7C15)) ^ (key >> np.uint64(29))
    return np.array([ei.shape[1], int(np.bitwise_xor.reduce(mixed)) if key.size else 0, int(mixed.sum(dtype=np.uint64))],
                    dtype=np.uint64)


def load_or_calc_ppr(edge_index, num_nodes: int, alpha: float = 0.15, eps: float = 5e-5, *, cache_root=None,
                     dataset: str = "graph", is_val: bool = False, device=None, num_threads: int = 0) -> CSR:
    """``get_ppr`` of the reference (calc_ppr_scores.py:245-270): load the cached matrix if present, otherwise run the
    producer (the GPU one when ``device`` is given, else the host one) and store the result."""
    import os
    path = None if cache_root is None else ppr_cache_path(cache_root, dataset, alpha, eps, is_val)
    fp = _edge_fingerprint(edge_index, num_nodes)
    if path is not None and os.path.isfile(path):
        z = np.load(path)
        if int(z["num_nodes"]) != int(num_nodes):
            raise _lib.LpfError(f"{path}: cached PPR is for {int(z['num_nodes'])} nodes, not {num_nodes}")
        # a cache is matched by NAME (dataset, alpha, eps) like the reference's; the edge fingerprint catches a file
        # that was computed on another edge set under the same name (files written before it existed are trusted)
        if "edge_fp" in z.files and not np.array_equal(z["edge_fp"], fp):
            raise _lib.LpfError(f"{path}: cached PPR was computed on a different edge set (fingerprint mismatch); "
                                "delete the file or use another dataset name")
        return CSR(z["rowptr"], z["col"], z["val"], int(num_nodes))
    csr = calc_ppr_gpu(edge_index, num_nodes, alpha, eps, device=device) if device is not None else \
        calc_ppr(edge_index, num_nodes, alpha, eps, num_threads)
    if path is not None:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = path + ".tmp.npz"
        np.savez(tmp, rowptr=csr.rowptr, col=csr.col, val=csr.val, num_nodes=np.int64(num_nodes),
                 alpha=np.float64(alpha), eps=np.float64(eps), edge_fp=fp)
        os.replace(tmp, path)
    return csr


def ppr_reference_cache_path(root_dir: str, dataset: str, alpha: float, eps: float, is_val: bool = False) -> str:
    """Exactly the reference's cache file (calc_ppr_scores.py:249-257):
    ``<root_dir>/node_subsets/ppr/<dataset>/sparse_adj-015_eps-5e-05[_val].pt``."""
    import os
    val_suf = "_val" if is_val else ""
    name = f"sparse_adj-{str(alpha).replace('.', '')}_eps-{str(eps).replace('.', '')}{val_suf}.pt"
    return os.path.join(root_dir, "node_subsets", "ppr", dataset, name)


def get_ppr(dataset, edge_index, num_nodes, alpha, eps, is_val, *, root_dir=None, device=None,
            num_threads: int = 0) -> torch.Tensor:
    """Drop-in for the reference's ``get_ppr(dataset, edge_index, num_nodes, alpha, eps, is_val)``
    (src/util/calc_ppr_scores.py:245-270): same six positional arguments, same cache directory and file NAME, same
    return type (a coalesced torch sparse COO tensor, what ``read_datasets.py:122-129`` stores in ``data['ppr']``).

    Cache: the reference ``torch.save``s a ``torch_sparse.SparseTensor`` at that path.  When ``torch_sparse`` is
    importable such a file is read (a cache written by the reference is picked up) and written in that format; without
    the package a ``.lpf.npz`` sibling with the same stem carries the CSR triplet (``load_or_calc_ppr``).
    ``root_dir``: the directory that holds ``node_subsets/`` (the reference uses its repo root); default: the current
    working directory.  ``device``: run the MI355X producer instead of the host one (bit-identical result)."""
    import os
    root_dir = os.getcwd() if root_dir is None else root_dir
    ref_path = ppr_reference_cache_path(root_dir, dataset, alpha, eps, is_val)
    try:
        from torch_sparse import SparseTensor  # noqa: F401  (only to read / write the reference's own cache format)
        have_ts = True
    except Exception:  # noqa: BLE001
        have_ts = False
    if have_ts and os.path.isfile(ref_path):
        print("PPR matrix exists. Loading from file...", flush=True)
        # the reference pickles a torch_sparse.SparseTensor: allow-list exactly that class (and its storage) instead of
        # unpickling whatever the file holds
        import pickle
        from torch_sparse import SparseTensor
        from torch_sparse.storage import SparseStorage
        try:
            with torch.serialization.safe_globals([SparseTensor, SparseStorage]):
                return torch.load(ref_path, weights_only=True).to_torch_sparse_coo_tensor()
        except (AttributeError, pickle.UnpicklingError, RuntimeError) as exc:
            # (torch < 2.5 has no safe_globals; the restricted unpickler may refuse the TorchScript storage class): the
            # cache is an optimisation -- recompute instead of failing, and say so
            print(f"could not load {ref_path} with the restricted unpickler ({exc}); recomputing", flush=True)
    csr = load_or_calc_ppr(edge_index, int(num_nodes), alpha, eps, cache_root=root_dir, dataset=dataset, is_val=is_val,
                           device=device, num_threads=num_threads)
    coo = csr.to_torch_sparse_coo()
    if have_ts:
        from torch_sparse import SparseTensor
        ix = coo.indices()
        print(f"Saving data to {ref_path}...", flush=True)
        os.makedirs(os.path.dirname(ref_path), exist_ok=True)   # (only when a file is actually written)
        torch.save(SparseTensor(row=ix[0], col=ix[1], value=coo.values(), sparse_sizes=(int(num_nodes),) * 2), ref_path)
    return coo


def ppr_coo(edge_index, num_nodes: int, alpha: float = 0.15, eps: float = 5e-5) -> torch.Tensor:
    """torch sparse COO PPR matrix without any cache (``calc_ppr(...).to_torch_sparse_coo()``)."""
    return calc_ppr(edge_index, num_nodes, alpha, eps).to_torch_sparse_coo()
