"""PPR producer: host C++/OpenMP Andersen push behind ``lpf_ppr_push_cpu`` (liblpformer_host.so).

Replaces the reference's numba ``calc_ppr`` + Python list packing (src/util/calc_ppr_scores.py:103-241) with the
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


def get_ppr(edge_index, num_nodes: int, alpha: float = 0.15, eps: float = 5e-5) -> torch.Tensor:
    """torch sparse COO PPR matrix, the object the reference stores in ``data['ppr']``
    (calc_ppr_scores.py:245-270, minus the on-disk cache)."""
    return calc_ppr(edge_index, num_nodes, alpha, eps).to_torch_sparse_coo()
