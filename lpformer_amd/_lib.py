"""ctypes binding of the two C-ABI shared objects (include/lpformer_hip.h).

There is no fallback: if ``liblpformer_hip.so`` is missing or a call fails, the caller gets an exception.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_HERE, "liblpformer_hip.so")
HOST_LIB_PATH = os.path.join(_HERE, "liblpformer_host.so")

ABI_VERSION = 9
FLAG_RELU = 1
SELECT_ERR_NODE_RANGE, SELECT_ERR_ITEM_CAP, SELECT_ERR_ENTRY_CAP = 1, 2, 4
ROWS_PERM_LB_WORDS = 1025      # LPF_ROWS_PERM_LB_WORDS (include/lpformer_hip.h)
SELECT4_BLOCK = 64             # LPF_SELECT4_BLOCK

i32, i64, f32, f64, u32, u64, vp = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_uint32, C.c_uint64, C.c_void_p

# name -> argument types (every entry point returns int unless listed in _RESTYPE)
HIP_PROTOTYPES = {
    "lpf_abi_version": [],
    "lpf_strerror": [C.c_int],
    "lpf_last_hip_error": [],
    "lpf_device_info": [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int],
    "lpf_gcn_norm_csr": [i64, vp, vp, vp, vp, vp, vp],
    "lpf_spmm_csr_f32": [i64, i32, vp, vp, vp, vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, vp, u32, vp, i64, vp],
    "lpf_gemm_f32": [i64, i32, i32, vp, i64, vp, i64, vp, vp, i64, vp, i64, u32, vp],
    "lpf_layernorm_bwd_f32": [i64, i32, vp, i64, vp, i64, vp, vp, i64, vp, vp, vp, vp],
    "lpf_layernorm_bwd_workspace_floats": [i32],
    "lpf_layernorm_relu_bwd_f32": [i64, i32, vp, i64, vp, i64, vp, vp, vp, i64, vp, vp, vp, vp, vp],
    "lpf_layernorm_relu_drop_bwd_f32": [i64, i32, vp, i64, vp, i64, vp, vp, f32, u64, vp, i64, vp, vp, vp, vp, vp],
    "lpf_gemm_tn_f32": [i64, i32, i32, vp, i64, vp, i64, vp, i64, vp, vp],
    "lpf_gemm_tn_colsum_f32": [i64, i32, i32, vp, i64, vp, i64, vp, i64, vp, vp, vp],
    "lpf_gemm_tn_workspace_floats": [i64, i32, i32],
    "lpf_gemm_f32_out_bf16": [i64, i32, i32, vp, i64, vp, i64, vp, vp, i64, vp, i64, u32, vp],
    "lpf_gcn_layer_fused_f32": [i32, i64, vp, i64, vp, vp, vp, vp, i64, vp, vp, i64, vp, vp, vp, vp, i64, vp, vp, u32, vp,
                                vp, vp, i64, vp],
    "lpf_spmm_row_parts_f32": [i32, vp, i64, vp, vp, vp, i64, vp, vp],
    "lpf_spmm_row_parts_bf16p": [i32, vp, i64, vp, vp, vp, i64, vp, vp],
    "lpf_gcn_layer_fused_train_f32": [i32, i64, vp, i64, vp, vp, vp, vp, i64, vp, vp, i64, vp, vp, vp, vp, i64, u32, vp, vp,
                                      vp, i64, vp, i64, f32, u64, vp],
    "lpf_gcn_layer_fused_bf16": [i32, i64, vp, i64, vp, vp, vp, vp, i64, vp, vp, i64, vp, vp, vp, vp, i64, vp, vp, u32, vp,
                                 vp, vp, i64, vp],
    "lpf_spmm_csr_bf16": [i64, i32, vp, vp, vp, vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, vp, u32, vp, i64, vp],
    "lpf_layernorm_f32": [i64, i32, vp, i64, vp, vp, vp, i64, u32, vp],
    "lpf_pair_gather_f32": [i64, i32, vp, i64, i64, vp, i64, vp, i64, vp, i64, vp],
    "lpf_select_plan_blocks": [i64],
    "lpf_select_plan": [i64, vp, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp],
    "lpf_select_run": [i64, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, f32, i32, vp, vp, i64,
                       i32, vp],
    "lpf_select3_plan": [i64, vp, i64, i64, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, i64, vp, vp, vp],
    "lpf_select3_run": [i64, vp, vp, vp, i64, vp, vp, vp, vp, f32, f32, f32, i32, vp, vp, i64, i32, vp],
    "lpf_select4": [i64, vp, i64, i64, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, f32, vp, vp, vp, vp, vp, i64, i32, vp],
    "lpf_select4_regions": [i64, vp, vp, vp, i64, vp, vp, i64, vp, vp],
    "lpf_select_export": [i64, vp, vp, i64, vp, vp, i64, i32, vp, vp, vp, vp, vp],
    "lpf_pair_scores_f32": [i32, vp, i64, vp, vp, vp, vp, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, i64, vp],
    "lpf_pair_softmax_gather_f32": [i32, i64, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, i64, vp, vp, vp],
    "lpf_pair_attention_fused_f32": [i32, i64, vp, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, vp, i64, vp],
    "lpf_pair_attention_fused_bf16": [i32, i64, vp, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, vp, i64, vp],
    "lpf_pair_attention_flip_f32": [i32, i64, vp, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, vp, i64, vp],
    "lpf_pair_attention_flip_zbf16": [i32, i64, vp, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, vp, i64, vp],
    "lpf_pair_attention_merge_f32": [i64, i32, i32, vp, vp, i64, vp, vp, vp, vp, vp, vp, i64, vp],
    "lpf_pair_attention_rows_f32": [i32, i64, vp, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp,
                                    i64, vp, i64, vp],
    "lpf_pair_attention_rows_zbf16": [i32, i64, vp, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp,
                                      i64, vp, i64, vp],
    "lpf_pair_attention_rows_perm_f32": [i32, i64, vp, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp,
                                         vp, i64, vp, i64, vp, vp, vp, vp],
    "lpf_pair_attention_rows_perm_zbf16": [i32, i64, vp, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, i32,
                                           vp, vp, i64, vp, i64, vp, vp, vp, vp],
    "lpf_pair_attention_rows4_f32": [i32, i64, vp, vp, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp,
                                     vp, vp, vp, vp, i32, vp, vp, i64, vp, i64, vp, vp, vp],
    "lpf_pair_attention_rows4_zbf16": [i32, i64, vp, vp, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp,
                                       vp, vp, vp, vp, i32, vp, vp, i64, vp, i64, vp, vp, vp],
    "lpf_tail_chain_rows_perm_f32": [i64, i32, i32, vp, i64, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                     vp, vp, vp],
    "lpf_tail_chain_rows_perm_bf16": [i64, i32, i32, vp, i64, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                      vp, vp, vp],
    "lpf_pair_rows_piece_floats": [i32],
    "lpf_tail_chain_rows_f32": [i64, i32, i32, vp, i64, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp],
    "lpf_tail_chain_rows_bf16": [i64, i32, i32, vp, i64, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp],
    "lpf_tail_chain_merge_f32": [i64, i32, i32, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp,
                                 vp, vp, vp],
    "lpf_tail_chain_merge_bf16": [i64, i32, i32, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp,
                                 vp, vp, vp],
    "lpf_rowdot_sigmoid_f32": [i64, i32, vp, i64, vp, f32, vp, vp, vp],
    "lpf_tail_chain_f32": [i64, i32, i32, vp, i64, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp,
                           vp, vp],
    "lpf_ppr_filter_count": [i64, vp, vp, i32, f32, vp, vp],
    "lpf_ppr_filter_fill": [i64, vp, vp, vp, i32, f32, vp, vp, vp, vp],
    "lpf_self_ppr": [i64, vp, vp, vp, vp, vp, vp, vp],
    "lpf_csr_lookup_f32": [i64, i64, vp, vp, vp, vp, vp, vp, vp],
    "lpf_ppr_push_workspace_bytes": [i64, i64, C.c_double, C.c_double],
    "lpf_ppr_push_f64": [i64, vp, vp, C.c_double, C.c_double, i64, vp, i64, vp, vp, i64, vp, vp, vp, vp],
    "lpf_ppr_pack_workspace_bytes": [i64, i64],
    "lpf_ppr_pack_csr": [i64, vp, vp, vp, vp, i64, vp, vp, vp, vp, i64, vp],
    "lpf_train_partial_blocks": [i64],
    "lpf_pe_hidden_fwd_f32": [i64, i32, vp, vp, vp, vp, vp, vp, vp, i64, vp],
    "lpf_pe_hidden_bwd_f32": [i64, i32, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp],
    "lpf_colsum_f32": [i64, i32, vp, i64, vp, vp, vp],
    "lpf_pair_attention_train_fwd_f32": [i64, i64, i32, vp, vp, vp, i64, vp, i64, vp, i64, vp, vp, vp, i64, vp, vp, vp,
                                         vp],
    "lpf_segment_rows_sum_f32": [i64, i32, vp, vp, vp, i64, vp, i64, vp],
    "lpf_pair_attention_train_bwd_f32": [i64, i64, i32, vp, vp, vp, i64, vp, i64, vp, i64, vp, vp, vp, i64, vp, vp, vp,
                                         vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, vp],
    "lpf_pair_scatter_add_f32": [i64, i32, vp, i64, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp],
    "lpf_dense_chain_f32": [i64, i32, vp, i64, vp, i64, i64, i32, vp, i32, vp, vp, i64, vp, vp, u32, vp, i32, vp, vp, i64,
                            vp, vp],
    "lpf_dense_chain_side_f32": [i64, i32, vp, i64, vp, i64, i64, i32, vp, i32, vp, vp, i64, vp, vp, u32, vp, i32, vp, vp, i64,
                            vp, vp, i64, i32, vp, i64, vp],
}
HOST_PROTOTYPES = {
    "lpf_ppr_push_cpu": [i64, vp, vp, f64, f64, vp, C.POINTER(vp), C.POINTER(vp), i32],
    "lpf_host_free": [vp],
    "lpf_host_abi_version": [],
}
_RESTYPE = {"lpf_strerror": C.c_char_p, "lpf_last_hip_error": C.c_char_p, "lpf_host_free": None,
            "lpf_ppr_push_workspace_bytes": C.c_int64, "lpf_select_plan_blocks": C.c_int64, "lpf_ppr_pack_workspace_bytes": C.c_int64,
            "lpf_gemm_tn_workspace_floats": C.c_int64, "lpf_layernorm_bwd_workspace_floats": C.c_int64,
            "lpf_train_partial_blocks": C.c_int64, "lpf_pair_rows_piece_floats": C.c_int64}


class LpfError(RuntimeError):
    pass


_hip = None
_host = None
_recorder = None   # set by `recording(...)`: the launches, pointers and stream hand-overs of the calls made meanwhile
_recorder_thread = None   # ... by THIS thread only: a concurrent sweep or PPR producer on another thread is not captured


def _recording():
    return _recorder if (_recorder is not None and _recorder_thread == threading.get_ident()) else None


class recording:
    """``with recording(rec):`` every entry point fetched through ``hip()`` is handed to ``rec.launch(name, fn)`` (which
    returns what the caller calls instead), every tensor whose address ``ptr()`` hands out to ``rec.keep(t)``, every
    ``stream_wait(a, b)`` to ``rec.wait(a, b)``.  ``lpformer_amd.PlannedScorer`` records one scoring step this way.  One
    recording at a time, from one thread (the hooks are module state)."""

    def __init__(self, rec):
        self.rec = rec

    def __enter__(self):
        global _recorder, _recorder_thread
        if _recorder is not None:
            raise LpfError("a recording is already in progress")
        _recorder, _recorder_thread = self.rec, threading.get_ident()
        return self.rec

    def __exit__(self, *exc):
        global _recorder, _recorder_thread
        _recorder = _recorder_thread = None
        return False


class _RecordingLib:
    def __init__(self, lib, rec):
        self._lib, self._rec = lib, rec

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        return self._rec.launch(name, fn) if name.startswith("lpf_") else fn


def _bind(lib, protos):
    for name, argtypes in protos.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, C.c_int)
    return lib


def hip():
    """The gfx950 kernel library.  Raises if it has not been built (python __graft_entry__.py build)."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise LpfError(f"{HIP_LIB_PATH} not found: build it with `make -C lpformer_amd/csrc` "
                           "(or __graft_entry__.build()); lpformer_amd has no non-HIP fallback")
        lib = _bind(C.CDLL(HIP_LIB_PATH), HIP_PROTOTYPES)
        if lib.lpf_abi_version() != ABI_VERSION:
            raise LpfError("liblpformer_hip.so ABI version mismatch; rebuild")
        _hip = lib
    rec = _recording()
    return _hip if rec is None else _RecordingLib(_hip, rec)


def host():
    """The host-side library (PPR producer)."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise LpfError(f"{HOST_LIB_PATH} not found: build it with `make -C lpformer_amd/csrc`")
        lib = _bind(C.CDLL(HOST_LIB_PATH), HOST_PROTOTYPES)
        if lib.lpf_host_abi_version() != ABI_VERSION:
            raise LpfError("liblpformer_host.so ABI version mismatch; rebuild")
        _host = lib
    return _host


def check(rc: int, what: str = ""):
    if rc != 0:
        lib = _hip if _hip is not None else hip()
        msg = lib.lpf_strerror(rc).decode()
        extra = lib.lpf_last_hip_error().decode()
        raise LpfError(f"{what or 'lpformer_hip call'} failed: {msg}" + (f" [{extra}]" if extra else ""))


def ptr(t):
    """Device/host pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    rec = _recording()
    if rec is not None:
        rec.keep(t)
    return t.data_ptr()


def stream_wait(waiter, waited):
    """``waiter.wait_stream(waited)`` -- the one way the scoring path hands work from one stream to another, so that a
    recording sees it."""
    rec = _recording()
    if rec is not None:
        rec.wait(waiter, waited)
    waiter.wait_stream(waited)


def device_info():
    cu, lds, wave = C.c_int(), C.c_int(), C.c_int()
    name = C.create_string_buffer(64)
    check(hip().lpf_device_info(C.byref(cu), C.byref(lds), C.byref(wave), name, 64), "lpf_device_info")
    return {"cu_count": cu.value, "lds_bytes_per_cu": lds.value, "wave_size": wave.value, "arch": name.value.decode()}
