"""ctypes binding of libgts_hip.so (C ABI declared in include/gts_hip.h).

There is NO CPU fallback: if the library is missing or a tensor is not on an AMD GPU the
call raises.  The oracle under /oracle is never reachable from here.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# GTS_LIB_PATH: another build of the same library (A/B runs of two builds in one session, tools/); the default is the in-tree build
LIB_PATH = os.environ.get("GTS_LIB_PATH") or os.path.join(_HERE, "libgts_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "gts_hip.h")
ABI_VERSION = 21

_p = ctypes.c_void_p
_i32 = ctypes.c_int32
_i64 = ctypes.c_int64
_f32 = ctypes.c_float
_f64 = ctypes.c_double

# name -> argtypes (restype is int32 unless noted)
SIGNATURES = {
    "gts_abi_version": [],
    "gts_error_string": [_i32],
    "gts_spmm_max_fwd_f32": [_p, _p, _p, _p, _p, _i32, _i32, _i64, _i64, _p],
    "gts_spmm_max_bwd_f32": [_p, _p, _p, _p, _p, _i32, _p, _p, _i64, _i64, _p],
    "gts_cluster_record_words": [_i32, _i32, _i32, _i32],
    "gts_cluster_schedule": [_p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _p, _i64, _p, _p, _p],
    "gts_cluster_lds_bytes": [_i32, _i32, _i32, _i32],
    "gts_cluster_uses_counters": [_i64, _i32, _i32, _i32, _i32],
    "gts_spmm_max_fwd_cluster_f32": [_p, _i64, _i32, _i32, _i32, _p, _p, _p, _i32, _i32, _i64, _i64, _p, _p],
    "gts_spmm_max_bwd_cluster_f32": [_p, _i64, _i32, _i32, _i32, _p, _p, _i32, _p, _i64, _i64, _p, _p],
    "gts_spmm_sum_f32": [_p, _p, _p, _p, _p, _p, _p, _i32, _i64, _i64, _p],
    "gts_gat_fwd_f32": [_p, _p, _p, _p, _p, _f32, _p, _p, _i32, _p, _p, _i64, _i64, _i64, _p],
    "gts_gat_scores_f32": [_p, _p, _p, _p, _p, _i64, _i64, _i64, _p],
    "gts_gat_reduce_workspace": [_i64, _i64],
    "gts_gat_act_bwd_f32": [_p, _p, _i32, _p, _p, _p, _i64, _i64, _i64, _p],
    "gts_gat_cluster_workspace": [_i64, _i32, _i32, _i64, _i32],
    "gts_gat_attn_f32": [_p, _p, _p, _p, _f32, _p, _i64, _i64, _i64, _p],
    "gts_gat_fwd_cluster_f32": [_p, _p, _p, _i64, _i32, _i32, _i32, _i32, _p, _p, _p, _f32, _p, _i32, _p, _p, _p, _i64,
                                _i64, _i64, _i64, _i64, _p],
    "gts_gat_bwd_edge_cluster_f32": [_p, _p, _p, _i64, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _f32, _p, _p, _p, _i64,
                                     _i64, _i64, _i64, _i64, _p],
    "gts_gat_bwd_src_cluster_f32": [_p, _p, _p, _i64, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _i64,
                                    _i64, _i64, _i64, _i64, _p],
    "gts_gat_param_grad_f32": [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p],
    "gts_gat_bwd_edge_f32": [_p, _p, _p, _p, _p, _p, _p, _f32, _p, _p, _i64, _i64, _i64, _p],
    "gts_gat_bwd_src_f32": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p],
    "gts_project_rows_i16": [_p, _p, _p, _p, _i64, _i64, _i32, _p],
    "gts_project_argmax_i16": [_p, _p, _p, _p, _i64, _i64, _i64, _p],
    "gts_project_argmax_occupancy_i16": [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p],
    "gts_crop_concat_f32": [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _p],
    "gts_argmax_scatter_i16": [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _p],
    "gts_adamw_f32": [_p, _p, _p, _p, _i64, _f64, _f64, _f64, _f64, _f64, _i64, _p],
    "gts_label_confusion_workspace": [_i64],
    "gts_label_confusion_i16": [_p, _p, _p, _p, _i64, _i64, _p],
    "gts_linear_fwd_f32": [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i32, _p, _p, _p],
    "gts_relu_bits_bytes": [_i64, _i64],
    "gts_relu_bits_pay": [_i64, _i64],
    "gts_linear_bwd_input_f32": [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p],
    "gts_linear_bwd_input_t_f32": [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p, _p],
    "gts_linear_bwd_input_t_act_workspace": [_i64, _i64],
    "gts_linear_bwd_input_t_act_f32": [_p, _p, _p, _p, _p, _i32, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p, _p],
    "gts_transpose_batch_f32": [_p, _p, _i32, _i64, _i64, _p],
    "gts_packed_weight_floats": [_i64, _i64],
    "gts_pack_weights_f32": [_p, _p, _p, _i32, _i64, _i64, _i32, _p],
    "gts_gat_fc_scores_workspace": [_i64, _i64, _i64],
    "gts_gat_fc_scores_f32": [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p, _p],
    "gts_linear_fwd_chain_f32": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i32, _i64, _i32, _p, _p, _p],
    "gts_linear_bwd_input_chain_t_f32": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p, _p],
    "gts_linear_bwd_weight_workspace": [_i64, _i64, _i64, _i32],
    "gts_linear_bwd_weight_f32": [_p, _p, _p, _p, _i32, _p, _i64, _i64, _i64, _i64, _p],
    "gts_sage_pool_stack_fwd_arena": [_i64, _p, _i32, _i32, _i32, _i32, _p],
    "gts_sage_pool_stack_fwd_f32": [_p, _p, _p, _i64, _i32, _i32, _i32, _p, _p, _i64, _p, _i32, _i32, _i32, _i32, _p, _i64, _p],
    "gts_sage_pool_stack_bwd_scratch": [_i64, _p, _i32, _i32],
    "gts_sage_pool_stack_bwd_f32": [_p, _p, _p, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _i64, _p, _i32, _i32, _i32, _p, _p, _p,
                                    _p, _i64, _p],
    "gts_set_option": [_i32, _i32],
    "gts_get_option": [_i32],
    "gts_weighted_ce_workspace": [_i64],
    "gts_weighted_ce_f32": [_p, _p, _p, _p, _p, _i64, _p, _i64, _i64, _p],
    "gts_collate_plan": [_p, _i32, _i64, _p, _i32, _p],
    "gts_collate_batch": [_p, _i32, _i64, _p, _i32, _p, _i64, _i32, _p],
}
_RESTYPE = {"gts_error_string": ctypes.c_char_p, "gts_linear_bwd_weight_workspace": _i64,
            "gts_weighted_ce_workspace": _i64, "gts_gat_reduce_workspace": _i64,
            "gts_label_confusion_workspace": _i64, "gts_gat_fc_scores_workspace": _i64,
            "gts_relu_bits_bytes": _i64, "gts_cluster_record_words": _i64, "gts_sage_pool_stack_fwd_arena": _i64, "gts_sage_pool_stack_bwd_scratch": _i64, "gts_cluster_lds_bytes": _i64,
            "gts_linear_bwd_input_t_act_workspace": _i64, "gts_gat_cluster_workspace": _i64,
            "gts_packed_weight_floats": _i64}

COLLATE_MAX_SCHEDULES = 6      # GTS_COLLATE_MAX_SCHEDULES


class CollateMember(ctypes.Structure):
    """gts_collate_member_t (include/gts_hip.h): one member of a batch, host pointers."""
    _fields_ = [("n_nodes", _i64), ("n_edges", _i64),
                ("indptr", _p), ("indices", _p), ("t_indptr", _p), ("t_indices", _p), ("t_slot", _p), ("t_pos", _p),
                ("features", _p), ("labels", _p), ("feat_bytes", _i32), ("label_bytes", _i32),
                ("sched_rec", _p * COLLATE_MAX_SCHEDULES), ("sched_clusters", _i64 * COLLATE_MAX_SCHEDULES),
                ("sched_loc_words", _i32 * COLLATE_MAX_SCHEDULES)]


class CollateKind(ctypes.Structure):
    """gts_collate_kind_t: the limits one kind of cluster schedule was built with."""
    _fields_ = [("max_rows", _i32), ("max_srcs", _i32), ("tagged", _i32), ("reserved", _i32)]


class CollatePlan(ctypes.Structure):
    """gts_collate_plan_t: sizes and byte offsets of the collated block."""
    _fields_ = [("total_bytes", _i64), ("n_nodes", _i64), ("n_edges", _i64), ("features", _i64), ("labels", _i64),
                ("csr", _i64 * 8), ("sched", _i64 * COLLATE_MAX_SCHEDULES),
                ("sched_clusters", _i64 * COLLATE_MAX_SCHEDULES), ("sched_loc_words", _i32 * COLLATE_MAX_SCHEDULES),
                ("sched_record_words", _i32 * COLLATE_MAX_SCHEDULES)]


_lib = None


class GtsError(RuntimeError):
    pass


def declared_symbols(header_path=HEADER_PATH):
    """Every function name declared in include/gts_hip.h."""
    with open(header_path) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(gts_[a-z0-9_]+)\s*\(", text)))


def load():
    """Load the HIP library once; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GtsError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (hipcc, gfx950). "
            "The gts operators have no CPU or PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError -> the .so is stale; surface it
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, _i32)
    if lib.gts_abi_version() != ABI_VERSION:
        raise GtsError(f"libgts_hip.so ABI {lib.gts_abi_version()} != expected {ABI_VERSION}: rebuild")
    # tuning knobs from the environment, e.g. GTS_OPTIONS="1=5,3=1" (option=value pairs, gts_set_option)
    for pair in filter(None, os.environ.get("GTS_OPTIONS", "").split(",")):
        opt, val = pair.split("=")
        if lib.gts_set_option(int(opt), int(val)) != 0:
            raise GtsError(f"GTS_OPTIONS: unknown option {opt}")
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().gts_error_string(code)
        raise GtsError(f"{what} failed with code {code}: {msg.decode() if msg else '?'}")


def require_device(*tensors):
    """All tensors must live on one HIP device and be contiguous."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise GtsError("gts operators run on MI355X only: got a CPU tensor "
                           "(there is no CPU fallback; use the oracle in tests)")
        if not t.is_contiguous():
            raise GtsError("gts operators need contiguous tensors")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise GtsError(f"tensors on different devices: {dev} vs {t.device}")
    return dev


def ptr(t):
    return None if t is None else t.data_ptr()


def current_stream():
    import torch

    return torch.cuda.current_stream().cuda_stream
