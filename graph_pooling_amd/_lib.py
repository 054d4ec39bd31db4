"""ctypes binding of libdiffpool_hip.so (include/diffpool_hip.h).

The library is the product: if it is missing this module raises at first use — there is no
PyTorch / CPU fallback behind these calls.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DP_LIB") or os.path.join(_HERE, "libdiffpool_hip.so")   # DP_LIB: a diagnostic build

DP_MAX_LAYERS = 8
DP_MAX_LEVELS = 4
DP_MAX_PRED = 4

F_ADD_SELF, F_NORMALIZE, F_RELU, F_BN, F_LAST_ONLY = 1, 2, 4, 8, 16
MODE_EVAL, MODE_TRAIN = 0, 1
ERR_DEVICE = -4                                   # DP_ERR_DEVICE
DEVERR_BARRIER, DEVERR_NONFINITE_GRAD = 1, 2      # DP_DEVERR_*
SAVE_S, SAVE_XPOOL, SAVE_ADJPOOL, SAVE_Z, SAVE_ZASSIGN, SAVE_ARGMAX = 0, 1, 2, 3, 4, 5


class StackCfg(C.Structure):
    _fields_ = [("n_layers", C.c_int),
                ("dims", C.c_int * (DP_MAX_LAYERS + 1)),
                ("w_off", C.c_long * DP_MAX_LAYERS),
                ("b_off", C.c_long * DP_MAX_LAYERS),
                ("drop_off", C.c_long * DP_MAX_LAYERS)]


class EncoderCfg(C.Structure):
    _fields_ = [("B", C.c_int), ("N", C.c_int),
                ("num_pooling", C.c_int),
                ("n_nodes", C.c_int * (DP_MAX_LEVELS + 1)),
                ("embed", StackCfg * (DP_MAX_LEVELS + 1)),
                ("assign", StackCfg * DP_MAX_LEVELS),
                ("assign_pred_w_off", C.c_long * DP_MAX_LEVELS),
                ("assign_pred_b_off", C.c_long * DP_MAX_LEVELS),
                ("n_pred", C.c_int),
                ("pred_dims", C.c_int * (DP_MAX_PRED + 2)),
                ("pred_w_off", C.c_long * (DP_MAX_PRED + 1)),
                ("pred_b_off", C.c_long * (DP_MAX_PRED + 1)),
                ("flags", C.c_int),
                ("readout", C.c_int),
                ("s2s_off", C.c_long * 6),
                ("mask_readout", C.c_int),
                ("n_params", C.c_long),
                ("n_graph_params", C.c_long),
                ("bn_world", C.c_int),
                ("exchange", C.c_void_p),
                ("exchange_user", C.c_void_p)]


# int (*dp_exchange_fn)(void* user, const void* local, void* gathered, size_t bytes_per_rank, void* stream)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


_P = C.c_void_p
_I = C.c_int
_L = C.c_long
_F = C.c_float
_Z = C.c_size_t

# name -> (restype, argtypes); mirrors include/diffpool_hip.h one to one
_PROTOS = {
    "dp_version": (_I, []),
    "dp_last_error_string": (C.c_char_p, []),
    "dp_device_error": (_I, [_I]),
    "dp_device_error_describe": (C.c_char_p, [_I]),
    "dp_profile_level0": (_I, [_I]),
    "dp_profile_level0_read": (_I, [_I, C.POINTER(C.c_double), C.POINTER(_I)]),
    "dp_sizeof_encoder_cfg": (_Z, []),
    "dp_bgemm_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _L, _L, _L, _I, _I, _F, _F, _I, _P]),
    "dp_bgemm_split_bf16": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _L, _L, _L, _I, _I, _F, _P]),
    "dp_adj_aggregate": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _I, _F, _P]),
    "dp_adj_pack_ld": (_I, [_I]),
    "dp_adj_pack_bytes": (_Z, [_I, _I]),
    "dp_adj_pack": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "dp_adj_aggregate_packed_workspace_bytes": (_Z, [_I, _I, _I]),
    "dp_adj_aggregate_packed": (_I, [_P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _F, _I, _P, _Z, _P]),
    "dp_gcn_layer_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "dp_gcn_layer_fwd": (_I, [_P, _I, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "dp_gcn_layer_bwd": (_I, [_P, _I, _P, _P, _P, _I, _P, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "dp_bn_node_workspace_bytes": (_Z, [_I, _I, _I]),
    "dp_bn_node_fwd": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _Z, _P]),
    "dp_bn_node_bwd": (_I, [_P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "dp_assign_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "dp_assign_softmax_mask_fwd": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _Z, _P]),
    "dp_assign_softmax_mask_bwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _P, _Z, _P]),
    "dp_pool_fwd": (_I, [_P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dp_pool_bwd_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "dp_pool_bwd": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _P, _Z, _P]),
    "dp_masked_max_fwd": (_I, [_P, _I, _P, _P, _I, _P, _I, _I, _I, _P]),
    "dp_masked_max_bwd": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P]),
    "dp_linkpred_workspace_bytes": (_Z, [_I, _I, _I]),
    "dp_linkpred_loss_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _Z, _P]),
    "dp_linkpred_loss_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _Z, _P]),
    "dp_cross_entropy_fwd": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "dp_cross_entropy_bwd": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "dp_set2set_save_bytes": (_Z, [_I, _I, _I]),
    "dp_set2set_fwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _Z, _P]),
    "dp_set2set_bwd_workspace_bytes": (_Z, [_I, _I, _I]),
    "dp_set2set_bwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P,
                            _I, _I, _I, _P, _Z, _P, _Z, _P]),
    "dp_mean_aggregate_fwd": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _P]),
    "dp_mean_aggregate_bwd": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _P]),
    "dp_csr_aggregate": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "dp_sparse_gcn_layer_workspace_bytes": (_Z, [_I, _I, _I]),
    "dp_sparse_gcn_layer_fwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _P, _Z, _P]),
    "dp_sparse_gcn_layer_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P, _Z,
                                     _P]),
    "dp_encoder_save_bytes": (_Z, [C.POINTER(EncoderCfg)]),
    "dp_encoder_workspace_bytes": (_Z, [C.POINTER(EncoderCfg)]),
    "dp_encoder_save_locate": (_I, [C.POINTER(EncoderCfg), _I, _I, C.POINTER(_Z), C.POINTER(_Z)]),
    "dp_encoder_forward": (_I, [C.POINTER(EncoderCfg), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P, _Z, _I, _P]),
    "dp_encoder_backward": (_I, [C.POINTER(EncoderCfg), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P, _Z, _I, _P]),
    "dp_encoder_forward_packed": (_I, [C.POINTER(EncoderCfg), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P, _Z, _I,
                                       _P]),
    "dp_encoder_backward_packed": (_I, [C.POINTER(EncoderCfg), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P, _Z, _I,
                                        _P]),
    "dp_build_batch": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dp_build_batch_packed": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dp_clip_adam_workspace_bytes": (_Z, []),
    "dp_clip_adam_step": (_I, [_P, _P, _P, _P, _L, _I, _F, _F, _F, _F, _F, _P, _P, _Z, _P]),
    "dp_clip_adam_step_counted": (_I, [_P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _F, _P, _P, _Z, _P]),
    "dp_gather_labels": (_I, [_P, _P, _I, _P]),
    "dp_loss_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "dp_loss_forward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "dp_loss_backward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _Z, _P]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS.keys())

_lib = None
_lock = threading.Lock()


def load():
    """Load the shared library (once). Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the DiffPool HIP extension is not built. Run "
                "graph_pooling_amd/csrc/build.sh (or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "There is no PyTorch fallback for this path.")
        # torch must load ITS libamdhip64 first: the .so then binds to that already-loaded runtime. Loading
        # ours first puts two HIP runtimes in one process and the second one finds no device.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(lib, name)      # AttributeError if the .so does not export what the header declares
            fn.restype = res
            fn.argtypes = args
        if lib.dp_sizeof_encoder_cfg() != C.sizeof(EncoderCfg):
            raise RuntimeError("dp_encoder_cfg layout mismatch between diffpool_hip.h and _lib.py: "
                               f"{lib.dp_sizeof_encoder_cfg()} vs {C.sizeof(EncoderCfg)}")
        _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().dp_last_error_string().decode("utf-8", "replace")
        raise RuntimeError(f"{what or 'libdiffpool_hip'} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_gpu_tensor(t, name):
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a tensor on the GPU: the DiffPool HIP path has no CPU implementation "
                           "(move the model and its inputs to 'cuda')")
