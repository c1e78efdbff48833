"""ctypes binding of libfvta_hip.so (the C ABI declared in include/fvta_hip.h).

No fallback: if the shared library is missing or a call fails, this raises.
PyTorch is used by the callers only to own device memory and streams; every
pointer handed over here is a raw device address.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# FVTA_LIB_PATH: a diagnostics build of the same library (tools/: stamped / ablated kernels), never a different backend
LIB_PATH = os.environ.get("FVTA_LIB_PATH") or os.path.join(_HERE, "csrc", "libfvta_hip.so")

F32, BF16, BF16X3 = 0, 1, 2


class FvtaError(RuntimeError):
    pass


class AttnDesc(Structure):
    _fields_ = [(n, c_int32) for n in ("N", "K", "T", "JQ", "w", "simi", "feat_order", "add_tanh")] + [("hinfo_stride", c_int64)]


class LstmDesc(Structure):
    _fields_ = [(n, c_int32) for n in ("B", "J", "in_", "d", "share_fw_bw", "precision", "training", "reserved", "dx_overwrite", "out_pads_persist")] + \
               [("out_skip", c_int64)]


class TimewarpDesc(Structure):
    _fields_ = [(n, c_int32) for n in ("N", "K", "T", "w", "warp_type")] + [("window_t", c_float)]


class EmbedDesc(Structure):
    _fields_ = [(n, c_int32) for n in ("ntok", "W", "cdim", "cwdim", "wdim", "VW", "VT", "VC", "height")] + \
               [("keep_prob", c_float), ("dropout_seed", ctypes.c_uint64)]


class ImgTransDesc(Structure):
    _fields_ = [(n, c_int32) for n in ("M", "idim", "tdim", "add_tanh")]


class ScorerDesc(Structure):
    _fields_ = [(n, c_int32) for n in ("N", "C", "w", "use_eu_output", "add_tanh", "xent_grad")]


P = c_void_p
_SIGS = {
    "fvta_version": (c_int, []),
    "fvta_last_error": (c_char_p, []),
    "fvta_abi_struct_bytes": (c_int64, [c_int32]),
    "fvta_attn_workspace_bytes": (c_size_t, [POINTER(AttnDesc)]),
    "fvta_attn_saved_bytes": (c_size_t, [POINTER(AttnDesc)]),
    "fvta_attn_fwd": (c_int, [POINTER(AttnDesc), P, P, P, P, P, P, P, P, P, P, P]),
    "fvta_attn_bwd": (c_int, [POINTER(AttnDesc), P, P, P, P, P, P, P, P, P, P, P, P, c_int, P, P]),
    "fvta_attn_fwd_shadow": (c_int, [POINTER(AttnDesc), P, P, P, P, P, P, P, P, P, P]),
    "fvta_attn_bwd_shadow": (c_int, [POINTER(AttnDesc), P, P, P, P, P, P, P, P, P, P, P, P, c_int, P, P]),
    "fvta_attn_fwd_tw": (c_int, [POINTER(AttnDesc), P, P, P, P, P, P, P, P, P, P, P, P]),
    "fvta_attn_bwd_tw": (c_int, [POINTER(AttnDesc), P, P, P, P, P, P, P, P, P, P, P, P, P, P, c_int, P, P]),
    "fvta_timewarp_bwd_att": (c_int, [POINTER(TimewarpDesc), P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "fvta_lstm_plan_bytes": (c_size_t, [POINTER(LstmDesc)]),
    "fvta_lstm_saved_bytes": (c_size_t, [POINTER(LstmDesc)]),
    "fvta_lstm_workspace_bytes": (c_size_t, [POINTER(LstmDesc)]),
    "fvta_lstm_plan": (c_int, [POINTER(LstmDesc), P, P, P, P, c_int64, P, P]),
    "fvta_lstm_plan_xdir": (c_int, [POINTER(LstmDesc), P, P, P, P, c_int64, c_int64, P, P]),
    "fvta_dropout_pair_fwd": (c_int, [P, P, c_int64, c_float, ctypes.c_uint64, P]),
    "fvta_dropout_pair_bwd": (c_int, [P, P, c_int64, c_float, ctypes.c_uint64, c_int32, P]),
    "fvta_bilstm_fwd": (c_int, [POINTER(LstmDesc), P, P, P, P, P, P, P, P, P, P]),
    "fvta_bilstm_bwd": (c_int, [POINTER(LstmDesc), P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "fvta_bilstm_bwd_overlap": (c_int, [POINTER(LstmDesc), P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "fvta_bilstm_bwd_hint": (c_int, [POINTER(LstmDesc), P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "fvta_lstm_shadow_rows": (c_int, [POINTER(LstmDesc), P, P, c_int64, P, P]),
    "fvta_rows_from_shadow": (c_int, [P, c_int64, c_int32, c_int64, P, P]),
    "fvta_lstm_last_state": (c_int, [POINTER(LstmDesc), P, P, c_int32, c_int32, P, P]),
    "fvta_lstm_last_state_bwd": (c_int, [POINTER(LstmDesc), P, P, c_int32, c_int32, P, P]),
    "fvta_scorer_ce_fwd": (c_int, [POINTER(ScorerDesc), P, P, P, P, P, P, P, P, P, P]),
    "fvta_scorer_ce_bwd": (c_int, [POINTER(ScorerDesc), P, P, P, P, P, P, P, P, c_float, P, P, P, P, P, P]),
    "fvta_attgru_fwd": (c_int, [c_int32, c_int32, P, P, P, P, P, P, P, P, P, P]),
    "fvta_attgru_bwd": (c_int, [c_int32, c_int32, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "fvta_timewarp_workspace_bytes": (c_size_t, [POINTER(TimewarpDesc)]),
    "fvta_timewarp_fwd": (c_int, [POINTER(TimewarpDesc), P, P, P, P, P, P, P, P, P, P, P]),
    "fvta_timewarp_bwd": (c_int, [POINTER(TimewarpDesc), P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "fvta_timewarp_fwd_shadow": (c_int, [POINTER(TimewarpDesc), P, P, P, P, P, P, P, P, P, P, P]),
    "fvta_timewarp_bwd_shadow": (c_int, [POINTER(TimewarpDesc), P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "fvta_embed_workspace_bytes": (c_size_t, [POINTER(EmbedDesc)]),
    "fvta_embed_fwd": (c_int, [POINTER(EmbedDesc), P, P, P, P, P, P, P, P, P, P, P]),
    "fvta_embed_bwd": (c_int, [POINTER(EmbedDesc), P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "fvta_image_trans_fwd": (c_int, [POINTER(ImgTransDesc), P, P, P, P, P, P, P]),
    "fvta_image_trans_bwd": (c_int, [POINTER(ImgTransDesc), P, P, P, P, P, P, P, P, P]),
    "fvta_adadelta_step": (c_int, [P, P, P, P, c_int64, c_float, c_float, c_float, c_float, P]),
    "fvta_adam_step": (c_int, [P, P, P, P, c_int64, c_float, c_float, c_float, c_float, c_int32, c_float, P]),
    "fvta_weight_decay": (c_int, [P, P, c_int64, c_float, P, P]),
    "fvta_probe_hbm_read": (c_int, [P, ctypes.c_size_t, P, P]),
    "fvta_probe_spin": (c_int, [c_int64, P]),
    "fvta_probe_hbm_mix": (c_int, [P, ctypes.c_size_t, c_int32, c_int32, P]),
    "fvta_softmax_fwd": (c_int, [P, P, c_int64, c_int32, P]),
    "fvta_softsel_fwd": (c_int, [P, P, P, c_int64, c_int32, c_int32, P]),
    "fvta_exp_mask": (c_int, [P, P, P, c_int64, P]),
    "fvta_linear_fwd": (c_int, [P, P, P, P, c_int64, c_int32, c_int32, c_int32, P]),
    "fvta_wsum_fwd": (c_int, [P, P, P, c_int64, c_int32, c_int32, P]),
    "fvta_dmn_features": (c_int, [P, P, P, P, c_int32, c_int32, c_int32, P]),
    "fvta_dmn_features_bwd": (c_int, [P, P, P, P, P, P, P, c_int32, c_int32, c_int32, P]),
    "fvta_relu_fwd": (c_int, [P, P, c_int64, P]),
    "fvta_relu_bwd": (c_int, [P, P, P, c_int64, P]),
    "fvta_linear_fwd_blk": (c_int, [P, P, P, P, c_int64, c_int32, c_int32, c_int32, c_int64, c_int64, P]),
    "fvta_linear_bwd_blk": (c_int, [P, P, P, P, P, P, P, c_int64, c_int32, c_int32, c_int32, c_int32, c_int64, c_int64, P]),
    "fvta_softmax_bwd": (c_int, [P, P, P, c_int64, c_int32, P]),
    "fvta_wsum_fwd_ld": (c_int, [P, P, P, c_int64, c_int32, c_int32, c_int64, P]),
    "fvta_wsum_bwd": (c_int, [P, P, P, P, P, c_int64, c_int32, c_int32, c_int64, P]),
    "fvta_linear_bwd": (c_int, [P, P, P, P, P, P, P, c_int64, c_int32, c_int32, c_int32, c_int32, P]),
    "fvta_attn_qside_fwd": (c_int, [P, P, P, c_int32, c_int32, c_int32, c_int32, P]),
    "fvta_attn_qside_bwd": (c_int, [P, P, P, P, P, c_int32, c_int32, c_int32, c_int32, P]),
    "fvta_attn_logits_bwd_workspace_bytes": (c_size_t, [POINTER(AttnDesc)]),
    "fvta_attn_logits_bwd": (c_int, [POINTER(AttnDesc), P, P, P, P, P, P, P, P, P, P]),
    "fvta_rows_reduce": (c_int, [P, P, c_int64, c_int32, c_int32, c_int64, c_float, c_int32, P]),
    "fvta_rows_broadcast": (c_int, [P, P, c_int64, c_int32, c_int32, c_int64, c_float, c_int32, P]),
    "fvta_attn_read_u": (c_int, [POINTER(AttnDesc), P, P, P]),
    "fvta_test_gemm": (c_int, [c_int32, c_int32, c_int32, c_int32, c_int32, P, P, P, P]),
    "fvta_profile_enable": (c_int, [c_int32]),
    "fvta_lstm_kernel_select": (c_int, [c_int32]),
    "fvta_lstm_bwd_kernel_counts": (c_int, [POINTER(c_int64)]),
    "fvta_attn_kernel_select": (c_int, [c_int32, c_int32]),
    "fvta_profile_collect": (c_int, [c_int32, POINTER(ctypes.c_double), POINTER(c_int64)]),
}

_lib = None


def exported_symbols():
    """Every entry point include/fvta_hip.h declares."""
    return sorted(_SIGS)


def load():
    """Load the library once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64 (SONAME libamdhip64.so.7).  Import it FIRST so that our
    # library binds to the SAME HIP runtime instance; loading ours first would pull in /opt/rocm's
    # copy as a second runtime, which then sees no device.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise FvtaError(
            "%s not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C fvta_memexqa_amd/csrc`). There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    # the descriptor structs are declared twice (include/fvta_hip.h and above): refuse a library whose layout differs
    for which, cls in enumerate((AttnDesc, LstmDesc, ScorerDesc, TimewarpDesc, EmbedDesc, ImgTransDesc)):
        have, want = ctypes.sizeof(cls), lib.fvta_abi_struct_bytes(which)
        if have != want:
            raise FvtaError("%s is %d bytes here but %d in %s: rebuild the library (descriptor layouts differ)"
                            % (cls.__name__, have, want, LIB_PATH))
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().fvta_last_error()
        raise FvtaError("%s failed (%d): %s" % (what, status, msg.decode() if msg else "?"))


def ptr(t):
    """Raw device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
