"""ctypes binding of liboasr.so (include/oasr.h).  The product path has NO fallback: if the HIP library is missing
or cannot be loaded every entry point raises (the driver's "native code not loaded" check must never be fooled)."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# OASR_LIB: A/B a differently built library of the SAME ABI (kernel experiments); never a fallback.
LIB_PATH = os.environ.get("OASR_LIB") or os.path.join(_HERE, "liboasr.so")
_lib = None


class NativeError(RuntimeError):
    pass


class Dims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
                                       "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")]


class Operand(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("ld", C.c_int64), ("rpb", C.c_int), ("bstride", C.c_int64), ("lead", C.c_int),
                ("kvalid", C.c_int), ("trail_from", C.c_int)]


class GemmArgs(C.Structure):
    _fields_ = [("A", Operand), ("B", Operand), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("ta", C.c_int),
                ("tb", C.c_int), ("alpha", C.c_float), ("bias", C.c_void_p), ("act", C.c_int), ("pos", C.c_void_p),
                ("pos_period", C.c_int), ("dgelu_u", C.c_void_p), ("ldu", C.c_int64), ("resid", C.c_void_p),
                ("ldr", C.c_int64), ("out", C.c_void_p), ("out_pre", C.c_void_p), ("ldc", C.c_int64),
                ("out_f32", C.c_void_p), ("ldc32", C.c_int64), ("beta", C.c_float), ("colsum", C.c_void_p), ("atomic", C.c_int),
                ("split_k", C.c_int), ("dgelu_deriv", C.c_int)]


class AttnArgs(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("ldq", C.c_int64), ("ldk", C.c_int64),
                ("ldv", C.c_int64), ("bsq", C.c_int64), ("bsk", C.c_int64), ("bsv", C.c_int64), ("o", C.c_void_p),
                ("ldo", C.c_int64), ("bso", C.c_int64), ("lse", C.c_void_p), ("o_lo", C.c_void_p), ("kv_len", C.c_void_p), ("B", C.c_int),
                ("H", C.c_int), ("Tq", C.c_int), ("Tk", C.c_int), ("causal", C.c_int), ("d_o", C.c_void_p),
                ("delta", C.c_void_p), ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p), ("dq_colsum", C.c_void_p),
                ("dv_colsum", C.c_void_p), ("colsum_scratch", C.c_void_p), ("qtile_flags", C.c_void_p),
                ("q_rows", C.c_void_p), ("k_rows", C.c_void_p), ("q_span", C.c_void_p)]


ABI_VERSION = 211  # include/oasr.h: OASR_ABI_VERSION (211: the KV cache's control tail is OASR_KV_TAIL_BYTES; 210: OASR_ERETRY from oasr_decode_check)
KV_TAIL_BYTES = 327680  # include/oasr.h: OASR_KV_TAIL_BYTES
ROWTAB = 16        # include/oasr.h: OASR_ROWTAB (entries per sample of a chunk-row table)


def _declare(lib):
    vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
    sig = {
        "oasr_last_error": (C.c_char_p, []),
        "oasr_version": (i32, []),
        "oasr_log_mel_workspace_bytes": (sz, [i32]),
        "oasr_log_mel": (i32, [vp, i32, i32, i32, vp, vp, vp]),
        "oasr_mel_filterbank": (i32, [vp]),
        "oasr_create": (vp, [C.POINTER(Dims)]),
        "oasr_create_ex": (vp, [C.POINTER(Dims), i32]),
        "oasr_create_ex2": (vp, [C.POINTER(Dims), i32, i32]),
        "oasr_compute_dtype": (i32, [vp]),
        "oasr_encode": (i32, [vp, vp, i32, vp, vp, sz, vp]),
        "oasr_kv_cache_bytes": (sz, [vp, i32]),
        "oasr_decode_step_workspace_bytes": (sz, [vp, i32]),
        "oasr_decode_begin": (i32, [vp, vp, i32, vp, vp]),
        "oasr_decode_step": (i32, [vp, vp, i32, i32, vp, vp, vp, sz, vp]),
        "oasr_decode_check": (i32, [vp, i32, vp, vp]),
        "oasr_decode_logits": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp, sz, vp]),
        "oasr_destroy": (None, [vp]),
        "oasr_param_count": (i32, [vp]),
        "oasr_param_info": (i32, [vp, i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(i64), C.POINTER(i32), C.POINTER(i64)]),
        "oasr_param_numel": (i64, [vp]),
        "oasr_segment_count": (i32, [vp]),
        "oasr_segment_info": (i32, [vp, i32, C.POINTER(i64), C.POINTER(i64)]),
        "oasr_bind": (i32, [vp, vp, vp, vp, vp, vp]),
        "oasr_shadow_bytes": (sz, [vp]),
        "oasr_bind_shadow": (i32, [vp, vp]),
        "oasr_refresh_shadow": (i32, [vp, vp]),
        "oasr_workspace_bytes": (sz, [vp, i32, i32, i32]),
        "oasr_forward": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp, sz, vp]),
        "oasr_train_fwd_bwd": (i32, [vp, vp, vp, vp, vp, i32, f32, f32, vp, i32, vp, vp, vp, sz, vp]),
        "oasr_train_fwd_bwd_s": (i32, [vp, vp, vp, vp, vp, i32, i32, f32, f32, vp, i32, vp, vp, vp, sz, vp]),
        "oasr_train_fwd_bwd_span": (i32, [vp, vp, vp, vp, vp, vp, i32, vp, i32, f32, f32, vp, i32, vp, vp, sz, vp]),
        "oasr_log_mel_raw": (i32, [vp, i32, i32, i32, vp, vp, vp, vp]),
        "oasr_sizeof_attn_args": (sz, []),
        "oasr_test_span_tables": (i32, [vp, i32, i32, vp, vp, vp, vp, vp, vp]),
        "oasr_train_fwd": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, sz, vp]),
        "oasr_train_bwd": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, sz, vp]),
        "oasr_zero_grad": (i32, [vp, vp]),
        "oasr_optim_step": (i32, [vp, f32, f32, f32, f32, f32, f32, f32, i64, vp, vp, vp]),
        "oasr_grad_sumsq_range": (i32, [vp, i64, i64, vp, vp, vp]),
        "oasr_optim_step_range": (i32, [vp, i64, i64, vp, vp, vp, f32, f32, f32, f32, f32, f32, f32, i64, vp]),
        "oasr_gemm": (i32, [C.POINTER(GemmArgs), vp]),
        "oasr_layernorm_fwd": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, vp]),
        "oasr_layernorm_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp]),
        "oasr_attention_fwd": (i32, [C.POINTER(AttnArgs), vp]),
        "oasr_attention_bwd": (i32, [C.POINTER(AttnArgs), vp]),
        "oasr_attention_scores": (i32, [C.POINTER(AttnArgs), i32, vp, vp]),
        "oasr_cross_entropy": (i32, [vp, i64, i32, vp, i64, i64, f32, vp, vp, vp, i32, vp]),
        "oasr_cast_f32_bf16": (i32, [vp, vp, i64, vp]),
        "oasr_pick_tokens": (i32, [vp, i64, i32, i64, vp, vp, vp, vp, vp]),
        "oasr_pick_tokens_ts": (i32, [vp, i64, i32, i64, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp, vp, vp]),
        "oasr_topk_tokens": (i32, [vp, i64, i32, i64, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
        "oasr_sample_tokens": (i32, [vp, i64, i32, i64, vp, vp, vp, i64, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp]),
        "oasr_probe_tr16": (i32, [vp, vp, vp]),
        "oasr_probe_lds_oob": (i32, [vp, vp, vp]),
        "oasr_profile_gemm": (i32, [i32]),
        "oasr_gemm_force_general": (i32, [i32]),
        "oasr_gemm_set_stagger": (i32, [i32, i32]),
        "oasr_gemm_set_variant": (i32, [i32]),
        "oasr_decode_set_ln_fold": (i32, [i32]),
        "oasr_span_set_side_streams": (i32, [i32]),
        "oasr_span_side_streams": (i32, []),
        "oasr_xcd_plan_debug": (i32, [i32, i32, i32, i32, i32, i32, i32, vp, i32]),
        "oasr_xcd_offsets_ok_debug": (i32, [vp, C.c_longlong, C.c_longlong, i32, i32, i32, i32]),
        "oasr_wide_supports_debug": (i32, [i32, i32, i32, i32, i32, i32, i32]),
        "oasr_wide_plan_debug": (i32, [i32, i32, i32, i32, i32, i32, vp, i32]),
        "oasr_attention_set_pingpong": (i32, [i32]),
        "oasr_profile_gemm_collect": (i32, [vp, vp, vp, C.c_char_p, i32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return list(sig)


EXPORTS = None


def lib():
    """Loads liboasr.so (once).  Raises NativeError when it is missing -- build it with __graft_entry__.build()."""
    global _lib, EXPORTS
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise NativeError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950).  There is no CPU/PyTorch fallback for this path.")
        try:
            _lib = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise NativeError(f"cannot load {LIB_PATH}: {e}") from e
        handle, _lib = _lib, None  # published only once it has passed every check below (a failed load must not leave a half-declared CDLL behind)
        # a library of another ABI generation reads structs of a different size through the same pointers, and may lack symbols this
        # binding declares: check the version FIRST (oasr_version exists in every generation), then the struct size, then declare
        vfn = getattr(handle, "oasr_version", None)
        ver = int(vfn()) if vfn is not None else None
        sfn = getattr(handle, "oasr_sizeof_attn_args", None)
        size = int(sfn()) if sfn is not None else None
        if ver != ABI_VERSION or size != C.sizeof(AttnArgs):
            raise NativeError(f"{LIB_PATH}: ABI version {ver} / oasr_attn_args of {size} bytes, this binding is written for version "
                              f"{ABI_VERSION} / {C.sizeof(AttnArgs)} bytes -- rebuild (__graft_entry__.build())")
        try:
            exports = _declare(handle)
        except AttributeError as e:
            raise NativeError(f"{LIB_PATH} (ABI {ver}) lacks an entry point this binding declares: {e} -- rebuild (__graft_entry__.build())") from e
        _lib, EXPORTS = handle, exports
    return _lib


def enable_testing_hooks():
    """Opt this process in to the kernel-selection setters of include/oasr_testing.h (they are inert otherwise)."""
    os.environ["OASR_TESTING_HOOKS"] = "1"


ERETRY = -4  # include/oasr.h: OASR_ERETRY


def check(rc, what=""):
    if rc != 0:
        msg = lib().oasr_last_error()
        raise NativeError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device (or host) address of a tensor; None -> NULL."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu(t, name="tensor"):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise NativeError(f"{name} must be a CUDA/HIP tensor: the MI355X-native path has no CPU fallback")
