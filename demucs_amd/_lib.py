"""ctypes binding of libdemucs_amd.so (C ABI declared in include/demucs_amd.h).

There is NO fallback: if the shared library is missing or fails to load, every entry point
raises.  The product never routes through PyTorch eager ops or the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DEMUCS_AMD_LIB") or os.path.join(_HERE, "libdemucs_amd.so")   # override: kernel A/B builds

MI_OK = 0


class MiTensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_int64)]


class MiConfig(C.Structure):
    _fields_ = [("n_sources", C.c_int32), ("segment_length", C.c_int32), ("max_batch", C.c_int32),
                ("dtype", C.c_int32)]


class MiProfileRow(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


class MiConvDesc(C.Structure):
    """mi_conv_desc of demucs_amd/csrc/gemm_conv.h (field order must match)."""
    _fields_ = [
        ("wt", C.c_void_p), ("M", C.c_int32), ("Mpad", C.c_int32), ("K", C.c_int32), ("Kpad", C.c_int32),
        ("ktab", C.c_void_p),
        ("x", C.c_void_p), ("x_bstride", C.c_int64),
        ("B", C.c_int32), ("D1", C.c_int32), ("D2", C.c_int32), ("O1", C.c_int32), ("O2", C.c_int32),
        ("S1", C.c_int32), ("S2", C.c_int32),
        ("pro", C.c_int32), ("pro_stats", C.c_void_p), ("pro_w", C.c_void_p), ("pro_b", C.c_void_p),
        ("row_mode", C.c_int32),
        ("epi", C.c_int32), ("flags", C.c_int32),
        ("bias", C.c_void_p), ("scale", C.c_void_p), ("res", C.c_void_p), ("emb", C.c_void_p),
        ("y", C.c_void_p), ("y_bstride", C.c_int64), ("y_cstride", C.c_int64),
        ("stats", C.c_void_p), ("gn_stats", C.c_void_p), ("gn_w", C.c_void_p), ("gn_b", C.c_void_p),
        ("out_len", C.c_int32), ("tile_m", C.c_int32), ("plain", C.c_int32), ("half", C.c_int32),
        ("o2_valid", C.c_int32), ("ktab_len", C.c_int32), ("sink", C.c_void_p), ("wx", C.c_void_p), ("tr_stride", C.c_int32), ("tr_pad", C.c_int32), ("x_ld", C.c_int32), ("x_ld_pad", C.c_int32), ("wh", C.c_void_p),
        ("xh", C.c_void_p), ("xh_n", C.c_int64), ("yh", C.c_void_p), ("yh_n", C.c_int64),
        ("wtap", C.c_void_p), ("ntaps", C.c_int32), ("tap_k2", C.c_int32), ("tap_pad1", C.c_int32), ("tap_pad2", C.c_int32),
        ("tap_dil1", C.c_int32), ("tap_dil2", C.c_int32), ("yh_pq", C.c_int64),
        ("dma_rows", C.c_int32), ("dma_rows_pad", C.c_int32),
    ]


# symbol -> (restype, argtypes); every symbol declared in include/demucs_amd.h
SIGNATURES = {
    "mi_model_create": (C.c_int, [C.POINTER(MiConfig), C.POINTER(MiTensorDesc), C.c_size_t, C.POINTER(C.c_void_p)]),
    "mi_model_destroy": (None, [C.c_void_p]),
    "mi_model_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "mi_model_forward_core": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "mi_model_tap": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.c_void_p]),
    "mi_hmodel_create": (C.c_int, [C.POINTER(MiConfig), C.POINTER(MiTensorDesc), C.c_size_t, C.POINTER(C.c_void_p)]),
    "mi_hmodel_destroy": (None, [C.c_void_p]),
    "mi_hmodel_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "mi_hmodel_tap": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.c_void_p]),
    "mi_hmodel_device_bytes": (C.c_int64, [C.c_void_p]),
    "mi_hmodel_status": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mi_set_two_streams": (C.c_int, [C.c_int32]),
    "mi_set_istft_fused": (C.c_int, [C.c_int32]),
    "mi_set_transpose_tiles": (C.c_int, [C.c_int32]),
    "mi_profile_begin": (C.c_int, [C.c_void_p]),
    "mi_profile_end": (C.c_int, [C.c_void_p, C.POINTER(MiProfileRow), C.c_int32, C.POINTER(C.c_int32), C.c_void_p]),
    "mi_model_device_bytes": (C.c_int64, [C.c_void_p]),
    "mi_segments_gather": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                     C.c_int64, C.c_void_p]),
    "mi_ola_accumulate": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    "mi_ola_finish": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                                C.c_int32, C.c_void_p, C.c_void_p]),
    "mi_mono_stats_scratch_bytes": (C.c_int32, []),
    "mi_mono_stats": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi_track_affine": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    "mi_prevent_clip": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi_two_stems": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "mi_stft_cac": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mi_istft_cac": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mi_conv_forward": (C.c_int, [C.POINTER(MiConvDesc), C.c_void_p]),
    "mi_resample_frac": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                   C.c_void_p]),
    "mi_conv_pack_split": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mi_conv_pack_half": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mi_f32_to_image": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "mi_conv_pack_tap": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mi_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                               C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
    "mi_attention_heads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "mi_attention_image": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
    "mi_lstm_seq": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "mi_gn_gelu": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_void_p]),
    "mi_gram_order": (C.c_int32, [C.c_int32]),
    "mi_gn_gelu_gram": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "mi_gram_finalize": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                   C.c_double, C.c_double, C.c_float, C.c_void_p, C.c_void_p]),
    "mi_layernorm_cf": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "mi_debug_set_post_launch_hook": (None, [C.c_void_p]),
    "mi_debug_last_conv_route": (C.c_int, []),
    "mi_last_error": (C.c_char_p, []),
    "mi_version": (C.c_char_p, []),
}

_lib = None
_lock = threading.Lock()


class EngineError(RuntimeError):
    """A libdemucs_amd call returned a non-zero status (the reference convention for this path
    is "exceptions propagate", demucs/apply.py:289-293)."""


def load() -> C.CDLL:
    """dlopen the engine (once).  Raises EngineError if it is absent: no fallback exists."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise EngineError(
                    f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(hipcc --offload-arch=gfx950).  demucs_amd has no CPU / eager fallback.")
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)          # AttributeError if the symbol is not exported
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def check(status: int, what: str) -> None:
    if status != MI_OK:
        msg = load().mi_last_error().decode(errors="replace")
        raise EngineError(f"{what} failed ({status}): {msg}")


def current_stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
