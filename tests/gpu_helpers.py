"""Helpers for the GPU parity tests: pack weights / gather tables for `mi_conv_forward` in Python
(an independent restatement of the packing done inside mi_model_create) and call the C ABI."""
import ctypes as C

import numpy as np
import torch

from demucs_amd import _lib

EPI_LINEAR, EPI_GLU, EPI_BIAS_STATS, EPI_STATS_ONLY, EPI_GN_GLU, EPI_CONVTR = range(6)
FLAG_GELU, FLAG_SCALE, FLAG_RES, FLAG_EMB, FLAG_TR_FREQ = 1, 2, 4, 8, 16
SLOTS = 32


def pick_tile(M):
    if M <= 32: return 32
    if M <= 64: return 64
    if M % 128 == 0: return 128
    if M % 96 == 0: return 96
    if M <= 96: return 96
    return 128


def rup(v, m):
    return (v + m - 1) // m * m


def glu_perm(M):
    """packed row m -> source row: (a_0, g_0, a_1, g_1, ...)"""
    return np.array([(m >> 1) + (M // 2 if m & 1 else 0) for m in range(M)])


def pack_w(W2d: torch.Tensor, bias, glu=False, tile=None):
    """W2d (M, K) -> device Wt (Kpad, Mpad), bias (Mpad)"""
    M, K = W2d.shape
    tile = tile or pick_tile(M)
    Mpad, Kpad = rup(M, tile), rup(K, 16)
    perm = glu_perm(M) if glu else np.arange(M)
    wt = torch.zeros(Kpad, Mpad, dtype=torch.float32)
    wt[:K, :M] = W2d[perm].t().float()
    b = torch.zeros(Mpad, dtype=torch.float32)
    if bias is not None:
        b[:M] = bias[perm].float()
    return wt.cuda(), b.cuda(), M, Mpad, K, Kpad, tile


def pack_vec(v, Mpad, glu=False):
    M = v.numel()
    perm = glu_perm(M) if glu else np.arange(M)
    out = torch.zeros(Mpad, dtype=torch.float32)
    out[:M] = v[perm].float()
    return out.cuda()


def ktab(Cin, K1, K2, dil1, dil2, pad1, pad2, chan_stride, D2, Kpad):
    Kpad = rup(Kpad, 32)          # the bf16 / fp16 main loop steps K by 32: entries past K are "never valid"
    t = np.zeros((Kpad, 4), dtype=np.int32)
    K = Cin * K1 * K2
    for k in range(Kpad):
        if k < K:
            ci, r = divmod(k, K1 * K2)
            k1, k2 = divmod(r, K2)
            d1, d2 = k1 * dil1 - pad1, k2 * dil2 - pad2
            t[k] = (ci * chan_stride + d1 * D2 + d2, d1, d2, ci)
        else:
            t[k] = (0, -(1 << 29), -(1 << 29), 0)
    return torch.from_numpy(t).cuda()


def conv_call(**kw):
    """Fill a MiConvDesc from keyword arguments (tensors -> data_ptr) and launch."""
    d = _lib.MiConvDesc()
    keep = []
    mode = kw.pop("x6", False)
    if mode in ("bf16", "f16"):
        # reduced-precision compute mode: bf16 / fp16 operand image of the same packed weights (gemm_half.hip)
        dt = {"bf16": 1, "f16": 2}[mode]
        wh = torch.empty(2 * rup(kw["Kpad"], 32) * kw["Mpad"], dtype=torch.uint8, device="cuda")
        _lib.check(_lib.load().mi_conv_pack_half(kw["wt"].data_ptr(), kw["Kpad"], kw["Mpad"], dt, wh.data_ptr(),
                                                 C.c_void_p(_lib.current_stream_ptr())), "mi_conv_pack_half")
        kw["wh"], kw["half"] = wh, dt
        kw["ktab_len"] = kw["ktab"].shape[0] if isinstance(kw.get("ktab"), torch.Tensor) else 0
        mode = False
    if mode and kw.get("tile_m") in (64, 96, 128):
        # split-bf16 image of the same packed weights: selects the 6-product bf16 MFMA main loop
        wx = torch.empty(6 * kw["Kpad"] * kw["Mpad"], dtype=torch.uint8, device="cuda")
        _lib.check(_lib.load().mi_conv_pack_split(kw["wt"].data_ptr(), kw["Kpad"], kw["Mpad"], kw["tile_m"], wx.data_ptr(),
                                                  C.c_void_p(_lib.current_stream_ptr())), "mi_conv_pack_split")
        kw["wx"] = wx
    for name, _ in _lib.MiConvDesc._fields_:
        v = kw.get(name, 0)
        if isinstance(v, torch.Tensor):
            assert v.is_cuda and v.is_contiguous(), name
            keep.append(v)
            v = v.data_ptr()
        setattr(d, name, v if v is not None else 0)
    _lib.check(_lib.load().mi_conv_forward(C.byref(d), C.c_void_p(_lib.current_stream_ptr())), "mi_conv_forward")
    torch.cuda.synchronize()
    return keep


def maxerr(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item()
