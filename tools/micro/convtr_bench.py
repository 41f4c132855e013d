"""Transposed convs of the decoders in a half mode: table-driven route against the operand-image (two-tap LDS-DMA) route, at the
model's shapes.  python tools/micro/convtr_bench.py [bf16|f16] [B]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from demucs_amd import _lib  # noqa: E402
from gpu_helpers import EPI_CONVTR, FLAG_GELU, FLAG_RES, FLAG_TR_FREQ, ktab, pack_w, rup  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 31
dt = {"bf16": 1, "f16": 2}[mode]
lib = _lib.load()
st = C.c_void_p(_lib.current_stream_ptr())


def desc(**kw):
    d = _lib.MiConvDesc()
    keep = []
    for name, _ in _lib.MiConvDesc._fields_:
        v = kw.get(name, 0)
        if isinstance(v, torch.Tensor):
            keep.append(v)
            v = v.data_ptr()
        setattr(d, name, v if v is not None else 0)
    return d, keep


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def case(freq, Cin, Co, Fr, T, last):
    g = torch.Generator().manual_seed(1)
    W2 = torch.randn(4 * Co, 2 * Cin, generator=g) * 0.05
    wt, bias, M, Mpad, K, Kpad, tile = pack_w(W2, torch.zeros(4 * Co))
    wh = torch.empty(2 * rup(Kpad, 32) * Mpad, dtype=torch.uint8, device="cuda")
    _lib.check(lib.mi_conv_pack_half(wt.data_ptr(), Kpad, Mpad, dt, wh.data_ptr(), st), "pack_half")
    pairs = (Cin // 8 * 2 + 3) // 4 * 4
    wtap = torch.empty(pairs * Mpad * 8, dtype=torch.int16, device="cuda")
    _lib.check(lib.mi_conv_pack_tap(wt.data_ptr(), Mpad, Cin, 2, dt, wtap.data_ptr(), st), "pack_tap")
    if freq:
        P, Pout = Fr * T, 4 * Fr * T
        x = torch.randn(B, Cin, P, device="cuda")
        common = dict(wt=wt, wh=wh, half=dt, M=M, Mpad=Mpad, K=K, Kpad=Kpad, x=x, x_bstride=Cin * P, B=B, D1=Fr, D2=T, O1=Fr + 1, O2=T,
                      S1=1, S2=1, row_mode=1, epi=EPI_CONVTR, flags=FLAG_TR_FREQ | (0 if last else FLAG_GELU | FLAG_RES), bias=bias,
                      y_bstride=Co * Pout, y_cstride=Pout, out_len=4 * Fr, tile_m=tile)
        kt = ktab(Cin, 2, 1, -1, 1, 0, 0, P, T, Kpad)
        tap = dict(tap_k2=1, tap_dil1=-1)
    else:
        L = T
        P, Pout = rup(L, 4), rup(4 * L, 4)
        x = torch.randn(B, Cin, P, device="cuda")
        common = dict(wt=wt, wh=wh, half=dt, M=M, Mpad=Mpad, K=K, Kpad=Kpad, x=x, x_bstride=Cin * P, B=B, D1=1, D2=L, O1=1, O2=L + 1,
                      S1=1, S2=1, epi=EPI_CONVTR, flags=(0 if last else FLAG_GELU | FLAG_RES), bias=bias, y_bstride=Co * Pout,
                      y_cstride=Pout, out_len=4 * L, tile_m=tile, x_ld=P)
        kt = ktab(Cin, 1, 2, 1, -1, 0, 0, P, P, Kpad)
        tap = dict(tap_k2=2, tap_dil2=-1)
    y = torch.empty(B, Co, Pout, device="cuda")
    res = None if last else torch.randn(B, Co, Pout, device="cuda")
    img = torch.empty(Cin // 8, B * P, 8, dtype=torch.int16, device="cuda")
    d0, k0 = desc(y=y, res=res, ktab=kt, ktab_len=kt.shape[0], **common)
    d1, k1 = desc(y=y, res=res, ktab=kt, ktab_len=kt.shape[0], xh=img, xh_n=B * P, wtap=wtap, ntaps=2, **tap, **common)
    t_g = timeit(lambda: _lib.check(lib.mi_conv_forward(C.byref(d0), st), "conv"))
    t_c = timeit(lambda: _lib.check(lib.mi_f32_to_image(x.data_ptr(), B, Cin, P, dt, img.data_ptr(), st), "img"))
    t_t = timeit(lambda: _lib.check(lib.mi_conv_forward(C.byref(d1), st), "conv"))
    cols = B * ((Fr + 1) * T if freq else T + 1)
    fl = 2.0 * M * K * cols
    by = 4.0 * x.numel() + 4.0 * y.numel() * (1 if last else 2)
    print(f"{'freq' if freq else 'time'} {Cin:4d}->{Co:4d} Fr {Fr:4d} T {T:6d} tile {tile:3d}: table {t_g:8.1f} us ({fl / t_g / 1e6:6.1f} TF/s, "
          f"{by / t_g / 1e3:6.1f} GB/s) | image pass {t_c:7.1f} us + taps {t_t:8.1f} us ({fl / t_t / 1e6:6.1f} TF/s, {by / t_t / 1e3:6.1f} GB/s)",
          flush=True)


for j, (Cin, Co) in enumerate([(384, 192), (192, 96), (96, 48), (48, 16)]):
    case(True, Cin, Co, [8, 32, 128, 512][j], 336, j == 3)
for j, (Cin, Co) in enumerate([(384, 192), (192, 96), (96, 48), (48, 8)]):
    case(False, Cin, Co, 1, [1344, 5375, 21499, 85995][j], j == 3)
