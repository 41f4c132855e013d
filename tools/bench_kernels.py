"""Micro-benchmarks of the engine's hot kernels at the shapes the htdemucs forward uses
(batch of 8 segments).  Run on the GPU box:  python tools/bench_kernels.py [filter]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from demucs_amd import _lib                                            # noqa: E402
from gpu_helpers import EPI_GLU, EPI_LINEAR, FLAG_GELU, conv_call, ktab, pack_w   # noqa: E402

B = int(os.environ.get("BENCH_B", "8"))


def time_fn(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def conv_case(name, M, Cin, K1, K2, D1, D2, pad1=0, pad2=0, glu=False, tile=0):
    K = Cin * K1 * K2
    W = torch.randn(M, K) * 0.05
    wt, bias, M_, Mpad, K_, Kpad, tile_ = pack_w(W, torch.randn(M), glu=glu, tile=tile or None)
    kt = ktab(Cin, K1, K2, 1, 1, pad1, pad2, D1 * D2, D2, Kpad)
    x = torch.randn(B, Cin, D1, D2, device="cuda")
    P = D1 * D2
    Mo = M // 2 if glu else M
    y = torch.empty(B, Mo, P, device="cuda")
    d = _lib.MiConvDesc()
    kw = dict(wt=wt, M=M, Mpad=Mpad, K=K, Kpad=Kpad, ktab=kt, x=x, x_bstride=Cin * P, B=B, D1=D1, D2=D2, O1=D1, O2=D2, S1=1, S2=1,
              row_mode=1 if D1 > 1 else 0, epi=EPI_GLU if glu else EPI_LINEAR, flags=0, bias=bias, y=y, y_bstride=Mo * P, y_cstride=P,
              tile_m=tile_, plain=1 if (K1 == 1 and K2 == 1) else 0)
    for f, _ in _lib.MiConvDesc._fields_:
        v = kw.get(f, 0)
        setattr(d, f, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    lib = _lib.load()
    st = C.c_void_p(_lib.current_stream_ptr())
    if os.environ.get("BENCH_X6", "1") != "0" and tile_ in (64, 96, 128):
        wx = torch.empty(6 * Kpad * Mpad, dtype=torch.uint8, device="cuda")
        _lib.check(lib.mi_conv_pack_split(wt.data_ptr(), Kpad, Mpad, tile_, wx.data_ptr(), st), "pack")
        d.wx = wx.data_ptr()
    ms = time_fn(lambda: _lib.check(lib.mi_conv_forward(C.byref(d), st), "conv"))
    flops = 2.0 * M * K * B * P
    print(f"{name:34s} M={M:5d} K={K:5d} N={B * P:7d} tile={tile_:3d}  {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TFLOP/s", flush=True)


def attn_case(name, Tq, Tk):
    lib = _lib.load()
    q = torch.randn(B, 512, Tq, device="cuda")
    kv = torch.randn(B, 1024, Tk, device="cuda")
    o = torch.empty(B, 512, Tq, device="cuda")
    st = C.c_void_p(_lib.current_stream_ptr())
    ms = time_fn(lambda: _lib.check(lib.mi_attention(q.data_ptr(), kv.data_ptr(), kv.data_ptr() + 512 * Tk * 4, o.data_ptr(), B, 8, Tq,
                                                     Tk, 512 * Tq, 1024 * Tk, 512 * Tq, st), "attn"))
    flops = 4.0 * B * 8 * Tq * Tk * 64
    print(f"{name:34s} Tq={Tq} Tk={Tk}  {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TFLOP/s", flush=True)


CASES = {
    "out_proj_f": lambda: conv_case("out_proj freq 512x512", 512, 512, 1, 1, 1, 2688),
    "ffn2_f": lambda: conv_case("ffn2 freq 512x2048", 512, 2048, 1, 1, 1, 2688),
    "qkv_f": lambda: conv_case("qkv freq 1536x512", 1536, 512, 1, 1, 1, 2688),
    "ffn1_f": lambda: conv_case("ffn1 freq 2048x512", 2048, 512, 1, 1, 1, 2688),
    "ffn1_t": lambda: conv_case("ffn1 time 2048x512", 2048, 512, 1, 1, 1, 1344),
    "out_proj_t": lambda: conv_case("out_proj time 512x512", 512, 512, 1, 1, 1, 1344),
    "dec0_rw": lambda: conv_case("dec0 rewrite 3x3 768x3456", 768, 384, 3, 3, 8, 336, 1, 1, glu=True),
    "dec1_rw": lambda: conv_case("dec1 rewrite 3x3 384x1728", 384, 192, 3, 3, 32, 336, 1, 1, glu=True),
    "dec2_rw": lambda: conv_case("dec2 rewrite 3x3 192x864", 192, 96, 3, 3, 128, 336, 1, 1, glu=True),
    "dec3_rw": lambda: conv_case("dec3 rewrite 3x3 96x432", 96, 48, 3, 3, 512, 336, 1, 1, glu=True),
    "enc0_rw": lambda: conv_case("enc0 rewrite 1x1 96x48", 96, 48, 1, 1, 512, 336, glu=True),
    "attn_ff": lambda: attn_case("attention self freq", 2688, 2688),
    "attn_tt": lambda: attn_case("attention self time", 1344, 1344),
    "attn_ft": lambda: attn_case("attention cross f<-t", 2688, 1344),
}

if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    torch.manual_seed(0)
    for k, fn in CASES.items():
        if pat in k:
            fn()
