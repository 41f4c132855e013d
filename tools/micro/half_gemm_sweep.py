"""Where does the bf16-operand linear GEMM spend its time?  Same M / N, K swept from one K step to the FFN's 2048, and
the epilogue varied (plain store, + residual, + GELU): separates the main loop from the epilogue.  GPU box:
    python tools/micro/half_gemm_sweep.py"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from demucs_amd import _lib
from gpu_helpers import EPI_LINEAR, FLAG_GELU, FLAG_RES, FLAG_SCALE, pack_w, ktab

B, P = 31, 2688
lib = _lib.load()
st = C.c_void_p(_lib.current_stream_ptr())


def run(M, K, flags, half):
    W = torch.randn(M, K) * 0.05
    wt, bias, M_, Mpad, K_, Kpad, tile = pack_w(W, torch.randn(M))
    x = torch.randn(B, K, P, device="cuda")
    y = torch.empty(B, M, P, device="cuda")
    res = torch.randn(B, M, P, device="cuda") if flags & FLAG_RES else None
    scale = torch.randn(Mpad, device="cuda")
    kt = ktab(K, 1, 1, 1, 1, 0, 0, P, P, Kpad)
    kw = dict(wt=wt, M=M, Mpad=Mpad, K=K, Kpad=Kpad, ktab=kt, ktab_len=kt.shape[0], x=x, x_bstride=K * P, B=B, D1=1, D2=P, O1=1, O2=P, S1=1, S2=1,
              epi=EPI_LINEAR, flags=flags, bias=bias, scale=scale, res=res, y=y, y_bstride=M * P, y_cstride=P, tile_m=tile, plain=1)
    if half:
        wh = torch.empty(2 * ((Kpad + 31) // 32 * 32) * Mpad, dtype=torch.uint8, device="cuda")
        _lib.check(lib.mi_conv_pack_half(wt.data_ptr(), Kpad, Mpad, half, wh.data_ptr(), st), "pack")
        kw.update(wh=wh, half=half)
    d = _lib.MiConvDesc()
    for f, _ in _lib.MiConvDesc._fields_:
        v = kw.get(f, 0)
        setattr(d, f, (v.data_ptr() if isinstance(v, torch.Tensor) else v) if v is not None else 0)
    fn = lambda: _lib.check(lib.mi_conv_forward(C.byref(d), st), "conv")
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    gb = 4.0 * B * P * (K + M * (2 if flags & FLAG_RES else 1)) / 1e9
    print(f"half={half} M={M:5d} K={K:5d} flags={flags:2d}: {ms * 1e3:8.1f} us  {2.0 * M * K * B * P / ms / 1e9:8.1f} TFLOP/s  {gb / ms * 1e3:7.1f} GB/s algorithmic", flush=True)


for half in (1, 0):
    for K in (32, 128, 512, 2048):
        run(1536, K, 0, half)
    run(512, 2048, FLAG_SCALE | FLAG_RES, half)
    run(2048, 512, FLAG_GELU, half)
    run(2048, 512, 0, half)
