"""Microbenchmark of attention_heads.hip on the transformer's shapes (31 segments x 8 heads; Tq / Tk = 2688 / 1344)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from demucs_amd import _lib

lib = _lib.load()
B, H = 31, 8
for dtype, hdt in ((1, torch.bfloat16), (2, torch.float16)):
    for Tq, Tk in ((2688, 2688), (1344, 1344), (2688, 1344), (1344, 2688)):
        q = torch.randn(B, H, Tq, 64, device="cuda").to(hdt)
        k = torch.randn(B, H, Tk, 64, device="cuda").to(hdt)
        v = torch.randn(B, H, Tk, 64, device="cuda").to(hdt)
        o = torch.empty(B, 512, Tq, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(3):
            _lib.check(lib.mi_attention_heads(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), B, H, Tq, Tk, Tq, Tk, dtype, st), "att")
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        n = 10
        for _ in range(n):
            lib.mi_attention_heads(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), B, H, Tq, Tk, Tq, Tk, dtype, st)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / n
        fl = 4.0 * B * H * Tq * Tk * 64
        print(f"dtype {dtype} Tq {Tq} Tk {Tk}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
