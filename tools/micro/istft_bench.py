"""Inverse STFT at the bench's size (31 segments x 4 sources): fused kernel vs the two separate kernels; run under
`rocprofv3 --kernel-trace --stats` for the kernel durations."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from demucs_amd import _lib  # noqa: E402

lib = _lib.load()
B, S, L = 31, 4, 343980
x = torch.randn(B, S * 4, 2048, 336, device="cuda")
out = torch.empty(B, S, 2, L, device="cuda")
for fused in (1, 0, 1, 0):
    lib.mi_set_istft_fused(fused)
    for _ in range(3):
        _lib.check(lib.mi_istft_cac(x.data_ptr(), B, S, L, out.data_ptr(), C.c_void_p(_lib.current_stream_ptr())), "mi_istft_cac")
print("ok", float(out.abs().mean()))
