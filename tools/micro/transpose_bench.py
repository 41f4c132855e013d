"""The layout changes around the transforms at the bench's size (31 segments, 4 sources): strip kernels (default) vs the 32 x 32
tile kernels; run under `rocprofv3 --kernel-trace --stats` for the kernel durations (cac_transpose[_strip]_kernel,
spec_transpose[_strip]_kernel)."""
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
mix = torch.randn(B, 2, L, device="cuda")
spec = torch.empty(B, 4, 2048, 336, device="cuda")
out = torch.empty(B, S, 2, L, device="cuda")
st = C.c_void_p(_lib.current_stream_ptr())
for tiles in (0, 1, 0, 1):     # 1 = all three round-3 kernels (2 / 4 select the cac / spec tile kernels alone)
    lib.mi_set_transpose_tiles(tiles)
    for _ in range(3):
        _lib.check(lib.mi_istft_cac(x.data_ptr(), B, S, L, out.data_ptr(), st), "mi_istft_cac")
        _lib.check(lib.mi_stft_cac(mix.data_ptr(), B, L, spec.data_ptr(), st), "mi_stft_cac")
print("ok", float(out.abs().mean()), float(spec.abs().mean()))
