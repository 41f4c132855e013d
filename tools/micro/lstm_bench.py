"""The persistent LSTM recurrence kernel alone (mi_lstm_seq mode 1) at the sizes of hdemucs_mmi's 44-second chunks and tail
chunk; run under `rocprofv3 --kernel-trace --stats` for kernel durations, MI_LSTM_DEBUG=1 prints the time block 0 spends polling."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from demucs_amd import _lib  # noqa: E402

lib = _lib.load()
for H, N in ((384, 50), (192, 95), (384, 4), (192, 7)):
    W = 200
    gen = torch.Generator().manual_seed(1)
    gx = torch.randn(N, 2, 4 * H, W, generator=gen).cuda()
    whh = np.ascontiguousarray((torch.randn(2, 4 * H, H, generator=gen) * (1.5 / H ** 0.5)).numpy())
    out = torch.empty(N, 2 * H, W, device="cuda")
    for rep in range(3):
        _lib.check(lib.mi_lstm_seq(gx.data_ptr(), whh.ctypes.data, N, H, W, out.data_ptr(), 1, C.c_void_p(_lib.current_stream_ptr())), "mi_lstm_seq")
    print(H, N, float(out.abs().mean()), flush=True)
