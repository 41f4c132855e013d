"""TEST INFRASTRUCTURE ONLY (imported by tests/, never by the product path).

CPU restatement of `julius.resample_frac` as used by demucs/audio.py:169-172.  julius (requirements.txt:
julius>=0.2.3) is not vendored in the reference and not importable here: PARITY UNPINNED — restated from the
published julius/resample.py (ResampleFrac.__init__/_init_kernels/forward) in float64-capable torch ops.
"""
import math

import torch
import torch.nn.functional as F


def resample_frac(x: torch.Tensor, old_sr: int, new_sr: int, zeros: int = 24, rolloff: float = 0.945, dtype=torch.float32):
    gcd = math.gcd(old_sr, new_sr)
    old_sr, new_sr = old_sr // gcd, new_sr // gcd
    if old_sr == new_sr:
        return x
    sr = min(new_sr, old_sr) * rolloff
    width = math.ceil(zeros * old_sr / sr)
    idx = torch.arange(-width, width + old_sr).to(dtype)
    kernels = []
    for i in range(new_sr):
        t = (-i / new_sr + idx / old_sr) * sr
        t = t.clamp(-zeros, zeros) * math.pi
        window = torch.cos(t / zeros / 2) ** 2
        kernel = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / t) * window
        kernels.append(kernel / kernel.sum())
    kernel = torch.stack(kernels).view(new_sr, 1, -1)
    shape, length = x.shape, x.shape[-1]
    xs = x.reshape(-1, length).to(dtype)
    xs = F.pad(xs[:, None], (width, width + old_sr), mode="replicate")
    ys = F.conv1d(xs, kernel, stride=old_sr)
    y = ys.transpose(1, 2).reshape(list(shape[:-1]) + [-1])
    return y[..., :int(math.floor(new_sr * length / old_sr))]
