"""`demucs.audio.convert_audio` for the MI355X engine (demucs/audio.py:137-172): channel conversion on the
host tensor, fractional sinc resampling (`julius.resample_frac`) as a HIP kernel.

julius is a pinned dependency of the reference (requirements.txt: julius>=0.2.3) that is neither vendored in
/root/reference nor installable here, so its algorithm is restated from the published source (julius/resample.py,
ResampleFrac: zeros=24, rolloff=0.945, cos^2-windowed sinc bank of new_sr phases, replicate padding, stride old_sr):
PARITY UNPINNED — checked against `oracle/resample_oracle.py` (the same restatement with torch conv1d) and against
analytic properties (identity at equal rates, sine amplitude / frequency preservation), not against julius itself.
"""
import ctypes as C
import math
from functools import lru_cache

import torch

from . import _lib

ZEROS = 24
ROLLOFF = 0.945


def convert_audio_channels(wav: torch.Tensor, channels: int = 2) -> torch.Tensor:
    """demucs/audio.py:137-166."""
    *shape, src_channels, length = wav.shape
    if src_channels == channels:
        pass
    elif channels == 1:
        wav = wav.mean(dim=-2, keepdim=True)
    elif src_channels == 1:
        wav = wav.expand(*shape, channels, length)
    elif src_channels >= channels:
        wav = wav[..., :channels, :]
    else:
        raise ValueError('The audio file has less channels than requested but is not mono.')
    return wav


@lru_cache(maxsize=16)
def sinc_bank(old_sr: int, new_sr: int, zeros: int = ZEROS, rolloff: float = ROLLOFF):
    """(width, kernels (new_sr, 2*width + old_sr) float32) of julius.ResampleFrac._init_kernels for REDUCED rates."""
    sr = min(new_sr, old_sr) * rolloff
    width = math.ceil(zeros * old_sr / sr)
    idx = torch.arange(-width, width + old_sr).float()
    kernels = []
    for i in range(new_sr):
        t = (-i / new_sr + idx / old_sr) * sr
        t = t.clamp_(-zeros, zeros)
        t *= math.pi
        window = torch.cos(t / zeros / 2) ** 2
        kernel = torch.where(t == 0, torch.tensor(1.0), torch.sin(t) / t) * window
        kernel.div_(kernel.sum())
        kernels.append(kernel)
    return width, torch.stack(kernels).contiguous()


def resample_frac(x: torch.Tensor, old_sr: int, new_sr: int, device="cuda") -> torch.Tensor:
    """`julius.resample_frac(x, old_sr, new_sr)` (default output length floor(new_sr * L / old_sr)) computed on `device`
    by the HIP kernel; the result comes back on x.device.  There is no CPU implementation in this package."""
    gcd = math.gcd(int(old_sr), int(new_sr))
    old, new = int(old_sr) // gcd, int(new_sr) // gcd
    if old == new:
        return x
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.EngineError("demucs_amd.audio.resample_frac only runs on a GPU device (MI355X)")
    width, bank = sinc_bank(old, new)
    shape, length = x.shape, x.shape[-1]
    out_len = int(math.floor(new * length / old))
    xs = x.reshape(-1, length).to(dev, torch.float32).contiguous()
    y = torch.empty(xs.shape[0], out_len, device=dev, dtype=torch.float32)
    if out_len > 0:
        table = bank.to(dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().mi_resample_frac(xs.data_ptr(), xs.shape[0], length, table.data_ptr(), old, new, width, y.data_ptr(),
                                                    out_len, C.c_void_p(_lib.current_stream_ptr())), "mi_resample_frac")
    return y.reshape(list(shape[:-1]) + [out_len]).to(x.device)


def convert_audio(wav: torch.Tensor, from_samplerate: int, to_samplerate: int, channels: int, device="cuda") -> torch.Tensor:
    """demucs/audio.py:169-172."""
    wav = convert_audio_channels(wav, channels)
    return resample_frac(wav, from_samplerate, to_samplerate, device)


# ---- after the separation: what demucs.separate does with the stems before any encoder sees them ----------------------
def prevent_clip(wav: torch.Tensor, mode="rescale") -> torch.Tensor:
    """demucs/audio.py:218-234, on whatever device the stems live (they stay in HBM with split=True on the engine)."""
    if mode is None or mode == "none":
        return wav
    assert wav.dtype.is_floating_point, "too late for clipping"
    if mode == "rescale":
        return wav / max(1.01 * wav.abs().max(), 1)
    if mode == "clamp":
        return wav.clamp(-0.99, 0.99)
    if mode == "tanh":
        return torch.tanh(wav)
    raise ValueError(f"Invalid mode {mode}")


def two_stems(origin: torch.Tensor, stems: dict, stem: str, other_method: str = "add") -> dict:
    """`--two-stems STEM` of demucs/separate.py:189-218 as a tensor function: returns {STEM: ..., "no_STEM": sum of the
    other stems} for other_method="add", {STEM: ..., "minus_STEM": origin - STEM} for "minus", {STEM: ...} for "none".
    The sum runs in dict order from zeros, like the reference."""
    if stem not in stems:
        raise KeyError(f"stem {stem!r} is not in the separated sources {list(stems)}")
    res = dict(stems)
    out = {}
    if other_method == "minus":
        out["minus_" + stem] = origin - res[stem]
    out[stem] = res.pop(stem)
    if other_method == "add":
        other = torch.zeros_like(next(iter(res.values())))
        for v in res.values():
            other += v
        out["no_" + stem] = other
    elif other_method not in ("minus", "none"):
        raise ValueError(f"Invalid other_method {other_method}")
    return out
