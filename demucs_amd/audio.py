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
_CLIP_MODES = {"rescale": 1, "clamp": 2, "tanh": 3}


def _engine_device(*tensors) -> torch.device:
    """The GPU the kernels run on: the device of the first device tensor, else the current GPU (host tensors -- what
    `Separator.separate_tensor(host wav)` returns -- are staged through it).  There is no CPU implementation."""
    for t in tensors:
        if t is not None and t.device.type == "cuda":
            return t.device
    if not torch.cuda.is_available():
        raise _lib.EngineError("demucs_amd.audio: prevent_clip / two_stems run on the GPU (MI355X) and none is available; "
                               "there is no CPU implementation in this package.")
    return torch.device("cuda", torch.cuda.current_device())


def _stage(t: torch.Tensor, dev: torch.device, what: str) -> torch.Tensor:
    """`t` as a contiguous float32 tensor on `dev` (the reference accepts any floating tensor on any device)."""
    if not t.dtype.is_floating_point:
        raise TypeError(f"{what}: a floating-point tensor is expected, got {t.dtype}")
    return t.to(device=dev, dtype=torch.float32).contiguous()


def prevent_clip(wav: torch.Tensor, mode="rescale") -> torch.Tensor:
    """demucs/audio.py:218-234 as a device kernel (`mi_prevent_clip`: the peak of "rescale" is reduced on the device and
    never visits the host).  Device stems (an engine separation with `split=True` and a device mix) are processed where
    they live; host stems (what `Separator.separate_tensor(host wav)` returns) are staged H2D / D2H around the kernel;
    other floating dtypes are computed in float32 and cast back.  NaN samples propagate like torch's `abs().max()`."""
    if mode is None or mode == "none":
        return wav
    assert wav.dtype.is_floating_point, "too late for clipping"
    if mode not in _CLIP_MODES:
        raise ValueError(f"Invalid mode {mode}")
    dev = _engine_device(wav)
    x = _stage(wav, dev, "prevent_clip")
    y = torch.empty_like(x)
    if x.numel():
        with torch.cuda.device(dev):
            peak = torch.empty(1, dtype=torch.int32, device=dev)
            _lib.check(_lib.load().mi_prevent_clip(x.data_ptr(), x.numel(), _CLIP_MODES[mode], peak.data_ptr(), y.data_ptr(),
                                                   C.c_void_p(_lib.current_stream_ptr())), "mi_prevent_clip")
    return y.to(device=wav.device, dtype=wav.dtype)            # like the reference: same device and dtype as the input


def two_stems(origin: torch.Tensor, stems: dict, stem: str, other_method: str = "add") -> dict:
    """`--two-stems STEM` of demucs/separate.py:189-218 as a tensor function on the stems' device (`mi_two_stems`): returns
    {STEM: ..., "no_STEM": 0 + the other stems in dict order} for other_method="add", {"minus_STEM": origin - STEM, STEM: ...}
    for "minus", {STEM: ...} for "none"."""
    if stem not in stems:
        raise KeyError(f"stem {stem!r} is not in the separated sources {list(stems)}")
    if other_method not in ("add", "minus", "none"):
        raise ValueError(f"Invalid other_method {other_method}")
    names = list(stems)
    out = {}
    if other_method in ("add", "minus"):
        if len(names) > 8:
            raise ValueError("two_stems: at most 8 stems")
        dev = _engine_device(*stems.values())
        tensors = [_stage(stems[k], dev, "two_stems") for k in names]
        n = tensors[0].numel()
        assert all(t.numel() == n for t in tensors)
        y = torch.empty_like(tensors[0])
        minus = other_method == "minus"
        org = _stage(origin, dev, "two_stems") if minus else None
        assert org is None or org.numel() == n
        ptrs = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
        with torch.cuda.device(y.device):
            _lib.check(_lib.load().mi_two_stems(ptrs, len(tensors), names.index(stem), org.data_ptr() if minus else None, int(minus), n,
                                                y.data_ptr(), C.c_void_p(_lib.current_stream_ptr())), "mi_two_stems")
        y = y.to(device=stems[stem].device, dtype=stems[stem].dtype)
        if minus:
            out["minus_" + stem] = y
    out[stem] = stems[stem]
    if other_method == "add":
        out["no_" + stem] = y
    return out
