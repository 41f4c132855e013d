"""`Separator`: the tensor-level face of the reference's public API (reference: demucs/api.py:53-319).

Only what sits on the tensor -> tensor path is provided: construction around an already built
model (the model zoo download of `demucs/pretrained.py` needs the network and is out of scope),
`update_parameter`, `separate_tensor` (api.py:241-291), the `samplerate / audio_channels / model`
properties and `list_models` for a local folder (api.py:322-347).  Audio file loading and the stem writers are not part of this path.

`separate_tensor` is a device path: ONE host -> device copy of the raw `wav`, the resampler kernel when `sr`
differs (`convert_audio`, demucs_amd/audio.py), the mono mean / unbiased std by a device reduction
(`mi_mono_stats`; the two scalars never visit the host), `(x - mean) / (std + 1e-8)` in place on the device
copy (`mi_track_affine`), `apply_model` on device-resident tensors, `x * std + mean` in place on the stems,
ONE device -> host copy of the stems into a pinned tensor.  The caller's host `wav` is never touched (the
reference normalises it in place and restores it, which leaves rounding differences of ~1e-7 behind); a device
`wav` is normalised and restored in place exactly as the reference does.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Callable, Dict, Optional, Tuple, Union

import torch

from . import _lib
from .audio import convert_audio
from .apply import BagOfModels, _is_engine, _to_host, apply_model

__all__ = ["Separator", "LoadModelError", "list_models"]


class LoadModelError(Exception):
    pass


def _with(d: Optional[dict], **subs) -> dict:
    out = dict(d) if d is not None else {}
    out.update(subs)
    return out


class Separator:
    def __init__(self, model, repo=None, device="cuda", shifts: int = 1, overlap: float = 0.25, split: bool = True,
                 segment: Optional[int] = None, jobs: int = 0, progress: bool = False,
                 callback: Optional[Callable[[dict], None]] = None, callback_arg: Optional[dict] = None):
        if isinstance(model, str):
            # api.py:99-104: `Separator(model=name, repo=folder)`; offline only "demucs_unittest" and local folders resolve
            from .pretrained import get_model
            from .states import ModelLoadingError
            try:
                model = get_model(model, repo)
            except ModelLoadingError as exc:
                raise LoadModelError(str(exc)) from exc
        self._model = model
        self._audio_channels = model.audio_channels
        self._samplerate = model.samplerate
        self.update_parameter(device=device, shifts=shifts, overlap=overlap, split=split, segment=segment, jobs=jobs,
                              progress=progress, callback=callback, callback_arg=callback_arg)

    def update_parameter(self, device=None, shifts=None, overlap=None, split=None, segment=-1, jobs=None, progress=None,
                         callback=-1, callback_arg=-1):
        """api.py:124-201: only the given parameters change (segment / callback use a sentinel because
        None is a legal value)."""
        if device is not None: self._device = device
        if shifts is not None: self._shifts = shifts
        if overlap is not None: self._overlap = overlap
        if split is not None: self._split = split
        if segment != -1:
            if segment is not None and segment <= 0:
                raise ValueError("segment must be greater than 0")       # api.py:166-170
            self._segment = segment
        if jobs is not None: self._jobs = jobs
        if progress is not None: self._progress = progress
        if callback != -1: self._callback = callback
        if callback_arg != -1: self._callback_arg = callback_arg

    def separate_tensor(self, wav: torch.Tensor, sr: Optional[int] = None) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """api.py:241-291.  `wav` (channels, length) float32 is normalised IN PLACE by the mono
        mean / std for the duration of the call and restored before returning."""
        if _is_engine(self._model) and torch.device(self._device).type == "cuda":
            return self._separate_on_device(wav, sr)
        if sr is not None and sr != self._samplerate:
            wav = convert_audio(wav, sr, self._samplerate, self._audio_channels, device=self._device)
        ref = wav.mean(0)
        wav -= ref.mean()
        wav /= ref.std() + 1e-8
        out = apply_model(self._model, wav[None], segment=self._segment, shifts=self._shifts, split=self._split,
                          overlap=self._overlap, device=self._device, num_workers=self._jobs, callback=self._callback,
                          callback_arg=_with(self._callback_arg, audio_length=wav.shape[1]), progress=self._progress)
        if out is None:
            raise KeyboardInterrupt
        out = out.to(wav.device)
        out *= ref.std() + 1e-8
        out += ref.mean()
        wav *= ref.std() + 1e-8
        wav += ref.mean()
        return wav, dict(zip(self._model.sources, out[0]))

    def _separate_on_device(self, wav: torch.Tensor, sr: Optional[int]):
        """The engine's `separate_tensor` (module docstring): everything between the one H2D and the one D2H runs on the GPU."""
        device = torch.device(self._device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        lib = _lib.load()
        host_in = wav.device.type == "cpu"
        with torch.cuda.device(device):
            stream = lambda: C.c_void_p(_lib.current_stream_ptr())          # noqa: E731
            dev = wav.to(device=device, dtype=torch.float32, non_blocking=True) if host_in else wav
            if sr is not None and sr != self._samplerate:
                dev = convert_audio(dev, sr, self._samplerate, self._audio_channels, device=device)
                if host_in:
                    wav = None                    # the reference returns the converted tensor: handed back from the device below
            if not dev.is_contiguous() or dev.dtype != torch.float32:
                dev = dev.contiguous().float()
            channels, length = dev.shape
            scratch = torch.empty(lib.mi_mono_stats_scratch_bytes(), dtype=torch.uint8, device=device)
            stats = torch.empty(2, dtype=torch.float32, device=device)
            _lib.check(lib.mi_mono_stats(dev.data_ptr(), channels, length, scratch.data_ptr(), stats.data_ptr(), stream()),
                       "mi_mono_stats")
            restore = dev.clone() if (host_in and wav is None) else None       # resampled host input: returned un-normalised
            _lib.check(lib.mi_track_affine(dev.data_ptr(), dev.numel(), stats.data_ptr(), 0, stream()), "mi_track_affine")
            out = apply_model(self._model, dev[None], segment=self._segment, shifts=self._shifts, split=self._split,
                              overlap=self._overlap, device=device, num_workers=self._jobs, callback=self._callback,
                              callback_arg=_with(self._callback_arg, audio_length=length), progress=self._progress)
            if out is None:
                raise KeyboardInterrupt
            out = out.contiguous()
            _lib.check(lib.mi_track_affine(out.data_ptr(), out.numel(), stats.data_ptr(), 1, stream()), "mi_track_affine")
            if host_in:
                stems = _to_host(out, device)
                if wav is None:
                    wav = _to_host(restore, device)
            else:
                _lib.check(lib.mi_track_affine(dev.data_ptr(), dev.numel(), stats.data_ptr(), 1, stream()), "mi_track_affine")
                stems, wav = out, dev
        return wav, dict(zip(self._model.sources, stems[0]))

    @property
    def samplerate(self):
        return self._samplerate

    @property
    def audio_channels(self):
        return self._audio_channels

    @property
    def model(self):
        return self._model


def list_models(repo: Optional[Union[str, Path]] = None) -> Dict[str, Dict[str, Union[str, Path]]]:
    """api.py:322-347: `{"single": {signature: package path}, "bag": {name: yaml path}}` of a local model folder.  Without `repo`
    the reference lists its remote zoo (`remote/files.txt` + the bag YAMLs it ships); offline only `demucs_unittest`, the one
    model that needs no download (pretrained.py:27-29), can be named."""
    if repo is None:
        return {"single": {"demucs_unittest": "built in (HDemucs(channels=4), pretrained.py:27-29)"}, "bag": {}}
    from .states import LocalRepo
    repo = Path(repo)
    if not repo.is_dir():
        raise LoadModelError(f"{repo} must exist and be a directory.")
    local = LocalRepo(repo)
    return {"single": dict(local._models), "bag": dict(local._bags)}
