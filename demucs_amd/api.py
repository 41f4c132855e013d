"""`Separator`: the tensor-level face of the reference's public API (reference: demucs/api.py:53-319).

Only what sits on the tensor -> tensor path is provided: construction around an already built
model (the model zoo download of `demucs/pretrained.py` needs the network and is out of scope),
`update_parameter`, `separate_tensor` with the reference's in-place normalise / restore contract
(api.py:265-291), and the `samplerate / audio_channels / model` properties.  Audio file loading,
resampling (`convert_audio`, julius) and stem writers are not part of this path.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch

from .audio import convert_audio
from .apply import BagOfModels, apply_model

__all__ = ["Separator", "LoadModelError"]


class LoadModelError(Exception):
    pass


def _with(d: Optional[dict], **subs) -> dict:
    out = dict(d) if d is not None else {}
    out.update(subs)
    return out


class Separator:
    def __init__(self, model, device="cuda", shifts: int = 1, overlap: float = 0.25, split: bool = True,
                 segment: Optional[int] = None, jobs: int = 0, progress: bool = False,
                 callback: Optional[Callable[[dict], None]] = None, callback_arg: Optional[dict] = None):
        if isinstance(model, str):
            raise LoadModelError(f"model zoo entry {model!r} cannot be fetched offline: pass a demucs_amd.HTDemucs or "
                                 "BagOfModels built from a locally loaded state dict")
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

    @property
    def samplerate(self):
        return self._samplerate

    @property
    def audio_channels(self):
        return self._audio_channels

    @property
    def model(self):
        return self._model
