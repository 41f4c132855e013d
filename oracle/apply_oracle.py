"""CPU ORACLE (test infrastructure, NOT product code): restatement of the reference's segment
scheduler `demucs.apply.apply_model` (reference: demucs/apply.py:29-322, demucs/utils.py:38-54).

Plain sequential Python over torch CPU tensors: bag loop -> shift loop -> overlapping-segment
loop -> padded leaf forward + centre trim, with the reference's callback events and its use of
Python's global `random` for the shift offsets.  Pinned by tests/golden/apply_*.npz (outputs of
the imported reference, see tools/make_golden.py).
"""
from __future__ import annotations

import random
from typing import Callable, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


class Window:
    """A (offset, length) view over the last axis of a tensor; `padded(n)` returns n samples
    centred on the view, filled with the underlying tensor's real neighbours where they exist
    and zeros elsewhere (TensorChunk, apply.py:82-124)."""

    def __init__(self, tensor, offset: int = 0, length: Optional[int] = None):
        total = tensor.shape[-1]
        assert 0 <= offset < total
        length = total - offset if length is None else min(total - offset, length)
        if isinstance(tensor, Window):
            self.tensor, self.offset = tensor.tensor, offset + tensor.offset
        else:
            self.tensor, self.offset = tensor, offset
        self.length = length

    @property
    def shape(self):
        s = list(self.tensor.shape)
        s[-1] = self.length
        return s

    def padded(self, target: int) -> Tensor:
        delta = target - self.length
        assert delta >= 0
        total = self.tensor.shape[-1]
        start = self.offset - delta // 2
        end = start + target
        cs, ce = max(0, start), min(total, end)
        return F.pad(self.tensor[..., cs:ce], (cs - start, end - ce))


def center_trim(t: Tensor, size: int) -> Tensor:
    """utils.py:38-54: odd remainder is removed on the right."""
    delta = t.shape[-1] - size
    if delta < 0:
        raise ValueError(f"tensor must be larger than reference. Delta is {delta}.")
    if delta:
        t = t[..., delta // 2:-(delta - delta // 2)]
    return t


class Bag:
    """BagOfModels container (apply.py:29-79)."""

    def __init__(self, models: Sequence, weights: Optional[List[List[float]]] = None):
        self.models = list(models)
        first = self.models[0]
        self.sources, self.samplerate, self.audio_channels = first.sources, first.samplerate, first.audio_channels
        self.weights = weights if weights is not None else [[1.0] * len(first.sources) for _ in self.models]


def transition_weight(segment_length: int, transition_power: float, dtype=torch.float32) -> Tensor:
    """Triangular cross-fade weight (apply.py:271-276): integer ramps divided by their max in
    float32, then raised to transition_power."""
    w = torch.cat([torch.arange(1, segment_length // 2 + 1),
                   torch.arange(segment_length - segment_length // 2, 0, -1)])
    return (w / w.max()) ** transition_power


def apply_model(model, mix, shifts: int = 1, split: bool = True, overlap: float = 0.25,
                transition_power: float = 1.0, segment=None,
                callback: Optional[Callable[[dict], None]] = None, callback_arg: Optional[dict] = None) -> Tensor:
    cb_arg = dict(callback_arg or {})
    cb_arg.update(model_idx_in_bag=0, shift_idx=0, segment_offset=0)         # apply.py:185-187
    kw = dict(shifts=shifts, split=split, overlap=overlap, transition_power=transition_power, segment=segment)
    if isinstance(model, Bag):                                               # apply.py:201-229
        estimates = 0.0
        totals = [0.0] * len(model.sources)
        cb_arg["models"] = len(model.models)
        for sub, w in zip(model.models, model.weights):
            idx = cb_arg["model_idx_in_bag"]
            sub_cb = (lambda d, i=idx: callback({**d, "model_idx_in_bag": i})) if callback else None
            out = apply_model(sub, mix, **kw, callback=sub_cb, callback_arg=cb_arg)
            for k, wk in enumerate(w):
                out[:, k] *= wk
                totals[k] += wk
            estimates = estimates + out
            cb_arg["model_idx_in_bag"] += 1
        for k in range(estimates.shape[1]):
            estimates[:, k] /= totals[k]
        return estimates
    cb_arg.setdefault("models", 1)
    assert transition_power >= 1, "transition_power < 1 leads to weird behavior."
    batch, channels, length = mix.shape
    if shifts:                                                               # apply.py:237-256
        kw["shifts"] = 0
        max_shift = int(0.5 * model.samplerate)
        win = mix if isinstance(mix, Window) else Window(mix)
        padded = win.padded(length + 2 * max_shift)
        out = 0.0
        for s in range(shifts):
            offset = random.randint(0, max_shift)
            shifted = Window(padded, offset, length + max_shift - offset)
            s_cb = (lambda d, i=s: callback({**d, "shift_idx": i})) if callback else None
            res = apply_model(model, shifted, **kw, callback=s_cb, callback_arg=cb_arg)
            out = out + res[..., max_shift - offset:]
        out /= shifts
        return out
    if split:                                                                # apply.py:257-301
        kw["split"] = False
        dtype = mix.tensor.dtype if isinstance(mix, Window) else mix.dtype
        out = torch.zeros(batch, len(model.sources), channels, length, dtype=dtype)
        sum_weight = torch.zeros(length, dtype=dtype)
        seg = model.segment if segment is None else segment
        assert seg is not None and seg > 0.0
        SL = int(model.samplerate * seg)
        stride = int((1 - overlap) * SL)
        weight = transition_weight(SL, transition_power).to(dtype)
        for off in range(0, length, stride):
            o_cb = (lambda d, i=off: callback({**d, "segment_offset": i})) if callback else None
            chunk_out = apply_model(model, Window(mix, off, SL), **kw, callback=o_cb, callback_arg=cb_arg)
            n = chunk_out.shape[-1]
            out[..., off:off + SL] += weight[:n] * chunk_out
            sum_weight[off:off + SL] += weight[:n]
        assert sum_weight.min() > 0
        out /= sum_weight
        return out
    # leaf (apply.py:302-322)
    if hasattr(model, "segment_length"):        # HTDemucs: pad to int(segment * sr), else to the training length (apply.py:305-308)
        valid = int(segment * model.samplerate) if segment is not None else model.segment_length
        if valid < length:
            raise ValueError(f"Given length {length} is longer than training length {valid}")
    elif hasattr(model, "valid_length"):
        valid = model.valid_length(length)
    else:                                        # HDemucs has no valid_length: the chunk runs at its own length (apply.py:309-310)
        valid = length
    win = mix if isinstance(mix, Window) else Window(mix)
    padded = win.padded(valid)
    if callback is not None:
        callback({**cb_arg, "state": "start"})
    out = model(padded)
    if callback is not None:
        callback({**cb_arg, "state": "end"})
    return center_trim(out, length)


def new_sdr(references: Tensor, estimates: Tensor) -> Tensor:
    """SDR of the MDX challenge definition (reference: demucs/evaluate.py:30-43), per (B,S)."""
    delta = 1e-7
    num = torch.sum(torch.square(references), dim=(2, 3)) + delta
    den = torch.sum(torch.square(references - estimates), dim=(2, 3)) + delta
    return 10 * torch.log10(num / den)


def separate_tensor(model, wav: Tensor, callback=None, callback_arg=None, **params):
    """Separator.separate_tensor at the model's own sample rate (demucs/api.py:265-291): `wav` (channels, length) is
    normalised IN PLACE by the mean / unbiased std of its mono mix-down (+1e-8), separated with `apply_model`, the stems
    and `wav` itself are mapped back; returns (wav, {source: stem})."""
    ref = wav.mean(0)
    wav -= ref.mean()
    wav /= ref.std() + 1e-8
    arg = dict(callback_arg or {})
    arg["audio_length"] = wav.shape[1]                                   # api.py:281-283
    out = apply_model(model, wav[None], callback=callback, callback_arg=arg, **params)
    out *= ref.std() + 1e-8
    out += ref.mean()
    wav *= ref.std() + 1e-8
    wav += ref.mean()
    return wav, dict(zip(model.sources, out[0]))
