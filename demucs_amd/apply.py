"""Segment scheduler: drop-in for `demucs.apply` (reference: demucs/apply.py:29-322,
demucs/utils.py:38-54,122-149) with the same names, argument meaning and error behaviour:
`apply_model`, `BagOfModels`, `TensorChunk`, `tensor_chunk`, plus `center_trim` and
`DummyPoolExecutor`.

Semantics kept from the reference
  * bag loop -> shift loop -> overlapping-segment loop -> padded leaf forward + centre trim;
  * `TensorChunk.padded` fills the padding with the underlying tensor's real neighbours;
  * triangular `weight ** transition_power` cross-fade, `out /= sum_weight`;
  * shift offsets come from Python's global `random`, and one `random.randrange(1)` is drawn per
    segment forward (the reference's transformer does that, transformer.py:680), so seeded runs
    see the same offsets as the reference;
  * callback events and their order; exceptions from a segment or a callback propagate;
  * the result is a new float32 tensor on `mix.device`; `mix` is never mutated.

MI355X-first execution: when the model is `demucs_amd.HTDemucs` on a GPU, the split branch keeps
the whole track resident in HBM, cuts all segments of a batch with one gather kernel, runs ONE
batched forward per `model.max_batch` segments, and overlap-adds on the device in the
reference's summation order (bit-identical to the sequential loop on the same per-segment
outputs).  Any other model object takes the generic per-segment route (plain torch tensor
bookkeeping around `model(padded)`), which is also what the CPU host-logic tests exercise.
"""
from __future__ import annotations

import ctypes as C
import os
import random
from concurrent.futures import CancelledError, ThreadPoolExecutor
from threading import Lock
from typing import Any, Callable, Dict, List, Optional, Sequence, Union

import torch
from torch.nn import functional as F

from . import _lib
from .hdemucs import HDemucs, MIN_LENGTH as _HDEMUCS_MIN_LENGTH
from .htdemucs import HTDemucs

__all__ = ["apply_model", "BagOfModels", "TensorChunk", "tensor_chunk", "center_trim", "DummyPoolExecutor"]


# ------------------------------------------------------------------------------------------------
# small pieces with the reference's names
# ------------------------------------------------------------------------------------------------
def center_trim(tensor: torch.Tensor, reference: Union[torch.Tensor, int]) -> torch.Tensor:
    """Trim the last axis to `reference` (a length or a tensor), centred; an odd remainder goes
    to the right (utils.py:38-54)."""
    size = reference.size(-1) if isinstance(reference, torch.Tensor) else int(reference)
    extra = tensor.size(-1) - size
    if extra < 0:
        raise ValueError(f"tensor must be larger than reference. Delta is {extra}.")
    if extra:
        left = extra // 2
        tensor = tensor[..., left:left + size]
    return tensor


class DummyPoolExecutor:
    """Lazy sequential stand-in for an executor: `submit` records the call, `result()` runs it
    (utils.py:122-149)."""

    class _Deferred:
        def __init__(self, owner, fn, args, kwargs):
            self._owner, self._fn, self._args, self._kwargs = owner, fn, args, kwargs

        def result(self):
            if not self._owner._alive:
                raise CancelledError()
            return self._fn(*self._args, **self._kwargs)

    def __init__(self, workers: int = 0):
        self._alive = True

    def submit(self, fn, *args, **kwargs):
        return DummyPoolExecutor._Deferred(self, fn, args, kwargs)

    def shutdown(self, *_, **__):
        self._alive = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return None


class TensorChunk:
    """Zero-copy (offset, length) window over the last axis of a tensor (apply.py:82-124)."""

    def __init__(self, tensor, offset: int = 0, length: Optional[int] = None):
        total = tensor.shape[-1]
        assert offset >= 0
        assert offset < total
        span = total - offset
        self.length = span if length is None else min(span, length)
        if isinstance(tensor, TensorChunk):         # windows of windows collapse onto the base tensor
            self.tensor, self.offset = tensor.tensor, tensor.offset + offset
        else:
            self.tensor, self.offset = tensor, offset
        self.device = tensor.device

    @property
    def shape(self) -> List[int]:
        dims = list(self.tensor.shape)
        dims[-1] = self.length
        return dims

    def window_start(self, target_length: int) -> int:
        """Base-tensor index of sample 0 of `padded(target_length)` (may be negative)."""
        delta = target_length - self.length
        assert delta >= 0
        return self.offset - delta // 2

    def padded(self, target_length: int) -> torch.Tensor:
        begin = self.window_start(target_length)
        end = begin + target_length
        total = self.tensor.shape[-1]
        lo, hi = max(0, begin), min(total, end)
        out = F.pad(self.tensor[..., lo:hi], (lo - begin, end - hi))
        assert out.shape[-1] == target_length
        return out


def tensor_chunk(tensor_or_chunk) -> TensorChunk:
    if isinstance(tensor_or_chunk, TensorChunk):
        return tensor_or_chunk
    assert isinstance(tensor_or_chunk, torch.Tensor)
    return TensorChunk(tensor_or_chunk)


class BagOfModels:
    """Weighted ensemble container (apply.py:29-79).  `weights[i][k]` is the weight of model i for
    source k.  Call `apply_model` on it; calling it directly raises like the reference."""

    def __init__(self, models: Sequence[Any], weights: Optional[List[List[float]]] = None,
                 segment: Optional[float] = None):
        assert len(models) > 0
        first = models[0]
        for other in models:
            assert other.sources == first.sources
            assert other.samplerate == first.samplerate
            assert other.audio_channels == first.audio_channels
            if segment is not None and not isinstance(other, HTDemucs) and segment > other.segment:
                other.segment = segment
        self.audio_channels, self.samplerate, self.sources = first.audio_channels, first.samplerate, first.sources
        self.models = list(models)
        if weights is None:
            weights = [[1.0 for _ in first.sources] for _ in models]
        else:
            assert len(weights) == len(models)
            for w in weights:
                assert len(w) == len(first.sources)
        self.weights = weights

    @property
    def max_allowed_segment(self) -> float:
        limit = float("inf")
        for m in self.models:
            if isinstance(m, HTDemucs):
                limit = min(limit, float(m.segment))
        return limit

    def forward(self, x):
        raise NotImplementedError("Call `apply_model` on this.")

    __call__ = forward


def _with(d: Optional[dict], **subs) -> dict:
    out = dict(d) if d is not None else {}
    out.update(subs)
    return out


def _model_device(model) -> Optional[torch.device]:
    try:
        return next(iter(model.parameters())).device
    except (AttributeError, StopIteration, TypeError):
        return None


def _transition_weight(segment_length: int, transition_power: float, device) -> torch.Tensor:
    """apply.py:271-276: integer ramps / max (float32), then ** transition_power."""
    ramp = torch.cat([torch.arange(1, segment_length // 2 + 1, device=device),
                      torch.arange(segment_length - segment_length // 2, 0, -1, device=device)])
    assert len(ramp) == segment_length
    return (ramp / ramp.max()) ** transition_power


# ------------------------------------------------------------------------------------------------
# apply_model
# ------------------------------------------------------------------------------------------------
def apply_model(model, mix: Union[torch.Tensor, TensorChunk], shifts: int = 1, split: bool = True,
                overlap: float = 0.25, transition_power: float = 1.0, progress: bool = False, device=None,
                num_workers: int = 0, segment: Optional[float] = None, pool=None, lock=None,
                callback: Optional[Callable[[dict], None]] = None, callback_arg: Optional[dict] = None) -> torch.Tensor:
    """Apply `model` to `mix` (B, channels, length); see the module docstring and the reference's
    docstring (apply.py:154-173) for the arguments."""
    device = mix.device if device is None else torch.device(device)
    if pool is None:
        pool = ThreadPoolExecutor(num_workers) if (num_workers > 0 and device.type == "cpu") else DummyPoolExecutor()
    if lock is None:
        lock = Lock()
    if device.type == "cuda" and split and _is_engine(model) and isinstance(mix, torch.Tensor):
        from . import distributed
        if distributed.sharding_active():
            # one process per GPU: every pass's segments are sharded over the ranks, ONE all-gather per call
            out = distributed.apply_model_sharded(model, mix, shifts=shifts, overlap=overlap, transition_power=transition_power,
                                                  segment=segment, device=device, callback=callback, callback_arg=callback_arg,
                                                  lock=lock)
            return out if mix.device == device else _to_host(out, device)
    if device.type == "cuda" and mix.device.type == "cpu" and _is_engine(model):
        return _apply_engine_from_host(model, mix, dict(
            shifts=shifts, split=split, overlap=overlap, transition_power=transition_power, progress=progress, device=device,
            num_workers=num_workers, segment=segment, pool=pool, lock=lock, callback=callback, callback_arg=callback_arg))
    callback_arg = _with(callback_arg, model_idx_in_bag=0, shift_idx=0, segment_offset=0)
    common: Dict[str, Any] = dict(shifts=shifts, split=split, overlap=overlap, transition_power=transition_power,
                                  progress=progress, device=device, pool=pool, segment=segment, lock=lock)

    if isinstance(model, BagOfModels):
        return _apply_bag(model, mix, common, callback, callback_arg)

    callback_arg.setdefault("models", 1)
    model.to(device)
    model.eval()
    assert transition_power >= 1, "transition_power < 1 leads to weird behavior."
    if shifts:
        return _apply_shifts(model, mix, shifts, common, callback, callback_arg)
    if split:
        return _apply_split(model, mix, common, callback, callback_arg)
    return _apply_leaf(model, mix, common, callback, callback_arg)


def _apply_bag(bag: BagOfModels, mix, common, callback, callback_arg) -> torch.Tensor:
    """apply.py:201-229: each model is applied on its own (its own random shifts), scaled per
    source, summed, and normalised by the per-source weight totals."""
    device = common["device"]
    callback_arg["models"] = len(bag.models)
    totals = [0.0] * len(bag.sources)
    estimates = None
    for sub, sub_weights in zip(bag.models, bag.weights):
        idx = callback_arg["model_idx_in_bag"]
        sub_cb = (lambda d, i=idx: callback(_with(d, model_idx_in_bag=i))) if callback else None
        home = _model_device(sub)
        sub.to(device)
        out = apply_model(sub, mix, **common, callback=sub_cb, callback_arg=callback_arg)
        if home is not None:
            sub.to(home)
        for k, w in enumerate(sub_weights):
            out[:, k, :, :] *= w
            totals[k] += w
        estimates = out if estimates is None else estimates.add_(out)
        del out
        callback_arg["model_idx_in_bag"] += 1
    assert isinstance(estimates, torch.Tensor)
    for k in range(estimates.shape[1]):
        estimates[:, k, :, :] /= totals[k]
    return estimates


def _apply_shifts(model, mix, shifts: int, common, callback, callback_arg) -> torch.Tensor:
    """apply.py:237-256: the "shift trick"."""
    kw = dict(common, shifts=0)
    length = mix.shape[-1]
    max_shift = int(0.5 * model.samplerate)
    padded_mix = tensor_chunk(mix).padded(length + 2 * max_shift)
    out = None
    for shift_idx in range(shifts):
        offset = random.randint(0, max_shift)
        shifted = TensorChunk(padded_mix, offset, length + max_shift - offset)
        s_cb = (lambda d, i=shift_idx: callback(_with(d, shift_idx=i))) if callback else None
        res = apply_model(model, shifted, **kw, callback=s_cb, callback_arg=callback_arg)
        piece = res[..., max_shift - offset:]
        out = piece.clone() if out is None else out.add_(piece)
    out /= shifts
    return out


def _segment_plan(model, length: int, overlap: float, segment):
    seg = model.segment if segment is None else segment
    assert seg is not None and seg > 0.0
    segment_length = int(model.samplerate * seg)
    stride = int((1 - overlap) * segment_length)
    return seg, segment_length, stride, list(range(0, length, stride))


def _apply_split(model, mix, common, callback, callback_arg) -> torch.Tensor:
    """apply.py:257-301."""
    if isinstance(model, (HTDemucs, HDemucs)) and common["device"].type == "cuda":
        return _apply_split_device(model, mix, common, callback, callback_arg)
    kw = dict(common, split=False)
    device, pool = common["device"], common["pool"]
    batch, channels, length = mix.shape
    _, segment_length, stride, offsets = _segment_plan(model, length, common["overlap"], common["segment"])
    out = torch.zeros(batch, len(model.sources), channels, length, device=mix.device)
    sum_weight = torch.zeros(length, device=mix.device)
    weight = _transition_weight(segment_length, common["transition_power"], device)
    futures = []
    for offset in offsets:
        chunk = TensorChunk(mix, offset, segment_length)
        o_cb = (lambda d, i=offset: callback(_with(d, segment_offset=i))) if callback else None
        futures.append((pool.submit(apply_model, model, chunk, **kw, callback_arg=callback_arg, callback=o_cb), offset))
    if common["progress"]:
        import tqdm
        scale = float(format(stride / model.samplerate, ".2f"))
        futures = tqdm.tqdm(futures, unit_scale=scale, ncols=120, unit="seconds")
    for future, offset in futures:
        try:
            chunk_out = future.result()
        except Exception:
            pool.shutdown(wait=True, cancel_futures=True)
            raise
        n = chunk_out.shape[-1]
        out[..., offset:offset + segment_length] += (weight[:n] * chunk_out).to(mix.device)
        sum_weight[offset:offset + segment_length] += weight[:n].to(mix.device)
    assert sum_weight.min() > 0
    out /= sum_weight
    return out


def _apply_leaf(model, mix, common, callback, callback_arg) -> torch.Tensor:
    """apply.py:302-322: pad to the valid length, forward under no_grad, centre-trim."""
    device, lock, segment = common["device"], common["lock"], common["segment"]
    length = mix.shape[-1]
    if isinstance(model, HTDemucs) and segment is not None:
        valid_length = int(segment * model.samplerate)
    elif hasattr(model, "valid_length"):
        valid_length = model.valid_length(length)
    else:
        valid_length = length
    padded_mix = tensor_chunk(mix).padded(valid_length).to(device)
    with lock:
        if callback is not None:
            callback(_with(callback_arg, state="start"))
    with torch.no_grad():
        out = model(padded_mix)
    with lock:
        if callback is not None:
            callback(_with(callback_arg, state="end"))
    assert isinstance(out, torch.Tensor)
    return center_trim(out, length)


# ------------------------------------------------------------------------------------------------
# device-resident split branch for the HIP engine
# ------------------------------------------------------------------------------------------------
def hdemucs_min_length() -> int:
    return _HDEMUCS_MIN_LENGTH


def _i64(values, device) -> torch.Tensor:
    # built on the host, ONE copy: torch.tensor(list, device=cuda) writes the elements one by one (a copy kernel each)
    return torch.tensor(list(values), dtype=torch.int64).to(device)


def _i32(values, device) -> torch.Tensor:
    return torch.tensor(list(values), dtype=torch.int32).to(device)


def _is_engine(model) -> bool:
    if isinstance(model, BagOfModels):
        return all(_is_engine(m) for m in model.models)
    return isinstance(model, (HTDemucs, HDemucs))


def _leaf_valid_length(model: HTDemucs, segment_length: int, segment) -> int:
    """Length the leaf pads a chunk to (apply.py:305-310): int(segment * sr) with a segment override, else the
    training length.  Raises the reference's error when it exceeds what the model was built for."""
    valid = int(segment * model.samplerate) if segment is not None else model.valid_length(segment_length)
    if valid > model.segment_length:
        raise ValueError(f"Given length {valid} is longer than training length {model.segment_length}")
    return valid


def device_split_accumulate(model: HTDemucs, base, chunk_offset: int, length: int, offsets: Sequence[int],
                            segment_length: int, valid_length: int, weight: torch.Tensor, acc, acc_origin: int,
                            on_start: Optional[Callable[[int], None]] = None, on_end: Optional[Callable[[int], None]] = None,
                            draw_rng: bool = True, base_origin: int = 0) -> None:
    """Run the segments at `offsets` (relative to the chunk that starts at `chunk_offset` of the
    device-resident track `base` (channels, total) and is `length` long) and add
    `weight[:n] * center_trim(model(padded_i), n)` into `acc` (rows, acc_len), whose sample 0 is
    chunk position `acc_origin`.  One gather + one batched forward + one overlap-add per
    `model.max_batch` segments.  `base` may be a WINDOW of the track whose sample 0 is track position
    `base_origin` (multi-GPU ranks hold only their span + halo): positions outside the window read as
    zero, exactly like positions outside the track, so the window must reach every sample the segments'
    padded windows touch, or the real track end.
    `on_start(offset)` / `on_end(offset)` bracket each segment's forward in the reference's event order
    (start, end, start, end ...): the first `start` of a batch fires before the batched forward is
    enqueued and no `end` fires before the batch has been COMPUTED (the stream is synchronised first when an `end`
    listener exists; without one nothing waits)."""
    if valid_length > model.segment_length:
        raise ValueError(f"Given length {valid_length} is longer than training length {model.segment_length}")
    if segment_length > valid_length or weight.numel() < segment_length:
        raise ValueError(f"segment length {segment_length} does not fit the padded length {valid_length} / the weight "
                         f"ramp ({weight.numel()})")
    # several TRACKS of equal geometry (a batch of mixes): `base` and `acc` are sequences; the segments of one offset group
    # of ALL tracks share one batched forward, as the reference forwards all tracks of an offset in one model call
    bases = [base] if isinstance(base, torch.Tensor) else list(base)
    accs = [acc] if isinstance(acc, torch.Tensor) else list(acc)
    assert len(bases) == len(accs) and all(b.shape == bases[0].shape for b in bases) and all(a.shape == accs[0].shape for a in accs)
    n_tracks = len(bases)
    lib = _lib.load()
    dev = bases[0].device
    channels, total = bases[0].shape
    rows = accs[0].shape[0]
    stream = lambda: C.c_void_p(_lib.current_stream_ptr())          # noqa: E731
    B = model.max_batch
    group = n_tracks if B >= n_tracks else 1      # tracks per forward: all of them, or one at a time when max_batch < n_tracks
    per = max(1, B // group)           # offsets per forward: per * group segments fill the batch either way
    SL = model.segment_length          # the engine's fixed forward length; a shorter leaf window is
    short = valid_length < SL          # right-padded with zeros like HTDemucs.forward does (htdemucs.py:534-537)
    seg_buf = torch.zeros(B, channels, SL, device=dev, dtype=torch.float32)
    cut_buf = torch.empty(B, channels, valid_length, device=dev, dtype=torch.float32) if short else seg_buf
    out_buf = torch.empty(B, rows // channels, channels, SL, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        for g0 in range(0, n_tracks, group):
            tracks = list(range(g0, min(n_tracks, g0 + group)))
            first_group = g0 == 0
            for i0 in range(0, len(offsets), per):
                offs = list(offsets[i0:i0 + per])
                nb = len(offs)
                lens = [min(length - o, segment_length) for o in offs]
                trims = [(valid_length - n) // 2 for n in lens]
                starts = [chunk_offset + o - t - base_origin for o, t in zip(offs, trims)]     # TensorChunk.padded window
                # index tensors stay referenced until the launches below are enqueued: the caching allocator
                # may hand a dropped tensor's block to the next allocation before the kernel has read it
                t_starts = _i64(starts, dev)
                for k, tr in enumerate(tracks):
                    dst = cut_buf[k * nb:(k + 1) * nb]
                    _lib.check(lib.mi_segments_gather(bases[tr].data_ptr(), total, channels, t_starts.data_ptr(), nb, valid_length,
                                                      dst.data_ptr(), dst.numel(), stream()), "mi_segments_gather")
                ntot = nb * len(tracks)
                if short:
                    seg_buf[:ntot, :, :valid_length] = cut_buf[:ntot]
                # the overlap-add's index tensors go to the device BEFORE the forward is enqueued: a host -> device copy issued
                # behind it would hold the host (and the next launch) until the forward has drained
                acc_offs = [o - acc_origin for o in offs]
                t_offs, t_lens, t_trims = _i64(acc_offs, dev), _i32(lens, dev), _i32(trims, dev)
                listen = first_group
                if on_start is not None and listen:
                    on_start(offs[0])
                model.forward_segments(seg_buf[:ntot], out_buf[:ntot])
                if on_end is not None and listen:
                    torch.cuda.current_stream(dev).synchronize()      # "end" means computed, as after the reference's blocking leaf
                if listen:
                    for k, o in enumerate(offs):
                        if draw_rng:
                            random.randrange(1)              # transformer.py:680, once per segment forward (whole batch of tracks)
                        if k and on_start is not None:
                            on_start(o)
                        if on_end is not None:
                            on_end(o)
                # a segment may hang over either end of `acc` (a rank's slab of a shifted pass): only the part inside counts
                span_lo, span_hi = max(0, min(acc_offs)), min(accs[0].shape[1], max(a + n for a, n in zip(acc_offs, lens)))
                if span_hi <= span_lo:
                    continue
                for k, tr in enumerate(tracks):
                    src = out_buf[k * nb:(k + 1) * nb]
                    _lib.check(lib.mi_ola_accumulate(accs[tr].data_ptr(), accs[tr].shape[1], rows, src.data_ptr(), SL, src.numel(),
                                                     t_offs.data_ptr(), t_lens.data_ptr(), t_trims.data_ptr(), nb, span_lo,
                                                     span_hi, weight.data_ptr(), weight.numel(), stream()), "mi_ola_accumulate")


def ragged_split_accumulate(model: HDemucs, base: torch.Tensor, chunk_offset: int, length: int, offsets: Sequence[int],
                            segment_length: int, weight: torch.Tensor, acc: torch.Tensor,
                            on_start: Optional[Callable[[int], None]] = None,
                            on_end: Optional[Callable[[int], None]] = None, acc_origin: int = 0, base_origin: int = 0) -> None:
    """`device_split_accumulate` for a model WITHOUT `valid_length` (HDemucs): the leaf forwards every chunk at its own
    length, unpadded (apply.py:309-310), so consecutive chunks of equal length -- all but the last of a track -- share one
    gather, one batched forward of up to `model.max_batch` chunks and one overlap-add; the shorter tail chunk gets its
    own, overlapped with the batched one on the model's side engine and stream when no listener needs ordered events.
    Events fire in the reference's order (start, end, start, end ...), as in `device_split_accumulate`; `acc_origin` and
    `base_origin` mean what they mean there (a multi-GPU rank's slab and its window of the track)."""
    lib = _lib.load()
    dev = base.device
    channels, total = base.shape
    rows = acc.shape[0]
    stream = lambda: C.c_void_p(_lib.current_stream_ptr())          # noqa: E731
    lens = [min(length - o, segment_length) for o in offsets]
    if weight.numel() < max(lens):
        raise ValueError(f"the weight ramp ({weight.numel()}) is shorter than a chunk ({max(lens)})")

    def gather(offs, n):
        seg = torch.empty(len(offs), channels, n, device=dev, dtype=torch.float32)
        t_starts = _i64([chunk_offset + o - base_origin for o in offs], dev)
        _lib.check(lib.mi_segments_gather(base.data_ptr(), total, channels, t_starts.data_ptr(), len(offs), n, seg.data_ptr(),
                                          seg.numel(), stream()), "mi_segments_gather")
        return seg, t_starts

    def ola_index(offs, n):          # built before the forward is enqueued (see device_split_accumulate)
        return _i64([o - acc_origin for o in offs], dev), _i32([n] * len(offs), dev), _i32([0] * len(offs), dev)

    def overlap_add(out, offs, n, idx):
        nb = len(offs)
        t_offs, t_lens, t_trims = idx
        # a chunk may hang over either end of `acc` (a rank's slab of a shifted pass): only the part inside counts
        span_lo, span_hi = max(0, offs[0] - acc_origin), min(acc.shape[1], offs[-1] - acc_origin + n)
        if span_hi <= span_lo:
            return
        _lib.check(lib.mi_ola_accumulate(acc.data_ptr(), acc.shape[1], rows, out.data_ptr(), n, out.numel(),
                                         t_offs.data_ptr(), t_lens.data_ptr(), t_trims.data_ptr(), nb, span_lo,
                                         span_hi, weight.data_ptr(), weight.numel(), stream()),
                   "mi_ola_accumulate")

    with torch.cuda.device(dev):
        # A track's tail chunk is shorter than the others, so it cannot join their batch, and a forward of this architecture
        # costs 1 600 dependent LSTM step launches whatever its size: without listeners (whose events must fire in order) it
        # runs on the model's single-item side engine and a side stream, under the batched forward of the full chunks; its
        # overlap-add still comes last, as in the reference's loop.
        side = None
        n_main = len(offsets)
        if (on_start is None and on_end is None and len(offsets) >= 2 and lens[-1] < lens[-2]
                and lens[-1] >= hdemucs_min_length() and os.environ.get("MI_NO_TAIL_OVERLAP") is None):
            main_stream, side = torch.cuda.current_stream(dev), model.side_stream()
            side.wait_stream(main_stream)                 # `base` and whatever produced it
            with torch.cuda.stream(side):
                tail_seg, tail_idx = gather([offsets[-1]], lens[-1])
                tail_ola = ola_index([offsets[-1]], lens[-1])
                tail_out = model(tail_seg, aux=True)
            n_main -= 1
        i = 0
        while i < n_main:
            n, j = lens[i], i + 1
            while j < n_main and j - i < model.max_batch and lens[j] == n:
                j += 1
            offs = list(offsets[i:j])
            i = j
            seg, t_starts = gather(offs, n)
            idx = ola_index(offs, n)
            if on_start is not None:
                on_start(offs[0])
            out = model(seg)
            if on_end is not None:
                torch.cuda.current_stream(dev).synchronize()      # "end" means computed
            for k, o in enumerate(offs):
                if k and on_start is not None:
                    on_start(o)
                if on_end is not None:
                    on_end(o)
            overlap_add(out, offs, n, idx)
        if side is not None:
            main_stream.wait_stream(side)
            for t in (tail_seg, tail_idx, tail_out, *tail_ola):
                t.record_stream(main_stream)                      # allocated under the side stream, consumed on this one
            overlap_add(tail_out, [offsets[-1]], lens[-1], tail_ola)


def finish_index(length: int, offsets: Sequence[int], segment_length: int, dev):
    """(offsets, lengths) of ALL segments of a chunk on the device, for `device_split_finish`; built ahead of the forwards so
    that the copy does not wait behind them."""
    return _i64(offsets, dev), _i32([min(length - o, segment_length) for o in offsets], dev)


def device_split_finish(acc: torch.Tensor, acc_origin: int, length: int, offsets: Sequence[int], segment_length: int,
                        weight: torch.Tensor, index=None) -> None:
    """`out /= sum_weight` on the device, sum_weight rebuilt from ALL segment offsets of the chunk."""
    lib = _lib.load()
    dev = acc.device
    with torch.cuda.device(dev):
        t_offs, t_lens = index if index is not None else finish_index(length, offsets, segment_length, dev)
        _lib.check(lib.mi_ola_finish(acc.data_ptr(), acc.shape[1], acc.shape[0], acc_origin, t_offs.data_ptr(),
                                     t_lens.data_ptr(), len(offsets), segment_length, weight.data_ptr(),
                                     C.c_void_p(_lib.current_stream_ptr())), "mi_ola_finish")


def _apply_split_device(model, mix, common, callback, callback_arg) -> torch.Tensor:
    device, lock = common["device"], common["lock"]
    chunk = tensor_chunk(mix)
    batch, channels, length = chunk.shape
    seg, segment_length, stride, offsets = _segment_plan(model, length, common["overlap"], common["segment"])
    ragged = isinstance(model, HDemucs)
    valid_length = None if ragged else _leaf_valid_length(model, segment_length, common["segment"])
    weight = _transition_weight(segment_length, common["transition_power"], device).to(torch.float32).contiguous()
    S = len(model.sources)
    on_device = mix.device == device
    out = (torch.zeros if on_device else torch.empty)(batch, S, channels, length, device=mix.device, dtype=torch.float32)
    bar = None
    if common["progress"]:
        import tqdm
        scale = float(format(stride / model.samplerate, ".2f"))
        bar = tqdm.tqdm(total=len(offsets), unit_scale=scale, ncols=120, unit="seconds")

    def on_start(offset):
        if callback is not None:
            with lock:
                callback(_with(callback_arg, segment_offset=offset, state="start"))

    def on_end(offset):
        if callback is not None:
            with lock:
                callback(_with(callback_arg, segment_offset=offset, state="end"))
        if bar is not None:
            bar.update(1)

    if not ragged:
        # the reference forwards all `batch` tracks of a segment offset in ONE model call (one RNG draw and one start / end event
        # pair per offset): here the tracks' segments of an offset group share one batched forward as well
        bases = [chunk.tensor[b].to(device=device, dtype=torch.float32).contiguous() for b in range(batch)]       # resident in HBM
        accs = [out[b].view(S * channels, length) if on_device else torch.zeros(S * channels, length, device=device, dtype=torch.float32)
                for b in range(batch)]
        fin = finish_index(length, offsets, segment_length, device)
        listen = callback is not None or bar is not None        # without a listener nothing waits for the forwards
        device_split_accumulate(model, bases, chunk.offset, length, offsets, segment_length, valid_length, weight, accs, 0,
                                on_start if listen else None, on_end if listen else None, draw_rng=True)
        for b in range(batch):
            device_split_finish(accs[b], 0, length, offsets, segment_length, weight, fin)
            if not on_device:
                out[b] = accs[b].view(S, channels, length).to(mix.device)
        if bar is not None:
            bar.close()
        return out
    for b in range(batch):
        # HDemucs (ragged route): one track at a time; events and the progress bar follow the first track
        first = b == 0
        base = chunk.tensor[b].to(device=device, dtype=torch.float32).contiguous()       # whole track resident in HBM
        acc = out[b].view(S * channels, length) if on_device else torch.zeros(S * channels, length, device=device, dtype=torch.float32)
        fin = finish_index(length, offsets, segment_length, device)
        listen = first and (callback is not None or bar is not None)     # no listener: the tail chunk may overlap the others
        ragged_split_accumulate(model, base, chunk.offset, length, offsets, segment_length, weight, acc,
                                on_start if listen else None, on_end if listen else None)
        device_split_finish(acc, 0, length, offsets, segment_length, weight, fin)
        if not on_device:
            out[b] = acc.view(S, channels, length).to(mix.device)
    if bar is not None:
        bar.close()
    model.check()          # a time-out of the LAST forward's recurrence would otherwise pass unnoticed (include/demucs_amd.h)
    return out


def _apply_engine_from_host(model, mix, kwargs) -> torch.Tensor:
    """Host `mix`, GPU engine: ONE H2D of the mix, the whole bag / shift / split recursion on device-resident tensors
    (averages accumulated in HBM), ONE D2H of the finished stems into a pinned host tensor.  Same result contract as the
    reference: a new float32 tensor on `mix.device`."""
    device = kwargs["device"]
    chunk = tensor_chunk(mix)
    src = chunk.tensor
    with torch.cuda.device(device):
        dev_src = src.to(device=device, dtype=torch.float32, non_blocking=True)
        dev_mix = TensorChunk(dev_src, chunk.offset, chunk.length) if isinstance(mix, TensorChunk) else dev_src
        out = apply_model(model, dev_mix, **kwargs)
    # the reference builds the result on `mix.device` in the split branch only (apply.py:259); a bare leaf
    # (and a shift average over bare leaves) stays on `device`
    return _to_host(out, device) if kwargs["split"] else out


def _to_host(out: torch.Tensor, device) -> torch.Tensor:
    """Device result -> new pinned host tensor (one DMA at PCIe rate instead of a staged pageable copy)."""
    with torch.cuda.device(device):
        host = torch.empty(out.shape, dtype=torch.float32, pin_memory=True)
        host.copy_(out, non_blocking=True)
        torch.cuda.current_stream(device).synchronize()
    return host
