"""Multi-GPU segment sharding: one process per GPU, `torch.distributed` (backend "nccl" = RCCL on
ROCm; "gloo" on CPU for tests).

The reference has no multi-GPU inference (`apply_model` takes one device, apply.py:145-153).  The
overlapping-segment loop is embarrassingly parallel (apply.py:278-285) and only couples through
the weighted overlap-add (apply.py:295-299), so:

  * the TRACK is cut into `world` contiguous intervals whose boundaries are segment offsets of the
    un-shifted plan; in every pass (bag member x shift) a rank runs the segments that START in its
    interval, batched on its GPU, so its results always land in the same slab
    [interval start, interval end + segment) of the track, whatever the shift;
  * a rank keeps only its window of the mix in HBM (its slab plus the padding halo), not the track;
  * per pass the rank's partial overlap-add is divided by the pass's summed weights (known to
    every rank from the offset list alone) and added, scaled by the bag weight / shift count, into
    the rank's contribution slab: the bag and shift averages are accumulated on the device;
  * ONE all-gather of the equally padded contribution slabs (RCCL over xGMI) per call, whatever
    the number of bag members and shifts; every rank adds the slabs into the full-track buffer in
    rank order.
With a single pass (one model, shifts=0) the slabs are gathered un-normalised and divided after
the stitch: with overlap <= 0.5 a sample is covered by at most two segments, so the float32
result is then bit-identical to the single-GPU (and to the reference's sequential) order.  With
several passes the seam samples between two ranks' slabs may differ from the single-GPU result in
the last bit (the division is applied per rank before the sum); everything else is identical.

`apply_model` (demucs_amd/apply.py) routes here by itself when a process group with more than
one rank is initialised, the model is the HIP engine on a GPU and `split=True`; `no_sharding()`
switches that off (e.g. to compute a single-process reference inside a distributed job).
"""
from __future__ import annotations

import contextlib
import os
import random
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import apply as _apply
from .htdemucs import HTDemucs

__all__ = ["shard_ranges", "apply_model_sharded", "no_sharding", "sharding_active", "track_intervals"]

_enabled = os.environ.get("DEMUCS_AMD_SHARD", "1") != "0"


@contextlib.contextmanager
def no_sharding():
    """Inside this block `apply_model` never shards, whatever process group exists."""
    global _enabled
    old, _enabled = _enabled, False
    try:
        yield
    finally:
        _enabled = old


def sharding_active(group=None) -> bool:
    return _enabled and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def shard_ranges(n_items: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous [lo, hi) index ranges, sizes differing by at most one, earlier ranks larger."""
    base, rem = divmod(n_items, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def track_intervals(length: int, stride: int, world: int) -> List[Tuple[int, int]]:
    """[lo, hi) track intervals per rank: boundaries are offsets of the un-shifted segment plan
    `range(0, length, stride)` cut into contiguous, equally sized index ranges.  Ranks beyond the
    number of segments get an empty interval at the end of the track."""
    offsets = list(range(0, length, stride))
    out = []
    for lo, hi in shard_ranges(len(offsets), world):
        a = offsets[lo] if lo < len(offsets) else length
        b = offsets[hi] if hi < len(offsets) else length
        out.append((a, b))
    return out


def _agree_on(value: int, device, group) -> int:
    """Rank 0's value on every rank (shift offsets come from each process's own Python RNG)."""
    t = torch.tensor([value], dtype=torch.int64, device=device if dist.get_backend(group) == "nccl" else "cpu")
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return int(t.item())


def apply_model_sharded(model, mix: torch.Tensor, shifts: int = 0, overlap: float = 0.25, transition_power: float = 1.0,
                        segment: Optional[float] = None, group=None, device=None,
                        callback: Optional[Callable[[dict], None]] = None, callback_arg: Optional[dict] = None,
                        lock=None) -> torch.Tensor:
    """`apply_model(model, mix, shifts=shifts, split=True, ...)` with the segments of every pass sharded over
    the ranks of `group`.  model: an engine `HTDemucs`, a `BagOfModels` of them, or (host tests) any model
    object `apply_model` accepts.  mix: (B, channels, length), identical on every rank, on `device` or
    on the host.  Returns the full (B, S, channels, length) result on `device` on EVERY rank.
    Callbacks fire on each rank for that rank's segments only (same dict keys as the reference)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    device = mix.device if device is None else torch.device(device)
    assert mix.dim() == 3, "mix must be (batch, channels, length)"
    assert transition_power >= 1, "transition_power < 1 leads to weird behavior."
    bag = isinstance(model, _apply.BagOfModels)
    models = list(model.models) if bag else [model]
    bag_weights = [list(w) for w in model.weights] if bag else [[1.0] * len(model.sources)]
    S = len(model.sources)
    batch, channels, length = mix.shape
    rows = S * channels
    sr = models[0].samplerate
    max_shift = int(0.5 * sr) if shifts else 0
    n_shift = max(shifts, 1)
    n_pass = len(models) * n_shift
    cb_arg = _apply._with(callback_arg, model_idx_in_bag=0, shift_idx=0, segment_offset=0, models=len(models))
    if lock is None:
        from threading import Lock
        lock = Lock()

    # every sub-model shares the segment plan of the call (apply.py passes `segment` down unchanged)
    _, segment_length, stride, _ = _apply._segment_plan(models[0], length, overlap, segment)
    for m in models[1:]:
        assert _apply._segment_plan(m, length, overlap, segment)[1] == segment_length, "bag members disagree on the segment length"
    intervals = track_intervals(length, stride, world)
    i_lo, i_hi = intervals[rank]
    last_owner = max(r for r, (a, b) in enumerate(intervals) if b > a)       # takes every start beyond its interval too
    slab_lo = [a for a, _ in intervals]
    slab_hi = [min(length, b + segment_length) if b > a else a for a, b in intervals]
    max_slab = max(h - l for l, h in zip(slab_lo, slab_hi))
    my_len = slab_hi[rank] - slab_lo[rank]
    engine = all(isinstance(m, HTDemucs) for m in models) and device.type == "cuda"
    weight = _apply._transition_weight(segment_length, transition_power, device).to(torch.float32).contiguous()

    contrib = torch.zeros(batch, rows, max_slab, device=device, dtype=torch.float32)
    if engine:
        valid = _apply._leaf_valid_length(models[0], segment_length, segment)
        # this rank's window of the mix: every sample a padded segment window of any pass can touch
        w_lo = max(0, i_lo - max_shift - valid)
        w_hi = min(length, (length if rank == last_owner else i_hi) + 2 * valid)
        window = mix[:, :, w_lo:w_hi].to(device=device, dtype=torch.float32).contiguous() if my_len else None
    else:
        padded_mix = _apply.tensor_chunk(mix).padded(length + 2 * max_shift)

    for mi, (sub, sub_w) in enumerate(zip(models, bag_weights)):
        sub.to(device)
        sub.eval()
        for si in range(n_shift):
            offset = random.randint(0, max_shift) if shifts else 0          # apply.py:245
            if shifts and world > 1:
                offset = _agree_on(offset, device, group)
            d = max_shift - offset                                          # virtual position v <-> track sample v - d
            lv = length + d                                                 # length of the shifted chunk
            offsets = list(range(0, lv, stride))
            for _ in offsets:
                random.randrange(1)      # transformer.py:680 once per segment forward of the pass, on EVERY rank: seeded
                #                          ranks stay in step with each other and with a seeded single-process run
            starts_t = [max(0, v - d) for v in offsets]
            mine = [v for v, t in zip(offsets, starts_t) if i_lo <= t and (t < i_hi or rank == last_owner)] if my_len else []
            arg = _apply._with(cb_arg, model_idx_in_bag=mi, shift_idx=si)

            def fire(state, v, _arg=arg):
                if callback is not None:
                    with lock:
                        callback(_apply._with(_arg, segment_offset=v, state=state))

            scale = torch.tensor([w / n_shift for w in sub_w], device=device, dtype=torch.float32)
            for b in range(batch):
                if not my_len:
                    continue
                first = b == 0
                v0 = slab_lo[rank] + d                                       # virtual position of slab sample 0
                part = torch.zeros(rows, my_len, device=device, dtype=torch.float32)
                if engine:
                    if mine:
                        _apply.device_split_accumulate(
                            sub, window[b], -d, lv, mine, segment_length, valid, weight, part, v0,
                            (lambda v: fire("start", v)) if first else None, (lambda v: fire("end", v)) if first else None,
                            draw_rng=False, base_origin=w_lo)
                else:
                    kw = dict(shifts=0, split=False, overlap=overlap, transition_power=transition_power, device=device,
                              segment=segment)
                    with no_sharding():
                        for v in mine:
                            chunk = _apply.TensorChunk(_apply.TensorChunk(padded_mix, offset, lv), v, segment_length)
                            if first:
                                fire("start", v)
                            state = random.getstate()                     # the leaf draws nothing the pass has not drawn above
                            out = _apply.apply_model(sub, chunk, **kw)[b].reshape(rows, -1).to(device)
                            random.setstate(state)
                            if first:
                                fire("end", v)
                            n = out.shape[-1]
                            lo = v - v0                                       # may be negative for the first shifted segment
                            a, c = max(lo, 0), min(lo + n, my_len)
                            part[:, a:c] += weight[a - lo:c - lo] * out[:, a - lo:c - lo]
                if n_pass > 1:
                    _normalise(part, v0, lv, offsets, segment_length, weight, engine)
                    contrib[b].view(S, channels, max_slab)[:, :, :my_len].add_(part.view(S, channels, my_len) * scale[:, None, None])
                else:
                    contrib[b, :, :my_len] = part
                del part

    if world > 1:
        gathered = torch.empty(world, batch, rows, max_slab, device=device, dtype=torch.float32)
        if dist.get_backend(group) == "nccl":
            dist.all_gather_into_tensor(gathered, contrib, group=group)    # ONE RCCL all-gather over xGMI
        else:
            dist.all_gather(list(gathered.unbind(0)), contrib, group=group)
    else:
        gathered = contrib[None]
    del contrib

    total = torch.zeros(batch, rows, length, device=device, dtype=torch.float32)
    for r, (s0, s1) in enumerate(zip(slab_lo, slab_hi)):               # rank order = ascending offsets
        if s1 > s0:
            total[:, :, s0:s1] += gathered[r, :, :, :s1 - s0]
    del gathered
    if n_pass == 1:
        offsets = list(range(0, length, stride))
        for b in range(batch):
            _normalise(total[b], 0, length, offsets, segment_length, weight, engine)
    total = total.view(batch, S, channels, length)
    if bag:
        totals = [sum(w[k] for w in bag_weights) for k in range(S)]
        total /= torch.tensor(totals, device=device, dtype=torch.float32)[None, :, None, None]
    return total


def _normalise(acc: torch.Tensor, acc_origin: int, length: int, offsets: Sequence[int], segment_length: int,
               weight: torch.Tensor, engine: bool) -> None:
    """acc (rows, n), whose sample 0 is chunk position acc_origin, /= the summed weights of ALL segments of the pass."""
    if engine:
        _apply.device_split_finish(acc, acc_origin, length, offsets, segment_length, weight)
        return
    n = acc.shape[1]
    sum_weight = torch.zeros(n, device=acc.device)
    for off in offsets:                                                 # ascending, float32: the reference's order
        m = min(length - off, segment_length)
        lo = off - acc_origin
        a, c = max(lo, 0), min(lo + m, n)
        if c > a:
            sum_weight[a:c] += weight[a - lo:c - lo]
    acc /= sum_weight
