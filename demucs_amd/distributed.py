"""Multi-GPU segment sharding: one process per GPU, `torch.distributed` (backend "nccl" = RCCL on
ROCm; "gloo" on CPU for tests).

The reference has no multi-GPU inference (`apply_model` takes one device, apply.py:145-153).  The
overlapping-segment loop is embarrassingly parallel (apply.py:278-285) and only couples through
the weighted overlap-add (apply.py:295-299), so:

  * the list of segment offsets is cut into `world` contiguous ranges;
  * rank r runs its segments (batched on its GPU) and accumulates `weight * out` into a slab
    covering [first offset, last offset + segment) of the track, un-normalised;
  * ONE all-gather of the equally padded slabs (RCCL over xGMI) gives every rank all slabs;
  * each rank adds the slabs into the full-track buffer in rank order and divides by the summed
    weights.  With overlap <= 0.5 a sample is covered by at most two segments, so the stitched
    float32 result is bit-identical to the single-GPU (and to the reference's sequential) order.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import apply as _apply
from .htdemucs import HTDemucs

__all__ = ["shard_ranges", "apply_model_sharded"]


def shard_ranges(n_items: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous [lo, hi) index ranges, sizes differing by at most one, earlier ranks larger."""
    base, rem = divmod(n_items, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def _slab_span(offsets: Sequence[int], lo: int, hi: int, length: int, segment_length: int) -> Tuple[int, int]:
    if hi <= lo:
        return 0, 0
    return offsets[lo], min(length, offsets[hi - 1] + segment_length)


def apply_model_sharded(model, mix: torch.Tensor, overlap: float = 0.25, transition_power: float = 1.0,
                        segment: Optional[float] = None, group=None, device=None) -> torch.Tensor:
    """Split-branch `apply_model(model, mix, shifts=0, split=True)` with the segments sharded over
    the ranks of `group`.  mix: (1, channels, length), identical on every rank (on `device` or on
    the host).  Returns the full (1, S, channels, length) result on `device` on EVERY rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    device = mix.device if device is None else torch.device(device)
    assert mix.dim() == 3 and mix.shape[0] == 1, "apply_model_sharded takes one track"
    assert transition_power >= 1, "transition_power < 1 leads to weird behavior."
    model.to(device)
    model.eval()
    _, channels, length = mix.shape
    _, segment_length, stride, offsets = _apply._segment_plan(model, length, overlap, segment)
    ranges = shard_ranges(len(offsets), world)
    spans = [_slab_span(offsets, lo, hi, length, segment_length) for lo, hi in ranges]
    max_span = max(b - a for a, b in spans)
    rows = len(model.sources) * channels
    weight = _apply._transition_weight(segment_length, transition_power, device).to(torch.float32).contiguous()
    engine = isinstance(model, HTDemucs) and device.type == "cuda"

    lo, hi = ranges[rank]
    a0, a1 = spans[rank]
    slab = torch.zeros(rows, max_span, device=device, dtype=torch.float32)
    mine = offsets[lo:hi]
    base = mix[0].to(device=device, dtype=torch.float32).contiguous()
    if engine:
        valid = int(segment * model.samplerate) if segment is not None else model.valid_length(segment_length)
        if mine:
            _apply.device_split_accumulate(model, base, 0, length, mine, segment_length, valid, weight, slab, a0)
    else:
        kw = dict(shifts=0, split=False, overlap=overlap, transition_power=transition_power, device=device, segment=segment)
        for off in mine:
            chunk = _apply.TensorChunk(mix, off, segment_length)
            out = _apply.apply_model(model, chunk, **kw)
            n = out.shape[-1]
            slab[:, off - a0:off - a0 + n] += (weight[:n] * out[0].reshape(rows, n)).to(device)

    if world > 1:
        gathered = torch.empty(world, rows, max_span, device=device, dtype=torch.float32)
        if dist.get_backend(group) == "nccl":
            dist.all_gather_into_tensor(gathered, slab, group=group)    # ONE RCCL all-gather over xGMI
        else:
            dist.all_gather(list(gathered.unbind(0)), slab, group=group)
    else:
        gathered = slab[None]

    total = torch.zeros(rows, length, device=device, dtype=torch.float32)
    for r, (s0, s1) in enumerate(spans):                                # rank order = ascending offsets
        if s1 > s0:
            total[:, s0:s1] += gathered[r, :, :s1 - s0]
    if engine:
        _apply.device_split_finish(total, 0, length, offsets, segment_length, weight)
    else:
        sum_weight = torch.zeros(length, device=device)
        for off in offsets:
            n = min(length - off, segment_length)
            sum_weight[off:off + n] += weight[:n]
        total /= sum_weight
    return total.view(1, len(model.sources), channels, length)
