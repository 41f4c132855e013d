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
    rank order.  A single pass of a plain model (no bag, no shifts) pipelines that exchange: one
    asynchronous all-gather per batched forward, of the part of the slab that forward completed,
    under the next forward (`_single_pass_pipelined`).
With a single pass of a plain model (no bag, shifts=0) the slabs are gathered un-normalised and divided after
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
from .hdemucs import HDemucs
from .htdemucs import HTDemucs

__all__ = ["shard_ranges", "apply_model_sharded", "no_sharding", "force_collectives", "sharding_active", "track_intervals",
           "rng_draws_per_forward", "collect_timing", "timing_summary"]

_enabled = os.environ.get("DEMUCS_AMD_SHARD", "1") != "0"
# DEMUCS_AMD_SHARD=force: take the sharded route and execute every collective even in a process group of ONE rank, so that
# the RCCL calls (broadcast, all_gather_into_tensor) run on a one-GPU box (tests/test_gpu_nccl.py, `bench.py --force-dist`)
_forced = os.environ.get("DEMUCS_AMD_SHARD") == "force"


@contextlib.contextmanager
def no_sharding():
    """Inside this block `apply_model` never shards, whatever process group exists."""
    global _enabled
    old, _enabled = _enabled, False
    try:
        yield
    finally:
        _enabled = old


@contextlib.contextmanager
def force_collectives():
    """Inside this block the sharded route and its collectives run for ANY process group, a single rank included."""
    global _forced
    old, _forced = _forced, True
    try:
        yield
    finally:
        _forced = old


# DEMUCS_AMD_SHARD_PIPELINE=0: one all-gather of whole slabs after all forwards (round 3's schedule) instead of one per batched forward
_pipelined = os.environ.get("DEMUCS_AMD_SHARD_PIPELINE", "1") != "0"
_timing = None          # a list while `collect_timing()` is active: one tuple of device events per sharded call


@contextlib.contextmanager
def collect_timing():
    """Inside this block every sharded call on a GPU records device events around its three phases -- the rank's own segment
    forwards, the all-gather, the stitch (+ normalisation); `timing_summary(events)` turns them into mean milliseconds
    (`bench.py --gpus N` reports them per rank-0 step: the serial tail of the sharded design)."""
    global _timing
    old, _timing = _timing, []
    try:
        yield _timing
    finally:
        _timing = old


def timing_summary(events) -> dict:
    if not events:
        return {}
    torch.cuda.synchronize()
    n = len(events)
    return {"calls": n,
            "segments_ms": round(sum(a.elapsed_time(b) for a, b, _, _ in events) / n, 3),
            "all_gather_ms": round(sum(b.elapsed_time(c) for _, b, c, _ in events) / n, 3),
            "stitch_ms": round(sum(c.elapsed_time(d) for _, _, c, d in events) / n, 3)}


def _mark(device):
    if _timing is None or device.type != "cuda":
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record(torch.cuda.current_stream(device))
    return e


def sharding_active(group=None) -> bool:
    return (_enabled and dist.is_available() and dist.is_initialized()
            and (dist.get_world_size(group) > 1 or _forced))


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


def rng_draws_per_forward(model) -> int:
    """How many `random.randrange(1)`-sized draws one forward of `model` takes from Python's global RNG: the reference's
    HTDemucs draws once in its transformer (transformer.py:680), its HDemucs draws nothing (hdemucs.py:689-794).  Any other
    model object states it with an integer attribute `rng_draws_per_forward` (default 0).  A rank only runs its own
    segments, so this is what keeps every rank's RNG stream -- and with it the shift offsets of the later passes -- equal
    to a seeded single-process run."""
    if isinstance(model, HTDemucs):
        return 1
    if isinstance(model, HDemucs):
        return 0
    return int(getattr(model, "rng_draws_per_forward", 0))


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
    # engine routes: "segments" = HTDemucs (every segment padded to the training length, one batched forward per max_batch
    # segments), "ragged" = HDemucs (no valid_length: every chunk at its own length, equal chunks batched); anything else is
    # a host model object on the generic per-segment route (CPU tests)
    kinds = ["segments" if isinstance(m, HTDemucs) else "ragged" if isinstance(m, HDemucs) else "generic" for m in models]
    engine = device.type == "cuda" and all(k != "generic" for k in kinds)
    if not engine:
        kinds = ["generic"] * len(models)
    weight = _apply._transition_weight(segment_length, transition_power, device).to(torch.float32).contiguous()
    # one pass of one plain model: slabs are gathered un-normalised and divided after the stitch (bit-identical to one GPU);
    # any shift (its virtual offsets differ from the un-shifted plan) or bag (per-source weights) takes the per-pass branch
    single = n_pass == 1 and not shifts and not bag

    ev0 = _mark(device)
    if single and _pipelined and kinds[0] in ("segments", "generic") and (world > 1 or (_forced and dist.is_initialized())):
        return _single_pass_pipelined(models[0], kinds[0], mix, device, group, world, rank, intervals, last_owner, slab_lo, slab_hi,
                                      segment_length, stride, weight, overlap, transition_power, segment, S, channels, callback, cb_arg,
                                      lock, ev0)
    contrib = torch.zeros(batch, rows, max_slab, device=device, dtype=torch.float32)
    if engine:
        valid = max([_apply._leaf_valid_length(m, segment_length, segment) for m, k in zip(models, kinds) if k == "segments"],
                    default=0)
        # this rank's window of the mix: every sample a (padded) segment window of any pass can touch
        w_lo = max(0, i_lo - max_shift - valid)
        w_hi = min(length, (length if rank == last_owner else i_hi + segment_length) + 2 * valid)
        window = mix[:, :, w_lo:w_hi].to(device=device, dtype=torch.float32).contiguous() if my_len else None
    else:
        padded_mix = _apply.tensor_chunk(mix).padded(length + 2 * max_shift)

    for mi, (sub, sub_w, kind) in enumerate(zip(models, bag_weights, kinds)):
        sub.to(device)
        sub.eval()
        draws = rng_draws_per_forward(sub)
        for si in range(n_shift):
            offset = random.randint(0, max_shift) if shifts else 0          # apply.py:245
            if shifts and (world > 1 or (_forced and dist.is_initialized())):
                offset = _agree_on(offset, device, group)
            d = max_shift - offset                                          # virtual position v <-> track sample v - d
            lv = length + d                                                 # length of the shifted chunk
            offsets = list(range(0, lv, stride))
            for _ in range(draws * len(offsets)):
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
                listen = first and callback is not None
                on_start, on_end = ((lambda v: fire("start", v)), (lambda v: fire("end", v))) if listen else (None, None)
                if kind == "segments":
                    if mine:
                        _apply.device_split_accumulate(
                            sub, window[b], -d, lv, mine, segment_length, _apply._leaf_valid_length(sub, segment_length, segment),
                            weight, part, v0, on_start, on_end, draw_rng=False, base_origin=w_lo)
                elif kind == "ragged":
                    if mine:
                        _apply.ragged_split_accumulate(sub, window[b], -d, lv, mine, segment_length, weight, part, on_start, on_end,
                                                       acc_origin=v0, base_origin=w_lo)
                else:
                    kw = dict(shifts=0, split=False, overlap=overlap, transition_power=transition_power, device=device,
                              segment=segment)
                    with no_sharding():
                        for v in mine:
                            chunk = _apply.TensorChunk(_apply.TensorChunk(padded_mix, offset, lv), v, segment_length)
                            if first:
                                fire("start", v)
                            state = random.getstate()                     # the pass's draws were taken above, on every rank
                            out = _apply.apply_model(sub, chunk, **kw)[b].reshape(rows, -1).to(device)
                            random.setstate(state)
                            if first:
                                fire("end", v)
                            n = out.shape[-1]
                            lo = v - v0                                       # may be negative for the first shifted segment
                            a, c = max(lo, 0), min(lo + n, my_len)
                            part[:, a:c] += weight[a - lo:c - lo] * out[:, a - lo:c - lo]
                if not single:
                    _normalise(part, v0, lv, offsets, segment_length, weight, engine)
                    contrib[b].view(S, channels, max_slab)[:, :, :my_len].add_(part.view(S, channels, my_len) * scale[:, None, None])
                else:
                    contrib[b, :, :my_len] = part
                del part

    ev1 = _mark(device)
    if world > 1 or (_forced and dist.is_initialized()):
        gathered = torch.empty(world, batch, rows, max_slab, device=device, dtype=torch.float32)
        if dist.get_backend(group) == "nccl":
            dist.all_gather_into_tensor(gathered, contrib, group=group)    # ONE RCCL all-gather over xGMI
        else:
            dist.all_gather(list(gathered.unbind(0)), contrib, group=group)
    else:
        gathered = contrib[None]
    del contrib
    ev2 = _mark(device)

    # Stitch in rank order (= ascending offsets).  A slab only overlaps its predecessor's tail (segment - stride samples): that
    # head is ADDED, everything behind it is COPIED -- the same values as zero-fill + add over the whole track (0 + a == a),
    # with a third of the traffic (no 5 GB zero-fill and no read of it for the 60-minute track)
    total = torch.empty(batch, rows, length, device=device, dtype=torch.float32)
    done = 0                                                          # total[..., :done] holds data
    for r, (s0, s1) in enumerate(zip(slab_lo, slab_hi)):
        if s1 <= s0:
            continue
        if s0 > done:
            total[:, :, done:s0].zero_()                              # cannot happen with overlap > 0; kept for safety
            done = s0
        head = min(done, s1)
        if head > s0:
            total[:, :, s0:head] += gathered[r, :, :, :head - s0]
        if s1 > head:
            total[:, :, head:s1] = gathered[r, :, :, head - s0:s1 - s0]
        done = max(done, s1)
    if done < length:
        total[:, :, done:].zero_()
    del gathered
    if single:
        offsets = list(range(0, length, stride))
        for b in range(batch):
            _normalise(total[b], 0, length, offsets, segment_length, weight, engine)
    total = total.view(batch, S, channels, length)
    if bag:
        totals = [sum(w[k] for w in bag_weights) for k in range(S)]
        total /= torch.tensor(totals, device=device, dtype=torch.float32)[None, :, None, None]
    if ev0 is not None:
        _timing.append((ev0, ev1, ev2, _mark(device)))
    for sub in models:                 # after the last collective: a rank that raised earlier would leave the others waiting in it
        if isinstance(sub, HDemucs):
            sub.check()                # a time-out of the last forward's recurrence would otherwise pass unnoticed
    return total


def _single_pass_pipelined(sub, kind, mix, device, group, world, rank, intervals, last_owner, slab_lo, slab_hi, segment_length, stride,
                           weight, overlap, transition_power, segment, S, channels, callback, cb_arg, lock, ev0):
    """One pass of a plain model (no bag, no shifts) with the exchange PIPELINED under the forwards.

    A rank runs its segments in groups of `max_batch` (one batched forward each).  After group g the part of its slab in front of
    group g + 1's first segment is final -- no later segment reaches back over it -- so that chunk is all-gathered at once
    (asynchronously: RCCL on its own stream) while the next group's forward runs; only the last chunk's exchange is exposed.  Every
    rank derives every rank's chunk plan from the offset list alone, so all ranks issue the same collectives with the same
    (padded) sizes.  The chunks are stitched as the whole slabs were (copied; the seam with the previous rank's tail added) and
    the track is divided by the summed weights afterwards: the same sums in the same order as the one-shot exchange and as one GPU.
    With 8 ranks on the 60-minute track (77 segments per rank = forwards of 32 + 32 + 13) five sixths of the 4.5 GB each rank
    receives move under compute."""
    batch, _, length = mix.shape
    rows = S * channels
    offsets = list(range(0, length, stride))
    per = max(1, int(getattr(sub, "max_batch", 4)))
    plans = []                      # per rank: (its groups of offsets, chunk bounds in slab coordinates)
    for r, (a, b) in enumerate(intervals):
        n = slab_hi[r] - slab_lo[r]
        own = [o for o in offsets if a <= o and (o < b or r == last_owner)] if n else []
        groups = [own[i:i + per] for i in range(0, len(own), per)]
        bounds = ([0] + [g[0] - slab_lo[r] for g in groups[1:]] + [n]) if groups else [0]
        plans.append((groups, bounds))
    G = max(len(p[0]) for p in plans)
    groups, bounds = plans[rank]
    my_len = slab_hi[rank] - slab_lo[rank]
    nccl = dist.get_backend(group) == "nccl"
    if kind == "segments":
        valid = _apply._leaf_valid_length(sub, segment_length, segment)
        i_lo, i_hi = intervals[rank]
        w_lo = max(0, i_lo - valid)
        w_hi = min(length, (length if rank == last_owner else i_hi + segment_length) + 2 * valid)
        window = mix[:, :, w_lo:w_hi].to(device=device, dtype=torch.float32).contiguous() if my_len else None
    else:
        padded_mix = _apply.tensor_chunk(mix).padded(length)
    sub.to(device)
    sub.eval()
    for _ in range(rng_draws_per_forward(sub) * len(offsets)):
        random.randrange(1)          # transformer.py:680, once per segment forward of the pass, on EVERY rank (see apply_model_sharded)
    arg = _apply._with(cb_arg, model_idx_in_bag=0, shift_idx=0)

    def fire(state, v):
        if callback is not None:
            with lock:
                callback(_apply._with(arg, segment_offset=v, state=state))

    v0 = slab_lo[rank]
    parts = [torch.zeros(rows, my_len, device=device, dtype=torch.float32) for _ in range(batch)] if my_len else []
    pending = []                    # (work, send buffer, receive buffer)
    for g in range(G):
        if g < len(groups):
            for b in range(batch):
                listen = b == 0 and callback is not None
                on_start, on_end = ((lambda v: fire("start", v)), (lambda v: fire("end", v))) if listen else (None, None)
                if kind == "segments":
                    _apply.device_split_accumulate(sub, window[b], 0, length, groups[g], segment_length, valid, weight, parts[b], v0,
                                                   on_start, on_end, draw_rng=False, base_origin=w_lo)
                else:
                    kw = dict(shifts=0, split=False, overlap=overlap, transition_power=transition_power, device=device, segment=segment)
                    with no_sharding():
                        for v in groups[g]:
                            chunk = _apply.TensorChunk(padded_mix, v, segment_length)
                            if b == 0:
                                fire("start", v)
                            state = random.getstate()                     # the pass's draws were taken above, on every rank
                            out = _apply.apply_model(sub, chunk, **kw)[b].reshape(rows, -1).to(device)
                            random.setstate(state)
                            if b == 0:
                                fire("end", v)
                            lo = v - v0
                            a, c = max(lo, 0), min(lo + out.shape[-1], my_len)
                            parts[b][:, a:c] += weight[a - lo:c - lo] * out[:, a - lo:c - lo]
        sizes = [p[1][g + 1] - p[1][g] if g < len(p[0]) else 0 for p in plans]
        width = max(sizes)
        send = torch.empty(batch, rows, width, device=device, dtype=torch.float32)
        if sizes[rank]:
            for b in range(batch):
                send[b, :, :sizes[rank]] = parts[b][:, bounds[g]:bounds[g + 1]]
        recv = torch.empty(world, batch, rows, width, device=device, dtype=torch.float32)
        if nccl:
            work = dist.all_gather_into_tensor(recv, send, group=group, async_op=True)      # RCCL over xGMI, under the next forward
        else:
            work = dist.all_gather(list(recv.unbind(0)), send, group=group, async_op=True)
        pending.append((work, send, recv))
    del parts
    ev1 = _mark(device)
    for work, _, _ in pending:
        work.wait()
    ev2 = _mark(device)
    # stitch in rank order, chunks in order: a piece only overlaps what lies in front of it (the previous rank's tail)
    total = torch.empty(batch, rows, length, device=device, dtype=torch.float32)
    done = 0
    for r, (r_groups, r_bounds) in enumerate(plans):
        for g in range(len(r_groups)):
            s0, s1 = slab_lo[r] + r_bounds[g], slab_lo[r] + r_bounds[g + 1]
            if s1 <= s0:
                continue
            src = pending[g][2][r]
            if s0 > done:
                total[:, :, done:s0].zero_()
                done = s0
            head = min(done, s1)
            if head > s0:
                total[:, :, s0:head] += src[:, :, :head - s0]
            if s1 > head:
                total[:, :, head:s1] = src[:, :, head - s0:s1 - s0]
            done = max(done, s1)
    if done < length:
        total[:, :, done:].zero_()
    del pending
    engine = kind == "segments"
    for b in range(batch):
        _normalise(total[b], 0, length, offsets, segment_length, weight, engine)
    if ev0 is not None:
        _timing.append((ev0, ev1, ev2, _mark(device)))
    return total.view(batch, S, channels, length)


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
