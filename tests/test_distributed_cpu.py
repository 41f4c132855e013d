"""N > 1 path on CPU: world_size-2 (and 3) `gloo` process groups running `apply_model_sharded`
with a cheap stand-in model; must be bit-identical to the single-process scheduler."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from demucs_amd.distributed import shard_ranges

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, length, overlap, out_path, case):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import random
    from demucs_amd import apply as P
    from demucs_amd.distributed import apply_model_sharded, no_sharding
    from test_apply_host import RaggedToy, ToyModel
    batch = 2 if case == "batch2" else 1
    mix = torch.randn(batch, 2, length, generator=torch.Generator().manual_seed(7))
    events = []
    if case in ("plain", "batch2", "ragged", "plain_mb2"):
        Toy = RaggedToy if case == "ragged" else ToyModel        # ragged: no valid_length, every chunk at its own length (HDemucs)
        model = Toy()
        if case == "plain_mb2":
            model.max_batch = 2          # groups of two segments: several pipelined all-gathers per rank, unequal group counts
        got = apply_model_sharded(model, mix, overlap=overlap, callback=lambda d: events.append(dict(d)))
        with no_sharding():
            want = P.apply_model(Toy(), mix, shifts=0, split=True, overlap=overlap)
        ok = torch.equal(got, want)                        # single pass: bit-identical
        ok = ok and all(e["state"] in ("start", "end") for e in events) and len(events) % 2 == 0
    else:
        w = [[1.0, 0.0, 0.5], [0.0, 1.0, 1.5]]
        shifts = {"shift1": 1, "bag1_shift1": 1, "bag1_noshift": 0}.get(case, 2)
        make = {"bag_shifts": lambda: P.BagOfModels([ToyModel(1.0), ToyModel(0.7)], w),
                "bag1_shift1": lambda: P.BagOfModels([ToyModel(0.9)], [[0.5, 2.0, 1.25]]),      # one member, non-unit weights
                "bag1_noshift": lambda: P.BagOfModels([ToyModel(0.9)], [[0.5, 2.0, 1.25]]),
                "ragged_shifts": lambda: RaggedToy()}.get(case, lambda: ToyModel())
        if rank == 1:
            random.seed(99)        # ranks that were NOT seeded alike still agree on the shift offsets (rank 0's are used)
        else:
            random.seed(5)
        got = apply_model_sharded(make(), mix, shifts=shifts, overlap=overlap, callback=lambda d: events.append(dict(d)))
        state_after = random.getstate()
        random.seed(5)
        with no_sharding():
            want = P.apply_model(make(), mix, shifts=shifts, split=True, overlap=overlap)
        # several passes: identical up to the last bit at the seams between two ranks' slabs
        ok = bool((got - want).abs().max() <= 2e-6) and got.shape == want.shape
        if rank == 0:              # a seeded rank consumed exactly the draws of the single-process run
            ok = ok and state_after == random.getstate()
        if case == "bag_shifts":
            ok = ok and {e["model_idx_in_bag"] for e in events} <= {0, 1} and all(e["models"] == 2 for e in events)
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    counts = [None] * world
    dist.all_gather_object(counts, len(events))
    if rank == 0:
        torch.save(dict(ok=all(flags), flags=flags, shape=tuple(got.shape), events=sum(counts)), out_path)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,length,overlap,case", [
    (2, 2500, 0.25, "plain"), (2, 300, 0.25, "plain"), (3, 4001, 0.1, "plain"), (2, 1700, 0.25, "batch2"), (2, 1337, 0.25, "ragged"),
    (2, 5000, 0.25, "plain_mb2"), (3, 5333, 0.25, "plain_mb2"), (3, 650, 0.25, "plain_mb2"),
    (2, 2500, 0.25, "shifts"), (3, 3111, 0.25, "bag_shifts"), (2, 390, 0.25, "bag_shifts"),
    # shifts=1 (the default of apply_model and Separator) and a one-member bag with non-unit weights must NOT take the
    # single-pass shortcut; a model that draws nothing per forward (HDemucs) must not advance the RNG per segment
    (2, 2500, 0.25, "shift1"), (3, 1777, 0.25, "shift1"), (2, 2100, 0.25, "bag1_shift1"), (2, 2100, 0.25, "bag1_noshift"),
    (2, 1337, 0.25, "ragged_shifts"), (3, 2650, 0.25, "ragged_shifts")])
def test_sharded_equals_single_process(tmp_path, world, length, overlap, case):
    out_path = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(world, _free_port(), length, overlap, out_path, case), nprocs=world, join=True)
    res = torch.load(out_path)
    assert res["ok"], res
    assert res["shape"] == (2 if case == "batch2" else 1, 3, 2, length)
    if case in ("plain", "batch2", "ragged", "plain_mb2"):       # every segment fired its start/end pair on exactly one rank
        assert res["events"] == 2 * len(range(0, length, int((1 - overlap) * 400)))


def test_track_intervals_cover_the_track_once():
    from demucs_amd.distributed import track_intervals
    for length, stride, world in [(7938000, 257985, 8), (158760000, 257985, 8), (300, 300, 2), (1000, 300, 3), (5, 300, 4)]:
        iv = track_intervals(length, stride, world)
        assert iv[0][0] == 0 and max(b for _, b in iv) == length
        for (a0, b0), (a1, b1) in zip(iv, iv[1:]):
            assert b0 == a1 or (a1 == b1 == length)
        assert all(a % stride == 0 or a == length for a, _ in iv)


def test_shard_ranges():
    assert shard_ranges(31, 8) == [(0, 4), (4, 8), (8, 12), (12, 16), (16, 20), (20, 24), (24, 28), (28, 31)]
    assert shard_ranges(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert shard_ranges(616, 8)[-1] == (539, 616)
