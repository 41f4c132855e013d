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


def _worker(rank, world, port, length, overlap, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from demucs_amd import apply as P
    from demucs_amd.distributed import apply_model_sharded
    from test_apply_host import ToyModel
    mix = torch.randn(1, 2, length, generator=torch.Generator().manual_seed(7))
    got = apply_model_sharded(ToyModel(), mix, overlap=overlap)
    want = P.apply_model(ToyModel(), mix, shifts=0, split=True, overlap=overlap)
    ok = torch.equal(got, want)
    flags = [None] * world
    dist.all_gather_object(flags, ok)
    if rank == 0:
        torch.save(dict(ok=all(flags), shape=tuple(got.shape)), out_path)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,length,overlap", [(2, 2500, 0.25), (2, 300, 0.25), (3, 4001, 0.1)])
def test_sharded_equals_single_process(tmp_path, world, length, overlap):
    out_path = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(world, _free_port(), length, overlap, out_path), nprocs=world, join=True)
    res = torch.load(out_path)
    assert res["ok"] and res["shape"] == (1, 3, 2, length)


def test_shard_ranges():
    assert shard_ranges(31, 8) == [(0, 4), (4, 8), (8, 12), (12, 16), (16, 20), (20, 24), (24, 28), (28, 31)]
    assert shard_ranges(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert shard_ranges(616, 8)[-1] == (539, 616)
