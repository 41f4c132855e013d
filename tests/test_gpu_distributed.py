"""Sharded multi-rank path on the real engine: two processes share the one GPU of the test box
(gloo process group carrying CUDA tensors; RCCL needs one GPU per rank, which the driver's 8-GPU
node provides) and must reproduce the single-process result bit for bit."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SL = 343980


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, length, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from demucs_amd import apply as P
    from demucs_amd.distributed import apply_model_sharded
    from demucs_amd.htdemucs import HTDemucs
    from demucs_amd.synth import synth_mix
    from demucs_amd.weights import HTDemucsConfig, synthetic_state_dict
    cfg = HTDemucsConfig()
    m = HTDemucs(cfg.sources, max_batch=2)
    m.load_state_dict(synthetic_state_dict(cfg, 4))
    mix = torch.from_numpy(synth_mix(50, length, "tones"))[None].cuda()
    got = apply_model_sharded(m, mix, overlap=0.25)
    want = P.apply_model(m, mix, shifts=0, split=True, overlap=0.25)
    ok = bool(torch.equal(got, want)) and got.device.type == "cuda"
    flags, diffs = [None] * world, [None] * world
    dist.all_gather_object(flags, ok)
    dist.all_gather_object(diffs, float((got - want).abs().max()))
    if rank == 0:
        torch.save(dict(ok=all(flags), shape=tuple(got.shape), max_abs_diff=max(diffs)), out_path)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,length", [(2, int(4.2 * SL)), (3, 2 * 257985 + 17)])
def test_sharded_engine_equals_single_process(tmp_path, world, length):
    out_path = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(world, _free_port(), length, out_path), nprocs=world, join=True)
    res = torch.load(out_path)
    assert res["ok"] and res["shape"] == (1, 4, 2, length), f"sharded != single process, max |diff| = {res['max_abs_diff']:.3e}"
