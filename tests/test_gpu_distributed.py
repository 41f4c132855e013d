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


def _worker(rank, world, port, length, out_path, case):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if case.startswith("hdemucs"):
        # The BLSTM recurrence is ONE persistent kernel per sequence whose workgroups wait for each other: with the ranks of this
        # test time-slicing ONE GPU, another rank's persistent kernel can keep part of a grid from becoming resident until the
        # kernel's 0.3 s time-out abandons the sequence (one process per GPU is the supported deployment; the engine
        # raises from mi_hmodel_status / the next forward).  What this test checks is the sharded scheduler, so the ranks use the
        # one-launch-per-step recurrence, which tests/test_gpu_lstm.py holds bit-identical to the persistent kernel.
        os.environ["MI_LSTM_STEPS"] = "1"
    import random
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from demucs_amd import apply as P
    from demucs_amd.distributed import no_sharding
    from demucs_amd.htdemucs import HTDemucs
    from demucs_amd.synth import synth_mix
    from demucs_amd.weights import HTDemucsConfig, synthetic_state_dict
    cfg = HTDemucsConfig()

    def make(seed, max_batch):
        m = HTDemucs(cfg.sources, max_batch=max_batch)
        m.load_state_dict(synthetic_state_dict(cfg, seed))
        return m
    if case == "plain":                                    # one pass: bit-identical
        m = make(4, 2)
        mix = torch.from_numpy(synth_mix(50, length, "tones"))[None].cuda()
        got = P.apply_model(m, mix, shifts=0, split=True, overlap=0.25)            # routed to the sharded engine by itself
        with no_sharding():
            want = P.apply_model(m, mix, shifts=0, split=True, overlap=0.25)
        tol = 0.0
    elif case == "bag_shifts":                             # 2 models x 2 shifts, host mix: last-bit differences at the seams only
        bag = P.BagOfModels([make(10, 2), make(11, 2)], [[1.0, 0.0, 0.5, 1.0], [0.0, 1.0, 0.5, 0.25]])
        mix = torch.from_numpy(synth_mix(51, length, "noise"))[None]
        random.seed(3)
        got = P.apply_model(bag, mix, shifts=2, split=True, overlap=0.25, device="cuda")
        random.seed(3)
        with no_sharding():
            want = P.apply_model(bag, mix, shifts=2, split=True, overlap=0.25, device="cuda")
        assert got.device.type == "cpu"
        tol = 2e-6
    elif case == "shift1":                                 # apply_model's default shifts=1: per-pass normalisation (one shifted pass)
        m = make(6, 2)
        mix = torch.from_numpy(synth_mix(52, length, "tones"))[None]
        random.seed(11)
        got = P.apply_model(m, mix, split=True, overlap=0.25, device="cuda")
        random.seed(11)
        with no_sharding():
            want = P.apply_model(m, mix, split=True, overlap=0.25, device="cuda")
        tol = 2e-6
    elif case in ("hdemucs", "hdemucs_shifts"):            # BASELINE configs[4]: hdemucs_mmi (bag of one, segment 44) on the ragged route
        from demucs_amd.hdemucs import HDemucs
        from demucs_amd.hdemucs_weights import HDemucsConfig, synthetic_hdemucs_state_dict
        hcfg = HDemucsConfig()
        # float32 on purpose: with the ranks of this test sharing ONE GPU, another process's 16-bit matrix-core kernels corrupt this
        # package's (tools/micro/mfma_neighbour.hip: fp32 MFMA never, 16-bit MFMA with VGPR accumulators always; DESIGN.md S8,
        # INTEGRATION.md "do not share the GPU") -- the f16 run of this case passed twice and failed once (7.5e-3) on unchanged
        # code.  The f16 engine goes through the sharded scheduler with one process on the GPU in tests/test_gpu_nccl.py.
        hm = HDemucs(hcfg.sources, max_batch=3, compute_dtype="f32")
        hm.load_state_dict(synthetic_hdemucs_state_dict(hcfg, 1))
        if case == "hdemucs":                              # one pass of a plain model: bit-identical
            hm.segment = 44
            mix = torch.from_numpy(synth_mix(53, length, "noise"))[None].cuda()
            got = P.apply_model(hm, mix, shifts=0, split=True, overlap=0.25)
            with no_sharding():
                want = P.apply_model(hm, mix, shifts=0, split=True, overlap=0.25)
            tol = 0.0
        else:                                              # shifts=2 from a host mix, seeded: HDemucs draws nothing per forward
            bag = P.BagOfModels([hm], [[1.0, 0.5, 2.0, 1.0]], segment=4)
            mix = torch.from_numpy(synth_mix(54, length, "tones"))[None]
            random.seed(21)
            got = P.apply_model(bag, mix, shifts=2, split=True, overlap=0.25, device="cuda", segment=4)
            state_after = random.getstate()
            random.seed(21)
            with no_sharding():
                want = P.apply_model(bag, mix, shifts=2, split=True, overlap=0.25, device="cuda", segment=4)
            assert rank != 0 or state_after == random.getstate(), "sharded HDemucs pass consumed a different RNG stream"
            tol = 2e-6
    else:                                                  # BASELINE configs[3]: the 60-minute track, 616 segments over the ranks
        m = make(0, 16)
        gen = torch.Generator(device="cuda").manual_seed(4)
        mix = torch.randn(1, 2, length, device="cuda", generator=gen) * 0.1
        got = P.apply_model(m, mix, shifts=0, split=True, overlap=0.25)
        with no_sharding():
            want = P.apply_model(m, mix, shifts=0, split=True, overlap=0.25)
        tol = 0.0
    diff = float((got - want).abs().max())
    ok = got.shape == want.shape and diff <= tol * max(1.0, float(want.abs().max())) and bool(torch.isfinite(got[..., ::97]).all())
    flags, diffs = [None] * world, [None] * world
    dist.all_gather_object(flags, bool(ok))
    dist.all_gather_object(diffs, diff)
    if rank == 0:
        torch.save(dict(ok=all(flags), shape=tuple(got.shape), max_abs_diff=max(diffs)), out_path)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,length,case", [(2, int(4.2 * SL), "plain"), (3, 2 * 257985 + 17, "plain"),
                                               (2, int(3.3 * SL), "bag_shifts"), (2, int(3.3 * SL), "shift1"),
                                               (2, 180 * 44100, "hdemucs"), (3, int(13.7 * 44100), "hdemucs_shifts"),
                                               (2, 3600 * 44100, "sixty_minutes")])
def test_sharded_engine_equals_single_process(tmp_path, world, length, case):
    out_path = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(world, _free_port(), length, out_path, case), nprocs=world, join=True)
    res = torch.load(out_path)
    assert res["ok"] and res["shape"] == (1, 4, 2, length), f"sharded != single process, max |diff| = {res['max_abs_diff']:.3e}"
