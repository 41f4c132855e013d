"""First-run safety of the RCCL lines: on the one-GPU test box a process group of ONE rank is initialised with
backend "nccl" (= RCCL on ROCm) in a fresh child process and the sharded scheduler is driven through its forced-collective
switch, so that `broadcast`, `all_gather_into_tensor`, `all_reduce(MAX)` and `barrier` all execute on RCCL with the same
tensors, dtypes and call order the driver's 8-GPU run uses; `bench.py`'s N > 1 branch runs under `torch.distributed.run`."""
import json
import os
import socket
import subprocess
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


def _worker(rank, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import random
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)          # before any other GPU work of this process
    from demucs_amd import apply as P
    from demucs_amd.distributed import force_collectives, no_sharding, sharding_active
    from demucs_amd.hdemucs import HDemucs
    from demucs_amd.hdemucs_weights import HDemucsConfig, synthetic_hdemucs_state_dict
    from demucs_amd.htdemucs import HTDemucs
    from demucs_amd.synth import synth_mix
    from demucs_amd.weights import HTDemucsConfig, synthetic_state_dict
    res = {}
    assert dist.get_backend() == "nccl" and not sharding_active()
    cfg = HTDemucsConfig()
    m = HTDemucs(cfg.sources, max_batch=4)
    m.load_state_dict(synthetic_state_dict(cfg, 4))
    mix = torch.from_numpy(synth_mix(50, int(2.6 * SL), "tones"))[None].cuda()
    with force_collectives():
        assert sharding_active()
        got = P.apply_model(m, mix, shifts=0, split=True, overlap=0.25)           # all_gather_into_tensor
        random.seed(8)
        got_s = P.apply_model(m, mix.cpu(), shifts=1, split=True, overlap=0.25, device="cuda")   # + broadcast of the shift offset
    with no_sharding():
        want = P.apply_model(m, mix, shifts=0, split=True, overlap=0.25)
        random.seed(8)
        want_s = P.apply_model(m, mix.cpu(), shifts=1, split=True, overlap=0.25, device="cuda")
    res["plain_equal"] = bool(torch.equal(got, want))
    res["shift1_diff"] = float((got_s - want_s).abs().max())
    hcfg = HDemucsConfig()
    hm = HDemucs(hcfg.sources, max_batch=2, compute_dtype="f16")
    hm.load_state_dict(synthetic_hdemucs_state_dict(hcfg, 1))
    hmix = torch.from_numpy(synth_mix(53, 20 * 44100, "noise"))[None].cuda()
    with force_collectives():
        hgot = P.apply_model(hm, hmix, shifts=0, split=True, overlap=0.25, segment=8)
    with no_sharding():
        hwant = P.apply_model(hm, hmix, shifts=0, split=True, overlap=0.25, segment=8)
    res["hdemucs_equal"] = bool(torch.equal(hgot, hwant))
    t = torch.tensor([1.5], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                                       # bench.py's max-over-ranks
    dist.barrier()
    torch.cuda.synchronize()
    res["all_reduce"] = float(t.item())
    torch.save(res, out_path)
    dist.destroy_process_group()


def test_rccl_collectives_execute_at_world_size_one(tmp_path):
    out_path = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(_free_port(), out_path), nprocs=1, join=True)
    res = torch.load(out_path)
    assert res["plain_equal"] and res["hdemucs_equal"] and res["shift1_diff"] <= 2e-6 and res["all_reduce"] == 1.5, res


def test_bench_multi_gpu_branch_under_torchrun():
    """`bench.py`'s N > 1 branch (nccl init with device_id, sharded steps, barrier fences, all_reduce(MAX) of the timing,
    strong-scaling line) launched exactly as the driver launches it, with one rank and --force-dist, on a 2-minute track."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--seconds", "120",
           "--steps", "2", "--warmup", "1", "--batch", "16"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    r = json.loads(line)
    assert r["scaling"] == "strong" and r["n_gpus"] == 1 and r["value"] > 200 and "roofline" in r
    assert "RCCL all-gather" in r["config"]["parallelism"]


def test_bench_launches_its_own_ranks_without_torchrun():
    """`python bench.py --gpus N` with NO torchrun environment (the shape of the driver's N = 1 command) must start its ranks
    itself: a child `torch.distributed.run` started before the parent makes any GPU call, the JSON line relayed, the child's
    exit code returned.  `--spawn` takes that path with ONE rank (this box has one GPU); `--force-dist` makes the rank run the
    N > 1 branch (RCCL group, sharded steps, strong-scaling line)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--force-dist", "--seconds", "120", "--steps", "2",
           "--warmup", "1", "--batch", "16"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), lines[:5]      # stdout is the ONE JSON line (RCCL's banner goes to stderr)
    r = json.loads(lines[0])
    assert r["scaling"] == "strong" and r["n_gpus"] == 1 and r["value"] > 200
    ph = r["sharded_step_phases_rank0"]
    assert ph["calls"] == 2 and ph["segments_ms"] > 0 and ph["all_gather_ms"] >= 0 and ph["stitch_ms"] > 0
