"""Host logic of bench.py that needs no GPU: `--gpus N` without a torchrun environment starts its own ranks as a CHILD
`torch.distributed.run` (never an exec), relays rank 0's JSON line and returns the child's exit code."""
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_self_launch_builds_the_torchrun_child(monkeypatch, capsys):
    import bench
    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"], seen["kw"] = cmd, kw
        return types.SimpleNamespace(returncode=7, stdout='{"metric": "x"}\n')

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1", "--spawn"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                   # the child's exit code is the parent's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "8", "--steps", "3", "--warmup", "1"]          # --spawn is not handed down
    assert seen["kw"]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"            # dmabuf IPC for RCCL on this pool
    assert capsys.readouterr().out.strip() == '{"metric": "x"}'             # the JSON line is relayed


def test_no_self_launch_under_torchrun(monkeypatch):
    """With WORLD_SIZE in the environment (the driver's torchrun form) the process is a rank: it must not spawn."""
    import bench
    called = []
    monkeypatch.setattr(bench, "self_launch", lambda n: called.append(n))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    with pytest.raises(Exception):                             # no GPU here: the rank fails at its first device call ...
        bench.main()
    assert called == []                                        # ... but never tried to launch ranks of its own
