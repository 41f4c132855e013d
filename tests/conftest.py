import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _usable_cores():
    """Threads the CPU side of the tests (the oracles) may use: the affinity mask capped by the cgroup CPU quota and by 16.  A GPU
    box shows all 256 logical CPUs of its host to a job that owns a 16-core share: torch's default of one thread per visible CPU
    makes every small float64 matmul of an oracle a 256-thread fork/join against the quota (the 400-step LSTM reference took 29 s)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    n = _usable_cores()
    os.environ.setdefault("OMP_NUM_THREADS", str(n))          # child processes (re-runs under other switches, rank launches)
    try:
        import torch
        if torch.get_num_threads() > n:
            torch.set_num_threads(n)
    except ImportError:
        pass


class Golden:
    """A tests/golden/*.npz fixture written by tools/make_golden.py (outputs of the imported
    reference): per tap a strided sample plus float64 sum / sum of squares."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))

    def meta(self, key, default=None):
        k = "meta/" + key
        return self.z[k] if k in self.z.files else default

    def taps(self, tag):
        return sorted({k.split("/")[1] for k in self.z.files if k.startswith(tag + "/")})

    def check(self, tag, tap, tensor, atol, rtol=0.0):
        """Compare `tensor` (torch or numpy, full resolution) with the stored sample and moments.
        Returns the max-abs error on the sample."""
        import torch
        t = tensor.detach().cpu() if isinstance(tensor, torch.Tensor) else torch.from_numpy(np.asarray(tensor))
        p = f"{tag}/{tap}"
        shape = tuple(int(v) for v in self.z[p + "/shape"])
        assert tuple(t.shape) == shape, (tap, tuple(t.shape), shape)
        stride = int(self.z[p + "/stride"])
        got = t.reshape(-1)[::stride].double().numpy()
        want = self.z[p + "/sample"].astype(np.float64)
        err = np.abs(got - want)
        tol = atol + rtol * np.abs(want)
        assert (err <= tol).all(), f"{tap}: max err {err.max():.3e} > tol (atol {atol:g}, rtol {rtol:g})"
        n = t.numel()
        rms = float(np.sqrt(float(self.z[p + "/sumsq"]) / n))
        d = t.double()
        assert abs(float(d.sum()) - float(self.z[p + "/sum"])) <= (atol + rtol * rms) * n, f"{tap}: sum mismatch"
        assert abs(float((d * d).sum()) - float(self.z[p + "/sumsq"])) <= 4 * (atol + rtol * rms) * rms * n + 1e-30, f"{tap}: sumsq mismatch"
        return float(err.max())


@pytest.fixture(scope="session")
def golden():
    return Golden
