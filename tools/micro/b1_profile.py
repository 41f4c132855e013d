"""rocprofv3 target: htdemucs forwards at B = 1 (float32): per-launch durations of a small-batch forward."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from demucs_amd.htdemucs import HTDemucs  # noqa: E402
from demucs_amd.weights import HTDemucsConfig, synthetic_state_dict  # noqa: E402

cfg = HTDemucsConfig()
m = HTDemucs(cfg.sources, max_batch=1, compute_dtype=sys.argv[1] if len(sys.argv) > 1 else "f32")
m.load_state_dict(synthetic_state_dict(cfg, 0))
m.to("cuda")
x = torch.randn(1, 2, 343980, device="cuda") * 0.1
for _ in range(4):
    y = m(x)
torch.cuda.synchronize()
