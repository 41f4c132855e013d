"""hdemucs_mmi architecture, 3-minute track in 44-second chunks: median wall time of apply_model (ms).  python tools/micro/hdemucs_time.py [dtype] [runs]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from demucs_amd import apply as P  # noqa: E402
from demucs_amd.hdemucs import HDemucs  # noqa: E402
from demucs_amd.hdemucs_weights import HDemucsConfig, synthetic_hdemucs_state_dict  # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else "f16"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 7
cfg = HDemucsConfig()
m = HDemucs(cfg.sources, max_batch=5, compute_dtype=dt)
m.load_state_dict(synthetic_hdemucs_state_dict(cfg, 0))
m.to("cuda")
bag = P.BagOfModels([m], segment=44)
mix = torch.randn(1, 2, 180 * 44100, device="cuda") * 0.1
ts = []
for i in range(runs + 2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = P.apply_model(bag, mix, shifts=0, overlap=0.25)
    torch.cuda.synchronize()
    if i >= 2:
        ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print(f"hdemucs_mmi {dt}: median {ts[len(ts) // 2]:.2f} ms  (min {ts[0]:.2f}, max {ts[-1]:.2f}; MI_LSTM_RB={os.environ.get('MI_LSTM_RB', '8')})", flush=True)
