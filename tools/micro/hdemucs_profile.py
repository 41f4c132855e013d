"""rocprofv3 target: hdemucs_mmi architecture, 3-minute track in 44-second chunks (equal chunks batched), given dtype."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from demucs_amd import apply as P
from demucs_amd.hdemucs import HDemucs
from demucs_amd.hdemucs_weights import HDemucsConfig, synthetic_hdemucs_state_dict
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
cfg = HDemucsConfig()
m = HDemucs(cfg.sources, max_batch=int(sys.argv[2]) if len(sys.argv) > 2 else 5, compute_dtype=dt)
m.load_state_dict(synthetic_hdemucs_state_dict(cfg, 0)); m.to("cuda")
bag = P.BagOfModels([m], segment=44)
mix = torch.randn(1, 2, 180 * 44100, device="cuda") * 0.1
for _ in range(2):
    out = P.apply_model(bag, mix, shifts=0, overlap=0.25)
torch.cuda.synchronize()
