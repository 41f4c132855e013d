"""Single process: does any split-bf16 / fp32 GEMM result change when the LDS of all CUs holds NaN patterns beforehand?"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from demucs_amd.htdemucs import HTDemucs
from demucs_amd.weights import HTDemucsConfig, synthetic_state_dict
from demucs_amd.synth import synth_mix
from demucs_amd import _lib
P = C.CDLL(os.path.join(ROOT, "tools", "micro", "liblds_poison.so"))
P.lds_poison.argtypes = [C.c_uint, C.c_void_p]
cfg = HTDemucsConfig()
m = HTDemucs(cfg.sources, max_batch=2); m.load_state_dict(synthetic_state_dict(cfg, 4)); m.to("cuda")
x = torch.stack([torch.from_numpy(synth_mix(50 + i, 343980, "tones")) for i in range(2)]).cuda()
ref = m.forward_segments(x).clone()
for pat in (0x7FC07FC0, 0xFFFFFFFF, 0x7F807F80, 0x00000000):
    P.lds_poison(pat, C.c_void_p(_lib.current_stream_ptr()))
    y = m.forward_segments(x).clone()
    torch.cuda.synchronize()
    print(f"pattern {pat:08x}: equal {torch.equal(y, ref)}  max diff {float((y - ref).abs().nan_to_num(nan=9e9).max()):.3e}  nans {int(torch.isnan(y).sum())}")
