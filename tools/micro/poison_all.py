"""Single process, deterministic: is any engine result changed when EVERY VGPR / AGPR / LDS word of the chip is overwritten
with a NaN (or all-ones, or zero) pattern between ALL of the engine's kernel launches?  (mi_debug_set_post_launch_hook.)
A kernel that consumes a register or LDS word it never wrote -- the ordinary explanation for "results change when another
process shares the GPU" -- shows up here as a mismatch, and the taps name the first stage that differs."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from demucs_amd.htdemucs import HTDemucs
from demucs_amd.weights import HTDemucsConfig, synthetic_state_dict
from demucs_amd.synth import synth_mix
from demucs_amd import _lib
P = C.CDLL(os.path.join(ROOT, "tools", "micro", "libpoison_all.so"))
P.poison_set_pattern.argtypes = [C.c_uint]
hook = C.cast(P.poison_all, C.c_void_p)
lib = _lib.load()
cfg = HTDemucsConfig()
TAPS = ["x0", "xt0", "enc0", "tenc0", "enc1", "tenc1", "enc2", "tenc2", "enc3", "tenc3", "tr_f", "tr_t", "yspec", "ytime"]
for dtype in ("f32", "bf16"):
    m = HTDemucs(cfg.sources, max_batch=2, compute_dtype=dtype); m.load_state_dict(synthetic_state_dict(cfg, 4)); m.to("cuda")
    x = torch.stack([torch.from_numpy(synth_mix(50 + i, 343980, "tones")) for i in range(2)]).cuda()
    ref = m.forward_segments(x).clone()
    ref_taps = {t: m.tap(t, 2).clone() for t in TAPS}
    for pat in (0x7FC07FC0, 0xFFFFFFFF, 0x00000000, 0x3F803F80):
        P.poison_set_pattern(pat)
        lib.mi_debug_set_post_launch_hook(hook)
        P.poison_all(C.c_void_p(_lib.current_stream_ptr()))
        y = m.forward_segments(x).clone()
        lib.mi_debug_set_post_launch_hook(None)
        torch.cuda.synchronize()
        bad = [t for t in TAPS if not torch.equal(m.tap(t, 2), ref_taps[t])]
        print(f"{dtype} pattern {pat:08x}: output equal {torch.equal(y, ref)}  max diff {float((y - ref).abs().nan_to_num(nan=9e9).max()):.3e}  "
              f"nans {int(torch.isnan(y).sum())}  first differing taps {bad[:4]}", flush=True)
    m.release()
print('poison kernel status (0 = every launch accepted):', P.poison_status())
