import sys, torch
sys.path.insert(0, ".")
from demucs_amd.htdemucs import HTDemucs
from demucs_amd.weights import HTDemucsConfig, synthetic_state_dict
from demucs_amd.synth import synth_mix
cfg = HTDemucsConfig()
m = HTDemucs(cfg.sources, max_batch=2)
m.load_state_dict(synthetic_state_dict(cfg, 4)); m.to("cuda")
SL = 343980
x = torch.from_numpy(synth_mix(50, 2 * SL, "tones")).reshape(2, 2, SL).contiguous().cuda() if False else torch.stack([torch.from_numpy(synth_mix(50 + i, SL, "tones")) for i in range(2)]).cuda()
a = m.forward_segments(x).clone(); b = m.forward_segments(x).clone()
print("repeat equal:", torch.equal(a, b), (a - b).abs().max().item())
c0 = m.forward_segments(x[:1]).clone(); c1 = m.forward_segments(x[1:]).clone()
print("b=1 vs b=2 item0:", torch.equal(a[:1], c0), (a[:1] - c0).abs().max().item())
print("b=1 vs b=2 item1:", torch.equal(a[1:], c1), (a[1:] - c1).abs().max().item())
for name in ["enc0_preemb", "spec_out", "time_out"]:
    pass
