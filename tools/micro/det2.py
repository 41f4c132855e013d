import sys, torch
sys.path.insert(0, ".")
from demucs_amd import apply as P
from demucs_amd.htdemucs import HTDemucs
from demucs_amd.weights import HTDemucsConfig, synthetic_state_dict
from demucs_amd.synth import synth_mix
cfg = HTDemucsConfig()
m = HTDemucs(cfg.sources, max_batch=2)
m.load_state_dict(synthetic_state_dict(cfg, 4)); m.to("cuda")
SL = 343980
x = torch.stack([torch.from_numpy(synth_mix(50 + i, SL, "tones")) for i in range(2)]).cuda()
c1 = m.forward_segments(x[1:]).clone()          # very first call: B = 1
a = m.forward_segments(x).clone()
c0 = m.forward_segments(x[:1]).clone()
print("first-call b=1 item1 vs b=2:", torch.equal(a[1:], c1), (a[1:] - c1).abs().max().item())
print("later b=1 item0 vs b=2:", torch.equal(a[:1], c0), (a[:1] - c0).abs().max().item())
# short ragged segment (17 valid samples, zero padded), alone vs second in a batch
z = torch.zeros(1, 2, SL, device="cuda"); z[..., :17] = x[0:1, :, :17]
zz = torch.cat([x[:1], z])
r1 = m.forward_segments(z).clone(); r2 = m.forward_segments(zz).clone()
print("ragged alone vs pos1:", torch.equal(r1, r2[1:]), (r1 - r2[1:]).abs().max().item())
for name in ("enc0_preemb",):
    pass
mix = torch.from_numpy(synth_mix(50, 2 * 257985 + 17, "tones"))[None].cuda()
w1 = P.apply_model(m, mix, shifts=0, split=True, overlap=0.25)
m2 = HTDemucs(cfg.sources, max_batch=1); m2.load_state_dict(synthetic_state_dict(cfg, 4)); m2.to("cuda")
w2 = P.apply_model(m2, mix, shifts=0, split=True, overlap=0.25)
print("apply max_batch 2 vs 1:", torch.equal(w1, w2), (w1 - w2).abs().max().item())
