"""Track-level parity of `demucs_amd.apply.apply_model` on a real MI355X: against the reference's
`apply_model` outputs (tests/golden/apply_*.npz, float64 and float32 reference runs) and
against this package's own generic per-segment route (bit-identical stitching)."""
import random

import numpy as np
import pytest
import torch

from demucs_amd import apply as P
from demucs_amd.htdemucs import HTDemucs
from demucs_amd.synth import synth_mix
from demucs_amd.weights import HTDemucsConfig, synthetic_state_dict

pytestmark = pytest.mark.gpu
SL = 343980
TOL = 1e-4          # north_star: <= 1e-4 max-abs sample deviation from the CPU reference

CASES = {
    "apply_one_segment": dict(wseeds=[0], mix=lambda: synth_mix(1, SL, "noise")),
    "apply_2p3_segments": dict(wseeds=[0], mix=lambda: synth_mix(2, int(2.3 * SL), "tones")),
    "apply_sl_plus_1": dict(wseeds=[1], mix=lambda: synth_mix(3, SL + 1, "noise")),
    "apply_shifts2": dict(wseeds=[0], mix=lambda: synth_mix(4, 300000, "tones")),
    "apply_bag2_shift1": dict(wseeds=[10, 11], mix=lambda: synth_mix(5, 400000, "noise")),
    "apply_bag4_onehot_shifts2": dict(wseeds=[10, 11, 12, 13], mix=lambda: synth_mix(9, 280000, "tones")),
    "apply_nosplit_short": dict(wseeds=[0], mix=lambda: synth_mix(6, 200000, "tones")),
    "apply_overlap10_tp2": dict(wseeds=[1], mix=lambda: synth_mix(8, int(1.5 * SL), "noise")),
}


def engine(wseed, max_batch=4):
    cfg = HTDemucsConfig()
    m = HTDemucs(cfg.sources, max_batch=max_batch)
    m.load_state_dict(synthetic_state_dict(cfg, wseed))
    return m


def golden_kwargs(g):
    return {k[len("meta/kw_"):]: g.z[k].item() for k in g.z.files if k.startswith("meta/kw_")}


@pytest.mark.parametrize("name", list(CASES))
def test_apply_model_matches_reference(golden, name):
    case, g = CASES[name], golden(name)
    kw = golden_kwargs(g)
    models = [engine(s) for s in case["wseeds"]]
    bw = g.meta("bag_weights")
    model = models[0] if bw is None else P.BagOfModels(models, bw.tolist())
    if g.meta("rseed") is not None:
        random.seed(int(g.meta("rseed")))
    mix = torch.from_numpy(case["mix"]())[None]
    mix0 = mix.clone()
    events = []
    out = P.apply_model(model, mix, device="cuda", callback=lambda d: events.append(dict(d)), **kw)
    # the split branch returns on mix.device; the bare leaf returns on `device` (apply.py:259,312-322)
    assert out.device.type == ("cpu" if kw.get("split", True) else "cuda")
    out = out.cpu()
    assert out.dtype == torch.float32 and torch.equal(mix, mix0)
    e64 = g.check("f64", "out", out, atol=TOL)
    e32 = g.check("f32", "out", out, atol=TOL)
    keys = ["model_idx_in_bag", "shift_idx", "segment_offset", "models", "state"]
    got = np.array([[str(e[k]) for k in keys] for e in events])
    assert got.shape == g.z["events"].shape and (got == g.z["events"]).all()
    print(f"{name}: max-abs vs reference f64 {e64:.2e}, vs reference f32 {e32:.2e}")


class PerSegment:
    """Routes an engine through the generic per-segment scheduler (not an HTDemucs instance)."""

    def __init__(self, m):
        self.m = m
        self.sources, self.samplerate, self.audio_channels, self.segment = m.sources, m.samplerate, m.audio_channels, m.segment

    def to(self, d):
        self.m.to(d)
        return self

    def eval(self):
        return self

    def valid_length(self, n):
        return self.m.valid_length(n)

    def __call__(self, x):
        return self.m(x)


@pytest.mark.parametrize("length,max_batch", [(int(3.4 * SL), 3), (SL // 2, 2), (2 * 257985 + 5, 8)])
def test_device_scheduler_is_bit_identical_to_per_segment_route(length, max_batch):
    """Batched forward + device overlap-add == the reference-ordered sequential loop on the same
    engine, for a mix living on the GPU (result stays on the GPU) and a batch of 2 tracks."""
    m = engine(2, max_batch)
    mix = torch.stack([torch.from_numpy(synth_mix(31, length, "tones")), torch.from_numpy(synth_mix(32, length, "noise"))]).cuda()
    fast = P.apply_model(m, mix, shifts=0, overlap=0.25)
    slow = P.apply_model(PerSegment(m), mix, shifts=0, overlap=0.25)
    assert fast.device.type == "cuda" and fast.shape == (2, 4, 2, length)
    assert torch.equal(fast, slow)


def test_more_tracks_than_max_batch_still_fills_the_batch():
    """A batch of 3 mixes through an engine with max_batch = 2: the tracks go one at a time, each forward still carries
    max_batch segments (4 segments per track -> 2 forwards per track, 6 in all; round 3 ran 12 one-segment forwards), and the
    result equals the per-track calls bit for bit."""
    m = engine(2, 2)
    length = int(2.3 * SL) + 7
    mix = torch.stack([torch.from_numpy(synth_mix(60 + k, length, "tones" if k & 1 else "noise")) for k in range(3)]).cuda()
    calls = []
    inner = m.forward_segments
    m.forward_segments = lambda seg, out: (calls.append(seg.shape[0]), inner(seg, out))[1]
    try:
        fast = P.apply_model(m, mix, shifts=0, overlap=0.25)
    finally:
        del m.forward_segments
    assert calls == [2] * 6, calls
    for k in range(3):
        assert torch.equal(fast[k], P.apply_model(m, mix[k:k + 1], shifts=0, overlap=0.25)[0])


def test_shorter_segment_override_and_errors():
    """`segment=5`: the leaf window is int(5 * sr) = 220500 samples, centred on the chunk; the model
    right-pads it to its training length (apply.py:304-305, htdemucs.py:534-537)."""
    m = engine(0, 2)
    mix = torch.from_numpy(synth_mix(40, 500000, "tones"))[None].cuda()
    fast = P.apply_model(m, mix, shifts=0, segment=5)
    seg_len, length = 220500, mix.shape[-1]
    stride = int(0.75 * seg_len)
    weight = P._transition_weight(seg_len, 1.0, mix.device)
    want = torch.zeros(1, 4, 2, length, device=mix.device)
    sw = torch.zeros(length, device=mix.device)
    for off in range(0, length, stride):
        chunk = P.TensorChunk(mix, off, seg_len)
        o = P.center_trim(m(chunk.padded(seg_len)), chunk.length)
        want[..., off:off + seg_len] += weight[:chunk.length] * o
        sw[off:off + seg_len] += weight[:chunk.length]
    want /= sw
    assert torch.equal(fast, want)
    with pytest.raises(ValueError):
        P.apply_model(m, mix, shifts=0, segment=9)              # longer than the training length (htdemucs.py:521-524)


def test_full_size_track_properties():
    """BASELINE configs[1] at full size (3-minute track, 31 segments, one 31-segment batched forward):
    finite, bit-identical to the per-segment route, and two randomly chosen segments of the stitched result
    agree with the float64 oracle run on those segments alone (interior samples that only they cover)."""
    from oracle import htdemucs_oracle as O
    cfg = HTDemucsConfig()
    sd = synthetic_state_dict(cfg, 0)
    m = HTDemucs(cfg.sources, max_batch=31)
    m.load_state_dict(sd)
    length = 180 * 44100
    mix = torch.from_numpy(synth_mix(1, length, "noise"))[None].cuda()
    out = P.apply_model(m, mix, shifts=0, overlap=0.25)
    assert out.shape == (1, 4, 2, length) and bool(torch.isfinite(out).all())
    m2 = HTDemucs(cfg.sources, max_batch=4)
    m2.load_state_dict(sd)
    slow = P.apply_model(PerSegment(m2), mix, shifts=0, overlap=0.25)
    assert torch.equal(out, slow)
    osd = O.to_torch_state(sd, torch.float64)
    stride = int(0.75 * SL)
    for k in (7, 22):
        off = k * stride
        seg = mix[..., off:off + SL].cpu().double()
        with torch.no_grad():
            want = O.htdemucs_forward(osd, seg, 4)
        lo, hi = SL - stride, stride                   # samples covered by segment k only (weight ratio = 1)
        err = (out[..., off + lo:off + hi].cpu().double() - want[..., lo:hi]).abs().max().item()
        assert err <= TOL, (k, err)


def _bag4(max_batch, wrap=lambda m: m, compute_dtype="f32"):
    cfg = HTDemucsConfig()
    models = []
    for seed in (10, 11, 12, 13):
        m = HTDemucs(cfg.sources, max_batch=max_batch, compute_dtype=compute_dtype)
        m.load_state_dict(synthetic_state_dict(cfg, seed))
        models.append(wrap(m.to("cuda")))
    onehot = [[1.0 if i == k else 0.0 for k in range(4)] for i in range(4)]          # remote/htdemucs_ft.yaml
    return P.BagOfModels(models, onehot)


def test_config3_full_size_bag_of_4_shifts_2():
    """BASELINE configs[2] at full size in float32: bag of 4 (one-hot weights, distinct weights), shifts=2, the 3-minute
    track handed over on the HOST, models already on the GPU: 4 x 2 x 31..32 segment forwards.  The device route (one
    H2D, batched forwards, averages in HBM, one D2H) must equal the reference-ordered per-segment route bit for bit,
    leave `mix` untouched, and each source must come from its own bag member."""
    length = 180 * 44100
    mix = torch.from_numpy(synth_mix(1, length, "noise"))[None]
    mix0 = mix.clone()
    bag = _bag4(32)
    random.seed(0)
    out = P.apply_model(bag, mix, shifts=2, overlap=0.25, device="cuda")
    state = random.getstate()
    assert out.device.type == "cpu" and out.shape == (1, 4, 2, length) and bool(torch.isfinite(out).all())
    assert torch.equal(mix, mix0)
    random.seed(0)
    slow = P.apply_model(_bag4(4, PerSegment), mix.cuda(), shifts=2, overlap=0.25, device="cuda")
    assert random.getstate() == state                       # same number of RNG draws in the same order
    assert torch.equal(out, slow.cpu())
    # one-hot weights: source k is exactly member k's own shift-averaged estimate
    random.seed(0)
    solo = P.apply_model(bag.models[0], mix, shifts=2, overlap=0.25, device="cuda")
    assert torch.equal(out[:, 0], solo[:, 0])


def test_config4_sixty_minute_track_single_gpu():
    """BASELINE configs[3]'s track on ONE GPU: L = 158 760 000 samples, 616 segments (apply.py:264-266 arithmetic).
    Finite, bit-identical to the per-segment route, two interior segments agree with the float64 oracle."""
    from oracle import htdemucs_oracle as O
    cfg = HTDemucsConfig()
    sd = synthetic_state_dict(cfg, 0)
    m = HTDemucs(cfg.sources, max_batch=32)
    m.load_state_dict(sd)
    length = 3600 * 44100
    stride = int(0.75 * SL)
    assert len(range(0, length, stride)) == 616
    gen = torch.Generator(device="cuda").manual_seed(4)
    mix = torch.randn(1, 2, length, device="cuda", generator=gen) * 0.1
    out = P.apply_model(m, mix, shifts=0, overlap=0.25)
    assert out.shape == (1, 4, 2, length) and out.device.type == "cuda"
    assert bool(torch.isfinite(out[0, :, :, ::13]).all())
    m2 = HTDemucs(cfg.sources, max_batch=4)
    m2.load_state_dict(sd)
    slow = P.apply_model(PerSegment(m2), mix, shifts=0, overlap=0.25)
    assert torch.equal(out, slow)
    del slow
    osd = O.to_torch_state(sd, torch.float64)
    for k in (0, 615, 333):
        off = k * stride
        n = min(SL, length - off)
        lo, hi = (0 if k == 0 else SL - stride), min(stride, n)      # samples covered by segment k only
        if hi - lo < 1000:
            continue
        seg = P.TensorChunk(mix, off, SL).padded(SL).cpu().double()       # the leaf's centred, neighbour-filled window
        with torch.no_grad():
            want = O.htdemucs_forward(osd, seg, 4)
        trim = (SL - n) // 2
        err = (out[..., off + lo:off + hi].cpu().double() - want[..., trim + lo:trim + hi]).abs().max().item()
        assert err <= TOL, (k, err)


def test_separator_separate_tensor_matches_reference_separator(golden):
    """`demucs_amd.api.Separator.separate_tensor` on the GPU against the reference's own
    `demucs.api.Separator.separate_tensor` (fixture made by tools/make_golden.py: loud DC-shifted input, shifts=1 with a
    seeded RNG).  Stems within the float32 bar scaled by the de-normalisation gain, `wav` restored in place, same callback
    dicts incl. `audio_length` and the caller's own keys."""
    from demucs_amd.api import Separator
    g = golden("separator_shift1")
    m = engine(int(g.meta("wseed")), 2)
    events = []
    sep = Separator(model=m, device="cuda", shifts=int(g.meta("kw_shifts")), overlap=float(g.meta("kw_overlap")),
                    split=bool(g.meta("kw_split")), callback=lambda d: events.append(dict(d)), callback_arg={"tag": "fixture"})
    wav = torch.from_numpy(3.0 * synth_mix(12, int(1.3 * SL), "tones") + 0.2)
    wav0 = wav.clone()
    random.seed(int(g.meta("rseed")))
    got_wav, stems = sep.separate_tensor(wav, sr=44100)
    assert got_wav is wav and float((wav - wav0).abs().max()) <= 1e-6
    assert list(stems) == m.sources
    out = torch.stack([stems[k] for k in m.sources])
    gain = float(wav0.mean(0).std())                                  # stems are apply_model's output x this
    e64 = g.check("f64", "out", out, atol=TOL * max(1.0, gain))
    g.check("f32", "out", out, atol=TOL * max(1.0, gain))
    g.check("f32", "wav", wav, atol=1e-6)
    keys = ["model_idx_in_bag", "shift_idx", "segment_offset", "models", "state", "audio_length", "tag"]
    got = np.array([[str(e[k]) for k in keys] for e in events])
    assert got.shape == g.z["events"].shape and (got == g.z["events"]).all()
    print(f"separator: max-abs vs reference f64 {e64:.2e} (de-normalisation gain {gain:.2f})")


def test_config3_bf16_mode_full_size_and_miniature(golden):
    """BASELINE configs[2] as written: htdemucs_ft-shaped bag of 4, shifts=2, overlap 0.25, bf16 compute mode.
    (1) the miniature with a reference golden (`apply_bag4_onehot_shifts2`, float64 reference run): SDR >= 28 dB;
    (2) the full 3-minute track: finite and >= 28 dB SDR per source against the float32 mode of the same engine."""
    from oracle import apply_oracle as A
    g = golden("apply_bag4_onehot_shifts2")
    kw = golden_kwargs(g)
    bag = _bag4(8, compute_dtype="bf16")
    random.seed(int(g.meta("rseed")))
    out = P.apply_model(bag, torch.from_numpy(CASES["apply_bag4_onehot_shifts2"]["mix"]())[None], device="cuda", **kw)
    stride = int(g.z["f64/out/stride"])
    want = g.z["f64/out/sample"].astype(np.float64)
    got = out.reshape(-1)[::stride].double().numpy()
    sdr = 10 * np.log10((np.sum(want * want) + 1e-7) / (np.sum((got - want) ** 2) + 1e-7))
    print(f"bag4 shifts2 bf16 miniature: max-abs {np.abs(got - want).max():.3e}  SDR {sdr:.1f} dB vs the float64 reference")
    assert sdr >= 28.0
    length = 180 * 44100
    mix = torch.from_numpy(synth_mix(1, length, "noise"))[None]
    random.seed(0)
    lo = P.apply_model(_bag4(32, compute_dtype="bf16"), mix, shifts=2, overlap=0.25, device="cuda")
    random.seed(0)
    hi = P.apply_model(_bag4(32), mix, shifts=2, overlap=0.25, device="cuda")
    assert lo.shape == (1, 4, 2, length) and bool(torch.isfinite(lo).all())
    sdrs = A.new_sdr(hi, lo)
    print(f"bag4 shifts2 bf16 full size: per-source SDR vs the float32 mode {[round(float(v), 1) for v in sdrs.flatten()]} dB, "
          f"max-abs {float((hi - lo).abs().max()):.3e}")
    assert float(sdrs.min()) >= 28.0


def test_config5_htdemucs_6s_fp16_track():
    """BASELINE configs[4], first half: the 6-source model (htdemucs_6s architecture) in the fp16 compute mode on a
    3-minute track.  Finite, (1, 6, 2, L), and >= 44 dB SDR per source against the float32 mode of the same engine
    (the fp16 floor measured against the reference's float64 output is in tests/test_gpu_model.py)."""
    from oracle import apply_oracle as A
    cfg6 = HTDemucsConfig(sources=["drums", "bass", "other", "vocals", "guitar", "piano"])
    sd = synthetic_state_dict(cfg6, 2)
    length = 180 * 44100
    mix = torch.from_numpy(synth_mix(1, length, "noise"))[None].cuda()
    outs = {}
    for dt in ("f16", "f32"):
        m = HTDemucs(cfg6.sources, max_batch=16, compute_dtype=dt)
        m.load_state_dict(sd)
        outs[dt] = P.apply_model(m, mix, shifts=0, overlap=0.25, device="cuda")
        m.release()
    assert outs["f16"].shape == (1, 6, 2, length) and bool(torch.isfinite(outs["f16"]).all())
    sdrs = A.new_sdr(outs["f32"].cpu(), outs["f16"].cpu())
    print(f"htdemucs_6s fp16 track: per-source SDR vs float32 mode {[round(float(v), 1) for v in sdrs.flatten()]} dB")
    assert float(sdrs.min()) >= 44.0
    # ... and against the FLOAT64 ORACLE (not the engine's own float32 mode) on the interior of one segment of the stitched track:
    # samples covered by segment k alone, so the stitched value is that segment's forward
    from oracle import htdemucs_oracle as O
    k, stride = 11, int(0.75 * SL)
    off, lo, hi = k * stride, SL - stride, stride
    with torch.no_grad():
        want = O.htdemucs_forward(O.to_torch_state(sd, torch.float64), mix[..., off:off + SL].cpu().double(), 6)[..., lo:hi]
    got = outs["f16"][..., off + lo:off + hi].cpu().double()
    sdr64 = 10 * torch.log10((want.pow(2).sum() + 1e-7) / ((got - want).pow(2).sum() + 1e-7))
    f32_err = float((outs["f32"][..., off + lo:off + hi].cpu().double() - want).abs().max())
    print(f"htdemucs_6s fp16 track, segment {k} interior vs the float64 oracle: SDR {float(sdr64):.1f} dB (float32 mode max-abs {f32_err:.2e})")
    assert float(sdr64) >= 44.0 and f32_err <= TOL
