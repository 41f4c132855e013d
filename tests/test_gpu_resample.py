"""mi_resample_frac (julius.resample_frac for demucs.audio.convert_audio) on the GPU against the CPU restatement,
and Separator.separate_tensor with a foreign sample rate (demucs/api.py:265-266).  Parity with julius itself is unpinned
(absent dependency, see demucs_amd/audio.py); tolerance 4e-6 relative = float32 summation-order noise of a ~200-tap filter."""
import math

import pytest
import torch

from demucs_amd import audio
from oracle import resample_oracle as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("old,new,length", [(48000, 44100, 100003), (22050, 44100, 5000), (44100, 16000, 44100), (8000, 44100, 777),
                                             (48000, 44100, 37)])
def test_kernel_matches_cpu_restatement(old, new, length):
    g = torch.Generator().manual_seed(length)
    x = torch.randn(2, 3, length, generator=g)
    got = audio.resample_frac(x, old, new)
    want = R.resample_frac(x, old, new, dtype=torch.float32)      # julius builds its filter bank in float32: same table
    assert got.shape == want.shape and got.device == x.device
    scale = max(1.0, float(want.abs().max()))
    assert float((got - want).abs().max()) < 4e-6 * scale          # float32 summation order of a ~200-tap filter
    exact = R.resample_frac(x, old, new, dtype=torch.float64)     # float64 table: the float32 table itself is 1e-5 off
    assert float((got.double() - exact).abs().max()) < 1e-4 * scale


def test_separator_resamples_then_separates():
    """separate_tensor(wav, sr=48000): convert_audio first, then the same normalise / apply / restore contract."""
    from demucs_amd.api import Separator
    from demucs_amd.htdemucs import HTDemucs
    from demucs_amd.weights import HTDemucsConfig, synthetic_state_dict
    cfg = HTDemucsConfig()
    m = HTDemucs(cfg.sources, max_batch=2)
    m.load_state_dict(synthetic_state_dict(cfg, 4))
    sep = Separator(model=m, device="cuda", shifts=0, overlap=0.25)
    g = torch.Generator().manual_seed(3)
    wav48 = torch.randn(1, 48000 * 3, generator=g) * 0.1               # mono, 3 s at 48 kHz
    wav, stems = sep.separate_tensor(wav48.clone(), sr=48000)
    n = math.floor(147 * wav48.shape[-1] / 160)
    assert wav.shape == (2, n) and set(stems) == set(cfg.sources)
    assert all(v.shape == (2, n) and bool(torch.isfinite(v).all()) for v in stems.values())
    want_wav = R.resample_frac(wav48.expand(2, -1), 48000, 44100, dtype=torch.float64)
    assert float((wav.double() - want_wav).abs().max()) < 1e-4          # restored after the in-place normalisation
    # same result as resampling by hand and calling the model-rate path
    wav2, stems2 = sep.separate_tensor(want_wav.float().clone())
    for k in stems:
        assert float((stems[k] - stems2[k]).abs().max()) < 1e-4


def test_clip_prevention_and_two_stems_on_the_stems_device():
    """SURVEY 8 f4, tensor half, on the GPU: the stems of an engine separation stay in HBM (device mix, split=True) and
    `prevent_clip` / `two_stems` (demucs/audio.py:218-234, demucs/separate.py:189-218) run there; results equal the same
    functions applied to host copies (bit-identical for rescale / clamp, last-bit for tanh), and the two-stems sum reconstructs the
    sum of the other stems."""
    from demucs_amd import apply as P
    from demucs_amd.htdemucs import HTDemucs
    from demucs_amd.synth import synth_mix
    from demucs_amd.weights import HTDemucsConfig, synthetic_state_dict
    cfg = HTDemucsConfig()
    m = HTDemucs(cfg.sources, max_batch=2)
    m.load_state_dict(synthetic_state_dict(cfg, 4))
    mix = (torch.from_numpy(synth_mix(3, 400000, "tones"))[None] * 4.0).cuda()          # loud: the stems clip
    out = P.apply_model(m, mix, shifts=0, overlap=0.25, device="cuda")
    stems = dict(zip(cfg.sources, out[0]))
    assert all(v.is_cuda for v in stems.values()) and float(out.abs().max()) > 1.0
    for mode in ("rescale", "clamp", "tanh", None):
        for k, v in stems.items():
            got = audio.prevent_clip(v, mode)
            want = audio.prevent_clip(v.cpu(), mode)
            # rescale / clamp are exact float32 ops; the device tanh and the host tanh may differ in the last bit
            assert got.is_cuda and (torch.allclose(got.cpu(), want, rtol=0, atol=2e-7) if mode == "tanh" else torch.equal(got.cpu(), want))
        if mode in ("rescale", "clamp", "tanh"):
            assert float(audio.prevent_clip(out, mode).abs().max()) <= 1.0
    two = audio.two_stems(mix[0], stems, "vocals")
    assert list(two) == ["vocals", "no_vocals"] and two["no_vocals"].is_cuda
    want = torch.zeros_like(stems["drums"])
    for k in ("drums", "bass", "other"):
        want += stems[k]
    assert torch.equal(two["no_vocals"], want)
    minus = audio.two_stems(mix[0], stems, "vocals", "minus")
    assert torch.equal(minus["minus_vocals"], mix[0] - stems["vocals"])
