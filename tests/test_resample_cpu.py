"""demucs.audio.convert_audio restatement (demucs/audio.py:137-172): channel conversion and the julius sinc resampler.
julius is absent from the reference tree and from this image: the resampler's parity is UNPINNED; these tests check the
oracle restatement (oracle/resample_oracle.py) against analytic properties and the host logic around it."""
import math

import pytest
import torch

from demucs_amd import audio
from oracle import resample_oracle as R


def test_channel_conversion_cases():
    x = torch.arange(12.0).reshape(3, 4)
    assert torch.equal(audio.convert_audio_channels(x, 3), x)
    assert torch.equal(audio.convert_audio_channels(x, 1), x.mean(0, keepdim=True))
    assert torch.equal(audio.convert_audio_channels(x[:1], 2), x[:1].expand(2, 4))
    assert torch.equal(audio.convert_audio_channels(x, 2), x[:2])
    with pytest.raises(ValueError):
        audio.convert_audio_channels(x[:2], 3)


@pytest.mark.parametrize("old,new", [(48000, 44100), (22050, 44100), (44100, 16000), (32000, 44100)])
def test_sinc_bank_is_normalised_and_oracle_keeps_a_sine(old, new):
    g = math.gcd(old, new)
    width, bank = audio.sinc_bank(old // g, new // g)
    assert bank.shape == (new // g, 2 * width + old // g)
    assert torch.allclose(bank.sum(1), torch.ones(new // g), atol=1e-6)
    f = 1000.0                                            # well below both Nyquist limits
    t = torch.arange(old) / old
    x = torch.sin(2 * math.pi * f * t)[None]
    y = R.resample_frac(x, old, new, dtype=torch.float64)[0]
    assert y.shape[-1] == math.floor(new * old / old)
    want = torch.sin(2 * math.pi * f * torch.arange(new, dtype=torch.float64) / new)
    core = slice(new // 10, -new // 10)                  # away from the replicate-padded edges
    assert float((y[core] - want[core]).abs().max()) < 2e-3


def test_oracle_identity_and_lengths():
    x = torch.randn(2, 1000)
    assert R.resample_frac(x, 44100, 44100) is x
    assert R.resample_frac(x, 48000, 44100).shape == (2, math.floor(147 * 1000 / 160))
    assert R.resample_frac(x, 8000, 44100).shape == (2, math.floor(441 * 1000 / 80))


def test_resample_needs_the_gpu_engine():
    from demucs_amd._lib import EngineError
    with pytest.raises(EngineError):
        audio.resample_frac(torch.randn(2, 100), 48000, 44100, device="cpu")
    x = torch.randn(2, 100)
    assert audio.resample_frac(x, 44100, 44100, device="cpu") is x       # equal rates: no engine involved


def test_prevent_clip_and_two_stems_host_logic_without_a_gpu():
    """demucs/audio.py:218-234 and the --two-stems branch of demucs/separate.py:189-218: the arithmetic runs in HIP kernels
    (tests/test_gpu_resample.py checks it against the reference's fixture); here only what needs no device -- pass-through
    modes, argument errors, and the loud refusal of host tensors (no CPU implementation)."""
    from demucs_amd._lib import EngineError
    w = torch.tensor([[0.5, -2.0, 1.5]])
    assert audio.prevent_clip(w, None) is w and audio.prevent_clip(w, "none") is w
    with pytest.raises(ValueError):
        audio.prevent_clip(w, "loud")
    with pytest.raises(EngineError):
        audio.prevent_clip(w, "rescale")
    stems = {k: torch.full((2, 4), float(i + 1)) for i, k in enumerate(["drums", "bass", "other", "vocals"])}
    origin = sum(stems.values())
    assert list(audio.two_stems(origin, stems, "bass", "none")) == ["bass"]
    assert set(stems) == {"drums", "bass", "other", "vocals"}                            # the caller's dict is not consumed
    with pytest.raises(KeyError):
        audio.two_stems(origin, stems, "piano")
    with pytest.raises(ValueError):
        audio.two_stems(origin, stems, "bass", "multiply")
    with pytest.raises(EngineError):
        audio.two_stems(origin, stems, "vocals", "add")
