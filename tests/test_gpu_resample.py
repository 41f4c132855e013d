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


def test_clip_prevention_and_two_stems_match_the_reference(golden):
    """SURVEY 8 f4 on the GPU against the REFERENCE: tests/golden/clip_two_stems.npz holds `demucs.audio.prevent_clip` in its
    three modes and the tensors the reference's `separate.main --two-stems vocals --other-method add|minus|none` hands to
    `save_audio` (tools/make_golden.py clip_fixture runs that code with stand-in Separator / save_audio).  The device kernels
    (`mi_prevent_clip`, `mi_two_stems`) must reproduce them: bit-exact for rescale / clamp / the sums, 2e-7 for tanh (libm)."""
    import numpy as np
    from conftest import GOLDEN
    import os
    z = np.load(os.path.join(GOLDEN, "clip_two_stems.npz"))
    names = [str(n) for n in z["meta/sources"]]
    stems = {k: torch.from_numpy(z[f"stem/{k}"]).cuda() for k in names}
    origin = torch.from_numpy(z["origin"]).cuda()
    for k in names + ["quiet"]:
        x = torch.from_numpy(z[f"stem/{k}"]).cuda()
        for mode in ("rescale", "clamp", "tanh"):
            key = f"clip/{mode}/{k}"
            if key not in z.files:
                continue
            got = audio.prevent_clip(x, mode)
            want = torch.from_numpy(z[key])
            assert got.is_cuda and got.shape == x.shape
            if mode == "tanh":
                assert float((got.cpu() - want).abs().max()) <= 2e-7
            else:
                assert torch.equal(got.cpu(), want), (mode, k, float((got.cpu() - want).abs().max()))
        assert audio.prevent_clip(x, None) is x and audio.prevent_clip(x, "none") is x
    assert float(torch.from_numpy(z["stem/vocals"]).abs().max()) > 1.0          # the fixture really clips
    for method, extra in (("add", "no_vocals"), ("minus", "minus_vocals"), ("none", None)):
        got = audio.two_stems(origin, stems, "vocals", method)
        want_keys = sorted(k.split("/")[-1] for k in z.files if k.startswith(f"two_stems/{method}/"))
        assert sorted(got) == want_keys, (sorted(got), want_keys)
        for k, v in got.items():
            assert v.is_cuda and torch.equal(v.cpu(), torch.from_numpy(z[f"two_stems/{method}/{k}"])), (method, k)
    with pytest.raises(ValueError):
        audio.prevent_clip(origin, "loud")
    with pytest.raises(KeyError):
        audio.two_stems(origin, stems, "kazoo")
    # host stems (what Separator.separate_tensor(host wav) returns) are staged through the GPU; other float dtypes are cast
    host = audio.prevent_clip(origin.cpu(), "rescale")
    assert host.device.type == "cpu" and torch.equal(host, audio.prevent_clip(origin, "rescale").cpu())
    half = audio.prevent_clip(origin.double(), "clamp")
    assert half.dtype == torch.float64 and half.is_cuda and float(half.abs().max()) <= 0.99 + 1e-7
    hs = audio.two_stems(origin.cpu(), {k: v.cpu() for k, v in stems.items()}, "vocals", "add")
    assert hs["no_vocals"].device.type == "cpu" and torch.equal(hs["no_vocals"], torch.from_numpy(z["two_stems/add/no_vocals"]))
    bad = origin.clone()
    bad[0, 5] = float("nan")                                  # torch's abs().max() propagates NaN: so does the device reduction
    assert bool(torch.isnan(audio.prevent_clip(bad, "rescale")).all())
