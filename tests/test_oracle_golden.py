"""The CPU oracle against the reference's outputs (tests/golden, made by tools/make_golden.py).

float64 oracle vs float64 reference must agree to ~1e-10 (same algorithm, same primitives);
float32 oracle vs float32 reference to float32 re-association noise.
"""
import random

import numpy as np
import pytest
import torch

from demucs_amd.synth import synth_mix
from demucs_amd.weights import HTDemucsConfig, synthetic_state_dict
from oracle import apply_oracle as A
from oracle import htdemucs_oracle as O

SL = 343980
CFG6 = HTDemucsConfig(sources=["drums", "bass", "other", "vocals", "guitar", "piano"])
SEG_CASES = {
    "seg_noise_w0": (HTDemucsConfig(), 0, lambda: synth_mix(123, SL, "noise")),
    "seg_tones_w1": (HTDemucsConfig(), 1, lambda: synth_mix(7, SL, "tones")),
    "seg_short_w0": (HTDemucsConfig(), 0, lambda: synth_mix(5, 100001, "tones")),
    "seg6_noise_w2": (CFG6, 2, lambda: synth_mix(11, SL, "noise")),
}


@pytest.mark.parametrize("name", list(SEG_CASES))
@pytest.mark.parametrize("tag,dtype,atol", [("f64", torch.float64, 2e-9), ("f32", torch.float32, 6e-5)])
def test_segment_forward_matches_reference(golden, name, tag, dtype, atol):
    if tag == "f64" and name not in ("seg_noise_w0", "seg_short_w0"):
        pytest.skip("float64 run kept to two cases to bound CPU time")
    cfg, wseed, mk = SEG_CASES[name]
    g = golden(name)
    sd = O.to_torch_state(synthetic_state_dict(cfg, wseed), dtype)
    mix = torch.from_numpy(mk()).to(dtype)[None]
    taps = {}
    with torch.no_grad():
        out = O.htdemucs_forward(sd, mix, len(cfg.sources), taps=taps)
    taps["out"] = out
    for tap in g.taps(tag):
        g.check(tag, tap, taps[tap], atol=atol, rtol=atol)


APPLY_CASES = {
    "apply_one_segment": dict(wseeds=[0], mix=lambda: synth_mix(1, SL, "noise")),
    "apply_2p3_segments": dict(wseeds=[0], mix=lambda: synth_mix(2, int(2.3 * SL), "tones")),
    "apply_sl_plus_1": dict(wseeds=[1], mix=lambda: synth_mix(3, SL + 1, "noise")),
    "apply_shifts2": dict(wseeds=[0], mix=lambda: synth_mix(4, 300000, "tones")),
    "apply_bag2_shift1": dict(wseeds=[10, 11], mix=lambda: synth_mix(5, 400000, "noise")),
    "apply_bag4_onehot_shifts2": dict(wseeds=[10, 11, 12, 13], mix=lambda: synth_mix(9, 280000, "tones")),
    "apply_nosplit_short": dict(wseeds=[0], mix=lambda: synth_mix(6, 200000, "tones")),
    "apply_overlap10_tp2": dict(wseeds=[1], mix=lambda: synth_mix(8, int(1.5 * SL), "noise")),
}


def golden_kwargs(g):
    kw = {}
    for k in g.z.files:
        if k.startswith("meta/kw_"):
            v = g.z[k].item()
            kw[k[len("meta/kw_"):]] = v
    return kw


@pytest.mark.parametrize("name", list(APPLY_CASES))
def test_apply_model_matches_reference(golden, name):
    case = APPLY_CASES[name]
    g = golden(name)
    cfg = HTDemucsConfig()
    kw = golden_kwargs(g)
    models = [O.OracleModel(synthetic_state_dict(cfg, s), cfg.sources) for s in case["wseeds"]]
    bw = g.meta("bag_weights")
    model = models[0] if bw is None else A.Bag(models, bw.tolist())
    if g.meta("rseed") is not None:
        random.seed(int(g.meta("rseed")))
    mix = torch.from_numpy(case["mix"]())[None]
    mix0 = mix.clone()
    events = []
    out = A.apply_model(model, mix, callback=lambda d: events.append(dict(d)), **kw)
    assert torch.equal(mix, mix0)
    g.check("f32", "out", out, atol=6e-5, rtol=6e-5)
    # scored against the float64 reference too (truth): float32 noise only
    g.check("f64", "out", out, atol=1e-4, rtol=1e-4)
    keys = ["model_idx_in_bag", "shift_idx", "segment_offset", "models", "state"]
    got = np.array([[str(e[k]) for k in keys] for e in events])
    assert got.shape == g.z["events"].shape and (got == g.z["events"]).all()


def test_window_padded_uses_real_neighbours():
    """TensorChunk.padded (apply.py:108-124): centred, neighbour-filled, zero outside the track."""
    t = torch.arange(20.0).view(1, 1, 20)
    w = A.Window(t, 12, 6)                      # samples 12..17
    p = w.padded(10)                            # delta 4 -> start 10
    assert p.flatten().tolist() == [10, 11, 12, 13, 14, 15, 16, 17, 18, 19]
    w = A.Window(t, 15, 100)                    # clipped to 5 samples 15..19
    assert w.length == 5
    p = w.padded(12)                            # delta 7 -> start 12, right side runs off the end
    assert p.flatten().tolist() == [12, 13, 14, 15, 16, 17, 18, 19, 0, 0, 0, 0]
    inner = A.Window(A.Window(t, 4, 10), 2, 3)  # nested windows compose offsets
    assert inner.offset == 6 and inner.padded(5).flatten().tolist() == [5, 6, 7, 8, 9]


def test_center_trim_and_errors():
    x = torch.arange(10.0)
    assert A.center_trim(x, 7).tolist() == [1, 2, 3, 4, 5, 6, 7]     # odd remainder dropped on the right
    with pytest.raises(ValueError):
        A.center_trim(x, 11)
    cfg = HTDemucsConfig()
    m = O.OracleModel(synthetic_state_dict(cfg, 0), cfg.sources)
    with pytest.raises(ValueError):                                    # htdemucs.py:521-524
        A.apply_model(m, torch.zeros(1, 2, SL + 5), shifts=0, split=False)
    with pytest.raises(AssertionError):                                # apply.py:235
        A.apply_model(m, torch.zeros(1, 2, 1000), shifts=0, transition_power=0.5)


def separator_input():
    return 3.0 * synth_mix(12, int(1.3 * SL), "tones") + 0.2


def test_separate_tensor_matches_reference_separator(golden):
    """tests/golden/separator_shift1.npz = the reference's own `Separator.separate_tensor` (api.py:241-291) run by
    tools/make_golden.py; checks the normalise / separate / restore contract and the callback dicts (audio_length)."""
    g = golden("separator_shift1")
    cfg = HTDemucsConfig()
    model = O.OracleModel(synthetic_state_dict(cfg, int(g.meta("wseed"))), cfg.sources)
    wav = torch.from_numpy(separator_input())
    wav0 = wav.clone()
    events = []
    random.seed(int(g.meta("rseed")))
    got_wav, stems = A.separate_tensor(model, wav, callback=lambda d: events.append(dict(d)), callback_arg={"tag": "fixture"},
                                       shifts=int(g.meta("kw_shifts")), overlap=float(g.meta("kw_overlap")), split=bool(g.meta("kw_split")))
    assert got_wav is wav and float((wav - wav0).abs().max()) <= 4 * float(g.z["f32/restore_err"]) + 1e-6
    assert list(stems) == cfg.sources
    out = torch.stack([stems[k] for k in cfg.sources])
    g.check("f32", "out", out, atol=2e-4, rtol=6e-5)              # outputs are ~3x the unit-scale cases
    g.check("f64", "out", out, atol=3e-4, rtol=1e-4)
    keys = ["model_idx_in_bag", "shift_idx", "segment_offset", "models", "state", "audio_length", "tag"]
    got = np.array([[str(e[k]) for k in keys] for e in events])
    assert got.shape == g.z["events"].shape and (got == g.z["events"]).all()


# ---- Hybrid Demucs v3 (`hdemucs_mmi` architecture) ---------------------------------------------------
from demucs_amd.hdemucs_weights import HDemucsConfig, hdemucs_layer_plan, synthetic_hdemucs_state_dict  # noqa: E402
from oracle import hdemucs_oracle as HO  # noqa: E402

HSEG = {"hseg_tones_10s_w0": (0, lambda: synth_mix(21, 441000, "tones")),
        "hseg_noise_odd_w1": (1, lambda: synth_mix(22, 233731, "noise")),
        "hseg_tiny_w0": (0, lambda: synth_mix(24, 1500, "tones")),
        "hseg_10smp_w1": (1, lambda: synth_mix(25, 10, "noise"))}        # 10 samples: the reference forwards any length >= 1


@pytest.mark.parametrize("name", list(HSEG))
@pytest.mark.parametrize("tag,dtype,atol", [("f64", torch.float64, 2e-9), ("f32", torch.float32, 8e-5)])
def test_hdemucs_forward_matches_reference(golden, name, tag, dtype, atol):
    """Every encoder / decoder tap (incl. the BLSTM + LocalState layers 4, 5, the merge layer, GroupNorm(4), the decoders
    that start from zeros) and the output of the reference's HDemucs, at two lengths (one odd)."""
    if tag == "f64" and name == "hseg_tones_10s_w0":
        pytest.skip("float64 run kept to the short case to bound CPU time")
    wseed, mk = HSEG[name]
    g = golden(name)
    cfg = HDemucsConfig()
    sd = {k: torch.from_numpy(v.copy()).to(dtype) for k, v in synthetic_hdemucs_state_dict(cfg, wseed).items()}
    taps = {}
    with torch.no_grad():
        out = HO.hdemucs_forward(sd, torch.from_numpy(mk()).to(dtype)[None], hdemucs_layer_plan(cfg), 4, taps=taps)
    taps["out"] = out
    checked = 0
    for tap in g.taps(tag):
        g.check(tag, tap, taps[tap], atol=atol, rtol=atol)
        checked += 1
    assert checked >= 23            # 6 enc + 6 dec + 5 tenc + 5 tdec + out


def test_hdemucs_oracle_at_the_unittest_width(golden):
    """The reference's own offline model, `demucs_unittest` = HDemucs(channels=4) (pretrained.py:27-29), with this repo's
    deterministic weights: the oracle follows the same layer plan at any width (fixture `hseg_unittest_w3`, 216 frames)."""
    g = golden("hseg_unittest_w3")
    cfg = HDemucsConfig(channels=4)
    for tag, dtype, atol in (("f64", torch.float64, 2e-9), ("f32", torch.float32, 8e-5)):
        sd = {k: torch.from_numpy(v.copy()).to(dtype) for k, v in synthetic_hdemucs_state_dict(cfg, 3).items()}
        taps = {}
        with torch.no_grad():
            out = HO.hdemucs_forward(sd, torch.from_numpy(synth_mix(27, 220623, "tones")).to(dtype)[None], hdemucs_layer_plan(cfg), 4, taps=taps)
        taps["out"] = out
        assert sum(1 for tap in g.taps(tag) if g.check(tag, tap, taps[tap], atol=atol, rtol=atol) is not None) >= 23


@pytest.mark.parametrize("name,mk", [("happly_10s_seg4", lambda: synth_mix(23, 449833, "tones")),
                                     ("happly_tail10", lambda: synth_mix(26, 396910, "noise"))])      # tail chunk of 10 samples
def test_hdemucs_apply_model_matches_reference(golden, name, mk):
    """apply_model around HDemucs: no valid_length, so every chunk runs at its own length (the last one shorter)."""
    g = golden(name)
    cfg = HDemucsConfig()
    model = HO.OracleHDemucs(synthetic_hdemucs_state_dict(cfg, int(g.meta("wseed"))), cfg)
    kw = golden_kwargs(g)
    mix = torch.from_numpy(mk())[None]
    events = []
    out = A.apply_model(model, mix, callback=lambda d: events.append(dict(d)), **kw)
    g.check("f32", "out", out, atol=8e-5, rtol=8e-5)
    g.check("f64", "out", out, atol=2e-4, rtol=1e-4)
    keys = ["model_idx_in_bag", "shift_idx", "segment_offset", "models", "state"]
    got = np.array([[str(e[k]) for k in keys] for e in events])
    assert got.shape == g.z["events"].shape and (got == g.z["events"]).all()


# ---- checkpoint packages written by the reference ------------------------------------------------------
def _package_state(path):
    """The float16 state of a golden package as float32 numpy arrays (zip read with weights_only=True: nothing executed)."""
    from demucs_amd import states
    pkg = states.read_package(path)
    return pkg, {k: v.float().numpy() for k, v in pkg["state"].items()}


def test_oracle_on_reference_package_htdemucs(golden):
    """The float64 oracle, fed the weights of the package the reference's `serialize_model` wrote, reproduces the forward of
    the reference model re-loaded from that package."""
    import os
    from conftest import GOLDEN
    g = golden("pkg_htdemucs")
    _, sd = _package_state(os.path.join(GOLDEN, "pkg_htdemucs.th"))
    mix = torch.from_numpy(synth_mix(31, int(g.meta("length")), "tones"))[None]
    with torch.no_grad():
        out = O.htdemucs_forward(O.to_torch_state(sd, torch.float64), mix.double(), 4)
    g.check("f64", "out", out, atol=2e-9, rtol=2e-9)
    g.check("f32", "out", out, atol=8e-5, rtol=8e-5)


def test_oracle_on_reference_package_hdemucs(golden):
    import os
    from conftest import GOLDEN
    g = golden("pkg_hdemucs")
    _, sd = _package_state(os.path.join(GOLDEN, "pkg_hdemucs.th"))
    mix = torch.from_numpy(synth_mix(32, int(g.meta("length")), "noise"))[None]
    osd = {k: torch.from_numpy(v).double() for k, v in sd.items()}
    with torch.no_grad():
        out = HO.hdemucs_forward(osd, mix.double(), hdemucs_layer_plan(HDemucsConfig()), 4)
    g.check("f64", "out", out, atol=2e-9, rtol=2e-9)
