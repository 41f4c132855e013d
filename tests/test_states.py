"""Checkpoint-package reader (demucs/states.py:50-107 -> demucs_amd/states.py).  No released checkpoint is available
offline.  PINNED by `tests/golden/pkg_htdemucs.th` / `pkg_hdemucs.th`: full-size packages assembled by the reference's own
`serialize_model` around reference models built with EVERY keyword of conf/config.yaml (tools/make_golden.py
`package_fixture`; periodic weights, members re-stored deflated), and by `ref_signatures.json`, the reference constructors'
keyword names and defaults.  The older cases write packages here in the same format with a stand-in module `demucs.htdemucs`
so that the pickled class reference carries the reference's qualified name."""
import sys
import types
import warnings
from fractions import Fraction

import pytest
import torch

from demucs_amd import states
from demucs_amd.htdemucs import HTDemucs
from demucs_amd.weights import HTDemucsConfig, synthetic_state_dict


@pytest.fixture()
def fake_reference_modules(monkeypatch):
    """`demucs.htdemucs.HTDemucs` / `demucs.hdemucs.HDemucs` as picklable names (what torch.save records for `klass`)."""
    pkg = types.ModuleType("demucs")
    made = {}
    for mod, cls in (("htdemucs", "HTDemucs"), ("hdemucs", "HDemucs")):
        m = types.ModuleType(f"demucs.{mod}")
        c = type(cls, (), {"__module__": f"demucs.{mod}"})
        setattr(m, cls, c)
        setattr(pkg, mod, m)
        monkeypatch.setitem(sys.modules, f"demucs.{mod}", m)
        made[cls] = c
    monkeypatch.setitem(sys.modules, "demucs", pkg)
    return made


def _package(klass, half):
    cfg = HTDemucsConfig()
    state = {k: torch.from_numpy(v).to(torch.half if half else torch.float32) for k, v in synthetic_state_dict(cfg, 7).items()}
    return {"klass": klass, "args": (list(cfg.sources),), "kwargs": {"segment": Fraction(39, 5), "dconv_mode": 3, "bottom_channels": 512, "t_dropout": 0.0, "t_cross_first": False,
                                                               "t_emb": "sin", "emb_smooth": True, "norm_starts": 4, "made_up_knob": 3},
            "state": state, "training_args": {"lr": 3e-4}}


@pytest.mark.parametrize("half", [False, True])
def test_file_roundtrip_without_executing_the_file(tmp_path, fake_reference_modules, half):
    path = tmp_path / "955717e8-8726e21a.th"
    torch.save(_package(fake_reference_modules["HTDemucs"], half), path)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        model = states.load_model(path)
    assert any("Dropping inexistant parameter made_up_knob" in str(x.message) for x in w)
    assert isinstance(model, HTDemucs) and model.sources == HTDemucsConfig().sources and model.segment == Fraction(39, 5)
    want = synthetic_state_dict(HTDemucsConfig(), 7)
    got = model.state_dict()
    assert set(got) == set(want)
    k = "crosstransformer.layers.0.linear1.weight"
    ref = torch.from_numpy(want[k])
    ref = ref.half().float() if half else ref
    assert torch.equal(torch.as_tensor(got[k]).float(), ref)


def test_strict_rejects_unknown_keywords_and_dict_input_works(fake_reference_modules):
    pkg = _package(fake_reference_modules["HTDemucs"], False)
    with pytest.raises(ValueError):
        states.load_model(dict(pkg), strict=True)
    model = states.load_model({**pkg, "klass": "demucs.htdemucs.HTDemucs", "kwargs": {"segment": Fraction(39, 5), "dconv_mode": 3,
                                                                                      "bottom_channels": 512}}, strict=True)
    assert isinstance(model, HTDemucs)


def test_other_architectures_and_quantised_states_are_refused(tmp_path, fake_reference_modules):
    path = tmp_path / "hd.th"
    torch.save({**_package(fake_reference_modules["HTDemucs"], False), "klass": "demucs.demucs.Demucs"}, path)
    with pytest.raises(ValueError, match="demucs.demucs.Demucs"):
        states.load_model(path)
    pkg = _package(fake_reference_modules["HTDemucs"], False)
    pkg["state"] = {"__quantized": True, "quantized": []}
    with pytest.raises(ValueError, match="quantised"):
        states.load_model(pkg)
    with pytest.raises(ValueError):
        states.load_model(12)


def test_arbitrary_pickled_objects_are_not_deserialised(tmp_path):
    class Evil:
        def __reduce__(self):
            return (print, ("executed from the checkpoint",))
    path = tmp_path / "evil.th"
    torch.save({"klass": Evil(), "args": (), "kwargs": {}, "state": {}}, path)
    with pytest.raises(Exception):
        states.read_package(path)


def test_local_repo_reads_signatures_checksums_and_bags(tmp_path, fake_reference_modules):
    import hashlib
    from demucs_amd.apply import BagOfModels
    sigs = ["f7e0c4bc", "d12395a8"]
    for i, sig in enumerate(sigs):
        tmp = tmp_path / f"{sig}.tmp"
        torch.save(_package(fake_reference_modules["HTDemucs"], True), tmp)
        digest = hashlib.sha256(tmp.read_bytes()).hexdigest()[:8]
        tmp.rename(tmp_path / (f"{sig}-{digest}.th" if i == 0 else f"{sig}.th"))
    (tmp_path / "two_ft.yaml").write_text("models: ['f7e0c4bc', 'd12395a8']\nweights: [[1., 0., 0., 0.], [0., 1., 1., 1.]]\n")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        repo = states.LocalRepo(tmp_path, max_batch=2)
        assert repo.has_model("two_ft") and repo.has_model("d12395a8") and not repo.has_model("nope")
        bag = repo.get_model("two_ft")
        assert isinstance(bag, BagOfModels) and len(bag.models) == 2 and bag.weights[1] == [0.0, 1.0, 1.0, 1.0]
        assert isinstance(repo.get_model("f7e0c4bc"), HTDemucs)
    with pytest.raises(states.ModelLoadingError):
        repo.get_model("nope")
    # demucs/api.py:322-347 list_models: {"single": {signature: path}, "bag": {name: path}}
    from demucs_amd.api import LoadModelError, list_models
    listed = list_models(tmp_path)
    assert sorted(listed["single"]) == sorted(sigs) and list(listed["bag"]) == ["two_ft"]
    assert listed["single"]["d12395a8"] == tmp_path / "d12395a8.th" and listed["bag"]["two_ft"] == tmp_path / "two_ft.yaml"
    assert list(list_models()["single"]) == ["demucs_unittest"] and list_models()["bag"] == {}
    with pytest.raises(LoadModelError):
        list_models(tmp_path / "missing")
    # a corrupted file no longer matches the checksum in its name
    bad = next(p for p in tmp_path.iterdir() if p.name.startswith("f7e0c4bc-"))
    bad.write_bytes(bad.read_bytes()[:-1] + b"\\0")
    with pytest.raises(states.ModelLoadingError, match="Invalid checksum"):
        states.LocalRepo(tmp_path).get_model("f7e0c4bc")


RELEASED = {"segment": Fraction(39, 5), "dconv_mode": 3, "bottom_channels": 512}


@pytest.mark.parametrize("flag", [{"t_cross_first": True}, {"t_emb": "scaled"}, {"t_norm_first": False}, {"t_gelu": False},
                                  {"t_sparse_self_attn": True}, {"use_train_segment": False}, {"norm_starts": 2}, {"cac": False},
                                  {"multi_freqs": [0.5]}, {"freq_emb": 0.0}])
def test_shape_preserving_architecture_flags_are_refused_not_dropped(fake_reference_modules, flag):
    """None of these changes a tensor shape, so load_state_dict could not notice them: the loader must."""
    pkg = _package(fake_reference_modules["HTDemucs"], False)
    pkg["kwargs"] = {**RELEASED, **flag}
    with pytest.raises(ValueError):
        states.load_model(pkg)


def test_known_reference_keywords_load_silently_and_omitted_ones_mean_reference_defaults(fake_reference_modules):
    from demucs_amd.weights import ENGINE_FIXED, INERT_KEYWORDS
    pkg = _package(fake_reference_modules["HTDemucs"], False)
    pkg["kwargs"] = {**RELEASED, **ENGINE_FIXED, **{k: 0 for k in INERT_KEYWORDS}, "norm_starts": 4, "multi_freqs": []}
    with warnings.catch_warnings():
        warnings.simplefilter("error")               # a real checkpoint (about 60 keywords) must not warn
        model = states.load_model(pkg)
    assert model.segment == Fraction(39, 5)
    # a package that omits dconv_mode / bottom_channels / segment asks for the reference defaults (1 / 0 / 10): another
    # architecture, refused instead of silently running the released one
    pkg["kwargs"] = {}
    with pytest.raises(ValueError):
        states.load_model(pkg)


def test_hdemucs_packages_load_into_the_hdemucs_engine_class(tmp_path, fake_reference_modules):
    """`hdemucs_mmi` (75fc33f5) is a `demucs.hdemucs.HDemucs` package: the reader builds the HDemucs engine class, keeps the
    reference's keyword rules (inert / fixed / unknown) and the bag YAML's `segment: 44` raises the member's segment."""
    from demucs_amd.apply import BagOfModels
    from demucs_amd.hdemucs import HDemucs
    from demucs_amd.hdemucs_weights import HDemucsConfig, synthetic_hdemucs_state_dict
    cfg = HDemucsConfig()
    state = {k: torch.from_numpy(v).half() for k, v in synthetic_hdemucs_state_dict(cfg, 1).items()}
    pkg = {"klass": fake_reference_modules["HDemucs"], "args": (list(cfg.sources),),
           "kwargs": {"channels": 48, "depth": 6, "dconv_init": 1e-3, "hybrid": True, "rescale": 0.1, "made_up": 1}, "state": state}
    path = tmp_path / "75fc33f5.th"
    torch.save(pkg, path)
    (tmp_path / "hdemucs_mmi.yaml").write_text("models: ['75fc33f5']\nsegment: 44\n")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        bag = states.LocalRepo(tmp_path).get_model("hdemucs_mmi")
    assert any("made_up" in str(x.message) for x in w)
    assert isinstance(bag, BagOfModels) and isinstance(bag.models[0], HDemucs) and bag.models[0].segment == 44
    assert not hasattr(bag.models[0], "valid_length")
    with pytest.raises(ValueError):
        states.load_model({**pkg, "kwargs": {"hybrid": False}})
    with pytest.raises(ValueError):
        states.load_model({**pkg, "kwargs": {"depth": 5}})


# ---- pinned by files the reference wrote (tools/make_golden.py package_fixture / signature_fixture) --------------------
import json       # noqa: E402
import os         # noqa: E402

import numpy as np    # noqa: E402

from conftest import GOLDEN    # noqa: E402


@pytest.mark.parametrize("name", ["pkg_htdemucs", "pkg_hdemucs"])
def test_reference_written_package_loads_without_warnings(name):
    """`states.load_model` on the package the reference's serialize_model + torch.save produced: every one of the 62 / 35
    keywords is understood (no "Dropping inexistant parameter" warning, nothing refused), the class maps to the engine class,
    and the state equals the float16-rounded periodic fill the fixture was made from."""
    from demucs_amd.hdemucs import HDemucs
    from demucs_amd.hdemucs_weights import HDemucsConfig, synthetic_hdemucs_state_dict
    path = os.path.join(GOLDEN, name + ".th")
    meta = np.load(os.path.join(GOLDEN, name + ".npz"))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        model = states.load_model(path)
    assert not w, [str(x.message) for x in w]
    pkg = states.read_package(path)
    assert sorted(pkg["kwargs"]) == list(meta["meta/kwargs"]) and pkg["args"] == ()
    assert all(v.dtype == torch.float16 for v in pkg["state"].values()) and len(pkg["state"]) == int(meta["meta/n_tensors"])
    wseed, period = int(meta["meta/wseed"]), int(meta["meta/period"])
    if name == "pkg_htdemucs":
        assert isinstance(model, HTDemucs) and model.segment == Fraction(39, 5)
        want = synthetic_state_dict(HTDemucsConfig(), wseed, period=period)
    else:
        assert isinstance(model, HDemucs) and model.segment == 40
        want = synthetic_hdemucs_state_dict(HDemucsConfig(), wseed, period=period)
    got = model.state_dict()
    assert list(got) == list(want)
    for k in list(want)[::7]:
        assert torch.equal(torch.as_tensor(got[k]).float(), torch.from_numpy(want[k]).half().float()), k


def test_keyword_tables_cover_the_reference_signatures():
    """Every keyword of the reference constructors (inspect.signature, dumped by tools/make_golden.py) is classified by the
    engine -- config field, inert in eval, or a fixed value -- and every default the engine fills in for an omitted keyword
    equals the reference's default; the released configuration (conf/config.yaml) is accepted value by value."""
    from demucs_amd import weights as W
    from demucs_amd.hdemucs import HDemucs
    from demucs_amd.hdemucs_weights import HDemucsConfig
    with open(os.path.join(GOLDEN, "ref_signatures.json")) as f:
        sig = json.load(f)

    def value(v):
        return Fraction(*v["Fraction"]) if isinstance(v, dict) and "Fraction" in v else v
    # HTDemucs
    fields = set(vars(HTDemucsConfig()))
    ref = {k: value(v) for k, v in sig["HTDemucs"]}
    for key, default in ref.items():
        if key == "sources":
            continue
        if key in fields:
            assert key in W.REFERENCE_DEFAULTS, f"HTDemucs config field {key} has no reference default recorded"
            assert W.REFERENCE_DEFAULTS[key] == default, (key, W.REFERENCE_DEFAULTS[key], default)
        else:
            assert key in W.INERT_KEYWORDS or key in W.ENGINE_FIXED or key == "norm_starts", f"HTDemucs keyword {key} is unclassified"
            if key in W.ENGINE_FIXED and key not in ("use_train_segment",):
                # the reference's default of a forward-changing keyword must be the value the engine implements
                assert W.check_reference_keyword(key, default), key
    assert set(W.REFERENCE_DEFAULTS) <= set(ref), set(W.REFERENCE_DEFAULTS) - set(ref)
    assert (set(W.INERT_KEYWORDS) | set(W.ENGINE_FIXED)) <= set(ref), (set(W.INERT_KEYWORDS) | set(W.ENGINE_FIXED)) - set(ref)
    for key, v in sig["config.yaml"]["htdemucs"].items():          # the released configuration, value by value
        if key not in fields:
            assert W.check_reference_keyword(key, v), key
    # HDemucs
    hfields = set(vars(HDemucsConfig()))
    href = {k: value(v) for k, v in sig["HDemucs"]}
    cfg = HDemucsConfig()
    for key, default in href.items():
        if key == "sources":
            continue
        assert key in hfields or key in HDemucs._INERT or key in HDemucs._FIXED, f"HDemucs keyword {key} is unclassified"
        if key in hfields:
            assert getattr(cfg, key) == default, (key, getattr(cfg, key), default)
        elif key in HDemucs._FIXED:
            assert HDemucs._FIXED[key] == default or (key == "multi_freqs" and not default), (key, default)
    assert (hfields - {"sources"}) | set(HDemucs._INERT) | set(HDemucs._FIXED) <= set(href)
    HDemucs(**{**sig["config.yaml"]["hdemucs"], "sources": ["a", "b"], "audio_channels": 2, "samplerate": 44100, "segment": 40})


def test_pretrained_get_model_offline():
    """demucs/pretrained.py:59-85 without the remote zoo: "demucs_unittest" (HDemucs(channels=4, sources=SOURCES), pretrained.py:27-29)
    resolves to an engine model object with the reference's state-dict schema at that width; any other name needs a local folder."""
    from demucs_amd.hdemucs_weights import HDemucsConfig, hdemucs_schema
    from demucs_amd.pretrained import SOURCES, get_model
    from demucs_amd.states import ModelLoadingError
    m = get_model("demucs_unittest")
    assert m.sources == SOURCES and m.cfg.channels == 4 and not hasattr(m, "valid_length")
    assert list(m.state_dict()) == list(hdemucs_schema(HDemucsConfig(channels=4)))
    assert m.state_dict()["encoder.5.conv.weight"].shape == (128, 64, 4)          # channels << level
    with pytest.raises(ModelLoadingError):
        get_model("htdemucs")
    with pytest.raises(ModelLoadingError):
        get_model("htdemucs", repo="/nonexistent/folder")
