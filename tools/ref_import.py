"""Import the upstream reference (read-only checkout at /root/reference) for GOLDEN GENERATION ONLY.

This module is tooling for the build container: it is never imported by the product
(`demucs_amd/`), by `bench.py`, by `__graft_entry__.py` or by any test.  The GPU box has no
/root/reference.  It registers empty stand-in modules for the reference's *non-hot-path*
imports that are not installed here (audio codecs, experiment manager, Wiener filter): none
of them is called by `apply_model` / `HTDemucs.forward` with cac=True, and
each stand-in raises if it ever is.
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


def _na(*a, **k):
    raise RuntimeError("stand-in for a non-hot-path dependency was called")


def _stub(name, **attrs):
    if name in sys.modules:
        return
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m


def import_reference():
    _stub("julius", resample_frac=_na)
    _stub("openunmix")
    _stub("openunmix.filtering", wiener=_na)
    _stub("dora")
    _stub("dora.log", fatal=_na, bold=lambda s: s)
    _stub("omegaconf", OmegaConf=types.SimpleNamespace(to_container=_na))
    _stub("lameenc")
    _stub("torchaudio")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import demucs.apply as ref_apply            # noqa
    import demucs.htdemucs as ref_htdemucs      # noqa
    return ref_apply, ref_htdemucs


def build_reference_htdemucs(cfg, state_dict_np, dtype=None):
    """Reference HTDemucs with the released htdemucs hyper-parameters, our weights loaded."""
    import torch
    _, ref_htdemucs = import_reference()
    model = ref_htdemucs.HTDemucs(
        sources=list(cfg.sources), dconv_mode=3, depth=4, t_layers=5, bottom_channels=512,
        t_dropout=0.02, segment=cfg.segment)
    sd = {k: torch.from_numpy(v.copy()) for k, v in state_dict_np.items()}
    missing, unexpected = model.load_state_dict(sd, strict=True), None
    model.eval()
    if dtype is not None:
        model = model.to(dtype)
    return model
