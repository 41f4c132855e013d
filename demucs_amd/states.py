"""Checkpoint packages of the reference (`demucs.states`, demucs/states.py:50-107) for the MI355X engine.

The reference stores `{"klass": <class>, "args": tuple, "kwargs": dict, "state": state_dict, ...}` with torch.save
(serialize_model, states.py:138-157) and loads it with an unrestricted `torch.load` (load_model, states.py:50-80).
Here the file is read with `torch.load(..., weights_only=True)`: nothing from the file is executed; the class reference is
resolved to an inert stand-in that only carries the qualified name, and `fractions.Fraction` (the `segment` keyword) is the
only callable allowed to be re-created.  Quantised states (`__quantized`, diffq) are refused: diffq is not available offline.

No released checkpoint can be fetched in this environment.  PINNED by full-size packages that the reference's own
`serialize_model` + `torch.save` wrote around reference models built with every keyword of conf/config.yaml
(tests/golden/pkg_htdemucs.th, pkg_hdemucs.th: tools/make_golden.py `package_fixture`), whose GPU forward through this reader
equals the reference's, and by the reference constructors' signatures (tests/golden/ref_signatures.json) against which the
keyword tables of weights.py / hdemucs.py are checked.
"""
from __future__ import annotations

import warnings
from fractions import Fraction
from pathlib import Path
from typing import Union

import torch

from .hdemucs import HDemucs
from .htdemucs import HTDemucs
from .weights import REFERENCE_DEFAULTS, check_reference_keyword

#: qualified names of the reference classes a package may name -> engine class that takes the same keywords
SUPPORTED = {"demucs.htdemucs.HTDemucs": HTDemucs, "demucs.hdemucs.HDemucs": HDemucs}


def _stub(qualname: str):
    """Inert class object whose pickled GLOBAL reference is `qualname`: lets weights_only=True unpickling resolve the
    package's `klass` without importing (or executing) anything from the reference."""
    module, name = qualname.rsplit(".", 1)
    return type(name, (), {"__module__": module, "__qualname__": name, "_mi_qualname": qualname})


_STUBS = {q: _stub(q) for q in SUPPORTED}
#: other reference model classes: recognised so that the error names them instead of failing inside the unpickler
_KNOWN_UNSUPPORTED = {q: _stub(q) for q in ("demucs.demucs.Demucs",)}


def _qualname(klass) -> str:
    if isinstance(klass, str):
        return klass
    return getattr(klass, "_mi_qualname", f"{getattr(klass, '__module__', '?')}.{getattr(klass, '__qualname__', klass)}")


def read_package(path: Union[str, Path]) -> dict:
    """The package dict of a reference checkpoint file, deserialised without executing anything from it."""
    allow = list(_STUBS.values()) + list(_KNOWN_UNSUPPORTED.values()) + [Fraction]
    with torch.serialization.safe_globals(allow):
        package = torch.load(str(path), map_location="cpu", weights_only=True)
    if not isinstance(package, dict) or not {"klass", "args", "kwargs", "state"} <= set(package):
        raise ValueError(f"{path}: not a demucs checkpoint package (expected klass / args / kwargs / state)")
    return package


def load_model(path_or_package, strict: bool = False, max_batch: int = 8):
    """demucs/states.py:50-80 for the engine: a dict (already loaded) or a path.  Unknown keywords are dropped with the
    reference's warning unless `strict`; architectures the engine does not implement raise ValueError."""
    if isinstance(path_or_package, dict):
        package = path_or_package
    elif isinstance(path_or_package, (str, Path)):
        package = read_package(path_or_package)
    else:
        raise ValueError(f"Invalid type for {path_or_package}.")
    qual = _qualname(package["klass"])
    if qual not in SUPPORTED:
        raise ValueError(f"checkpoint class {qual} is not implemented by the MI355X engine (supported: {sorted(SUPPORTED)})")
    klass = SUPPORTED[qual]
    args, kwargs = tuple(package["args"]), dict(package["kwargs"])
    if len(args) > 1:
        raise ValueError("checkpoint package passes hyper-parameters positionally; only `sources` may be positional")
    if klass is HDemucs:
        # HDemucs: the engine class itself sorts the reference's keywords (config field / inert / fixed value / unknown);
        # unknown names follow the reference's warn-and-drop rule unless `strict`.  Omitted keywords mean the reference's
        # defaults, which ARE the hdemucs_mmi architecture the engine implements.
        known = set(vars(HDemucs(["_"]).cfg)) | HDemucs._INERT | set(HDemucs._FIXED)
        for key in list(kwargs):
            if key not in known and key != "sources":
                if strict:
                    raise ValueError(f"unknown HDemucs keyword {key!r} in the checkpoint package")
                warnings.warn("Dropping inexistant parameter " + key)
                del kwargs[key]
        model = klass(*args, max_batch=min(max_batch, 16), **kwargs)
        set_state(model, package["state"])
        return model
    cfg_fields = set(vars(klass(["_"]).cfg))
    for key in list(kwargs):
        if key in cfg_fields or key == "sources":
            continue
        # a reference keyword outside the engine's config: honoured by construction (dropped silently) or a
        # ValueError when its value selects an architecture the engine does not implement -- never dropped blindly,
        # because load_state_dict cannot notice flags that keep every tensor shape (t_cross_first, t_norm_first, ...)
        if check_reference_keyword(key, kwargs[key], kwargs.get("depth", REFERENCE_DEFAULTS["depth"])):
            del kwargs[key]
        elif strict:
            raise ValueError(f"unknown HTDemucs keyword {key!r} in the checkpoint package")
        else:                                   # unknown to the reference's constructor too: its warn-and-drop rule
            warnings.warn("Dropping inexistant parameter " + key)
            del kwargs[key]
    # keywords the package omits mean the REFERENCE's defaults (segment=10, dconv_mode=1, bottom_channels=0, ...),
    # not the released-model values the engine class defaults to
    for key, default in REFERENCE_DEFAULTS.items():
        kwargs.setdefault(key, default)
    model = klass(*args, max_batch=max_batch, **kwargs)
    set_state(model, package["state"])
    return model


def set_state(model: HTDemucs, state: dict) -> dict:
    """demucs/states.py:97-107 (float32 or float16 state dicts; diffq-quantised states are refused)."""
    if state.get("__quantized"):
        raise ValueError("diffq-quantised checkpoints are not supported (diffq is unavailable offline)")
    model.load_state_dict(state)
    return state


# ---- local model repository (demucs/repo.py:75-146: LocalRepo + BagOnlyRepo) ------------------------------------------
class ModelLoadingError(RuntimeError):
    """demucs/repo.py:26."""


def check_checksum(path: Path, checksum: str):
    """demucs/repo.py:29-40: the suffix in the file name is the leading hex digits of the file's sha256."""
    import hashlib
    with open(path, "rb") as file:
        digest = hashlib.file_digest(file, "sha256").hexdigest() if hasattr(hashlib, "file_digest") else \
            hashlib.sha256(file.read()).hexdigest()
    if not digest.startswith(checksum):
        raise ModelLoadingError(f"Invalid checksum for file {path}, expected {checksum} but got {digest[:len(checksum)]}")


class LocalRepo:
    """A folder of `<signature>[-<sha256 prefix>].th` packages and `<name>.yaml` bags, as `demucs.pretrained.get_model(name,
    repo=folder)` reads it (demucs/pretrained.py:59-85 with LocalRepo / BagOnlyRepo / AnyModelRepo).  Nothing is downloaded."""

    def __init__(self, root: Union[str, Path], max_batch: int = 8):
        self.root, self.max_batch = Path(root), max_batch
        self._models, self._checksums, self._bags = {}, {}, {}
        for file in sorted(self.root.iterdir()):
            if file.suffix == ".th":
                if "-" in file.stem:
                    sig, checksum = file.stem.split("-")
                    self._checksums[sig] = checksum
                else:
                    sig = file.stem
                if sig in self._models:
                    raise ModelLoadingError(f"Duplicate pre-trained model exist for signature {sig}. Please delete all but one.")
                self._models[sig] = file
            elif file.suffix == ".yaml":
                self._bags[file.stem] = file

    def has_model(self, name_or_sig: str) -> bool:
        return name_or_sig in self._models or name_or_sig in self._bags

    def _single(self, sig: str) -> HTDemucs:
        try:
            file = self._models[sig]
        except KeyError:
            raise ModelLoadingError(f"Could not find pre-trained model with signature {sig}.")
        if sig in self._checksums:
            check_checksum(file, self._checksums[sig])
        return load_model(file, max_batch=self.max_batch)

    def get_model(self, name_or_sig: str):
        """A single model by signature, or a `BagOfModels` by YAML name (models / weights / segment keys)."""
        import yaml

        from .apply import BagOfModels
        if name_or_sig in self._models:
            return self._single(name_or_sig)
        if name_or_sig not in self._bags:
            raise ModelLoadingError(f"{name_or_sig} is neither a single pre-trained model or a bag of models.")
        with open(self._bags[name_or_sig]) as f:
            bag = yaml.safe_load(f)
        models = [self._single(sig) for sig in bag["models"]]
        return BagOfModels(models, bag.get("weights"), bag.get("segment"))
