"""`demucs.pretrained.get_model` for the MI355X engine (reference: demucs/pretrained.py:23-85).

Offline there is no model zoo (the reference fetches checkpoints from dl.fbaipublicfiles.com): what remains is
  * `get_model("demucs_unittest")` -- the reference builds `HDemucs(channels=4, sources=SOURCES)` with random weights
    (pretrained.py:27-29, the model its own CI separates a test file with); here the same architecture on the engine with this
    repo's deterministic weight fill, so that two calls give the same model;
  * `get_model(name, repo=folder)` -- packages / bag YAMLs of a local folder (demucs_amd.states.LocalRepo; nothing is downloaded).
Any other name without a repo raises `ModelLoadingError`: the network fetch is out of scope.
"""
from pathlib import Path
from typing import Optional, Union

from .hdemucs import HDemucs
from .hdemucs_weights import HDemucsConfig, synthetic_hdemucs_state_dict
from .states import LocalRepo, ModelLoadingError

SOURCES = ["drums", "bass", "other", "vocals"]
DEFAULT_MODEL = "htdemucs"

__all__ = ["SOURCES", "DEFAULT_MODEL", "demucs_unittest", "get_model"]


def demucs_unittest(max_batch: int = 1, weight_seed: int = 0) -> HDemucs:
    """pretrained.py:27-29."""
    cfg = HDemucsConfig(sources=list(SOURCES), channels=4)
    model = HDemucs(cfg.sources, max_batch=max_batch, channels=4)
    model.load_state_dict(synthetic_hdemucs_state_dict(cfg, weight_seed))
    return model


def get_model(name: str, repo: Optional[Union[str, Path]] = None, max_batch: int = 8):
    """pretrained.py:59-85 without the remote repository."""
    if name == "demucs_unittest":
        return demucs_unittest()
    if repo is None:
        raise ModelLoadingError(f"pre-trained model {name!r} lives in the reference's remote model zoo, which cannot be fetched here: "
                                "pass repo=<folder with the .th packages / bag .yaml files>")
    repo = Path(repo)
    if not repo.is_dir():
        raise ModelLoadingError(f"{repo} must exist and be a directory.")
    model = LocalRepo(repo, max_batch=max_batch).get_model(name)
    model.eval()
    return model
