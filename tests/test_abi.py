"""The C-ABI shared library loads on a machine without a GPU and exports every symbol that
include/demucs_amd.h declares (no compute calls here)."""
import ctypes
import os
import re

from demucs_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "demucs_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 15 and "mi_model_forward" in names
    lib = _lib.load()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/demucs_amd.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert lib.mi_version().startswith(b"demucs_amd")


def test_error_path_without_gpu_work():
    lib = _lib.load()
    rc = lib.mi_model_forward(None, None, None, 1, None)           # null handle: rejected before any HIP call
    assert rc == -1 and b"null handle" in lib.mi_last_error()
    assert ctypes.sizeof(_lib.MiConvDesc) % 8 == 0
