"""Host-side `HDemucs`: the Hybrid Demucs v3 model object (`hdemucs_mmi` architecture) backed by the gfx950 engine.

Mirrors what `apply_model` / `BagOfModels` / `Separator` use on the reference's `demucs.hdemucs.HDemucs`
(reference: demucs/hdemucs.py:338-794): attributes `.sources .samplerate .audio_channels .segment`, NO `valid_length`
(so the leaf hands every chunk over at its own length, demucs/apply.py:309-310), `to`, `eval`, `load_state_dict` /
`state_dict` with the reference's key schema (395 tensors), and `__call__(mix)` mapping float32 `(B, 2, n)` on a GPU to
`(B, S, 2, n)` for any n >= 1 (short inputs follow pad1d's zero-then-reflect rule, hdemucs.py:29-36).  One `mi_hmodel_forward` call per forward; no PyTorch
implementation of the network exists in this package and there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from .hdemucs_weights import HDemucsConfig, hdemucs_schema

__all__ = ["HDemucs"]

MIN_LENGTH = 1               # demucs_amd/csrc/hmodel.hip: kMinLength (the reference forwards any length >= 1 too)


class HDemucs:
    COMPUTE_DTYPES = {"f32": 0, "bf16": 1, "f16": 2}

    #: reference keywords that cannot change an eval-mode forward of this architecture
    _INERT = frozenset(("rescale", "dconv_init", "emb_smooth", "wiener_iters", "end_iters", "wiener_residual", "multi_freqs_depth"))
    #: reference keywords whose value the engine hard-codes
    _FIXED = dict(channels_time=None, cac=True, rewrite=True, hybrid=True, hybrid_old=False, multi_freqs=None)

    def __init__(self, sources: List[str], max_batch: int = 1, compute_dtype: str = "f32", **kwargs):
        cfg = HDemucsConfig(sources=list(sources))
        for k, v in kwargs.items():
            if hasattr(cfg, k):
                setattr(cfg, k, v)
            elif k in self._INERT:
                continue
            elif k in self._FIXED:
                if not (v == self._FIXED[k] or (k == "multi_freqs" and not v)):
                    raise ValueError(f"unsupported HDemucs argument {k}={v!r}: the MI355X engine implements {k}={self._FIXED[k]!r}")
            else:
                raise ValueError(f"unsupported HDemucs argument {k!r}")
        cfg.validate()
        if compute_dtype not in self.COMPUTE_DTYPES:
            raise ValueError(f"compute_dtype must be one of {sorted(self.COMPUTE_DTYPES)}, got {compute_dtype!r}")
        self.cfg = cfg
        self.sources = list(sources)
        self.samplerate = cfg.samplerate
        self.audio_channels = cfg.audio_channels
        self.segment = cfg.segment                  # a bag may raise it (BagOfModels: remote/hdemucs_mmi.yaml sets 44)
        self.max_batch = int(max_batch)
        self.compute_dtype = compute_dtype
        self._schema = hdemucs_schema(cfg)
        self._state: Optional["OrderedDict[str, np.ndarray]"] = None
        self._handles: Dict[tuple, tuple] = {}          # (device, aux) -> (handle, max_length); aux: the single-item side engine
        self._side_streams: Dict[torch.device, "torch.cuda.Stream"] = {}
        self._device: Optional[torch.device] = None
        self.training = False

    # ---- nn.Module-like surface ---------------------------------------------------------------------------------
    def load_state_dict(self, state, strict: bool = True):
        missing = [k for k in self._schema if k not in state]
        unexpected = [k for k in state if k not in self._schema]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:4]}, unexpected {unexpected[:4]}")
        new: "OrderedDict[str, np.ndarray]" = OrderedDict()
        for k, shape in self._schema.items():
            v = state[k]
            if isinstance(v, torch.Tensor):
                v = v.detach().to("cpu", torch.float32).numpy()
            v = np.ascontiguousarray(v, dtype=np.float32)
            if tuple(v.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(v.shape)} vs model {tuple(shape)}")
            new[k] = v
        self._state = new
        self._release()
        return self

    def state_dict(self):
        if self._state is None:
            raise RuntimeError("HDemucs has no weights: call load_state_dict first")
        return OrderedDict((k, torch.from_numpy(v.copy())) for k, v in self._state.items())

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("demucs_amd.HDemucs is inference-only")
        return self

    def parameters(self):
        yield torch.empty(0, device=self._device or torch.device("cpu"))

    def to(self, device):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self._device = device          # handles are cached per device (see HTDemucs.to)
        return self

    # ---- engine handle ------------------------------------------------------------------------------------------
    def _release(self):
        handles, self._handles = self._handles, {}
        for (dev, _aux), (h, _) in handles.items():
            with torch.cuda.device(dev):
                _lib.load().mi_hmodel_destroy(C.c_void_p(h))

    release = _release

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _ensure_handle(self, length: int, aux: bool = False) -> int:
        if self._state is None:
            raise RuntimeError("HDemucs has no weights: call load_state_dict first")
        if self._device is None or self._device.type != "cuda":
            raise _lib.EngineError("demucs_amd.HDemucs only runs on a GPU device (MI355X); call .to('cuda'). "
                                   "There is no CPU implementation in this package.")
        key = (self._device, bool(aux))
        have = self._handles.get(key)
        if have is not None and have[1] >= length:
            return have[0]
        lib = _lib.load()
        if have is not None:                            # a longer chunk than the workspace was sized for: re-create
            with torch.cuda.device(self._device):
                lib.mi_hmodel_destroy(C.c_void_p(have[0]))
            del self._handles[key]
        max_length = max(int(length), int(float(self.segment) * self.samplerate), MIN_LENGTH)
        names = list(self._state)
        descs = (_lib.MiTensorDesc * len(names))()
        for i, k in enumerate(names):
            v = self._state[k]
            descs[i].name = k.encode()
            descs[i].data = v.ctypes.data
            descs[i].numel = v.size
        cfg = _lib.MiConfig(len(self.sources), max_length, 1 if aux else self.max_batch, self.COMPUTE_DTYPES[self.compute_dtype])
        h = C.c_void_p()
        with torch.cuda.device(self._device):
            _lib.check(lib.mi_hmodel_create(C.byref(cfg), descs, len(names), C.byref(h)), "mi_hmodel_create")
        self._handles[key] = (h.value, max_length)
        return h.value

    def side_stream(self) -> "torch.cuda.Stream":
        """Stream on which `apply_model` runs a track's shorter tail chunk (on the single-item side engine: `forward(..., aux=True)`)
        while the batched forward of the full chunks runs on the caller's stream."""
        st = self._side_streams.get(self._device)
        if st is None:
            st = self._side_streams[self._device] = torch.cuda.Stream(self._device)
        return st

    def device_bytes(self) -> int:
        lib = _lib.load()
        return sum(int(lib.mi_hmodel_device_bytes(C.c_void_p(h))) for (dev, _aux), (h, _) in self._handles.items() if dev == self._device)

    def check(self) -> None:
        """Wait for the current stream and raise if a forward of this model lost its BLSTM recurrence to a time-out (the engine
        notices by itself only when the NEXT forward starts): `apply_model` calls this before it hands a track's stems on."""
        lib = _lib.load()
        for (dev, _aux), (h, _) in self._handles.items():
            if dev == self._device:
                _lib.check(lib.mi_hmodel_status(C.c_void_p(h), C.c_void_p(_lib.current_stream_ptr())), "mi_hmodel_status")

    def profile_begin(self) -> None:
        """Per-kernel-class HIP-event timing of the MAIN engine handle from here to profile_end (bench.py)."""
        h = self._handles[(self._device, False)][0]
        _lib.check(_lib.load().mi_profile_begin(C.c_void_p(h)), "mi_profile_begin")

    def profile_end(self):
        rows = (_lib.MiProfileRow * 128)()
        n = C.c_int32()
        h = self._handles[(self._device, False)][0]
        with torch.cuda.device(self._device):
            _lib.check(_lib.load().mi_profile_end(C.c_void_p(h), rows, 128, C.byref(n), C.c_void_p(_lib.current_stream_ptr())),
                       "mi_profile_end")
        return [dict(name=rows[i].name.decode(), launches=rows[i].launches, ms=rows[i].ms, flops=rows[i].flops, bytes=rows[i].bytes)
                for i in range(n.value)]

    def tap(self, name: str, batch: int) -> torch.Tensor:
        """Copy of an internal activation of the last forward (parity tests), shape (batch, numel)."""
        lib, n = _lib.load(), C.c_int64()
        h = C.c_void_p(self._handles[(self._device, False)][0])
        _lib.check(lib.mi_hmodel_tap(h, name.encode(), None, batch, C.byref(n), None), "mi_hmodel_tap")
        out = torch.empty(batch, n.value, device=self._device, dtype=torch.float32)
        with torch.cuda.device(self._device):
            _lib.check(lib.mi_hmodel_tap(h, name.encode(), out.data_ptr(), batch, C.byref(n), C.c_void_p(_lib.current_stream_ptr())),
                       "mi_hmodel_tap")
        return out

    # ---- forward --------------------------------------------------------------------------------------------------
    def __call__(self, mix: torch.Tensor, aux: bool = False) -> torch.Tensor:
        """HDemucs.forward in eval mode (hdemucs.py:689-794): float32 (B, 2, n) -> (B, S, 2, n), any n >= 1.
        aux=True runs on the single-item side engine (its own workspace), so that it may overlap a forward of the main one
        on another stream; the results are the same bit for bit."""
        if mix.dim() != 3 or mix.shape[1] != self.audio_channels:
            raise ValueError(f"expected (B, {self.audio_channels}, n), got {tuple(mix.shape)}")
        if mix.dtype != torch.float32:
            raise TypeError("mix must be float32")
        B, _, length = mix.shape
        if length < MIN_LENGTH:
            raise ValueError(f"demucs_amd.HDemucs needs at least {MIN_LENGTH} samples per forward, got {length}")
        handle = self._ensure_handle(length, aux)
        if mix.device != self._device:
            raise _lib.EngineError(f"mix is on {mix.device} but the model is on {self._device}")
        mix = mix.contiguous()
        out = torch.empty(B, len(self.sources), self.audio_channels, length, device=mix.device, dtype=torch.float32)
        lib = _lib.load()
        with torch.cuda.device(self._device):
            step = 1 if aux else self.max_batch
            for b0 in range(0, B, step):
                nb = min(step, B - b0)
                _lib.check(lib.mi_hmodel_forward(C.c_void_p(handle), mix[b0:b0 + nb].data_ptr(), out[b0:b0 + nb].data_ptr(), nb, length,
                                                 C.c_void_p(_lib.current_stream_ptr())), "mi_hmodel_forward")
        return out

    forward = __call__
