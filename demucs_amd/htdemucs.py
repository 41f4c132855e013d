"""Host-side `HTDemucs`: the object `apply_model` drives, backed by the gfx950 engine.

Mirrors the interface the reference's `apply_model` / `Separator` use on a model
(reference: demucs/htdemucs.py:27-133,511-660, demucs/apply.py:233-236,262-264,304-317):
attributes `.sources .samplerate .audio_channels .segment`, `valid_length`, `to`, `eval`,
`load_state_dict` / `state_dict` with the reference's key schema, and `__call__(mix)` mapping
float32 `(B, 2, n <= segment)` on a GPU device to `(B, S, 2, n)`.

The forward is one `mi_model_forward` call into libdemucs_amd.so; there is no PyTorch
implementation of the network in this package and no CPU path.
"""
from __future__ import annotations

import ctypes as C
import random
from collections import OrderedDict
from fractions import Fraction
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from .weights import HTDemucsConfig, check_reference_keyword, htdemucs_schema

__all__ = ["HTDemucs"]


class HTDemucs:
    #: compute modes of the engine (mi_config.dtype): operand type of the matrix-core GEMMs and of attention;
    #: accumulation, statistics, norms, softmax and the STFT / iSTFT are float32 in every mode
    COMPUTE_DTYPES = {"f32": 0, "bf16": 1, "f16": 2}

    def __init__(self, sources: List[str], segment=Fraction(39, 5), max_batch: int = 8, compute_dtype: str = "f32", **kwargs):
        cfg = HTDemucsConfig(sources=list(sources), segment=Fraction(segment) if not isinstance(segment, Fraction) else segment)
        for k, v in kwargs.items():            # accept the reference's keyword names, reject other architectures
            if hasattr(cfg, k):
                setattr(cfg, k, v)
            elif not check_reference_keyword(k, v, cfg.depth):    # raises when the value selects another architecture
                raise ValueError(f"unsupported HTDemucs argument {k!r}")
        cfg.validate()
        self.cfg = cfg
        self.sources = list(sources)
        self.samplerate = cfg.samplerate
        self.audio_channels = cfg.audio_channels
        self.segment = cfg.segment
        self.use_train_segment = True
        self.max_batch = int(max_batch)
        if compute_dtype not in self.COMPUTE_DTYPES:
            raise ValueError(f"compute_dtype must be one of {sorted(self.COMPUTE_DTYPES)}, got {compute_dtype!r}")
        self.compute_dtype = compute_dtype
        self._schema = htdemucs_schema(cfg)
        self._state: Optional["OrderedDict[str, np.ndarray]"] = None
        self._handles: Dict[torch.device, int] = {}      # engine handle per GPU the model has been used on
        self._device: Optional[torch.device] = None
        self.training = False

    # ---- nn.Module-like surface ---------------------------------------------------------------
    def load_state_dict(self, state: Dict[str, "np.ndarray | torch.Tensor"], strict: bool = True):
        """Accepts the reference checkpoint's `state` (float32 or float16, states.py:83-107)."""
        missing = [k for k in self._schema if k not in state]
        unexpected = [k for k in state if k not in self._schema]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:4]}..., unexpected {unexpected[:4]}..."
                               if len(missing) + len(unexpected) > 8 else
                               f"Error(s) in loading state_dict: missing {missing}, unexpected {unexpected}")
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

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        if self._state is None:
            raise RuntimeError("HTDemucs has no weights: call load_state_dict first")
        return OrderedDict((k, torch.from_numpy(v.copy())) for k, v in self._state.items())

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("demucs_amd.HTDemucs is inference-only")
        return self

    def parameters(self):
        """A single placeholder tensor on the model's device (apply.py:213 asks for the device)."""
        dev = self._device or torch.device("cpu")
        yield torch.empty(0, device=dev)

    def to(self, device):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        # Engine handles are cached per device: `apply_model` moves every sub-model of a bag to the compute device and
        # back to its "home" on each call (apply.py:213-218); re-packing 42 M weights each time would dominate.
        # `release()` frees them explicitly.
        self._device = device
        return self

    def valid_length(self, length: int) -> int:
        """reference: demucs/htdemucs.py:511-525"""
        training_length = int(self.segment * self.samplerate)
        if training_length < length:
            raise ValueError(f"Given length {length} is longer than training length {training_length}")
        return training_length

    @property
    def segment_length(self) -> int:
        return int(self.segment * self.samplerate)

    # ---- engine handle ------------------------------------------------------------------------
    def _release(self):
        handles, self._handles = self._handles, {}
        for dev, h in handles.items():
            with torch.cuda.device(dev):
                _lib.load().mi_model_destroy(C.c_void_p(h))

    def release(self):
        """Free the packed weights (and, with the last handle of its kind, the shared workspace) on every GPU."""
        self._release()

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _ensure_handle(self) -> int:
        if self._device in self._handles:
            return self._handles[self._device]
        if self._state is None:
            raise RuntimeError("HTDemucs has no weights: call load_state_dict first")
        if self._device is None or self._device.type != "cuda":
            raise _lib.EngineError("demucs_amd.HTDemucs only runs on a GPU device (MI355X); call .to('cuda'). "
                                   "There is no CPU implementation in this package.")
        lib = _lib.load()
        names = list(self._state)
        descs = (_lib.MiTensorDesc * len(names))()
        for i, k in enumerate(names):
            v = self._state[k]
            descs[i].name = k.encode()
            descs[i].data = v.ctypes.data
            descs[i].numel = v.size
        cfg = _lib.MiConfig(len(self.sources), self.segment_length, self.max_batch, self.COMPUTE_DTYPES[self.compute_dtype])
        h = C.c_void_p()
        with torch.cuda.device(self._device):
            _lib.check(lib.mi_model_create(C.byref(cfg), descs, len(names), C.byref(h)), "mi_model_create")
        self._handles[self._device] = h.value
        return h.value

    def set_compute_dtype(self, compute_dtype: str):
        if compute_dtype not in self.COMPUTE_DTYPES:
            raise ValueError(f"compute_dtype must be one of {sorted(self.COMPUTE_DTYPES)}, got {compute_dtype!r}")
        if compute_dtype != self.compute_dtype:
            self.compute_dtype = compute_dtype
            self._release()

    def set_max_batch(self, max_batch: int):
        if max_batch != self.max_batch:
            self.max_batch = int(max_batch)
            self._release()

    def device_bytes(self) -> int:
        return int(_lib.load().mi_model_device_bytes(C.c_void_p(self._ensure_handle())))

    def profile_begin(self) -> None:
        _lib.check(_lib.load().mi_profile_begin(C.c_void_p(self._ensure_handle())), "mi_profile_begin")

    def profile_end(self):
        """[{name, launches, ms, flops, bytes}] per kernel class since profile_begin (synchronises)."""
        rows = (_lib.MiProfileRow * 128)()
        n = C.c_int32()
        with torch.cuda.device(self._device):
            _lib.check(_lib.load().mi_profile_end(C.c_void_p(self._ensure_handle()), rows, 128, C.byref(n),
                                                  C.c_void_p(_lib.current_stream_ptr())), "mi_profile_end")
        return [dict(name=rows[i].name.decode(), launches=rows[i].launches, ms=rows[i].ms, flops=rows[i].flops,
                     bytes=rows[i].bytes) for i in range(n.value)]

    def tap(self, name: str, batch: int) -> torch.Tensor:
        """Copy of an internal activation of the last forward (parity tests), shape (batch, numel)."""
        lib, h, n = _lib.load(), C.c_void_p(self._ensure_handle()), C.c_int64()
        _lib.check(lib.mi_model_tap(h, name.encode(), None, batch, C.byref(n), None), "mi_model_tap")
        out = torch.empty(batch, n.value, device=self._device, dtype=torch.float32)
        with torch.cuda.device(self._device):
            _lib.check(lib.mi_model_tap(h, name.encode(), out.data_ptr(), batch, C.byref(n),
                                        C.c_void_p(_lib.current_stream_ptr())), "mi_model_tap")
        return out

    # ---- forward --------------------------------------------------------------------------------
    def forward_segments(self, mix: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """mix (B, 2, segment_length) float32 on the model device -> (B, S, 2, segment_length).
        No RNG side effect (see __call__)."""
        SL = self.segment_length
        if mix.dim() != 3 or mix.shape[1] != self.audio_channels or mix.shape[2] != SL:
            raise ValueError(f"expected (B, {self.audio_channels}, {SL}), got {tuple(mix.shape)}")
        if mix.dtype != torch.float32:
            raise TypeError("mix must be float32")
        handle = self._ensure_handle()
        if mix.device != self._device:
            raise _lib.EngineError(f"mix is on {mix.device} but the model is on {self._device}")
        mix = mix.contiguous()
        B = mix.shape[0]
        if out is None:
            out = torch.empty(B, len(self.sources), self.audio_channels, SL, device=mix.device, dtype=torch.float32)
        lib = _lib.load()
        with torch.cuda.device(self._device):
            for b0 in range(0, B, self.max_batch):
                nb = min(self.max_batch, B - b0)
                _lib.check(lib.mi_model_forward(C.c_void_p(handle), mix[b0:b0 + nb].data_ptr(), out[b0:b0 + nb].data_ptr(),
                                                nb, C.c_void_p(_lib.current_stream_ptr())), "mi_model_forward")
        return out

    def forward_core(self, mag: Optional[torch.Tensor], mix: torch.Tensor):
        """HTDemucs.forward_core (htdemucs.py:662-759): (spec_out (B,S,4,2048,T), time_out (B,S,2,L)).
        `mag` (B,4,2048,T) is the caller's `_magnitude(_spec(mix))` and feeds the frequency branch as given (the
        fork's ONNX / web tools compute it with their own STFT); None = the engine's own STFT of `mix`."""
        SL = self.segment_length
        if mix.dim() != 3 or mix.shape[1] != self.audio_channels or mix.shape[2] != SL or mix.dtype != torch.float32:
            raise ValueError(f"expected float32 (B, {self.audio_channels}, {SL}), got {tuple(mix.shape)} {mix.dtype}")
        B, S, T = mix.shape[0], len(self.sources), -(-SL // 1024)
        if mag is not None:
            if tuple(mag.shape) != (B, 4, 2048, T) or mag.dtype != torch.float32:
                raise ValueError(f"mag must be float32 (B, 4, 2048, {T}), got {tuple(mag.shape)} {mag.dtype}")
            if mag.device != mix.device:
                raise ValueError("mag and mix must live on the same device")
            mag = mag.contiguous()
        handle = self._ensure_handle()
        if mix.device != self._device:
            raise _lib.EngineError(f"mix is on {mix.device} but the model is on {self._device}")
        mix = mix.contiguous()
        spec = torch.empty(B, S, 4, 2048, T, device=mix.device, dtype=torch.float32)
        tout = torch.empty(B, S, self.audio_channels, SL, device=mix.device, dtype=torch.float32)
        lib = _lib.load()
        with torch.cuda.device(self._device):
            for b0 in range(0, B, self.max_batch):
                nb = min(self.max_batch, B - b0)
                _lib.check(lib.mi_model_forward_core(C.c_void_p(handle), mix[b0:b0 + nb].data_ptr(),
                                                     mag[b0:b0 + nb].data_ptr() if mag is not None else None,
                                                     spec[b0:b0 + nb].data_ptr(), tout[b0:b0 + nb].data_ptr(), nb,
                                                     C.c_void_p(_lib.current_stream_ptr())), "mi_model_forward_core")
        return spec, tout

    def __call__(self, mix: torch.Tensor) -> torch.Tensor:
        """HTDemucs.forward in eval mode (htdemucs.py:527-660): shorter inputs are right-padded with
        zeros to the training length and the output cropped back."""
        random.randrange(1)      # the reference draws its sin-embedding shift from Python's global
        #                          RNG on every forward (transformer.py:680); keep `shifts` offsets in step
        length = mix.shape[-1]
        SL = self.segment_length
        if length > SL:
            raise ValueError(f"Given length {length} is longer than training length {SL}")
        if length < SL:
            mix = torch.nn.functional.pad(mix, (0, SL - length))
        out = self.forward_segments(mix)
        return out[..., :length] if length < SL else out

    forward = __call__
