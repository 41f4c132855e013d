"""State-dict schema of the Hybrid Demucs v3 architecture (`hdemucs_mmi`: the reference's `HDemucs` with its default
hyper-parameters, channels=48, depth=6) and the deterministic synthetic weight fill for it.

The schema mirrors `HDemucs(sources).state_dict()` of the reference (demucs/hdemucs.py:366-580 for the layer plan,
demucs/hdemucs.py:69-122,256-302 for the layer modules, demucs/demucs.py:20-33,133-149,157-181 for the DConv branch
with its BLSTM and LocalState): 395 tensors, 83.6 M parameters for 4 sources; `tools/make_golden.py` asserts name and
shape equality against the imported reference.  Weights come from the same name-keyed counter fill as the htdemucs
family (`weights.py`), with LayerScale O(1) so that parity is not blind to the DConv / LSTM / attention branches.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

from .weights import _name_seed, periodic_uniform

__all__ = ["HDemucsConfig", "hdemucs_schema", "hdemucs_layer_plan", "synthetic_hdemucs_state_dict"]


@dataclass
class HDemucsConfig:
    """Hyper-parameters of the reference constructor (demucs/hdemucs.py:366-410); defaults = `hdemucs_mmi`
    (conf/config.yaml:126-165, remote/hdemucs_mmi.yaml)."""
    sources: List[str] = field(default_factory=lambda: ["drums", "bass", "other", "vocals"])
    audio_channels: int = 2
    channels: int = 48
    growth: int = 2
    nfft: int = 4096
    depth: int = 6
    kernel_size: int = 8
    stride: int = 4
    time_stride: int = 2
    context: int = 1
    context_enc: int = 0
    norm_starts: int = 4
    norm_groups: int = 4
    dconv_mode: int = 1
    dconv_depth: int = 2
    dconv_comp: int = 4
    dconv_attn: int = 4
    dconv_lstm: int = 4
    freq_emb: float = 0.2
    emb_scale: float = 10.0
    samplerate: int = 44100
    segment: float = 40

    def validate(self) -> None:
        bad = []
        if self.audio_channels != 2: bad.append("audio_channels")
        # channels: 48 = hdemucs_mmi (matrix-pipe LSTM / LocalState kernels); 4 = the reference's own `demucs_unittest` model
        # (pretrained.py:27-29), same engine with the generic recurrence / attention kernels (hidden <= 32, head dimension <= 8)
        if self.channels not in (4, 48) or self.growth != 2: bad.append("channels/growth")
        if self.nfft != 4096 or self.depth != 6: bad.append("nfft/depth")
        if self.kernel_size != 8 or self.stride != 4 or self.time_stride != 2: bad.append("kernel_size/stride/time_stride")
        if self.context != 1 or self.context_enc != 0: bad.append("context")
        if self.norm_starts != 4 or self.norm_groups != 4: bad.append("norm_*")
        if (self.dconv_mode, self.dconv_depth, self.dconv_comp, self.dconv_attn, self.dconv_lstm) != (1, 2, 4, 4, 4):
            bad.append("dconv_*")
        if float(self.freq_emb) != 0.2 or float(self.emb_scale) != 10.0: bad.append("freq_emb/emb_scale")
        if self.samplerate != 44100: bad.append("samplerate")
        if not 1 <= len(self.sources) <= 8: bad.append("sources")
        if bad:
            raise ValueError("unsupported HDemucs hyper-parameters for the MI355X path: " + ", ".join(bad))


def hdemucs_layer_plan(cfg: HDemucsConfig):
    """Per encoder index the facts the reference constructor derives (demucs/hdemucs.py:476-578):
    dict(freq, last_freq, ker, stri, pad, norm, lstm, attn, chin_z, chout_z, chin, chout, has_tenc, tenc_empty)."""
    plan = []
    S = len(cfg.sources)
    chin = cfg.audio_channels
    chin_z = 2 * chin
    chout = chout_z = cfg.channels
    freqs = cfg.nfft // 2
    for index in range(cfg.depth):
        freq = freqs > 1
        ker, stri = (cfg.kernel_size, cfg.stride) if freq else (cfg.time_stride * 2, cfg.time_stride)
        pad, last_freq = True, False
        if freq and freqs <= cfg.kernel_size:
            ker, pad, last_freq = freqs, False, True
        if last_freq:
            chout_z = max(chout, chout_z)
            chout = chout_z
        plan.append(dict(index=index, freq=freq, last_freq=last_freq, ker=ker, stri=stri, pad=pad, norm=index >= cfg.norm_starts,
                         lstm=index >= cfg.dconv_lstm, attn=index >= cfg.dconv_attn, chin_z=chin_z, chout_z=chout_z, chin=chin,
                         chout=chout, has_tenc=freq, tenc_empty=last_freq, freqs_in=freqs))
        if index == 0:
            chin = cfg.audio_channels * S
            chin_z = 2 * chin
        plan[-1]["dec_out_z"], plan[-1]["dec_out"] = chin_z, chin          # decoder mirror: chout_z -> chin_z (as updated at index 0)
        chin, chin_z = chout, chout_z
        chout, chout_z = int(cfg.growth * chout), int(cfg.growth * chout_z)
        if freq:
            freqs = 1 if freqs <= cfg.kernel_size else freqs // cfg.stride
    return plan


def _dconv_schema(out, prefix: str, C: int, comp: int, depth: int, lstm: bool, attn: bool):
    h = int(C / comp)
    for d in range(depth):
        p = f"{prefix}.dconv.layers.{d}"
        out[f"{p}.0.weight"] = (h, C, 3); out[f"{p}.0.bias"] = (h,)
        out[f"{p}.1.weight"] = (h,); out[f"{p}.1.bias"] = (h,)
        i = 3
        if lstm:                                   # BLSTM(hidden, layers=2) + Linear(2h, h)  (demucs.py:26-33)
            for layer in range(2):
                for suffix in ("", "_reverse"):
                    out[f"{p}.{i}.lstm.weight_ih_l{layer}{suffix}"] = (4 * h, h if layer == 0 else 2 * h)
                    out[f"{p}.{i}.lstm.weight_hh_l{layer}{suffix}"] = (4 * h, h)
                    out[f"{p}.{i}.lstm.bias_ih_l{layer}{suffix}"] = (4 * h,)
                    out[f"{p}.{i}.lstm.bias_hh_l{layer}{suffix}"] = (4 * h,)
            out[f"{p}.{i}.linear.weight"] = (h, 2 * h); out[f"{p}.{i}.linear.bias"] = (h,)
            i += 1
        if attn:                                   # LocalState(hidden, heads=4, ndecay=4)  (demucs.py:162-181)
            for n, m in (("content", h), ("query", h), ("key", h), ("query_decay", 16), ("proj", h)):
                out[f"{p}.{i}.{n}.weight"] = (m, h, 1); out[f"{p}.{i}.{n}.bias"] = (m,)
            i += 1
        out[f"{p}.{i}.weight"] = (2 * C, h, 1); out[f"{p}.{i}.bias"] = (2 * C,)
        out[f"{p}.{i + 1}.weight"] = (2 * C,); out[f"{p}.{i + 1}.bias"] = (2 * C,)
        out[f"{p}.{i + 3}.scale"] = (C,)


def hdemucs_schema(cfg: HDemucsConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """name -> shape, in the reference's registration order (encoder, decoder, tencoder, tdecoder, freq_emb)."""
    cfg.validate()
    plan = hdemucs_layer_plan(cfg)
    enc: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    dec: List["OrderedDict[str, Tuple[int, ...]]"] = []
    tenc: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    tdec: List["OrderedDict[str, Tuple[int, ...]]"] = []
    ctx = 1 + 2 * cfg.context
    for L in plan:
        i, p = L["index"], f"encoder.{L['index']}"
        kshape = (L["ker"], 1) if L["freq"] else (L["ker"],)
        one = (1, 1) if L["freq"] else (1,)
        enc[f"{p}.conv.weight"] = (L["chout_z"], L["chin_z"]) + kshape; enc[f"{p}.conv.bias"] = (L["chout_z"],)
        if L["norm"]:
            enc[f"{p}.norm1.weight"] = (L["chout_z"],); enc[f"{p}.norm1.bias"] = (L["chout_z"],)
        enc[f"{p}.rewrite.weight"] = (2 * L["chout_z"], L["chout_z"]) + one; enc[f"{p}.rewrite.bias"] = (2 * L["chout_z"],)
        if L["norm"]:
            enc[f"{p}.norm2.weight"] = (2 * L["chout_z"],); enc[f"{p}.norm2.bias"] = (2 * L["chout_z"],)
        if cfg.dconv_mode & 1:
            _dconv_schema(enc, p, L["chout_z"], cfg.dconv_comp, cfg.dconv_depth, L["lstm"], L["attn"])
        d: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
        q = "decoder.{j}"
        d[f"{q}.conv_tr.weight"] = (L["chout_z"], L["dec_out_z"]) + kshape; d[f"{q}.conv_tr.bias"] = (L["dec_out_z"],)
        if L["norm"]:
            d[f"{q}.norm2.weight"] = (L["dec_out_z"],); d[f"{q}.norm2.bias"] = (L["dec_out_z"],)
        rk = (ctx, ctx) if L["freq"] else (ctx,)
        d[f"{q}.rewrite.weight"] = (2 * L["chout_z"], L["chout_z"]) + rk; d[f"{q}.rewrite.bias"] = (2 * L["chout_z"],)
        if L["norm"]:
            d[f"{q}.norm1.weight"] = (2 * L["chout_z"],); d[f"{q}.norm1.bias"] = (2 * L["chout_z"],)
        dec.insert(0, d)
        if L["has_tenc"]:
            pt = f"tencoder.{i}"
            tenc[f"{pt}.conv.weight"] = (L["chout"], L["chin"], cfg.kernel_size); tenc[f"{pt}.conv.bias"] = (L["chout"],)
            td: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
            qt = "tdecoder.{j}"
            td[f"{qt}.conv_tr.weight"] = (L["chout"], L["dec_out"], cfg.kernel_size); td[f"{qt}.conv_tr.bias"] = (L["dec_out"],)
            if L["norm"]:
                td[f"{qt}.norm2.weight"] = (L["dec_out"],); td[f"{qt}.norm2.bias"] = (L["dec_out"],)
            if not L["tenc_empty"]:
                if L["norm"]:
                    tenc[f"{pt}.norm1.weight"] = (L["chout"],); tenc[f"{pt}.norm1.bias"] = (L["chout"],)
                tenc[f"{pt}.rewrite.weight"] = (2 * L["chout"], L["chout"], 1); tenc[f"{pt}.rewrite.bias"] = (2 * L["chout"],)
                if L["norm"]:
                    tenc[f"{pt}.norm2.weight"] = (2 * L["chout"],); tenc[f"{pt}.norm2.bias"] = (2 * L["chout"],)
                if cfg.dconv_mode & 1:
                    _dconv_schema(tenc, pt, L["chout"], cfg.dconv_comp, cfg.dconv_depth, L["lstm"], L["attn"])
                td[f"{qt}.rewrite.weight"] = (2 * L["chout"], L["chout"], ctx); td[f"{qt}.rewrite.bias"] = (2 * L["chout"],)
                if L["norm"]:
                    td[f"{qt}.norm1.weight"] = (2 * L["chout"],); td[f"{qt}.norm1.bias"] = (2 * L["chout"],)
            tdec.insert(0, td)
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict(enc)
    for j, d in enumerate(dec):
        for k, v in d.items():
            out[k.format(j=j)] = v
    out.update(tenc)
    for j, d in enumerate(tdec):
        for k, v in d.items():
            out[k.format(j=j)] = v
    out["freq_emb.embedding.weight"] = (cfg.nfft // 2 // cfg.stride, cfg.channels)
    return out


def synthetic_hdemucs_state_dict(cfg: HDemucsConfig, seed: int = 0, period=None) -> "OrderedDict[str, np.ndarray]":
    """Deterministic float32 weights keyed by tensor name; gains keep every stage O(1) (checked when the goldens are made)."""
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in hdemucs_schema(cfg).items():
        n = int(np.prod(shape))
        u = periodic_uniform(_name_seed(name, seed), n, period) * 2.0 - 1.0
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "scale":                                    # LayerScale: O(1), otherwise the DConv branch is invisible
            v = 1.0 + 0.5 * u
        elif name.startswith("freq_emb"):
            v = 0.1 * u
        elif "query_decay.weight" in name:                     # reference init: small weights, bias -2 (wide window)
            v = u * 0.3 * np.sqrt(3.0 / shape[1])
        elif "query_decay.bias" in name:
            v = -1.0 + 0.5 * u
        elif leaf.startswith("weight") and len(shape) >= 2:    # conv / linear / LSTM matrices
            fan_in = shape[0] * 2 if "conv_tr" in name else int(np.prod(shape[1:]))
            if "conv_tr" in name and shape[2] == 4:            # k = 4, s = 2: two taps reach each output as well
                fan_in = shape[0] * 2
            glu_fed = ".rewrite." in name or (".dconv.layers" in name and len(shape) == 3 and shape[2] == 1)
            gain = 1.5 if glu_fed else 1.2 if ".conv." in name else 1.0
            v = u * gain * np.sqrt(3.0 / fan_in)
        elif leaf.startswith("weight"):                        # norm affine
            v = 1.0 + 0.25 * u
        else:                                                  # biases (conv, linear, LSTM, norm)
            v = 0.1 * u
        sd[name] = v.astype(np.float32).reshape(shape)
    return sd
