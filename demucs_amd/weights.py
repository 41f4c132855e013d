"""State-dict schema of the htdemucs architecture family and a deterministic,
machine-independent synthetic weight fill.

The schema mirrors what `HTDemucs.state_dict()` of the reference yields for the
released htdemucs hyper-parameters (reference: demucs/htdemucs.py:56-418,
demucs/hdemucs.py:69-122,256-302, demucs/demucs.py:133-149,
demucs/transformer.py:271-329,380-447,588-646; key list SURVEY.md App. B).

There is no network in the build/bench environment, so real checkpoints are never
available: every parity fixture and every benchmark uses the fill below, which is a
pure function of (tensor name, seed) so that the reference (in the golden-generation
container), the CPU oracle and the HIP path all see bit-identical weights.
LayerScale parameters are drawn O(1) instead of the reference's 1e-4 / 1e-3 init,
otherwise parity would be blind to the transformer and DConv branches.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from dataclasses import dataclass, field
from fractions import Fraction
from typing import Dict, List, Tuple

import numpy as np

__all__ = ["HTDemucsConfig", "htdemucs_schema", "synthetic_state_dict", "counter_uniform",
           "counter_normal", "periodic_uniform", "REFERENCE_DEFAULTS", "ENGINE_FIXED", "INERT_KEYWORDS", "check_reference_keyword"]


@dataclass
class HTDemucsConfig:
    """Hyper-parameters of the supported family: htdemucs / htdemucs_ft / htdemucs_6s.

    Only `sources` varies between released models (4 or 6 stems); everything else is
    the `955717e8` architecture (reference: demucs/grids/mmi.py:15-30,46-51,
    conf/config.yaml:195-271, docs/training.md:202).
    """
    sources: List[str] = field(default_factory=lambda: ["drums", "bass", "other", "vocals"])
    audio_channels: int = 2
    channels: int = 48
    growth: int = 2
    nfft: int = 4096
    depth: int = 4
    kernel_size: int = 8
    stride: int = 4
    context: int = 1
    freq_emb: float = 0.2
    emb_scale: float = 10.0
    dconv_mode: int = 3
    dconv_depth: int = 2
    dconv_comp: int = 8
    bottom_channels: int = 512
    t_layers: int = 5
    t_heads: int = 8
    t_hidden_scale: float = 4.0
    t_max_period: float = 10000.0
    t_weight_pos_embed: float = 1.0
    samplerate: int = 44100
    segment: Fraction = Fraction(39, 5)

    def validate(self) -> None:
        bad = []
        if self.audio_channels != 2: bad.append("audio_channels")
        if self.channels != 48: bad.append("channels")
        if self.growth != 2: bad.append("growth")
        if self.nfft != 4096: bad.append("nfft")
        if self.depth != 4: bad.append("depth")
        if self.kernel_size != 8 or self.stride != 4: bad.append("kernel_size/stride")
        if self.context != 1: bad.append("context")
        if self.dconv_mode != 3 or self.dconv_depth != 2 or self.dconv_comp != 8: bad.append("dconv_*")
        if self.bottom_channels != 512: bad.append("bottom_channels")
        if self.t_layers != 5 or self.t_heads != 8 or self.t_hidden_scale != 4.0: bad.append("t_*")
        if self.samplerate != 44100: bad.append("samplerate")
        if float(self.freq_emb) != 0.2 or float(self.emb_scale) != 10.0: bad.append("freq_emb/emb_scale")
        if float(self.t_max_period) != 10000.0 or float(self.t_weight_pos_embed) != 1.0: bad.append("t_max_period/t_weight_pos_embed")
        if len(self.sources) < 1 or len(self.sources) > 8: bad.append("sources")
        if bad:
            raise ValueError("unsupported HTDemucs hyper-parameters for the MI355X path: " + ", ".join(bad))

    @property
    def segment_length(self) -> int:
        return int(self.samplerate * self.segment)


# ---------------------------------------------------------------------------------------
# Every keyword of the reference constructor (demucs/htdemucs.py:56-133) that is NOT a field
# of HTDemucsConfig, sorted by what it means for the engine.
# ---------------------------------------------------------------------------------------
#: defaults of the reference constructor for the HTDemucsConfig fields (a checkpoint package that omits a
#: keyword means THIS value, not the released-model value the engine class defaults to)
REFERENCE_DEFAULTS = dict(audio_channels=2, channels=48, growth=2, nfft=4096, depth=4, freq_emb=0.2, emb_scale=10,
                          kernel_size=8, stride=4, context=1, dconv_mode=1, dconv_depth=2, dconv_comp=8,
                          bottom_channels=0, t_layers=5, t_hidden_scale=4.0, t_heads=8, t_max_period=10000.0,
                          t_weight_pos_embed=1.0, samplerate=44100, segment=10)

#: keywords whose value changes the forward: the engine hard-codes the value on the right (the released htdemucs
#: family, conf/config.yaml:195-271) and refuses anything else
ENGINE_FIXED = dict(channels_time=None, cac=True, rewrite=True, multi_freqs=None, time_stride=2, context_enc=0,
                    t_emb="sin", t_norm_in=True, t_norm_in_group=False, t_group_norm=False, t_norm_first=True,
                    t_norm_out=True, t_layer_scale=True, t_gelu=True, t_sin_random_shift=0,
                    t_sparse_self_attn=False, t_sparse_cross_attn=False, t_cross_first=False, use_train_segment=True)

#: keywords that cannot change an eval-mode forward of the architecture above (training knobs, initialisation, and
#: parameters of code paths that ENGINE_FIXED switches off): accepted with any value
INERT_KEYWORDS = frozenset((
    "t_dropout", "rescale", "dconv_init", "t_weight_decay", "t_lr", "emb_smooth",            # training / init only
    "wiener_iters", "end_iters", "wiener_residual",                                          # unused when cac=True
    "multi_freqs_depth",                                                                     # unused without multi_freqs
    "norm_groups",                                                                           # unused: norm_starts >= depth
    "t_max_positions", "t_cape_mean_normalize", "t_cape_augment", "t_cape_glob_loc_scale",   # t_emb != "sin" only
    "t_mask_type", "t_mask_random_seed", "t_sparse_attn_window", "t_global_window", "t_sparsity", "t_auto_sparsity"))


def check_reference_keyword(key: str, value, depth: int = 4) -> bool:
    """True if `key` is a reference keyword outside HTDemucsConfig that the engine honours by construction (drop it
    silently); raises ValueError when its value selects an architecture the engine does not implement; False for a
    name unknown to the reference as well."""
    if key in INERT_KEYWORDS:
        return True
    if key == "norm_starts":                      # GroupNorm inside HEnc/HDecLayer from this layer on: never, or refuse
        if value < depth:
            raise ValueError(f"unsupported HTDemucs argument norm_starts={value!r} (< depth {depth}: the engine has no "
                             "in-layer GroupNorm)")
        return True
    if key in ENGINE_FIXED:
        want = ENGINE_FIXED[key]
        same = (value == want) or (key == "multi_freqs" and not value)
        if not same:
            raise ValueError(f"unsupported HTDemucs argument {key}={value!r}: the MI355X engine implements {key}={want!r}")
        return True
    return False


def htdemucs_schema(cfg: HTDemucsConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """name -> shape for every tensor of the state dict (533 tensors for 4 sources)."""
    cfg.validate()
    S = len(cfg.sources)
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    chans = [cfg.channels * cfg.growth ** i for i in range(cfg.depth)]        # 48 96 192 384
    comp = cfg.dconv_comp

    def dconv(prefix: str, C: int) -> None:
        h = C // comp
        for d in range(cfg.dconv_depth):
            p = f"{prefix}.dconv.layers.{d}"
            out[f"{p}.0.weight"] = (h, C, 3)
            out[f"{p}.0.bias"] = (h,)
            out[f"{p}.1.weight"] = (h,)
            out[f"{p}.1.bias"] = (h,)
            out[f"{p}.3.weight"] = (2 * C, h, 1)
            out[f"{p}.3.bias"] = (2 * C,)
            out[f"{p}.4.weight"] = (2 * C,)
            out[f"{p}.4.bias"] = (2 * C,)
            out[f"{p}.6.scale"] = (C,)

    # module registration order of the reference: encoder, decoder, tencoder, tdecoder,
    # freq_emb, channel_*sampler*, crosstransformer (htdemucs.py:244-248,360,370-384).
    cin_z = 2 * cfg.audio_channels
    for i, C in enumerate(chans):
        p = f"encoder.{i}"
        out[f"{p}.conv.weight"] = (C, cin_z, 8, 1)
        out[f"{p}.conv.bias"] = (C,)
        out[f"{p}.rewrite.weight"] = (2 * C, C, 1, 1)
        out[f"{p}.rewrite.bias"] = (2 * C,)
        dconv(p, C)
        cin_z = C
    for j in range(cfg.depth):
        C = chans[cfg.depth - 1 - j]
        Cout = chans[cfg.depth - 2 - j] if j < cfg.depth - 1 else 2 * cfg.audio_channels * S
        p = f"decoder.{j}"
        out[f"{p}.conv_tr.weight"] = (C, Cout, 8, 1)
        out[f"{p}.conv_tr.bias"] = (Cout,)
        out[f"{p}.rewrite.weight"] = (2 * C, C, 3, 3)
        out[f"{p}.rewrite.bias"] = (2 * C,)
        dconv(p, C)
    cin = cfg.audio_channels
    for i, C in enumerate(chans):
        p = f"tencoder.{i}"
        out[f"{p}.conv.weight"] = (C, cin, 8)
        out[f"{p}.conv.bias"] = (C,)
        out[f"{p}.rewrite.weight"] = (2 * C, C, 1)
        out[f"{p}.rewrite.bias"] = (2 * C,)
        dconv(p, C)
        cin = C
    for j in range(cfg.depth):
        C = chans[cfg.depth - 1 - j]
        Cout = chans[cfg.depth - 2 - j] if j < cfg.depth - 1 else cfg.audio_channels * S
        p = f"tdecoder.{j}"
        out[f"{p}.conv_tr.weight"] = (C, Cout, 8)
        out[f"{p}.conv_tr.bias"] = (Cout,)
        out[f"{p}.rewrite.weight"] = (2 * C, C, 3)
        out[f"{p}.rewrite.bias"] = (2 * C,)
        dconv(p, C)
    out["freq_emb.embedding.weight"] = (cfg.nfft // 2 // cfg.stride, chans[0])
    Ct, Cb = chans[-1], cfg.bottom_channels
    for n in ("channel_upsampler", "channel_downsampler", "channel_upsampler_t", "channel_downsampler_t"):
        a, b = (Cb, Ct) if "up" in n else (Ct, Cb)
        out[f"{n}.weight"] = (a, b, 1)
        out[f"{n}.bias"] = (a,)
    D = Cb
    H = int(D * cfg.t_hidden_scale)
    for n in ("norm_in", "norm_in_t"):
        out[f"crosstransformer.{n}.weight"] = (D,)
        out[f"crosstransformer.{n}.bias"] = (D,)
    for branch in ("layers", "layers_t"):
        for k in range(cfg.t_layers):
            p = f"crosstransformer.{branch}.{k}"
            cross = k % 2 == 1
            attn = "cross_attn" if cross else "self_attn"
            if cross:
                # nn.Module registration order in CrossTransformerEncoderLayer (transformer.py:417-447)
                out[f"{p}.{attn}.in_proj_weight"] = (3 * D, D)
                out[f"{p}.{attn}.in_proj_bias"] = (3 * D,)
                out[f"{p}.{attn}.out_proj.weight"] = (D, D)
                out[f"{p}.{attn}.out_proj.bias"] = (D,)
                out[f"{p}.linear1.weight"] = (H, D); out[f"{p}.linear1.bias"] = (H,)
                out[f"{p}.linear2.weight"] = (D, H); out[f"{p}.linear2.bias"] = (D,)
                for n in ("norm1", "norm2", "norm3", "norm_out"):
                    out[f"{p}.{n}.weight"] = (D,); out[f"{p}.{n}.bias"] = (D,)
                out[f"{p}.gamma_1.scale"] = (D,); out[f"{p}.gamma_2.scale"] = (D,)
            else:
                out[f"{p}.{attn}.in_proj_weight"] = (3 * D, D)
                out[f"{p}.{attn}.in_proj_bias"] = (3 * D,)
                out[f"{p}.{attn}.out_proj.weight"] = (D, D)
                out[f"{p}.{attn}.out_proj.bias"] = (D,)
                out[f"{p}.linear1.weight"] = (H, D); out[f"{p}.linear1.bias"] = (H,)
                out[f"{p}.linear2.weight"] = (D, H); out[f"{p}.linear2.bias"] = (D,)
                for n in ("norm1", "norm2", "norm_out"):
                    out[f"{p}.{n}.weight"] = (D,); out[f"{p}.{n}.bias"] = (D,)
                out[f"{p}.gamma_1.scale"] = (D,); out[f"{p}.gamma_2.scale"] = (D,)
    return out


# ---------------------------------------------------------------------------------------
# counter-based PRNG: value i of stream `seed` = splitmix64(seed + i * golden) -> double.
# Pure integer arithmetic in numpy uint64, so it is identical on every machine / version.
# ---------------------------------------------------------------------------------------
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + _GOLD
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def counter_uniform(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """n doubles in [0, 1): element i depends only on (seed, offset + i)."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset, offset + n, dtype=np.uint64)
        key = _splitmix64(np.full(1, seed & 0xFFFFFFFFFFFFFFFF, dtype=np.uint64))[0]
        bits = _splitmix64(idx * _GOLD + key)
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def counter_normal(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """Approximately N(0,1): sum of 4 uniforms, centred and scaled (bounded, exactly portable;
    no libm transcendental whose last bit could differ between machines)."""
    acc = np.zeros(n, dtype=np.float64)
    for k in range(4):
        acc += counter_uniform(seed * 4 + k + 0x5151, n, offset)
    return (acc - 2.0) * np.sqrt(3.0)


def _name_seed(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) << 20) ^ (seed * 0x2545F4914F6CDD1D & 0xFFFFFFFFFFFFFFFF)


_GAINS = {"rewrite": 1.5, "dconv3": 1.5, "conv_tr": 1.0, "conv": 1.2, "default": 1.0}


def _gain(name: str) -> float:
    """Per-kind gains chosen so that activations stay O(1) through the U-Net (checked in
    tools/make_golden.py's per-stage rms table); GLU-fed convs get more (GLU halves energy)."""
    if ".rewrite." in name:
        return _GAINS["rewrite"]
    if ".dconv.layers" in name:
        return _GAINS["dconv3"] if ".3." in name else _GAINS["default"]
    if ".conv_tr." in name:
        return _GAINS["conv_tr"]
    if ".conv." in name:
        return _GAINS["conv"]
    return _GAINS["default"]


def periodic_uniform(seed: int, n: int, period=None) -> np.ndarray:
    """`counter_uniform(seed, n)`, or with `period` its first `period` values repeated: a tensor filled this way deflates
    ~200:1, which lets a full-size checkpoint package written by the reference live under tests/golden/ in ~1 MB."""
    if period is None or n <= period:
        return counter_uniform(seed, n)
    return np.resize(counter_uniform(seed, period), n)


def synthetic_state_dict(cfg: HTDemucsConfig, seed: int = 0, period=None) -> "OrderedDict[str, np.ndarray]":
    """Deterministic float32 weights keyed by tensor name (see module docstring)."""
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in htdemucs_schema(cfg).items():
        n = int(np.prod(shape))
        u = periodic_uniform(_name_seed(name, seed), n, period) * 2.0 - 1.0          # U(-1, 1)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "scale":                                  # LayerScale: O(1)  (fact 8)
            v = 1.0 + 0.5 * u
        elif name.startswith("freq_emb"):
            v = 0.1 * u                                      # x emb_scale(10) x 0.2 at use
        elif leaf in ("weight", "in_proj_weight") and len(shape) >= 2:
            if "conv_tr" in name:
                fan_in = shape[0] * 2                        # 2 taps reach each output
            else:
                fan_in = int(np.prod(shape[1:]))
            v = u * _gain(name) * np.sqrt(3.0 / fan_in)
        elif leaf == "weight":                               # norm affine
            v = 1.0 + 0.25 * u
        else:                                                # biases (conv, linear, norm)
            v = 0.1 * u
        sd[name] = v.astype(np.float32).reshape(shape)
    return sd
