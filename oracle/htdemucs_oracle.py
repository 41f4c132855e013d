"""CPU ORACLE (test infrastructure, NOT product code).

A from-scratch CPU restatement of the reference's segmented HTDemucs inference path, written
as plain functions over a state dict.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it; the product (`demucs_amd/`) never does and has
no CPU fallback.

Every function cites the reference lines it restates (paths relative to /root/reference).
The arithmetic primitives are torch CPU ops in float32 or float64 (`dtype=`); the STFT/iSTFT
are restated from their definitions (framing + rfft / irfft + overlap-add) rather than through
`torch.stft`, so they are an independent check of spec.py's conventions.

Pinned by: `tests/golden/*.npz`, produced by `tools/make_golden.py` from the imported
reference run in the build container (fp32 and fp64); see tests/test_oracle_golden.py.
The reference's own test-suite holds no numeric fixture for this path, so
those goldens are the only pin.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
NFFT = 4096
HOP = 1024

# bench.py's cpu_baseline leg sets this: the SAME library primitives the reference calls (th.stft / th.istft,
# spec.py:17-25,38-45; the fused attention inside nn.MultiheadAttention, transformer.py:418-419,506) instead of the
# from-definition restatements below, so that the timed CPU port runs at the reference's speed.  Parity tests keep
# it False: the restatements are the independent check.
FAST_PRIMITIVES = False


# --------------------------------------------------------------------------------------
# DSP (reference: demucs/spec.py:11-47, demucs/htdemucs.py:420-450, demucs/hdemucs.py:23-40)
# --------------------------------------------------------------------------------------
def hann_periodic(n: int, dtype) -> Tensor:
    """spec.py:19,41 build `th.hann_window(n_fft)` in float32 and only then cast `.to(x)`,
    so the window carries float32 rounding even in a float64 run; restated the same way."""
    k = torch.arange(n, dtype=torch.float32)
    return (0.5 - 0.5 * torch.cos(k * (2.0 * math.pi / n))).to(dtype)


def reflect_pad(x: Tensor, left: int, right: int) -> Tensor:
    """pad1d(mode='reflect') incl. the short-input fallback (hdemucs.py:23-40)."""
    length = x.shape[-1]
    max_pad = max(left, right)
    if length <= max_pad:
        extra = max_pad - length + 1
        extra_r = min(right, extra)
        extra_l = extra - extra_r
        x = F.pad(x, (extra_l, extra_r))
        left, right = left - extra_l, right - extra_r
    n = x.shape[-1]
    idx = torch.arange(-left, n + right)
    idx = torch.where(idx < 0, -idx, idx)
    idx = torch.where(idx >= n, 2 * (n - 1) - idx, idx)
    return x[..., idx]


def stft_cac(mix: Tensor) -> Tensor:
    """_spec + _magnitude(cac) (htdemucs.py:420-461; spec.py:11-27).
    mix (B,C,L) -> (B,2C,2048,le) real, channel order [c0.re, c0.im, c1.re, c1.im]."""
    B, C, L = mix.shape
    le = int(math.ceil(L / HOP))
    pad = HOP // 2 * 3
    x = reflect_pad(mix, pad, pad + le * HOP - L)              # htdemucs.py:433-435
    if FAST_PRIMITIVES:
        z = torch.stft(x.reshape(-1, x.shape[-1]), NFFT, HOP, window=hann_periodic(NFFT, mix.dtype), win_length=NFFT,
                       normalized=True, center=True, return_complex=True, pad_mode="reflect")
        z = z.view(B, C, NFFT // 2 + 1, -1)[..., :-1, 2:2 + le]
        return torch.view_as_real(z).permute(0, 1, 4, 2, 3).reshape(B, C * 2, NFFT // 2, le)
    x = reflect_pad(x, NFFT // 2, NFFT // 2)                    # th.stft(center=True, reflect)
    frames = x.unfold(-1, NFFT, HOP)                            # (B,C,le+4,4096)
    assert frames.shape[-2] == le + 4
    w = hann_periodic(NFFT, mix.dtype)
    z = torch.fft.rfft(frames * w, dim=-1) / math.sqrt(NFFT)    # normalized=True
    z = z[..., 2:2 + le, :NFFT // 2]                            # drop edge frames & Nyquist (:437-439)
    z = z.permute(0, 1, 3, 2)                                   # (B,C,F,T)
    m = torch.view_as_real(z).permute(0, 1, 4, 2, 3)            # (B,C,2,F,T)   (:457-458)
    return m.reshape(B, C * 2, NFFT // 2, le)


def istft_from_cac(x: Tensor, length: int) -> Tensor:
    """_mask(cac) + _ispec (htdemucs.py:442-450,463-471; spec.py:30-47).
    x (B,S,2C,2048,T) real -> (B,S,C,length)."""
    B, S, C2, Fr, T = x.shape
    z = x.reshape(B, S, C2 // 2, 2, Fr, T).permute(0, 1, 2, 4, 5, 3).contiguous()
    z = torch.view_as_complex(z)                                # (B,S,C,Fr,T)
    z = F.pad(z, (2, 2, 0, 1))                                  # +Nyquist bin, 2+2 frames (:444-445)
    pad = HOP // 2 * 3
    le = HOP * int(math.ceil(length / HOP)) + 2 * pad
    n_frames = T + 4
    w = hann_periodic(NFFT, x.dtype)
    if FAST_PRIMITIVES:
        y = torch.istft(z.reshape(-1, Fr + 1, n_frames), NFFT, HOP, window=w, win_length=NFFT, normalized=True, length=le,
                        center=True)
        return y.view(B, S, C2 // 2, le)[..., pad:pad + length]
    fr = torch.fft.irfft(z.transpose(-1, -2) * math.sqrt(NFFT), n=NFFT, dim=-1) * w   # (B,S,C,T+4,4096)
    total = NFFT + HOP * (n_frames - 1)
    y = torch.zeros(B, S, C2 // 2, total, dtype=x.dtype)
    env = torch.zeros(total, dtype=x.dtype)
    w2 = w * w
    for t in range(n_frames):                                   # overlap-add
        y[..., t * HOP:t * HOP + NFFT] += fr[..., t, :]
        env[t * HOP:t * HOP + NFFT] += w2
    y = y[..., NFFT // 2:NFFT // 2 + le] / env[NFFT // 2:NFFT // 2 + le]
    return y[..., pad:pad + length]


# --------------------------------------------------------------------------------------
# U-Net blocks (reference: demucs/hdemucs.py:69-157,256-335; demucs/demucs.py:86-154)
# --------------------------------------------------------------------------------------
def dconv(sd: Dict[str, Tensor], p: str, x: Tensor) -> Tensor:
    """DConv residual branch on (R,C,T) rows (demucs.py:133-154), depth 2, dilation 1,2."""
    for d in range(2):
        q = f"{p}.dconv.layers.{d}"
        dil = 2 ** d
        y = F.conv1d(x, sd[f"{q}.0.weight"], sd[f"{q}.0.bias"], dilation=dil, padding=dil)
        y = F.group_norm(y, 1, sd[f"{q}.1.weight"], sd[f"{q}.1.bias"], eps=1e-5)
        y = F.gelu(y)
        y = F.conv1d(y, sd[f"{q}.3.weight"], sd[f"{q}.3.bias"])
        y = F.group_norm(y, 1, sd[f"{q}.4.weight"], sd[f"{q}.4.bias"], eps=1e-5)
        y = F.glu(y, dim=1)
        y = sd[f"{q}.6.scale"][:, None] * y                    # LayerScale (transformer.py:251-255)
        x = x + y
    return x


def _rows(y: Tensor):
    B, C, Fr, T = y.shape
    return y.permute(0, 2, 1, 3).reshape(-1, C, T), (B, Fr, C, T)


def _unrows(y: Tensor, shp):
    B, Fr, C, T = shp
    return y.view(B, Fr, C, T).permute(0, 2, 1, 3)


def enc_freq(sd, i: int, x: Tensor) -> Tensor:
    """HEncLayer freq=True, norm=Identity (hdemucs.py:123-157)."""
    p = f"encoder.{i}"
    y = F.conv2d(x, sd[f"{p}.conv.weight"], sd[f"{p}.conv.bias"], stride=(4, 1), padding=(2, 0))
    y = F.gelu(y)
    r, shp = _rows(y)
    y = _unrows(dconv(sd, p, r), shp)
    z = F.conv2d(y, sd[f"{p}.rewrite.weight"], sd[f"{p}.rewrite.bias"])
    return F.glu(z, dim=1)


def enc_time(sd, i: int, x: Tensor) -> Tensor:
    """HEncLayer freq=False (hdemucs.py:132-157)."""
    p = f"tencoder.{i}"
    le = x.shape[-1]
    if le % 4:
        x = F.pad(x, (0, 4 - le % 4))
    y = F.gelu(F.conv1d(x, sd[f"{p}.conv.weight"], sd[f"{p}.conv.bias"], stride=4, padding=2))
    y = dconv(sd, p, y)
    z = F.conv1d(y, sd[f"{p}.rewrite.weight"], sd[f"{p}.rewrite.bias"])
    return F.glu(z, dim=1)


def dec_freq(sd, j: int, x: Tensor, skip: Tensor, last: bool) -> Tensor:
    """HDecLayer freq=True, context=1, context_freq=True (hdemucs.py:304-335)."""
    p = f"decoder.{j}"
    x = x + skip
    y = F.glu(F.conv2d(x, sd[f"{p}.rewrite.weight"], sd[f"{p}.rewrite.bias"], padding=1), dim=1)
    r, shp = _rows(y)
    y = _unrows(dconv(sd, p, r), shp)
    z = F.conv_transpose2d(y, sd[f"{p}.conv_tr.weight"], sd[f"{p}.conv_tr.bias"], stride=(4, 1))
    z = z[..., 2:-2, :]
    return z if last else F.gelu(z)


def dec_time(sd, j: int, x: Tensor, skip: Tensor, length: int, last: bool) -> Tensor:
    """HDecLayer freq=False (hdemucs.py:304-335)."""
    p = f"tdecoder.{j}"
    x = x + skip
    y = F.glu(F.conv1d(x, sd[f"{p}.rewrite.weight"], sd[f"{p}.rewrite.bias"], padding=1), dim=1)
    y = dconv(sd, p, y)
    z = F.conv_transpose1d(y, sd[f"{p}.conv_tr.weight"], sd[f"{p}.conv_tr.bias"], stride=4)
    z = z[..., 2:2 + length]
    return z if last else F.gelu(z)


# --------------------------------------------------------------------------------------
# Cross-domain transformer (reference: demucs/transformer.py:19-70,258-268,339-377,466-512,648-683)
# --------------------------------------------------------------------------------------
def sin_embedding_1d(length: int, dim: int, dtype, max_period: float = 10000.0) -> Tensor:
    """create_sin_embedding with shift=0 (transformer.py:19-34) -> (length, dim).
    The reference evaluates this table in float32 whatever the model dtype (integer aranges,
    true division -> float32); the table is part of the algorithm, so the oracle does the same
    and only then casts."""
    pos = torch.arange(length).view(-1, 1)
    half = dim // 2
    adim = torch.arange(half).view(1, -1)
    phase = pos / (max_period ** (adim / (half - 1)))
    return torch.cat([torch.cos(phase), torch.sin(phase)], dim=-1).to(dtype)


def sin_embedding_2d(d_model: int, height: int, width: int, dtype, max_period: float = 10000.0) -> Tensor:
    """create_2d_sin_embedding (transformer.py:37-70) -> (d_model, height, width); float32
    arithmetic like the reference, then cast."""
    pe = torch.zeros(d_model, height, width)
    half = d_model // 2
    div = torch.exp(torch.arange(0.0, half, 2) * -(math.log(max_period) / half))
    pw = torch.arange(0.0, width).unsqueeze(1)
    ph = torch.arange(0.0, height).unsqueeze(1)
    pe[0:half:2] = torch.sin(pw * div).t().unsqueeze(1).repeat(1, height, 1)
    pe[1:half:2] = torch.cos(pw * div).t().unsqueeze(1).repeat(1, height, 1)
    pe[half::2] = torch.sin(ph * div).t().unsqueeze(2).repeat(1, 1, width)
    pe[half + 1::2] = torch.cos(ph * div).t().unsqueeze(2).repeat(1, 1, width)
    return pe.to(dtype)


def mha(sd, p: str, q: Tensor, k: Tensor, heads: int = 8) -> Tensor:
    """nn.MultiheadAttention(batch_first, no mask, need_weights=False): packed in_proj rows
    [Wq;Wk;Wv], softmax(QK^T/sqrt(d))V, out_proj (called transformer.py:418-419,506)."""
    B, Tq, D = q.shape
    Tk = k.shape[1]
    W, b = sd[f"{p}.in_proj_weight"], sd[f"{p}.in_proj_bias"]
    Q = F.linear(q, W[:D], b[:D]).view(B, Tq, heads, D // heads).transpose(1, 2)
    K = F.linear(k, W[D:2 * D], b[D:2 * D]).view(B, Tk, heads, D // heads).transpose(1, 2)
    V = F.linear(k, W[2 * D:], b[2 * D:]).view(B, Tk, heads, D // heads).transpose(1, 2)
    if FAST_PRIMITIVES:
        o = F.scaled_dot_product_attention(Q, K, V).transpose(1, 2).reshape(B, Tq, D)
    else:
        att = torch.softmax(Q @ K.transpose(-1, -2) / math.sqrt(D // heads), dim=-1)
        o = (att @ V).transpose(1, 2).reshape(B, Tq, D)
    return F.linear(o, sd[f"{p}.out_proj.weight"], sd[f"{p}.out_proj.bias"])


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[f"{p}.weight"], sd[f"{p}.bias"], eps=1e-5)


def _gn_tokens(sd, p, x):
    """MyGroupNorm(1, D) over (T, D) jointly (transformer.py:258-268)."""
    return F.group_norm(x.transpose(1, 2), 1, sd[f"{p}.weight"], sd[f"{p}.bias"], eps=1e-5).transpose(1, 2)


def _ffn(sd, p, x):
    return F.linear(F.gelu(F.linear(x, sd[f"{p}.linear1.weight"], sd[f"{p}.linear1.bias"])),
                    sd[f"{p}.linear2.weight"], sd[f"{p}.linear2.bias"])


def self_layer(sd, p: str, x: Tensor) -> Tensor:
    """MyTransformerEncoderLayer, norm_first, norm_out, layer_scale (transformer.py:363-370)."""
    xn = _ln(sd, f"{p}.norm1", x)
    x = x + sd[f"{p}.gamma_1.scale"] * mha(sd, f"{p}.self_attn", xn, xn)
    x = x + sd[f"{p}.gamma_2.scale"] * _ffn(sd, p, _ln(sd, f"{p}.norm2", x))
    return _gn_tokens(sd, f"{p}.norm_out", x)


def cross_layer(sd, p: str, q: Tensor, k: Tensor) -> Tensor:
    """CrossTransformerEncoderLayer, norm_first (transformer.py:493-497)."""
    x = q + sd[f"{p}.gamma_1.scale"] * mha(sd, f"{p}.cross_attn", _ln(sd, f"{p}.norm1", q), _ln(sd, f"{p}.norm2", k))
    x = x + sd[f"{p}.gamma_2.scale"] * _ffn(sd, p, _ln(sd, f"{p}.norm3", x))
    return _gn_tokens(sd, f"{p}.norm_out", x)


def cross_transformer(sd, x: Tensor, xt: Tensor, taps=None):
    """CrossTransformerEncoder.forward (transformer.py:648-676). x (B,C,Fr,T1), xt (B,C,T2)."""
    B, C, Fr, T1 = x.shape
    pe2 = sin_embedding_2d(C, Fr, T1, x.dtype).permute(2, 1, 0).reshape(1, T1 * Fr, C)   # (t1 fr) c
    x = x.permute(0, 3, 2, 1).reshape(B, T1 * Fr, C)
    x = _ln(sd, "crosstransformer.norm_in", x) + pe2
    T2 = xt.shape[-1]
    xt = xt.permute(0, 2, 1)
    xt = _ln(sd, "crosstransformer.norm_in_t", xt) + sin_embedding_1d(T2, C, x.dtype)[None]
    for idx in range(5):
        pf, pt = f"crosstransformer.layers.{idx}", f"crosstransformer.layers_t.{idx}"
        if idx % 2 == 0:
            x, xt = self_layer(sd, pf, x), self_layer(sd, pt, xt)
        else:
            old_x = x
            x = cross_layer(sd, pf, x, xt)
            xt = cross_layer(sd, pt, xt, old_x)
        if taps is not None:
            taps[f"tr{idx}_f"] = x.transpose(1, 2)      # channel-first (B, C, tokens) view
            taps[f"tr{idx}_t"] = xt.transpose(1, 2)
    x = x.view(B, T1, Fr, C).permute(0, 3, 2, 1)
    xt = xt.permute(0, 2, 1)
    return x, xt


# --------------------------------------------------------------------------------------
# Whole model (reference: demucs/htdemucs.py:527-660)
# --------------------------------------------------------------------------------------
def htdemucs_forward(sd: Dict[str, Tensor], mix: Tensor, n_sources: int = 4, segment_length: int = 343980,
                     taps: Optional[dict] = None, mag_override: Optional[Tensor] = None) -> Tensor:
    """mix (B,2,n<=segment_length) -> (B,S,2,n).  `sd` tensors must have mix's dtype."""
    length = mix.shape[-1]
    length_pre_pad = None
    if length < segment_length:                                    # :534-537
        length_pre_pad = length
        mix = F.pad(mix, (0, segment_length - length))
    elif length > segment_length:
        raise ValueError(f"Given length {length} is longer than training length {segment_length}")
    B = mix.shape[0]
    S = n_sources
    # forward_core (htdemucs.py:662-690) takes `mag` as an input: mag_override stands for a caller-computed one
    mag = stft_cac(mix) if mag_override is None else mag_override.to(mix.dtype)
    if taps is not None: taps["stft"] = mag
    mean = mag.mean(dim=(1, 2, 3), keepdim=True)
    std = mag.std(dim=(1, 2, 3), keepdim=True)                     # unbiased (:546)
    x = (mag - mean) / (1e-5 + std)
    meant = mix.mean(dim=(1, 2), keepdim=True)
    stdt = mix.std(dim=(1, 2), keepdim=True)
    xt = (mix - meant) / (1e-5 + stdt)
    saved, saved_t, lengths_t = [], [], []
    for i in range(4):
        lengths_t.append(xt.shape[-1])
        xt = enc_time(sd, i, xt)
        saved_t.append(xt)
        x = enc_freq(sd, i, x)
        if taps is not None and i == 0:
            taps["enc0_preemb"] = x
        if i == 0:                                                 # :577-582, hdemucs.py:60-66
            emb = (sd["freq_emb.embedding.weight"] * 10.0).t()[None, :, :, None]
            x = x + 0.2 * emb
        saved.append(x)
        if taps is not None:
            taps[f"enc{i}"] = x; taps[f"tenc{i}"] = xt
    b, c, f, t = x.shape
    x = F.conv1d(x.reshape(b, c, f * t), sd["channel_upsampler.weight"], sd["channel_upsampler.bias"]).view(b, -1, f, t)
    xt = F.conv1d(xt, sd["channel_upsampler_t.weight"], sd["channel_upsampler_t.bias"])
    x, xt = cross_transformer(sd, x, xt, taps)
    x = F.conv1d(x.reshape(b, -1, f * t), sd["channel_downsampler.weight"], sd["channel_downsampler.bias"]).view(b, -1, f, t)
    xt = F.conv1d(xt, sd["channel_downsampler_t.weight"], sd["channel_downsampler_t.bias"])
    if taps is not None:
        taps["bott_f"] = x; taps["bott_t"] = xt
    for j in range(4):
        x = dec_freq(sd, j, x, saved.pop(), last=j == 3)
        xt = dec_time(sd, j, xt, saved_t.pop(), lengths_t.pop(), last=j == 3)
        if taps is not None:
            taps[f"dec{j}"] = x; taps[f"tdec{j}"] = xt
    Fq, T = x.shape[-2:]
    x = x.view(B, S, -1, Fq, T) * std[:, None] + mean[:, None]     # :625-626
    if taps is not None: taps["spec_out"] = x                      # forward_core outputs (htdemucs.py:752-759)
    if taps is not None: taps["time_out"] = xt.view(B, S, -1, segment_length) * stdt[:, None] + meant[:, None]
    x = istft_from_cac(x, segment_length)
    if taps is not None: taps["istft"] = x
    xt = xt.view(B, S, -1, segment_length) * stdt[:, None] + meant[:, None]
    out = xt + x
    if length_pre_pad:
        out = out[..., :length_pre_pad]
    return out


def to_torch_state(sd_np: dict, dtype=torch.float32) -> Dict[str, Tensor]:
    return {k: torch.from_numpy(v.copy()).to(dtype) for k, v in sd_np.items()}


class OracleModel:
    """Minimal model object with the attributes `apply_model` needs (apply.py:233-236,262-264)."""

    def __init__(self, sd_np: dict, sources: List[str], dtype=torch.float32, segment=None):
        from fractions import Fraction
        self.sd = to_torch_state(sd_np, dtype)
        self.dtype = dtype
        self.sources = list(sources)
        self.samplerate = 44100
        self.audio_channels = 2
        self.segment = Fraction(39, 5) if segment is None else segment

    @property
    def segment_length(self) -> int:
        return int(self.samplerate * self.segment)

    def __call__(self, mix: Tensor) -> Tensor:
        import random
        random.randrange(1)      # transformer.py:680 draws from Python's global RNG every forward
        with torch.no_grad():
            return htdemucs_forward(self.sd, mix.to(self.dtype), len(self.sources), self.segment_length)
