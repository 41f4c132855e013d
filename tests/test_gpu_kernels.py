"""Kernel-level parity on a real MI355X, through the C ABI, against float64 CPU math
(the oracle's definitions for STFT/iSTFT; torch CPU float64 conv / attention for the GEMM paths).

Tolerances (float32 kernels vs float64 truth, inputs O(1)): stated per test; they are a few
units of float32 rounding times sqrt(K) of the contraction length.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from demucs_amd import _lib
from demucs_amd.synth import synth_mix
from oracle import htdemucs_oracle as O

pytestmark = pytest.mark.gpu
SL = 343980


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return _lib.load()


def stream():
    return C.c_void_p(_lib.current_stream_ptr())


_HALF_MODE = [None]      # set by the `x6` fixture: "bf16" / "f16" while a reduced-precision case runs


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    t = torch.randn(*shape, generator=g, dtype=torch.float64) * scale
    if _HALF_MODE[0]:
        # values exactly representable in bf16 AND fp16: the operand rounding of the reduced-precision GEMMs is then the
        # identity, products are exact in float32, and the SAME tolerances as the float32 kernels apply to them
        t = t.float().bfloat16().double()
        t = torch.where(t.abs() < 2.0 ** -14, torch.zeros_like(t), t)
    return t


def chained_tol(base):
    """Tolerance of a test whose kernel feeds COMPUTED values (not `rnd` ones) back into a matrix product: the operand
    rounding (2^-9 relative for bf16, 2^-12 for fp16) is then visible."""
    return {None: base, "bf16": max(base, 3e-2), "f16": max(base, 4e-3)}[_HALF_MODE[0]]


# ------------------------------------------------------------------------------------------------
def test_stft_matches_oracle(lib):
    mix = torch.stack([torch.from_numpy(synth_mix(3, SL, "tones")), torch.from_numpy(synth_mix(4, SL, "noise"))])
    want = O.stft_cac(mix.double())
    out = torch.empty(2, 4, 2048, 336, device="cuda")
    mixd = mix.cuda()
    _lib.check(lib.mi_stft_cac(mixd.data_ptr(), 2, SL, out.data_ptr(), stream()), "mi_stft_cac")
    err = (out.cpu().double() - want).abs().max().item()
    assert err < 3e-6 * max(1.0, want.abs().max().item()), err     # 4096-point float32 FFT


def test_istft_matches_oracle_and_roundtrip(lib):
    x = rnd(1, 4, 4, 2048, 336, seed=1)
    want = O.istft_from_cac(x, SL)
    out = torch.empty(1, 4, 2, SL, device="cuda")
    xd = x.float().cuda()
    _lib.check(lib.mi_istft_cac(xd.data_ptr(), 1, 4, SL, out.data_ptr(), stream()), "mi_istft_cac")
    err = (out.cpu().double() - want).abs().max().item()
    assert err < 2e-5 * want.abs().max().item(), err
    # size-independent property: iSTFT(STFT(x)) == x on the interior (samples covered by four kept
    # frames), up to the Nyquist bin that `_spec` drops (htdemucs.py:437); tight against the oracle's
    # own round trip everywhere.
    mix = torch.from_numpy(synth_mix(9, SL, "tones"))[None]
    spec = torch.empty(1, 4, 2048, 336, device="cuda")
    mixd = mix.cuda()
    _lib.check(lib.mi_stft_cac(mixd.data_ptr(), 1, SL, spec.data_ptr(), stream()), "mi_stft_cac")
    back = torch.empty(1, 1, 2, SL, device="cuda")
    _lib.check(lib.mi_istft_cac(spec.data_ptr(), 1, 1, SL, back.data_ptr(), stream()), "mi_istft_cac")
    back = back.cpu()[0, 0]
    rt = (back - mix[0])[:, 1536:342528].abs().max().item()
    assert rt < 2e-3, rt
    oracle_rt = O.istft_from_cac(O.stft_cac(mix.double())[:, None], SL)[0, 0]
    assert (back.double() - oracle_rt).abs().max().item() < 5e-6


@pytest.mark.parametrize("L", [SL, 44100 * 3 + 17, 5000, 1023 * 7])
def test_fused_istft_is_bit_identical_to_the_separate_kernels(lib, L):
    """The default inverse STFT keeps the overlap-add of a run of frames in registers (istft_fused_kernel); the two separate
    kernels (frames to memory, then a gather) add the same four frames per sample in the same ascending order: equal bits, at
    the segment length, at lengths that are no multiple of the hop and at a short input."""
    T = -(-L // 1024)
    x = rnd(2, 3, 4, 2048, T, seed=5).float().cuda()
    outs = []
    for fused in (1, 0):
        old = lib.mi_set_istft_fused(fused)
        try:
            out = torch.full((2, 3, 2, L), float("nan"), device="cuda")
            _lib.check(lib.mi_istft_cac(x.data_ptr(), 2, 3, L, out.data_ptr(), stream()), "mi_istft_cac")
        finally:
            lib.mi_set_istft_fused(old)
        outs.append(out.cpu())
    assert bool(torch.isfinite(outs[0]).all()) and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("L", [SL, 44100 * 3 + 17, 5000, 44 * 44100 + 3])
def test_strip_transposes_are_bit_identical_to_the_tile_kernels(lib, L):
    """Both layout changes around the transforms (cac_transpose / spec_transpose) move 32 bins x all frames per workgroup by
    default; the 32 x 32 tile kernels compute the same values with the same arithmetic: equal bits for the normalised CaC
    spectrogram and for the waveform, at the segment length (one strip, 16-byte accesses), at frame counts that are no multiple of
    four (4-byte accesses), at a short input and at a 44-second chunk (several strips per row)."""
    T = -(-L // 1024)
    mix = torch.from_numpy(synth_mix(6, L, "noise"))[None].repeat(2, 1, 1).contiguous().cuda()
    x = rnd(2, 2, 4, 2048, T, seed=8).float().cuda()
    got = []
    for tiles in (0, 1):                        # default kernels, the round-3 kernels
        old = lib.mi_set_transpose_tiles(tiles)
        try:
            spec = torch.full((2, 4, 2048, T), float("nan"), device="cuda")
            _lib.check(lib.mi_stft_cac(mix.data_ptr(), 2, L, spec.data_ptr(), stream()), "mi_stft_cac")
            wav = torch.full((2, 2, 2, L), float("nan"), device="cuda")
            _lib.check(lib.mi_istft_cac(x.data_ptr(), 2, 2, L, wav.data_ptr(), stream()), "mi_istft_cac")
        finally:
            lib.mi_set_transpose_tiles(old)
        got.append((spec.cpu(), wav.cpu()))
    assert bool(torch.isfinite(got[0][0]).all()) and bool(torch.isfinite(got[0][1]).all())
    for other in got[1:]:
        assert torch.equal(got[0][0], other[0]) and torch.equal(got[0][1], other[1])


# ------------------------------------------------------------------------------------------------
from gpu_helpers import (EPI_BIAS_STATS, EPI_CONVTR, EPI_GLU, EPI_GN_GLU, EPI_LINEAR, EPI_STATS_ONLY, FLAG_EMB,  # noqa: E402
                         FLAG_GELU, FLAG_RES, FLAG_SCALE, FLAG_TR_FREQ, SLOTS, conv_call, ktab, maxerr, pack_vec, pack_w)


@pytest.fixture(params=[False, True, "bf16", "f16"], ids=["fp32mfma", "bf16x6", "bf16", "f16"])
def x6(request):
    """Every GEMM main loop: native fp32 MFMA, the exact 3-term bf16 split with 6 products (gemm_x6.hip), and the
    reduced-precision compute modes with bf16 / fp16 operands (gemm_half.hip)."""
    _HALF_MODE[0] = request.param if isinstance(request.param, str) else None
    yield request.param
    _HALF_MODE[0] = None


def test_conv_freq_strided_gelu(lib, x6):
    """HEncLayer conv on the frequency axis: Conv2d k=(8,1) s=(4,1) p=(2,0) + GELU (hdemucs.py:110,136,144)."""
    B, Cin, Cout, Fr, T = 2, 12, 96, 64, 48
    x, W, b = rnd(B, Cin, Fr, T, seed=2), rnd(Cout, Cin, 8, 1, seed=3, scale=0.1), rnd(Cout, seed=4)
    want = F.gelu(F.conv2d(x, W, b, stride=(4, 1), padding=(2, 0)))
    wt, bias, M, Mpad, K, Kpad, tile = pack_w(W.reshape(Cout, -1), b)
    kt = ktab(Cin, 8, 1, 1, 1, 2, 0, Fr * T, T, Kpad)
    y = torch.empty(B, Cout, Fr // 4, T, device="cuda")
    P = Fr // 4 * T
    conv_call(x6=x6, wt=wt, M=M, Mpad=Mpad, K=K, Kpad=Kpad, ktab=kt, x=x.float().cuda(), x_bstride=Cin * Fr * T, B=B, D1=Fr, D2=T,
              O1=Fr // 4, O2=T, S1=4, S2=1, row_mode=1, epi=EPI_LINEAR, flags=FLAG_GELU, bias=bias, y=y, y_bstride=Cout * P,
              y_cstride=P, tile_m=tile)
    assert maxerr(y, want) < 2e-5


def test_conv_time_strided_ragged(lib, x6):
    """HEncLayer conv on the time axis with a length that is not a multiple of the stride
    (right zero padding, hdemucs.py:132-136): L=5375 -> 1344."""
    B, Cin, Cout, L = 2, 6, 48, 5375
    x, W, b = rnd(B, Cin, L, seed=5), rnd(Cout, Cin, 8, seed=6, scale=0.2), rnd(Cout, seed=7)
    want = F.conv1d(F.pad(x, (0, 1)), W, b, stride=4, padding=2)
    Lo = want.shape[-1]
    wt, bias, M, Mpad, K, Kpad, tile = pack_w(W.reshape(Cout, -1), b)
    kt = ktab(Cin, 1, 8, 1, 1, 0, 2, L, L, Kpad)
    y = torch.empty(B, Cout, Lo, device="cuda")
    conv_call(x6=x6, wt=wt, M=M, Mpad=Mpad, K=K, Kpad=Kpad, ktab=kt, x=x.float().cuda(), x_bstride=Cin * L, B=B, D1=1, D2=L, O1=1, O2=Lo,
              S1=1, S2=4, epi=EPI_LINEAR, bias=bias, y=y, y_bstride=Cout * Lo, y_cstride=Lo, tile_m=tile)
    assert maxerr(y, want) < 2e-5


def test_conv_3x3_glu_and_emb(lib, x6):
    """HDecLayer rewrite Conv2d 3x3 + GLU (hdemucs.py:294,313), plus the additive per-(c, fr) table."""
    B, C, Fr, T = 1, 24, 16, 80
    x, W, b = rnd(B, C, Fr, T, seed=8), rnd(2 * C, C, 3, 3, seed=9, scale=0.1), rnd(2 * C, seed=10)
    emb = rnd(C, Fr, seed=11)
    want = F.glu(F.conv2d(x, W, b, padding=1), dim=1) + emb[None, :, :, None]
    wt, bias, M, Mpad, K, Kpad, tile = pack_w(W.reshape(2 * C, -1), b, glu=True)
    kt = ktab(C, 3, 3, 1, 1, 1, 1, Fr * T, T, Kpad)
    y = torch.empty(B, C, Fr, T, device="cuda")
    P = Fr * T
    conv_call(x6=x6, wt=wt, M=M, Mpad=Mpad, K=K, Kpad=Kpad, ktab=kt, x=x.float().cuda(), x_bstride=C * P, B=B, D1=Fr, D2=T, O1=Fr, O2=T,
              S1=1, S2=1, row_mode=1, epi=EPI_GLU, flags=FLAG_EMB, emb=emb.float().cuda().contiguous(), bias=bias, y=y,
              y_bstride=C * P, y_cstride=P, tile_m=tile)
    assert maxerr(y, want) < 2e-5


@pytest.mark.parametrize("Cc,Fr,T,pitch", [(48, 12, 336, 336), (64, 5, 140, 140), (96, 3, 61, 64), (192, 2, 37, 40)])
def test_float32_3x3_glu_on_the_dma_tap_route(lib, Cc, Fr, T, pitch):
    """The float32 decoders' 3 x 3 + GLU rewrite conv when its geometry is stated in the descriptor (ntaps / tap_k2 / tap_pad*):
    the main loop moves shifted runs of the input rows global -> LDS by DMA and writes the conv's zero padding at the row ends in
    LDS (gemm_conv.hip conv_gemm_dmatap_kernel; 96- and 128-row tiles), instead of walking the gather table.  Against F.conv2d in
    float64 and, bit for bit, against the table-driven route (same products, same k order); with a row pitch wider than the valid
    width the padding columns hold NaN: they must never reach a valid output."""
    B = 2
    x, W, b = rnd(B, Cc, Fr, T, seed=31), rnd(2 * Cc, Cc, 3, 3, seed=32, scale=0.05), rnd(2 * Cc, seed=33)
    want = F.glu(F.conv2d(x, W, b, padding=1), dim=1)
    wt, bias, M, Mpad, K, Kpad, tile = pack_w(W.reshape(2 * Cc, -1), b, glu=True)
    assert tile in (96, 128)
    xp = torch.full((B, Cc, Fr, pitch), float("nan"))
    xp[..., :T] = x.float()
    P = Fr * pitch
    # 32 floats of slack on both sides, as the engine's decoder-input buffers carry (the shifted runs start one sample early)
    buf = torch.full((B * Cc * P + 64,), float("nan"), device="cuda")
    buf[32:32 + B * Cc * P] = xp.reshape(-1).cuda()
    xin = buf[32:32 + B * Cc * P]
    kt = ktab(Cc, 3, 3, 1, 1, 1, 1, P, pitch, Kpad)
    outs = []
    for geo in (dict(ntaps=9, tap_k2=3, tap_pad1=1, tap_pad2=1), {}):
        y = torch.full((B, Cc, Fr, pitch), float("nan"), device="cuda")
        d = _lib.MiConvDesc()
        kw = dict(wt=wt, M=M, Mpad=Mpad, K=K, Kpad=Kpad, ktab=kt, x=xin, x_bstride=Cc * P, B=B, D1=Fr, D2=T, O1=Fr, O2=pitch, S1=1, S2=1,
                  row_mode=1, epi=EPI_GLU, bias=bias, y=y, y_bstride=Cc * P, y_cstride=P, tile_m=tile, o2_valid=T if pitch != T else 0,
                  x_ld=pitch if pitch != T else 0, **geo)
        for name, _ in _lib.MiConvDesc._fields_:
            v = kw.get(name, 0)
            setattr(d, name, v.data_ptr() if isinstance(v, torch.Tensor) else v)
        _lib.check(lib.mi_conv_forward(C.byref(d), stream()), "mi_conv_forward")
        torch.cuda.synchronize()
        assert lib.mi_debug_last_conv_route() == (2 if geo else 0)       # shifted-run DMA taps / table-driven gather
        outs.append(y[..., :T].cpu())
    assert bool(torch.isfinite(outs[0]).all())
    assert maxerr(outs[0], want) < 2e-5
    assert torch.equal(outs[0], outs[1])


def test_float32_k3_glu_time_branch_on_the_dma_tap_route(lib):
    """The time-branch rewrite conv (k = 3, padding 1) on rows whose pitch exceeds their valid length (21 499 -> 21 500)."""
    B, Cc, L, pitch = 2, 96, 21499, 21500
    x, W, b = rnd(B, Cc, L, seed=34), rnd(2 * Cc, Cc, 3, seed=35, scale=0.05), rnd(2 * Cc, seed=36)
    want = F.glu(F.conv1d(x, W, b, padding=1), dim=1)
    wt, bias, M, Mpad, K, Kpad, tile = pack_w(W.reshape(2 * Cc, -1), b, glu=True)
    buf = torch.full((B * Cc * pitch + 64,), float("nan"), device="cuda")
    xp = torch.full((B, Cc, pitch), float("nan"))
    xp[..., :L] = x.float()
    buf[32:32 + B * Cc * pitch] = xp.reshape(-1).cuda()
    xin = buf[32:32 + B * Cc * pitch]
    kt = ktab(Cc, 1, 3, 1, 1, 0, 1, pitch, pitch, Kpad)
    y = torch.full((B, Cc, pitch), float("nan"), device="cuda")
    d = _lib.MiConvDesc()
    kw = dict(wt=wt, M=M, Mpad=Mpad, K=K, Kpad=Kpad, ktab=kt, x=xin, x_bstride=Cc * pitch, B=B, D1=1, D2=L, O1=1, O2=pitch, S1=1, S2=1,
              epi=EPI_GLU, bias=bias, y=y, y_bstride=Cc * pitch, y_cstride=pitch, tile_m=tile, o2_valid=L, x_ld=pitch, ntaps=3, tap_k2=3,
              tap_pad1=0, tap_pad2=1)
    for name, _ in _lib.MiConvDesc._fields_:
        v = kw.get(name, 0)
        setattr(d, name, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    _lib.check(lib.mi_conv_forward(C.byref(d), stream()), "mi_conv_forward")
    torch.cuda.synchronize()
    assert lib.mi_debug_last_conv_route() == 2
    got = y[..., :L].cpu()
    assert bool(torch.isfinite(got).all()) and maxerr(got, want) < 2e-5


def _conv_desc_call(lib, **kw):
    d = _lib.MiConvDesc()
    for name, _ in _lib.MiConvDesc._fields_:
        v = kw.get(name, 0)
        setattr(d, name, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    _lib.check(lib.mi_conv_forward(C.byref(d), stream()), "mi_conv_forward")
    torch.cuda.synchronize()
    return lib.mi_debug_last_conv_route()        # 0 table-driven gather, 2 DMA shifted-run taps, 3 DMA row taps (include/demucs_amd.h)


@pytest.mark.parametrize("Cout,Fr,T,pitch", [(96, 64, 48, 48), (128, 16, 336, 336), (48, 32, 61, 64), (192, 8, 37, 40)])
def test_float32_strided_frequency_conv_on_the_dma_row_route(lib, Cout, Fr, T, pitch):
    """Round 4: float32 layers whose taps move along rows only (`dma_rows`) run an LDS-DMA main loop (gemm_conv.hip
    conv_gemm_dmarow_kernel; 128-, 96- and 64-row tiles) -- here the encoder's Conv2d k=(8,1) s=(4,1) p=(2,0) + GELU
    (hdemucs.py:110,136,144).  Against F.conv2d in float64 and, bit for bit, against the table-driven gather (same products, same
    k order); with a row pitch wider than the valid width the padding columns hold NaN and must never reach a valid output."""
    B, Cin = 2, 24
    x, W, b = rnd(B, Cin, Fr, T, seed=41), rnd(Cout, Cin, 8, 1, seed=42, scale=0.1), rnd(Cout, seed=43)
    want = F.gelu(F.conv2d(x, W, b, stride=(4, 1), padding=(2, 0)))
    wt, bias, M, Mpad, K, Kpad, tile = pack_w(W.reshape(Cout, -1), b)
    assert tile in (64, 96, 128)
    xp = torch.full((B, Cin, Fr, pitch), float("nan"))
    xp[..., :T] = x.float()
    xin = xp.cuda().contiguous()
    kt = ktab(Cin, 8, 1, 1, 1, 2, 0, Fr * pitch, pitch, Kpad)
    P = Fr // 4 * pitch
    outs = []
    for rows in (1, 0):
        y = torch.full((B, Cout, Fr // 4, pitch), float("nan"), device="cuda")
        route = _conv_desc_call(lib, wt=wt, M=M, Mpad=Mpad, K=K, Kpad=Kpad, ktab=kt, x=xin, x_bstride=Cin * Fr * pitch, B=B, D1=Fr, D2=T,
                                O1=Fr // 4, O2=pitch, S1=4, S2=1, row_mode=1, epi=EPI_LINEAR, flags=FLAG_GELU, bias=bias, y=y, y_bstride=Cout * P,
                                y_cstride=P, tile_m=tile, o2_valid=T if pitch != T else 0, x_ld=pitch if pitch != T else 0, dma_rows=rows)
        assert route == (3 if rows else 0)          # the two outputs really come from two main loops
        outs.append(y[..., :T].cpu())
    assert bool(torch.isfinite(outs[0]).all())
    assert maxerr(outs[0], want) < 2e-5
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("Co,skip_on", [(32, True), (24, True), (16, False), (48, False)])
def test_float32_frequency_transposed_conv_on_the_dma_row_route(lib, Co, skip_on):
    """ConvTranspose2d k=(8,1) s=(4,1) + crop (+ GELU + skip) as the 4-phase GEMM with its two taps (rows q, q - 1) by LDS-DMA:
    M = 4 Co = 128 / 96 / 64 / 192 rows.  Float64 reference and bit identity with the table-driven gather."""
    B, Cc, Fr, T = 2, 40, 9, 132
    x, W, b = rnd(B, Cc, Fr, T, seed=44), rnd(Cc, Co, 8, 1, seed=45, scale=0.2), rnd(Co, seed=46)
    skip = rnd(B, Co, 4 * Fr, T, seed=47)
    want = F.conv_transpose2d(x, W, b, stride=(4, 1))[..., 2:-2, :]
    if skip_on:
        want = F.gelu(want) + skip
    Wr = W.reshape(Cc, Co, 8)
    W2 = torch.zeros(4 * Co, 2 * Cc, dtype=torch.float64)       # row 4co+r, col 2ci+j  <- W[ci][co][r+4j]
    for r in range(4):
        for j in range(2):
            W2[r::4, j::2] = Wr[:, :, r + 4 * j].t()
    wt, bias, M, Mpad, K, Kpad, tile = pack_w(W2, b.repeat_interleave(4))
    assert tile in (64, 96, 128)
    kt = ktab(Cc, 2, 1, -1, 1, 0, 0, Fr * T, T, Kpad)
    xin, res = x.float().cuda().contiguous(), skip.float().cuda().contiguous()
    outs = []
    for rows in (1, 0):
        y = torch.full((B, Co, 4 * Fr, T), float("nan"), device="cuda")
        route = _conv_desc_call(lib, wt=wt, M=M, Mpad=Mpad, K=K, Kpad=Kpad, ktab=kt, x=xin, x_bstride=Cc * Fr * T, B=B, D1=Fr, D2=T, O1=Fr + 1,
                                O2=T, S1=1, S2=1, row_mode=1, epi=EPI_CONVTR, flags=FLAG_TR_FREQ | (FLAG_GELU | FLAG_RES if skip_on else 0),
                                res=res if skip_on else 0, bias=bias, y=y, y_bstride=Co * 4 * Fr * T, y_cstride=4 * Fr * T, out_len=4 * Fr,
                                tile_m=tile, dma_rows=rows)
        assert route == (3 if rows else 0)
        outs.append(y.cpu())
    assert maxerr(outs[0], want) < 2e-5
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("Cc,gn", [(64, False), (128, False), (192, True)])
def test_float32_pointwise_glu_layers_on_the_dma_row_route(lib, Cc, gn):
    """1x1 conv + GLU (the encoders' rewrite, hdemucs.py:117,149-150; with the frequency embedding's additive table) and the DConv
    tail 1x1 -> GroupNorm -> GLU -> LayerScale -> + x (demucs.py:141-143,151-154) as plain layers: 96- / 128-row tiles of the
    LDS-DMA row loop.  Float64 reference; bit identity with the table-driven gather (the same layer declared non-plain)."""
    B, Fr, T = 2, 6, 140
    P = Fr * T
    K = 32 if gn else Cc
    x, W, b = rnd(B, K, Fr, T, seed=48), rnd(2 * Cc, K, 1, 1, seed=49, scale=0.2), rnd(2 * Cc, seed=50)
    z = F.conv2d(x, W, b)
    wt, bias, M, Mpad, K_, Kpad, tile = pack_w(W.reshape(2 * Cc, K), b, glu=True)
    assert tile == 128 and K_ == Kpad
    kt = ktab(K, 1, 1, 1, 1, 0, 0, P, T, Kpad)
    xin = x.float().cuda().contiguous()
    extra = {}
    if gn:
        g2w, g2b, ls, res = 1 + 0.2 * rnd(2 * Cc, seed=51), 0.1 * rnd(2 * Cc, seed=52), 1 + 0.3 * rnd(Cc, seed=53), rnd(B, Cc, Fr, T, seed=54)
        rows = z.permute(0, 2, 1, 3).reshape(B * Fr, 2 * Cc, T)
        mean, var = rows.mean(dim=(1, 2)), rows.var(dim=(1, 2), unbiased=False)
        st2 = torch.stack([mean, 1.0 / torch.sqrt(var + 1e-5)], 1).float().cuda().contiguous()
        zn = F.group_norm(rows, 1, g2w, g2b, eps=1e-5).view(B, Fr, 2 * Cc, T).permute(0, 2, 1, 3)
        want = res + ls[None, :, None, None] * F.glu(zn, dim=1)
        extra = dict(epi=EPI_GN_GLU, gn_stats=st2, gn_w=pack_vec(g2w, Mpad, glu=True), gn_b=pack_vec(g2b, Mpad, glu=True),
                     scale=ls.float().cuda(), res=res.float().cuda().contiguous())
    else:
        emb = rnd(Cc, Fr, seed=55)
        want = F.glu(z, dim=1) + emb[None, :, :, None]
        extra = dict(epi=EPI_GLU, flags=FLAG_EMB, emb=emb.float().cuda().contiguous())
    outs = []
    for plain in (1, 0):
        y = torch.full((B, Cc, Fr, T), float("nan"), device="cuda")
        route = _conv_desc_call(lib, wt=wt, M=M, Mpad=Mpad, K=K, Kpad=Kpad, ktab=kt, x=xin, x_bstride=K * P, B=B, D1=Fr, D2=T, O1=Fr, O2=T, S1=1,
                                S2=1, row_mode=1, bias=bias, y=y, y_bstride=Cc * P, y_cstride=P, tile_m=tile, plain=plain, **extra)
        assert route == (3 if plain else 0)
        outs.append(y.cpu())
    assert maxerr(outs[0], want) < (1e-4 if gn else 2e-5)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("h,dil,Fr,T,pitch", [(24, 1, 6, 336, 336), (24, 2, 5, 61, 64), (48, 2, 3, 140, 140), (48, 1, 1, 21499, 21500)])
def test_float32_dilated_k3_conv_with_row_statistics_on_the_dma_tap_route(lib, h, dil, Fr, T, pitch):
    """The DConv blocks' first conv (demucs.py:138: Conv1d C -> C / 8, k = 3, dilation = padding = 1 / 2) with the row-statistics
    epilogue, on the 32- / 64-row tiles of the DMA tap loop (conv_gemm_dmatap_kernel<.., BIAS_STATS, 3, DIL>): shifted runs, up to
    two samples from outside the row at either end re-zeroed in LDS.  Frequency-branch rows (b, fr) and a time-branch row with a
    pitch wider than its valid length (NaN in the padding).  Float64 reference, bit identity of the stored tensor with the
    table-driven gather, statistics to 1e-9 relative (float64 atomics: summation order differs between launches)."""
    B, Cc = 2, 8 * h
    x, W, b = rnd(B, Cc, Fr, T, seed=61), rnd(h, Cc, 1, 3, seed=62, scale=0.1), rnd(h, seed=63)
    want = F.conv2d(x, W, b, padding=(0, dil), dilation=(1, dil))
    wt, bias, M, Mpad, K, Kpad, tile = pack_w(W.reshape(h, -1), b)
    assert tile in (32, 64)
    P = Fr * pitch
    xp = torch.full((B, Cc, Fr, pitch), float("nan"))
    xp[..., :T] = x.float()
    buf = torch.full((B * Cc * P + 64,), float("nan"), device="cuda")       # 32 floats of slack on both sides, as the engine's buffers carry
    buf[32:32 + B * Cc * P] = xp.reshape(-1).cuda()
    xin = buf[32:32 + B * Cc * P]
    kt = ktab(Cc, 1, 3, 1, dil, 0, dil, P, pitch, Kpad)
    row_mode, nrows = (1, B * Fr) if Fr > 1 else (0, B)
    outs, sts = [], []
    for geo in (dict(ntaps=3, tap_k2=3, tap_pad1=0, tap_pad2=dil, tap_dil2=dil), {}):
        y = torch.full((B, h, Fr, pitch), float("nan"), device="cuda")
        stats = torch.zeros(nrows, SLOTS, 2, dtype=torch.float64, device="cuda")
        route = _conv_desc_call(lib, wt=wt, M=M, Mpad=Mpad, K=K, Kpad=Kpad, ktab=kt, x=xin, x_bstride=Cc * P, B=B, D1=Fr, D2=T, O1=Fr, O2=pitch,
                                S1=1, S2=1, row_mode=row_mode, epi=EPI_BIAS_STATS, bias=bias, y=y, y_bstride=h * P, y_cstride=P, stats=stats,
                                tile_m=tile, o2_valid=T if pitch != T else 0, x_ld=pitch if pitch != T else 0, **geo)
        assert route == (2 if geo else 0)
        outs.append(y[..., :T].cpu())
        sts.append(stats.sum(1).cpu())
    assert bool(torch.isfinite(outs[0]).all())
    assert maxerr(outs[0], want) < 6e-6 * float(want.abs().max())          # K = 576 / 1 152 float32 products, outputs up to ~8
    assert torch.equal(outs[0], outs[1])
    rows = want.permute(0, 2, 1, 3).reshape(nrows, -1) if row_mode else want.reshape(B, -1)
    ref = torch.stack([rows.sum(1), (rows ** 2).sum(1)], 1)
    for st in sts:
        assert ((st - ref).abs() / ref.abs().clamp_min(1.0)).max().item() < 2e-5        # float32 values, float32 per-thread partial sums
    assert ((sts[0] - sts[1]).abs() / sts[1].abs().clamp_min(1.0)).max().item() < 1e-9


def test_linear_scale_residual_big_k(lib, x6):
    """nn.Linear on channel-first tokens with LayerScale + residual epilogue (transformer.py:364-367):
    M=512, K=2048 exercises the 128-row tile and a long contraction."""
    B, M, K, Tn = 2, 512, 2048, 300
    x, W, b = rnd(B, K, Tn, seed=12), rnd(M, K, seed=13, scale=0.03), rnd(M, seed=14)
    g, r = rnd(M, seed=15), rnd(B, M, Tn, seed=16)
    want = r + g[None, :, None] * (torch.einsum("mk,bkt->bmt", W, x) + b[None, :, None])
    wt, bias, M_, Mpad, K_, Kpad, tile = pack_w(W, b)
    kt = ktab(K, 1, 1, 1, 1, 0, 0, Tn, Tn, Kpad)
    y = torch.empty(B, M, Tn, device="cuda")
    conv_call(x6=x6, wt=wt, M=M, Mpad=Mpad, K=K, Kpad=Kpad, ktab=kt, x=x.float().cuda(), x_bstride=K * Tn, B=B, D1=1, D2=Tn, O1=1, O2=Tn,
              S1=1, S2=1, plain=1, epi=EPI_LINEAR, flags=FLAG_SCALE | FLAG_RES, scale=pack_vec(g, Mpad), res=r.float().cuda(), bias=bias,
              y=y, y_bstride=M * Tn, y_cstride=Tn, tile_m=tile)
    assert maxerr(y, want) < 6e-5
    # the same layer with MI_FLAG_STATS (256): sum / sum of squares of the stored result per item, float64 atomics into
    # stats[item][32 slots][2] -- what the GroupNorm after a transformer layer reads (no pass of its own over the tensor)
    if x6 is not True:                    # the split-bf16 main loop is not instantiated with it (launch_conv falls back)
        stats = torch.zeros(B, 32, 2, dtype=torch.float64, device="cuda")
        y2 = torch.empty(B, M, Tn, device="cuda")
        conv_call(x6=x6, wt=wt, M=M, Mpad=Mpad, K=K, Kpad=Kpad, ktab=kt, x=x.float().cuda(), x_bstride=K * Tn, B=B, D1=1, D2=Tn, O1=1,
                  O2=Tn, S1=1, S2=1, plain=1, epi=EPI_LINEAR, flags=FLAG_SCALE | FLAG_RES | 256, scale=pack_vec(g, Mpad), res=r.float().cuda(),
                  bias=bias, y=y2, y_bstride=M * Tn, y_cstride=Tn, tile_m=tile, stats=stats)
        assert torch.equal(y2, y)
        got = stats.sum(1).cpu()
        ref = torch.stack([y.double().sum((1, 2)).cpu(), (y.double() ** 2).sum((1, 2)).cpu()], 1)
        assert ((got - ref).abs() / ref.abs().clamp_min(1.0)).max().item() < 2e-6, (got, ref)


@pytest.mark.parametrize("freq", [True, False])
def test_conv_transpose_4phase(lib, freq, x6):
    """ConvTranspose k=8 s=4 + crop + GELU + skip add (hdemucs.py:287,326-334) as a 4-phase GEMM."""
    if freq:
        B, C, Co, Fr, T = 2, 24, 12, 8, 40
        x, W, b = rnd(B, C, Fr, T, seed=17), rnd(C, Co, 8, 1, seed=18, scale=0.2), rnd(Co, seed=19)
        skip = rnd(B, Co, 4 * Fr, T, seed=20)
        want = F.gelu(F.conv_transpose2d(x, W, b, stride=(4, 1))[..., 2:-2, :]) + skip
    else:
        B, C, Co, L, Lout = 2, 24, 12, 345, 1377        # ragged: the reference crops [2 : 2 + length]
        x, W, b = rnd(B, C, L, seed=21), rnd(C, Co, 8, seed=22, scale=0.2), rnd(Co, seed=23)
        skip = rnd(B, Co, Lout, seed=24)
        want = F.gelu(F.conv_transpose1d(x, W, b, stride=4)[..., 2:2 + Lout]) + skip
    Wr = W.reshape(C, Co, 8)
    W2 = torch.zeros(4 * Co, 2 * C, dtype=torch.float64)       # row 4co+r, col 2ci+j  <- W[ci][co][r+4j]
    for r in range(4):
        for j in range(2):
            W2[r::4, j::2] = Wr[:, :, r + 4 * j].t()
    wt, bias, M, Mpad, K, Kpad, tile = pack_w(W2, b.repeat_interleave(4))
    if freq:
        kt = ktab(C, 2, 1, -1, 1, 0, 0, Fr * T, T, Kpad)
        y = torch.empty(B, Co, 4 * Fr, T, device="cuda")
        conv_call(x6=x6, wt=wt, M=M, Mpad=Mpad, K=K, Kpad=Kpad, ktab=kt, x=x.float().cuda(), x_bstride=C * Fr * T, B=B, D1=Fr, D2=T,
                  O1=Fr + 1, O2=T, S1=1, S2=1, row_mode=1, epi=EPI_CONVTR, flags=FLAG_TR_FREQ | FLAG_GELU | FLAG_RES,
                  res=skip.float().cuda(), bias=bias, y=y, y_bstride=Co * 4 * Fr * T, y_cstride=4 * Fr * T, out_len=4 * Fr,
                  tile_m=tile)
    else:
        kt = ktab(C, 1, 2, 1, -1, 0, 0, L, L, Kpad)
        y = torch.empty(B, Co, Lout, device="cuda")
        conv_call(x6=x6, wt=wt, M=M, Mpad=Mpad, K=K, Kpad=Kpad, ktab=kt, x=x.float().cuda(), x_bstride=C * L, B=B, D1=1, D2=L, O1=1,
                  O2=L + 1, S1=1, S2=1, epi=EPI_CONVTR, flags=FLAG_GELU | FLAG_RES, res=skip.float().cuda(), bias=bias, y=y,
                  y_bstride=Co * Lout, y_cstride=Lout, out_len=Lout, tile_m=tile)
    assert maxerr(y, want) < 2e-5


@pytest.mark.parametrize("freq", [True, False])
@pytest.mark.parametrize("mode", ["bf16", "f16"])
def test_conv_transpose_on_operand_image(lib, freq, mode):
    """Half modes: the transposed conv as a two-tap gather on the operand image of its input (gemm_tap.hip, taps q and q - 1 by
    LDS-DMA) against the table-driven route on the same rounded operands and against float64; the float32 -> image pass
    (`mi_f32_to_image`) is part of the route.  Frequency axis (an extra output row) and ragged time axis with a row pitch."""
    hdt = torch.bfloat16 if mode == "bf16" else torch.float16
    dt = {"bf16": 1, "f16": 2}[mode]
    if freq:
        B, C, Co, Fr, T = 2, 24, 12, 8, 40
        x, W, b = rnd(B, C, Fr, T, seed=17), rnd(C, Co, 8, 1, seed=18, scale=0.2), rnd(Co, seed=19)
        skip = rnd(B, Co, 4 * Fr, T, seed=20)
        want = F.gelu(F.conv_transpose2d(x.to(hdt).double(), W.to(hdt).double(), b.double(), stride=(4, 1))[..., 2:-2, :]) + skip
        P, xs = Fr * T, x
    else:
        B, C, Co, L, Lp, Lout = 2, 24, 12, 345, 348, 1377
        x, W, b = rnd(B, C, L, seed=21), rnd(C, Co, 8, seed=22, scale=0.2), rnd(Co, seed=23)
        skip = rnd(B, Co, Lout, seed=24)
        want = F.gelu(F.conv_transpose1d(x.to(hdt).double(), W.to(hdt).double(), b.double(), stride=4)[..., 2:2 + Lout]) + skip
        P = Lp
        xs = torch.zeros(B, C, Lp)
        xs[..., :L] = x
    Wr = W.reshape(C, Co, 8)
    W2 = torch.zeros(4 * Co, 2 * C, dtype=torch.float64)
    for r in range(4):
        for j in range(2):
            W2[r::4, j::2] = Wr[:, :, r + 4 * j].t()
    wt, bias, M, Mpad, K, Kpad, tile = pack_w(W2, b.repeat_interleave(4))
    xd = xs.float().cuda().contiguous()
    img = torch.empty(C // 8, B * P, 8, dtype=torch.int16, device="cuda")
    _lib.check(lib.mi_f32_to_image(xd.data_ptr(), B, C, P, dt, img.data_ptr(), stream()), "mi_f32_to_image")
    torch.cuda.synchronize()
    assert torch.equal(img, xs.to(hdt).permute(1, 0, 2, 3).reshape(C // 8, 8, B * P).permute(0, 2, 1).contiguous().view(torch.int16).cuda()
                       if freq else xs.to(hdt).permute(1, 0, 2).reshape(C // 8, 8, B * P).permute(0, 2, 1).contiguous().view(torch.int16).cuda())
    pairs = (C // 8 * 2 + 3) // 4 * 4
    wtap = torch.empty(pairs * Mpad * 8, dtype=torch.int16, device="cuda")
    _lib.check(lib.mi_conv_pack_tap(wt.data_ptr(), Mpad, C, 2, dt, wtap.data_ptr(), stream()), "mi_conv_pack_tap")
    if freq:
        common = dict(x6=mode, wt=wt, M=M, Mpad=Mpad, K=K, Kpad=Kpad, ktab=ktab(C, 2, 1, -1, 1, 0, 0, Fr * T, T, Kpad), x=xd, x_bstride=C * P, B=B,
                      D1=Fr, D2=T, O1=Fr + 1, O2=T, S1=1, S2=1, row_mode=1, epi=EPI_CONVTR, flags=FLAG_TR_FREQ | FLAG_GELU | FLAG_RES,
                      res=skip.float().cuda(), bias=bias, y_bstride=Co * 4 * Fr * T, y_cstride=4 * Fr * T, out_len=4 * Fr, tile_m=tile)
        tap = dict(tap_k2=1, tap_dil1=-1)
        y_ref, y_tap = torch.empty(B, Co, 4 * Fr, T, device="cuda"), torch.empty(B, Co, 4 * Fr, T, device="cuda")
    else:
        common = dict(x6=mode, wt=wt, M=M, Mpad=Mpad, K=K, Kpad=Kpad, ktab=ktab(C, 1, 2, 1, -1, 0, 0, Lp, Lp, Kpad), x=xd, x_bstride=C * Lp, B=B,
                      D1=1, D2=L, O1=1, O2=L + 1, S1=1, S2=1, epi=EPI_CONVTR, flags=FLAG_GELU | FLAG_RES, res=skip.float().cuda(), bias=bias,
                      y_bstride=Co * Lout, y_cstride=Lout, out_len=Lout, tile_m=tile, x_ld=Lp)
        tap = dict(tap_k2=2, tap_dil2=-1)
        y_ref, y_tap = torch.empty(B, Co, Lout, device="cuda"), torch.empty(B, Co, Lout, device="cuda")
    conv_call(y=y_ref, **common)
    conv_call(y=y_tap, xh=img, xh_n=B * P, wtap=wtap, ntaps=2, **tap, **common)
    e_ref, e_tap = maxerr(y_ref, want), maxerr(y_tap, want)
    print(f"transposed conv {mode} freq={freq}: gather route {e_ref:.2e}, image route {e_tap:.2e}")
    assert e_tap < 2e-5 and e_ref < 2e-5 and maxerr(y_tap, y_ref) < 1e-5


def _phase_image(v, axis_len, pq, freq, T, hdt):
    """Host restatement of gemm_conv.h MI_FLAG_IMG4: v (B, C, Len[, T]) float -> int16 image [(C/8) * 4][B * pq][8]; slot
    q = i // 4 + (i % 4 >= 2) of plane i % 4 (frequency rows: position q * T + t)."""
    B, C = v.shape[:2]
    img = torch.zeros(C // 8 * 4, B * pq, 8, dtype=hdt)
    for i in range(axis_len):
        rho, q = i % 4, i // 4 + (1 if i % 4 >= 2 else 0)
        for b in range(B):
            if freq:
                blk = v[b, :, i, :].to(hdt).reshape(C // 8, 8, T).permute(0, 2, 1)                  # (oct, T, 8)
                img[rho::4][:, b * pq + q * T:b * pq + (q + 1) * T, :] = blk
            else:
                img[rho::4][:, b * pq + q, :] = v[b, :, i].to(hdt).reshape(C // 8, 8)
    return img.view(torch.int16)


@pytest.mark.parametrize("freq", [True, False])
@pytest.mark.parametrize("mode", ["bf16", "f16"])
def test_encoder_conv_on_phase_split_image(lib, freq, mode):
    """Half modes, encoder levels 1-3: the previous level's 1x1 + GLU epilogue also writes its result as a PHASE-SPLIT 16-bit image
    (MI_FLAG_IMG4), on which the strided conv (k = 8, s = 4, pad 2; hdemucs.py:110,132-136) is a stride-1 two-tap conv over 4 C
    channels run by LDS-DMA (gemm_tap.hip).  Checks (a) the epilogue's image against the host restatement of the layout, built from
    the float32 output of the same launch (never-written slots stay zero), (b) the conv on that image against float64 on the
    rounded operands, bias + GELU included; frequency rows and a ragged time axis (L = 1373 -> 344)."""
    hdt = torch.bfloat16 if mode == "bf16" else torch.float16
    dt = {"bf16": 1, "f16": 2}[mode]
    _HALF_MODE[0] = mode
    try:
        B, C, Co = 2, 32, 40                   # 2 C output rows of the producer: a multiple of 32 (whole 16-channel tiles)
        if freq:
            Fr, T = 32, 40
            Len, P, Q = Fr, Fr * T, Fr // 4 + 1
            pq = Q * T
            x = rnd(B, 2 * C, Fr, T, seed=31)
        else:
            L, Lp = 1373, 1376
            Len, P, Q = L, Lp, (L + 3) // 4 + 1
            pq = (Q + 3) // 4 * 4
            x = rnd(B, 2 * C, Lp, seed=31)
            x[..., L:] = 0
        Wr, br = rnd(2 * C, 2 * C, seed=32, scale=0.15), rnd(2 * C, seed=33)
        # (a) producer: 1x1 conv 2C -> 2C + GLU -> C channels, float32 y and the phase image
        wt, bias, M, Mpad, K, Kpad, tile = pack_w(Wr, br, glu=True)
        kt = ktab(2 * C, 1, 1, 1, 1, 0, 0, P, P if not freq else T, Kpad)
        y = torch.zeros(B, C, *((Fr, T) if freq else (Lp,)), device="cuda")
        img = torch.zeros(C // 8 * 4, B * pq, 8, dtype=torch.int16, device="cuda")
        geo = dict(D1=Fr, D2=T, O1=Fr, O2=T, row_mode=1) if freq else dict(D1=1, D2=L, O1=1, O2=Lp, o2_valid=L, x_ld=Lp)
        conv_call(x6=mode, wt=wt, M=M, Mpad=Mpad, K=K, Kpad=Kpad, ktab=kt, x=x.float().cuda(), x_bstride=2 * C * P, B=B, S1=1, S2=1, plain=1,
                  epi=EPI_GLU, flags=512 | (FLAG_TR_FREQ if freq else 0), bias=bias, y=y, y_bstride=C * P, y_cstride=P, tile_m=tile,
                  yh=img, yh_n=B * pq, yh_pq=pq, **geo)
        xd = x.double()
        z = torch.einsum("mk,bk...->bm...", Wr.double(), xd) + br.double().reshape(1, -1, *([1] * (x.dim() - 2)))
        want_y = F.glu(z, dim=1)
        yv = y.cpu() if freq else y.cpu()[..., :L]
        assert maxerr(yv.cuda(), want_y if freq else want_y[..., :L]) < 2e-5
        want_img = _phase_image(yv, Len, pq, freq, T if freq else 0, hdt)
        assert torch.equal(img.cpu(), want_img), "phase-split image differs from the layout restatement"
        # (b) consumer: strided conv C -> Co on the image
        W, b = rnd(Co, C, 8, seed=34, scale=0.2), rnd(Co, seed=35)
        yr = yv.to(hdt).double()                                   # the rounded operand the image holds
        if freq:
            want = F.gelu(F.conv2d(yr, W.to(hdt).double()[..., None], b.double(), stride=(4, 1), padding=(2, 0)))
            Lo = Fr // 4
        else:
            want = F.gelu(F.conv1d(F.pad(yr, (0, (-L) % 4)), W.to(hdt).double(), b.double(), stride=4, padding=2))
            Lo = want.shape[-1]
        wt2d = torch.zeros(Co, 8 * C, dtype=torch.float64)         # k' = ((oct * 4 + rho) * 8 + c8) * 2 + j  <-  tap 4 (j - (rho >= 2)) + rho + 2
        for ci in range(C):
            for rho in range(4):
                for j in range(2):
                    tau = 4 * (j - (1 if rho >= 2 else 0)) + rho + 2
                    wt2d[:, (((ci // 8) * 4 + rho) * 8 + ci % 8) * 2 + j] = W[:, ci, tau]
        wt2, bias2, M2, Mpad2, K2, Kpad2, tile2 = pack_w(wt2d, b)
        pairs = (4 * C // 8 * 2 + 3) // 4 * 4
        wtap = torch.empty(pairs * Mpad2 * 8, dtype=torch.int16, device="cuda")
        _lib.check(lib.mi_conv_pack_tap(wt2.data_ptr(), Mpad2, 4 * C, 2, dt, wtap.data_ptr(), stream()), "mi_conv_pack_tap")
        kt2 = ktab(C, 8, 1, 1, 1, 2, 0, 8, 8, Kpad2)               # unused by the image route (the descriptor still carries a table)
        if freq:
            y2 = torch.empty(B, Co, Lo, T, device="cuda")
            geo2 = dict(D1=Q, D2=T, O1=Lo, O2=T, row_mode=1, tap_k2=1, y_bstride=Co * Lo * T, y_cstride=Lo * T)
        else:
            Lop = (Lo + 3) // 4 * 4
            y2 = torch.zeros(B, Co, Lop, device="cuda")
            geo2 = dict(D1=1, D2=pq, x_ld=pq, O1=1, O2=Lop, o2_valid=Lo, tap_k2=2, y_bstride=Co * Lop, y_cstride=Lop)
        conv_call(x6=mode, wt=wt2, M=M2, Mpad=Mpad2, K=K2, Kpad=Kpad2, ktab=kt2, x=y, x_bstride=C * P, B=B, S1=1, S2=1, epi=EPI_LINEAR,
                  flags=FLAG_GELU, bias=bias2, y=y2, tile_m=tile2, xh=img, xh_n=B * pq, wtap=wtap, ntaps=2, **geo2)
        got = y2 if freq else y2[..., :Lo]
        err = maxerr(got, want)
        print(f"encoder conv on the phase-split image, {mode}, freq={freq}: {err:.2e} vs float64 on the rounded operands")
        assert err < 3e-5
    finally:
        _HALF_MODE[0] = None


@pytest.mark.parametrize("freq", [True, False])
def test_dconv_layer_three_passes(lib, freq, x6):
    """One DConv residual layer (demucs.py:138-143,151-154): dilated conv3 + per-row statistics,
    in-place GroupNorm(1)+GELU pass, 1x1 with statistics-only pass, then GroupNorm + GLU + LayerScale +
    residual epilogue.  Frequency branch rows are (b, fr); time branch rows are b."""
    C, h, dil = 48, 6, 2
    if freq:
        B, Fr, T = 2, 5, 336
        x4 = rnd(B, C, Fr, T, seed=25)
        rows_x = x4.permute(0, 2, 1, 3).reshape(B * Fr, C, T)
        D1, D2, row_mode, nrows = Fr, T, 1, B * Fr
    else:
        B, L = 2, 2000
        x4 = rnd(B, C, L, seed=25)
        rows_x = x4
        D1, D2, row_mode, nrows = 1, L, 0, B
    W0, b0 = rnd(h, C, 3, seed=26, scale=0.2), rnd(h, seed=27)
    g1w, g1b = 1 + 0.2 * rnd(h, seed=28), 0.1 * rnd(h, seed=29)
    W3, b3 = rnd(2 * C, h, 1, seed=30, scale=0.5), rnd(2 * C, seed=31)
    g2w, g2b = 1 + 0.2 * rnd(2 * C, seed=32), 0.1 * rnd(2 * C, seed=33)
    ls = 1 + 0.3 * rnd(C, seed=34)
    y = F.conv1d(rows_x, W0, b0, dilation=dil, padding=dil)
    y = F.gelu(F.group_norm(y, 1, g1w, g1b, eps=1e-5))
    y = F.group_norm(F.conv1d(y, W3, b3), 1, g2w, g2b, eps=1e-5)
    want_rows = rows_x + ls[:, None] * F.glu(y, dim=1)
    want = want_rows.view(B, Fr, C, T).permute(0, 2, 1, 3) if freq else want_rows

    P = D1 * D2
    xd = x4.float().cuda().contiguous()
    # pass 1: conv3 + bias, per-row statistics
    wt0, bias0, M0, Mpad0, K0, Kpad0, tile0 = pack_w(W0.reshape(h, -1), b0)
    kt0 = ktab(C, 1, 3, 1, dil, 0, dil, P, D2, Kpad0)
    hid = torch.empty(B, h, P, device="cuda")
    stats = torch.zeros(nrows, SLOTS, 2, dtype=torch.float64, device="cuda")
    conv_call(x6=x6, wt=wt0, M=M0, Mpad=Mpad0, K=K0, Kpad=Kpad0, ktab=kt0, x=xd, x_bstride=C * P, B=B, D1=D1, D2=D2, O1=D1, O2=D2, S1=1,
              S2=1, row_mode=row_mode, epi=EPI_BIAS_STATS, bias=bias0, y=hid, y_bstride=h * P, y_cstride=P, stats=stats,
              tile_m=tile0)
    s = stats.sum(1).cpu()
    cnt = h * D2 if freq else h * P
    mean = s[:, 0] / cnt
    var = s[:, 1] / cnt - mean ** 2
    ref_h = F.conv1d(rows_x, W0, b0, dilation=dil, padding=dil)
    assert (mean - ref_h.mean(dim=(1, 2))).abs().max() < 1e-6
    assert (var - ref_h.var(dim=(1, 2), unbiased=False)).abs().max() < 1e-5
    st1 = torch.stack([mean, 1.0 / torch.sqrt(var + 1e-5)], 1).float().cuda().contiguous()
    # pass 2: statistics of the 1x1 output (nothing stored)
    wt3, bias3, M3, Mpad3, K3, Kpad3, tile3 = pack_w(W3.reshape(2 * C, h), b3, glu=True)
    kt3 = ktab(h, 1, 1, 1, 1, 0, 0, P, D2, Kpad3)
    stats.zero_()
    g1wd, g1bd = g1w.float().cuda(), g1b.float().cuda()
    _lib.check(lib.mi_gn_gelu(hid.data_ptr(), B, h, h, D1, D2, row_mode, st1.data_ptr(), g1wd.data_ptr(), g1bd.data_ptr(), stream()),
               "mi_gn_gelu")
    torch.cuda.synchronize()
    common = dict(wt=wt3, M=M3, Mpad=Mpad3, K=K3, Kpad=Kpad3, ktab=kt3, x=hid, x_bstride=h * P, B=B, D1=D1, D2=D2, O1=D1, O2=D2,
                  S1=1, S2=1, row_mode=row_mode, bias=bias3, tile_m=tile3, plain=1)
    conv_call(x6=x6, epi=EPI_STATS_ONLY, stats=stats, **common)
    s = stats.sum(1).cpu()
    cnt = 2 * C * (D2 if freq else P)
    mean2 = s[:, 0] / cnt
    var2 = s[:, 1] / cnt - mean2 ** 2
    st2 = torch.stack([mean2, 1.0 / torch.sqrt(var2 + 1e-5)], 1).float().cuda().contiguous()
    # pass 3: recompute, GroupNorm + GLU + LayerScale + residual
    out = torch.empty_like(xd)
    conv_call(x6=x6, epi=EPI_GN_GLU, gn_stats=st2, gn_w=pack_vec(g2w, Mpad3, glu=True), gn_b=pack_vec(g2b, Mpad3, glu=True),
              scale=ls.float().cuda(), res=xd, y=out, y_bstride=C * P, y_cstride=P, **common)
    assert maxerr(out, want) < chained_tol(3e-5)


def test_gelu_erf_form_accuracy(lib):
    """`gelu_exact` (csrc/common.h) through mi_gn_gelu with identity statistics, on a dense grid and on tiny arguments,
    against the float64 erf form of F.gelu (reference: nn.GELU default in demucs/demucs.py:116 and transformer.py:339).
    Bar: 1.5e-7 |x| + 1e-9 -- tighter than float32 F.gelu on the host reaches."""
    n = 1 << 20
    x = torch.cat([torch.linspace(-9, 9, n - 4096, dtype=torch.float64), torch.logspace(-8, 0, 2048, dtype=torch.float64),
                   -torch.logspace(-8, 0, 2048, dtype=torch.float64)]).float()
    xd = x.clone().cuda().view(1, 1, 1, n)
    st = torch.tensor([[0.0, 1.0]], device="cuda")
    one, zero = torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")
    _lib.check(lib.mi_gn_gelu(xd.data_ptr(), 1, 1, 1, 1, n, 0, st.data_ptr(), one.data_ptr(), zero.data_ptr(), stream()), "mi_gn_gelu")
    torch.cuda.synchronize()
    x64 = x.double()
    want = 0.5 * x64 * (1 + torch.erf(x64 * math.sqrt(0.5)))
    err = (xd.view(-1).cpu().double() - want).abs()
    rel = (err / (1.5e-7 * x64.abs() + 1e-9)).max().item()
    print(f"gelu: max abs err {err.max():.2e}, worst err / (1.5e-7 |x| + 1e-9) = {rel:.2f}")
    assert rel <= 1.0


@pytest.mark.parametrize("case", ["freq_h24", "time_h48", "freq_h96"])
def test_gn_gelu_gram_gives_second_groupnorm_statistics(lib, case):
    """DConv's implicit-GEMM route (csrc/model.hip run_dconv): GroupNorm(1) + GELU of the hidden tensor in place and, from the
    Gram sums of the same pass, the statistics of GroupNorm(1, 2C) after the 1x1 conv z = W g + b (demucs/demucs.py:139-142),
    against a float64 evaluation of z.  Pitch-padded rows, ragged column tiles, both row modes, several accumulator slots."""
    B, h, hp, D1, D2, pitch, row_mode, slots, C2 = {"freq_h24": (2, 24, 32, 3, 200, 208, 1, 1, 384),
                                                     "time_h48": (2, 48, 48, 1, 1000, 1000, 0, 8, 768),
                                                     "freq_h96": (1, 96, 96, 2, 150, 152, 1, 1, 768)}[case]
    rows = B * D1 if row_mode else B
    x = rnd(B, hp, D1, pitch, seed=70, scale=1.5).float()
    st1 = torch.stack([rnd(rows, seed=71, scale=0.2), rnd(rows, seed=72).abs() + 0.5], 1).float().contiguous()
    w, bb = (rnd(h, seed=73) + 1.0).float(), rnd(h, seed=74, scale=0.3).float()
    W, b1 = rnd(C2, h, seed=75, scale=0.4).float().double(), rnd(C2, seed=76, scale=0.3).float().double()
    # float64 reference
    xs = x[:, :h, :, :D2].double()
    srow = st1.double().view(B, D1, 2) if row_mode else st1.double().view(B, 1, 2).expand(B, D1, 2)
    g = F.gelu((xs - srow[:, None, :, 0:1]) * srow[:, None, :, 1:2] * w.double()[None, :, None, None] + bb.double()[None, :, None, None])
    z = torch.einsum("mh,bhdt->bmdt", W, g) + b1[None, :, None, None]
    zr = z.permute(0, 2, 1, 3).reshape(rows, -1) if row_mode else z.reshape(B, -1)
    mean, var = zr.mean(1), zr.var(1, unbiased=False)
    # weights of the accumulators as csrc/model.hip load_dconv builds them
    HP = lib.mi_gram_order(h)
    A = W.t() @ W
    wt = torch.zeros(HP, HP, dtype=torch.float64)
    for i in range(h):
        for k in range(h):
            if k // 32 >= i // 32:
                wt[i, k] = A[i, k] * (2.0 if k // 32 > i // 32 else 1.0)
    wt[:h, h] = 2.0 * (W.t() @ b1)
    ct = torch.zeros(HP, dtype=torch.float64)
    ct[:h] = W.sum(0)
    cols = D2 if row_mode else D1 * D2
    xd, std, wd, bd = x.clone().cuda(), st1.cuda(), w.cuda(), bb.cuda()
    gram = torch.zeros(rows * slots * HP * HP, dtype=torch.float64, device="cuda")
    out = torch.empty(rows, 2, device="cuda")
    _lib.check(lib.mi_gn_gelu_gram(xd.data_ptr(), B, h, hp, D1, D2, pitch, row_mode, std.data_ptr(), wd.data_ptr(), bd.data_ptr(),
                                   gram.data_ptr(), slots, stream()), "mi_gn_gelu_gram")
    wtd, ctd = wt.cuda(), ct.cuda()
    _lib.check(lib.mi_gram_finalize(gram.data_ptr(), rows, h, slots, wtd.data_ptr(), ctd.data_ptr(), float(b1.sum()), float((b1 * b1).sum()),
                                    float(cols), float(cols * C2), 1e-5, out.data_ptr(), stream()), "mi_gram_finalize")
    torch.cuda.synchronize()
    assert torch.allclose(xd[:, :h, :, :D2].cpu().double(), g, rtol=3e-6, atol=3e-6)           # normalised + GELU in place
    assert torch.equal(xd[:, :, :, D2:].cpu(), x[:, :, :, D2:]) and torch.equal(xd[:, h:].cpu(), x[:, h:])      # pads untouched
    got = out.cpu().double()
    want_rstd = 1.0 / torch.sqrt(var + 1e-5)
    e_mean, e_rstd = (got[:, 0] - mean).abs().max().item(), ((got[:, 1] - want_rstd).abs() / want_rstd).max().item()
    print(f"{case}: mean err {e_mean:.2e}, rstd rel err {e_rstd:.2e}")
    assert e_mean < 2e-6 * max(1.0, float(mean.abs().max())) and e_rstd < 2e-6
    assert float(gram.abs().max()) == 0.0                                    # accumulators re-zeroed


@pytest.mark.parametrize("mode", ["bf16", "f16"])
def test_ffn_operand_image_round_trip(lib, mode):
    """Half modes, transformer FFN (transformer.py:339-343 linear1 -> GELU -> linear2): lin1's epilogue writes the hidden
    tensor as lin2's 16-bit operand image (MI_FLAG_IMG) and lin2 consumes it by DMA (mi_conv_desc.xh).  The image must hold
    exactly the rounding of the float32 result the plain epilogue stores, laid out [rows / 8][columns][8]; lin2 on the image
    must equal lin2 on the float32 tensor (same operands after rounding).  Ragged column count, 128- and 256-row tiles."""
    B, P, Kin, H = 2, 168, 64, 256                       # N = 336: two full column tiles and a ragged one
    N = B * P
    x = rnd(B, Kin, P, seed=80).float().cuda()
    W1, b1 = rnd(H, Kin, seed=81, scale=0.3), rnd(H, seed=82, scale=0.2)
    wt1, bias1, M1, Mpad1, K1, Kpad1, _ = pack_w(W1, b1, tile=128)
    ident = torch.stack([torch.zeros(N), torch.ones(N)], 1).float().cuda().contiguous()      # LayerNorm fold with mean 0, rstd 1
    c1 = torch.zeros(Mpad1, device="cuda")
    common1 = dict(wt=wt1, M=M1, Mpad=Mpad1, K=K1, Kpad=Kpad1, ktab=ktab(Kin, 1, 1, 1, 1, 0, 0, P, P, Kpad1), x=x, x_bstride=Kin * P,
                   B=B, D1=1, D2=P, O1=1, O2=P, S1=1, S2=1, bias=bias1, epi=EPI_LINEAR, scale=c1, pro_stats=ident, tile_m=128, plain=1, x6=mode)
    hid = torch.empty(B, H, P, device="cuda")
    conv_call(flags=32 | FLAG_GELU, y=hid, y_bstride=H * P, y_cstride=P, **common1)
    img = torch.zeros(H // 8, N, 8, dtype=torch.int16, device="cuda")
    conv_call(flags=32 | FLAG_GELU | 64, yh=img, yh_n=N, y_bstride=H * P, y_cstride=P, **common1)
    hdt = torch.bfloat16 if mode == "bf16" else torch.float16
    want_img = hid.permute(1, 0, 2).reshape(H // 8, 8, N).permute(0, 2, 1).to(hdt).contiguous().view(torch.int16)
    assert torch.equal(img, want_img)
    for M2 in (256, 128):                                # 256-row and 128-row tiles of the image-input kernel
        W2, b2 = rnd(M2, H, seed=83, scale=0.2), rnd(M2, seed=84, scale=0.2)
        wt2, bias2, _, Mpad2, K2, Kpad2, _ = pack_w(W2, b2, tile=128)
        scale, res = pack_vec(rnd(M2, seed=85) + 1.0, Mpad2), rnd(B, M2, P, seed=86).float().cuda()
        common2 = dict(wt=wt2, M=M2, Mpad=Mpad2, K=K2, Kpad=Kpad2, ktab=ktab(H, 1, 1, 1, 1, 0, 0, P, P, Kpad2), x=hid, x_bstride=H * P,
                       B=B, D1=1, D2=P, O1=1, O2=P, S1=1, S2=1, bias=bias2, epi=EPI_LINEAR, flags=FLAG_SCALE | FLAG_RES, scale=scale, res=res,
                       y_bstride=M2 * P, y_cstride=P, tile_m=128, plain=1, x6=mode)
        y_ref, y_img = torch.empty(B, M2, P, device="cuda"), torch.empty(B, M2, P, device="cuda")
        conv_call(y=y_ref, **common2)
        conv_call(y=y_img, xh=img, xh_n=N, **common2)
        assert maxerr(y_img, y_ref) < 1e-5


@pytest.mark.parametrize("dtype", [0, 1, 2], ids=["f32", "bf16", "f16"])
def test_attention_matches_softmax(lib, dtype):
    """softmax(QK^T/8)V per head on channel-first q/k/v, ragged Tq (not a multiple of 128), cross
    lengths, and one spiked key forcing a large running-max jump mid-stream."""
    B, H, Tq, Tk = 2, 8, 200, 320
    q, k, v = rnd(B, 512, Tq, seed=40), rnd(B, 512, Tk, seed=41), rnd(B, 512, Tk, seed=42)
    k[:, :, 170] *= 6.0
    Q = q.view(B, H, 64, Tq).transpose(2, 3)
    K = k.view(B, H, 64, Tk).transpose(2, 3)
    V = v.view(B, H, 64, Tk).transpose(2, 3)
    want = (torch.softmax(Q @ K.transpose(-1, -2) / 8.0, dim=-1) @ V).transpose(2, 3).reshape(B, 512, Tq)
    kv = torch.cat([k, v], 1).float().cuda().contiguous()          # (B, 1024, Tk) like the packed KV projection
    o = torch.empty(B, 512, Tq, device="cuda")
    qd = q.float().cuda()
    _lib.check(lib.mi_attention(qd.data_ptr(), kv.data_ptr(), kv.data_ptr() + 512 * Tk * 4, o.data_ptr(), B, H, Tq, Tk, 512 * Tq,
                                1024 * Tk, 512 * Tq, dtype, stream()), "mi_attention")
    torch.cuda.synchronize()
    # half modes: Q, K, V and the probabilities are rounded to the operand type (2^-9 / 2^-12 relative), softmax in float32
    assert maxerr(o, want) < [2e-5, 6e-2, 8e-3][dtype]        # scores reach |s| ~ 30 at the spiked key: 2^-9 (2^-12) relative on them moves a probability by several percent
    if dtype:
        # the same result as out_proj's 16-bit operand image: exactly the rounding of the float32 output, [channel / 8][b Tq + q][8]
        img = torch.zeros(512 // 8, B * Tq, 8, dtype=torch.int16, device="cuda")
        _lib.check(lib.mi_attention_image(qd.data_ptr(), kv.data_ptr(), kv.data_ptr() + 512 * Tk * 4, img.data_ptr(), B * Tq, B, H, Tq, Tk,
                                          512 * Tq, 1024 * Tk, dtype, stream()), "mi_attention_image")
        torch.cuda.synchronize()
        hdt = torch.bfloat16 if dtype == 1 else torch.float16
        want_img = o.permute(1, 0, 2).reshape(64, 8, B * Tq).permute(0, 2, 1).to(hdt).contiguous().view(torch.int16)
        assert torch.equal(img, want_img)


@pytest.mark.parametrize("dtype", [1, 2], ids=["bf16", "f16"])
@pytest.mark.parametrize("Tq,Tk,pitch,spike", [(200, 320, 64, True), (200, 320, 0, False), (333, 97, 8, False), (64, 2688, 0, False)])
def test_attention_heads_matches_softmax(lib, dtype, Tq, Tk, pitch, spike):
    """`mi_attention_heads` (attention_heads.hip): the half modes' attention on the 16-bit per-head token-major operands the
    in-projections write (K / V tiles by LDS-DMA, swizzled LDS image, transposed V reads).  The float64 softmax of the SAME
    rounded operands is the reference, so only the probabilities' rounding (2^-9 / 2^-12 relative) and float32 accumulation
    remain: tight bounds on un-spiked data; the spiked case (|score| ~ 30) gets the bound of test_attention_matches_softmax.
    Ragged Tq / Tk (last tile masked, rows past Tk read from the zero page), row pitches above the token counts, a long key axis."""
    B, H = 2, 8
    hdt = torch.bfloat16 if dtype == 1 else torch.float16
    q, k, v = rnd(B, H, Tq, 64, seed=140), rnd(B, H, Tk, 64, seed=141), rnd(B, H, Tk, 64, seed=142)
    if spike:
        k[:, :, 170] *= 6.0
    qh, kh, vh = q.to(hdt), k.to(hdt), v.to(hdt)
    Q, K, V = qh.double(), kh.double(), vh.double()
    want = (torch.softmax(Q @ K.transpose(-1, -2) / 8.0, dim=-1) @ V).transpose(2, 3).reshape(B, 512, Tq)
    Tqp, Tkp = Tq + pitch, Tk + pitch

    def padded(t, Tp):           # rows past T hold NaNs: they must never reach a product
        out = torch.full((B, H, Tp, 64), float("nan"), dtype=hdt)
        out[:, :, :t.shape[2]] = t
        return out.cuda().contiguous()
    qd, kd, vd = padded(qh, Tqp), padded(kh, Tkp), padded(vh, Tkp)
    o = torch.empty(B, 512, Tq, device="cuda")
    _lib.check(lib.mi_attention_heads(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), o.data_ptr(), B, H, Tq, Tk, Tqp, Tkp, dtype, stream()),
               "mi_attention_heads")
    torch.cuda.synchronize()
    err = maxerr(o, want)
    print(f"attention_heads dtype {dtype} Tq {Tq} Tk {Tk}: max-abs vs float64 softmax of the rounded operands {err:.2e}")
    assert bool(torch.isfinite(o).all())
    assert err < ([0, 6e-2, 8e-3][dtype] if spike else [0, 6e-3, 8e-4][dtype])


@pytest.mark.parametrize("mode", ["bf16", "f16"])
@pytest.mark.parametrize("C,Fr,T,Tp", [(16, 6, 40, 40), (48, 5, 77, 77), (24, 1, 301, 304)], ids=["3x3", "3x3-wide", "k3-pitched"])
def test_tap_image_conv_matches_gather_route(lib, mode, C, Fr, T, Tp):
    """gemm_tap.hip: the decoders' k x k GLU convs in the half modes read their input as a 16-bit operand image
    [C / 8][positions][8] and gather the taps by LDS-DMA (k ordered channel-octet, tap, channel % 8).  Same rounded operands as
    the table-driven float32 -> 16-bit route of gemm_half.hip, so the results agree to float32 summation-order noise; and the
    float64 convolution of the ROUNDED operands bounds both.  3 x 3 with frame edges, a ragged column tile, and the k = 3
    time-branch form with a padded row pitch (image columns past the valid length hold NaN: they must never be gathered)."""
    B = 2
    time = Fr == 1
    hdt = torch.bfloat16 if mode == "bf16" else torch.float16
    dt = {"bf16": 1, "f16": 2}[mode]
    x = rnd(B, C, Fr, T, seed=201)
    W, b = (rnd(2 * C, C, 1, 3, seed=202, scale=0.2) if time else rnd(2 * C, C, 3, 3, seed=202, scale=0.1)), rnd(2 * C, seed=203)
    xr, Wr = x.to(hdt).double(), W.to(hdt).double()
    z = F.conv2d(xr, Wr, b.double(), padding=(0, 1) if time else 1)
    want = z[:, :C] * torch.sigmoid(z[:, C:])
    ntaps = W.shape[2] * W.shape[3]
    wt, bias, M, Mpad, K, Kpad, tile = pack_w(W.reshape(2 * C, -1), b, glu=True)
    P = Fr * Tp
    # the input as its operand image: position n = b * Fr * Tp + fr * Tp + t; pitch columns are NaN
    xp = torch.full((B, C, Fr, Tp), float("nan"))
    xp[..., :T] = x
    img = xp.to(hdt).permute(1, 0, 2, 3).reshape(C // 8, 8, B * P).permute(0, 2, 1).contiguous().cuda()
    pairs = (C // 8 * ntaps + 3) // 4 * 4
    wtap = torch.empty(pairs * Mpad * 8, dtype=torch.int16, device="cuda")
    _lib.check(lib.mi_conv_pack_tap(wt.data_ptr(), Mpad, C, ntaps, dt, wtap.data_ptr(), stream()), "mi_conv_pack_tap")
    xd = torch.zeros(B, C, Fr, Tp)
    xd[..., :T] = x
    xd = xd.float().cuda()
    common = dict(x6=mode, wt=wt, M=M, Mpad=Mpad, K=K, Kpad=Kpad, ktab=ktab(C, W.shape[2], 3, 1, 1, 0 if time else 1, 1, P, Tp, Kpad), x=xd,
                  x_bstride=C * P, B=B, D1=Fr, D2=T, O1=Fr, O2=Tp, S1=1, S2=1, row_mode=0 if time else 1, epi=EPI_GLU, bias=bias,
                  y_bstride=C * P, y_cstride=P, tile_m=tile, o2_valid=T if Tp != T else 0, x_ld=Tp if Tp != T else 0)
    y_ref, y_tap = torch.zeros(B, C, Fr, Tp, device="cuda"), torch.zeros(B, C, Fr, Tp, device="cuda")
    conv_call(y=y_ref, **common)
    conv_call(y=y_tap, xh=img, xh_n=B * P, wtap=wtap, ntaps=ntaps, tap_k2=3, tap_pad1=0 if time else 1, tap_pad2=1, **common)
    e_ref, e_tap = maxerr(y_ref[..., :T], want), maxerr(y_tap[..., :T], want)
    print(f"tap conv {mode} C {C} Fr {Fr} T {T}: gather route {e_ref:.2e}, image route {e_tap:.2e} vs float64 on the rounded operands")
    assert bool(torch.isfinite(y_tap[..., :T]).all())
    assert e_tap < 2e-5 and e_ref < 2e-5 and maxerr(y_tap[..., :T], y_ref[..., :T]) < 1e-5


def test_layernorm_channel_first(lib):
    B, Cn, Tn = 2, 512, 333
    x = rnd(B, Cn, Tn, seed=50) * 3 + 40.0          # large mean: checks the shifted one-pass variance
    w, b, pe = rnd(Cn, seed=51), rnd(Cn, seed=52), rnd(Cn, Tn, seed=53)
    want = F.layer_norm(x.transpose(1, 2), (Cn,), w, b, eps=1e-5).transpose(1, 2) + pe[None]
    y = torch.empty(B, Cn, Tn, device="cuda")
    xd, wd, bd, ped = x.float().cuda(), w.float().cuda(), b.float().cuda(), pe.float().cuda()
    _lib.check(lib.mi_layernorm_cf(xd.data_ptr(), B, Cn, Tn, wd.data_ptr(), bd.data_ptr(), ped.data_ptr(), y.data_ptr(), stream()),
               "mi_layernorm_cf")
    torch.cuda.synchronize()
    assert maxerr(y, want) < 3e-5
