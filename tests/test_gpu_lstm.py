"""The LSTM recurrence of hdemucs_mmi's BLSTM blocks (demucs/demucs.py:20-67 -> nn.LSTM: zero initial state, gates i, f, g, o)
through the C ABI (`mi_lstm_seq`): the persistent kernel the engine uses (lstm.hip: one launch for all time steps, hidden state
exchanged between workgroups as tagged granules) and the one-launch-per-step chain, against torch's own nn.LSTM in float64 fed
with the same pre-activations -- at the sizes the 44-second chunks produce (95 / 50 sequences of 200 steps), at ragged sizes
(sequence counts that are not multiples of 16, one step, several launches of sequence tiles) and bit for bit against each other."""
import ctypes as C

import numpy as np
import pytest
import torch

from demucs_amd import _lib

pytestmark = pytest.mark.gpu


def reference(gx, whh, H):
    """float64 recurrence: gx (N, 2, 4H, W), whh (2, 4H, H) -> (N, 2H, W)."""
    N, _, _, W = gx.shape
    g64, w64 = gx.double(), whh.double()
    out = torch.zeros(N, 2 * H, W, dtype=torch.float64)
    for d in range(2):
        h = torch.zeros(N, H, dtype=torch.float64)
        c = torch.zeros(N, H, dtype=torch.float64)
        for s in range(W):
            t = W - 1 - s if d else s
            a = g64[:, d, :, t] + h @ w64[d].t()
            i, f, g, o = a[:, :H], a[:, H:2 * H], a[:, 2 * H:3 * H], a[:, 3 * H:]
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            out[:, d * H:(d + 1) * H, t] = h
    return out


def run(gx, whh, H, mode):
    N, W = gx.shape[0], gx.shape[3]
    gxd = gx.cuda().contiguous()
    out = torch.full((N, 2 * H, W), float("nan"), device="cuda")
    wh = np.ascontiguousarray(whh.numpy(), dtype=np.float32)
    _lib.check(_lib.load().mi_lstm_seq(gxd.data_ptr(), wh.ctypes.data, N, H, W, out.data_ptr(), mode,
                                       C.c_void_p(_lib.current_stream_ptr())), "mi_lstm_seq")
    return out.cpu()


@pytest.mark.parametrize("H,N,W", [(192, 95, 200), (384, 50, 200), (192, 7, 200), (384, 4, 200), (192, 1, 1), (384, 17, 3),
                                    (192, 33, 37), (384, 150, 24), (192, 300, 16)])
def test_persistent_recurrence_matches_float64_and_the_step_chain(H, N, W):
    gen = torch.Generator().manual_seed(H + 7 * N + W)
    gx = torch.randn(N, 2, 4 * H, W, generator=gen)
    whh = torch.randn(2, 4 * H, H, generator=gen) * (1.5 / H ** 0.5)         # spectral radius ~1: the state really recurs
    want = reference(gx, whh, H)
    fast = run(gx, whh, H, 1)
    slow = run(gx, whh, H, 0)
    assert bool(torch.isfinite(fast).all())
    err = float((fast.double() - want).abs().max())
    print(f"lstm H={H} N={N} W={W}: persistent kernel vs float64 {err:.2e}; vs the step chain "
          f"{float((fast - slow).abs().max()):.2e}")
    assert err <= 2e-5 and float((slow.double() - want).abs().max()) <= 2e-5
    assert torch.equal(fast, slow)              # same products, same summation order: bit-identical


def test_back_to_back_calls_reuse_the_granule_buffers():
    """Tags restart at 1 in every launch: the granule buffers are zeroed per launch, so a second sequence on the same stream must
    not see the first one's granules as valid."""
    H, N, W = 192, 20, 50
    gen = torch.Generator().manual_seed(3)
    whh = torch.randn(2, 4 * H, H, generator=gen) * (1.5 / H ** 0.5)
    a = torch.randn(N, 2, 4 * H, W, generator=gen)
    b = torch.randn(N, 2, 4 * H, W, generator=gen)
    ra, rb = run(a, whh, H, 1), run(b, whh, H, 1)
    assert torch.equal(ra, run(a, whh, H, 0)) and torch.equal(rb, run(b, whh, H, 0))
