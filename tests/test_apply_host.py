"""Host logic of `demucs_amd.apply` on CPU with a cheap stand-in model (generic per-segment
route): must equal the oracle's restatement of the reference scheduler bit for bit, emit the
same callback events, and keep the reference's error behaviour."""
import random
from fractions import Fraction

import pytest
import torch

from demucs_amd import apply as P
from oracle import apply_oracle as A


class ToyModel:
    """Deterministic, position-dependent, non-linear stand-in with the attributes apply_model needs."""
    sources = ["a", "b", "c"]
    samplerate = 100
    audio_channels = 2
    segment = Fraction(4, 1)            # 400 samples
    rng_draws_per_forward = 1           # like the reference's HTDemucs (transformer.py:680); read by the sharded scheduler

    def __init__(self, gain=1.0):
        self.gain = gain
        self.calls = 0

    segment_length = 400

    def valid_length(self, length):
        if length > 400:
            raise ValueError(f"Given length {length} is longer than training length 400")
        return 400

    def to(self, device):
        return self

    def eval(self):
        return self

    def parameters(self):
        yield torch.empty(0)

    def __call__(self, mix):
        random.randrange(1)
        self.calls += 1
        ramp = torch.linspace(0.5, 1.5, mix.shape[-1])
        outs = [torch.tanh(self.gain * (k + 1) * mix * ramp) + 0.01 * k for k in range(3)]
        return torch.stack(outs, 1)


class RaggedToy:
    """Stand-in WITHOUT `valid_length` / `segment_length`, like the reference's HDemucs: the leaf forwards every chunk at its
    own length, unpadded (demucs/apply.py:309-310); no RNG draw per forward."""
    sources = ["a", "b", "c"]
    samplerate = 100
    audio_channels = 2
    segment = Fraction(4, 1)

    def to(self, device):
        return self

    def eval(self):
        return self

    def parameters(self):
        yield torch.empty(0)

    def __call__(self, mix):
        n = mix.shape[-1]
        ramp = torch.linspace(0.5, 1.5, n) * (1.0 + 0.001 * n)          # depends on the chunk length it was given
        return torch.stack([torch.tanh((k + 1) * mix * ramp) + 0.01 * k for k in range(3)], 1)


def both(model_p, model_o, mix, rseed=None, **kw):
    ev_p, ev_o = [], []
    if rseed is not None:
        random.seed(rseed)
    out_p = P.apply_model(model_p, mix, callback=lambda d: ev_p.append(dict(d)), **kw)
    if rseed is not None:
        random.seed(rseed)
    kw.pop("num_workers", None)
    out_o = A.apply_model(model_o, mix, callback=lambda d: ev_o.append(dict(d)), **kw)
    return out_p, out_o, ev_p, ev_o


@pytest.mark.parametrize("length", [400, 401, 399, 1000, 37, 920])
@pytest.mark.parametrize("overlap,tp", [(0.25, 1.0), (0.1, 2.0), (0.6, 1.0)])
def test_split_matches_oracle_bit_exact(length, overlap, tp):
    mix = torch.randn(2, 2, length, generator=torch.Generator().manual_seed(length))
    mix0 = mix.clone()
    out_p, out_o, ev_p, ev_o = both(ToyModel(), ToyModel(), mix, shifts=0, split=True, overlap=overlap, transition_power=tp)
    assert torch.equal(mix, mix0)
    assert out_p.shape == (2, 3, 2, length) and torch.equal(out_p, out_o)
    assert ev_p == ev_o and len(ev_p) == 2 * len(range(0, length, int((1 - overlap) * 400)))


@pytest.mark.parametrize("length,kw", [(1000, {}), (401, {}), (37, {}), (920, dict(segment=2.5)), (777, dict(shifts=2))])
def test_model_without_valid_length_matches_oracle_bit_exact(length, kw):
    mix = torch.randn(2, 2, length, generator=torch.Generator().manual_seed(length))
    args = dict(dict(shifts=0, split=True, overlap=0.25), **kw)
    out_p, out_o, ev_p, ev_o = both(RaggedToy(), RaggedToy(), mix, rseed=3, **args)
    assert out_p.shape == (2, 3, 2, length) and torch.equal(out_p, out_o) and ev_p == ev_o


def test_shifts_and_bag_match_oracle_bit_exact():
    mix = torch.randn(1, 2, 777, generator=torch.Generator().manual_seed(1))
    out_p, out_o, ev_p, ev_o = both(ToyModel(), ToyModel(), mix, rseed=5, shifts=3, split=True, overlap=0.25)
    assert torch.equal(out_p, out_o) and ev_p == ev_o
    w = [[1.0, 0.0, 0.5], [0.0, 1.0, 1.5]]
    bag_p = P.BagOfModels([ToyModel(1.0), ToyModel(0.7)], w)
    bag_o = A.Bag([ToyModel(1.0), ToyModel(0.7)], w)
    out_p, out_o, ev_p, ev_o = both(bag_p, bag_o, mix, rseed=9, shifts=1, split=True, overlap=0.25)
    assert torch.equal(out_p, out_o) and ev_p == ev_o
    assert {e["model_idx_in_bag"] for e in ev_p} == {0, 1} and all(e["models"] == 2 for e in ev_p)
    with pytest.raises(NotImplementedError):
        bag_p(mix)


def test_no_split_and_tensor_chunk_input():
    mix = torch.randn(1, 2, 900, generator=torch.Generator().manual_seed(2))
    out_p, out_o, _, _ = both(ToyModel(), ToyModel(), mix[..., :300], shifts=0, split=False)
    assert torch.equal(out_p, out_o) and out_p.shape[-1] == 300
    chunk_p, chunk_o = P.TensorChunk(mix, 100, 650), A.Window(mix, 100, 650)
    out_p = P.apply_model(ToyModel(), chunk_p, shifts=0, split=True)
    out_o = A.apply_model(ToyModel(), chunk_o, shifts=0, split=True)
    assert torch.equal(out_p, out_o) and out_p.shape[-1] == 650


def test_thread_pool_matches_sequential():
    mix = torch.randn(1, 2, 2000, generator=torch.Generator().manual_seed(3))
    seq = P.apply_model(ToyModel(), mix, shifts=0, split=True)
    par = P.apply_model(ToyModel(), mix, shifts=0, split=True, num_workers=3)
    assert torch.equal(seq, par)


def test_tensor_chunk_semantics():
    t = torch.arange(20.0).view(1, 1, 20)
    c = P.TensorChunk(t, 12, 6)
    assert c.shape == [1, 1, 6] and c.padded(10).flatten().tolist() == list(range(10, 20))
    c = P.TensorChunk(t, 15, 100)
    assert c.length == 5 and c.padded(12).flatten().tolist() == [12, 13, 14, 15, 16, 17, 18, 19, 0, 0, 0, 0]
    inner = P.TensorChunk(P.TensorChunk(t, 4, 10), 2, 3)
    assert inner.offset == 6 and inner.tensor is t and inner.padded(5).flatten().tolist() == [5, 6, 7, 8, 9]
    assert P.tensor_chunk(inner) is inner and isinstance(P.tensor_chunk(t), P.TensorChunk)
    with pytest.raises(AssertionError):
        P.TensorChunk(t, 20)
    with pytest.raises(AssertionError):
        c.padded(3)


def test_center_trim_and_dummy_pool():
    x = torch.arange(10.0)
    assert P.center_trim(x, 7).tolist() == [1, 2, 3, 4, 5, 6, 7]
    assert P.center_trim(x, torch.zeros(4)).tolist() == [3, 4, 5, 6]
    assert P.center_trim(x, 10) is x
    with pytest.raises(ValueError):
        P.center_trim(x, 11)
    pool = P.DummyPoolExecutor()
    fut = pool.submit(lambda a, b=1: a + b, 2, b=5)
    assert fut.result() == 7
    pool.shutdown()
    from concurrent.futures import CancelledError
    with pytest.raises(CancelledError):
        fut.result()


def test_error_behaviour():
    m = ToyModel()
    with pytest.raises(AssertionError):
        P.apply_model(m, torch.zeros(1, 2, 100), shifts=0, transition_power=0.5)        # apply.py:235
    with pytest.raises(ValueError):
        P.apply_model(m, torch.zeros(1, 2, 500), shifts=0, split=False)                 # longer than the training length

    def boom(d):
        if d["state"] == "end" and d["segment_offset"] > 0:
            raise KeyboardInterrupt("stop")                                             # docs/api.md: abort via callback
    with pytest.raises(KeyboardInterrupt):
        P.apply_model(m, torch.zeros(1, 2, 1000), shifts=0, callback=boom)

    class Broken(ToyModel):
        def __call__(self, mix):
            raise RuntimeError("segment failed")
    with pytest.raises(RuntimeError, match="segment failed"):                           # apply.py:289-293
        P.apply_model(Broken(), torch.zeros(1, 2, 1000), shifts=0, num_workers=2)
    with pytest.raises(AssertionError):
        P.BagOfModels([ToyModel()], [[1.0, 2.0]])


def test_engine_model_has_no_cpu_path():
    from demucs_amd.htdemucs import HTDemucs
    from demucs_amd.weights import HTDemucsConfig, synthetic_state_dict
    cfg = HTDemucsConfig()
    m = HTDemucs(cfg.sources)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 2, 343980))                 # no weights
    m.load_state_dict(synthetic_state_dict(cfg, 0))
    with pytest.raises(RuntimeError, match="no CPU"):
        P.apply_model(m, torch.zeros(1, 2, 1000), shifts=0, device="cpu")
    assert set(m.state_dict()) == set(synthetic_state_dict(cfg, 0))
    with pytest.raises(RuntimeError):
        m.load_state_dict({"encoder.0.conv.weight": torch.zeros(1)})
    with pytest.raises(ValueError):
        HTDemucs(cfg.sources, depth=6)


def test_separator_tensor_contract():
    """api.py:265-291: normalise by the mono mean/std in place, separate, restore the input."""
    from demucs_amd.api import LoadModelError, Separator
    wav = torch.randn(2, 900, generator=torch.Generator().manual_seed(8)) * 3 + 0.5
    wav0 = wav.clone()
    events = []
    sep = Separator(ToyModel(), device="cpu", shifts=0, callback=lambda d: events.append(d), callback_arg={"tag": 1})
    got_wav, stems = sep.separate_tensor(wav)
    assert got_wav is wav and torch.allclose(wav, wav0, atol=1e-6)
    assert list(stems) == ["a", "b", "c"] and stems["a"].shape == (2, 900)
    ref = wav0.mean(0)
    norm = (wav0 - ref.mean()) / (ref.std() + 1e-8)
    want = A.apply_model(ToyModel(), norm[None], shifts=0) * (ref.std() + 1e-8) + ref.mean()
    assert torch.allclose(torch.stack(list(stems.values())), want[0], atol=1e-6)
    assert events and events[0]["audio_length"] == 900 and events[0]["tag"] == 1
    assert sep.samplerate == 100 and sep.audio_channels == 2 and sep.model is not None
    sep.update_parameter(segment=None, shifts=1)
    with pytest.raises(ValueError):
        sep.update_parameter(segment=0)
    from demucs_amd._lib import EngineError
    with pytest.raises(EngineError):                      # resampling is a HIP kernel: no CPU implementation to fall back to
        sep.separate_tensor(wav, sr=48000)
    with pytest.raises(LoadModelError):
        Separator("htdemucs")


def test_device_scheduler_refuses_oversized_windows_before_any_launch():
    """`device_split_accumulate` sizes its segment buffer for `model.segment_length`: a padded window longer than that
    (a segment override above the training length, demucs/apply.py:305-310 -> htdemucs.py:531-533 raises there too) or a
    weight ramp shorter than a segment must be refused on the host, before the library is even loaded -- the gather
    kernel would otherwise write past the buffer.  Runs without a GPU: the checks come first."""
    class Stub:
        segment_length, max_batch, samplerate, sources = 1000, 4, 44100, ["a"]

    base, acc, w = torch.zeros(2, 5000), torch.zeros(2, 5000), torch.ones(1000)
    with pytest.raises(ValueError, match="longer than training length"):
        P.device_split_accumulate(Stub(), base, 0, 5000, [0, 750], 1000, 1200, w, acc, 0)
    with pytest.raises(ValueError, match="does not fit"):
        P.device_split_accumulate(Stub(), base, 0, 5000, [0, 750], 1000, 900, w, acc, 0)
    with pytest.raises(ValueError, match="does not fit"):
        P.device_split_accumulate(Stub(), base, 0, 5000, [0, 750], 1000, 1000, torch.ones(10), acc, 0)
    with pytest.raises(ValueError, match="weight ramp"):
        P.ragged_split_accumulate(Stub(), base, 0, 5000, [0, 750], 1000, torch.ones(10), acc)
