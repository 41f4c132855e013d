"""SURVEY 8 f2 on the GPU: the checkpoint packages the REFERENCE wrote (tests/golden/pkg_*.th: `serialize_model` +
`torch.save` around reference models built with every keyword of conf/config.yaml, tools/make_golden.py package_fixture) go
through `demucs_amd.states.load_model` into the engine, whose forward must equal the forward of the reference model
re-loaded from the same package (stored samples, float64 truth and float32).  Tolerance: north_star's 1e-4 max-abs."""
import os
import warnings

import pytest
import torch

from conftest import GOLDEN
from demucs_amd import states
from demucs_amd.synth import synth_mix

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("name,seed,kind", [("pkg_htdemucs", 31, "tones"), ("pkg_hdemucs", 32, "noise")])
def test_reference_written_package_runs_on_the_engine(golden, name, seed, kind):
    g = golden(name)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        model = states.load_model(os.path.join(GOLDEN, name + ".th"), max_batch=1)
    assert not w, [str(x.message) for x in w]
    model.to("cuda").eval()
    mix = torch.from_numpy(synth_mix(seed, int(g.meta("length")), kind))[None].cuda()
    out = model(mix)
    e64 = g.check("f64", "out", out, atol=TOL)
    e32 = g.check("f32", "out", out, atol=TOL)
    print(f"{name}: engine forward of the reference-written package, max-abs vs reference f64 {e64:.2e}, f32 {e32:.2e}")
