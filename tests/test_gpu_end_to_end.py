"""BASELINE config 5 stand-in: the four ops inside an optimisation loop (samples/fit_texture_synth.py).
Numeric parity of every op is covered elsewhere; this catches gradient-quality regressions that
per-op parity would miss (wrong sign, a gradient routed to the wrong tensor, a stale work buffer)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_texture_fit_converges(dr):
    spec = importlib.util.spec_from_file_location("fit_texture_synth", os.path.join(ROOT, "samples", "fit_texture_synth.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    r = mod.fit(iters=60, res=128, ref_res=256, tex_size=128, seed=1, lr=3e-2)
    assert r["loss_last"] < 0.25 * r["loss_first"], r
    assert r["tex_rmse_after"] < 0.8 * r["tex_rmse_before"], r
