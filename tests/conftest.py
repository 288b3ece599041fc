import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE_ROOT = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ref():
    """The reference itself compiled for the host (oracle/_ref, see oracle/ref.py).  Built on demand where
    the reference checkout exists; on the GPU box the prebuilt binary travels with the snapshot.  Tests that
    need it are skipped only when neither is there."""
    from oracle import ref as _ref
    if os.path.isdir(os.path.join(REFERENCE_ROOT, "csrc")):
        _ref.build()
    if not _ref.available():
        pytest.skip("oracle/_ref not built and %s absent" % REFERENCE_ROOT)
    _ref.lib()
    return _ref


@pytest.fixture()
def python_host_layer():
    """The Python host layer (torch/_plugin.py) serves every call of the test, as where csrc_host/nvdr_torch_host.cpp is not built."""
    from nvdiffrast_amd.torch import _plugin
    _plugin.set_host_layer("python")
    yield
    _plugin.set_host_layer("compiled")


@pytest.fixture(scope="session")
def raw_oracle():
    """The C oracle without the reference cross-check."""
    import oracle as _oracle
    _oracle.build()
    _oracle.lib()
    return _oracle


@pytest.fixture(scope="session")
def oracle(raw_oracle):
    """The CPU oracle (test infrastructure), pinned: every call is also run through the reference's own code
    when oracle/_ref is available and must agree with it (oracle/pinned.py)."""
    from oracle import ref as _ref
    from oracle.pinned import PinnedOracle
    if os.path.isdir(os.path.join(REFERENCE_ROOT, "csrc")):
        _ref.build()
    po = PinnedOracle(raw_oracle)
    pytest._nvdr_pinned = po
    return po


# ---- the parity bars, stated once (DESIGN.md section 2 quotes this block) ------------------------------------------------
# integers (triangle ids, U32 depths): identical.
# forward floats of ONE op on identical inputs (barycentrics, z/w, attributes, texture samples, blended colours): 1e-5 abs.
# gradients of ONE op on identical inputs: 1e-5 * max(1, |g|_inf) -- every gradient on this path is a SUM of per-pixel
#   terms (the reference's with f32 atomics in launch order, the oracle's in f64, this library's in fixed point), and an f32
#   sum cannot be defined better than the ulp of its largest term: position gradients of the benchmark scene reach 1.8e4,
#   where 1e-5 abs would be a hundredth of an ulp.
# a CHAIN of k ops compared end to end: every op is handed its predecessor's output, which is defined only to that
#   op's bar, and passes the difference on through its own Jacobian: k * the single-op gradient bar (CHAIN_OPS = 4 for
#   rasterize -> interpolate -> texture -> antialias), and CHAIN_VALUE_TOL = 2e-5 abs for the colours at the end of the
#   forward chain: an interpolated uv is defined to about 1 ulp (6e-8) by the order of three f32 products, and a texture
#   turns a uv difference into a colour difference of (texels per unit uv at the sampled level) x (difference of
#   neighbouring texels) -- for a random 2048^2 texture sampled at about one texel per pixel that is up to
#   6e-8 x 1024 x 1 = 6e-5 in the worst case and 9e-6 at the worst pixel measured.
# gradients that sum MANY thousands of per-pixel terms per element, compared with the f64-summing oracle at batch scale (32 x 512^2
#   and up; tests/test_gpu_work_order.py): BIG_SUM_FACTOR = 2 times the single-op bar, i.e. 2e-5 of the tensor's magnitude -- the
#   bracket oracle/pinned.py pins the oracle to the reference's own f32-atomic sums with.  Every per-pixel term carries its own f32
#   rounding (1e-7 of the term); thousands of them with mixed signs leave 1e-5 of the LARGEST element on an element that is itself a
#   small difference of large terms (measured: one element of 48 k at 1.04 of the single-op bar).
# texture gradients w.r.t. uv / uv_da are compared WHERE THE REFERENCE FUNCTION IS CONTINUOUS: bilinear sampling has a discontinuous uv
#   gradient at texel boundaries, mip selection a discontinuous level gradient where the footprint crosses a level or the clamp; a
#   pixel whose uv lies within one ulp -- or whose footprint uv_da within DA_NUDGE = 4e-6 relative (the level is a logarithm of a
#   sum of squares of uv_da, evaluated in f32 with v_log_f32 here and in f64 by the oracle: ~1e-6 absolute on the level) -- of such
#   a place gets either side from either implementation.  `discontinuous_pixels` finds them from the ORACLE ALONE -- pixels whose
#   oracle gradient moves by more than the bar when the inputs are nudged by that much -- independent of the implementation under
#   test; they are excluded BY THAT NAMED
#   CRITERION (a few in eight million), and their number is bounded by the tests.  No other element of any tensor is exempted.
ATOL = 1e-5
CHAIN_OPS = 4
CHAIN_VALUE_TOL = 2e-5
BIG_SUM_FACTOR = 2
DA_NUDGE = 4e-6


def grad_tol(g, ops=1):
    import numpy as np
    return ops * ATOL * max(1.0, float(np.abs(g).max()))


_MARGINS = {}


def discontinuous_pixels(oracle, tex, uv, g_col, uv_da, kw, patterns=4, seed=0):
    """[N,H,W] bool: pixels at which the ORACLE's own texture gradient w.r.t. uv / uv_da changes by more than the single-op bar
    when uv is moved by one ulp and uv_da by DA_NUDGE relative (each component up or down, `patterns` random sign patterns and the
    two uniform ones): the reference function is discontinuous within rounding distance of the input there (see the bars above)."""
    import numpy as np
    base = oracle.texture_grad(tex, uv, g_col, uv_da, **kw)
    tol_uv, tol_da = grad_tol(base["uv"]), grad_tol(base["uv_da"])
    bad = np.zeros(uv.shape[:3], bool)
    rng = np.random.default_rng(seed)
    signs = [(np.ones(uv.shape[-1]), np.ones(uv_da.shape[-1])), (-np.ones(uv.shape[-1]), -np.ones(uv_da.shape[-1]))]
    signs += [(rng.choice([-1.0, 1.0], size=uv.shape[-1]), rng.choice([-1.0, 1.0], size=uv_da.shape[-1])) for _ in range(patterns)]
    for su, sd in signs:
        uv2 = np.nextafter(uv, (su * np.inf).astype(np.float32)).astype(np.float32)
        da2 = (uv_da * (1.0 + sd * DA_NUDGE).astype(np.float32)).astype(np.float32)
        g2 = oracle.texture_grad(tex, uv2, g_col, da2, **kw)
        bad |= (np.abs(g2["uv"] - base["uv"]) > tol_uv).any(-1) | (np.abs(g2["uv_da"] - base["uv_da"]) > tol_da).any(-1)
    return bad


def within(name, got, want, tol, where=None):
    """Assert |got - want| <= tol everywhere (`where`: a boolean mask over the leading dimensions selecting the elements that are
    compared -- only ever the named criterion `discontinuous_pixels`) and remember the margin: the GPU run's summary lists, per
    named check, the worst error as a fraction of its tolerance."""
    import numpy as np
    frac = 0.0
    d = np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64))
    if where is not None:
        d = d[where]
    worst = float(d.max(initial=0.0)) / tol
    bad = float((d > tol).mean()) if d.size else 0.0
    rec = _MARGINS.setdefault(name, [0, 0.0, 0.0])
    rec[0] += 1
    rec[1] = max(rec[1], worst)
    rec[2] = max(rec[2], bad)
    assert bad <= frac, (name, "fraction above tolerance", bad, "worst error / tolerance", worst)


def pytest_terminal_summary(terminalreporter):
    """How much of the run was checked against the reference itself."""
    if _MARGINS:
        terminalreporter.write_line("parity margins (checks, worst error / tolerance, worst fraction of elements above it): " +
                                    ", ".join("%s x%d %.2f %.1e" % (k, v[0], v[1], v[2]) for k, v in sorted(_MARGINS.items())))
    try:
        from oracle.pinned import PinnedOracle  # noqa: F401
    except Exception:  # noqa: BLE001
        return
    po = getattr(pytest, "_nvdr_pinned", None)
    if po is not None and po.stats:
        terminalreporter.write_line("oracle pinned to the reference (%s): " % ("enabled" if po.enabled else "DISABLED") +
                                    ", ".join("%s x%d (worst %.2f of tol)" % (k, v[0], v[1]) for k, v in sorted(po.stats.items())))


@pytest.fixture(scope="session")
def dr():
    """The product API on cuda:0; fails loudly when the GPU or the HIP library is missing."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test without a GPU"
    from nvdiffrast_amd import _capi
    _capi.load()
    import nvdiffrast_amd.torch as dr_
    return dr_
