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
ATOL = 1e-5
CHAIN_OPS = 4
CHAIN_VALUE_TOL = 2e-5


def grad_tol(g, ops=1):
    import numpy as np
    return ops * ATOL * max(1.0, float(np.abs(g).max()))


_MARGINS = {}


def within(name, got, want, tol, frac=0.0):
    """Assert |got - want| <= tol everywhere (or everywhere but a fraction `frac` of the elements) and remember the
    margin: the GPU run's summary lists, per named check, the worst error as a fraction of its tolerance."""
    import numpy as np
    d = np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64))
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
