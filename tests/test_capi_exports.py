"""CPU-side checks of the drop-in boundary: the library builds for gfx950 without a GPU, loads,
and exports exactly the entry points include/nvdr_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nvdr_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nvdr_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from nvdiffrast_amd import _build, _capi
    _build.build()                                  # hipcc cross-compiles for gfx950 here
    return _capi.load()


def test_header_symbols_are_all_exported(lib):
    names = _declared()
    assert len(names) >= 9
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/nvdr_hip.h but not exported"


def test_binding_table_matches_header():
    from nvdiffrast_amd import _capi
    assert sorted(_capi.SIGNATURES) == _declared()


def test_abi_version_and_error_string(lib):
    assert lib.nvdr_abi_version() >= 1
    assert isinstance(lib.nvdr_last_error(), bytes)


def test_scratch_query_is_pure_host_code(lib):
    # Runs without a GPU: pure arithmetic on the host.
    small = lib.nvdr_rasterize_scratch_bytes(1, 100, 64, 64)
    big = lib.nvdr_rasterize_scratch_bytes(64, 10000, 512, 512)
    assert 0 < small < big
    assert lib.nvdr_rasterize_scratch_bytes(0, 100, 64, 64) == 0


def test_bad_arguments_are_rejected_before_any_launch(lib):
    # Null pointers / empty shapes must come back as NVDR_ERR_ARG with a message, not crash.
    rc = lib.nvdr_rasterize_fwd(None, None, None, 1, 1, 3, 1, 1, 8, 8, None, None, None, 0, None, None, None)
    assert rc == 1
    assert b"null pointer" in lib.nvdr_last_error()
    rc = lib.nvdr_interpolate_fwd(None, None, None, None, 1, 1, 1, 3, 4, 1, 8, 8, 0, None, 0, None, None, None)
    assert rc == 1


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "nvdiffrast_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "nvdr_oracle" not in src, f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from nvdiffrast_amd import _capi
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "lib_path", lambda: str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no fallback"):
        _capi.load()


def test_cpu_tensors_are_rejected_like_the_reference():
    import torch
    from nvdiffrast_amd.torch import _plugin
    pos = torch.zeros(1, 3, 4)
    tri = torch.zeros(1, 3, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="must reside on the same GPU device"):
        _plugin.interpolate_fwd(torch.zeros(1, 3, 2), torch.zeros(1, 4, 4, 4), tri)
    with pytest.raises(RuntimeError, match="must reside on the same GPU device"):
        _plugin.rasterize_grad(pos, tri, torch.zeros(1, 4, 4, 4), torch.zeros(1, 4, 4, 4))
