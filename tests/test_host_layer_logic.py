"""The compiled host layer's bookkeeping and autograd nodes, exercised where there is no GPU.

csrc_host/nvdr_torch_host.cpp is built a second time (into a temporary directory, -DNVDR_HOST_TEST_BUILD: CPU tensors, no
stream) and bound to tests/host_stub/nvdr_stub.c, an implementation of the eleven C-ABI entry points it calls on top of the CPU
oracle.  The public functions of ops.py then run end to end on the CPU, and what is checked is the HOST logic:

  * forward values and every gradient equal the oracle's chain rule (fused and separate backward, with and without pixel
    differentials, instanced and range mode, broadcast attributes);
  * the share of the position gradient prepared by interpolate's backward is ADDED to what other consumers of rast contribute;
  * anything that could observe rast's gradient (hook, retain_grad, autograd.grad capture) switches to the separate kernels and
    sees the reference's values;
  * tile-flag records die with the tensor's version / storage; stale state from an interrupted backward pass is not used;
  * calls the layer must decline come back as None (the Python layer then words the error).
The GPU suite runs the same public functions on the shipped module (tests/test_gpu_host_layer.py)."""
import ctypes
import os
import shutil
import subprocess
import sys
import sysconfig

import numpy as np
import pytest
import torch

import oracle
from nvdiffrast_amd.utils import m10k_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SYMBOLS = ("nvdr_last_error", "nvdr_get_option", "nvdr_log", "nvdr_rasterize_scratch_bytes_pool", "nvdr_rasterize_pool_peak_offset",
           "nvdr_tile_flags_bytes", "nvdr_rasterize_fwd", "nvdr_rasterize_grad", "nvdr_interpolate_fwd", "nvdr_interpolate_grad",
           "nvdr_interpolate_rasterize_grad", "nvdr_texture_mip_info", "nvdr_texture_construct_mip", "nvdr_texture_fwd",
           "nvdr_texture_grad", "nvdr_texture_grad_scratch_bytes", "nvdr_antialias_fwd", "nvdr_antialias_grad")
C_RAST_FWD, C_RAST_GRAD, C_INTERP_FWD, C_INTERP_GRAD, C_FUSED, C_CLEAN = range(6)


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    """(module, stub library): the test build of the host layer bound to the oracle-backed stub."""
    oracle.build()
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("g++ not available")
    from torch.utils import cpp_extension
    d = str(tmp_path_factory.mktemp("hostbuild"))
    stub = os.path.join(d, "libnvdr_stub.so")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-shared", "-fPIC", os.path.join(ROOT, "tests", "host_stub", "nvdr_stub.c"), "-o", stub,
                           "-L" + os.path.join(ROOT, "oracle"), "-l:libnvdr_oracle.so", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    mod = os.path.join(d, "_nvdr_host_cputest.so")
    cmd = [cxx, "-O1", "-std=c++17", "-fPIC", "-shared", "-DNVDR_HOST_TEST_BUILD=1", "-DTORCH_EXTENSION_NAME=_nvdr_host_cputest",
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    cmd += ["-I" + p for p in cpp_extension.include_paths()] + ["-I" + sysconfig.get_paths()["include"]]
    cmd += [os.path.join(ROOT, "nvdiffrast_amd", "csrc_host", "nvdr_torch_host.cpp"), "-o", mod,
            "-L" + tlib, "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python", "-Wl,-rpath," + tlib]
    subprocess.check_call(cmd)
    sys.path.insert(0, d)
    try:
        import _nvdr_host_cputest as m
    finally:
        sys.path.remove(d)
    lib = ctypes.CDLL(stub)
    lib.nvdr_stub_calls.restype = ctypes.c_int
    m.init({n: ctypes.cast(getattr(lib, n), ctypes.c_void_p).value for n in SYMBOLS})
    return m, lib


@pytest.fixture()
def dr(host, monkeypatch):
    """nvdiffrast_amd.torch with the test build installed as its compiled host layer."""
    import nvdiffrast_amd.torch as dr
    from nvdiffrast_amd.torch import _plugin
    m, _ = host
    monkeypatch.setitem(_plugin._host_state, "mod", m)
    monkeypatch.setitem(_plugin._host_state, "enabled", True)
    m.set_fused(True); m.set_skip(True); m.set_verify(False)
    return dr


class _Ctx:
    """A RasterizeCudaContext without the GPU its constructor asks for."""

    def __new__(cls, dr):
        from nvdiffrast_amd.torch import _plugin
        c = object.__new__(dr.RasterizeCudaContext)
        c.cpp_wrapper = _plugin.RasterizeCRStateWrapper(0)
        c.active_depth_peeler = None
        return c


def _scene(n=2, seed=3, res=(32, 40)):
    b = m10k_batch(n, seed=seed, nx=6, ny=4)
    rng = np.random.default_rng(seed)
    G = rng.normal(size=(n,) + res + (b["attr"].shape[-1],)).astype(np.float32)
    return b, res, G


def _t(a, grad=False):
    return torch.from_numpy(np.ascontiguousarray(a)).requires_grad_(grad)


def _close(a, b, what):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else a
    tol = 2e-5 * max(1.0, float(np.abs(b).max()))
    assert a.shape == b.shape and np.abs(a - b).max() <= tol, "%s: %.3g > %.3g" % (what, np.abs(a - b).max(), tol)


def _oracle_chain(b, res, G, extra_rast_grad=None):
    ro, rdbo = oracle.rasterize(b["pos"], b["tri"], res)
    oo, _ = oracle.interpolate(b["attr"], ro, b["tri"])
    ga, gr, _ = oracle.interpolate_grad(b["attr"], ro, b["tri"], G)
    if extra_rast_grad is not None:
        gr = gr + extra_rast_grad
    gp = oracle.rasterize_grad(b["pos"], b["tri"], ro, gr)
    return ro, rdbo, oo, ga, gr, gp


def test_forward_and_fused_backward_equal_the_oracle(host, dr):
    m, lib = host
    b, res, G = _scene()
    ctx = _Ctx(dr)
    pos, attr, tri = _t(b["pos"], True), _t(b["attr"], True), _t(b["tri"])
    c0 = m.counters()
    f0 = lib.nvdr_stub_calls(C_FUSED)
    rast, rast_db = dr.rasterize(ctx, pos, tri, res)
    out, out_da = dr.interpolate(attr, rast, tri)
    assert rast.grad_fn.name() == "NvdrRasterizeBackward" and out.grad_fn.name() == "NvdrInterpolateBackward"
    assert out_da.shape == (2,) + res + (0,)
    torch.autograd.backward(out, _t(G))
    ro, rdbo, oo, ga, gr, gp = _oracle_chain(b, res, G)
    assert (rast.detach().numpy()[..., 3] != ro[..., 3]).sum() == 0
    _close(rast, ro, "rast"); _close(rast_db, rdbo, "rast_db"); _close(out, oo, "out")
    _close(attr.grad, ga, "g_attr"); _close(pos.grad, gp, "g_pos")
    c1 = m.counters()
    assert c1["fused"] == c0["fused"] + 1 and c1["fused_alone"] == c0["fused_alone"] + 1 and c1["separate"] == c0["separate"]
    assert lib.nvdr_stub_calls(C_FUSED) == f0 + 1
    assert c1["fast_forward"] == c0["fast_forward"] + 2


def test_fused_off_runs_the_two_kernels(host, dr):
    m, lib = host
    from nvdiffrast_amd.torch import _plugin
    b, res, G = _scene(seed=5)
    ctx = _Ctx(dr)
    pos, attr, tri = _t(b["pos"], True), _t(b["attr"], True), _t(b["tri"])
    _plugin.set_fused_backward("off")
    try:
        i0, r0, f0 = lib.nvdr_stub_calls(C_INTERP_GRAD), lib.nvdr_stub_calls(C_RAST_GRAD), lib.nvdr_stub_calls(C_FUSED)
        rast, _ = dr.rasterize(ctx, pos, tri, res)
        out, _ = dr.interpolate(attr, rast, tri)
        torch.autograd.backward(out, _t(G))
        assert (lib.nvdr_stub_calls(C_INTERP_GRAD), lib.nvdr_stub_calls(C_RAST_GRAD), lib.nvdr_stub_calls(C_FUSED)) == (i0 + 1, r0 + 1, f0)
    finally:
        _plugin.set_fused_backward("auto")
    _, _, _, ga, _, gp = _oracle_chain(b, res, G)
    _close(attr.grad, ga, "g_attr"); _close(pos.grad, gp, "g_pos")


def test_other_consumers_of_rast_are_added_to_the_prepared_share(host, dr):
    """A mask made from rast's barycentrics next to interpolate: rast's gradient has two contributors.  The compiled layer keeps
    interpolate's prepared share and adds rasterize_grad of the OTHER contribution (linearity) -- the sum is the reference's."""
    m, lib = host
    b, res, G = _scene(seed=7)
    ctx = _Ctx(dr)
    pos, attr, tri = _t(b["pos"], True), _t(b["attr"], True), _t(b["tri"])
    rng = np.random.default_rng(0)
    Wm = rng.normal(size=(2,) + res + (4,)).astype(np.float32)
    c0 = m.counters()
    rast, _ = dr.rasterize(ctx, pos, tri, res)
    out, _ = dr.interpolate(attr, rast, tri)
    loss = (out * _t(G)).sum() + (rast * _t(Wm)).sum()
    loss.backward()
    _, _, _, ga, _, gp = _oracle_chain(b, res, G, extra_rast_grad=Wm)
    _close(attr.grad, ga, "g_attr"); _close(pos.grad, gp, "g_pos")
    c1 = m.counters()
    assert c1["fused"] == c0["fused"] + 1 and c1["fused_plus"] == c0["fused_plus"] + 1 and c1["fused_alone"] == c0["fused_alone"]


@pytest.mark.parametrize("how", ["hook", "retain_grad", "autograd_grad", "node_prehook"])
def test_whoever_looks_at_rasts_gradient_sees_the_reference_values(host, dr, how):
    m, lib = host
    b, res, G = _scene(seed=11)
    ctx = _Ctx(dr)
    pos, attr, tri = _t(b["pos"], True), _t(b["attr"], True), _t(b["tri"])
    rast, _ = dr.rasterize(ctx, pos, tri, res)
    out, _ = dr.interpolate(attr, rast, tri)
    ro, _, _, ga, gr, gp = _oracle_chain(b, res, G)
    seen = []
    c0 = m.counters()
    if how == "hook":
        rast.register_hook(lambda g: seen.append(g.clone()))
        torch.autograd.backward(out, _t(G))
    elif how == "retain_grad":
        rast.retain_grad()
        torch.autograd.backward(out, _t(G))
        seen.append(rast.grad)
    elif how == "node_prehook":
        rast.grad_fn.register_prehook(lambda gs: seen.append(gs[0].clone()))
        torch.autograd.backward(out, _t(G))
    else:
        g_rast, g_pos, g_attr = torch.autograd.grad(out, [rast, pos, attr], _t(G))
        seen.append(g_rast)
        _close(g_pos, gp, "g_pos"); _close(g_attr, ga, "g_attr")
    assert m.counters()["fused"] == c0["fused"]              # nothing was prepared
    _close(seen[0], gr, "g_rast as seen by " + how)
    if how != "autograd_grad":
        _close(pos.grad, gp, "g_pos"); _close(attr.grad, ga, "g_attr")


def test_backward_with_inputs_still_fuses_and_partial_passes_do_not_leak(host, dr):
    m, lib = host
    b, res, G = _scene(seed=13)
    ctx = _Ctx(dr)
    pos, attr, tri = _t(b["pos"], True), _t(b["attr"], True), _t(b["tri"])
    rast, _ = dr.rasterize(ctx, pos, tri, res)
    out, _ = dr.interpolate(attr, rast, tri)
    _, _, _, ga, gr, gp = _oracle_chain(b, res, G)
    # (1) a pass that never reaches the rasterize node: nothing may be prepared (it would be left behind)
    c0 = m.counters()
    (g_attr,) = torch.autograd.grad(out, [attr], _t(G), retain_graph=True)
    _close(g_attr, ga, "g_attr")
    assert m.counters()["fused"] == c0["fused"]
    # (2) backward(inputs=[pos]): the rasterize node runs and captures nothing -> fused
    torch.autograd.backward(out, _t(G), inputs=[pos], retain_graph=True)
    _close(pos.grad, gp, "g_pos (inputs=[pos])")
    assert m.counters()["fused"] == c0["fused"] + 1
    # (3) the same graph again, plain: accumulates
    torch.autograd.backward(out, _t(G))
    _close(pos.grad, 2 * gp, "g_pos accumulated"); _close(attr.grad, ga, "g_attr")


def test_pixel_differentials_range_mode_and_broadcast(host, dr):
    m, lib = host
    b, res, G = _scene(n=2, seed=17)
    ctx = _Ctx(dr)
    rng = np.random.default_rng(1)
    # range mode: one vertex buffer, two ranges of the triangle list; attributes shared ([V, A])
    pos2 = b["pos"][0].copy()
    T = b["tri"].shape[0]
    ranges = np.array([[0, T // 2], [T // 2, T - T // 2]], np.int32)
    attr2 = b["attr"][0].copy()
    A = attr2.shape[-1]
    Gda = rng.normal(size=(2,) + res + (2 * A,)).astype(np.float32)
    pos, attr, tri = _t(pos2, True), _t(attr2, True), _t(b["tri"])
    c0 = m.counters()
    rast, rast_db = dr.rasterize(ctx, pos, tri, res, ranges=_t(ranges))
    out, out_da = dr.interpolate(attr, rast, tri, rast_db=rast_db, diff_attrs="all")
    torch.autograd.backward([out, out_da], [_t(G), _t(Gda)])
    ro, rdbo = oracle.rasterize(pos2, b["tri"], res, ranges=ranges)
    oo, odao = oracle.interpolate(attr2, ro, b["tri"], rast_db=rdbo, diff_attrs="all")
    ga, gr, grdb = oracle.interpolate_grad(attr2, ro, b["tri"], G, rast_db=rdbo, dda=Gda, diff_attrs="all")
    gp = oracle.rasterize_grad(pos2, b["tri"], ro, gr, grdb)
    assert (rast.detach().numpy()[..., 3] != ro[..., 3]).sum() == 0
    _close(out, oo, "out"); _close(out_da, odao, "out_da"); _close(attr.grad, ga, "g_attr"); _close(pos.grad, gp, "g_pos")
    assert m.counters()["fused"] == c0["fused"] + 1
    # a selected list of differentials, grad_db=False: rast_db's gradient does not reach pos
    pos, attr = _t(pos2, True), _t(attr2, True)
    rast, rast_db = dr.rasterize(ctx, pos, tri, res, ranges=_t(ranges), grad_db=False)
    out, out_da = dr.interpolate(attr, rast, tri, rast_db=rast_db, diff_attrs=[1, -1])
    Gd2 = np.ascontiguousarray(Gda[..., :4])
    torch.autograd.backward([out, out_da], [_t(G), _t(Gd2)])
    ga, gr, grdb = oracle.interpolate_grad(attr2, ro, b["tri"], G, rast_db=rdbo, dda=Gd2, diff_attrs=[1, -1])
    gp = oracle.rasterize_grad(pos2, b["tri"], ro, gr)
    _close(attr.grad, ga, "g_attr (list)"); _close(pos.grad, gp, "g_pos (grad_db=False)")


def test_two_interpolations_of_one_rast_both_prepare(host, dr):
    m, lib = host
    b, res, G = _scene(seed=19)
    ctx = _Ctx(dr)
    rng = np.random.default_rng(2)
    attr_b = rng.normal(size=b["attr"].shape[:-1] + (3,)).astype(np.float32)
    Gb = rng.normal(size=(2,) + res + (3,)).astype(np.float32)
    pos, attr, attr2, tri = _t(b["pos"], True), _t(b["attr"], True), _t(attr_b, True), _t(b["tri"])
    c0 = m.counters()
    rast, _ = dr.rasterize(ctx, pos, tri, res)
    o1, _ = dr.interpolate(attr, rast, tri)
    o2, _ = dr.interpolate(attr2, rast, tri)
    torch.autograd.backward([o1, o2], [_t(G), _t(Gb)])
    ro, _ = oracle.rasterize(b["pos"], b["tri"], res)
    ga1, gr1, _ = oracle.interpolate_grad(b["attr"], ro, b["tri"], G)
    ga2, gr2, _ = oracle.interpolate_grad(attr_b, ro, b["tri"], Gb)
    gp = oracle.rasterize_grad(b["pos"], b["tri"], ro, gr1 + gr2)
    _close(attr.grad, ga1, "g_attr 1"); _close(attr2.grad, ga2, "g_attr 2"); _close(pos.grad, gp, "g_pos")
    c1 = m.counters()
    assert c1["fused"] == c0["fused"] + 2 and c1["fused_alone"] == c0["fused_alone"] + 1


def test_records_die_with_version_and_storage(host, dr):
    m, lib = host
    b, res, G = _scene(seed=23)
    ctx = _Ctx(dr)
    pos, attr, tri = _t(b["pos"]), _t(b["attr"]), _t(b["tri"])
    rast, rast_db = dr.rasterize(ctx, pos, tri, res)
    assert rast.grad_fn is None                                   # nothing requires a gradient: no node
    f = m.flags_of(rast, m.KIND_RAST)
    assert f is not None and f.dtype == torch.uint8 and m.flags_of(rast, m.KIND_ZERO) is None
    th, tw = (res[0] + 7) // 8, (res[1] + 7) // 8
    want = torch.nn.functional.max_pool2d((rast[..., 3] > 0).float()[:, None], 8, ceil_mode=True)[:, 0] > 0
    assert torch.equal(f[:2 * th * tw].view(2, th, tw) != 0, want)
    assert m.flags_of(rast.detach(), m.KIND_RAST) is not None     # the same storage and version
    assert m.flags_of(rast.clone(), m.KIND_RAST) is None and m.flags_of(rast[:1], m.KIND_RAST) is None and m.flags_of(rast_db, m.KIND_RAST) is None
    out, out_da = dr.interpolate(attr, rast, tri)
    assert m.flags_of(out, m.KIND_ZERO) is not None and m.flags_of(out, m.KIND_ZERO).data_ptr() == f.data_ptr()
    rast.mul_(1.0)                                                # version counter moves: the record is void
    assert m.flags_of(rast, m.KIND_RAST) is None
    from nvdiffrast_amd.torch import _plugin
    _plugin.set_tile_skipping(False)
    try:
        assert m.flags_of(out, m.KIND_ZERO) is None
    finally:
        _plugin.set_tile_skipping(True)
    del rast, out
    r2, _ = dr.rasterize(ctx, pos, tri, res)                      # (the sweep on attach drops the dead records)
    assert m.flags_of(r2, m.KIND_RAST) is not None


def test_in_place_change_of_a_saved_tensor_is_reported(host, dr):
    b, res, G = _scene(seed=29)
    ctx = _Ctx(dr)
    pos, attr, tri = _t(b["pos"], True), _t(b["attr"], True), _t(b["tri"])
    rast, _ = dr.rasterize(ctx, pos, tri, res)
    out, _ = dr.interpolate(attr, rast, tri)
    with torch.no_grad():
        rast.mul_(2.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        torch.autograd.backward(out, _t(G))


def test_a_rast_that_is_not_rasterizes_own_is_never_fused(host, dr):
    m, lib = host
    b, res, G = _scene(seed=31)
    ctx = _Ctx(dr)
    pos, attr, tri = _t(b["pos"], True), _t(b["attr"], True), _t(b["tri"])
    rast, _ = dr.rasterize(ctx, pos, tri, res)
    c0 = m.counters()
    out, _ = dr.interpolate(attr, rast * 1.0, tri)                # a copy: its grad_fn is torch's multiplication
    torch.autograd.backward(out, _t(G))
    _, _, _, ga, gr, gp = _oracle_chain(b, res, G)
    _close(attr.grad, ga, "g_attr"); _close(pos.grad, gp, "g_pos")
    assert m.counters()["fused"] == c0["fused"]
    # another triangle tensor with the same contents (pose-style scripts interpolate with their own index buffer): not fused either
    pos.grad = attr.grad = None
    rast, _ = dr.rasterize(ctx, pos, tri, res)
    out, _ = dr.interpolate(attr, rast, tri.clone())
    torch.autograd.backward(out, _t(G))
    _close(pos.grad, gp, "g_pos"); assert m.counters()["fused"] == c0["fused"]


def test_no_grad_and_unused_outputs(host, dr):
    b, res, G = _scene(seed=37)
    ctx = _Ctx(dr)
    pos, attr, tri = _t(b["pos"], True), _t(b["attr"], True), _t(b["tri"])
    with torch.no_grad():
        rast, _ = dr.rasterize(ctx, pos, tri, res)
        out, _ = dr.interpolate(attr, rast, tri)
    assert rast.grad_fn is None and out.grad_fn is None and not out.requires_grad
    # only rast_db is used downstream; grad_db=False -> pos receives nothing at all
    rast, rast_db = dr.rasterize(ctx, pos, tri, res, grad_db=False)
    rast_db.sum().backward()
    assert pos.grad is None
    rast, rast_db = dr.rasterize(ctx, pos, tri, res)
    Wd = np.random.default_rng(3).normal(size=tuple(rast_db.shape)).astype(np.float32)
    (rast_db * _t(Wd)).sum().backward()
    ro, _ = oracle.rasterize(b["pos"], b["tri"], res)
    _close(pos.grad, oracle.rasterize_grad(b["pos"], b["tri"], ro, np.zeros_like(ro), Wd), "g_pos from rast_db alone")


def test_depth_peeling_layers(host, dr):
    b, res, G = _scene(seed=41)
    ctx = _Ctx(dr)
    pos, tri = _t(b["pos"]), _t(b["tri"])
    layers = []
    with dr.DepthPeeler(ctx, pos, tri, res) as peeler:
        for _ in range(3):
            layers.append(peeler.rasterize_next_layer()[0].numpy().copy())
    depth = None
    for k in range(3):
        want, _, depth = oracle.rasterize(b["pos"], b["tri"], res, peel_depth=depth, return_depth=True)
        assert (layers[k][..., 3] != want[..., 3]).sum() == 0, "layer %d" % k
    assert (layers[0][..., 3] != layers[1][..., 3]).any()


def test_clean_scratch_flag_and_declined_calls(host, dr):
    m, lib = host
    b, res, G = _scene(seed=43)
    ctx = _Ctx(dr)
    pos, attr, tri = _t(b["pos"]), _t(b["attr"]), _t(b["tri"])
    k0 = lib.nvdr_stub_calls(C_CLEAN)
    dr.rasterize(ctx, pos, tri, res)
    dr.rasterize(ctx, pos, tri, res)                              # same layout as the previous successful call: clean
    dr.rasterize(ctx, pos, tri, (res[0] + 8, res[1]))             # another layout: not clean
    assert lib.nvdr_stub_calls(C_CLEAN) == k0 + 1
    st = ctx.cpp_wrapper.host_state(m)
    assert st.scratch_bytes > 0 and not st.captured
    empty = torch.empty((0, 2), dtype=torch.int32)
    assert m.rasterize(st, pos.double(), tri, 8, 8, empty, True, -1) is None          # dtype
    assert m.rasterize(st, pos[:, :, :3], tri, 8, 8, empty, True, -1) is None         # shape / contiguity
    assert m.rasterize(st, pos, tri.long(), 8, 8, empty, True, -1) is None
    assert m.rasterize(st, pos, tri, 0, 8, empty, True, -1) is None
    assert m.rasterize(st, pos[0], tri, 8, 8, empty, True, -1) is None                # range mode without ranges
    rast, rast_db = dr.rasterize(ctx, pos, tri, res)
    assert m.interpolate(attr.double(), rast, tri, None, False, []) is None
    assert m.interpolate(attr[:1].expand(3, -1, -1), rast, tri, None, False, []) is None    # minibatch mismatch (3 vs 2), non-contiguous
    assert m.interpolate(attr, rast[..., :3], tri, None, False, []) is None
    assert m.interpolate(attr, rast, tri, rast_db[:, :8], True, []) is None
    assert m.interpolate(attr, rast, tri, rast_db, False, list(range(33))) is None     # IP_MAX_DIFF_ATTRS
    m.set_verify(True)
    try:
        assert m.interpolate(attr, rast, tri, None, False, []) is None                  # the checking mode is the Python layer's
    finally:
        m.set_verify(False)


@pytest.mark.parametrize("mode", ["linear-mipmap-linear", "linear-mipmap-nearest", "linear", "nearest", "bias-only", "cube"])
def test_texture_and_antialias_through_the_compiled_layer(host, dr, mode):
    """The full pipeline of config 3 on the CPU stub: every forward value and every gradient equals the oracle's chain."""
    m, lib = host
    b, res, G = _scene(seed=47)
    ctx = _Ctx(dr)
    rng = np.random.default_rng(4)
    V = b["pos"].shape[1]
    cube = mode == "cube"
    uvattr = (rng.normal(size=(V, 3)) if cube else rng.uniform(0, 1, size=(V, 2))).astype(np.float32)
    tex_np = rng.uniform(size=(1, 6, 16, 16, 3) if cube else (1, 32, 16, 3)).astype(np.float32)
    pos, uva, tex, tri = _t(b["pos"], True), _t(uvattr, True), _t(tex_np, True), _t(b["tri"])
    n = b["pos"].shape[0]
    bias_np = rng.uniform(0, 2, size=(n,) + res).astype(np.float32)
    bias = _t(bias_np, True)
    f0, g0 = lib.nvdr_stub_calls(6), lib.nvdr_stub_calls(7)
    rast, rast_db = dr.rasterize(ctx, pos, tri, res)
    uv, uv_da = dr.interpolate(uva, rast, tri, rast_db=rast_db, diff_attrs="all")
    kw = dict(boundary_mode="cube" if cube else "wrap")
    if mode in ("linear", "nearest"):
        col = dr.texture(tex, uv, filter_mode=mode, **kw)
    elif mode == "bias-only":
        col = dr.texture(tex, uv, mip_level_bias=bias, filter_mode="linear-mipmap-linear", **kw)
    elif cube:
        col = dr.texture(tex, uv, uv_da, filter_mode="linear-mipmap-linear", **kw)
    else:
        col = dr.texture(tex, uv, uv_da, mip_level_bias=bias, filter_mode=mode, **kw)
    assert col.grad_fn.name() == "NvdrTextureBackward"
    from nvdiffrast_amd.torch import _plugin
    topo = _plugin.TopologyHashWrapper()
    topo.ev_hash = torch.zeros(16, dtype=torch.int32)       # (built on the GPU by the product; the stub's oracle has its own edge map)
    aa = dr.antialias(col, rast, pos, tri, topology_hash=topo, pos_gradient_boost=2.0)
    assert aa.grad_fn.name() == "NvdrAntialiasBackward"
    dy = rng.normal(size=tuple(aa.shape)).astype(np.float32)
    aa.backward(_t(dy))
    assert (lib.nvdr_stub_calls(6), lib.nvdr_stub_calls(7)) == (f0 + 1, g0 + 1)
    # the oracle's chain on the same inputs
    ro, rdbo = oracle.rasterize(b["pos"], b["tri"], res)
    uvo, uvdao = oracle.interpolate(uvattr, ro, b["tri"], rast_db=rdbo, diff_attrs="all")
    okw = dict(boundary_mode="cube" if cube else "wrap")
    if mode in ("linear", "nearest"):
        okw.update(filter_mode=mode)
    elif mode == "bias-only":
        okw.update(mip_level_bias=bias_np, filter_mode="linear-mipmap-linear")
    elif cube:
        okw.update(uv_da=uvdao, filter_mode="linear-mipmap-linear")
    else:
        okw.update(uv_da=uvdao, mip_level_bias=bias_np, filter_mode=mode)
    colo = oracle.texture(tex_np, uvo, **okw)
    aao = oracle.antialias(colo, ro, b["pos"], b["tri"])
    _close(col, colo, "col"); _close(aa, aao, "aa")
    g_col, g_pos_aa = oracle.antialias_grad(colo, ro, b["pos"], b["tri"], dy)
    tg = oracle.texture_grad(tex_np, uvo, g_col, **okw)
    g_tex, g_uv, g_uv_da, g_bias = tg["tex"], tg["uv"], tg["uv_da"], tg["mip_level_bias"]
    _close(tex.grad, g_tex, "g_tex")
    if mode not in ("nearest",):
        ga, gr, grdb = oracle.interpolate_grad(uvattr, ro, b["tri"], g_uv, rast_db=rdbo,
                                                dda=(g_uv_da if g_uv_da is not None else np.zeros_like(uvdao)), diff_attrs="all")
        gp = oracle.rasterize_grad(b["pos"], b["tri"], ro, gr, grdb) + 2.0 * g_pos_aa
        _close(uva.grad, ga, "g_uvattr"); _close(pos.grad, gp, "g_pos")
    else:
        assert uva.grad is None
        _close(pos.grad, 2.0 * g_pos_aa, "g_pos (antialias only)")
    if mode in ("bias-only", "linear-mipmap-linear"):
        _close(bias.grad, g_bias, "g_bias")


def test_texture_declines_what_it_does_not_serve(host, dr):
    m, lib = host
    rng = np.random.default_rng(5)
    tex, uv = _t(rng.uniform(size=(1, 8, 8, 3)).astype(np.float32)), _t(rng.uniform(size=(1, 4, 4, 2)).astype(np.float32))
    assert m.texture(tex, uv, None, None, None, 0, [], False, 1, 1, True) is not None
    assert m.texture(tex.double(), uv, None, None, None, 0, [], False, 1, 1, True) is None
    assert m.texture(tex, uv[..., :1], None, None, None, 0, [], False, 1, 1, True) is None
    assert m.texture(tex, uv, None, None, None, 0, [], False, 3, 1, True) is None                 # mipmapped without uv_da / bias / mip
    mip = m.construct_mip(tex, -1, False)
    assert mip is not None and mip.numel() == 3 * (16 + 4 + 1)
    uv_da = _t(np.zeros((1, 4, 4, 4), np.float32))
    assert m.texture(tex, uv, uv_da, None, mip, -1, [1, 8, 8, 3], False, 3, 1, True) is not None
    assert m.texture(tex, uv, uv_da, None, mip, -1, [1, 8, 8, 4], False, 3, 1, True) is None      # wrapper made for another texture
    assert m.texture(tex, uv, uv_da, None, mip[:-1], -1, [1, 8, 8, 3], False, 3, 1, True) is None
    assert m.construct_mip(_t(np.zeros((1, 6, 8, 3), np.float32)), -1, False) is None               # odd extent above 1: the Python layer words it
