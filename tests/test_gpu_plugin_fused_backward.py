"""The fused backward of rasterize -> interpolate for a caller that binds `_plugin` the way the reference's ops.py binds its
pybind module: interpolate_grad[_da] prepares the position gradient, rasterize_grad[_db] hands it
out (`_plugin.py`, section "fused backward"; VERDICT r4 "missing" 4).

The binding below is written for this test in the reference's calling convention (nvdiffrast/torch/ops.py:66-90, 146-190):
outputs nobody used arrive in backward as materialised zeros, `grad_db` is decided by the caller, no extra arguments.
Every gradient is compared with the oracle (pinned to the reference), at the bars of tests/conftest.py."""
import gc

import numpy as np
import pytest
import torch
from conftest import grad_tol, within

from nvdiffrast_amd.utils import m10k_batch

pytestmark = pytest.mark.gpu


def _t(a, dev="cuda"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _kernels(lib, _capi, fn):
    lib.nvdr_profile_reset()
    lib.nvdr_profile_enable(1)
    try:
        fn()
        torch.cuda.synchronize()
        return set(_capi.profile_read())
    finally:
        lib.nvdr_profile_enable(0)
        lib.nvdr_profile_reset()


def _binding(P, materialize=True):
    """Autograd functions over plugin `P` with the reference's call lists.  materialize=False: the one line INTEGRATION.md
    section 1 suggests on top (gradients of unused outputs arrive as None)."""
    empty_ranges = torch.empty((0, 2), dtype=torch.int32)

    class Rasterize(torch.autograd.Function):
        @staticmethod
        def forward(ctx, state, pos, tri, res, grad_db):
            ctx.set_materialize_grads(materialize)
            out, out_db = P.rasterize_fwd_cuda(state, pos, tri, res, empty_ranges, -1)
            ctx.save_for_backward(pos, tri, out)
            ctx.grad_db = grad_db
            return out, out_db

        @staticmethod
        def backward(ctx, dy, ddb):
            pos, tri, out = ctx.saved_tensors
            g = P.rasterize_grad_db(pos, tri, out, dy, ddb) if ctx.grad_db else P.rasterize_grad(pos, tri, out, dy)
            return None, g, None, None, None

    class Interpolate(torch.autograd.Function):
        @staticmethod
        def forward(ctx, attr, rast, tri):
            out, out_da = P.interpolate_fwd(attr, rast, tri)
            ctx.save_for_backward(attr, rast, tri)
            return out, out_da

        @staticmethod
        def backward(ctx, dy, _):
            attr, rast, tri = ctx.saved_tensors
            g_attr, g_rast = P.interpolate_grad(attr, rast, tri, dy)
            return g_attr, g_rast, None

    class InterpolateDa(torch.autograd.Function):
        @staticmethod
        def forward(ctx, attr, rast, tri, rast_db, diff_all, diff_list):
            out, out_da = P.interpolate_fwd_da(attr, rast, tri, rast_db, diff_all, diff_list)
            ctx.save_for_backward(attr, rast, tri, rast_db)
            ctx.diff = (diff_all, diff_list)
            return out, out_da

        @staticmethod
        def backward(ctx, dy, dda):
            attr, rast, tri, rast_db = ctx.saved_tensors
            g_attr, g_rast, g_rast_db = P.interpolate_grad_da(attr, rast, tri, dy, rast_db, dda, *ctx.diff)
            return g_attr, g_rast, None, g_rast_db, None, None

    return Rasterize, Interpolate, InterpolateDa


def test_the_reference_style_binding_gets_the_fused_backward(dr, oracle):
    from nvdiffrast_amd import _capi
    from nvdiffrast_amd.torch import _plugin
    lib = _capi.load()
    Rasterize, Interpolate, _ = _binding(_plugin)
    N, res = 2, (128, 136)
    b = m10k_batch(N, seed=41, nx=30, ny=15)
    rng = np.random.default_rng(5)
    G = rng.normal(size=(N,) + res + (4,)).astype(np.float32)
    tri = _t(b["tri"])
    ro, _ = oracle.rasterize(b["pos"], b["tri"], res)
    ga, gr, _ = oracle.interpolate_grad(b["attr"], ro, b["tri"], G)
    gp = oracle.rasterize_grad(b["pos"], b["tri"], ro, gr)
    got = {}

    def step(state, extra=None, grad_db=True, hook=None, drop_rast=False):
        pos = _t(b["pos"]).requires_grad_(True)
        attr = _t(b["attr"]).requires_grad_(True)
        rast, rast_db = Rasterize.apply(state, pos, tri, res, grad_db)
        out, _ = Interpolate.apply(attr, rast, tri)
        loss = (out * _t(G)).sum()
        if extra is not None:
            loss = loss + extra(rast, rast_db)
        if hook is not None:
            rast.register_hook(hook)
        if drop_rast:
            del rast, rast_db, out
            gc.collect()
        loss.backward()
        got["pos"], got["attr"] = pos.grad, attr.grad

    # (1) nothing else reads rast: one fused kernel, the pass over the materialised zero ddb, no two-kernel pair
    state = _plugin.RasterizeCRStateWrapper(0)
    before = _plugin.fused_backward_count()
    names = _kernels(lib, _capi, lambda: step(state))
    assert "interp_raster_grad" in names and "raster_grad_db_only" in names, names
    assert not ({"interp_grad", "raster_grad", "raster_grad_db"} & names), names
    after = _plugin.fused_backward_count()
    assert after["used"] == before["used"] + 1 and after["materialized"] == before["materialized"]
    within("plugin fused: g_pos", got["pos"].cpu().numpy(), gp, grad_tol(gp))
    within("plugin fused: g_attr", got["attr"].cpu().numpy(), ga, grad_tol(ga))

    # (1a) the binding with set_materialize_grads(False): ddb arrives as None, no pass over zeros at all
    R2, I2, _ = _binding(_plugin, materialize=False)

    def step_line(state, only_db=False):
        pos = _t(b["pos"]).requires_grad_(True)
        attr = _t(b["attr"]).requires_grad_(True)
        rast, rast_db = R2.apply(state, pos, tri, res, True)
        out, _ = I2.apply(attr, rast, tri)
        ((rast_db * rast_db).sum() if only_db else (out * _t(G)).sum()).backward()
        got["pos"], got["attr"] = pos.grad, attr.grad
    names = _kernels(lib, _capi, lambda: step_line(_plugin.RasterizeCRStateWrapper(0)))
    assert names & {"interp_raster_grad", "raster_grad_db_only", "raster_grad", "raster_grad_db", "interp_grad"} == {"interp_raster_grad"}, names
    within("plugin fused, no materialised zeros: g_pos", got["pos"].cpu().numpy(), gp, grad_tol(gp))
    names = _kernels(lib, _capi, lambda: step_line(_plugin.RasterizeCRStateWrapper(0), only_db=True))       # dy None, ddb real
    assert "raster_grad_db_only" in names and not ({"raster_grad", "raster_grad_db"} & names), names
    rdb_o = oracle.rasterize(b["pos"], b["tri"], res)[1]
    gp_only = oracle.rasterize_grad(b["pos"], b["tri"], ro, np.zeros_like(ro), 2.0 * rdb_o)
    within("plugin, only rast_db used: g_pos", got["pos"].cpu().numpy(), gp_only, grad_tol(gp_only))

    # (1b) grad_db=False: rasterize_grad, nothing to add
    names = _kernels(lib, _capi, lambda: step(state, grad_db=False))
    assert names & {"interp_raster_grad", "raster_grad_db_only", "raster_grad", "interp_grad"} == {"interp_raster_grad"}, names
    within("plugin fused, grad_db=False: g_pos", got["pos"].cpu().numpy(), gp, grad_tol(gp))

    # (1c) the caller's own `rast` is gone when rasterize's node runs: the stand-in carries the exchange
    before = _plugin.fused_backward_count()
    step(state, drop_rast=True)
    assert _plugin.fused_backward_count()["used"] == before["used"] + 1
    within("plugin fused, rast dropped: g_pos", got["pos"].cpu().numpy(), gp, grad_tol(gp))

    # (2) a real gradient for rast_db from somewhere else: its share is added to the prepared gradient
    wdb = rng.normal(size=(N,) + res + (4,)).astype(np.float32)
    wdb[0, 30:90] = 0.0
    gp_db = oracle.rasterize_grad(b["pos"], b["tri"], ro, gr, wdb)
    assert np.abs(gp_db - gp).max() > 1e-3 * np.abs(gp).max()
    before = _plugin.fused_backward_count()
    names = _kernels(lib, _capi, lambda: step(state, extra=lambda rast, rast_db: (rast_db * _t(wdb)).sum()))
    assert "interp_raster_grad" in names and "raster_grad_db_only" in names and "raster_grad_db" not in names, names
    assert _plugin.fused_backward_count()["used"] == before["used"] + 1
    within("plugin fused + ddb: g_pos", got["pos"].cpu().numpy(), gp_db, grad_tol(gp_db))

    # (3) an in-place hook on rast's gradient: the object that arrives is the stand-in, but edited -> recomputed from what it holds
    def double(g):
        g.mul_(2.0)
        return g
    gp_2 = oracle.rasterize_grad(b["pos"], b["tri"], ro, 2.0 * gr)
    state_h = _plugin.RasterizeCRStateWrapper(0)
    before = _plugin.fused_backward_count()
    step(state_h, hook=double)
    assert _plugin.fused_backward_count()["discarded"] == before["discarded"] + 1
    within("plugin fused, edited stand-in: g_pos", got["pos"].cpu().numpy(), gp_2, grad_tol(gp_2))

    # (4) a second contributor to rast's gradient: prepared, discarded, recomputed from the sum; then the context stops preparing
    wmask = rng.normal(size=(N,) + res).astype(np.float32)
    gr2 = gr.copy(); gr2[..., 0] += wmask
    gp2 = oracle.rasterize_grad(b["pos"], b["tri"], ro, gr2)
    extra = lambda rast, rast_db: (rast[..., 0] * _t(wmask)).sum()                        # noqa: E731
    before = _plugin.fused_backward_count()
    names = _kernels(lib, _capi, lambda: step(state, extra=extra))
    assert "interp_raster_grad" in names and "raster_grad_db" in names, names
    assert _plugin.fused_backward_count()["discarded"] == before["discarded"] + 1
    within("plugin fused discarded: g_pos", got["pos"].cpu().numpy(), gp2, grad_tol(gp2))
    within("plugin fused discarded: g_attr", got["attr"].cpu().numpy(), ga, grad_tol(ga))
    names = _kernels(lib, _capi, lambda: step(state, extra=extra))
    assert "interp_raster_grad" not in names and {"interp_grad", "raster_grad_db"} <= names, names
    within("plugin, after discard: g_pos", got["pos"].cpu().numpy(), gp2, grad_tol(gp2))
    _plugin.set_fused_backward("auto")                                                    # re-arm (other tests share nothing with `state`, but the epoch is global)

    # (5) switched off: the reference's two kernels
    _plugin.set_fused_backward("off")
    try:
        names = _kernels(lib, _capi, lambda: step(_plugin.RasterizeCRStateWrapper(0)))
        assert "interp_raster_grad" not in names and {"interp_grad", "raster_grad_db"} <= names, names
        within("plugin, fused off: g_pos", got["pos"].cpu().numpy(), gp, grad_tol(gp))
    finally:
        _plugin.set_fused_backward("auto")

    # (6) who looks at rast's gradient sees the reference's values (autograd.grad on rast), computed on demand
    pos = _t(b["pos"]).requires_grad_(True)
    attr = _t(b["attr"]).requires_grad_(True)
    rast, _ = Rasterize.apply(_plugin.RasterizeCRStateWrapper(0), pos, tri, res, True)
    out, _ = Interpolate.apply(attr, rast, tri)
    (g_rast,) = torch.autograd.grad((out * _t(G)).sum(), [rast])
    within("plugin fused: g_rast on demand", (g_rast + 0).cpu().numpy(), gr, grad_tol(gr))


def test_the_reference_style_binding_fuses_the_differential_pair(dr, oracle):
    from nvdiffrast_amd import _capi
    from nvdiffrast_amd.torch import _plugin
    lib = _capi.load()
    Rasterize, _, InterpolateDa = _binding(_plugin)
    N, res = 2, (128, 128)
    b = m10k_batch(N, seed=42, nx=30, ny=15, attrs=2)
    rng = np.random.default_rng(6)
    G = rng.normal(size=(N,) + res + (2,)).astype(np.float32)
    Gd = rng.normal(size=(N,) + res + (4,)).astype(np.float32)
    tri = _t(b["tri"])
    ro, rdbo = oracle.rasterize(b["pos"], b["tri"], res)
    ga, gr, grdb = oracle.interpolate_grad(b["uv"], ro, b["tri"], G, rdbo, Gd, "all")
    gp = oracle.rasterize_grad(b["pos"], b["tri"], ro, gr, grdb)
    got = {}

    def step(state, extra=None, grad_db=True):
        pos = _t(b["pos"]).requires_grad_(True)
        uv = _t(b["uv"]).requires_grad_(True)
        rast, rast_db = Rasterize.apply(state, pos, tri, res, grad_db)
        out, out_da = InterpolateDa.apply(uv, rast, tri, rast_db, True, [])
        loss = (out * _t(G)).sum() + (out_da * _t(Gd)).sum()
        if extra is not None:
            loss = loss + extra(rast_db)
        loss.backward()
        got["pos"], got["uv"] = pos.grad, uv.grad

    state = _plugin.RasterizeCRStateWrapper(0)
    before = _plugin.fused_backward_count()
    names = _kernels(lib, _capi, lambda: step(state))
    assert "interp_raster_grad_da" in names and not ({"raster_grad_db", "raster_grad_db_only", "interp_grad_da"} & names), names
    assert _plugin.fused_backward_count()["used"] == before["used"] + 1
    within("plugin fused da: g_pos", got["pos"].cpu().numpy(), gp, grad_tol(gp))
    within("plugin fused da: g_uv", got["uv"].cpu().numpy(), ga, grad_tol(ga))

    # grad_db=False: the prepared gradient holds rast_db's share, which this caller does not want -> recomputed without it
    gp_nodb = oracle.rasterize_grad(b["pos"], b["tri"], ro, gr)
    before = _plugin.fused_backward_count()
    step(_plugin.RasterizeCRStateWrapper(0), grad_db=False)
    assert _plugin.fused_backward_count()["discarded"] == before["discarded"] + 1
    within("plugin fused da, grad_db=False: g_pos", got["pos"].cpu().numpy(), gp_nodb, grad_tol(gp_nodb))
    _plugin.set_fused_backward("auto")

    # a second contributor to rast_db's gradient
    w = rng.normal(size=(N,) + res + (4,)).astype(np.float32)
    gp2 = oracle.rasterize_grad(b["pos"], b["tri"], ro, gr, grdb + w)
    names = _kernels(lib, _capi, lambda: step(state, lambda rdb: (rdb * _t(w)).sum()))
    assert "interp_raster_grad_da" in names and "raster_grad_db" in names, names
    within("plugin fused da discarded: g_pos", got["pos"].cpu().numpy(), gp2, grad_tol(gp2))
    _plugin.set_fused_backward("auto")


def test_db_only_pass_is_the_difference_of_the_two_gradients(dr, oracle):
    """nvdr_rasterize_grad with dy == NULL adds exactly rast_db's share (linearity of rasterize_grad_db in (dy, ddb))."""
    from nvdiffrast_amd import _capi
    N, res = 3, (72, 200)
    b = m10k_batch(N, seed=43, nx=24, ny=12)
    rng = np.random.default_rng(7)
    ro, _ = oracle.rasterize(b["pos"], b["tri"], res)
    ddb = rng.normal(size=(N,) + res + (4,)).astype(np.float32)
    ddb[1] = 0.0
    ddb[2, :, 100:] = 0.0
    want = oracle.rasterize_grad(b["pos"], b["tri"], ro, np.zeros_like(ro), ddb)
    pos, tri, out, d = _t(b["pos"]), _t(b["tri"]), _t(ro), _t(ddb)
    g = torch.zeros_like(pos)
    V, T = pos.shape[1], tri.shape[0]
    rc = _capi.load().nvdr_rasterize_grad(pos.data_ptr(), tri.data_ptr(), out.data_ptr(), None, d.data_ptr(), 1, N, V, T,
                                          res[0], res[1], g.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    _capi.check(rc, "rasterize_grad")
    within("db-only pass", g.cpu().numpy(), want, grad_tol(want))
    # the same through the plugin: dy None is what a binding with set_materialize_grads(False) passes for an unused rast
    from nvdiffrast_amd.torch import _plugin
    within("plugin, dy None", _plugin.rasterize_grad_db(pos, tri, out, None, d).cpu().numpy(), want, grad_tol(want))
    assert not _plugin.rasterize_grad_db(pos, tri, out, None, None).any()
    with pytest.raises(RuntimeError, match="ddb must have shape"):
        _plugin.rasterize_grad_db(pos, tri, out, None, d[:, :-1])
    rc = _capi.load().nvdr_rasterize_grad(pos.data_ptr(), tri.data_ptr(), out.data_ptr(), None, None, 1, N, V, T,
                                          res[0], res[1], g.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    assert rc != 0                                                                          # neither gradient: refused
