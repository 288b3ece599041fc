"""Launches that walk the work order behind the tile flags (include/nvdr_hip.h `tile_flags`; csrc/nvdr_device.hpp
decode_block_ordered): kept for batches of at least 2048 bins of 64x64 pixels, so these tests are the only ones at that size.
The fused backward pass, interpolate / rasterize backward, the texture kernels and the antialias discontinuity pass through
the operator layer, every op against the oracle on the HIP path's own inputs (oracle/chain.py explains why).  The checker is
the C oracle WITHOUT the per-call cross-check against the reference emulation (`raw_oracle`): eight million pixels per call are
beyond what that emulation does in test time, and the background's texel collects six million f32 terms there, whose order
of summation alone moves it by 7e-5 of its value (tests/test_gpu_reference_direct.py covers that case at its own bar)."""
import numpy as np
import pytest
import torch
from conftest import ATOL, BIG_SUM_FACTOR, discontinuous_pixels, grad_tol, within

from nvdiffrast_amd.torch import _plugin
from nvdiffrast_amd.utils import m10k_batch

pytestmark = pytest.mark.gpu


def _t(a, dev="cuda"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("n,res", [(32, (512, 512)), (37, (500, 460))])       # 2048 bins exactly; border bins + tile width not a multiple of 8
def test_every_consumer_on_an_ordered_launch(dr, raw_oracle, n, res):
    oracle = raw_oracle
    b = m10k_batch(n, seed=90 + n, nx=24, ny=14)
    V = b["pos"].shape[1]
    rng = np.random.default_rng(n)
    uvattr = rng.uniform(0, 1, size=(V, 2)).astype(np.float32)
    tex_np = rng.uniform(size=(1, 128, 128, 3)).astype(np.float32)
    assert _plugin.tile_flags_bytes(n, *res) > (n * ((res[0] + 7) // 8) * ((res[1] + 7) // 8) + 15) // 16 * 16       # an order is kept

    ctx = dr.RasterizeCudaContext()
    pos = _t(b["pos"]).requires_grad_(True)
    tri = _t(b["tri"])
    uva = _t(uvattr).requires_grad_(True)
    tex = _t(tex_np).requires_grad_(True)
    rast, rast_db = dr.rasterize(ctx, pos, tri, res)
    uv, uv_da = dr.interpolate(uva, rast, tri, rast_db=rast_db, diff_attrs="all")
    uv.retain_grad(); uv_da.retain_grad()
    col = dr.texture(tex, uv, uv_da, filter_mode="linear-mipmap-linear", boundary_mode="wrap")
    col.retain_grad()
    aa = dr.antialias(col, rast, pos, tri)
    dy = rng.normal(size=aa.shape).astype(np.float32)
    used = _plugin.fused_backward_count()["used"]
    aa.backward(_t(dy))
    assert _plugin.fused_backward_count()["used"] == used + 1            # the fused pair ran (on the ordered launch)

    # forward, op by op
    ro, rdbo = oracle.rasterize(b["pos"], b["tri"], res)
    rh = _np(rast)
    assert (rh[..., 3] != ro[..., 3]).sum() == 0
    uvo, uvdao = oracle.interpolate(uvattr, rh, b["tri"], _np(rast_db), "all")
    within("ordered: uv", _np(uv), uvo, ATOL)
    within("ordered: uv_da", _np(uv_da), uvdao, grad_tol(uvdao))
    kw = dict(filter_mode="linear-mipmap-linear", boundary_mode="wrap")
    within("ordered: texture", _np(col), oracle.texture(tex_np, _np(uv), _np(uv_da), **kw), ATOL)
    within("ordered: antialias", _np(aa), oracle.antialias(_np(col), rh, b["pos"], b["tri"]), ATOL)
    # backward, op by op, each from the upstream gradient the HIP path itself produced
    g_col, g_pos_aa = oracle.antialias_grad(_np(col), rh, b["pos"], b["tri"], dy)
    within("ordered: g_col", _np(col.grad), g_col, grad_tol(g_col))
    g = oracle.texture_grad(tex_np, _np(uv), _np(col.grad), _np(uv_da), **kw)
    within("ordered: g_tex", _np(tex.grad), g["tex"], grad_tol(g["tex"]))
    # (eight million pixels: compared wherever the reference function is continuous within an ulp of uv / uv_da -- conftest.py)
    ok = ~discontinuous_pixels(oracle, tex_np, _np(uv), _np(col.grad), _np(uv_da), kw)
    assert (~ok).mean() <= 1e-4, (~ok).sum()
    within("ordered: g_uv", _np(uv.grad), g["uv"], grad_tol(g["uv"]), where=ok)
    within("ordered: g_uv_da", _np(uv_da.grad), g["uv_da"], grad_tol(g["uv_da"]), where=ok)
    ga, gr, grdb = oracle.interpolate_grad(uvattr, rh, b["tri"], _np(uv.grad), _np(rast_db), _np(uv_da.grad), "all")
    within("ordered: g_uvattr", _np(uva.grad), ga, grad_tol(ga))
    gp = oracle.rasterize_grad(b["pos"], b["tri"], rh, gr, grdb) + g_pos_aa
    within("ordered: g_pos", _np(pos.grad), gp, grad_tol(gp, 2))


def test_separate_backward_kernels_on_an_ordered_launch(dr, raw_oracle):
    """interpolate_grad and rasterize_grad (the two-kernel path: fused backward switched off) with the flags of a 2048-bin batch."""
    oracle = raw_oracle
    n, res = 32, (512, 512)
    b = m10k_batch(n, seed=7, nx=24, ny=14)
    rng = np.random.default_rng(5)
    attr = rng.uniform(-1, 1, size=(b["pos"].shape[1], 4)).astype(np.float32)
    dy = rng.normal(size=(n,) + res + (4,)).astype(np.float32)
    ctx = dr.RasterizeCudaContext()
    rast, _ = dr.rasterize(ctx, _t(b["pos"]), _t(b["tri"]), res)
    flags = _plugin.flags_of(rast)
    g_attr, g_rast = _plugin.interpolate_grad(_t(attr), rast, _t(b["tri"]), _t(dy), tile_flags=flags)
    g_pos = _plugin.rasterize_grad(_t(b["pos"]), _t(b["tri"]), rast, g_rast, tile_flags=flags)
    rh = _np(rast)
    ga, gr, _ = oracle.interpolate_grad(attr, rh, b["tri"], dy)
    within("ordered, separate: g_attr", _np(g_attr), ga, grad_tol(ga))
    within("ordered, separate: g_rast", _np(g_rast), gr, grad_tol(gr))
    gp = oracle.rasterize_grad(b["pos"], b["tri"], rh, _np(g_rast))
    within("ordered, separate: g_pos", _np(g_pos), gp, grad_tol(gp, BIG_SUM_FACTOR))      # (batch-scale sums: conftest.py)
    # and the same as the same kernels walking the image (no flags at all): only the order of the f32 atomics between blocks
    # differs, a tenth of the bar
    a2, r2 = _plugin.interpolate_grad(_t(attr), rast, _t(b["tri"]), _t(dy), tile_flags=False)
    p2 = _plugin.rasterize_grad(_t(b["pos"]), _t(b["tri"]), rast, g_rast, tile_flags=False)
    assert torch.equal(r2, g_rast)
    within("ordered vs image order: g_attr", _np(g_attr), _np(a2), 0.1 * grad_tol(ga))
    within("ordered vs image order: g_pos", _np(g_pos), _np(p2), 0.1 * grad_tol(gp))


@pytest.mark.parametrize("case", ["nothing visible", "every bin covered"])
def test_orders_without_a_second_part(dr, raw_oracle, case):
    """The two degenerate orders of a 2048-bin batch: no bin with a covered tile (all geometry outside the viewport), and no bin
    without one (one quad over the whole image): every consumer still visits every pixel exactly once."""
    oracle = raw_oracle
    n, res = 32, (512, 512)
    rng = np.random.default_rng(11)
    quad = np.array([[-1.2, -1.2, 0.1, 1], [1.2, -1.2, 0.3, 1], [1.2, 1.2, 0.5, 1], [-1.2, 1.2, 0.2, 1]], np.float32)
    if case == "nothing visible":
        quad[:, 0] += 5.0
    pos_np = np.repeat(quad[None], n, 0).copy()
    pos_np[:, :, :2] *= rng.uniform(0.95, 1.05, size=(n, 1, 1)).astype(np.float32)
    tri_np = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    attr_np = rng.uniform(-1, 1, size=(4, 2)).astype(np.float32)
    tex_np = rng.uniform(size=(1, 32, 32, 3)).astype(np.float32)
    ctx = dr.RasterizeCudaContext()
    pos = _t(pos_np).requires_grad_(True)
    attr = _t(attr_np).requires_grad_(True)
    tex = _t(tex_np).requires_grad_(True)
    tri = _t(tri_np)
    rast, rast_db = dr.rasterize(ctx, pos, tri, res)
    grid = _plugin.tile_flags_grid(_plugin.flags_of(rast), n, *res)
    assert int(grid.sum()) == (0 if case == "nothing visible" else grid.numel())
    uv, uv_da = dr.interpolate(attr, rast, tri, rast_db=rast_db, diff_attrs="all")
    uv.retain_grad(); uv_da.retain_grad()
    col = dr.texture(tex, uv, uv_da, filter_mode="linear-mipmap-linear")
    col.retain_grad()
    aa = dr.antialias(col, rast, pos, tri)
    dy = rng.normal(size=aa.shape).astype(np.float32)
    aa.backward(_t(dy))
    rh = _np(rast)
    ro, _ = oracle.rasterize(pos_np, tri_np, res)
    assert (rh[..., 3] != ro[..., 3]).sum() == 0
    uvo, uvdao = oracle.interpolate(attr_np, rh, tri_np, _np(rast_db), "all")
    within("degenerate order: uv", _np(uv), uvo, ATOL)
    within("degenerate order: uv_da", _np(uv_da), uvdao, grad_tol(uvdao))
    kw = dict(filter_mode="linear-mipmap-linear")
    within("degenerate order: texture", _np(col), oracle.texture(tex_np, _np(uv), _np(uv_da), **kw), ATOL)
    within("degenerate order: antialias", _np(aa), oracle.antialias(_np(col), rh, pos_np, tri_np), ATOL)
    g_col, g_pos_aa = oracle.antialias_grad(_np(col), rh, pos_np, tri_np, dy)
    within("degenerate order: g_col", _np(col.grad), g_col, grad_tol(g_col))
    g = oracle.texture_grad(tex_np, _np(uv), _np(col.grad), _np(uv_da), **kw)
    within("degenerate order: g_tex", _np(tex.grad), g["tex"], grad_tol(g["tex"], 2))     # eight million terms on a handful of texels
    ok = ~discontinuous_pixels(oracle, tex_np, _np(uv), _np(col.grad), _np(uv_da), kw)
    assert (~ok).mean() <= 1e-4, (~ok).sum()
    within("degenerate order: g_uv", _np(uv.grad), g["uv"], grad_tol(g["uv"]), where=ok)
    within("degenerate order: g_uv_da", _np(uv_da.grad), g["uv_da"], grad_tol(g["uv_da"]), where=ok)
    ga, gr, grdb = oracle.interpolate_grad(attr_np, rh, tri_np, _np(uv.grad), _np(rast_db), _np(uv_da.grad), "all")
    within("degenerate order: g_attr", _np(attr.grad), ga, grad_tol(ga, 2))
    gp = oracle.rasterize_grad(pos_np, tri_np, rh, gr, grdb) + g_pos_aa
    within("degenerate order: g_pos", _np(pos.grad), gp, grad_tol(gp, 2))
