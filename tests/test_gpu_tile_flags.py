"""Tile occupancy flags (include/nvdr_hip.h `tile_flags`; nvdr_device.hpp TileFlags): the rasterizer notes per 8x8 tile
whether any pixel shows a triangle, and the kernels that read rast skip the empty tiles -- only while the rast tensor is,
untouched, the one rasterize() returned."""
import numpy as np
import pytest
import torch
from conftest import ATOL, grad_tol, within

from nvdiffrast_amd.torch import _plugin
from nvdiffrast_amd.utils import m10k_batch, stress_triangles

# These tests look INSIDE the Python host layer (the records on the tensors, the stand-in gradient, the discard rule): they run
# with that layer serving every call.  tests/test_gpu_host_layer.py asks the same questions of the compiled layer.
pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("python_host_layer")]


def _t(a, dev="cuda"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _expected_flags(rast):
    """max over each 8x8 tile of (id > 0), from the rast tensor itself."""
    occ = (rast[..., 3] > 0)
    n, h, w = occ.shape
    hp, wp = (h + 7) // 8 * 8, (w + 7) // 8 * 8
    pad = torch.zeros((n, hp, wp), dtype=torch.bool, device=occ.device)
    pad[:, :h, :w] = occ
    return pad.view(n, hp // 8, 8, wp // 8, 8).any(4).any(2).to(torch.uint8)


@pytest.mark.parametrize("res,n,kind", [((512, 512), 3, "mesh"), ((100, 77), 2, "mesh"), ((8, 2056), 1, "mesh"), ((2100, 300), 1, "mesh"),
                                        ((256, 256), 2, "soup"), ((64, 64), 40, "mesh"), ((520, 1030), 2, "strip"),
                                        ((520, 330), 45, "mesh"), ((64, 128), 1100, "mesh"),        # >= 2048 bins: with a work order
                                        ((64, 64), 4200, "mesh"), ((128, 192), 1500, "mesh")])      # > 4096 bins: the order made by two launches of several workgroups
def test_flags_describe_the_rast_tensor(dr, res, n, kind):
    """Every in-image tile is written exactly once -- by the bin's own workgroup, by the workgroup that clears an empty bin
    for it, by the last part of a shared bin -- and says whether the tile shows a triangle (incl. >2048 px tiled viewports,
    sizes that are not multiples of 8 or 64, and the depth-peeling instantiations)."""
    if kind == "soup":
        b = stress_triangles(n, T=400, res=res[0], seed=5)
        pos, tri = b["pos"], b["tri"]
    else:
        b = m10k_batch(n, seed=50 + n, nx=30, ny=16)
        pos, tri = b["pos"].copy(), b["tri"]
        if kind == "strip":
            pos[..., 0] *= 0.05                                   # thousands of triangles per bin: shared bins in a small launch
    ctx = dr.RasterizeCudaContext()
    t_pos, t_tri = _t(pos), _t(tri)
    rast, _ = dr.rasterize(ctx, t_pos, t_tri, res)
    from nvdiffrast_amd.torch import _plugin
    buf = ctx.cpp_wrapper.last_flags
    assert buf.dtype == torch.uint8 and buf.dim() == 1
    flags = _plugin.tile_flags_grid(buf, n, *res)
    want = _expected_flags(rast)
    assert torch.equal(flags, want), int((flags != want).sum())
    assert 0 < int(want.sum()) < want.numel()                     # both kinds of tile occur
    assert rast._nvdr_origin.flags_for(rast) is buf
    _check_order(buf, want, n, res)
    with dr.DepthPeeler(ctx, t_pos, t_tri, res) as peeler:
        for _ in range(2):
            r, _ = peeler.rasterize_next_layer()
            want = _expected_flags(r)
            assert torch.equal(_plugin.tile_flags_grid(ctx.cpp_wrapper.last_flags, n, *res), want)
            _check_order(ctx.cpp_wrapper.last_flags, want, n, res)


def _check_order(buf, tile_grid, n, res):
    """The work order behind the flags (include/nvdr_hip.h): every 64x64-pixel bin exactly once, those with a covered tile
    first and in image-major order, then the others, and the count of the former."""
    th, tw = tile_grid.shape[1:]
    by, bx = (res[0] + 63) // 64, (res[1] + 63) // 64
    nb = n * by * bx
    off = (n * th * tw + 15) // 16 * 16
    if not 2048 <= nb <= 65536 or max(res) > 2048:                # no order kept for this size (nvdr_device.hpp kOrderMinBins)
        assert buf.numel() == off
        return
    rc = (off + 4 * (nb + 1) + 7) // 8 * 8                        # behind the order: the rasterizer's byte per bin and tile row
    assert buf.numel() == rc + 8 * nb
    rows = buf[rc:].view(nb, 8).cpu().numpy()
    order = buf[off:].view(torch.int32).cpu().numpy()
    pad = torch.zeros((n, by * 8, bx * 8), dtype=torch.uint8, device=tile_grid.device)
    pad[:, :th, :tw] = tile_grid
    cov = pad.view(n, by, 8, bx, 8).amax(dim=(2, 4)).reshape(-1).cpu().numpy().astype(bool)
    ncov = int(cov.sum())
    want_rows = pad.view(n, by, 8, bx, 8).amax(dim=4).permute(0, 1, 3, 2).reshape(nb, 8).cpu().numpy()      # [bin][tile row]
    inside = (np.arange(by * 8).reshape(by, 8) < th)[None, :, None, :].repeat(n, 0).repeat(bx, 2).reshape(nb, 8)
    assert np.array_equal(rows[inside] != 0, want_rows[inside] != 0)
    assert order[nb] == ncov
    assert np.array_equal(order[:ncov], np.nonzero(cov)[0]) and np.array_equal(order[ncov:nb], np.nonzero(~cov)[0])


def test_consumers_skip_empty_tiles_only_for_the_untouched_rast(dr, oracle):
    """interpolate / antialias / the backward kernels through the operator layer (flags attached) against the oracle, and the
    same after an in-place edit of rast: the version counter moved, the flags are dropped, and the kernels read the edited
    tensor -- a triangle painted into a tile that the flags call empty must show up."""
    from nvdiffrast_amd import _capi
    from nvdiffrast_amd.torch import _plugin
    N, res = 2, (128, 192)
    b = m10k_batch(N, seed=61, nx=20, ny=10)
    pos_np = b["pos"].copy(); pos_np[..., :2] *= 0.6                         # plenty of empty tiles around the mesh
    rng = np.random.default_rng(6)
    G = rng.normal(size=(N,) + res + (4,)).astype(np.float32)
    tri = _t(b["tri"])
    ctx = dr.RasterizeCudaContext()
    pos = _t(pos_np).requires_grad_(True)
    attr = _t(b["attr"]).requires_grad_(True)
    rast, rast_db = dr.rasterize(ctx, pos, tri, res)
    flags = rast._nvdr_origin.flags_for(rast)                       # the buffer the consumers are handed
    assert flags is not None
    grid = _plugin.tile_flags_grid(flags, N, *res)
    assert int((grid == 0).sum()) > grid.numel() // 4
    out, out_da = dr.interpolate(attr, rast, tri, rast_db=rast_db, diff_attrs="all")
    col = torch.rand((N,) + res + (3,), device="cuda")
    aa = dr.antialias(col, rast, pos, tri)
    ((out * _t(G)).sum() + (out_da ** 2).sum() + (aa * aa).sum()).backward()
    ro, rdbo = oracle.rasterize(pos_np, b["tri"], res)
    oo, odao = oracle.interpolate(b["attr"], ro, b["tri"], rdbo, "all")
    within("flags: interpolate", out.detach().cpu().numpy(), oo, ATOL)
    within("flags: interpolate da", out_da.detach().cpu().numpy(), odao, grad_tol(odao))
    aao = oracle.antialias(col.cpu().numpy(), ro, pos_np, b["tri"])
    within("flags: antialias", aa.detach().cpu().numpy(), aao, ATOL)
    ga, gr, grdb = oracle.interpolate_grad(b["attr"], ro, b["tri"], G, rdbo, 2 * odao, "all")
    _gc, gpa = oracle.antialias_grad(col.cpu().numpy(), ro, pos_np, b["tri"], 2 * aao)
    gp = oracle.rasterize_grad(pos_np, b["tri"], ro, gr, grdb) + gpa
    within("flags: g_attr", attr.grad.cpu().numpy(), ga, grad_tol(ga))
    within("flags: g_pos", pos.grad.cpu().numpy(), gp, grad_tol(gp, 2))

    # the same kernels with and without the flags give identical bits where no atomics are involved
    a = _plugin.interpolate_fwd(attr.detach(), rast.detach(), tri, tile_flags=flags)[0]
    bb = _plugin.interpolate_fwd(attr.detach(), rast.detach(), tri, tile_flags=False)[0]
    assert torch.equal(a, bb)

    # an in-place edit: a triangle id painted into an empty tile
    empty = torch.nonzero(grid[0] == 0)[0]
    ty, tx = int(empty[0]), int(empty[1])
    r2, _ = dr.rasterize(ctx, _t(pos_np), tri, res)
    assert r2._nvdr_origin.flags_for(r2) is not None
    with torch.no_grad():
        r2[0, ty * 8 + 3, tx * 8 + 2] = torch.tensor([0.25, 0.5, 0.0, 7.0], device="cuda")
    assert r2._nvdr_origin.flags_for(r2) is None                             # version counter moved
    o2, _ = dr.interpolate(attr.detach(), r2, tri)
    want, _ = oracle.interpolate(b["attr"], r2.cpu().numpy(), b["tri"])
    assert np.abs(want[0, ty * 8 + 3, tx * 8 + 2]).max() > 0
    within("edited rast: interpolate", o2.cpu().numpy(), want, ATOL)
    # a copy of rast has no flags either (nothing is assumed about tensors the rasterizer did not hand out itself)
    assert getattr(rast.detach().clone(), "_nvdr_origin", None) is None


@pytest.mark.parametrize("res", [(96, 128), (104, 72)])          # (72 pixels: nine tile columns -- the light gradient kernel reads flag PAIRS from dwords)
@pytest.mark.parametrize("fm", ["nearest", "linear", "linear-mipmap-nearest", "linear-mipmap-linear"])
@pytest.mark.parametrize("bm", ["wrap", "clamp", "zero"])
def test_texture_takes_uv_of_empty_tiles_as_zero_only_while_it_is(dr, oracle, fm, bm, res):
    """interpolate()'s outputs carry the rasterizer's flags (zeros on empty tiles); texture() forward and backward then do
    not read uv / uv_da there.  Same results as the oracle on the full tensors; after an in-place edit of uv inside an empty
    tile the flags are dropped and the edit shows."""
    N = 2
    b = m10k_batch(N, seed=71, nx=20, ny=10, attrs=2)
    pos_np = b["pos"].copy(); pos_np[..., :2] *= 0.55
    rng = np.random.default_rng(8)
    tex_np = rng.uniform(size=(1, 64, 64, 3)).astype(np.float32)
    dy = rng.normal(size=(N,) + res + (3,)).astype(np.float32)
    tri = _t(b["tri"])
    ctx = dr.RasterizeCudaContext()
    mip = "mipmap" in fm
    ro, rdbo = oracle.rasterize(pos_np, b["tri"], res)
    uvo, uvdao = oracle.interpolate(b["uv"], ro, b["tri"], rdbo, "all")

    def run(edit=None):
        rast, rast_db = dr.rasterize(ctx, _t(pos_np), tri, res)
        uv, uv_da = dr.interpolate(_t(b["uv"]), rast, tri, rast_db=rast_db, diff_attrs="all")
        assert uv._nvdr_zero_tiles.flags is rast._nvdr_origin.flags and uv_da._nvdr_zero_tiles.flags is uv._nvdr_zero_tiles.flags
        if edit is not None:
            with torch.no_grad():
                edit(uv)
        uv.requires_grad_(True); uv_da.requires_grad_(True)
        tex = _t(tex_np).requires_grad_(True)
        col = dr.texture(tex, uv, uv_da if mip else None, filter_mode=fm, boundary_mode=bm)
        col.backward(_t(dy))
        return col, tex.grad, uv.grad, (uv_da.grad if fm == "linear-mipmap-linear" else None), rast

    col, g_tex, g_uv, g_da, rast = run()
    kw = dict(filter_mode=fm, boundary_mode=bm)
    want = oracle.texture(tex_np, uvo, uvdao if mip else None, **kw)
    g = oracle.texture_grad(tex_np, uvo, dy, uvdao if mip else None, **kw)
    within("zero tiles: texture", col.detach().cpu().numpy(), want, ATOL)
    within("zero tiles: g_tex", g_tex.cpu().numpy(), g["tex"], grad_tol(g["tex"]))
    if fm != "nearest":
        within("zero tiles: g_uv", g_uv.cpu().numpy(), g["uv"], grad_tol(g["uv"]))
    if g_da is not None:
        within("zero tiles: g_uv_da", g_da.cpu().numpy(), g["uv_da"], grad_tol(g["uv_da"]))

    e = torch.nonzero(_plugin.tile_flags_grid(rast._nvdr_origin.flags, *rast.shape[:3])[1] == 0)[0]
    ty, tx = int(e[0]), int(e[1])

    def paint(uv):
        uv[1, ty * 8 + 1, tx * 8 + 5] = torch.tensor([0.43, 0.27], device="cuda")
    col2 = run(paint)[0]
    uv2 = uvo.copy(); uv2[1, ty * 8 + 1, tx * 8 + 5] = (0.43, 0.27)
    want2 = oracle.texture(tex_np, uv2, uvdao if mip else None, **kw)
    assert np.abs(want2 - want).max() > 1e-3
    within("zero tiles dropped: texture", col2.detach().cpu().numpy(), want2, ATOL)


def test_views_of_the_rasterizers_outputs_are_ordinary_tensors(dr, oracle):
    """A view shares storage and version counter with the tensor rasterize() / interpolate() returned, but not its shape:
    slices and reshapes of rast and uv go through the consumers like any tensor (no flags, no fused gradient, no error)."""
    b = m10k_batch(3, seed=77, nx=20, ny=12)
    V = b["pos"].shape[1]
    rng = np.random.default_rng(2)
    uvattr = rng.uniform(0, 1, size=(V, 2)).astype(np.float32)
    tex = rng.uniform(0, 1, size=(1, 64, 64, 3)).astype(np.float32)
    ctx = dr.RasterizeCudaContext()
    t_pos = _t(b["pos"]).requires_grad_(True)
    t_tri, t_uv, t_tex = _t(b["tri"]), _t(uvattr), _t(tex)
    rast, _ = dr.rasterize(ctx, t_pos, t_tri, (64, 128))
    origin = rast._nvdr_origin
    first = rast[:1]                                               # same data_ptr, same version
    assert origin.flags_for(first) is None and origin.flags_for(rast.view(6, 32, 128, 4)) is None
    ro, _ = oracle.rasterize(b["pos"], b["tri"], (64, 128))
    uv1, _ = dr.interpolate(t_uv, first, t_tri)
    want, _ = oracle.interpolate(uvattr, ro[:1], b["tri"])
    within("view of rast: interpolate", uv1.detach().cpu().numpy(), want, ATOL)
    halves, _ = dr.interpolate(t_uv, rast.view(6, 32, 128, 4), t_tri)
    want_all, _ = oracle.interpolate(uvattr, ro, b["tri"])
    within("reshaped rast: interpolate", halves.detach().cpu().numpy().reshape(3, 64, 128, 2), want_all, ATOL)
    # a slice of interpolate's output carries no zero-tile record either
    uv, _ = dr.interpolate(t_uv, rast, t_tri)
    col = dr.texture(t_tex, uv[:2], filter_mode="linear")
    within("view of uv: texture", col.detach().cpu().numpy(), oracle.texture(tex, uv[:2].detach().cpu().numpy(), filter_mode="linear"), ATOL)
    # and the sliced graph differentiates (two-kernel path: the slice is not the rasterizer's own tensor)
    before = _plugin_counts()
    uv1.sum().backward()
    assert _plugin_counts()["used"] == before["used"] and t_pos.grad is not None and bool(torch.isfinite(t_pos.grad).all())


def _plugin_counts():
    from nvdiffrast_amd.torch import _plugin
    return dict(_plugin.fused_backward_count())


def test_the_plugin_finds_the_flags_itself(dr, oracle):
    """INTEGRATION.md section 1's literal binding -- the reference's ops.py calling `_plugin.interpolate_fwd(attr, rast, tri)` and
    friends with the reference's own argument lists -- gets the empty-tile skipping too: the record travels with the tensors
    rasterize_fwd_cuda / interpolate_fwd* returned.  Observed through the verification mode, which counts (and
    checks) every use of flags; a detached alias and autograd's copy of a saved output are recognised through the live owner."""
    N, res = 2, (128, 192)
    b = m10k_batch(N, seed=61, nx=20, ny=10, attrs=2)
    pos_np = b["pos"].copy(); pos_np[..., :2] *= 0.6
    rng = np.random.default_rng(3)
    tex_np = rng.uniform(size=(1, 64, 64, 3)).astype(np.float32)
    tri, pos, uvattr, tex = _t(b["tri"]), _t(pos_np), _t(b["uv"]), _t(tex_np)
    state = _plugin.RasterizeCRStateWrapper(0)
    no_ranges = torch.empty((0, 2), dtype=torch.int32)
    mips = _plugin.texture_construct_mip(tex, -1, False)
    _plugin.set_tile_flag_verification(True)
    try:
        n0 = _plugin.tile_flag_verifications()
        rast, rast_db = _plugin.rasterize_fwd_cuda(state, pos, tri, res, no_ranges, -1)
        uv, uv_da = _plugin.interpolate_fwd_da(uvattr, rast, tri, rast_db, True, [])
        col = _plugin.texture_fwd_mip(tex, uv, uv_da, torch.tensor([]), mips, [], 3, 1)
        topo = _plugin.antialias_construct_topology_hash(tri)
        aa, work = _plugin.antialias_fwd(col, rast, pos, tri, topo)
        dy = torch.randn_like(col)
        g = _plugin.texture_grad_linear_mipmap_linear(tex, uv, dy, uv_da, torch.tensor([]), mips, [], 3, 1)
        ga, gr, grdb = _plugin.interpolate_grad_da(uvattr, rast, tri, g[1], rast_db, g[2], True, [])
        gp = _plugin.rasterize_grad_db(pos, tri, rast.detach(), gr, grdb)           # an alias of rast: found through its live owner
        assert _plugin.tile_flag_verifications() - n0 == 1 + 2 + 1 + 2 + 1 + 1      # interpolate, texture (uv, uv_da), antialias, texture grad, interpolate grad, rasterize grad
        # the same values as with the flags handed over explicitly and as without any
        flags = state.last_flags
        _plugin.set_tile_flag_verification(False)
        uv_x, uvda_x = _plugin.interpolate_fwd_da(uvattr, rast, tri, rast_db, True, [], tile_flags=flags)
        uv_n, uvda_n = _plugin.interpolate_fwd_da(uvattr, rast, tri, rast_db, True, [], tile_flags=False)
        assert torch.equal(uv, uv_x) and torch.equal(uv, uv_n) and torch.equal(uv_da, uvda_n)
        col_n = _plugin.texture_fwd_mip(tex, uv_n, uvda_n, torch.tensor([]), mips, [], 3, 1)    # no record on these
        assert torch.equal(col, col_n)
        ro, rdbo = oracle.rasterize(pos_np, b["tri"], res)
        uvo, uvdao = oracle.interpolate(b["uv"], ro, b["tri"], rdbo, "all")
        within("plugin-level flags: uv", uv.cpu().numpy(), uvo, ATOL)
        gao, gro, grdbo = oracle.interpolate_grad(b["uv"], ro, b["tri"], g[1].cpu().numpy(), rdbo, g[2].cpu().numpy(), "all")
        within("plugin-level flags: g_rast", gr.cpu().numpy(), gro, grad_tol(gro))
        gpo = oracle.rasterize_grad(pos_np, b["tri"], ro, gr.cpu().numpy(), grdb.cpu().numpy())
        within("plugin-level flags: g_pos", gp.cpu().numpy(), gpo, grad_tol(gpo))
        # switched off altogether: nothing is looked up, nothing verified
        _plugin.set_tile_flag_verification(True)
        _plugin.set_tile_skipping(False)
        n1 = _plugin.tile_flag_verifications()
        uv_o, _ = _plugin.interpolate_fwd_da(uvattr, rast, tri, rast_db, True, [])
        assert _plugin.tile_flag_verifications() == n1 and torch.equal(uv_o, uv)
    finally:
        _plugin.set_tile_skipping(True)
        _plugin.set_tile_flag_verification(False)


def test_verification_mode_catches_writes_the_version_counter_does_not_see(dr):
    """The one deviation from the reference the flags carry: a write through `.data` (or an external kernel) leaves the
    version counter alone, so the record stays valid and the consumers skip tiles that are no longer empty.  The verification
    mode re-derives the flags from the tensor actually passed and refuses; `set_tile_skipping(False)` is the public way out."""
    N, res = 1, (64, 64)
    b = m10k_batch(N, seed=5, nx=10, ny=6)
    pos_np = b["pos"].copy(); pos_np[..., :2] *= 0.4
    ctx = dr.RasterizeCudaContext()
    tri, attr = _t(b["tri"]), _t(b["attr"])
    rast, _ = dr.rasterize(ctx, _t(pos_np), tri, res)
    grid = _plugin.tile_flags_grid(rast._nvdr_origin.flags, N, *res)
    e = torch.nonzero(grid[0] == 0)[0]
    ty, tx = int(e[0]), int(e[1])
    rast.data[0, ty * 8 + 2, tx * 8 + 2] = torch.tensor([0.2, 0.3, 0.0, 5.0], device="cuda")     # does not bump rast._version
    painted = (0, ty * 8 + 2, tx * 8 + 2)
    out_skipping, _ = dr.interpolate(attr, rast, tri)
    assert float(out_skipping[painted].abs().max()) == 0.0                   # the hazard: the painted pixel is skipped
    _plugin.set_tile_flag_verification(True)
    try:
        with pytest.raises(RuntimeError, match="tile flags disagree"):
            dr.interpolate(attr, rast, tri)
    finally:
        _plugin.set_tile_flag_verification(False)
    _plugin.set_tile_skipping(False)
    try:
        out_full, _ = dr.interpolate(attr, rast, tri)
    finally:
        _plugin.set_tile_skipping(True)
    assert float(out_full[painted].abs().max()) > 0.0                        # read like any tensor, as the reference would
