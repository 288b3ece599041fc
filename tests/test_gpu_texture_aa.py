"""GPU parity: HIP texture / antialias (through the C ABI) vs the CPU oracle, plus the whole
four-op chain (BASELINE config 3's op graph at a size the oracle finishes in seconds).

Bars (stated once in tests/conftest.py): sampled values within 1e-5 abs; gradients within 1e-5 * max(1, |g|_inf); a
chain of four ops compared end to end k = 4 times that, colours at the end of the chain 2e-5.  No element is exempted."""
import numpy as np
import pytest
import torch
from conftest import CHAIN_OPS, CHAIN_VALUE_TOL, grad_tol, within

from nvdiffrast_amd.utils import m10k_batch

pytestmark = pytest.mark.gpu
ATOL = 1e-5


def _t(a, dev="cuda"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _tol(ref):
    return ATOL * max(1.0, float(np.abs(ref).max()))


def _close(a, b, tol):
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    assert d.max() <= tol, (d.max(), tol)


FILTERS = ["nearest", "linear", "linear-mipmap-nearest", "linear-mipmap-linear"]


@pytest.mark.parametrize("bm", ["wrap", "clamp", "zero"])
@pytest.mark.parametrize("fm", FILTERS)
@pytest.mark.parametrize("C,tex_n", [(1, 1), (2, 2), (3, 1), (4, 2), (5, 1)])
def test_texture_forward_backward(dr, oracle, fm, bm, C, tex_n):
    rng = np.random.default_rng(100 + C)
    N, H, W = 2, 37, 29
    tex = rng.uniform(size=(tex_n, 32, 64, C)).astype(np.float32)
    uv = rng.uniform(-0.3, 1.3, size=(N, H, W, 2)).astype(np.float32)
    mip = "mipmap" in fm
    uv_da = (rng.normal(size=(N, H, W, 4)) * 0.05).astype(np.float32) if mip else None
    bias = rng.uniform(-0.5, 0.5, size=(N, H, W)).astype(np.float32) if mip else None
    dy = rng.normal(size=(N, H, W, C)).astype(np.float32)
    dy[0, :3] = 0.0                                          # all-zero upstream gradient rows take the early-out
    kw = dict(filter_mode=fm, boundary_mode=bm)

    t_tex = _t(tex).requires_grad_(True)
    t_uv = _t(uv).requires_grad_(True)
    t_da = _t(uv_da).requires_grad_(True) if mip else None
    t_bias = _t(bias).requires_grad_(True) if mip else None
    out = dr.texture(t_tex, t_uv, t_da, t_bias, **kw)
    out.backward(_t(dy))

    oo = oracle.texture(tex, uv, uv_da, bias, **kw)
    g = oracle.texture_grad(tex, uv, dy, uv_da, bias, **kw)
    _close(out.detach().cpu().numpy(), oo, ATOL)
    _close(t_tex.grad.cpu().numpy(), g["tex"], _tol(g["tex"]))
    if fm == "nearest":
        assert t_uv.grad is None or float(t_uv.grad.abs().max()) == 0.0
    else:
        _close(t_uv.grad.cpu().numpy(), g["uv"], _tol(g["uv"]))
    if fm == "linear-mipmap-linear":
        _close(t_da.grad.cpu().numpy(), g["uv_da"], _tol(g["uv_da"]))
        _close(t_bias.grad.cpu().numpy(), g["mip_level_bias"], _tol(g["mip_level_bias"]))
    elif mip:
        assert t_da.grad is None and t_bias.grad is None


def test_bias_only_and_uvda_only(dr, oracle):
    rng = np.random.default_rng(7)
    tex = rng.uniform(size=(1, 64, 64, 3)).astype(np.float32)
    uv = rng.uniform(size=(2, 16, 16, 2)).astype(np.float32)
    uv_da = (rng.normal(size=(2, 16, 16, 4)) * 0.04).astype(np.float32)
    bias = rng.uniform(0.0, 4.0, size=(2, 16, 16)).astype(np.float32)
    for da, b in ((uv_da, None), (None, bias)):
        for fm in ("linear-mipmap-nearest", "linear-mipmap-linear"):
            o = dr.texture(_t(tex), _t(uv), None if da is None else _t(da), None if b is None else _t(b), filter_mode=fm)
            _close(o.cpu().numpy(), oracle.texture(tex, uv, da, b, filter_mode=fm), ATOL)
    # max_mip_level limits the chain; 0 degrades to plain bilinear (ops.py:411-412)
    o = dr.texture(_t(tex), _t(uv), _t(uv_da), filter_mode="linear-mipmap-linear", max_mip_level=2)
    _close(o.cpu().numpy(), oracle.texture(tex, uv, uv_da, filter_mode="linear-mipmap-linear", max_mip_level=2), ATOL)
    o = dr.texture(_t(tex), _t(uv), _t(uv_da), filter_mode="linear-mipmap-linear", max_mip_level=0)
    _close(o.cpu().numpy(), oracle.texture(tex, uv, filter_mode="linear"), ATOL)


def test_mip_construction_and_reuse(dr, oracle):
    rng = np.random.default_rng(8)
    for shape in [(2, 64, 16, 3), (1, 8, 128, 4), (1, 2, 2, 1)]:
        tex = rng.uniform(size=shape).astype(np.float32)
        w = dr.texture_construct_mip(_t(tex))
        levels = oracle.texture_build_mip(tex)
        flat = np.concatenate([l.reshape(-1) for l in levels]) if levels else np.zeros(0, np.float32)
        assert w.mip.numel() == flat.size
        assert np.abs(w.mip.cpu().numpy() - flat).max() <= 1e-6
    tex = rng.uniform(size=(1, 32, 32, 2)).astype(np.float32)
    uv = rng.uniform(size=(1, 9, 9, 2)).astype(np.float32)
    da = (rng.normal(size=(1, 9, 9, 4)) * 0.06).astype(np.float32)
    w = dr.texture_construct_mip(_t(tex))
    a = dr.texture(_t(tex), _t(uv), _t(da), mip=w)
    b = dr.texture(_t(tex), _t(uv), _t(da))
    assert torch.equal(a, b)
    with pytest.raises(RuntimeError):
        dr.texture_construct_mip(_t(rng.uniform(size=(1, 12, 8, 1)).astype(np.float32)))     # 12 -> 6 -> 3: odd
    with pytest.raises(RuntimeError, match="mip does not match texture size"):
        dr.texture(_t(rng.uniform(size=(1, 16, 16, 2)).astype(np.float32)), _t(uv), _t(da), mip=w)
    with pytest.raises(RuntimeError, match="square in cube map mode"):
        dr.texture(torch.zeros(1, 6, 4, 8, 3, device="cuda"), torch.zeros(1, 2, 2, 3, device="cuda"), boundary_mode="cube")


@pytest.mark.parametrize("shape", [(1, 1, 4096, 4), (1, 2, 4096, 3), (1, 4096, 1, 4), (2, 2, 2048, 2), (1, 1, 8192, 1),
                                   (1, 4, 1024, 3), (3, 64, 64, 1), (1, 1, 2, 1), (1, 128, 2, 4)])
def test_mip_chain_of_thin_textures(dr, oracle, shape):
    """Once one extent has reached 1 a level is HALF of the one before, not a quarter (texture.cpp:77-98): the launch that
    builds the small levels at the end of the chain in LDS must size its two buffers for that (1 x 4096 x 4: level 1 has
    8192 elements, level 2 has 4096).  Every level against the oracle, and the mipmapped sample / gradient through them."""
    rng = np.random.default_rng(sum(shape))
    tex = rng.uniform(size=shape).astype(np.float32)
    w = dr.texture_construct_mip(_t(tex))
    levels = oracle.texture_build_mip(tex)
    flat = np.concatenate([l.reshape(-1) for l in levels])
    assert w.mip.numel() == flat.size
    got = w.mip.cpu().numpy()
    off = 0
    for k, l in enumerate(levels, start=1):
        assert np.abs(got[off:off + l.size] - l.reshape(-1)).max() <= 1e-6, ("mip level", k, l.shape)
        off += l.size
    n = shape[0]
    uv = rng.uniform(size=(n, 8, 8, 2)).astype(np.float32)
    bias = rng.uniform(0.0, len(levels), size=(n, 8, 8)).astype(np.float32)
    t_tex = _t(tex).requires_grad_(True)
    out = dr.texture(t_tex, _t(uv), None, _t(bias), filter_mode="linear-mipmap-linear")
    dy = rng.normal(size=tuple(out.shape)).astype(np.float32)
    out.backward(_t(dy))
    _close(out.detach().cpu().numpy(), oracle.texture(tex, uv, None, bias, filter_mode="linear-mipmap-linear"), ATOL)
    g = oracle.texture_grad(tex, uv, dy, None, bias, filter_mode="linear-mipmap-linear")
    _close(t_tex.grad.cpu().numpy(), g["tex"], _tol(g["tex"]))


def test_custom_mip_stack_gradients(dr, oracle):
    rng = np.random.default_rng(9)
    tex = rng.uniform(size=(1, 16, 16, 2)).astype(np.float32)
    levels = [rng.uniform(size=(1, 16 >> k, 16 >> k, 2)).astype(np.float32) for k in range(1, 4)]
    uv = rng.uniform(size=(2, 11, 13, 2)).astype(np.float32)
    bias = rng.uniform(0.1, 2.9, size=(2, 11, 13)).astype(np.float32)
    dy = rng.normal(size=(2, 11, 13, 2)).astype(np.float32)
    t_tex = _t(tex).requires_grad_(True)
    t_lv = [_t(l).requires_grad_(True) for l in levels]
    out = dr.texture(t_tex, _t(uv), None, _t(bias), mip=t_lv, filter_mode="linear-mipmap-linear")
    out.backward(_t(dy))
    g = oracle.texture_grad(tex, uv, dy, None, bias, mip=levels, filter_mode="linear-mipmap-linear")
    _close(out.detach().cpu().numpy(), oracle.texture(tex, uv, None, bias, mip=levels, filter_mode="linear-mipmap-linear"), ATOL)
    _close(t_tex.grad.cpu().numpy(), g["tex"], _tol(g["tex"]))
    for k in range(3):
        _close(t_lv[k].grad.cpu().numpy(), g["mip"][k], _tol(g["mip"][k]))


# ------------------------------------------------------------------------------ cube maps

def _cube_dirs(rng, N, H, W):
    v = rng.normal(size=(N, H, W, 3)).astype(np.float32)
    v[0, 0] = np.array([1, 0.98, 0.1]) + rng.normal(size=(W, 3)) * 0.04       # hugging an edge
    v[0, 1] = np.array([1, -1, 1]) + rng.normal(size=(W, 3)) * 0.03           # hugging a corner
    v[0, 2] = np.array([-1, -1, -1]) + rng.normal(size=(W, 3)) * 0.03
    v[0, 3, 0] = 0.0                                                          # invalid direction
    return v.astype(np.float32)


@pytest.mark.parametrize("fm", FILTERS)
@pytest.mark.parametrize("C,tex_n", [(3, 1), (4, 2), (1, 1), (2, 2), (5, 1)])    # every k_tex_fwd_cube<.., C_CT> variant
def test_cube_forward_backward(dr, oracle, fm, C, tex_n):
    rng = np.random.default_rng(200 + C)
    N, H, W = 2, 23, 19
    tex = rng.uniform(size=(tex_n, 6, 16, 16, C)).astype(np.float32)
    v = _cube_dirs(rng, N, H, W)
    mip = "mipmap" in fm
    da = (rng.normal(size=(N, H, W, 6)) * 0.2).astype(np.float32) if mip else None
    bias = rng.uniform(-0.5, 0.5, size=(N, H, W)).astype(np.float32) if mip else None
    dy = rng.normal(size=(N, H, W, C)).astype(np.float32)
    dy[1, :2] = 0.0
    kw = dict(filter_mode=fm, boundary_mode="cube")
    t_tex = _t(tex).requires_grad_(True)
    t_v = _t(v).requires_grad_(True)
    t_da = _t(da).requires_grad_(True) if mip else None
    t_bias = _t(bias).requires_grad_(True) if mip else None
    out = dr.texture(t_tex, t_v, t_da, t_bias, **kw)
    out.backward(_t(dy))
    oo = oracle.texture(tex, v, da, bias, **kw)
    g = oracle.texture_grad(tex, v, dy, da, bias, **kw)
    _close(out.detach().cpu().numpy(), oo, ATOL)
    _close(t_tex.grad.cpu().numpy(), g["tex"], _tol(g["tex"]))
    if fm != "nearest":
        _close(t_v.grad.cpu().numpy(), g["uv"], _tol(g["uv"]))
    if fm == "linear-mipmap-linear":
        _close(t_da.grad.cpu().numpy(), g["uv_da"], _tol(g["uv_da"]))
        _close(t_bias.grad.cpu().numpy(), g["mip_level_bias"], _tol(g["mip_level_bias"]))


def test_cube_mips(dr, oracle):
    rng = np.random.default_rng(210)
    tex = rng.uniform(size=(2, 6, 8, 8, 3)).astype(np.float32)
    w = dr.texture_construct_mip(_t(tex), cube_mode=True)
    flat = np.concatenate([l.reshape(-1) for l in oracle.texture_build_mip(tex)])
    assert np.abs(w.mip.cpu().numpy() - flat).max() <= 1e-6
    v = _cube_dirs(rng, 2, 9, 9)
    da = (rng.normal(size=(2, 9, 9, 6)) * 0.3).astype(np.float32)
    a = dr.texture(_t(tex), _t(v), _t(da), mip=w, boundary_mode="cube")
    b = dr.texture(_t(tex), _t(v), _t(da), boundary_mode="cube")
    assert torch.equal(a, b)
    # custom stack with its own gradients
    levels = [rng.uniform(size=(2, 6, 8 >> k, 8 >> k, 3)).astype(np.float32) for k in (1, 2)]
    t_lv = [_t(l).requires_grad_(True) for l in levels]
    dy = rng.normal(size=(2, 9, 9, 3)).astype(np.float32)
    o = dr.texture(_t(tex), _t(v), _t(da), mip=t_lv, boundary_mode="cube")
    o.backward(_t(dy))
    g = oracle.texture_grad(tex, v, dy, da, mip=levels, boundary_mode="cube")
    for k in range(2):
        _close(t_lv[k].grad.cpu().numpy(), g["mip"][k], _tol(g["mip"][k]))


# ------------------------------------------------------------------------------ antialias

def _scene(N=2, res=(96, 128), seed=31):
    b = m10k_batch(N, seed=seed, nx=24, ny=12)
    return b, res


def test_antialias_forward_backward(dr, oracle):
    b, res = _scene()
    rng = np.random.default_rng(1)
    ro, _ = oracle.rasterize(b["pos"], b["tri"], res)
    color = rng.uniform(size=(2,) + res + (3,)).astype(np.float32)
    dy = rng.normal(size=color.shape).astype(np.float32)
    t_col = _t(color).requires_grad_(True)
    t_pos = _t(b["pos"]).requires_grad_(True)
    tri = _t(b["tri"])
    out = dr.antialias(t_col, _t(ro), t_pos, tri)
    out.backward(_t(dy))
    oo = oracle.antialias(color, ro, b["pos"], b["tri"])
    gc, gp = oracle.antialias_grad(color, ro, b["pos"], b["tri"], dy)
    assert (oo != color).any(-1).sum() > 200                      # the scene has silhouettes
    _close(out.detach().cpu().numpy(), oo, ATOL)
    _close(t_col.grad.cpu().numpy(), gc, _tol(gc))
    _close(t_pos.grad.cpu().numpy(), gp, _tol(gp))
    # prebuilt topology hash and gradient boost
    h = dr.antialias_construct_topology_hash(tri)
    t_pos2 = _t(b["pos"]).requires_grad_(True)
    out2 = dr.antialias(_t(color), _t(ro), t_pos2, tri, topology_hash=h, pos_gradient_boost=3.0)
    out2.backward(_t(dy))
    _close(out2.detach().cpu().numpy(), oo, ATOL)
    within("antialias boosted g_pos", t_pos2.grad.cpu().numpy(), 3.0 * gp, grad_tol(3.0 * gp))     # the bar of the boosted gradient itself


def test_antialias_range_mode_and_split_vertices(dr, oracle):
    # shared [V,4] positions (range mode) and a mesh whose triangles do not share vertex indices
    # (0.71, not 0.7: with 0.7 the top edge lies exactly on a pixel boundary, where the reference's blend weight is
    # +-0 or +-2^-22 depending on one rounding and its position gradient -- which does not scale with the weight --
    # jumps accordingly; see DESIGN.md "knife-edge silhouettes" and tests/test_ref_pins_oracle.py)
    pos = np.array([[-0.71, -0.71, 0, 1], [0.71, -0.71, 0.2, 1], [0.71, 0.71, 0, 1], [-0.71, -0.71, 0, 1], [0.71, 0.71, 0, 1], [-0.71, 0.71, -0.1, 1]], np.float32)
    tri = np.array([[0, 1, 2], [3, 4, 5]], np.int32)
    ranges = np.array([[0, 2], [1, 1]], np.int32)
    res = (40, 56)
    ro, _ = oracle.rasterize(pos, tri, res, ranges=ranges)
    rng = np.random.default_rng(2)
    color = rng.uniform(size=(2,) + res + (4,)).astype(np.float32)
    dy = rng.normal(size=color.shape).astype(np.float32)
    t_pos = _t(pos).requires_grad_(True)
    out = dr.antialias(_t(color), _t(ro), t_pos, _t(tri))
    out.backward(_t(dy))
    _close(out.detach().cpu().numpy(), oracle.antialias(color, ro, pos, tri), ATOL)
    gc, gp = oracle.antialias_grad(color, ro, pos, tri, dy)
    _close(t_pos.grad.cpu().numpy(), gp, _tol(gp))


def test_full_chain_config3(dr, oracle):
    """rasterize -> interpolate(uv, diff_attrs='all') -> texture(trilinear) -> antialias, forward and
    backward, against the oracle chain (BASELINE config 3's graph, reduced size)."""
    N, res = 2, (128, 128)
    b = m10k_batch(N, seed=41, nx=40, ny=20)
    rng = np.random.default_rng(3)
    tex = rng.uniform(size=(1, 256, 256, 3)).astype(np.float32)
    G = rng.normal(size=(N,) + res + (3,)).astype(np.float32)
    pos = _t(b["pos"]).requires_grad_(True)
    uvattr = _t(b["uv"]).requires_grad_(True)
    t_tex = _t(tex).requires_grad_(True)
    tri = _t(b["tri"])
    ctx = dr.RasterizeCudaContext()
    rast, rast_db = dr.rasterize(ctx, pos, tri, res)
    uv, uv_da = dr.interpolate(uvattr, rast, tri, rast_db=rast_db, diff_attrs="all")
    col = dr.texture(t_tex, uv, uv_da, filter_mode="linear-mipmap-linear")
    out = dr.antialias(col, rast, pos, tri)
    out.backward(_t(G))

    ro, rdbo = oracle.rasterize(b["pos"], b["tri"], res)
    uvo, uvdao = oracle.interpolate(b["uv"], ro, b["tri"], rast_db=rdbo, diff_attrs="all")
    colo = oracle.texture(tex, uvo, uvdao, filter_mode="linear-mipmap-linear")
    outo = oracle.antialias(colo, ro, b["pos"], b["tri"])
    assert (rast[..., 3].detach().cpu().numpy() != ro[..., 3]).sum() == 0
    _close(uv_da.detach().cpu().numpy(), uvdao, _tol(uvdao))
    within("chain vs oracle: col", col.detach().cpu().numpy(), colo, CHAIN_VALUE_TOL)       # end of a forward chain (conftest.py)
    within("chain vs oracle: aa", out.detach().cpu().numpy(), outo, CHAIN_VALUE_TOL)

    g_col, g_pos_aa = oracle.antialias_grad(colo, ro, b["pos"], b["tri"], G)
    tg = oracle.texture_grad(tex, uvo, g_col, uvdao, filter_mode="linear-mipmap-linear")
    g_uvattr, g_rast, g_rdb = oracle.interpolate_grad(b["uv"], ro, b["tri"], tg["uv"], rast_db=rdbo, dda=tg["uv_da"], diff_attrs="all")
    g_pos = oracle.rasterize_grad(b["pos"], b["tri"], ro, g_rast, ddb=g_rdb) + g_pos_aa
    _close(t_tex.grad.cpu().numpy(), tg["tex"], _tol(tg["tex"]))
    within("chain vs oracle: g_uvattr", uvattr.grad.cpu().numpy(), g_uvattr, grad_tol(g_uvattr, CHAIN_OPS))   # four ops end to end
    within("chain vs oracle: g_pos", pos.grad.cpu().numpy(), g_pos, grad_tol(g_pos, CHAIN_OPS))


@pytest.mark.parametrize("bm", ["wrap", "clamp", "zero"])
@pytest.mark.parametrize("fm", ["linear", "linear-mipmap-nearest", "linear-mipmap-linear"])
def test_texture_backward_on_constant_uv_regions(dr, oracle, fm, bm):
    """Background of a render: interpolate() gives uv = 0 and uv_da = 0 there, every pixel of a wave hits the same four
    texels.  Whole waves of such pixels take k_tex_grad's uniform-wave path; mixed waves (region borders, a second
    constant value, a non-zero footprint, zero upstream gradients) must fall back to the general one."""
    rng = np.random.default_rng(321)
    N, H, W, C = 2, 48, 80, 3
    tex = rng.uniform(size=(1, 32, 64, C)).astype(np.float32)
    uv = rng.uniform(-0.2, 1.2, size=(N, H, W, 2)).astype(np.float32)
    mip = "mipmap" in fm
    uv_da = (rng.normal(size=(N, H, W, 4)) * 0.05).astype(np.float32)
    uv[0, :, :48] = 0.0; uv_da[0, :, :48] = 0.0                    # background, texel corner (0,0): all four wrap taps
    uv[1, 8:40, 16:64] = np.array([0.37, 0.61], np.float32); uv_da[1, 8:40, 16:64] = 0.0   # a flat region elsewhere
    uv[1, :8, :32] = 0.0                                            # constant uv but a NON-zero footprint: general path
    bias = rng.uniform(-0.5, 0.5, size=(N, H, W)).astype(np.float32) if fm == "linear-mipmap-linear" else None
    dy = rng.normal(size=(N, H, W, C)).astype(np.float32)
    dy[0, 16:24, :16] = 0.0                                         # zero upstream gradients inside the background
    kw = dict(filter_mode=fm, boundary_mode=bm)
    t_tex = _t(tex).requires_grad_(True)
    t_uv = _t(uv).requires_grad_(True)
    t_da = _t(uv_da).requires_grad_(True) if mip else None
    t_bias = _t(bias).requires_grad_(True) if bias is not None else None
    out = dr.texture(t_tex, t_uv, t_da, t_bias, **kw)
    out.backward(_t(dy))
    g = oracle.texture_grad(tex, uv, dy, uv_da if mip else None, bias, **kw)
    _close(t_tex.grad.cpu().numpy(), g["tex"], _tol(g["tex"]))
    _close(t_uv.grad.cpu().numpy(), g["uv"], _tol(g["uv"]))
    if fm == "linear-mipmap-linear":
        _close(t_da.grad.cpu().numpy(), g["uv_da"], _tol(g["uv_da"]))
        _close(t_bias.grad.cpu().numpy(), g["mip_level_bias"], _tol(g["mip_level_bias"]))


@pytest.mark.parametrize("bm", ["wrap", "clamp", "zero"])
@pytest.mark.parametrize("C,tex_n", [(3, 1), (4, 2), (1, 1), (6, 1)])
def test_texture_backward_two_level_reduction_of_constant_regions(dr, oracle, bm, C, tex_n):
    """k_tex_grad with caller scratch (include/nvdr_hip.h): waves whose pixels share one texel quad leave per-wave records,
    k_tex_grad_fold merges them.  Several constant regions with different quads (runs that change inside a fold wave's
    walk), region borders through waves, a region whose footprint lies outside a zero-boundary texture (first tap without
    a texel, or none at all), zero upstream gradients, and the result with the scratch withheld (one-level path) next to it."""
    from nvdiffrast_amd.torch import _plugin
    rng = np.random.default_rng(400 + C)
    N, H, W = max(2, tex_n), 176, 208
    tex = rng.uniform(size=(tex_n, 32, 64, C)).astype(np.float32)
    uv = rng.uniform(-0.2, 1.2, size=(N, H, W, 2)).astype(np.float32)
    uv_da = (rng.normal(size=(N, H, W, 4)) * 0.05).astype(np.float32)

    def flat(n, ys, xs, u, v):
        uv[n, ys, xs] = np.array([u, v], np.float32); uv_da[n, ys, xs] = 0.0
    flat(0, slice(0, 176), slice(0, 160), 0.0, 0.0)            # background: quad around texel corner (0, 0)
    flat(0, slice(40, 90), slice(30, 100), 0.37, 0.61)         # another constant inside it (borders cut through 8x8 wave tiles)
    flat(1, slice(0, 64), slice(0, 208), -0.004, 0.5)          # left of the texture: first tap has no texel under 'zero'
    flat(1, slice(64, 128), slice(0, 208), -0.7, -0.7)         # far outside: no tap has a texel under 'zero'
    flat(1, slice(128, 176), slice(16, 200), 0.999, 0.999)     # last texel: wraps / clamps / loses three taps
    dy = rng.normal(size=(N, H, W, C)).astype(np.float32)
    dy[0, 100:140, 10:70] = 0.0
    kw = dict(filter_mode="linear-mipmap-linear", boundary_mode=bm)
    g = oracle.texture_grad(tex, uv, dy, uv_da, None, **kw)

    def run():
        t_tex = _t(tex).requires_grad_(True)
        t_uv = _t(uv).requires_grad_(True)
        t_da = _t(uv_da).requires_grad_(True)
        dr.texture(t_tex, t_uv, t_da, **kw).backward(_t(dy))
        return t_tex.grad.cpu().numpy(), t_uv.grad.cpu().numpy(), t_da.grad.cpu().numpy()

    lib = _plugin._capi.load()
    blocks = 11 * 13 * N                                      # 16x16-pixel blocks: four records each + one byte each (256-B aligned parts)
    assert lib.nvdr_texture_grad_scratch_bytes(N, H, W, C) == (4 * blocks * (8 + C) * 4 + 255) // 256 * 256 + (blocks + 255) // 256 * 256
    got = run()
    within("two-level tex grad: g_tex", got[0], g["tex"], _tol(g["tex"]))
    within("two-level tex grad: g_uv", got[1], g["uv"], _tol(g["uv"]))
    within("two-level tex grad: g_uv_da", got[2], g["uv_da"], _tol(g["uv_da"]))
    # the one-level path (no scratch): same results up to the order of the f32 sums
    try:
        _plugin._TEX_GRAD_SCRATCH = False
        one = run()
    finally:
        _plugin._TEX_GRAD_SCRATCH = True
    within("one-level tex grad: g_tex", one[0], g["tex"], _tol(g["tex"]))
    # (the two paths run different kernels since round 5 -- the dense-window kernel sums a pixel's uv gradient level by level, the
    # general one channel by channel -- so their pixel gradients agree to rounding, not bit for bit)
    within("one-level tex grad: g_uv", one[1], g["uv"], _tol(g["uv"]))
    within("one-level tex grad: g_uv_da", one[2], g["uv_da"], _tol(g["uv_da"]))
