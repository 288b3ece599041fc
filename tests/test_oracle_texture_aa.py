"""The CPU oracle's texture and antialias restatements checked against independent statements of
the same maths (numpy bilinear sampling, box-filter mips, adjoint identities, central
differences, analytic coverage).  These property tests predate oracle/_ref; every oracle call in them is
now also cross-checked against the reference itself by the pinned `oracle` fixture.  No GPU needed."""
import numpy as np
import pytest


def _np_bilinear_wrap(tex, uv):
    n, h, w, c = tex.shape
    out = np.zeros(uv.shape[:3] + (c,), np.float64)
    for z in range(uv.shape[0]):
        t = tex[0 if n == 1 else z].astype(np.float64)
        u = uv[z, ..., 0].astype(np.float64); v = uv[z, ..., 1].astype(np.float64)
        u = (u - np.floor(u)) * w - 0.5
        v = (v - np.floor(v)) * h - 0.5
        iu, iv = np.floor(u).astype(int), np.floor(v).astype(int)
        fu, fv = (u - iu)[..., None], (v - iv)[..., None]
        a = t[iv % h, iu % w]; b = t[iv % h, (iu + 1) % w]
        c_ = t[(iv + 1) % h, iu % w]; d = t[(iv + 1) % h, (iu + 1) % w]
        out[z] = (a * (1 - fu) + b * fu) * (1 - fv) + (c_ * (1 - fu) + d * fu) * fv
    return out


def test_linear_wrap_matches_numpy_bilinear(oracle):
    rng = np.random.default_rng(1)
    tex = rng.uniform(size=(2, 16, 8, 3)).astype(np.float32)
    uv = rng.uniform(-1.5, 2.5, size=(2, 9, 7, 2)).astype(np.float32)
    o = oracle.texture(tex, uv, filter_mode="linear", boundary_mode="wrap")
    assert np.abs(o - _np_bilinear_wrap(tex, uv)).max() < 2e-5


def test_boundary_modes(oracle):
    tex = np.arange(16, dtype=np.float32).reshape(1, 4, 4, 1) + 1.0
    uv = np.array([[-0.3, 0.5], [1.2, 0.5], [0.5, -0.2], [0.5, 0.5]], np.float32).reshape(1, 1, 4, 2)
    z = oracle.texture(tex, uv, filter_mode="nearest", boundary_mode="zero")[0, 0, :, 0]
    assert z[0] == 0 and z[1] == 0 and z[2] == 0 and z[3] == tex[0, 2, 2, 0]
    c = oracle.texture(tex, uv, filter_mode="linear", boundary_mode="clamp")[0, 0, :, 0]
    # clamped to the centre of the border texel: u=-0.3 -> column 0, v=0.5 -> between rows 1 and 2
    assert np.isclose(c[0], 0.5 * (tex[0, 1, 0, 0] + tex[0, 2, 0, 0]))
    assert np.isclose(c[1], 0.5 * (tex[0, 1, 3, 0] + tex[0, 2, 3, 0]))
    w = oracle.texture(tex, uv, filter_mode="nearest", boundary_mode="wrap")[0, 0, :, 0]
    assert w[0] == tex[0, 2, 2, 0] and w[1] == tex[0, 2, 0, 0]        # -0.3 -> 0.7 -> col 2; 1.2 -> 0.2 -> col 0


def test_mip_chain_is_box_filter_and_odd_sizes_are_rejected(oracle):
    rng = np.random.default_rng(2)
    tex = rng.uniform(size=(2, 8, 32, 2)).astype(np.float32)
    L, lw, lh, off, total = oracle.texture_mip_info(tex.shape)
    assert L == 5 and lw == [32, 16, 8, 4, 2, 1] and lh == [8, 4, 2, 1, 1, 1]
    mips = oracle.texture_build_mip(tex)
    cur = tex.astype(np.float64)
    for m in mips:
        n, h, w, c = cur.shape
        cur = cur.reshape(n, max(h // 2, 1), 2 if h > 1 else 1, max(w // 2, 1), 2 if w > 1 else 1, c).mean((2, 4))
        assert m.shape == cur.shape and np.abs(m - cur).max() < 1e-6
    assert oracle.texture_mip_info((1, 8, 8, 1), max_mip_level=2)[0] == 2
    with pytest.raises(ValueError):
        oracle.texture_mip_info((1, 12, 8, 1))                          # 12 -> 6 -> 3 is odd


def test_mip_level_follows_the_footprint(oracle):
    # A texture whose level-k mip is the constant k: the trilinear result then IS the mip level.
    S = 64
    tex = np.zeros((1, S, S, 1), np.float32)
    L = 6
    levels = [np.full((1, S >> k, S >> k, 1), float(k), np.float32) for k in range(1, L + 1)]
    uv = np.full((1, 1, 5, 2), 0.37, np.float32)
    scales = np.array([0.5, 1.0, 2.0, 3.0, 8.0], np.float32) / S            # footprint in texels -> level log2(s)
    uv_da = np.zeros((1, 1, 5, 4), np.float32)
    uv_da[0, 0, :, 0] = scales; uv_da[0, 0, :, 3] = scales * 0.25            # major axis = du/dX
    o = oracle.texture(tex, uv, uv_da, mip=levels, filter_mode="linear-mipmap-linear")[0, 0, :, 0]
    assert np.allclose(o, np.maximum(np.log2(scales * S), 0.0), atol=1e-5)
    bias = np.full((1, 1, 5), 1.5, np.float32)
    o2 = oracle.texture(tex, uv, uv_da, bias, mip=levels, filter_mode="linear-mipmap-linear")[0, 0, :, 0]
    assert np.allclose(o2, np.clip(np.log2(scales * S) + 1.5, 0, L), atol=1e-5)
    o3 = oracle.texture(tex, uv, None, bias, mip=levels, filter_mode="linear-mipmap-nearest")[0, 0, :, 0]
    assert np.all(o3 == 1.0)                                                 # floor(1.5), texture_kernel.cu:577


@pytest.mark.parametrize("fm", ["nearest", "linear", "linear-mipmap-nearest", "linear-mipmap-linear"])
@pytest.mark.parametrize("bm", ["wrap", "clamp", "zero"])
def test_texture_gradients(oracle, fm, bm):
    rng = np.random.default_rng(3)
    tex = rng.uniform(size=(1, 16, 16, 2)).astype(np.float32)
    uv = rng.uniform(-0.2, 1.2, size=(2, 6, 6, 2)).astype(np.float32)
    mip = "mipmap" in fm
    uv_da = (rng.normal(size=(2, 6, 6, 4)) * 0.08).astype(np.float32) if mip else None
    bias = rng.uniform(-0.3, 0.3, size=(2, 6, 6)).astype(np.float32) if mip else None
    dy = rng.normal(size=(2, 6, 6, 2)).astype(np.float32)
    kw = dict(filter_mode=fm, boundary_mode=bm)
    f = lambda t=tex, u=uv, d=uv_da, b=bias: oracle.texture(t, u, d, b, **kw).astype(np.float64)
    g = oracle.texture_grad(tex, uv, dy, uv_da, bias, **kw)

    # The op is linear in the texture: <dy, f(tex + e)> - <dy, f(tex)> == <g_tex, e> exactly (up to rounding).
    e = rng.normal(size=tex.shape).astype(np.float32)
    lhs = ((f(t=tex + e) - f()) * dy).sum()
    assert np.isclose(lhs, (g["tex"].astype(np.float64) * e).sum(), rtol=2e-4, atol=2e-4)

    if fm == "nearest":
        assert g["uv"] is None
        return
    # Central differences for uv on pixels that stay inside one bilinear cell / mip level.
    eps = 1e-3
    fd = np.zeros_like(uv, dtype=np.float64)
    for k in range(2):
        up = uv.copy(); up[..., k] += eps
        um = uv.copy(); um[..., k] -= eps
        fd[..., k] = ((f(u=up) - f(u=um)) * dy).sum(-1) / (2 * eps)
    err = np.abs(fd - g["uv"])
    assert np.median(err) < 2e-2 * max(1.0, np.abs(g["uv"]).max())
    assert (err < 5e-2 * max(1.0, np.abs(g["uv"]).max())).mean() > 0.8      # kinks at texel borders excepted

    if fm == "linear-mipmap-linear":
        fdb = np.zeros_like(bias, dtype=np.float64)
        bp = bias + eps; bm_ = bias - eps
        fdb = ((f(b=bp) - f(b=bm_)) * dy).sum(-1) / (2 * eps)
        errb = np.abs(fdb - g["mip_level_bias"])
        assert (errb < 1e-2 * max(1.0, np.abs(fdb).max())).mean() > 0.85
        fda = np.zeros_like(uv_da, dtype=np.float64)
        for k in range(4):
            dp = uv_da.copy(); dp[..., k] += eps * 0.1
            dm = uv_da.copy(); dm[..., k] -= eps * 0.1
            fda[..., k] = ((f(d=dp) - f(d=dm)) * dy).sum(-1) / (2 * eps * 0.1)
        erra = np.abs(fda - g["uv_da"])
        assert (erra < 3e-2 * max(1.0, np.abs(fda).max())).mean() > 0.85
    else:
        assert g["uv_da"] is None and g["mip_level_bias"] is None


def test_custom_mip_stack_gets_its_own_gradients(oracle):
    rng = np.random.default_rng(4)
    tex = rng.uniform(size=(1, 8, 8, 1)).astype(np.float32)
    levels = [rng.uniform(size=(1, 8 >> k, 8 >> k, 1)).astype(np.float32) for k in range(1, 4)]
    uv = rng.uniform(size=(1, 5, 5, 2)).astype(np.float32)
    bias = rng.uniform(0.2, 2.5, size=(1, 5, 5)).astype(np.float32)
    dy = rng.normal(size=(1, 5, 5, 1)).astype(np.float32)
    g = oracle.texture_grad(tex, uv, dy, None, bias, mip=levels, filter_mode="linear-mipmap-linear")
    f = lambda lv: oracle.texture(tex, uv, None, bias, mip=lv, filter_mode="linear-mipmap-linear").astype(np.float64)
    for k in range(3):
        e = rng.normal(size=levels[k].shape).astype(np.float32)
        lv2 = [l.copy() for l in levels]; lv2[k] = lv2[k] + e
        assert np.isclose(((f(lv2) - f(levels)) * dy).sum(), (g["mip"][k].astype(np.float64) * e).sum(), rtol=1e-3, atol=1e-4)
    # with the internally built chain the level gradients are folded into g_tex instead
    g2 = oracle.texture_grad(tex, uv, dy, None, bias, filter_mode="linear-mipmap-linear")
    e = rng.normal(size=tex.shape).astype(np.float32)
    f2 = lambda t: oracle.texture(t, uv, None, bias, filter_mode="linear-mipmap-linear").astype(np.float64)
    assert g2["mip"] is None
    assert np.isclose(((f2(tex + e) - f2(tex)) * dy).sum(), (g2["tex"].astype(np.float64) * e).sum(), rtol=1e-3, atol=1e-4)


# ------------------------------------------------------------------------------ antialias

def _one_triangle(res=32):
    pos = np.array([[[-0.6, -0.7, 0, 1], [0.7, -0.5, 0, 1], [-0.2, 0.8, 0, 1]]], np.float32)
    tri = np.array([[0, 1, 2]], np.int32)
    return pos, tri, res


def test_antialias_coverage_matches_the_analytic_area(oracle):
    pos, tri, res = _one_triangle()
    r, _ = oracle.rasterize(pos, tri, (res, res))
    color = (r[..., 3:4] > 0).astype(np.float32)
    o = oracle.antialias(color, r, pos, tri)
    p = pos[0, :, :2].astype(np.float64) * res / 2
    area = 0.5 * abs((p[1, 0] - p[0, 0]) * (p[2, 1] - p[0, 1]) - (p[2, 0] - p[0, 0]) * (p[1, 1] - p[0, 1]))
    assert abs(color.sum() - area) > 1.0                     # aliased coverage is off by more than a pixel
    assert abs(o.sum() - area) < 0.6                         # the blended one is not
    changed = (o != color).any(-1)
    assert 20 < changed.sum() < 4 * 3 * res and o.min() >= 0.0 and o.max() <= 1.0


def test_antialias_leaves_interior_edges_alone(oracle):
    # Two coplanar triangles sharing an edge: the shared edge is not a silhouette.
    pos = np.array([[[-0.7, -0.7, 0, 1], [0.7, -0.7, 0, 1], [0.7, 0.7, 0, 1], [-0.7, 0.7, 0, 1]]], np.float32)
    tri = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    r, _ = oracle.rasterize(pos, tri, (32, 32))
    rng = np.random.default_rng(5)
    colors = rng.uniform(size=(3, 3)).astype(np.float32)
    color = colors[r[..., 3].astype(int)]                    # background / tri 0 / tri 1 in different colours
    o = oracle.antialias(color, r, pos, tri)
    ids = r[0, ..., 3]
    inner = (ids[:-1, :-1] > 0) & (ids[1:, :-1] > 0) & (ids[:-1, 1:] > 0)
    assert np.array_equal(o[0, :-1, :-1][inner], color[0, :-1, :-1][inner])
    # same mesh with split vertices: every edge is a silhouette now and the diagonal gets blended
    pos2 = pos[:, [0, 1, 2, 0, 2, 3]]
    tri2 = np.array([[0, 1, 2], [3, 4, 5]], np.int32)
    o2 = oracle.antialias(color, r, pos2, tri2)
    assert (o2[0, :-1, :-1][inner] != color[0, :-1, :-1][inner]).any()


def test_antialias_gradients(oracle):
    pos, tri, res = _one_triangle(24)
    rng = np.random.default_rng(6)
    r, _ = oracle.rasterize(pos, tri, (res, res))
    color = rng.uniform(size=(1, res, res, 3)).astype(np.float32) * 0.2 + (r[..., 3:4] > 0) * 0.7
    color = color.astype(np.float32)
    dy = rng.normal(size=color.shape).astype(np.float32)
    g_color, g_pos = oracle.antialias_grad(color, r, pos, tri, dy)
    # linear in colour: adjoint identity
    e = rng.normal(size=color.shape).astype(np.float32)
    f = lambda c, p=pos: oracle.antialias(c, r, p, tri).astype(np.float64)
    assert np.isclose(((f(color + e) - f(color)) * dy).sum(), (g_color.astype(np.float64) * e).sum(), rtol=1e-3)
    assert np.all(g_pos[..., 2] == 0)
    # Position gradient: with colour = coverage and dy = 1 the loss is the blended area, whose
    # derivative w.r.t. the vertices is known in closed form (d area / d p_i = half the opposite
    # edge rotated by 90 degrees).  The op's gradient is that of the per-pixel linear model, so it
    # agrees to a few percent, not exactly.
    for n in (24, 64):
        rr, _ = oracle.rasterize(pos, tri, (n, n))
        cov = (rr[..., 3:4] > 0).astype(np.float32)
        _, gp = oracle.antialias_grad(cov, rr, pos, tri, np.ones_like(cov))
        p = pos[0, :, :2].astype(np.float64)
        s = 0.5 * (n / 2) ** 2
        exp = s * np.array([[p[1, 1] - p[2, 1], p[2, 0] - p[1, 0]],
                            [p[2, 1] - p[0, 1], p[0, 0] - p[2, 0]],
                            [p[0, 1] - p[1, 1], p[1, 0] - p[0, 0]]])
        assert np.abs(gp[0, :, :2] - exp).max() < 0.08 * np.abs(exp).max()
        # w-gradient of a vertex = -(x, y) . (gx, gy) / w  (antialias.cu:531-532) with w = 1 here
        assert np.allclose(gp[0, :, 3], -(pos[0, :, 0] * gp[0, :, 0] + pos[0, :, 1] * gp[0, :, 1]), rtol=1e-3, atol=1e-2)


# ------------------------------------------------------------------------------ cube maps

# OpenGL cube-map convention (what texture_kernel.cu:87-110 implements): face -> (major axis, sign,
# s axis, s sign, t axis, t sign), stated here independently of the oracle's table.
_GL_FACES = [(0, +1, 2, -1, 1, -1), (0, -1, 2, +1, 1, -1), (1, +1, 0, +1, 2, +1),
             (1, -1, 0, +1, 2, -1), (2, +1, 0, +1, 1, -1), (2, -1, 0, -1, 1, -1)]


def _texel_center(face, ix, iy, w):
    """3D position of a texel centre on the cube [-1,1]^3 (also defined one texel outside the face)."""
    ma, ms, sa, ss, ta, ts = _GL_FACES[face]
    p = np.zeros(3)
    p[ma] = ms
    p[sa] = ss * ((2 * ix + 1) / w - 1.0)
    p[ta] = ts * ((2 * iy + 1) / w - 1.0)
    return p


def test_cube_face_lookup_and_edge_fold(oracle):
    import ctypes
    lib = oracle.lib()
    lib.nvdro_cube_texel.restype = ctypes.c_longlong
    rng = np.random.default_rng(11)
    # face lookup agrees with the GL table
    for _ in range(200):
        v = rng.normal(size=3).astype(np.float32)
        s = ctypes.c_float(); t = ctypes.c_float()
        f = lib.nvdro_cube_index(v.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), ctypes.byref(s), ctypes.byref(t))
        ma, ms, sa, ss, ta, ts = _GL_FACES[f]
        assert ma == int(np.argmax(np.abs(v))) and ms == np.sign(v[ma])
        assert np.isclose(s.value, 0.5 * ss * v[sa] / abs(v[ma]) + 0.5, atol=1e-6)
        assert np.isclose(t.value, 0.5 * ts * v[ta] / abs(v[ma]) + 0.5, atol=1e-6)
    zero = np.zeros(3, np.float32)
    assert lib.nvdro_cube_index(zero.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), ctypes.byref(s), ctypes.byref(t)) == -1
    # a texel one step beyond an edge folds onto the real texel nearest to where it would have been
    for w in (1, 2, 4, 5):
        centers = {(f, x, y): _texel_center(f, x, y, w) for f in range(6) for x in range(w) for y in range(w)}
        for f in range(6):
            for i in range(w):
                for (ix, iy) in ((-1, i), (w, i), (i, -1), (i, w)):
                    got = lib.nvdro_cube_texel(f, ix, iy, w)
                    virt = _texel_center(f, ix, iy, w)
                    best = min((k for k in centers if k[0] != f), key=lambda k: np.sum((centers[k] - virt) ** 2))
                    assert got == best[1] + w * (best[2] + w * best[0]), (w, f, ix, iy)
            for (ix, iy) in ((-1, -1), (w, -1), (-1, w), (w, w)):
                assert lib.nvdro_cube_texel(f, ix, iy, w) == -1           # no fourth texel at a corner
            assert lib.nvdro_cube_texel(f, 0, w - 1, w) == 0 + w * (w - 1 + w * f)


def _dir_texture(rng, S, C):
    return rng.uniform(size=(1, 6, S, S, C)).astype(np.float32)


def test_cube_sampling_is_seamless_and_corner_aware(oracle):
    # A texture that is a linear function of the texel centre's 3D position is reproduced (to O(1/S))
    # everywhere, including across edges and at corners, only if every fold picks the right neighbour.
    S = 16
    alpha = np.array([0.3, -0.5, 0.8])
    tex = np.zeros((1, 6, S, S, 1), np.float32)
    for f in range(6):
        for y in range(S):
            for x in range(S):
                tex[0, f, y, x, 0] = alpha @ _texel_center(f, x, y, S)
    rng = np.random.default_rng(12)
    v = rng.normal(size=(1, 40, 40, 3)).astype(np.float32)
    # add directions hugging edges and corners
    v[0, 0, :, :] = np.array([1, 1, 0]) + rng.normal(size=(40, 3)) * 0.02
    v[0, 1, :, :] = np.array([1, -1, 1]) + rng.normal(size=(40, 3)) * 0.02
    v[0, 2, :, :] = np.array([-1, -1, -1]) + rng.normal(size=(40, 3)) * 0.02
    o = oracle.texture(tex, v, filter_mode="linear", boundary_mode="cube")[..., 0]
    p = v / np.abs(v).max(-1, keepdims=True)                      # point on the cube surface
    assert np.abs(o - p @ alpha).max() < 2.5 * np.abs(alpha).sum() / S
    # nearest: a texture holding the face index returns the face of the direction
    texf = np.zeros((1, 6, 4, 4, 1), np.float32)
    for f in range(6):
        texf[0, f] = f
    of = oracle.texture(texf, v, filter_mode="nearest", boundary_mode="cube")[..., 0]
    ma = np.abs(v).argmax(-1)
    expect = 2 * ma + (np.take_along_axis(v, ma[..., None], -1)[..., 0] < 0)
    assert np.array_equal(of, expect.astype(np.float32))
    # invalid direction -> zeros, no gradient
    z = np.zeros((1, 1, 1, 3), np.float32)
    assert oracle.texture(tex, z, filter_mode="linear", boundary_mode="cube").sum() == 0.0


@pytest.mark.parametrize("fm", ["nearest", "linear", "linear-mipmap-nearest", "linear-mipmap-linear"])
def test_cube_gradients(oracle, fm):
    rng = np.random.default_rng(13)
    S, C = 8, 2
    tex = _dir_texture(rng, S, C)
    v = rng.normal(size=(2, 7, 7, 3)).astype(np.float32)
    v[0, 0] = np.array([1, 0.97, 0.2]) + rng.normal(size=(7, 3)) * 0.05        # near an edge
    v[0, 1] = np.array([1, -1, 1]) + rng.normal(size=(7, 3)) * 0.04            # near a corner
    v = v.astype(np.float32)
    mip = "mipmap" in fm
    da = (rng.normal(size=(2, 7, 7, 6)) * 0.15).astype(np.float32) if mip else None
    bias = rng.uniform(-0.3, 0.3, size=(2, 7, 7)).astype(np.float32) if mip else None
    dy = rng.normal(size=(2, 7, 7, C)).astype(np.float32)
    kw = dict(filter_mode=fm, boundary_mode="cube")
    f = lambda t=tex, u=v, d=da, b=bias: oracle.texture(t, u, d, b, **kw).astype(np.float64)
    g = oracle.texture_grad(tex, v, dy, da, bias, **kw)
    e = rng.normal(size=tex.shape).astype(np.float32)
    assert np.isclose(((f(t=tex + e) - f()) * dy).sum(), (g["tex"].astype(np.float64) * e).sum(), rtol=3e-4, atol=3e-4)
    if fm == "nearest":
        return
    assert g["uv"].shape == v.shape
    eps = 1e-3
    fd = np.zeros(v.shape, np.float64)
    for k in range(3):
        up = v.copy(); up[..., k] += eps
        um = v.copy(); um[..., k] -= eps
        fd[..., k] = ((f(u=up) - f(u=um)) * dy).sum(-1) / (2 * eps)
    err = np.abs(fd - g["uv"])
    scale = max(1.0, np.abs(fd).max())
    assert np.median(err) < 2e-2 * scale and (err < 6e-2 * scale).mean() > 0.75
    if fm == "linear-mipmap-linear":
        assert g["uv_da"].shape == da.shape
        fda = np.zeros(da.shape, np.float64)
        for k in range(6):
            dp = da.copy(); dp[..., k] += 1e-4
            dm = da.copy(); dm[..., k] -= 1e-4
            fda[..., k] = ((f(d=dp) - f(d=dm)) * dy).sum(-1) / 2e-4
        erra = np.abs(fda - g["uv_da"])
        assert (erra < 4e-2 * max(1.0, np.abs(fda).max())).mean() > 0.8
        fdb = ((f(b=bias + eps) - f(b=bias - eps)) * dy).sum(-1) / (2 * eps)
        assert (np.abs(fdb - g["mip_level_bias"]) < 2e-2 * max(1.0, np.abs(fdb).max())).mean() > 0.8


def test_cube_mip_level_gradient_wrt_direction(oracle):
    # With a texture whose level-k mip is the constant k the output IS the mip level, so g_uv is purely
    # d(level)/d(direction): checks the second-derivative path (texture_kernel.cu:235-317).
    S = 32
    tex = np.zeros((1, 6, S, S, 1), np.float32)
    levels = [np.full((1, 6, S >> k, S >> k, 1), float(k), np.float32) for k in range(1, 6)]
    rng = np.random.default_rng(14)
    v = (rng.normal(size=(1, 6, 6, 3)) + np.array([0.2, 0.1, 1.5])).astype(np.float32)
    da = (rng.normal(size=(1, 6, 6, 6)) * 0.3).astype(np.float32)
    kw = dict(mip=levels, filter_mode="linear-mipmap-linear", boundary_mode="cube")
    f = lambda u: oracle.texture(tex, u, da, **kw).astype(np.float64)[..., 0]
    lvl = f(v)
    assert lvl.min() > 0.02 and lvl.max() < 4.98                  # away from the clamps
    g = oracle.texture_grad(tex, v, np.ones((1, 6, 6, 1), np.float32), da, **kw)
    eps = 1e-3
    for k in range(3):
        up = v.copy(); up[..., k] += eps
        um = v.copy(); um[..., k] -= eps
        fd = (f(up) - f(um)) / (2 * eps)
        assert np.abs(fd - g["uv"][..., k]).max() < 2e-2 * max(1.0, np.abs(fd).max())


def test_pipeline_fixture_is_reproduced(oracle):
    """tests/golden/pipeline_small.npz (made by tests/golden/make_pipeline_fixture.py) pins the oracle's output
    on a whole forward+backward op chain: any change to the restatement shows up here."""
    import importlib.util
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_pipeline_fixture", os.path.join(here, "make_pipeline_fixture.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    fx = np.load(os.path.join(here, "pipeline_small.npz"))
    i = gen.inputs()
    for k, v in i.items():
        assert np.array_equal(fx["in_" + k], v), k                      # the generator is deterministic
    o = gen.run_oracle(i)
    assert np.array_equal(o["rast"][..., 3], fx["out_rast"][..., 3])     # triangle ids: exact
    for k, v in o.items():
        ref = fx["out_" + k]
        assert np.abs(v - ref).max() <= 1e-6 * max(1.0, float(np.abs(ref).max())), k
