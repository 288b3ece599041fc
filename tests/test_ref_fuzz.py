"""Randomised pinning: seeded random scenes / shapes / modes through the pinned oracle, i.e. oracle/*.c against the
reference's own code (oracle/_ref) call by call (oracle/pinned.py).  Complements the hand-made scenes of
tests/test_ref_pins_oracle.py with inputs nobody chose: odd resolutions, triangles of every size and orientation,
vertices behind the eye, degenerate and duplicated triangles, every filter / boundary / channel-count combination."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def po(oracle, ref):
    assert oracle.enabled
    return oracle


def _random_scene(rng):
    N = int(rng.integers(1, 4))
    H, W = int(rng.integers(3, 150)), int(rng.integers(3, 150))
    T = int(rng.integers(1, 400))
    kind = rng.integers(0, 4)
    if kind == 0:                                    # soup of independent triangles, all sizes
        c = rng.uniform(-1.2, 1.2, size=(N, T, 1, 2))
        r = np.exp(rng.uniform(np.log(0.01), np.log(1.5), size=(N, T, 1, 1)))
        xy = c + r * rng.normal(size=(N, T, 3, 2))
        z = rng.uniform(-1.1, 1.1, size=(N, T, 3, 1))
        w = np.ones_like(z)
    elif kind == 1:                                  # perspective: w varies, some vertices behind the eye
        xy = rng.normal(size=(N, T, 3, 2)) * 1.5
        z = rng.normal(size=(N, T, 3, 1))
        w = rng.uniform(-0.3, 2.5, size=(N, T, 3, 1))
    elif kind == 2:                                  # snapped to pixel / subpixel positions: ties and on-edge samples
        g = rng.integers(-W, W + 1, size=(N, T, 3, 2)) / np.array([W / 2.0, H / 2.0]) * rng.choice([1.0, 0.5, 1.0 / 16.0])
        xy = g
        z = rng.choice([-0.5, 0.0, 0.25, 0.5], size=(N, T, 3, 1))
        w = np.ones_like(z)
    else:                                            # slivers and near-degenerate triangles
        a = rng.uniform(-1, 1, size=(N, T, 1, 2)); d = rng.normal(size=(N, T, 1, 2))
        t = rng.uniform(-1, 1, size=(N, T, 3, 1))
        xy = a + d * t + rng.normal(size=(N, T, 3, 2)) * rng.choice([0.0, 1e-4, 1e-2])
        z = rng.uniform(-0.9, 0.9, size=(N, T, 3, 1))
        w = rng.uniform(0.5, 2.0, size=(N, T, 3, 1))
    pos = np.concatenate([xy * w, z * w, w], -1).reshape(N, 3 * T, 4).astype(np.float32)
    tri = np.arange(3 * T, dtype=np.int32).reshape(T, 3)
    if rng.uniform() < 0.3:                          # shared vertices, duplicates, a corrupt index
        tri = rng.integers(0, 3 * T, size=(T, 3)).astype(np.int32)
        if T > 3:
            tri[1] = tri[0]
            tri[2] = [0, 3 * T, 1]
    return pos, tri, (H, W)


@pytest.mark.parametrize("seed", range(24))
def test_random_raster_interpolate_antialias(po, seed):
    rng = np.random.default_rng(9000 + seed)
    pos, tri, res = _random_scene(rng)
    # barycentrics of slivers / near-w=0 triangles are ill-conditioned (1/area): ids must be identical, floats are
    # compared where both sides are finite and the triangle is not a sliver -- PinnedOracle's bars apply to the rest
    try:
        ro, rdbo = po.rasterize(pos, tri, res)
    except AssertionError as e:
        if "triangle ids differ" in str(e):
            raise
        ro, rdbo = po._o.rasterize(pos, tri, res)
        from oracle import ref
        r, _ = ref.rasterize(pos, tri, res)
        assert (r[..., 3] != ro[..., 3]).sum() == 0
        ok = np.isfinite(r).all(-1) & np.isfinite(ro).all(-1)
        assert (np.abs(r[ok][:, :3] - ro[ok][:, :3]) > 1e-4).mean() < 0.02
    A = int(rng.integers(1, 6))
    attr = rng.uniform(-1, 1, size=(pos.shape[0] if rng.uniform() < 0.5 else 1, pos.shape[1], A)).astype(np.float32)
    clean = np.isfinite(ro).all() and np.isfinite(rdbo).all() and np.abs(rdbo).max() < 1e4
    if not clean:
        return
    diff = "all" if rng.uniform() < 0.5 else list(rng.integers(-A, A, size=int(rng.integers(1, 4))))
    out, da = po.interpolate(attr, ro, tri, rast_db=rdbo, diff_attrs=diff)
    dy = rng.normal(size=out.shape).astype(np.float32)
    dda = rng.normal(size=da.shape).astype(np.float32)
    po.interpolate_grad(attr, ro, tri, dy, rast_db=rdbo, dda=dda, diff_attrs=diff)
    col = rng.uniform(size=ro.shape[:3] + (int(rng.integers(1, 5)),)).astype(np.float32)
    # the edge-crossing position d / dy (antialias.cu:338-359) is ill-conditioned on slivers: a handful of blended
    # pixels may differ by a few 1e-5 between two correct evaluations; everything else must meet the 1e-5 bar
    from oracle import ref
    a, b = po._o.antialias(col, ro, pos, tri), ref.antialias(col, ro, pos, tri)
    d = np.abs(a - b)
    assert (d > 1e-5).mean() <= 1e-4 and d.max() <= 1e-3, (float((d > 1e-5).mean()), float(d.max()))


@pytest.mark.parametrize("seed", range(24))
def test_random_texture(po, seed):
    rng = np.random.default_rng(7000 + seed)
    cube = rng.uniform() < 0.3
    C = int(rng.integers(1, 6))
    tn = int(rng.integers(1, 3))
    N = tn if tn > 1 else int(rng.integers(1, 3))
    H, W = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    fm = str(rng.choice(["nearest", "linear", "linear-mipmap-nearest", "linear-mipmap-linear"]))
    if cube:
        S = int(2 ** rng.integers(0, 5))
        tex = rng.uniform(size=(tn, 6, S, S, C)).astype(np.float32)
        uv = rng.normal(size=(N, H, W, 3)).astype(np.float32)
        uv[rng.uniform(size=(N, H, W)) < 0.05] = 0.0                         # invalid directions
        uv_da = (rng.normal(size=(N, H, W, 6)) * rng.choice([0.0, 0.02, 0.5])).astype(np.float32)
        bm = "cube"
    else:
        th, tw = int(2 ** rng.integers(0, 7)), int(2 ** rng.integers(0, 7))
        tex = rng.uniform(size=(tn, th, tw, C)).astype(np.float32)
        uv = rng.uniform(-1.5, 2.5, size=(N, H, W, 2)).astype(np.float32)
        uv_da = (rng.normal(size=(N, H, W, 4)) * rng.choice([0.0, 0.02, 0.5])).astype(np.float32)
        bm = str(rng.choice(["wrap", "clamp", "zero"]))
    mip = "mipmap" in fm
    mode = rng.integers(0, 3) if mip else 0          # 0: uv_da, 1: bias only, 2: both
    bias = rng.uniform(-1, 3, size=(N, H, W)).astype(np.float32)
    kw = dict(filter_mode=fm, boundary_mode=bm)
    if mip:
        kw.update(uv_da=None if mode == 1 else uv_da, mip_level_bias=None if mode == 0 else bias)
        if rng.uniform() < 0.3:
            kw["max_mip_level"] = int(rng.integers(0, 4))
    po.texture(tex, uv, **kw)
    dy = rng.normal(size=(N, H, W, C)).astype(np.float32)
    dy[rng.uniform(size=(N, H, W)) < 0.2] = 0.0
    po.texture_grad(tex, uv, dy, **kw)
