"""CPU tests that pin the C oracle to THE REFERENCE ITSELF (oracle/_ref: the reference's glue, kernels and
CudaRaster compiled unmodified for the host, see oracle/ref.py and oracle/refshim/).

Every ``oracle.X(...)`` call below goes through ``oracle.pinned.PinnedOracle``, which runs the same call through
the reference and raises unless both agree (triangle ids identical, forward floats within 1e-5, gradients within
2e-5 of the tensor's magnitude).  The scenes are the ones the GPU parity tests feed to the HIP kernels
(tests/test_gpu_*.py), so "HIP == oracle" there plus "oracle == reference" here closes the chain; on the GPU box
the same cross-check runs inside the GPU tests themselves, because the prebuilt oracle/_ref travels with them.

Skipped only where neither /root/reference nor a prebuilt oracle/_ref exists."""
import os

import numpy as np
import pytest

from nvdiffrast_amd.utils import m10k_batch, stress_triangles


@pytest.fixture(scope="module")
def po(oracle, ref):
    assert oracle.enabled, "oracle/_ref is present but pinning is disabled"
    return oracle


# --------------------------------------------------------------------------- the reference's own golden
def test_reference_build_reproduces_its_golden_image(ref):
    """samples/torch/triangle.py:19-30 through the reference's CudaRaster + shader + interpolate kernels on the
    CPU == docs/img/tri.png, all 65,536 pixels (validates the shim itself, PTX emulation included)."""
    from PIL import Image
    pos = np.array([[[-0.8, -0.8, 0, 1], [0.8, -0.8, 0, 1], [-0.8, 0.8, 0, 1]]], np.float32)
    col = np.array([[[1, 0, 0], [0, 1, 0], [0, 0, 1]]], np.float32)
    tri = np.array([[0, 1, 2]], np.int32)
    for variant in ("fma", "nofma"):
        rast, _ = ref.rasterize(pos, tri, (256, 256), variant=variant)
        out, _ = ref.interpolate(col, rast, tri, variant=variant)
        img = np.clip(np.rint(out[0, ::-1] * 255), 0, 255).astype(np.uint8)
        g = np.array(Image.open(os.path.join(os.path.dirname(__file__), "golden", "tri.png")))
        assert (img != g).sum() == 0
    assert ref.lib("fma").nvdr_ref_uses_fma() == 1 and ref.lib("nofma").nvdr_ref_uses_fma() == 0


def test_reference_glue_error_messages_survive(ref):
    """The glue's TORCH_CHECKs are live in the CPU build (same text a CUDA build raises)."""
    P = ref.plugin()
    pos = np.zeros((1, 3, 4), np.float32)
    with pytest.raises(RuntimeError, match=r"tri must have shape \[>0, 3\]"):
        P.rasterize_fwd_cuda(P.RasterizeCRStateWrapper(), pos, np.zeros((1, 4), np.int32), (8, 8), np.zeros((0, 2), np.int32), -1)
    with pytest.raises(RuntimeError, match="resolution must be"):
        P.rasterize_fwd_cuda(P.RasterizeCRStateWrapper(), pos, np.zeros((1, 3), np.int32), (0, 8), np.zeros((0, 2), np.int32), -1)
    with pytest.raises(RuntimeError):
        P.texture_construct_mip(np.zeros((1, 12, 8, 1), np.float32), -1, False)      # 12 -> 6 -> 3: odd extent


# --------------------------------------------------------------------------- rasterize forward (the bit-exact part)
@pytest.mark.parametrize("res", [(512, 512), (250, 333), (64, 72), (8, 8), (5, 3)])
def test_lattice_mesh(po, res):
    b = m10k_batch(2, seed=11)
    po.rasterize(b["pos"], b["tri"], res)


def test_stress_overdraw(po):
    s = stress_triangles(2, T=4000, res=256, seed=3)
    po.rasterize(s["pos"], s["tri"], (256, 256))


def test_clipping_and_huge_triangles(po):
    rng = np.random.default_rng(5)
    T = 600
    pos = rng.normal(size=(2, 3 * T, 4)).astype(np.float32) * np.array([2.0, 2.0, 1.5, 1.0], np.float32)
    pos[..., 3] = rng.uniform(-0.5, 2.0, size=pos.shape[:2])
    tri = np.arange(3 * T, dtype=np.int32).reshape(T, 3)
    # ids must be identical; barycentrics of clipped slivers near w = 0 are ill-conditioned, so the float
    # comparison is made on the raw oracle/reference pair with the GPU test's bar (1e-4 where finite)
    ro, _ = po._o.rasterize(pos, tri, (128, 200))
    from oracle import ref
    r, _ = ref.rasterize(pos, tri, (128, 200))
    assert (ro[..., 3] != r[..., 3]).sum() == 0
    ok = np.isfinite(ro).all(-1) & np.isfinite(r).all(-1)
    assert np.abs(ro[ok][:, :3] - r[ok][:, :3]).max() <= 1e-4


def test_depth_ties_and_duplicates(po):
    b = m10k_batch(1, seed=2, nx=20, ny=10)
    tri = np.concatenate([b["tri"], b["tri"][::-1], b["tri"]], 0)
    r, _ = po.rasterize(b["pos"], tri, (160, 160))
    assert (r[..., 3][r[..., 3] > 0] > 2 * b["tri"].shape[0]).all()


def test_range_mode(po):
    b = m10k_batch(1, seed=4, nx=30, ny=20)
    T = b["tri"].shape[0]
    ranges = np.array([[0, T], [100, 500], [T - 7, 7], [3, 0]], np.int32)
    po.rasterize(b["pos"][0], b["tri"], (96, 128), ranges=ranges)


def test_many_triangles_in_one_bin(po):
    rng = np.random.default_rng(9)
    T = 5000
    c = rng.uniform(-0.1, 0.1, size=(T, 1, 2))
    xy = c + rng.uniform(-0.05, 0.05, size=(T, 3, 2))
    z = rng.uniform(-0.9, 0.9, size=(T, 3, 1))
    pos = np.concatenate([xy, z, np.ones_like(z)], -1).reshape(1, -1, 4).astype(np.float32)
    tri = np.arange(3 * T, dtype=np.int32).reshape(T, 3)
    po.rasterize(pos, tri, (256, 256))


def test_more_than_32_images(po):
    """Per-image parameters beyond the 32 embedded in the launch block (RasterImpl.cpp:267-272)."""
    b = m10k_batch(40, seed=13, nx=12, ny=8)
    po.rasterize(b["pos"], b["tri"], (40, 48))


def test_depth_peeling_layers(po):
    """DepthPeeler semantics (ops.py:141-204, FineRaster.inl:253-258,349): four layers of a high-overdraw scene."""
    s = stress_triangles(2, T=1500, res=128, seed=8)
    layers = po.rasterize_layers(s["pos"], s["tri"], (128, 128), 4)
    cov = [int((l[0][..., 3] > 0).sum()) for l in layers]
    assert cov[0] >= cov[1] >= cov[2] >= cov[3] > 0
    assert any((layers[k][0][..., 3] != layers[0][0][..., 3]).any() for k in (1, 2, 3))


def test_viewport_tiling_beyond_2048(po):
    """torch_rasterize.cpp:99-124: 2100 x 2500 is rasterised as 2 x 2 viewport tiles."""
    b = m10k_batch(1, seed=30, nx=12, ny=8)
    po.rasterize(b["pos"], b["tri"], (2100, 2500))


def test_out_of_range_indices_and_degenerates(po):
    pos = np.array([[[-1, -1, 0, 1], [1, -1, 0, 1], [0, 1, 0, 1], [0.5, 0.5, 0, 1]]], np.float32)
    tri = np.array([[0, 1, 2], [0, 1, 7], [-1, 1, 2], [1, 1, 2], [0, 3, 2]], np.int32)
    po.rasterize(pos, tri, (16, 16))


# --------------------------------------------------------------------------- rasterize backward, interpolate
@pytest.mark.parametrize("with_db", [True, False])
def test_raster_interp_chain_gradients(po, with_db):
    """The headline op graph (rasterize + interpolate fwd + bwd) on the benchmark mesh at 2 x 256^2."""
    N, res = 2, (256, 256)
    b = m10k_batch(N, seed=21)
    rng = np.random.default_rng(0)
    G = rng.normal(size=(N,) + res + (4,)).astype(np.float32)
    Gdb = rng.normal(size=(N,) + res + (4,)).astype(np.float32) * 0.01
    ro, rdbo = po.rasterize(b["pos"], b["tri"], res)
    po.interpolate(b["attr"], ro, b["tri"])
    _g_attr, g_rast, _ = po.interpolate_grad(b["attr"], ro, b["tri"], G)
    po.rasterize_grad(b["pos"], b["tri"], ro, g_rast, ddb=Gdb if with_db else None)


def test_raster_grad_on_overdraw_and_range_mode(po):
    rng = np.random.default_rng(3)
    s = stress_triangles(1, T=800, res=96, seed=5)
    ro, _ = po.rasterize(s["pos"], s["tri"], (96, 96))
    dy = rng.normal(size=ro.shape).astype(np.float32)
    ddb = rng.normal(size=ro.shape).astype(np.float32) * 0.01
    po.rasterize_grad(s["pos"], s["tri"], ro, dy, ddb=ddb)
    b = m10k_batch(1, seed=4, nx=30, ny=20)
    T = b["tri"].shape[0]
    ranges = np.array([[0, T], [100, 500]], np.int32)
    ro, _ = po.rasterize(b["pos"][0], b["tri"], (64, 80), ranges=ranges)
    dy = rng.normal(size=ro.shape).astype(np.float32)
    po.rasterize_grad(b["pos"][0], b["tri"], ro, dy, ddb=dy[..., ::-1].copy() * 0.01)


@pytest.mark.parametrize("A", [1, 2, 3, 4, 7])
def test_interpolate_forward_and_backward(po, A):
    rng = np.random.default_rng(40 + A)
    b = m10k_batch(2, seed=6, attrs=A)
    ro, rdbo = po.rasterize(b["pos"], b["tri"], (128, 160))
    per_item = np.repeat(b["attr"], 2, 0) * np.array([1.0, 0.5], np.float32).reshape(2, 1, 1)
    for attr in (b["attr"], per_item, b["attr"][0]):            # broadcast [1,V,A], instanced [N,V,A], plain [V,A]
        po.interpolate(attr, ro, b["tri"])
        for diff in ("all", [A - 1, 0, -1]):
            out, da = po.interpolate(attr, ro, b["tri"], rast_db=rdbo, diff_attrs=diff)
            dy = rng.normal(size=out.shape).astype(np.float32)
            dda = rng.normal(size=da.shape).astype(np.float32)
            po.interpolate_grad(attr, ro, b["tri"], dy, rast_db=rdbo, dda=dda, diff_attrs=diff)
        po.interpolate_grad(attr, ro, b["tri"], rng.normal(size=ro.shape[:3] + (A,)).astype(np.float32))


# --------------------------------------------------------------------------- texture
FILTERS = ["nearest", "linear", "linear-mipmap-nearest", "linear-mipmap-linear"]


@pytest.mark.parametrize("bm", ["wrap", "clamp", "zero"])
@pytest.mark.parametrize("fm", FILTERS)
@pytest.mark.parametrize("C,tex_n", [(1, 1), (2, 2), (3, 1), (4, 2), (5, 1)])
def test_texture_2d_matrix(po, fm, bm, C, tex_n):
    """The 60-case matrix of tests/test_gpu_texture_aa.py::test_texture_forward_backward, same seeds."""
    rng = np.random.default_rng(100 + C)
    N, H, W = 2, 37, 29
    tex = rng.uniform(size=(tex_n, 32, 64, C)).astype(np.float32)
    uv = rng.uniform(-0.3, 1.3, size=(N, H, W, 2)).astype(np.float32)
    mip = "mipmap" in fm
    uv_da = (rng.normal(size=(N, H, W, 4)) * 0.05).astype(np.float32) if mip else None
    bias = rng.uniform(-0.5, 0.5, size=(N, H, W)).astype(np.float32) if mip else None
    dy = rng.normal(size=(N, H, W, C)).astype(np.float32)
    dy[0, :3] = 0.0
    po.texture(tex, uv, uv_da, bias, filter_mode=fm, boundary_mode=bm)
    po.texture_grad(tex, uv, dy, uv_da, bias, filter_mode=fm, boundary_mode=bm)


def test_texture_bias_only_uvda_only_and_level_limits(po):
    rng = np.random.default_rng(7)
    tex = rng.uniform(size=(1, 64, 64, 3)).astype(np.float32)
    uv = rng.uniform(size=(2, 16, 16, 2)).astype(np.float32)
    uv_da = (rng.normal(size=(2, 16, 16, 4)) * 0.04).astype(np.float32)
    bias = rng.uniform(0.0, 4.0, size=(2, 16, 16)).astype(np.float32)
    dy = rng.normal(size=(2, 16, 16, 3)).astype(np.float32)
    for da, b in ((uv_da, None), (None, bias)):
        for fm in ("linear-mipmap-nearest", "linear-mipmap-linear"):
            po.texture(tex, uv, da, b, filter_mode=fm)
            po.texture_grad(tex, uv, dy, da, b, filter_mode=fm)
    po.texture(tex, uv, uv_da, filter_mode="linear-mipmap-linear", max_mip_level=2)
    po.texture_grad(tex, uv, dy, uv_da, filter_mode="linear-mipmap-linear", max_mip_level=2)


def test_mip_construction_shapes(po):
    rng = np.random.default_rng(8)
    for shape in [(2, 64, 16, 3), (1, 8, 128, 4), (1, 2, 2, 1), (1, 6, 8, 8, 2)]:
        po.texture_build_mip(rng.uniform(size=shape).astype(np.float32))
    po.texture_build_mip(rng.uniform(size=(1, 32, 32, 2)).astype(np.float32), 3)


def test_custom_mip_stack(po):
    rng = np.random.default_rng(9)
    tex = rng.uniform(size=(1, 16, 16, 2)).astype(np.float32)
    levels = [rng.uniform(size=(1, 16 >> k, 16 >> k, 2)).astype(np.float32) for k in range(1, 4)]
    uv = rng.uniform(size=(2, 11, 13, 2)).astype(np.float32)
    da = (rng.normal(size=(2, 11, 13, 4)) * 0.2).astype(np.float32)
    dy = rng.normal(size=(2, 11, 13, 2)).astype(np.float32)
    for fm in ("linear-mipmap-nearest", "linear-mipmap-linear"):
        po.texture(tex, uv, da, mip=levels, filter_mode=fm)
        g = po.texture_grad(tex, uv, dy, da, mip=levels, filter_mode=fm)
        assert g["mip"] is not None and len(g["mip"]) == 3


def _directions(rng, shape, near_edges=False):
    d = rng.normal(size=shape + (3,)).astype(np.float32)
    if near_edges:                                   # push many directions onto face edges and cube corners
        m = rng.uniform(size=shape) < 0.5
        a = np.abs(d)
        d = np.where(m[..., None], np.sign(d) * (a.max(-1, keepdims=True) * rng.uniform(0.97, 1.0, size=shape + (3,))), d).astype(np.float32)
    return d


@pytest.mark.parametrize("fm", FILTERS)
@pytest.mark.parametrize("C,tex_n", [(1, 1), (2, 1), (3, 2), (4, 2)])
def test_texture_cube_matrix(po, fm, C, tex_n):
    """Cube maps, all filters, channel counts that select the float/float2/float4 kernel instances
    (texture_kernel.cu:803-838), tex batch 1 and 2 (the corner-texel handling of texture_kernel.cu:431-432
    depends on the slice index), directions concentrated on edges and corners."""
    rng = np.random.default_rng(200 + 10 * C + tex_n)
    N, H, W = 2, 23, 19
    tex = rng.uniform(size=(tex_n, 6, 16, 16, C)).astype(np.float32)
    uv = _directions(rng, (N, H, W), near_edges=True)
    mip = "mipmap" in fm
    uv_da = (rng.normal(size=(N, H, W, 6)) * 0.05).astype(np.float32) if mip else None
    bias = rng.uniform(-0.5, 0.5, size=(N, H, W)).astype(np.float32) if mip else None
    dy = rng.normal(size=(N, H, W, C)).astype(np.float32)
    po.texture(tex, uv, uv_da, bias, filter_mode=fm, boundary_mode="cube")
    po.texture_grad(tex, uv, dy, uv_da, bias, filter_mode=fm, boundary_mode="cube")


# --------------------------------------------------------------------------- antialias
def test_antialias_forward_backward(po):
    rng = np.random.default_rng(12)
    for N, res, kw in ((2, (96, 128), dict(seed=14, nx=24, ny=12)), (1, (64, 64), dict(seed=15, nx=8, ny=6))):
        b = m10k_batch(N, attrs=3, **kw)
        ro, _ = po.rasterize(b["pos"], b["tri"], res)
        col, _ = po.interpolate(b["attr"], ro, b["tri"])
        po.antialias(col, ro, b["pos"], b["tri"])
        dy = rng.normal(size=col.shape).astype(np.float32)
        po.antialias_grad(col, ro, b["pos"], b["tri"], dy)


def test_antialias_on_overdraw_and_range_mode(po):
    rng = np.random.default_rng(13)
    s = stress_triangles(1, T=600, res=96, seed=6)
    ro, _ = po.rasterize(s["pos"], s["tri"], (96, 96))
    col = rng.uniform(size=ro.shape[:3] + (4,)).astype(np.float32)
    po.antialias(col, ro, s["pos"], s["tri"])
    po.antialias_grad(col, ro, s["pos"], s["tri"], rng.normal(size=col.shape).astype(np.float32))
    b = m10k_batch(1, seed=4, nx=30, ny=20)
    T = b["tri"].shape[0]
    ranges = np.array([[0, T], [100, 500]], np.int32)
    ro, _ = po.rasterize(b["pos"][0], b["tri"], (64, 80), ranges=ranges)
    col = rng.uniform(size=ro.shape[:3] + (3,)).astype(np.float32)
    po.antialias(col, ro, b["pos"][0], b["tri"])
    po.antialias_grad(col, ro, b["pos"][0], b["tri"], rng.normal(size=col.shape).astype(np.float32))


# --------------------------------------------------------------------------- whole chain (BASELINE config 3's op graph)
def test_four_op_chain(po):
    rng = np.random.default_rng(17)
    N, res = 2, (160, 160)
    b = m10k_batch(N, seed=23)
    tex = rng.uniform(size=(1, 256, 256, 3)).astype(np.float32)
    ro, rdbo = po.rasterize(b["pos"], b["tri"], res)
    uv, uvda = po.interpolate(b["uv"], ro, b["tri"], rast_db=rdbo, diff_attrs="all")
    col = po.texture(tex, uv, uvda, filter_mode="linear-mipmap-linear")
    aa = po.antialias(col, ro, b["pos"], b["tri"])
    dy = rng.normal(size=aa.shape).astype(np.float32)
    g_col, _g_pos_aa = po.antialias_grad(col, ro, b["pos"], b["tri"], dy)
    g = po.texture_grad(tex, uv, g_col, uvda, filter_mode="linear-mipmap-linear")
    _ga, g_rast, g_rdb = po.interpolate_grad(b["uv"], ro, b["tri"], g["uv"], rast_db=rdbo, dda=g["uv_da"], diff_attrs="all")
    po.rasterize_grad(b["pos"], b["tri"], ro, g_rast, ddb=g_rdb)


def test_knife_edge_silhouettes_are_not_defined_by_the_reference(po, ref):
    """A silhouette edge lying EXACTLY on a pixel boundary gives the reference a blend weight of +-0 or +-2^-22
    depending on a single rounding (antialias.cu:307-365), and its backward pass treats the two cases differently:
    weight bits == 0 -> item skipped (:409), anything else -> full position gradient, which does not scale with the
    weight (:519-546).  The reference's own two builds (FMA contraction on / off, oracle/refshim/build.py) therefore
    disagree with each other by O(100) on such an input, while every forward image agrees to 1e-5.  Documented
    ambiguity: parity of antialias position gradients is claimed for edges in general position only."""
    pos = np.array([[[-0.7, -0.7, 0, 1], [0.7, 0.7, 0, 1], [-0.7, 0.7, -0.1, 1]]], np.float32)   # y = 0.7 -> row boundary 34.0 of 40
    tri = np.array([[0, 1, 2]], np.int32)
    res = (40, 56)
    rng = np.random.default_rng(2)
    color = rng.uniform(size=(1,) + res + (4,)).astype(np.float32)
    dy = rng.normal(size=color.shape).astype(np.float32)
    ro, _ = po.rasterize(pos, tri, res)
    fwd = [ref.antialias(color, ro, pos, tri, variant=v) for v in ("fma", "nofma")] + [po._o.antialias(color, ro, pos, tri)]
    assert np.abs(fwd[0] - fwd[1]).max() <= 1e-5 and np.abs(fwd[0] - fwd[2]).max() <= 1e-5
    g = [ref.antialias_grad(color, ro, pos, tri, dy, variant=v)[1] for v in ("fma", "nofma")]
    assert np.abs(g[0] - g[1]).max() > 10.0
    # general position (no edge on a pixel boundary or through pixel centres -- the diagonal of the triangle above has
    # slope 5/7 in pixels and passes through centres, the other knife edge: weight exactly +-1/2 kills the gradient,
    # :541-546): both builds and the oracle agree
    pos2 = pos.copy(); pos2[0, :, :2] += np.array([[0.0131, -0.0072], [-0.0057, 0.0113], [0.0091, 0.0039]], np.float32)
    ro2, _ = po.rasterize(pos2, tri, res)
    g2 = [ref.antialias_grad(color, ro2, pos2, tri, dy, variant=v)[1] for v in ("fma", "nofma")]
    assert np.abs(g2[0] - g2[1]).max() <= 2e-5 * np.abs(g2[0]).max()
    po.antialias_grad(color, ro2, pos2, tri, dy)


def test_lut_coverage_equals_the_integer_fill_rule(ref):
    """the reference's LIVE fine-raster coverage is the LUT path (Util.inl:214-300,
    cover8x8_exact_fast with flip bits from cover8x8_selectFlips), while the oracle and the HIP kernel implement the
    integer rule that Util.inl:304-359 states without a LUT: pixel (x, y) of a tile is inside an edge iff
        (ox - 16x) * dy - (oy - 16y) * dx - [dy > 0 or (dy == 0 and dx <= 0)]  >=  0
    (subpixel units; the bracket is the top-left exclusion).  Both reference functions are compiled from the
    reference's own Util.inl (oracle/refshim/cover_probe.cpp) and compared with each other and with that rule on
    400k edges: random ones inside a 2048-px viewport, axis-aligned ones, and edges through pixel centres."""
    import ctypes
    rng = np.random.default_rng(314)
    n = 300000
    base = rng.integers(-16384, 16384 - 128, size=(n, 2)) & ~127                     # tile origins: multiples of 8 px
    v0 = rng.integers(-16384, 16385, size=(n, 2))
    v1 = rng.integers(-16384, 16385, size=(n, 2))
    near = rng.uniform(size=n) < 0.5                                                  # half of the edges close to their tile
    v0[near] = base[near] + rng.integers(-200, 328, size=(int(near.sum()), 2))
    v1[near] = v0[near] + rng.integers(-400, 401, size=(int(near.sum()), 2))
    e = np.concatenate([v0 - base, v1 - v0], 1)
    # axis-aligned and pixel-centre-aligned edges (the fill rule's tie cases)
    k = 100000
    o = rng.integers(-20, 140, size=(k, 2)) * rng.choice([1, 8, 16], size=(k, 1))
    d = rng.integers(-64, 65, size=(k, 2)) * rng.choice([1, 16], size=(k, 1))
    d[rng.uniform(size=k) < 0.3, 0] = 0
    d[rng.uniform(size=k) < 0.3, 1] = 0
    e = np.ascontiguousarray(np.concatenate([e, np.concatenate([o, d], 1)], 0).astype(np.int32))
    e = e[(e[:, 2] != 0) | (e[:, 3] != 0)]
    m = np.zeros((e.shape[0], 2), np.uint64)
    rc = ref.lib().nvdr_ref_cover8x8_probe(e.ctypes.data_as(ctypes.c_void_p), m.ctypes.data_as(ctypes.c_void_p), int(e.shape[0]))
    assert rc == 0
    assert (m[:, 0] != m[:, 1]).sum() == 0, "LUT path and non-LUT statement disagree"
    ox, oy, dx, dy = [e[:, c].astype(np.int64) for c in range(4)]
    excl = ((dy > 0) | ((dy == 0) & (dx <= 0))).astype(np.int64)
    rule = np.zeros(e.shape[0], np.uint64)
    for y in range(8):
        for x in range(8):
            rule |= (((ox - 16 * x) * dy - (oy - 16 * y) * dx - excl) >= 0).astype(np.uint64) << np.uint64(x + 8 * y)
    assert (rule != m[:, 0]).sum() == 0, "integer fill rule differs from the reference's LUT coverage"
    assert 0.05 < float((rule != 0).mean()) < 0.95                                    # the sample is not trivial


def test_integer_depth_surfaces_are_identical(po):
    """Beyond "the same triangle wins": the U32 depth of every covered pixel equals the value in the reference's depth
    buffer (read back from its CudaRaster context) -- depth-plane setup (Util.inl:184-210), clipped sub-triangles,
    vertex depths that float rounding pushed past 2^32 (the conversion saturates on CUDA), range mode."""
    b = m10k_batch(2, seed=11)
    po.rasterize_ids(b["pos"], b["tri"], (250, 333))
    s = stress_triangles(2, T=3000, res=192, seed=3)
    po.rasterize_ids(s["pos"], s["tri"], (192, 192))
    rng = np.random.default_rng(5)
    T = 600
    pos = rng.normal(size=(2, 3 * T, 4)).astype(np.float32) * np.array([2.0, 2.0, 1.5, 1.0], np.float32)
    pos[..., 3] = rng.uniform(-0.5, 2.0, size=pos.shape[:2])
    po.rasterize_ids(pos, np.arange(3 * T, dtype=np.int32).reshape(T, 3), (128, 200))
    b = m10k_batch(1, seed=4, nx=30, ny=20)
    po.rasterize_ids(b["pos"][0], b["tri"], (96, 128), ranges=np.array([[0, 1160], [100, 500]], np.int32))
    # the triangle the randomised test found: near- and far-clipped, one vertex depth above 2^32 after rounding
    tri264 = np.array([[[4.076959133148193, -4.071649074554443, -4.839981555938721, 2.0703206062316895],
                        [0.7489162087440491, 1.5174667835235596, 1.0366004705429077, 0.7550548315048218],
                        [-0.015080906450748444, 0.00025000300956889987, 0.01805366948246956, 0.012354218401014805]]], np.float32)
    ids, _ = po.rasterize_ids(tri264, np.array([[0, 1, 2]], np.int32), (118, 10))
    assert (ids > 0).sum() > 500
