"""GPU parity: HIP rasterize / interpolate (through the C ABI) vs the CPU oracle.

Bars (north_star): triangle-id channel bit-exact; barycentrics / z/w / attribute values and
gradients within 1e-5 abs (position gradients, whose magnitude is O(100), within
1e-5 * max(1, |g|_inf) -- an f32 atomic sum cannot do better than its own ulp)."""
import numpy as np
import pytest
import torch

from nvdiffrast_amd.utils import m10k_batch, stress_triangles

pytestmark = pytest.mark.gpu

ATOL = 1e-5


def _t(a, dev="cuda"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _raster_pair(dr, oracle, pos, tri, res, ranges=None):
    ctx = dr.RasterizeCudaContext()
    r, rdb = dr.rasterize(ctx, _t(pos), _t(tri), res, ranges=None if ranges is None else torch.from_numpy(ranges))
    ro, rdbo = oracle.rasterize(pos, tri, res, ranges=ranges)
    return r.cpu().numpy(), rdb.cpu().numpy(), ro, rdbo


def _check_raster(r, rdb, ro, rdbo, db_tol=None):
    assert r.shape == ro.shape
    mism = int((r[..., 3] != ro[..., 3]).sum())
    assert mism == 0, f"{mism} triangle-id mismatches"
    assert np.abs(r[..., :3] - ro[..., :3]).max() <= ATOL
    if db_tol is None:
        db_tol = ATOL * max(1.0, float(np.abs(rdbo).max()))
    assert np.abs(rdb - rdbo).max() <= db_tol


def test_triangle_sample(dr, oracle):
    """BASELINE config 1: samples/torch/triangle.py inputs; oracle is pinned to tri.png."""
    pos = np.array([[[-0.8, -0.8, 0, 1], [0.8, -0.8, 0, 1], [-0.8, 0.8, 0, 1]]], np.float32)
    col = np.array([[[1, 0, 0], [0, 1, 0], [0, 0, 1]]], np.float32)
    tri = np.array([[0, 1, 2]], np.int32)
    r, rdb, ro, rdbo = _raster_pair(dr, oracle, pos, tri, (256, 256))
    _check_raster(r, rdb, ro, rdbo)
    out, _ = dr.interpolate(_t(col), _t(r), _t(tri))
    img = np.clip(np.rint(out.cpu().numpy()[0, ::-1] * 255), 0, 255).astype(np.uint8)
    from PIL import Image
    import os
    g = np.array(Image.open(os.path.join(os.path.dirname(__file__), "golden", "tri.png")))
    assert (img != g).sum() == 0


@pytest.mark.parametrize("res", [(512, 512), (250, 333), (64, 72), (8, 8), (5, 3)])
def test_lattice_mesh_ids_exact(dr, oracle, res):
    b = m10k_batch(3, seed=11)
    r, rdb, ro, rdbo = _raster_pair(dr, oracle, b["pos"], b["tri"], res)
    _check_raster(r, rdb, ro, rdbo)


def test_stress_overdraw(dr, oracle):
    s = stress_triangles(2, T=4000, res=256, seed=3)
    r, rdb, ro, rdbo = _raster_pair(dr, oracle, s["pos"], s["tri"], (256, 256))
    _check_raster(r, rdb, ro, rdbo)


@pytest.mark.parametrize("order", ["front_to_back", "back_to_front", "shuffled"])
def test_depth_cull_under_layered_overdraw(dr, oracle, order):
    """k_fine drops a (triangle, tile) pair whose depth plane lies behind everything a fully covered tile already holds
    (FineRaster.inl:13-34).  Layers of screen-filling quads -- tiles are covered after the first -- with flat, gently and steeply
    tilted depth planes (slopes on both sides of the limit up to which the wrapped U32 plane is evaluated per tile), sheets that
    intersect, exact duplicates (depth ties: the higher id wins) and slivers; ids and z/w must be the reference's whatever the
    order of submission, in one pass and through three peeled layers."""
    rng = np.random.default_rng(77)
    quads = []
    L = 24
    for i in range(L):
        z0 = -0.9 + 1.8 * i / (L - 1)
        kind = i % 4
        tilt = [0.0, 0.02, 0.6, 1.7][kind]                      # flat / gentle / steep / spanning the whole depth range
        zc = np.clip(np.array([z0 - tilt, z0 + tilt * 0.3, z0 + tilt, z0 - tilt * 0.5]), -0.99, 0.99)
        xy = np.array([[-1.1, -1.1], [1.1, -1.1], [1.1, 1.1], [-1.1, 1.1]]) * (1.0 if kind != 3 else 0.7)
        quads.append(np.concatenate([xy, zc[:, None], np.ones((4, 1))], 1))
    quads.append(quads[5].copy())                               # a coplanar duplicate
    sl = stress_triangles(1, T=300, res=128, seed=12)           # small triangles in between, independent depths
    P = np.stack(quads).astype(np.float32)                      # [Q,4,4]
    idx = {"front_to_back": np.arange(len(P)), "back_to_front": np.arange(len(P))[::-1], "shuffled": rng.permutation(len(P))}[order]
    P = P[idx]
    pos = np.concatenate([P.reshape(-1, 4), sl["pos"][0]], 0)[None]
    tq = np.concatenate([np.array([[0, 1, 2], [0, 2, 3]]) + 4 * q for q in range(len(P))], 0)
    tri = np.concatenate([tq, sl["tri"] + 4 * len(P)], 0).astype(np.int32)
    res = (200, 136)                                            # several bins, partial tiles at the borders
    r, rdb, ro, rdbo = _raster_pair(dr, oracle, pos, tri, res)
    _check_raster(r, rdb, ro, rdbo)
    assert (ro[..., 3] > 0).mean() > 0.99
    ctx = dr.RasterizeCudaContext()
    peel = None
    with dr.DepthPeeler(ctx, _t(pos), _t(tri), res) as peeler:
        for layer in range(3):
            rl, rldb = peeler.rasterize_next_layer()
            ol, oldb, peel = oracle.rasterize(pos, tri, res, peel_depth=peel, return_depth=True)
            _check_raster(rl.cpu().numpy(), rldb.cpu().numpy(), ol, oldb)


def test_clipping_and_huge_triangles(dr, oracle):
    """Triangles crossing every frustum plane incl. w<=0 vertices (clipper path, pool slots)."""
    rng = np.random.default_rng(5)
    T = 600
    pos = rng.normal(size=(2, 3 * T, 4)).astype(np.float32) * np.array([2.0, 2.0, 1.5, 1.0], np.float32)
    pos[..., 3] = rng.uniform(-0.5, 2.0, size=pos.shape[:2])
    tri = np.arange(3 * T, dtype=np.int32).reshape(T, 3)
    r, rdb, ro, rdbo = _raster_pair(dr, oracle, pos, tri, (128, 200))
    mism = int((r[..., 3] != ro[..., 3]).sum())
    assert mism == 0
    # vertices with w <= 0 can make a pixel's barycentrics non-finite in the reference's formula; where they are finite
    # the 1e-5 bar holds (measured: 4e-7 against the oracle, 5e-7 against the reference itself)
    ok = np.isfinite(ro).all(-1) & np.isfinite(r).all(-1)
    assert np.abs(r[ok][:, :3] - ro[ok][:, :3]).max() <= ATOL


def test_depth_ties_and_duplicates(dr, oracle):
    """Coplanar duplicates: the highest triangle index must win (FineRaster.inl:152-172)."""
    b = m10k_batch(1, seed=2, nx=20, ny=10)
    tri = np.concatenate([b["tri"], b["tri"][::-1], b["tri"]], 0)
    r, rdb, ro, rdbo = _raster_pair(dr, oracle, b["pos"], tri, (160, 160))
    _check_raster(r, rdb, ro, rdbo)
    assert (r[..., 3][r[..., 3] > 0] > 2 * b["tri"].shape[0]).all()


def test_range_mode(dr, oracle):
    b = m10k_batch(1, seed=4, nx=30, ny=20)
    pos = b["pos"][0]
    T = b["tri"].shape[0]
    ranges = np.array([[0, T], [100, 500], [T - 7, 7], [3, 0]], np.int32)
    r, rdb, ro, rdbo = _raster_pair(dr, oracle, pos, b["tri"], (96, 128), ranges=ranges)
    _check_raster(r, rdb, ro, rdbo)


def test_large_list_multi_round(dr, oracle):
    """More triangles in one 64x64 bin than the LDS list holds -> several filter/raster rounds."""
    rng = np.random.default_rng(9)
    T = 5000
    c = rng.uniform(-0.1, 0.1, size=(T, 1, 2))
    xy = c + rng.uniform(-0.05, 0.05, size=(T, 3, 2))
    z = rng.uniform(-0.9, 0.9, size=(T, 3, 1))
    pos = np.concatenate([xy, z, np.ones_like(z)], -1).reshape(1, -1, 4).astype(np.float32)
    tri = np.arange(3 * T, dtype=np.int32).reshape(T, 3)
    r, rdb, ro, rdbo = _raster_pair(dr, oracle, pos, tri, (256, 256))
    _check_raster(r, rdb, ro, rdbo)


def test_depth_peeling(dr, oracle):
    s = stress_triangles(2, T=1500, res=128, seed=8)
    pos, tri = _t(s["pos"]), _t(s["tri"])
    ctx = dr.RasterizeCudaContext()
    peel = None
    with dr.DepthPeeler(ctx, pos, tri, (128, 128)) as peeler:
        for layer in range(4):
            r, rdb = peeler.rasterize_next_layer()
            ro, rdbo, depth = oracle.rasterize(s["pos"], s["tri"], (128, 128), peel_depth=peel, return_depth=True)
            peel = depth
            _check_raster(r.cpu().numpy(), rdb.cpu().numpy(), ro, rdbo)
    assert isinstance(dr.rasterize(ctx, pos, tri, (128, 128)), tuple)


@pytest.mark.parametrize("A", [1, 2, 3, 4, 7])
def test_interpolate_forward(dr, oracle, A):
    b = m10k_batch(2, seed=6, attrs=A)
    ro, rdbo = oracle.rasterize(b["pos"], b["tri"], (128, 160))
    for attr in (b["attr"], np.repeat(b["attr"], 2, 0) * np.array([1.0, 0.5], np.float32).reshape(2, 1, 1)):
        out, da = dr.interpolate(_t(attr), _t(ro), _t(b["tri"]))
        oo, _ = oracle.interpolate(attr, ro, b["tri"])
        assert da.shape == (2, 128, 160, 0)
        assert np.abs(out.cpu().numpy() - oo).max() <= ATOL
        out, da = dr.interpolate(_t(attr), _t(ro), _t(b["tri"]), rast_db=_t(rdbo), diff_attrs="all")
        oo, dao = oracle.interpolate(attr, ro, b["tri"], rast_db=rdbo, diff_attrs="all")
        assert np.abs(out.cpu().numpy() - oo).max() <= ATOL
        assert np.abs(da.cpu().numpy() - dao).max() <= ATOL * max(1.0, np.abs(dao).max())
    lst = [A - 1, 0, -1]
    out, da = dr.interpolate(_t(b["attr"]), _t(ro), _t(b["tri"]), rast_db=_t(rdbo), diff_attrs=lst)
    oo, dao = oracle.interpolate(b["attr"], ro, b["tri"], rast_db=rdbo, diff_attrs=lst)
    assert np.abs(da.cpu().numpy() - dao).max() <= ATOL * max(1.0, np.abs(dao).max())


def _grad_tol(ref):
    return ATOL * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("grad_db", [True, False])
def test_raster_interp_backward(dr, oracle, grad_db):
    """The benchmark's op graph at a size the oracle finishes quickly."""
    N, res = 3, (256, 256)
    b = m10k_batch(N, seed=21)
    rng = np.random.default_rng(0)
    G = rng.normal(size=(N, res[0], res[1], 4)).astype(np.float32)
    Gdb = rng.normal(size=(N, res[0], res[1], 4)).astype(np.float32) * 0.01
    pos = _t(b["pos"]).requires_grad_(True)
    attr = _t(b["attr"]).requires_grad_(True)
    tri = _t(b["tri"])
    ctx = dr.RasterizeCudaContext()
    rast, rast_db = dr.rasterize(ctx, pos, tri, res, grad_db=grad_db)
    out, _ = dr.interpolate(attr, rast, tri)
    loss = (out * _t(G)).sum() + (rast_db * _t(Gdb)).sum()
    loss.backward()

    ro, rdbo = oracle.rasterize(b["pos"], b["tri"], res)
    g_attr, g_rast, _ = oracle.interpolate_grad(b["attr"], ro, b["tri"], G)
    g_pos = oracle.rasterize_grad(b["pos"], b["tri"], ro, g_rast, ddb=Gdb if grad_db else None)
    assert np.abs(attr.grad.cpu().numpy() - g_attr).max() <= _grad_tol(g_attr)
    assert np.abs(pos.grad.cpu().numpy() - g_pos).max() <= _grad_tol(g_pos)


def test_interpolate_backward_da(dr, oracle):
    N, res = 2, (96, 128)
    b = m10k_batch(N, seed=22, attrs=3)
    rng = np.random.default_rng(1)
    ro, rdbo = oracle.rasterize(b["pos"], b["tri"], res)
    for diff in ("all", [2, 0]):
        D = 3 if diff == "all" else 2
        G = rng.normal(size=(N, res[0], res[1], 3)).astype(np.float32)
        Gda = rng.normal(size=(N, res[0], res[1], 2 * D)).astype(np.float32)
        attr = _t(b["attr"]).requires_grad_(True)
        rast = _t(ro).requires_grad_(True)
        rast_db = _t(rdbo).requires_grad_(True)
        out, da = dr.interpolate(attr, rast, _t(b["tri"]), rast_db=rast_db, diff_attrs=diff)
        ((out * _t(G)).sum() + (da * _t(Gda)).sum()).backward()
        g_attr, g_rast, g_rdb = oracle.interpolate_grad(b["attr"], ro, b["tri"], G, rast_db=rdbo, dda=Gda, diff_attrs=diff)
        assert np.abs(attr.grad.cpu().numpy() - g_attr).max() <= _grad_tol(g_attr)
        assert np.abs(rast.grad.cpu().numpy() - g_rast).max() <= _grad_tol(g_rast)
        assert np.abs(rast_db.grad.cpu().numpy() - g_rdb).max() <= _grad_tol(g_rdb)


def test_viewport_tiling_beyond_2048(dr, oracle):
    """Images larger than one 2048^2 viewport are rasterised in viewport tiles (torch_rasterize.cpp:99-124)."""
    b = m10k_batch(1, seed=30, nx=12, ny=8)
    r, rdb, ro, rdbo = _raster_pair(dr, oracle, b["pos"], b["tri"], (2100, 2500))
    _check_raster(r, rdb, ro, rdbo)


def test_bins_shared_by_several_workgroups(dr, oracle):
    """A mesh squeezed into a narrow strip puts thousands of triangles into single 64x64-px bins; in a small launch such
    bins are shared by up to four workgroups (k_order helper items; k_fine merges the parts' keys through memory and the
    last part shades).  Ids and U32 depths must not notice -- also through depth peeling (the PEEL / depth-surface
    instantiations of the shared path)."""
    b = m10k_batch(2, seed=4)
    pos = b["pos"].copy()
    pos[..., 0] *= 0.04                                     # 10,000 triangles in a strip ~20 px wide: >2000 per bin
    res = (512, 512)
    r, rdb, ro, rdbo = _raster_pair(dr, oracle, pos, b["tri"], res)
    ids = ro[..., 3]
    counts = np.bincount(ids[ids > 0].astype(np.int64)).size
    assert counts > 1000                                    # the strip really shows many different triangles
    _check_raster(r, rdb, ro, rdbo)

    layers = oracle.rasterize_layers(pos, b["tri"], res, 3)
    ctx = dr.RasterizeCudaContext()
    with dr.DepthPeeler(ctx, _t(pos), _t(b["tri"]), res) as peeler:
        for k in range(3):
            rk, _ = peeler.rasterize_next_layer()
            want, _, depth = layers[k]
            assert (rk[..., 3].cpu().numpy() != want[..., 3]).sum() == 0, "layer %d" % k
            got_depth = ctx.cpp_wrapper.depth.cpu().numpy().view(np.uint32)[:, :res[0], :res[1]]
            cov = want[..., 3] > 0
            assert (got_depth[cov] != np.asarray(depth).view(np.uint32)[:, :res[0], :res[1]][cov]).sum() == 0, "layer %d depth" % k


def test_shared_bins_stress_with_poisoned_exchange_buffers(dr, oracle):
    """200 calls through the shared-bin path: four strip scenes in rotation, the exchange buffers
    overwritten with zeros before every call (a stale key of 0 would win every minimum, so a part that read keys before
    their owner had published them shows up as a wrong id), and unrelated work on a second stream to vary which part of a
    bin arrives last.  Ids identical to the oracle's in every call."""
    scenes = []
    for k, squeeze in enumerate((0.04, 0.025, 0.06, 0.035)):
        b = m10k_batch(2, seed=40 + k)
        pos = b["pos"].copy()
        pos[..., k % 2] *= squeeze                          # vertical and horizontal strips
        ro, _ = oracle.rasterize(pos, b["tri"], (512, 512))
        scenes.append((_t(pos), _t(b["tri"]), _t(ro[..., 3].copy())))
    ctx = dr.RasterizeCudaContext()
    side = torch.cuda.Stream()
    noise = torch.empty(1 << 22, device="cuda")
    bad = 0
    for it in range(200):
        pos, tri, want = scenes[it % 4]
        ctx.cpp_wrapper.poison_scratch(0)
        with torch.cuda.stream(side):
            for _ in range(it % 5):
                noise.add_(1.0)
        r, _ = dr.rasterize(ctx, pos, tri, (512, 512))
        bad += int((r[..., 3] != want).sum().item())
    torch.cuda.synchronize()
    assert bad == 0


@pytest.mark.parametrize("res", [(512, 512), (2048, 2048), (200, 328)])
def test_long_and_short_edges_share_waves(dr, oracle, res):
    """k_fine walks coverage two pixels per instruction in 16-bit halves when every edge of a wave's 64 (triangle, tile)
    pairs is at most 125 px long, and in 32 bits otherwise (raster.hip raster_pairs; tests/test_coverage_pk16.py states the
    bound).  A soup that interleaves triangles of 1 ... 8 px with triangles of 100 ... 1500 px -- so that waves hold both kinds,
    tiles are covered completely by the long ones (the immediate depth bound) and the short ones pass through both walks --
    must give the reference's ids and U32 depths at an ordinary size, at the largest single viewport, and at a size that
    is no multiple of the tile."""
    rng = np.random.default_rng(res[0] * 7 + res[1])
    H, W = res
    T = 1200
    c = rng.uniform(-1.05, 1.05, size=(T, 1, 2))
    small = rng.uniform(1.0, 8.0, size=(T, 1, 1))
    large = np.exp(rng.uniform(np.log(100.0), np.log(1500.0), size=(T, 1, 1)))
    size = np.where((np.arange(T) % 3 == 0)[:, None, None], large, small) * (2.0 / max(H, W))
    ang = rng.uniform(0, 2 * np.pi, size=(T, 1, 1)) + np.array([0, 2.1, 4.2]).reshape(1, 3, 1) + rng.uniform(-0.6, 0.6, size=(T, 3, 1))
    xy = c + size * np.concatenate([np.cos(ang), np.sin(ang)], -1)
    z = rng.uniform(-0.95, 0.95, size=(T, 3, 1))
    pos = np.concatenate([xy, z, np.ones_like(z)], -1).reshape(1, -1, 4).astype(np.float32)
    # snap some vertices onto pixel centres / corners: edges exactly through samples
    pos[0, ::7, :2] = np.round(pos[0, ::7, :2] * (W / 2)) / (W / 2)
    tri = np.arange(3 * T, dtype=np.int32).reshape(T, 3)
    ids_o, depth_o = oracle.rasterize_ids(pos, tri, res)
    ctx = dr.RasterizeCudaContext()
    with dr.DepthPeeler(ctx, _t(pos), _t(tri), res) as peeler:
        r, _ = peeler.rasterize_next_layer()
        r2, _ = peeler.rasterize_next_layer()
    depth = ctx.cpp_wrapper.peel.cpu().numpy().view(np.uint32)            # layer 0's surface (the peeler swapped them)
    got = r[..., 3].cpu().numpy()
    assert (got != ids_o[:, :H, :W].astype(np.float32)).sum() == 0, "triangle ids differ"
    cov = ids_o[:, :H, :W] > 0
    assert cov.mean() > 0.5
    assert (depth[:, :H, :W][cov] != depth_o[:, :H, :W][cov]).sum() == 0, "U32 depth surface differs"
    plain, _ = dr.rasterize(ctx, _t(pos), _t(tri), res)                    # the non-peeling instantiation
    assert torch.equal(plain[..., 3], r[..., 3])
    assert (r2[..., 3] != r[..., 3]).any()                                 # the second layer exists and is another one
