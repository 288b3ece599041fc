"""Per-bin triangle lists of large meshes (raster.hip k_binscan / k_binfill / k_fine<..., LIST>; the reference's bin stage,
BinRaster.inl:60-170,319-377).  Meshes of 32768 triangles or more get a list per 64x64-pixel bin wherever scanning the bin's
range of triangle slots would cost more; ids must not notice -- in index order, shuffled, in range mode, with clipped
triangles in the pool, with more (triangle, bin) pairs than the list buffer holds, and through depth peeling."""
import numpy as np
import pytest
import torch

from nvdiffrast_amd.utils import big_mesh_batch, stress_triangles

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _ids(dr, pos, tri, res, ranges=None):
    ctx = dr.RasterizeCudaContext()
    r, _ = dr.rasterize(ctx, _t(pos), _t(tri), res, ranges=None if ranges is None else torch.from_numpy(ranges))
    return r.cpu().numpy()


@pytest.mark.parametrize("shuffle", [False, True])
def test_large_mesh_ids_match_the_oracle(dr, raw_oracle, shuffle):
    b = big_mesh_batch(2, nx=300, ny=200, shuffle=shuffle)               # 120 000 triangles
    res = (512, 384)
    ro, _ = raw_oracle.rasterize(b["pos"], b["tri"], res)
    r = _ids(dr, b["pos"], b["tri"], res)
    assert (r[..., 3] != ro[..., 3]).sum() == 0
    assert np.abs(r[..., :3] - ro[..., :3]).max() <= 1e-5
    assert (ro[..., 3] > 0).mean() > 0.1


def test_shuffled_and_ordered_meshes_render_the_same_surface(dr):
    a = big_mesh_batch(1, nx=256, ny=128, shuffle=False)
    s = big_mesh_batch(1, nx=256, ny=128, shuffle=True)
    ra = _ids(dr, a["pos"], a["tri"], (256, 256))
    rs = _ids(dr, s["pos"], s["tri"], (256, 256))
    assert ((ra[..., 3] > 0) == (rs[..., 3] > 0)).all()
    # same triangles under other numbers: look the vertex triples up
    ia = a["tri"][ra[..., 3].astype(np.int64)[ra[..., 3] > 0] - 1]
    is_ = s["tri"][rs[..., 3].astype(np.int64)[rs[..., 3] > 0] - 1]
    # interior pixels agree; where two triangles tie in depth the higher INDEX wins, and the indices differ between the two
    assert (ia != is_).any(axis=1).mean() < 1e-3


def test_large_soup_with_clipped_triangles_and_big_triangles(dr, raw_oracle):
    """Independent triangles of all sizes (some cover many bins, some cross the frustum: pool slots) in numbers that switch the
    lists on."""
    rng = np.random.default_rng(11)
    b = stress_triangles(1, T=40000, res=256, seed=5)
    pos = b["pos"].copy()
    big = rng.choice(40000, size=60, replace=False)                     # a few triangles blown up beyond the frustum
    for t in big:
        c = pos[0, 3 * t:3 * t + 3, :2].mean(0)
        pos[0, 3 * t:3 * t + 3, :2] = c + (pos[0, 3 * t:3 * t + 3, :2] - c) * 40.0
    pos[0, 3 * big[:20], 2] = 1.5                                        # and some through the far plane
    res = (256, 256)
    ro, _ = raw_oracle.rasterize(pos, b["tri"], res)
    r = _ids(dr, pos, b["tri"], res)
    assert (r[..., 3] != ro[..., 3]).sum() == 0


def test_large_mesh_in_range_mode(dr, raw_oracle):
    b = big_mesh_batch(1, nx=200, ny=100, shuffle=True)                  # 40 000 triangles
    pos = b["pos"][0]
    T = b["tri"].shape[0]
    ranges = np.array([[0, T], [1000, 35000], [T - 100, 100]], np.int32)
    res = (192, 320)
    ro, _ = raw_oracle.rasterize(pos, b["tri"], res, ranges=ranges)
    r = _ids(dr, pos, b["tri"], res, ranges=ranges)
    assert (r[..., 3] != ro[..., 3]).sum() == 0


def test_large_mesh_depth_peeling(dr, raw_oracle):
    b = stress_triangles(1, T=35000, res=128, seed=9)
    ctx = dr.RasterizeCudaContext()
    peel = None
    with dr.DepthPeeler(ctx, _t(b["pos"]), _t(b["tri"]), (128, 128)) as peeler:
        for _ in range(3):
            r, _ = peeler.rasterize_next_layer()
            ro, _, depth = raw_oracle.rasterize(b["pos"], b["tri"], (128, 128), peel_depth=peel, return_depth=True)
            peel = depth
            assert (r.cpu().numpy()[..., 3] != ro[..., 3]).sum() == 0
