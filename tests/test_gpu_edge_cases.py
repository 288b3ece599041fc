"""GPU edge cases: inputs the reference guards against explicitly (corrupt indices, empty ranges,
degenerate and non-finite geometry, zero gradients) and size extremes.  Checked against the oracle
wherever it defines the result."""
import numpy as np
import pytest
import torch

from nvdiffrast_amd.utils import m10k_batch

pytestmark = pytest.mark.gpu


def _t(a, dev="cuda"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _ids_equal(dr, oracle, pos, tri, res, ranges=None):
    ctx = dr.RasterizeCudaContext()
    r, rdb = dr.rasterize(ctx, _t(pos), _t(tri), res, ranges=None if ranges is None else torch.from_numpy(ranges))
    ro, rdbo = oracle.rasterize(pos, tri, res, ranges=ranges)
    r = r.cpu().numpy()
    assert (r[..., 3] != ro[..., 3]).sum() == 0
    ok = np.isfinite(ro[..., :3]).all(-1) & np.isfinite(r[..., :3]).all(-1)   # non-finite geometry: the NaN pattern is not specified
    assert np.abs(r[..., :3][ok] - ro[..., :3][ok]).max(initial=0.0) <= 1e-5
    return r, ro


def test_empty_and_partial_ranges(dr, oracle):
    b = m10k_batch(1, seed=3, nx=20, ny=10)
    pos = b["pos"][0]
    T = b["tri"].shape[0]
    ranges = np.array([[0, 0], [5, 1], [T - 3, 3], [0, T]], np.int32)
    r, ro = _ids_equal(dr, oracle, pos, b["tri"], (64, 64), ranges)
    assert (r[0, ..., 3] == 0).all()                      # zero triangles -> background only
    assert set(np.unique(r[1, ..., 3])) <= {0.0, 6.0}     # ids stay global triangle numbers (+1)


def test_corrupt_indices_are_skipped(dr, oracle):
    b = m10k_batch(2, seed=4, nx=16, ny=8)
    tri = b["tri"].copy()
    V = b["pos"].shape[1]
    tri[3] = [0, V, 1]; tri[10] = [-1, 2, 3]; tri[11] = [5, 5, 5]; tri[12] = [7, 8, 7]
    r, ro = _ids_equal(dr, oracle, b["pos"], tri, (96, 80))
    assert not np.isin(r[..., 3], [4.0, 11.0, 12.0, 13.0]).any()
    # interpolate / antialias tolerate the same table (their own index guards)
    out, _ = dr.interpolate(_t(b["attr"]), _t(r), _t(tri))
    oo, _ = oracle.interpolate(b["attr"], r, tri)
    assert np.abs(out.cpu().numpy() - oo).max() <= 1e-5
    col = np.random.default_rng(0).uniform(size=r.shape[:3] + (3,)).astype(np.float32)
    aa = dr.antialias(_t(col), _t(r), _t(b["pos"]), _t(tri))
    assert np.abs(aa.cpu().numpy() - oracle.antialias(col, r, b["pos"], tri)).max() <= 1e-5


def test_degenerate_and_nonfinite_geometry(dr, oracle):
    rng = np.random.default_rng(5)
    pos = rng.uniform(-1, 1, size=(1, 60, 4)).astype(np.float32)
    pos[..., 3] = rng.uniform(0.5, 2.0, size=(1, 60))
    pos[0, 3] = pos[0, 4]                                   # zero-area triangles
    pos[0, 6, 3] = 0.0                                      # w = 0
    pos[0, 9, 0] = np.nan
    pos[0, 12, 1] = np.inf
    pos[0, 15, 3] = -1.0                                    # behind the eye
    pos[0, 18:21, :2] *= 1e4                                # far outside the viewport
    tri = np.arange(60, dtype=np.int32).reshape(20, 3)
    r, ro = _ids_equal(dr, oracle, pos, tri, (40, 40))
    assert np.isfinite(r[..., 3]).all()
    # backward on the same scene must not produce NaN where the oracle does not
    ctx = dr.RasterizeCudaContext()
    p = _t(pos).requires_grad_(True)
    rr, rdb = dr.rasterize(ctx, p, _t(tri), (40, 40))
    G = rng.normal(size=rr.shape).astype(np.float32)
    (rr * _t(G)).sum().backward()
    go = oracle.rasterize_grad(pos, tri, ro, G, ddb=np.zeros_like(G))
    g = p.grad.cpu().numpy()
    fin = np.isfinite(go) & np.isfinite(g)
    assert fin.mean() > 0.8
    assert np.abs(g[fin] - go[fin]).max() <= 1e-5 * max(1.0, np.abs(go[fin]).max())


def test_zero_upstream_gradients(dr):
    b = m10k_batch(1, seed=6, nx=12, ny=6)
    pos = _t(b["pos"]).requires_grad_(True)
    attr = _t(b["attr"]).requires_grad_(True)
    tri = _t(b["tri"])
    ctx = dr.RasterizeCudaContext()
    rast, rast_db = dr.rasterize(ctx, pos, tri, (32, 32))
    out, _ = dr.interpolate(attr, rast, tri)
    (out * 0.0).sum().backward()
    assert float(pos.grad.abs().max()) == 0.0 and float(attr.grad.abs().max()) == 0.0
    tex = torch.rand(1, 8, 8, 3, device="cuda", requires_grad=True)
    uv = torch.rand(1, 5, 5, 2, device="cuda", requires_grad=True)
    da = (torch.rand(1, 5, 5, 4, device="cuda") * 0.1).requires_grad_(True)
    o = dr.texture(tex, uv, da)
    (o * 0.0).sum().backward()
    assert float(tex.grad.abs().max()) == 0.0 and float(uv.grad.abs().max()) == 0.0 and float(da.grad.abs().max()) == 0.0


def test_size_extremes(dr, oracle):
    tri = np.array([[0, 1, 2]], np.int32)
    pos = np.array([[[-1, -1, 0, 1], [3, -1, 0, 1], [-1, 3, 0, 1]]], np.float32)      # covers the whole viewport
    for res in [(1, 1), (1, 7), (9, 1), (3, 2048), (2050, 5)]:
        r, ro = _ids_equal(dr, oracle, pos, tri, res)
        assert (r[..., 3] == 1).all()
    # many images, one pixel each
    posN = np.repeat(pos, 300, 0)
    r, ro = _ids_equal(dr, oracle, posN, tri, (1, 1))
    assert r.shape == (300, 1, 1, 4)
    # a single triangle index table entry referencing the same vertex thrice renders nothing
    r, ro = _ids_equal(dr, oracle, pos, np.array([[1, 1, 1]], np.int32), (8, 8))
    assert (r[..., 3] == 0).all()


def test_many_attributes_and_diff_list(dr, oracle):
    b = m10k_batch(2, seed=8, nx=16, ny=8, attrs=37)
    ro, rdbo = oracle.rasterize(b["pos"], b["tri"], (48, 48))
    lst = [0, -1, 5, 36, 100, -40]                       # out-of-range entries yield zeros (interpolate.cu:102-106)
    attr = _t(b["attr"]).requires_grad_(True)
    out, da = dr.interpolate(attr, _t(ro), _t(b["tri"]), rast_db=_t(rdbo), diff_attrs=lst)
    oo, dao = oracle.interpolate(b["attr"], ro, b["tri"], rast_db=rdbo, diff_attrs=lst)
    assert np.abs(out.detach().cpu().numpy() - oo).max() <= 1e-5
    assert np.abs(da.detach().cpu().numpy() - dao).max() <= 1e-5 * max(1.0, np.abs(dao).max())
    rng = np.random.default_rng(9)
    G = rng.normal(size=oo.shape).astype(np.float32); Gda = rng.normal(size=dao.shape).astype(np.float32)
    ((out * _t(G)).sum() + (da * _t(Gda)).sum()).backward()
    ga, _, _ = oracle.interpolate_grad(b["attr"], ro, b["tri"], G, rast_db=rdbo, dda=Gda, diff_attrs=lst)
    assert np.abs(attr.grad.cpu().numpy() - ga).max() <= 1e-5 * max(1.0, np.abs(ga).max())
    with pytest.raises(RuntimeError, match="too many entries in diff_attrs"):
        dr.interpolate(attr, _t(ro), _t(b["tri"]), rast_db=_t(rdbo), diff_attrs=list(range(33)))


def test_error_messages_match_the_reference(dr):
    ctx = dr.RasterizeCudaContext()
    pos = torch.zeros(1, 3, 4, device="cuda"); tri = torch.zeros(1, 3, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError, match=r"resolution must be \[>0, >0\]"):
        dr.rasterize(ctx, pos, tri, (0, 8))
    with pytest.raises(RuntimeError, match="must be int32 tensors"):
        dr.rasterize(ctx, pos, tri.long(), (8, 8))
    with pytest.raises(RuntimeError, match=r"instance mode - pos must have shape \[>0, >0, 4\]"):
        dr.rasterize(ctx, torch.zeros(1, 3, 3, device="cuda"), tri, (8, 8))
    with pytest.raises(RuntimeError, match="must be contiguous tensors"):
        dr.rasterize(ctx, torch.zeros(1, 4, 3, device="cuda").transpose(1, 2), tri, (8, 8))
    with pytest.raises(RuntimeError, match=r"range mode - ranges must have shape \[>0, 2\]"):
        dr.rasterize(ctx, torch.zeros(3, 4, device="cuda"), tri, (8, 8))
    with dr.DepthPeeler(ctx, pos, tri, (8, 8)) as peeler:
        assert isinstance(dr.rasterize(ctx, pos, tri, (8, 8)), RuntimeError)      # returned, not raised (ops.py:131-132)
        with pytest.raises(RuntimeError, match="multiple depth peelers"):
            dr.DepthPeeler(ctx, pos, tri, (8, 8)).__enter__()
        peeler.rasterize_next_layer()


def test_context_reuse_across_layouts_and_scenes(dr, oracle):
    """One context, many calls: the rasterizer's control block is cleaned by the call itself (no memset
    per call, include/nvdr_hip.h `scratch_clean`).  Alternating scenes, batch sizes, resolutions (incl. a
    tiled >2048 viewport and a clipped scene that uses the sub-triangle pool) and repeated identical
    calls must all reproduce the oracle."""
    ctx = dr.RasterizeCudaContext()
    a = m10k_batch(3, seed=11, nx=24, ny=12)
    b = m10k_batch(2, seed=12, nx=10, ny=30)
    near = b["pos"].copy(); near[..., 2] -= 0.8 * np.abs(near[..., 3])          # pushes part of the mesh through the near plane
    cases = [(a["pos"], a["tri"], (72, 120)), (a["pos"], a["tri"], (72, 120)), (b["pos"], b["tri"], (72, 120)),
             (near, b["tri"], (64, 64)), (near, b["tri"], (64, 64)), (a["pos"][:1], a["tri"], (200, 136)),
             (a["pos"], a["tri"], (72, 120)), (b["pos"][:1], b["tri"], (8, 2056)), (b["pos"][:1], b["tri"], (8, 2056)),
             (a["pos"], a["tri"], (72, 120))]
    want = {}
    for i, (pos, tri, res) in enumerate(cases):
        key = (pos.tobytes()[:64], pos.shape, res)
        if key not in want:
            want[key] = oracle.rasterize(pos, tri, res)[0]
        r, _ = dr.rasterize(ctx, _t(pos), _t(tri), res)
        r = r.cpu().numpy()
        assert (r[..., 3] != want[key][..., 3]).sum() == 0, f"call {i}"
        assert np.abs(r[..., :3] - want[key][..., :3]).max() <= 1e-5, f"call {i}"


def _encode_ids(ids):
    """triidx_to_float (common.h:192-193): exact floats up to 2^24, bit-offset floats above."""
    ids = np.asarray(ids, np.int64)
    small = ids.astype(np.float32)
    big = (np.int64(0x4A800000) + ids).astype(np.uint32).view(np.float32)
    return np.where(ids <= 0x01000000, small, big).astype(np.float32)


def test_triangle_ids_above_2_pow_24_survive_the_float_channel(dr, raw_oracle):
    """common.h:186-193: a triangle id above 2^24 cannot be stored in an f32 as a number, so the id channel carries
    it as a bit-offset float.  A mesh of 2^24 + 16 triangles whose LAST 16 are visible: interpolate (fwd + grad),
    rasterize_grad and antialias (fwd + grad) must decode the ids exactly as the oracle does.
    The rast tensor is made from a rasterization of the 16 visible triangles with the id channel re-based."""
    from nvdiffrast_amd.torch import _plugin
    oracle = raw_oracle                                     # the reference under its CPU shim would run 2^24 fibres for the hash
    T = (1 << 24) + 16
    small = m10k_batch(1, seed=5, nx=4, ny=2)               # 3 x 5 lattice: 16 triangles, 15 vertices
    assert small["tri"].shape[0] == 16
    res = (96, 96)
    ro, rdbo = oracle.rasterize(small["pos"], small["tri"], res)
    ids = ro[..., 3].astype(np.int64)
    big_ids = np.where(ids > 0, ids + (T - 16), 0)
    assert big_ids.max() > (1 << 24) and (big_ids > (1 << 24)).sum() > 500
    rast = ro.copy()
    rast[..., 3] = _encode_ids(big_ids)
    tri = np.zeros((T, 3), np.int32)
    tri[T - 16:] = small["tri"]
    rng = np.random.default_rng(3)
    A = 4
    attr = rng.uniform(size=(1, small["pos"].shape[1], A)).astype(np.float32)
    G = rng.normal(size=(1,) + res + (A,)).astype(np.float32)
    t_tri, t_rast = _t(tri), _t(rast)

    # interpolate forward + backward
    t_attr = _t(attr).requires_grad_(True)
    t_rast_g = _t(rast).requires_grad_(True)
    out, _ = dr.interpolate(t_attr, t_rast_g, t_tri)
    out.backward(_t(G))
    oo, _ = oracle.interpolate(attr, rast, tri)
    ga, gr, _ = oracle.interpolate_grad(attr, rast, tri, G)
    assert np.abs(oo).max() > 0.1                                                     # the big ids really resolve to triangles
    assert np.abs(out.detach().cpu().numpy() - oo).max() <= 1e-5
    assert np.abs(t_attr.grad.cpu().numpy() - ga).max() <= 1e-5 * max(1.0, np.abs(ga).max())
    assert np.abs(t_rast_g.grad.cpu().numpy() - gr).max() <= 1e-5 * max(1.0, np.abs(gr).max())

    # rasterize backward (with the pixel-differential gradients)
    dy = rng.normal(size=rast.shape).astype(np.float32)
    ddb = rng.normal(size=rast.shape).astype(np.float32)
    gp = _plugin.rasterize_grad_db(_t(small["pos"]), t_tri, t_rast, _t(dy), _t(ddb)).cpu().numpy()
    gpo = oracle.rasterize_grad(small["pos"], tri, rast, dy, ddb)
    assert np.abs(gpo).max() > 0 and np.abs(gp - gpo).max() <= 1e-5 * max(1.0, np.abs(gpo).max())

    # antialias forward + backward (topology hash over all 2^24 + 16 triangles)
    col = rng.uniform(size=(1,) + res + (3,)).astype(np.float32)
    t_col = _t(col).requires_grad_(True)
    t_pos = _t(small["pos"]).requires_grad_(True)
    aa = dr.antialias(t_col, t_rast, t_pos, t_tri)
    dya = rng.normal(size=col.shape).astype(np.float32)
    aa.backward(_t(dya))
    aao = oracle.antialias(col, rast, small["pos"], tri)
    gc, gpa = oracle.antialias_grad(col, rast, small["pos"], tri, dya)
    assert np.abs(aao - col).max() > 1e-3                                             # silhouettes were found and blended
    assert np.abs(aa.detach().cpu().numpy() - aao).max() <= 1e-5
    assert np.abs(t_col.grad.cpu().numpy() - gc).max() <= 1e-5 * max(1.0, np.abs(gc).max())
    assert np.abs(t_pos.grad.cpu().numpy() - gpa).max() <= 1e-5 * max(1.0, np.abs(gpa).max())
