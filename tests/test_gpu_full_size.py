"""Full-size runs (BASELINE.json sizes) checked through size-independent properties, because the CPU
oracle would take minutes there: value ranges, conservation laws, linearity of the backward pass,
forward determinism, and agreement of the oracle on a sample of the items."""
import numpy as np
import pytest
import torch

from nvdiffrast_amd.utils import m10k_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def headline(dr):
    N, R = 64, 512
    b = m10k_batch(N, seed=20240, attrs=4)
    dev = torch.device("cuda")
    t = dict(pos=torch.from_numpy(b["pos"]).to(dev), tri=torch.from_numpy(b["tri"]).to(dev),
             attr=torch.from_numpy(b["attr"]).to(dev), uv=torch.from_numpy(b["uv"]).to(dev))
    return b, t, N, R


def test_headline_config_properties(dr, oracle, headline):
    b, t, N, R = headline
    ctx = dr.RasterizeCudaContext()
    pos = t["pos"].clone().requires_grad_(True)
    attr = t["attr"].clone().requires_grad_(True)
    rast, rast_db = dr.rasterize(ctx, pos, t["tri"], (R, R))
    out, _ = dr.interpolate(attr, rast, t["tri"])
    ids = rast[..., 3].detach()
    cov = ids > 0
    T = t["tri"].shape[0]
    assert float(ids.min()) == 0.0 and float(ids.max()) <= T and torch.equal(ids, ids.round())
    assert 0.2 < float(cov.float().mean()) < 0.7
    u, v = rast[..., 0], rast[..., 1]
    assert float(u.min()) >= 0.0 and float(v.min()) >= 0.0 and float((u + v).max()) <= 1.0 + 1e-6
    assert float(rast[..., 2].abs().max()) <= 1.0
    assert float(rast[~cov].abs().max()) == 0.0 and float(rast_db[~cov].abs().max()) == 0.0 and float(out[~cov].abs().max()) == 0.0
    # attributes are convex combinations of vertex attributes in [0,1]
    assert float(out.min()) >= -1e-6 and float(out.max()) <= 1.0 + 1e-6

    # forward is deterministic (visibility is an order-independent minimum)
    rast2, rast_db2 = dr.rasterize(ctx, pos, t["tri"], (R, R))
    assert torch.equal(rast, rast2) and torch.equal(rast_db, rast_db2)

    # conservation: barycentric weights sum to one, so the attribute gradient sums to the upstream
    # gradient over covered pixels
    G = torch.randn_like(out)
    (g_attr,) = torch.autograd.grad(out, attr, G, retain_graph=True)
    expect = (G * cov[..., None]).double().sum((0, 1, 2))
    got = g_attr.double().sum((0, 1))
    assert torch.allclose(got, expect, rtol=1e-4, atol=1e-2)

    # the backward pass is linear in the upstream gradient
    G2 = torch.randn_like(out)
    gp1, ga1 = torch.autograd.grad(out, (pos, attr), G, retain_graph=True)
    gp2, ga2 = torch.autograd.grad(out, (pos, attr), G2, retain_graph=True)
    gp3, ga3 = torch.autograd.grad(out, (pos, attr), G + 2.0 * G2, retain_graph=True)
    assert torch.allclose(ga3, ga1 + 2.0 * ga2, rtol=1e-4, atol=1e-4 * float(ga3.abs().max()))
    assert torch.allclose(gp3, gp1 + 2.0 * gp2, rtol=1e-4, atol=1e-4 * float(gp3.abs().max()))
    assert float(gp1[..., 2].abs().max()) == 0.0            # no gradient reaches clip-space z

    # a sample of the items against the oracle (ids exact, gradients to tolerance)
    sel = [0, 31, 63]
    ro, _ = oracle.rasterize(b["pos"][sel], b["tri"], (R, R))
    r = rast[sel].detach().cpu().numpy()
    assert (r[..., 3] != ro[..., 3]).sum() == 0
    assert np.abs(r[..., :3] - ro[..., :3]).max() <= 1e-5
    Gs = G[sel].cpu().numpy()
    _, gr, _ = oracle.interpolate_grad(b["attr"], ro, b["tri"], Gs)
    gpo = oracle.rasterize_grad(b["pos"][sel], b["tri"], ro, gr)
    assert np.abs(gp1[sel].cpu().numpy() - gpo).max() <= 1e-5 * max(1.0, np.abs(gpo).max())


def test_range_mode_equals_instanced_at_full_size(dr, headline):
    b, t, N, R = headline
    ctx = dr.RasterizeCudaContext()
    n = 8
    V, T = b["pos"].shape[1], b["tri"].shape[0]
    # one vertex buffer holding n copies of the mesh, one triangle range per image
    pos_flat = t["pos"][:n].reshape(n * V, 4).contiguous()
    tri_flat = torch.cat([t["tri"] + i * V for i in range(n)], 0).contiguous()
    ranges = torch.tensor([[i * T, T] for i in range(n)], dtype=torch.int32)
    ra, _ = dr.rasterize(ctx, pos_flat, tri_flat, (R, R), ranges=ranges)
    rb, _ = dr.rasterize(ctx, t["pos"][:n].contiguous(), t["tri"], (R, R))
    ida = ra[..., 3]; idb = rb[..., 3]
    off = torch.arange(n, device=ida.device, dtype=ida.dtype).view(n, 1, 1) * T
    assert torch.equal(torch.where(ida > 0, ida - off, ida), idb)
    assert torch.equal(ra[..., :3], rb[..., :3])


def test_config3_full_size_properties(dr, headline):
    """batch 32 @1024^2, 2048^2 mipmapped texture + antialias: finite, bounded, conservative."""
    b, t, N, R = headline
    N, R = 32, 1024
    ctx = dr.RasterizeCudaContext()
    pos = t["pos"][:N].clone().requires_grad_(True)
    tex = torch.rand(1, 2048, 2048, 3, device="cuda").requires_grad_(True)
    rast, rast_db = dr.rasterize(ctx, pos, t["tri"], (R, R))
    uv, uv_da = dr.interpolate(t["uv"], rast, t["tri"], rast_db=rast_db, diff_attrs="all")
    col = dr.texture(tex, uv, uv_da, filter_mode="linear-mipmap-linear")
    col = col * (rast[..., 3:] > 0)
    out = dr.antialias(col, rast, pos, t["tri"])
    assert torch.isfinite(out).all() and float(out.min()) >= -1e-5 and float(out.max()) <= 1.0 + 1e-5
    cov = rast[..., 3] > 0
    # antialiasing moves colour across silhouettes only: interior pixels far from any id change keep theirs
    same = (rast[:, 1:-1, 1:-1, 3] == rast[:, :-2, 1:-1, 3]) & (rast[:, 1:-1, 1:-1, 3] == rast[:, 2:, 1:-1, 3]) \
        & (rast[:, 1:-1, 1:-1, 3] == rast[:, 1:-1, :-2, 3]) & (rast[:, 1:-1, 1:-1, 3] == rast[:, 1:-1, 2:, 3])
    assert torch.equal(out[:, 1:-1, 1:-1][same], col[:, 1:-1, 1:-1][same])
    G = torch.randn_like(out)
    g_tex, g_pos = torch.autograd.grad(out, (tex, pos), G, retain_graph=True)
    assert torch.isfinite(g_tex).all() and torch.isfinite(g_pos).all()
    # texture weights (bilinear x trilinear x mip pull) sum to one per covered pixel and the antialias
    # blend only moves weight between pixels: the texture gradient sums to the upstream gradient that
    # reaches covered pixels through `col`
    (g_col,) = torch.autograd.grad(out, col, G)
    expect = (g_col * cov[..., None]).double().sum((0, 1, 2))
    assert torch.allclose(g_tex.double().sum((0, 1, 2)), expect, rtol=2e-4, atol=5e-2)


@pytest.mark.parametrize("name", ["t1m", "t1m_shuffled", "s10k_1024"])
def test_million_triangle_scenes_equal_the_reference_fixture(dr, name):
    """tests/golden/t1m_reference.npz: what the REFERENCE's own rasterizer (oracle/_ref; tests/golden/make_t1m_fixture.py) makes of
    item 0 of bench.py's million-triangle scenes (per-bin triangle lists, SHADE launches) and of an S10k stress item at 1024^2.
    The HIP path must give the same id image (SHA-256) and the same sampled barycentrics."""
    import importlib.util
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_t1m_fixture", os.path.join(here, "golden", "make_t1m_fixture.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    fx = np.load(os.path.join(here, "golden", "t1m_reference.npz"))
    (pos, tri, res), = [(p, t, r) for n, p, t, r in mk.scenes() if n == name]
    ctx = dr.RasterizeCudaContext()
    rast, _ = dr.rasterize(ctx, torch.from_numpy(pos).cuda(), torch.from_numpy(tri).cuda(), res)
    r = rast.cpu().numpy()
    assert mk.digest(r) == bytes(fx[name + "/ids_sha256"]).decode(), "id image differs from the reference's"
    yx, want = fx[name + "/sample_yx"], fx[name + "/sample_rast"]
    got = r[0, yx[:, 0], yx[:, 1]]
    assert (got[:, 3] != want[:, 3]).sum() == 0 and np.abs(got[:, :3] - want[:, :3]).max() <= 1e-5
