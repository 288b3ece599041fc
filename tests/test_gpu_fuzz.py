"""GPU side of the randomised pinning (tests/test_ref_fuzz.py): the same seeded random scenes through the HIP path,
compared with the pinned oracle (which cross-checks itself against the reference on the spot).  Triangle ids AND the
U32 depth surface (read from the DepthPeeler's layer-0 surface) must be identical; floats meet the usual bars wherever
the scene is well conditioned."""
import numpy as np
import pytest
from conftest import within
import torch

from test_ref_fuzz import _random_scene

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("seed", range(24))
def test_random_scenes_ids_and_depth(dr, oracle, seed):
    rng = np.random.default_rng(9000 + seed)
    pos, tri, res = _random_scene(rng)
    ids_o, depth_o = oracle.rasterize_ids(pos, tri, res)
    ctx = dr.RasterizeCudaContext()
    with dr.DepthPeeler(ctx, _t(pos), _t(tri), res) as peeler:
        r, rdb = peeler.rasterize_next_layer()
    depth = ctx.cpp_wrapper.depth.cpu().numpy().view(np.uint32)
    H, W = res
    got_ids = r[..., 3].cpu().numpy()
    assert (got_ids != ids_o[:, :H, :W].astype(np.float32)).sum() == 0, "triangle ids differ"
    cov = ids_o[:, :H, :W] > 0
    assert (depth[:, :H, :W][cov] != depth_o[:, :H, :W][cov]).sum() == 0, "U32 depth surface differs"
    ro, rdbo = oracle._o.rasterize(pos, tri, res)
    ok = np.isfinite(ro).all(-1)
    within("fuzz rast u,v,z/w", r.cpu().numpy()[ok][:, :3], ro[ok][:, :3], 1e-5)


@pytest.mark.parametrize("seed", range(24))
def test_random_texture(dr, oracle, seed):
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
        uv[rng.uniform(size=(N, H, W)) < 0.05] = 0.0
        uv_da = (rng.normal(size=(N, H, W, 6)) * rng.choice([0.0, 0.02, 0.5])).astype(np.float32)
        bm = "cube"
    else:
        th, tw = int(2 ** rng.integers(0, 7)), int(2 ** rng.integers(0, 7))
        tex = rng.uniform(size=(tn, th, tw, C)).astype(np.float32)
        uv = rng.uniform(-1.5, 2.5, size=(N, H, W, 2)).astype(np.float32)
        uv_da = (rng.normal(size=(N, H, W, 4)) * rng.choice([0.0, 0.02, 0.5])).astype(np.float32)
        bm = str(rng.choice(["wrap", "clamp", "zero"]))
    mip = "mipmap" in fm
    mode = rng.integers(0, 3) if mip else 0
    bias = rng.uniform(-1, 3, size=(N, H, W)).astype(np.float32)
    kw = dict(filter_mode=fm, boundary_mode=bm)
    okw = dict(kw)
    da_in = bias_in = None
    if mip:
        da_in = None if mode == 1 else uv_da
        bias_in = None if mode == 0 else bias
        okw.update(uv_da=da_in, mip_level_bias=bias_in)
        if rng.uniform() < 0.3:
            kw["max_mip_level"] = okw["max_mip_level"] = int(rng.integers(0, 4))
    dy = rng.normal(size=(N, H, W, C)).astype(np.float32)
    dy[rng.uniform(size=(N, H, W)) < 0.2] = 0.0
    t_tex = _t(tex).requires_grad_(True)
    t_uv = _t(uv).requires_grad_(True)
    out = dr.texture(t_tex, t_uv, None if da_in is None else _t(da_in), None if bias_in is None else _t(bias_in), **kw)
    out.backward(_t(dy))
    oo = oracle.texture(tex, uv, **okw)
    g = oracle.texture_grad(tex, uv, dy, **okw)
    tol = lambda x: 1e-5 * max(1.0, float(np.abs(x).max()))                  # noqa: E731
    within("fuzz texture out", out.detach().cpu().numpy(), oo, 1e-5)                # (no element is exempted)
    within("fuzz texture g_tex", t_tex.grad.cpu().numpy(), g["tex"], tol(g["tex"]))
    if g["uv"] is not None:
        within("fuzz texture g_uv", t_uv.grad.cpu().numpy(), g["uv"], tol(g["uv"]))
