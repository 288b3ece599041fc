"""BASELINE config 5 stand-in: the four ops inside an optimisation loop (samples/fit_texture_synth.py).
Numeric parity of every op is covered elsewhere; this catches gradient-quality regressions that
per-op parity would miss (wrong sign, a gradient routed to the wrong tensor, a stale work buffer)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name="fit_texture_synth"):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "samples", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("graph", [False, True])
def test_texture_fit_converges(dr, graph):
    """graph=True: the whole iteration (two renders, backward, Adam) is captured into one hipGraph and
    replayed -- possible because every op is asynchronous, allocation-free at the C ABI and never
    synchronises the host (the reference's rasterizer does, RasterImpl.cpp:367)."""
    r = _load().fit(iters=60, res=128, ref_res=256, tex_size=128, seed=1, lr=3e-2, graph=graph)
    assert r["loss_last"] < 0.25 * r["loss_first"], r
    assert r["tex_rmse_after"] < 0.8 * r["tex_rmse_before"], r


def test_graph_replay_matches_eager(dr):
    import numpy as np
    import torch
    from nvdiffrast_amd.utils import m10k_batch
    b = m10k_batch(2, seed=5, nx=24, ny=12)
    dev = torch.device("cuda")
    pos = torch.from_numpy(b["pos"]).to(dev).requires_grad_(True)
    attr = torch.from_numpy(b["attr"]).to(dev).requires_grad_(True)
    tri = torch.from_numpy(b["tri"]).to(dev)
    G = torch.randn(2, 96, 96, 4, device=dev)
    ctx = dr.RasterizeCudaContext()

    def step():
        pos.grad = None; attr.grad = None
        rast, _ = dr.rasterize(ctx, pos, tri, (96, 96))
        out, _ = dr.interpolate(attr, rast, tri)
        torch.autograd.backward(out, G)
        return out

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    eager = step().detach().clone(); g_pos = pos.grad.clone(); g_attr = attr.grad.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out_g = step()
    pos.grad.zero_(); attr.grad.zero_()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_g, eager)
    assert torch.allclose(pos.grad, g_pos, rtol=1e-5, atol=1e-5 * float(g_pos.abs().max()))
    assert torch.allclose(attr.grad, g_attr, rtol=1e-5, atol=1e-5 * float(g_attr.abs().max()))


def test_cube_geometry_is_recovered_through_silhouette_gradients(dr):
    """Vertex positions get gradients only through antialias (and through rasterize's barycentrics):
    recovering a perturbed cube from 32x32 renders needs both to be right (cf. samples/torch/cube.py)."""
    # The loop is not bit-reproducible (f32 atomics) and Adam amplifies that: over runs of ONE seed the vertex error after
    # 300 iterations was observed between 1e-5 and 4e-4, with one run in ~15 still far from converged (0.08; 32x32-pixel
    # renders of an axis-aligned cube keep hitting the knife-edge cases of DESIGN.md section 2).  The test therefore
    # asks for convergence in one of two attempts; the reference's own cube.py on this package is
    # tests/test_gpu_reference_samples.py::test_cube_py.
    mod = _load("fit_cube_synth")
    for seed, iters in ((2, 300), (3, 500)):
        r = mod.fit(iters=iters, res=32, batch=8, seed=seed)
        assert r["pos_err_before"] > 0.15
        if r["pos_err_after"] < 5e-3 and r["col_err_after"] < 5e-3 and r["loss_last"] < 1e-3 * r["loss_first"]:
            return
    raise AssertionError(r)


def test_cube_map_is_learned_from_reflections(dr):
    """Every texel of all six faces -- edge and corner texels included -- must receive consistent
    gradients for the environment map to be recovered (cf. samples/torch/envphong.py)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "samples"))
    r = _load("fit_envmap_synth").fit(iters=200, res=96, env_size=8, seed=3)
    assert r["env_rmse_before"] > 0.2 and r["env_rmse_after"] < 2e-3, r


def test_pose_is_recovered_from_a_single_image(dr):
    """Batched candidate search + gradient descent on the rotation of a face-coloured cube
    (cf. samples/torch/pose.py).  interpolate runs with its own index buffer (per-face colours) while
    rasterize / antialias use the position topology; the pose gradient exists only through antialias."""
    r = _load("fit_pose_synth").fit(res=64, search=14, candidates=32, descent=300, seed=1)       # short search: descent has work left
    assert r["err_deg_initial"] > 20, r
    assert r["err_deg_after_search"] < 25, r
    assert r["err_deg_final"] < 0.2 and r["err_deg_final"] <= r["err_deg_after_search"], r


def test_pipeline_matches_committed_fixture(dr):
    """The whole op chain (rasterize -> interpolate with differentials -> trilinear texture -> antialias, forward
    and backward through autograd) against the committed vectors of tests/golden/pipeline_small.npz."""
    fx = np.load(os.path.join(ROOT, "tests", "golden", "pipeline_small.npz"))
    dev = torch.device("cuda", 0)
    t = lambda k: torch.from_numpy(fx["in_" + k]).to(dev)
    pos = t("pos").requires_grad_(True); uvattr = t("uv").requires_grad_(True); tex = t("tex").requires_grad_(True)
    tri = t("tri")
    H, W = fx["out_rast"].shape[1:3]
    ctx = dr.RasterizeCudaContext(device=dev)
    rast, rast_db = dr.rasterize(ctx, pos, tri, (H, W))
    uv, uv_da = dr.interpolate(uvattr, rast, tri, rast_db=rast_db, diff_attrs="all")
    col = dr.texture(tex, uv, uv_da, filter_mode="linear-mipmap-linear")
    out = dr.antialias(col, rast, pos, tri)
    torch.autograd.backward(out, t("g_out"))
    got = dict(rast=rast, rast_db=rast_db, uv=uv, uv_da=uv_da, col=col, out=out,
               g_tex=tex.grad, g_uvattr=uvattr.grad, g_pos=pos.grad)
    assert np.array_equal(rast[..., 3].detach().cpu().numpy(), fx["out_rast"][..., 3])          # ids: bit exact
    for k, v in got.items():
        ref = fx["out_" + k]
        err = np.abs(v.detach().cpu().numpy() - ref).max()
        tol = 1e-5 * max(1.0, float(np.abs(ref).max()))          # measured: <= 0.26 of this for the colours, <= 0.1 for the gradients
        assert err <= tol, (k, err, tol)
