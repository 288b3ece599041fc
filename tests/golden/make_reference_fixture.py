#!/usr/bin/env python3
"""Generates tests/golden/reference_pipeline.npz FROM THE REFERENCE ITSELF: the reference's own Python layer
(/root/reference/nvdiffrast/torch/ops.py, unmodified) running on the reference's own C++/CUDA sources compiled for
the host (oracle/_ref, see oracle/refshim/), autograd included.  Needs the reference checkout, so it runs in the
build container; the vectors are committed so that the GPU box -- where /root/reference does not exist -- can check
the HIP path against them (tests/test_gpu_end_to_end.py::test_pipeline_matches_reference_fixture), and the CPU suite
checks the oracle against them (tests/test_reference_fixture.py).

Scenes (all small): the four-op chain of BASELINE config 3 (rasterize -> interpolate(diff_attrs='all') -> trilinear
texture -> antialias, loss = sum(out * G), backward to pos / uv attributes / texture); the headline chain
(rasterize -> interpolate, backward); a cube-map lookup with two texture slices; three depth-peeling layers; a
range-mode render.

    python tests/golden/make_reference_fixture.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nvdiffrast_amd.utils import m10k_batch, stress_triangles     # noqa: E402

RES = (40, 56)


def inputs():
    b = m10k_batch(2, seed=77, nx=14, ny=7, attrs=3)
    rng = np.random.default_rng(78)
    s = stress_triangles(1, T=150, res=48, seed=79)
    v = rng.normal(size=(2, 12, 12, 3)).astype(np.float32)
    return dict(pos=b["pos"], tri=b["tri"], uv=b["uv"], attr=b["attr"],
                tex=rng.uniform(size=(1, 32, 32, 3)).astype(np.float32),
                g_out=rng.normal(size=(2,) + RES + (3,)).astype(np.float32),
                g_attr_out=rng.normal(size=(2,) + RES + (3,)).astype(np.float32),
                cube_tex=rng.uniform(size=(2, 6, 8, 8, 3)).astype(np.float32),
                cube_dir=(np.sign(v) * rng.uniform(0.6, 1.0, size=v.shape)).astype(np.float32),
                cube_da=(rng.normal(size=(2, 12, 12, 6)) * 0.05).astype(np.float32),
                g_cube=rng.normal(size=(2, 12, 12, 3)).astype(np.float32),
                peel_pos=s["pos"], peel_tri=s["tri"],
                ranges=np.array([[0, b["tri"].shape[0]], [20, 90]], np.int32))


def run(dr, i, dev="cpu"):
    """The scenes through an nvdiffrast.torch-compatible module `dr`; returns {name: ndarray}."""
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)      # noqa: E731
    N = lambda t: t.detach().cpu().numpy()                               # noqa: E731
    o = {}
    ctx = dr.RasterizeCudaContext()
    tri = T(i["tri"])
    # config-3 chain
    pos = T(i["pos"]).requires_grad_(True)
    uvattr = T(i["uv"]).requires_grad_(True)
    tex = T(i["tex"]).requires_grad_(True)
    rast, rast_db = dr.rasterize(ctx, pos, tri, RES)
    uv, uv_da = dr.interpolate(uvattr, rast, tri, rast_db=rast_db, diff_attrs="all")
    col = dr.texture(tex, uv, uv_da, filter_mode="linear-mipmap-linear")
    out = dr.antialias(col, rast, pos, tri)
    (out * T(i["g_out"])).sum().backward()
    o.update(rast=N(rast), rast_db=N(rast_db), uv=N(uv), uv_da=N(uv_da), col=N(col), out=N(out),
             g_pos=N(pos.grad), g_uvattr=N(uvattr.grad), g_tex=N(tex.grad))
    # headline chain
    pos2 = T(i["pos"]).requires_grad_(True)
    attr = T(i["attr"]).requires_grad_(True)
    r2, _ = dr.rasterize(ctx, pos2, tri, RES)
    a2, _ = dr.interpolate(attr, r2, tri)
    (a2 * T(i["g_attr_out"])).sum().backward()
    o.update(h_out=N(a2), h_g_pos=N(pos2.grad), h_g_attr=N(attr.grad))
    # cube map, two texture slices, corner-heavy directions
    ctex = T(i["cube_tex"]).requires_grad_(True)
    cdir = T(i["cube_dir"]).requires_grad_(True)
    cda = T(i["cube_da"]).requires_grad_(True)
    c = dr.texture(ctex, cdir, cda, filter_mode="linear-mipmap-linear", boundary_mode="cube")
    (c * T(i["g_cube"])).sum().backward()
    o.update(cube_out=N(c), cube_g_tex=N(ctex.grad), cube_g_dir=N(cdir.grad), cube_g_da=N(cda.grad))
    # depth peeling
    with dr.DepthPeeler(ctx, T(i["peel_pos"]), T(i["peel_tri"]), (48, 48)) as peeler:
        for k in range(3):
            r, rdb = peeler.rasterize_next_layer()
            o["peel%d" % k] = N(r)
    # range mode
    r, rdb = dr.rasterize(ctx, T(i["pos"][0]), tri, RES, ranges=torch.from_numpy(i["ranges"]))
    o.update(range_rast=N(r), range_rast_db=N(rdb))
    return o


if __name__ == "__main__":
    from oracle import ref, ref_torch
    ref.build()
    dr = ref_torch.reference_on_cpu("fma")
    i = inputs()
    o = run(dr, i)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_pipeline.npz")
    np.savez_compressed(path, **{"in_" + k: v for k, v in i.items()}, **{"out_" + k: v for k, v in o.items()})
    print(path, os.path.getsize(path), "bytes; coverage %.2f; reference root %s" %
          (float((o["rast"][..., 3] > 0).mean()), ref.lib().nvdr_ref_reference_root().decode()))
