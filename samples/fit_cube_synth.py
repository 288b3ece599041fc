#!/usr/bin/env python3
"""Geometry fitting through silhouettes (in the spirit of the reference's samples/torch/cube.py):
vertex positions and vertex colours of a cube are recovered from low-resolution renders of the
true cube under random rotations.  Position gradients exist only because `antialias` makes pixel
colours depend continuously on where silhouette edges fall, so this is the end-to-end check of the
rasterize -> interpolate -> antialias gradient chain.

    python samples/fit_cube_synth.py [--iters 400] [--res 32] [--batch 8]
Prints one JSON line: image loss and geometric / colour error before and after.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvdiffrast_amd.torch as dr                                                  # noqa: E402
from nvdiffrast_amd.utils.synth import perspective, translation, random_pose        # noqa: E402


def cube_mesh():
    v = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32) * 0.5
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    tri = np.array([t for q in quads for t in ((q[0], q[1], q[2]), (q[0], q[2], q[3]))], np.int32)
    col = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], np.float32)
    return v, tri, col


def render(ctx, mvp, pos, tri, col, res, topo):
    """mvp [N,4,4], pos [V,3], col [V,3] -> [N,res,res,3]"""
    posw = torch.cat([pos, torch.ones_like(pos[:, :1])], 1)
    clip = torch.matmul(posw[None], mvp.transpose(1, 2)).contiguous()
    rast, _ = dr.rasterize(ctx, clip, tri, (res, res))
    img, _ = dr.interpolate(col[None], rast, tri)
    return dr.antialias(img, rast, clip, tri, topology_hash=topo)


def fit(iters=400, res=32, batch=8, seed=0, lr=2e-2, device="cuda"):
    dev = torch.device(device)
    v_np, tri_np, c_np = cube_mesh()
    rng = np.random.default_rng(seed)
    tri = torch.from_numpy(tri_np).to(dev)
    pos_true = torch.from_numpy(v_np).to(dev)
    col_true = torch.from_numpy(c_np).to(dev)
    pos = (pos_true + torch.from_numpy(rng.normal(scale=0.12, size=v_np.shape).astype(np.float32)).to(dev)).requires_grad_(True)
    col = torch.full_like(col_true, 0.5).requires_grad_(True)
    ctx = dr.RasterizeCudaContext(device=dev)
    topo = dr.antialias_construct_topology_hash(tri)
    opt = torch.optim.Adam([pos, col], lr=lr)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda i: max(0.05, 10 ** (-i / iters)))
    proj = perspective(x=0.4, n=1.0, f=20.0) @ translation(0, 0, -3.0)
    err = lambda: (float((pos.detach() - pos_true).abs().max()), float((col.detach() - col_true).abs().max()))
    e0 = err()
    losses = []
    t0 = time.perf_counter()
    for it in range(iters):
        mvp = torch.from_numpy(np.stack([(proj @ random_pose(rng, 0.0)).astype(np.float32) for _ in range(batch)])).to(dev)
        with torch.no_grad():
            target = render(ctx, mvp, pos_true, tri, col_true, res, topo)
        img = render(ctx, mvp, pos, tri, col, res, topo)
        loss = torch.mean((img - target) ** 2)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step(); sched.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    e1 = err()
    return dict(iters=iters, res=res, batch=batch, loss_first=float(np.mean(losses[:5])), loss_last=float(np.mean(losses[-5:])),
                pos_err_before=e0[0], pos_err_after=e1[0], col_err_before=e0[1], col_err_after=e1[1],
                iters_per_s=round(iters / (time.perf_counter() - t0), 1))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--res", type=int, default=32)
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    print(json.dumps(fit(a.iters, a.res, a.batch)))
