#!/usr/bin/env python3
"""Environment-map learning through cube-map texturing (in the spirit of the reference's
samples/torch/envphong.py): a mirror sphere is rendered under random views; the per-pixel reflection
vector looks up a cube map (`texture(..., boundary_mode='cube')`), and the cube map is learned from
renders made with the true one.  Exercises the cube-map forward and backward kernels (edge folds, corner
texels) inside an optimisation loop.

    python samples/fit_envmap_synth.py [--iters 300] [--res 128] [--env 16]
Prints one JSON line: image loss and cube-map RMSE before and after.
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
from fit_texture_synth import uv_sphere                                             # noqa: E402


def true_envmap(S):
    """Smooth directional pattern on the cube [6,S,S,3] (a function of the texel's direction)."""
    t = (np.arange(S) + 0.5) / S * 2 - 1
    tt, ss = np.meshgrid(t, t, indexing="ij")
    faces = [(1 + 0 * ss, -tt, -ss), (-1 + 0 * ss, -tt, ss), (ss, 1 + 0 * ss, tt), (ss, -1 + 0 * ss, -tt), (ss, -tt, 1 + 0 * ss), (-ss, -tt, -1 + 0 * ss)]
    env = np.zeros((6, S, S, 3), np.float32)
    for f, (x, y, z) in enumerate(faces):
        d = np.stack([x, y, z], -1); d /= np.linalg.norm(d, axis=-1, keepdims=True)
        env[f, ..., 0] = 0.5 + 0.5 * np.sin(3 * d[..., 0] + 2 * d[..., 1])
        env[f, ..., 1] = 0.5 + 0.5 * d[..., 1]
        env[f, ..., 2] = 0.5 + 0.5 * np.cos(4 * d[..., 2] - d[..., 0])
    return env


def render(ctx, mvp, eye, pos, tri, env, res):
    posw = torch.cat([pos, torch.ones_like(pos[:, :1])], 1)
    clip = torch.matmul(posw, mvp.t())[None]
    rast, rast_db = dr.rasterize(ctx, clip, tri, (res, res))
    # attributes: world-space position = normal on the unit sphere; pixel derivatives drive the mip level
    nrm, nrm_da = dr.interpolate(pos[None], rast, tri, rast_db=rast_db, diff_attrs="all")
    view = nrm - eye                                                    # from the eye to the surface point
    view = view / view.norm(dim=-1, keepdim=True).clamp_min(1e-8)
    n = nrm / nrm.norm(dim=-1, keepdim=True).clamp_min(1e-8)
    refl = (view - 2.0 * (view * n).sum(-1, keepdim=True) * n).contiguous()
    col = dr.texture(env[None], refl, filter_mode="linear", boundary_mode="cube")
    return col * torch.clamp(rast[..., -1:], 0, 1)


def fit(iters=300, res=128, env_size=16, seed=0, lr=3e-2, device="cuda"):
    dev = torch.device(device)
    pos_np, _, tri_np = uv_sphere(32, 64)
    pos = torch.from_numpy(pos_np).to(dev)
    tri = torch.from_numpy(tri_np).to(dev)
    env_true = torch.from_numpy(true_envmap(env_size)).to(dev)
    env = torch.full_like(env_true, 0.5).requires_grad_(True)
    ctx = dr.RasterizeCudaContext(device=dev)
    opt = torch.optim.Adam([env], lr=lr)
    proj = perspective(x=0.4, n=1.0, f=20.0) @ translation(0, 0, -3.0)
    rng = np.random.default_rng(seed)
    rmse = lambda: float(torch.sqrt(torch.mean((env.detach() - env_true) ** 2)))
    r0 = rmse()
    losses = []
    t0 = time.perf_counter()
    for it in range(iters):
        pose = random_pose(rng, 0.0)
        mvp = torch.from_numpy((proj @ pose).astype(np.float32)).to(dev)
        eye = torch.from_numpy((np.linalg.inv(translation(0, 0, -3.0) @ pose) @ np.array([0, 0, 0, 1.0]))[:3].astype(np.float32)).to(dev)
        with torch.no_grad():
            target = render(ctx, mvp, eye, pos, tri, env_true, res)
        img = render(ctx, mvp, eye, pos, tri, env, res)
        loss = torch.mean((img - target) ** 2)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    return dict(iters=iters, res=res, env=env_size, loss_first=float(np.mean(losses[:5])), loss_last=float(np.mean(losses[-5:])),
                env_rmse_before=r0, env_rmse_after=rmse(), iters_per_s=round(iters / (time.perf_counter() - t0), 1))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--env", type=int, default=16)
    a = ap.parse_args()
    print(json.dumps(fit(a.iters, a.res, a.env)))
