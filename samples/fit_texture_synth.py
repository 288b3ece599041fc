#!/usr/bin/env python3
"""End-to-end optimisation loop over all four ops (BASELINE config 5; stands in for the reference's
samples/torch/earth.py, whose data file earth.npz is not in the reference checkout).

A UV sphere carries a procedural ground-truth texture.  Each iteration renders it from a random
view twice -- at `ref_res` with the true texture (box-filtered down to `res`) and at `res` with the
texture being learned -- and takes an Adam step on the L2 image difference.  The render is
    rasterize -> interpolate(uv, diff_attrs='all') -> texture(linear-mipmap-linear) -> antialias
so every forward and backward kernel of the path runs every iteration.

    python samples/fit_texture_synth.py [--iters 200] [--res 256] [--ref-res 1024] [--tex 512] [--graph]
Prints one JSON line: first/last loss, texture RMSE before/after, iterations per second.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvdiffrast_amd.torch as dr                                     # noqa: E402
from nvdiffrast_amd.utils.synth import perspective, translation, random_pose   # noqa: E402


def uv_sphere(n_lat=48, n_lon=96):
    """Unit sphere with a seam: (n_lat+1) x (n_lon+1) vertices, uv = (lon, lat) in [0,1]."""
    lat = np.linspace(0.0, np.pi, n_lat + 1)
    lon = np.linspace(0.0, 2.0 * np.pi, n_lon + 1)
    la, lo = np.meshgrid(lat, lon, indexing="ij")
    pos = np.stack([np.sin(la) * np.cos(lo), np.cos(la), np.sin(la) * np.sin(lo)], -1).reshape(-1, 3)
    uv = np.stack([lo / (2.0 * np.pi), la / np.pi], -1).reshape(-1, 2)
    row = n_lon + 1
    i, j = np.meshgrid(np.arange(n_lat), np.arange(n_lon), indexing="ij")
    v00 = (i * row + j).reshape(-1)
    tri = np.concatenate([np.stack([v00, v00 + row, v00 + row + 1], -1), np.stack([v00, v00 + row + 1, v00 + 1], -1)], 0)
    return pos.astype(np.float32), uv.astype(np.float32), tri.astype(np.int32)


def procedural_texture(size):
    """Smooth continents-and-stripes pattern, [size, size, 3] in [0,1], periodic in u."""
    v, u = np.meshgrid(np.linspace(0, 1, size, endpoint=False), np.linspace(0, 1, size, endpoint=False), indexing="ij")
    r = 0.5 + 0.5 * np.sin(2 * np.pi * (3 * u + 0.3 * np.sin(2 * np.pi * 2 * v)))
    g = 0.5 + 0.5 * np.sin(2 * np.pi * 5 * v) * np.cos(2 * np.pi * 2 * u)
    b = 0.5 + 0.5 * np.cos(2 * np.pi * (4 * u - 3 * v))
    checker = ((np.floor(u * 16) + np.floor(v * 16)) % 2) * 0.15
    return np.clip(np.stack([r, g, b], -1) * 0.85 + checker[..., None], 0, 1).astype(np.float32)


def render(ctx, mvp, pos, tri, uv, tex, res, topo):
    posw = torch.cat([pos, torch.ones_like(pos[:, :1])], 1)
    clip = torch.matmul(posw, mvp.t())[None]
    rast, rast_db = dr.rasterize(ctx, clip, tri, (res, res))
    texc, texd = dr.interpolate(uv[None], rast, tri, rast_db=rast_db, diff_attrs="all")
    color = dr.texture(tex[None], texc, texd, filter_mode="linear-mipmap-linear", max_mip_level=6)
    color = color * torch.clamp(rast[..., -1:], 0, 1)                 # mask the background
    return dr.antialias(color, rast, clip, tri, topology_hash=topo)


def fit(iters=200, res=256, ref_res=1024, tex_size=512, seed=0, lr=1e-2, device="cuda", graph=False):
    dev = torch.device(device)
    pos_np, uv_np, tri_np = uv_sphere()
    pos = torch.from_numpy(pos_np).to(dev)
    uv = torch.from_numpy(uv_np).to(dev)
    tri = torch.from_numpy(tri_np).to(dev)
    tex_true = torch.from_numpy(procedural_texture(tex_size)).to(dev)
    tex_opt = torch.full_like(tex_true, 0.5).requires_grad_(True)
    ctx = dr.RasterizeCudaContext(device=dev)
    topo = dr.antialias_construct_topology_hash(tri)
    opt = torch.optim.Adam([tex_opt], lr=lr)
    proj = perspective(x=0.4, n=1.0, f=20.0) @ translation(0, 0, -3.5)
    rng = np.random.default_rng(seed)
    rmse0 = float(torch.sqrt(torch.mean((tex_opt.detach() - tex_true) ** 2)))
    mvp = torch.zeros(4, 4, device=dev)
    loss_buf = torch.zeros((), device=dev)

    def one_iteration():
        with torch.no_grad():
            ref = render(ctx, mvp, pos, tri, uv, tex_true, ref_res, topo)
            k = ref_res // res
            ref = ref.reshape(1, res, k, res, k, 3).mean((2, 4))
        img = render(ctx, mvp, pos, tri, uv, tex_opt, res, topo)
        loss = torch.mean((img - ref) ** 2)
        opt.zero_grad(set_to_none=False)
        loss.backward()
        opt.step()
        loss_buf.copy_(loss.detach())

    def new_view():
        mvp.copy_(torch.from_numpy((proj @ random_pose(rng, 0.0)).astype(np.float32)), non_blocking=False)

    g = None
    if graph:
        # Every op of the path is asynchronous and allocation-free at the C-ABI level, so a whole
        # iteration (two renders, backward, Adam) captures into one hipGraph; only the view matrix
        # changes between replays.  Launch-bound at these sizes: see the printed it/s with and without.
        opt = torch.optim.Adam([tex_opt], lr=lr, capturable=True)
        new_view()
        from nvdiffrast_amd.torch.graph import StepGraph
        g = StepGraph(one_iteration, warmup=3)                    # warm-up (allocations, scratch growth) on a side stream, then the recording
    losses = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(iters):
        new_view()
        if g is not None:
            g.replay()
        else:
            one_iteration()
        if it < 5 or it >= iters - 5:
            losses.append(float(loss_buf))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rmse1 = float(torch.sqrt(torch.mean((tex_opt.detach() - tex_true) ** 2)))
    return dict(iters=iters, res=res, ref_res=ref_res, tex=tex_size, graph=bool(graph), loss_first=float(np.mean(losses[:5])),
                loss_last=float(np.mean(losses[-5:])), tex_rmse_before=rmse0, tex_rmse_after=rmse1,
                iters_per_s=round(iters / dt, 1))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--ref-res", type=int, default=1024)
    ap.add_argument("--tex", type=int, default=512)
    ap.add_argument("--graph", action="store_true", help="capture one iteration into a hipGraph and replay it")
    a = ap.parse_args()
    print(json.dumps(fit(a.iters, a.res, a.ref_res, a.tex, graph=a.graph)))
