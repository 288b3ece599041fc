#!/usr/bin/env python3
"""Pose recovery from one image (in the spirit of the reference's samples/torch/pose.py): the unknown
rotation of a cube with six differently coloured faces is found from a single target render.

Two phases, both through rasterize -> interpolate -> antialias:
  1. search   K candidate rotations are rendered in ONE instanced rasterize call per iteration
              (the batch axis is the candidate axis): the incumbent composed with random rotations of
              shrinking angle, half of them additionally composed with a random element of the cube's
              24-element rotation group (the image loss has local minima at face-permuted poses);
  2. descent  Adam on an axis-angle correction of the best candidate; the gradient reaches the pose
              only through `antialias` (silhouette position) and the interpolated colours.
Colours are per face, so `interpolate` is called with its own index buffer (a colour index per
triangle corner) while rasterize / antialias use the 8-vertex position topology -- the same split
the reference sample uses.

    python samples/fit_pose_synth.py [--res 64] [--search 40] [--descent 200]
Prints one JSON line with the angular error (degrees) after each phase.
"""
import argparse
import itertools
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvdiffrast_amd.torch as dr                                                  # noqa: E402
from nvdiffrast_amd.utils.synth import perspective, translation                    # noqa: E402


def cube():
    pos = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32) * 0.6
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    pos_idx = np.array([t for q in quads for t in ((q[0], q[1], q[2]), (q[0], q[2], q[3]))], np.int32)
    col = np.array([[1, .1, .1], [.1, 1, .1], [.1, .1, 1], [1, 1, .1], [1, .1, 1], [.1, 1, 1]], np.float32)
    col_idx = np.repeat(np.arange(6, dtype=np.int32), 2)[:, None].repeat(3, 1)       # both triangles of face f use colour f
    return pos, pos_idx, col, np.ascontiguousarray(col_idx)


def cube_group():
    """The 24 proper rotations that map the cube onto itself: signed permutation matrices with det +1."""
    out = []
    for perm in itertools.permutations(range(3)):
        for signs in itertools.product((1, -1), repeat=3):
            m = np.zeros((3, 3), np.float32)
            for r in range(3):
                m[r, perm[r]] = signs[r]
            if np.linalg.det(m) > 0:
                out.append(m)
    assert len(out) == 24
    return np.stack(out)


def rodrigues(w):
    """axis-angle [..., 3] -> rotation matrices [..., 3, 3] (differentiable, fine at w = 0)."""
    th2 = (w * w).sum(-1)[..., None, None]
    th = torch.sqrt(th2 + 1e-20)
    z = torch.zeros_like(w[..., 0])
    K = torch.stack([torch.stack([z, -w[..., 2], w[..., 1]], -1),
                     torch.stack([w[..., 2], z, -w[..., 0]], -1),
                     torch.stack([-w[..., 1], w[..., 0], z], -1)], -2)
    a = torch.sin(th) / th
    b = (1 - torch.cos(th)) / (th2 + 1e-20)
    eye = torch.eye(3, device=w.device, dtype=w.dtype).expand(K.shape)
    return eye + a * K + b * (K @ K)


def random_rotations(rng, n, max_angle):
    """n rotations with uniformly random axes and angles in [0, max_angle]."""
    axis = rng.normal(size=(n, 3))
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    return (axis * rng.uniform(0, max_angle, size=(n, 1))).astype(np.float32)


def angle_deg(Ra, Rb):
    c = (torch.trace(Ra.T @ Rb) - 1) / 2
    return float(torch.rad2deg(torch.acos(c.clamp(-1, 1))))


def render(ctx, R, vp, pos, pos_idx, col, col_idx, res, topo):
    """R [N,3,3] object rotations, vp [4,4] view-projection -> images [N,res,res,3]"""
    world = pos[None] @ R.transpose(1, 2)                                          # [N,V,3]
    clip = torch.cat([world, torch.ones_like(world[..., :1])], -1) @ vp.T
    clip = clip.contiguous()
    rast, _ = dr.rasterize(ctx, clip, pos_idx, (res, res))
    img, _ = dr.interpolate(col[None], rast, col_idx)
    return dr.antialias(img, rast, clip, pos_idx, topology_hash=topo)


def image_loss(img, target):
    d2 = ((img - target) ** 2).sum(-1)
    return (d2 / (d2 + 0.25)).mean(dim=(1, 2))                                     # saturating: far-off pixels do not dominate


def fit(res=64, search=40, candidates=32, descent=200, seed=0, lr=0.02, device="cuda"):
    dev = torch.device(device)
    rng = np.random.default_rng(seed)
    p_np, pi_np, c_np, ci_np = cube()
    pos, pos_idx = torch.from_numpy(p_np).to(dev), torch.from_numpy(pi_np).to(dev)
    col, col_idx = torch.from_numpy(c_np).to(dev), torch.from_numpy(ci_np).to(dev)
    group = torch.from_numpy(cube_group()).to(dev)
    vp = torch.from_numpy((perspective(x=0.4, n=1.0, f=20.0) @ translation(0, 0, -3.5)).astype(np.float32)).to(dev)
    ctx = dr.RasterizeCudaContext(device=dev)
    topo = dr.antialias_construct_topology_hash(pos_idx)

    # A target is only identifiable if three faces are visible (a face-on cube looks the same after
    # quarter turns about the view axis): draw until the view direction is well off every face axis.
    while True:
        R_true = rodrigues(torch.from_numpy(random_rotations(rng, 1, np.pi)).to(dev))[0]
        if float(R_true[2].abs().min()) > 0.3:                                     # row 2 = view axis in object coordinates
            break
    with torch.no_grad():
        target = render(ctx, R_true[None], vp, pos, pos_idx, col, col_idx, res, topo)
    R_best = rodrigues(torch.from_numpy(random_rotations(rng, 1, np.pi)).to(dev))[0]
    e_init = angle_deg(R_best, R_true)
    t0 = time.perf_counter()

    # ---- phase 1: batched stochastic search ---------------------------------------------
    with torch.no_grad():
        best = float(image_loss(render(ctx, R_best[None], vp, pos, pos_idx, col, col_idx, res, topo), target)[0])
        for it in range(search):
            spread = np.pi * (0.02 ** (it / max(search - 1, 1)))                   # 180 deg -> 3.6 deg
            noise = rodrigues(torch.from_numpy(random_rotations(rng, candidates, spread)).to(dev))
            sym = group[torch.from_numpy(rng.integers(0, 24, size=candidates)).to(dev)]
            sym[: candidates // 2] = torch.eye(3, device=dev)                      # half of them stay in the current basin
            cand = R_best[None] @ noise @ sym
            losses = image_loss(render(ctx, cand, vp, pos, pos_idx, col, col_idx, res, topo), target)
            k = int(torch.argmin(losses))
            if float(losses[k]) < best:
                best, R_best = float(losses[k]), cand[k].clone()
    e_search = angle_deg(R_best, R_true)

    # ---- phase 2: gradient descent on an axis-angle correction ------------------------
    w = torch.zeros(3, device=dev, requires_grad=True)
    opt = torch.optim.Adam([w], lr=lr)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda i: 0.05 ** (i / max(descent, 1)))
    last = None
    for it in range(descent):
        R = R_best @ rodrigues(w)
        loss = image_loss(render(ctx, R[None], vp, pos, pos_idx, col, col_idx, res, topo), target)[0]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step(); sched.step()
        last = float(loss.detach())
    torch.cuda.synchronize()
    R_fit = (R_best @ rodrigues(w)).detach()
    return dict(res=res, search_iters=search, candidates=candidates, descent_iters=descent,
                err_deg_initial=round(e_init, 3), err_deg_after_search=round(e_search, 3),
                err_deg_final=round(angle_deg(R_fit, R_true), 4), loss_after_search=best, loss_final=last,
                seconds=round(time.perf_counter() - t0, 2))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=64)
    ap.add_argument("--search", type=int, default=40)
    ap.add_argument("--candidates", type=int, default=32)
    ap.add_argument("--descent", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    print(json.dumps(fit(a.res, a.search, a.candidates, a.descent, a.seed)))
