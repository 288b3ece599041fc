"""Deterministic synthetic workloads: no datasets, no checkpoints.

``m10k_batch`` is the benchmark's geometry: a jittered 101x51 vertex lattice (V = 5151,
T = 10000, shared vertices) seen through a per-item random rigid pose and a perspective
projection.  ``stress_triangles`` is the high-overdraw variant (independent triangles).
Everything is numpy on the host; callers move the arrays to the GPU.
"""
import numpy as np


def perspective(x=0.1, n=1.0, f=50.0):
    """OpenGL-style frustum with half-width ``x`` at the near plane (cf. samples/torch/util.py:16-20)."""
    m = np.zeros((4, 4), np.float32)
    m[0, 0] = m[1, 1] = n / x
    m[2, 2] = -(f + n) / (f - n)
    m[2, 3] = -(2.0 * f * n) / (f - n)
    m[3, 2] = -1.0
    return m


def translation(x, y, z):
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = (x, y, z)
    return m


def random_pose(rng, t):
    """Random orthonormal frame + translation in [-t, t]^3 (recipe of samples/torch/util.py:42-50,
    drawn from ``rng`` instead of the global numpy state)."""
    m = rng.normal(size=(3, 3))
    m[1] = np.cross(m[0], m[2])
    m[2] = np.cross(m[0], m[1])
    m /= np.linalg.norm(m, axis=1, keepdims=True)
    out = np.eye(4)
    out[:3, :3] = m
    out[:3, 3] = rng.uniform(-t, t, size=3)
    return out


def lattice_mesh(rng, nx=100, ny=50, extent=0.9, jitter=0.4):
    """(nx+1) x (ny+1) vertex lattice over [-extent, extent]^2 with two triangles per cell.

    Returns (verts [V,3] f32, tri [T,3] i32, uv [V,2] f32 lattice coordinates in [0,1])."""
    xs = np.linspace(-extent, extent, nx + 1)
    ys = np.linspace(-extent, extent, ny + 1)
    gx, gy = np.meshgrid(xs, ys, indexing="xy")             # [ny+1, nx+1]
    cell = np.array([2 * extent / nx, 2 * extent / ny])
    jx = rng.uniform(-jitter, jitter, size=gx.shape) * cell[0]
    jy = rng.uniform(-jitter, jitter, size=gy.shape) * cell[1]
    x = gx + jx
    y = gy + jy
    z = 0.3 * np.sin(3.0 * x) * np.cos(2.0 * y) + rng.uniform(-0.05, 0.05, size=x.shape)
    verts = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    uv = np.stack([(gx + extent) / (2 * extent), (gy + extent) / (2 * extent)], -1).reshape(-1, 2).astype(np.float32)
    row = nx + 1
    cy, cx = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    v00 = (cy * row + cx).reshape(-1)
    v10, v01, v11 = v00 + 1, v00 + row, v00 + row + 1
    tri = np.concatenate([np.stack([v00, v10, v11], -1), np.stack([v00, v11, v01], -1)], 0)
    # interleave the two triangles of each cell so neighbouring ids are neighbours on screen
    tri = tri.reshape(2, -1, 3).transpose(1, 0, 2).reshape(-1, 3).astype(np.int32)
    return verts, tri, uv


def m10k_batch(N, seed=20240, nx=100, ny=50, attrs=4, pose_seed=None):
    """Benchmark geometry: returns dict(pos [N,V,4], tri [T,3], attr [1,V,attrs], uv [1,V,2]).
    The mesh depends on ``seed`` only; item n's pose on ``pose_seed + n`` (default: seed + n)."""
    rng = np.random.default_rng(seed)
    verts, tri, uv = lattice_mesh(rng, nx, ny)
    attr = rng.uniform(0.0, 1.0, size=(1, verts.shape[0], attrs)).astype(np.float32)
    vh = np.concatenate([verts, np.ones((verts.shape[0], 1), np.float32)], 1).astype(np.float64)
    proj = perspective(x=0.4, n=1.0, f=50.0).astype(np.float64) @ translation(0, 0, -3.5).astype(np.float64)
    pos = np.empty((N, verts.shape[0], 4), np.float32)
    for n in range(N):
        pose = random_pose(np.random.default_rng((seed if pose_seed is None else pose_seed) + n), 0.25)
        pos[n] = (vh @ (proj @ pose).T).astype(np.float32)
    return dict(pos=pos, tri=tri, attr=attr, uv=uv[None])


def small_rotation(rng, max_deg):
    """Rotation by at most ``max_deg`` degrees about a random axis (Rodrigues), as a 4x4 matrix."""
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    a = np.deg2rad(rng.uniform(-max_deg, max_deg))
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    out = np.eye(4)
    out[:3, :3] = np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)
    return out


def dense_batch(N, seed=20240, nx=100, ny=50, attrs=4, distance=1.45, max_deg=12.0):
    """The benchmark's mesh with the camera pulled in until it fills the image (coverage >= 0.9, overdraw ~1):
    the regime in which no consumer of ``rast`` can skip anything.  Same dict as ``m10k_batch``."""
    rng = np.random.default_rng(seed)
    verts, tri, uv = lattice_mesh(rng, nx, ny)
    attr = rng.uniform(0.0, 1.0, size=(1, verts.shape[0], attrs)).astype(np.float32)
    vh = np.concatenate([verts, np.ones((verts.shape[0], 1), np.float32)], 1).astype(np.float64)
    proj = perspective(x=0.4, n=1.0, f=50.0).astype(np.float64) @ translation(0, 0, -distance).astype(np.float64)
    pos = np.empty((N, verts.shape[0], 4), np.float32)
    for n in range(N):
        pose = small_rotation(np.random.default_rng(seed + 7000 + n), max_deg)
        pos[n] = (vh @ (proj @ pose).T).astype(np.float32)
    return dict(pos=pos, tri=tri, attr=attr, uv=uv[None])


def big_mesh_batch(N, nx=1000, ny=500, seed=20240, attrs=4, shuffle=False):
    """T = 2*nx*ny triangles (default one million) of the same kind of lattice, random poses as in ``m10k_batch``;
    ``shuffle`` permutes the rows of ``tri`` -- the same surface, but its index order no longer says anything about where a
    triangle lies on the screen (a triangle soup as far as the rasterizer's binning is concerned)."""
    b = m10k_batch(N, seed=seed, nx=nx, ny=ny, attrs=attrs)
    if shuffle:
        b["tri"] = np.ascontiguousarray(b["tri"][np.random.default_rng(seed + 1).permutation(b["tri"].shape[0])])
    return b


def stress_triangles(N, T=10000, res=512, seed=20240):
    """S10k: independent triangles, centres U(-1,1)^2, edge length log-uniform [2,64] px,
    z U(-0.9,0.9), w = 1.  Returns dict(pos [N,3T,4], tri [T,3])."""
    rng = np.random.default_rng(seed)
    pos = np.empty((N, 3 * T, 4), np.float32)
    for n in range(N):
        c = rng.uniform(-1, 1, size=(T, 1, 2))
        size = np.exp(rng.uniform(np.log(2.0), np.log(64.0), size=(T, 1, 1))) * (2.0 / res)
        ang = rng.uniform(0, 2 * np.pi, size=(T, 1, 1)) + np.array([0, 2.1, 4.2]).reshape(1, 3, 1) \
            + rng.uniform(-0.5, 0.5, size=(T, 3, 1))
        xy = c + size * np.concatenate([np.cos(ang), np.sin(ang)], -1)
        z = rng.uniform(-0.9, 0.9, size=(T, 3, 1))
        pos[n] = np.concatenate([xy, z, np.ones_like(z)], -1).reshape(-1, 4)
    tri = np.arange(3 * T, dtype=np.int32).reshape(T, 3)
    return dict(pos=pos, tri=tri)
