#!/usr/bin/env python3
"""Time-bounded soak of the rasterizer's integer stage: random scenes (the generator of tests/test_ref_fuzz.py, widened to
several bins per image, thousands of triangles and resolutions that are no multiple of anything) through the HIP path, triangle
ids and the U32 depth surface against the C oracle, bit for bit -- the production instantiation (plain rasterize), the depth-peeling
one (layers 0 and 1), and range mode.  Prints one line per mismatch (seed, what, count) and a summary; exit code 1 on any.
    python tools/fuzz_soak.py [seconds] [first seed]
    NVDR_DEBUG=268435456 python tools/fuzz_soak.py ...      # per-bin triangle lists for every mesh (k_fine<LIST>, k_binfill)
The oracle is the checker here (this is a test tool, like tests/); nothing of the product imports it."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
import nvdiffrast_amd.torch as dr


def scene(rng):
    N = int(rng.integers(1, 5))
    H, W = int(rng.integers(3, 700)), int(rng.integers(3, 700))
    T = int(np.exp(rng.uniform(0, np.log(6000))))
    kind = rng.integers(0, 5)
    if kind == 0:                                    # soup of independent triangles, all sizes
        c = rng.uniform(-1.2, 1.2, size=(N, T, 1, 2))
        r = np.exp(rng.uniform(np.log(0.003), np.log(1.5), size=(N, T, 1, 1)))
        xy = c + r * rng.normal(size=(N, T, 3, 2))
        z = rng.uniform(-1.1, 1.1, size=(N, T, 3, 1)); w = np.ones_like(z)
    elif kind == 1:                                  # perspective: w varies, some vertices behind the eye
        xy = rng.normal(size=(N, T, 3, 2)) * 1.5
        z = rng.normal(size=(N, T, 3, 1)); w = rng.uniform(-0.3, 2.5, size=(N, T, 3, 1))
    elif kind == 2:                                  # snapped to pixel / subpixel positions: ties and on-edge samples
        xy = rng.integers(-W, W + 1, size=(N, T, 3, 2)) / np.array([W / 2.0, H / 2.0]) * rng.choice([1.0, 0.5, 1.0 / 16.0])
        z = rng.choice([-0.5, 0.0, 0.25, 0.5], size=(N, T, 3, 1)); w = np.ones_like(z)
    elif kind == 3:                                  # slivers and near-degenerate triangles
        a = rng.uniform(-1, 1, size=(N, T, 1, 2)); d = rng.normal(size=(N, T, 1, 2))
        t = rng.uniform(-1, 1, size=(N, T, 3, 1))
        xy = a + d * t + rng.normal(size=(N, T, 3, 2)) * rng.choice([0.0, 1e-4, 1e-2])
        z = rng.uniform(-0.9, 0.9, size=(N, T, 3, 1)); w = rng.uniform(0.5, 2.0, size=(N, T, 3, 1))
    else:                                            # layered overdraw: many screen-filling triangles at close depths + small ones
        big = rng.uniform() < 0.5
        c = rng.uniform(-0.8, 0.8, size=(N, T, 1, 2))
        r = np.where(rng.uniform(size=(N, T, 1, 1)) < (0.2 if big else 0.02), 2.0, 0.05)
        xy = c + r * rng.normal(size=(N, T, 3, 2))
        z = rng.uniform(-0.2, 0.2, size=(N, T, 3, 1)) * rng.choice([1.0, 1e-3]); w = np.ones_like(z)
    pos = np.concatenate([xy * w, z * w, w], -1).reshape(N, 3 * T, 4).astype(np.float32)
    tri = np.arange(3 * T, dtype=np.int32).reshape(T, 3)
    if rng.uniform() < 0.3:                          # shared vertices, duplicates, a corrupt index
        tri = rng.integers(0, 3 * T, size=(T, 3)).astype(np.int32)
        if T > 3:
            tri[1] = tri[0]; tri[2] = [0, 3 * T, 1]
    return pos, tri, (H, W)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    t0, n, bad, pix = time.time(), 0, 0, 0
    ctx = dr.RasterizeCudaContext()
    while time.time() - t0 < seconds:
        rng = np.random.default_rng(seed)
        pos, tri, res = scene(rng)
        H, W = res
        ids_o, depth_o = oracle.rasterize_ids(pos, tri, res)
        want = ids_o[:, :H, :W].astype(np.float32)
        cov = ids_o[:, :H, :W] > 0
        P, Tt = t(pos), t(tri)

        def report(what, k):
            nonlocal bad
            if k:
                bad += 1
                print("MISMATCH seed %d %s: %d (N=%d res=%s T=%d)" % (seed, what, k, pos.shape[0], res, tri.shape[0]), flush=True)
        r, _ = dr.rasterize(ctx, P, Tt, res)                                        # the production instantiation
        report("ids (plain)", int((r[..., 3].cpu().numpy() != want).sum()))
        with dr.DepthPeeler(ctx, P, Tt, res) as peeler:
            r0, _ = peeler.rasterize_next_layer()
            d0 = ctx.cpp_wrapper.depth.cpu().numpy().view(np.uint32).copy()
            r1, _ = peeler.rasterize_next_layer()
        report("ids (peel layer 0)", int((r0[..., 3].cpu().numpy() != want).sum()))
        report("depth (peel layer 0)", int((d0[:, :H, :W][cov] != depth_o[:, :H, :W][cov]).sum()))
        ids1, _ = oracle.rasterize_ids(pos, tri, res, peel_depth=depth_o)
        report("ids (peel layer 1)", int((r1[..., 3].cpu().numpy() != ids1[:, :H, :W].astype(np.float32)).sum()))
        if seed % 3 == 0 and pos.shape[0] > 1:                                         # range mode: item n draws a slice of the triangles
            T = tri.shape[0]
            cuts = np.sort(rng.integers(0, T + 1, size=pos.shape[0] + 1))
            ranges = np.stack([cuts[:-1], cuts[1:] - cuts[:-1]], 1).astype(np.int32)
            if (ranges[:, 1] > 0).any():
                ids_r, _ = oracle.rasterize_ids(pos[0], tri, res, ranges=ranges)
                rr, _ = dr.rasterize(ctx, t(pos[0]), Tt, res, ranges=torch.from_numpy(ranges))
                report("ids (range mode)", int((rr[..., 3].cpu().numpy() != ids_r[:, :H, :W].astype(np.float32)).sum()))
        n += 1; seed += 1; pix += cov.size
    print("fuzz_soak: %d scenes, %.1f Mpixels, %d mismatching checks, %.0f s, NVDR_DEBUG=%s" % (n, pix / 1e6, bad, time.time() - t0, os.environ.get("NVDR_DEBUG", "")))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
