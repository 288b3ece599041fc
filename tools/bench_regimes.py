#!/usr/bin/env python3
"""The regimes the benchmark scene hides: rasterize+interpolate fwd+bwd on
  dense         the M10k mesh with the camera pulled in (coverage ~1, overdraw ~1), 64 @512^2
  s10k          SURVEY 8(d)'s stress variant, 64 @512^2
  t1m           a one-million-triangle lattice mesh in index order, 2 @1024^2
  t1m_shuffled  the same mesh with the rows of `tri` permuted
One JSON line per regime: step time, per-kernel hipEvent times of the library, coverage, ids against the C oracle on one item.
    python tools/bench_regimes.py [names...]
"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nvdiffrast_amd.torch as dr
from nvdiffrast_amd import _capi
from nvdiffrast_amd.utils import dense_batch, big_mesh_batch, stress_triangles, m10k_batch


def scene(name):
    if name == "dense":
        return dense_batch(64), 512
    if name == "ch":
        return m10k_batch(64), 512
    if name == "c2":
        return m10k_batch(16), 512
    if name == "s10k":
        b = stress_triangles(64, T=10000, res=512)
        b["attr"] = np.random.default_rng(3).uniform(size=(1, b["pos"].shape[1], 4)).astype(np.float32)
        return b, 512
    if name in ("t1m", "t1m_shuffled"):
        return big_mesh_batch(2, shuffle=name.endswith("shuffled")), 1024
    raise SystemExit("unknown regime " + name)


def run(name, steps=10, check=True):
    dev = torch.device("cuda", 0)
    b, R = scene(name)
    N = b["pos"].shape[0]
    pos = torch.from_numpy(b["pos"]).to(dev).requires_grad_(True)
    tri = torch.from_numpy(b["tri"]).to(dev)
    attr = torch.from_numpy(b["attr"]).to(dev).requires_grad_(True)
    G = torch.randn(N, R, R, 4, device=dev)
    ctx = dr.RasterizeCudaContext(device=dev)

    def step():
        pos.grad = None; attr.grad = None
        rast, _ = dr.rasterize(ctx, pos, tri, (R, R))
        out, _ = dr.interpolate(attr, rast, tri)
        torch.autograd.backward(out, G)
        return rast

    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): rast = step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / steps * 1e3
    lib = _capi.load(); lib.nvdr_profile_reset(); lib.nvdr_profile_enable(1)
    for _ in range(10): step()
    torch.cuda.synchronize(); prof = _capi.profile_read(); lib.nvdr_profile_enable(0); lib.nvdr_profile_reset()
    doc = {"regime": name, "items": N, "res": R, "triangles": int(tri.shape[0]), "ms_per_step": round(ms, 4),
           "Gpix_per_s": round(N * R * R / ms / 1e6, 2), "coverage": round(float((rast[..., 3] > 0).float().mean()), 4),
           "kernels_ms": {k: round(v[0] / v[1], 4) for k, v in prof.items()}}
    if check:
        import oracle
        ro, _ = oracle.rasterize(b["pos"][:1], b["tri"], (R, R))
        rh = rast[:1].detach().cpu().numpy()
        doc["tri_id_mismatches_item0"] = int((rh[..., 3] != ro[..., 3]).sum())
        doc["bary_max_abs_err_item0"] = float(np.abs(rh[..., :3] - ro[..., :3]).max())
    print(json.dumps(doc), flush=True)
    return doc


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if not a.startswith("--")]
    for nm in (names or ["ch", "dense", "s10k", "t1m", "t1m_shuffled"]):
        run(nm, check="--no-check" not in sys.argv)
