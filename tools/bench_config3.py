#!/usr/bin/env python3
"""BASELINE config 3: 10k-tri mesh + 2048^2 mipmapped texture() fwd+bwd + antialias(), batch 32 @1024^2.
Prints per-kernel hipEvent times (library profile hooks) and the step time.  Development/measurement tool."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvdiffrast_amd.torch as dr
from nvdiffrast_amd import _capi
from nvdiffrast_amd.utils import m10k_batch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--tex", type=int, default=2048)
ap.add_argument("--steps", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda", 0)
N, R = args.batch, args.res
b = m10k_batch(N)
pos = torch.from_numpy(b["pos"]).to(dev).requires_grad_(True)
uvattr = torch.from_numpy(b["uv"]).to(dev).requires_grad_(True)
tri = torch.from_numpy(b["tri"]).to(dev)
rng = np.random.default_rng(5)
tex = torch.from_numpy(rng.uniform(size=(1, args.tex, args.tex, 3)).astype(np.float32)).to(dev).requires_grad_(True)
G = torch.randn((N, R, R, 3), device=dev)
ctx = dr.RasterizeCudaContext(device=dev)
topo = dr.antialias_construct_topology_hash(tri)

def step():
    pos.grad = uvattr.grad = tex.grad = None
    rast, rast_db = dr.rasterize(ctx, pos, tri, (R, R))
    uv, uv_da = dr.interpolate(uvattr, rast, tri, rast_db=rast_db, diff_attrs="all")
    col = dr.texture(tex, uv, uv_da, filter_mode="linear-mipmap-linear")
    out = dr.antialias(col, rast, pos, tri, topology_hash=topo)
    torch.autograd.backward(out, G)

for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / args.steps * 1e3
lib = _capi.load()
lib.nvdr_profile_reset(); lib.nvdr_profile_enable(1)
for _ in range(5):
    step()
torch.cuda.synchronize()
prof = _capi.profile_read()
lib.nvdr_profile_enable(0)
P = N * R * R
print(json.dumps({"config": "C3", "batch": N, "res": R, "tex": args.tex, "ms_per_step": round(ms, 3),
                  "Mpix_per_s": round(P / ms / 1e3, 1),
                  "kernels_ms": {k: round(v[0] / max(v[1], 1) * (v[1] / 5), 4) for k, v in prof.items()}}))
