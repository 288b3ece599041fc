#!/usr/bin/env python3
"""S10k stress variant: 10,000 independent triangles per image, edge length log-uniform in
[2,64] px, heavy overdraw, no shared vertices; rasterize+interpolate fwd+bwd @512^2 batch 64."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvdiffrast_amd.torch as dr
from nvdiffrast_amd import _capi
from nvdiffrast_amd.utils import stress_triangles

N, R = 64, 512
dev = torch.device("cuda", 0)
b = stress_triangles(N, T=10000, res=R)
pos = torch.from_numpy(b["pos"]).to(dev).requires_grad_(True)
tri = torch.from_numpy(b["tri"]).to(dev)
attr = torch.rand(1, pos.shape[1], 4, device=dev, requires_grad=True)
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
for _ in range(10): rast = step()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
lib = _capi.load(); lib.nvdr_profile_reset(); lib.nvdr_profile_enable(1)
for _ in range(5): step()
torch.cuda.synchronize(); prof = _capi.profile_read(); lib.nvdr_profile_enable(0)
cov = float((rast[..., 3] > 0).float().mean())
print(json.dumps({"workload": "S10k", "ms_per_step": round(ms, 3), "Mpix_per_s": round(N * R * R / ms / 1e3, 1), "coverage": round(cov, 3),
                  "kernels_ms": {k: round(v[0] / v[1], 4) for k, v in prof.items()}}))
