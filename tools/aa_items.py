#!/usr/bin/env python3
"""Antialias work-item statistics of BASELINE config 3 (how many pixel pairs, how many with a blend)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvdiffrast_amd.torch as dr
from nvdiffrast_amd.torch import _plugin
from nvdiffrast_amd.utils import m10k_batch
dev = torch.device("cuda", 0)
N, R = 32, 1024
b = m10k_batch(N)
pos = torch.from_numpy(b["pos"]).to(dev); tri = torch.from_numpy(b["tri"]).to(dev)
ctx = dr.RasterizeCudaContext(device=dev)
rast, _ = dr.rasterize(ctx, pos, tri, (R, R))
col = torch.rand((N, R, R, 3), device=dev)
topo = dr.antialias_construct_topology_hash(tri)
out, work = _plugin.antialias_fwd(col, rast, pos, tri, topo)
w = work.view(torch.int32).view(-1, 4) if work.dtype != torch.int32 else work.view(-1, 4)
cnt = int(w[0, 0])
items = w[1:1 + cnt]
nz = int((items[:, 3] != 0).sum())
cov = float((rast[..., 3] > 0).float().mean())
print(json.dumps({"pixels": N * R * R, "coverage": round(cov, 3), "items": cnt, "items_per_pixel": round(cnt / (N * R * R), 4),
                  "items_with_blend": nz, "blend_fraction": round(nz / max(cnt, 1), 4)}))
