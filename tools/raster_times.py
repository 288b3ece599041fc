import sys, json, torch, numpy as np
sys.path.insert(0, '/root/repo')
import nvdiffrast_amd.torch as dr
from nvdiffrast_amd import _capi
from nvdiffrast_amd.utils import m10k_batch
lib = _capi.load()
dev = torch.device("cuda", 0)
b = m10k_batch(64)
pos = torch.from_numpy(b["pos"]).to(dev); tri = torch.from_numpy(b["tri"]).to(dev)
ctx = dr.RasterizeCudaContext()
for _ in range(3): dr.rasterize(ctx, pos, tri, (512, 512))
torch.cuda.synchronize()
lib.nvdr_profile_reset(); lib.nvdr_profile_enable(1)
for _ in range(10): dr.rasterize(ctx, pos, tri, (512, 512))
torch.cuda.synchronize()
pr = _capi.profile_read()
print({k: round(v[0]/v[1], 4) for k, v in pr.items()})
