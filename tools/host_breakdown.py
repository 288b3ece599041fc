"""Where the host time of one tiny step goes, call by call (perf_counter around rasterize / interpolate / backward, no device
synchronisation inside the loop).  NVDR_DEBUG=2097152: the library returns before launching.  python tools/host_breakdown.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvdiffrast_amd.torch as dr  # noqa: E402
from nvdiffrast_amd.torch import _plugin  # noqa: E402
from nvdiffrast_amd.utils import m10k_batch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda", 0)
b = m10k_batch(1, seed=1, nx=16, ny=8)
pos = torch.from_numpy(b["pos"]).to(dev).requires_grad_(True)
attr = torch.from_numpy(b["attr"]).to(dev).requires_grad_(True)
tri = torch.from_numpy(b["tri"]).to(dev)
G = torch.randn(1, 64, 64, 4, device=dev)
ctx = dr.RasterizeCudaContext(device=dev)
pc = time.perf_counter
acc = [0.0] * 5
for it in range(steps + 100):
    t0 = pc()
    pos.grad = None
    attr.grad = None
    t1 = pc()
    rast, _ = dr.rasterize(ctx, pos, tri, (64, 64))
    t2 = pc()
    out, _ = dr.interpolate(attr, rast, tri)
    t3 = pc()
    torch.autograd.backward(out, G)
    t4 = pc()
    del rast, out, _
    t5 = pc()
    if it >= 100:
        for k, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            acc[k] += d
    if it % 500 == 0:
        torch.cuda.synchronize()
torch.cuda.synchronize()
names = ("grad=None", "rasterize", "interpolate", "backward", "del")
print(_plugin.host_layer_name(), "us per step:", {n: round(a / steps * 1e6, 1) for n, a in zip(names, acc)}, "sum", round(sum(acc) / steps * 1e6, 1))
