"""NVDR_DEBUG=2097152 makes the library return before launching anything: the loop then measures the host alone.
Where the HOST time of one headline step goes (a batch too small to keep the GPU busy, so the step time is the
host's): cProfile over the eager loop.  Run on a GPU box: python tools/host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvdiffrast_amd.torch as dr  # noqa: E402
from nvdiffrast_amd.utils import m10k_batch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda", 0)
b = m10k_batch(1, seed=1, nx=16, ny=8)
pos = torch.from_numpy(b["pos"]).to(dev).requires_grad_(True)
attr = torch.from_numpy(b["attr"]).to(dev).requires_grad_(True)
tri = torch.from_numpy(b["tri"]).to(dev)
G = torch.randn(1, 64, 64, 4, device=dev)
ctx = dr.RasterizeCudaContext(device=dev)


def step():
    pos.grad = None
    attr.grad = None
    rast, _ = dr.rasterize(ctx, pos, tri, (64, 64))
    out, _ = dr.interpolate(attr, rast, tri)
    torch.autograd.backward(out, G)


for _ in range(50):
    step()
torch.cuda.synchronize()
# (a device synchronisation every 500 steps, outside the clock: a loop that enqueues for thousands of steps without ever waiting
# measures the HIP runtime recycling its signal / kernarg pools -- 57 instead of 34 us per step with the launches off -- which no
# real loop does: it is throttled by its kernels)
spent = 0.0
for i0 in range(0, steps, 500):
    t0 = time.perf_counter()
    for _ in range(min(500, steps - i0)):
        step()
    spent += time.perf_counter() - t0
    torch.cuda.synchronize()
print("us per step: %.1f" % (spent / steps * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
