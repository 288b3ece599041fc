"""Host time of one tiny step through a REFERENCE-STYLE binding of `_plugin` (autograd functions with the reference's call lists,
tests/test_gpu_plugin_fused_backward.py `_binding`): what a user of the reference's own ops.py pays per step.
NVDR_HOST=0: the plugin's entry points in Python; default: its forward entry points served by the compiled layer."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nvdiffrast_amd.torch as dr  # noqa: E402
from nvdiffrast_amd.torch import _plugin  # noqa: E402
from nvdiffrast_amd.utils import m10k_batch  # noqa: E402
from test_gpu_plugin_fused_backward import _binding  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda", 0)
b = m10k_batch(1, seed=1, nx=16, ny=8)
pos = torch.from_numpy(b["pos"]).to(dev).requires_grad_(True)
attr = torch.from_numpy(b["attr"]).to(dev).requires_grad_(True)
tri = torch.from_numpy(b["tri"]).to(dev)
G = torch.randn(1, 64, 64, 4, device=dev)
state = dr.RasterizeCudaContext(device=dev).cpp_wrapper
Rasterize, Interpolate, _ = _binding(_plugin, materialize=False)
spent = 0.0
for i0 in range(-500, steps, 500):
    t0 = time.perf_counter()
    for _ in range(500):
        pos.grad = None; attr.grad = None
        rast, _db = Rasterize.apply(state, pos, tri, (64, 64), True)
        out, _da = Interpolate.apply(attr, rast, tri)
        torch.autograd.backward(out, G)
    if i0 >= 0:
        spent += time.perf_counter() - t0
    torch.cuda.synchronize()
print(_plugin.host_layer_name(), "reference-style binding, us per step: %.1f" % (spent / steps * 1e6))
