"""Development tool: how much of k_fine / the backward pass is the imbalance between the XCDs' image ranges?
The benchmark batch as it is (XCD chunk = 8 consecutive images) against the same 64 images dealt to the chunks by covered
area (largest first, always to the lightest chunk): same total work, equal work per XCD."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvdiffrast_amd.torch as dr
from nvdiffrast_amd import _capi
from nvdiffrast_amd.utils import m10k_batch

lib = _capi.load()
dev = torch.device("cuda", 0)
b = m10k_batch(64)
pos = torch.from_numpy(b["pos"]).to(dev); tri = torch.from_numpy(b["tri"]).to(dev)
attr = torch.rand(pos.shape[1], 4, device=dev)
ctx = dr.RasterizeCudaContext()
G = torch.randn(64, 512, 512, 4, device=dev)


def run(p, label):
    p = p.clone().requires_grad_(True)
    a = attr.clone().requires_grad_(True)
    def step():
        rast, _ = dr.rasterize(ctx, p, tri, (512, 512))
        out, _ = dr.interpolate(a, rast, tri)
        p.grad = None; a.grad = None
        torch.autograd.backward(out, G)
        return rast
    for _ in range(3): rast = step()
    torch.cuda.synchronize()
    lib.nvdr_profile_reset(); lib.nvdr_profile_enable(1)
    for _ in range(10): step()
    torch.cuda.synchronize()
    pr = _capi.profile_read(); lib.nvdr_profile_enable(0)
    cov = (rast[..., 3] > 0).float().mean(dim=(1, 2)).cpu().numpy()
    print(label, {k: round(v[0] / v[1] * 1e3, 1) for k, v in pr.items()}, "coverage per XCD chunk", np.round(cov.reshape(8, 8).sum(1), 2))
    return cov


cov = run(pos, "as is   ")
# deal the images to 8 chunks of 8 by covered area
order = np.argsort(-cov)
chunks = [[] for _ in range(8)]; load = np.zeros(8)
for i in order:
    free = [c for c in range(8) if len(chunks[c]) < 8]
    c = min(free, key=lambda k: load[k])
    chunks[c].append(int(i)); load[c] += cov[i]
perm = torch.tensor([i for c in chunks for i in c], device=dev)
run(pos[perm].contiguous(), "balanced")
