import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import oracle
from oracle import ref
import nvdiffrast_amd.torch as dr
from nvdiffrast_amd.utils import m10k_batch
b = m10k_batch(2, seed=4, nx=16, ny=8)
tri = b["tri"].copy(); V = b["pos"].shape[1]
tri[3] = [0, V, 1]; tri[10] = [-1, 2, 3]; tri[11] = [5, 5, 5]; tri[12] = [7, 8, 7]
dev = torch.device("cuda", 0)
ctx = dr.RasterizeCudaContext()
r, _ = dr.rasterize(ctx, torch.from_numpy(b["pos"]).to(dev), torch.from_numpy(tri).to(dev), (96, 80))
r = r.cpu().numpy()
ro, _ = oracle.rasterize(b["pos"], tri, (96, 80))
print("ids equal", (r[..., 3] == ro[..., 3]).all(), "max float diff", np.abs(r - ro).max())
col = np.random.default_rng(0).uniform(size=r.shape[:3] + (3,)).astype(np.float32)
for name, rr in (("gpu r", r), ("oracle r", ro)):
    for i in range(3):
        a = oracle.antialias(col, rr, b["pos"], tri)
        c = ref.antialias(col, rr, b["pos"], tri)
        d = np.abs(a - c)
        print(name, i, d.max(), (d > 1e-5).sum(), np.argwhere(d > 1e-5)[:3].tolist())
        if d.max() > 1e-5:
            np.savez("/root/repo/gpurun_out/r04k/aa_case.npz", r=rr, col=col, pos=b["pos"], tri=tri, a=a, c=c)
