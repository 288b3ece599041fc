import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import nvdiffrast_amd.torch as dr
from nvdiffrast_amd import _capi
rng = np.random.default_rng(0)
T = 1_000_000; N = 2; R = 1024
c = rng.uniform(-1, 1, size=(T, 1, 2)); sz = rng.uniform(1.0, 6.0, size=(T, 1, 1)) * (2.0 / R)
ang = rng.uniform(0, 2 * np.pi, size=(T, 1, 1)) + np.array([0, 2.1, 4.2]).reshape(1, 3, 1)
xy = c + sz * np.concatenate([np.cos(ang), np.sin(ang)], -1)
z = rng.uniform(-0.9, 0.9, size=(T, 3, 1))
p1 = np.concatenate([xy, z, np.ones_like(z)], -1).reshape(-1, 4).astype(np.float32)
pos = torch.from_numpy(np.stack([p1, p1[::-1].copy()])).cuda().requires_grad_(True)
tri = torch.arange(3 * T, dtype=torch.int32).reshape(T, 3).cuda()
attr = torch.rand(1, 3 * T, 3, device='cuda', requires_grad=True)
ctx = dr.RasterizeCudaContext()
lib = _capi.load()
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rast, _ = dr.rasterize(ctx, pos, tri, (R, R))
    out, _ = dr.interpolate(attr, rast, tri)
    out.sum().backward()
    torch.cuda.synchronize(); print("iter", it, "ms", (time.perf_counter() - t0) * 1e3)
lib.nvdr_profile_reset(); lib.nvdr_profile_enable(1)
rast, _ = dr.rasterize(ctx, pos, tri, (R, R)); out, _ = dr.interpolate(attr, rast, tri); out.sum().backward()
torch.cuda.synchronize(); print({k: round(v[0] / v[1], 3) for k, v in _capi.profile_read().items()})
print("coverage", float((rast[..., 3] > 0).float().mean()), "distinct ids", int(rast[..., 3].unique().numel()))
def tm(label, f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize()
    print(label, round((time.perf_counter() - t0) * 1e3, 2), "ms"); return r
lib.nvdr_profile_enable(0)
rast, rdb = tm("rasterize", lambda: dr.rasterize(ctx, pos, tri, (R, R)))
out, _ = tm("interpolate", lambda: dr.interpolate(attr, rast, tri))
loss = tm("sum", lambda: out.sum())
tm("backward", lambda: loss.backward())
