import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvdiffrast_amd.torch as dr
import oracle
fm, bm = sys.argv[1], sys.argv[2]
rng = np.random.default_rng(321)
N, H, W, C = 2, 48, 80, 3
tex = rng.uniform(size=(1, 32, 64, C)).astype(np.float32)
uv = rng.uniform(-0.2, 1.2, size=(N, H, W, 2)).astype(np.float32)
mip = "mipmap" in fm
uv_da = (rng.normal(size=(N, H, W, 4)) * 0.05).astype(np.float32)
uv[0, :, :48] = 0.0; uv_da[0, :, :48] = 0.0
uv[1, 8:40, 16:64] = np.array([0.37, 0.61], np.float32); uv_da[1, 8:40, 16:64] = 0.0
uv[1, :8, :32] = 0.0
bias = rng.uniform(-0.5, 0.5, size=(N, H, W)).astype(np.float32) if fm == "linear-mipmap-linear" else None
dy = rng.normal(size=(N, H, W, C)).astype(np.float32)
dy[0, 16:24, :16] = 0.0
kw = dict(filter_mode=fm, boundary_mode=bm)
_t = lambda a: torch.from_numpy(a).cuda()
t_tex = _t(tex).requires_grad_(True); t_uv = _t(uv).requires_grad_(True)
t_da = _t(uv_da).requires_grad_(True) if mip else None
t_bias = _t(bias).requires_grad_(True) if bias is not None else None
out = dr.texture(t_tex, t_uv, t_da, t_bias, **kw)
out.backward(_t(dy))
g = oracle.texture_grad(tex, uv, dy, uv_da if mip else None, bias, **kw)
e = np.abs(t_tex.grad.cpu().numpy() - g["tex"])
print("tex err max", e.max(), "at", np.argwhere(e > 1e-2)[:12].tolist())
idx = np.argwhere(e > 1e-2)
for i in idx[:6]:
    print(tuple(i), "got", t_tex.grad.cpu().numpy()[tuple(i)], "want", g["tex"][tuple(i)])
eu = np.abs(t_uv.grad.cpu().numpy() - g["uv"]).max(-1)
print("uv err max", eu.max(), "pixels", np.argwhere(eu > 1e-3)[:10].tolist())
