import sys, json, numpy as np, torch
sys.path.insert(0, '/root/repo')
import nvdiffrast_amd.torch as dr
from nvdiffrast_amd import _capi
from nvdiffrast_amd.utils import m10k_batch
dev = torch.device('cuda', 0)
N, R = 32, 1024
b = m10k_batch(N)
pos = torch.from_numpy(b['pos']).to(dev); uvattr = torch.from_numpy(b['uv']).to(dev); tri = torch.from_numpy(b['tri']).to(dev)
tex = torch.from_numpy(np.random.default_rng(5).uniform(size=(1, 2048, 2048, 3)).astype(np.float32)).to(dev).requires_grad_(True)
ctx = dr.RasterizeCudaContext(device=dev)
rast, rdb = dr.rasterize(ctx, pos, tri, (R, R))
uv, uvda = dr.interpolate(uvattr, rast, tri, rast_db=rdb, diff_attrs='all')
uv = uv.detach().requires_grad_(True); uvda = uvda.detach().requires_grad_(True)
G = torch.randn((N, R, R, 3), device=dev)
mask = (rast[..., 3:4] > 0).float()
lib = _capi.load()
def run(g, tag):
    for _ in range(2):
        tex.grad = None; uv.grad = None; uvda.grad = None
        col = dr.texture(tex, uv, uvda, filter_mode='linear-mipmap-linear')
        col.backward(g)
    torch.cuda.synchronize()
    lib.nvdr_profile_reset(); lib.nvdr_profile_enable(1)
    for _ in range(5):
        tex.grad = None; uv.grad = None; uvda.grad = None
        col = dr.texture(tex, uv, uvda, filter_mode='linear-mipmap-linear')
        col.backward(g)
    torch.cuda.synchronize()
    prof = _capi.profile_read(); lib.nvdr_profile_enable(0)
    print(tag, {k: round(v[0]/v[1], 4) for k, v in prof.items()})
print('coverage', float(mask.mean()))
run(G, 'dense G')
run(G * mask, 'G masked to covered pixels')
run(G * (1 - mask), 'G only on background')
# hypothesis check: is the background's cost the cross-block contention on the same 4 texels?  Give every 16x16 block
# its own constant uv (still uniform inside each wave) and compare.
by, bx = torch.meshgrid(torch.arange(R, device=dev) // 16, torch.arange(R, device=dev) // 16, indexing='ij')
uv_blk = torch.stack([(bx.float() * 7.0 + 0.5) / 2048.0, (by.float() * 5.0 + 0.5) / 2048.0], -1)[None].expand(N, -1, -1, -1)
bg = (1 - mask)
uv2 = (uv.detach() * mask + uv_blk * bg).contiguous().requires_grad_(True)
uv_saved = uv
uv = uv2
run(G * bg, 'G only on background, per-block distinct uv')
