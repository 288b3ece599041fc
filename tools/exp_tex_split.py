"""Where k_tex_grad's time goes on config 3's tensors (run under NVDR_DEBUG switches; see tools/README.md):
    python tools/exp_tex_split.py            dense / covered-only / background-only upstream gradients
NVDR_DEBUG bits of texture.hip: 2048 no table clear + flush, 65536 no LDS adds, 131072 no slot lookups and no scatter."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
        dr.texture(tex, uv, uvda, filter_mode='linear-mipmap-linear').backward(g)
    torch.cuda.synchronize()
    lib.nvdr_profile_reset(); lib.nvdr_profile_enable(1)
    for _ in range(5):
        tex.grad = None; uv.grad = None; uvda.grad = None
        dr.texture(tex, uv, uvda, filter_mode='linear-mipmap-linear').backward(g)
    torch.cuda.synchronize()
    prof = _capi.profile_read(); lib.nvdr_profile_enable(0)
    print(os.environ.get('NVDR_DEBUG', '0'), tag, {k: round(v[0] / v[1], 4) for k, v in prof.items() if 'grad' in k})
print('coverage', float(mask.mean()))
run(G, 'dense')
run(G * mask, 'covered only')
run(G * (1 - mask), 'background only')
