import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, '/root/repo')
import nvdiffrast_amd.torch as dr
from nvdiffrast_amd import _capi
from nvdiffrast_amd.utils import m10k_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64      # run with NVDR_DEBUG=64
lib = _capi.load()
lib.nvdr_debug_buffer.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda", 0)
b = m10k_batch(N)
pos = torch.from_numpy(b["pos"]).to(dev); tri = torch.from_numpy(b["tri"]).to(dev)
ctx = dr.RasterizeCudaContext()
for _ in range(2): dr.rasterize(ctx, pos, tri, (512, 512))
torch.cuda.synchronize()
nwg = N * 64; W = 8
buf = torch.zeros(nwg * W * 8, dtype=torch.int64, device=dev)
lib.nvdr_debug_buffer(buf.data_ptr())
dr.rasterize(ctx, pos, tri, (512, 512)); torch.cuda.synchronize()
lib.nvdr_debug_buffer(None)
d = buf.cpu().numpy().reshape(nwg, W, 8).astype(np.float64)
cand = d[:, 0, 2]; surv = d[:, :, 3].sum(1); cnt = d[:, 0, 4]
print("total tris in lists %.0f, (triangle, tile) pairs %.0f, pairs with a non-empty coverage mask %.0f" % (cnt.sum(), cand.sum(), surv.sum()))
i = np.argsort(-cand)[:6]
for k in i: print("WG", k, "cnt", cnt[k], "cand", cand[k], "surv", surv[k])
print("pairs per tri mean %.2f; non-empty per tri %.2f" % (cand.sum() / cnt.sum(), surv.sum() / cnt.sum()))
