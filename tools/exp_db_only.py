#!/usr/bin/env python3
"""Development: time of the rast_db-only pass of rasterize_grad (dy == NULL) over a zero ddb at the headline batch, with and
without tile flags, next to the ordinary rasterize_grad_db over the same tensors."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvdiffrast_amd import _capi
from nvdiffrast_amd.torch import _plugin
from nvdiffrast_amd.utils import m10k_batch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
b = m10k_batch(N)
pos = torch.from_numpy(b["pos"]).to(dev)
tri = torch.from_numpy(b["tri"]).to(dev)
state = _plugin.RasterizeCRStateWrapper(0)
rast, rast_db = _plugin.rasterize_fwd_cuda(state, pos, tri, (512, 512), torch.empty((0, 2), dtype=torch.int32), -1)
flags = state.last_flags
zeros = torch.zeros_like(rast_db)
dy = torch.randn_like(rast)
g = torch.zeros_like(pos)
lib = _capi.load()
V, T = pos.shape[1], tri.shape[0]
st = torch.cuda.current_stream().cuda_stream


def run(dy_, ddb_, fl):
    return lib.nvdr_rasterize_grad(pos.data_ptr(), tri.data_ptr(), rast.data_ptr(), _capi.ptr(dy_), _capi.ptr(ddb_), 1, N, V, T, 512, 512,
                                   g.data_ptr(), _capi.ptr(fl), st)


def t(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("record found:", _plugin._record_of(rast, "rast") is not None, " auto flags:", _plugin._auto_flags("x", None, "rast", rast) is flags)
print("db-only, zero ddb, flags   : %.1f us" % t(lambda: run(None, zeros, flags)))
print("db-only, zero ddb, no flags: %.1f us" % t(lambda: run(None, zeros, None)))
print("grad_db, zero ddb, flags   : %.1f us" % t(lambda: run(dy, zeros, flags)))
print("grad,            flags     : %.1f us" % t(lambda: run(dy, None, flags)))
print("zeros_like(rast_db)        : %.1f us" % t(lambda: torch.zeros_like(rast_db)))
