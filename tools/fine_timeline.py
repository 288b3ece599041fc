"""Development tool: per-workgroup phase timing of k_fine on the benchmark scene (uses nvdr_debug_buffer)."""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvdiffrast_amd.torch as dr
from nvdiffrast_amd import _capi
from nvdiffrast_amd.utils import m10k_batch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lib = _capi.load()
lib.nvdr_debug_buffer.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda", 0)
b = m10k_batch(N)
pos = torch.from_numpy(b["pos"]).to(dev); tri = torch.from_numpy(b["tri"]).to(dev)
ctx = dr.RasterizeCudaContext()
for _ in range(3):
    dr.rasterize(ctx, pos, tri, (512, 512))
torch.cuda.synchronize()
nwg = N * 64
WAVES = 8
buf = torch.zeros(nwg * WAVES * 8, dtype=torch.int64, device=dev)
lib.nvdr_debug_buffer(buf.data_ptr())
dr.rasterize(ctx, pos, tri, (512, 512))
torch.cuda.synchronize()
lib.nvdr_debug_buffer(None)
d = buf.cpu().numpy().reshape(nwg, WAVES, 8).astype(np.float64)
t0 = d[:, :, 0].min()
start = (d[:, 0, 0] - t0) / 100.0          # wall_clock64 = 100 MHz -> us
end = (d[:, :, 6].max(1) - t0) / 100.0
dur = end - start
rounds = d[:, 0, 1]; filt = d[:, :, 2].max(1) / 100.0; rast = d[:, :, 3].max(1) / 100.0; cnt = d[:, 0, 4]
shade = (d[:, :, 6] - d[:, :, 5]).max(1) / 100.0
print("kernel span us", end.max(), "first start", start.min(), "last start", start.max())
print("WG duration us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()))
print("filter us (max wave): mean %.1f p90 %.1f max %.1f" % (filt.mean(), np.percentile(filt, 90), filt.max()))
print("raster us (max wave): mean %.1f p90 %.1f max %.1f" % (rast.mean(), np.percentile(rast, 90), rast.max()))
print("shade+store us: mean %.1f p90 %.1f max %.1f" % (shade.mean(), np.percentile(shade, 90), shade.max()))
print("rounds: mean %.2f max %d; list total mean %.0f max %d" % (rounds.mean(), rounds.max(), cnt.mean(), cnt.max()))
idx = np.argsort(-dur)[:8]
for i in idx:
    print("WG", i, "img", i // 64, "bin", i % 64, "start %.1f dur %.1f filt %.1f rast %.1f shade %.1f rounds %d cnt %d" % (start[i], dur[i], filt[i], rast[i], shade[i], rounds[i], cnt[i]))
# concurrency: how many WGs alive over time
ts = np.linspace(0, end.max(), 20)
print("alive WGs over time:", [(int(((start <= t) & (end > t)).sum())) for t in ts])
emp = cnt == 0
print("empty WGs: %d, their duration mean %.1f" % (emp.sum(), dur[emp].mean()))
# per-XCD chunks of the work order (item index = position in the order; 8 contiguous chunks)
per = (nwg + 7) // 8
for x in range(8):
    sl = slice(x * per, min((x + 1) * per, nwg))
    print("XCD chunk %d: last end %.1f us, sum of WG durations %.0f us, non-empty %d, triangles listed %d"
          % (x, end[sl].max(), dur[sl].sum(), int((cnt[sl] > 0).sum()), int(cnt[sl].sum())))
# occupancy seen by each starting workgroup inside its XCD chunk (128 slots per XCD = 32 CUs x 4)
for x in range(8):
    sl = slice(x * per, min((x + 1) * per, nwg))
    s_, e_ = start[sl], end[sl]
    order_ = np.argsort(s_)
    alive_at_start = np.array([int(((s_ <= s_[i]) & (e_ > s_[i])).sum()) for i in order_])
    qs = [int(alive_at_start[int(f * (len(order_) - 1))]) for f in (0.1, 0.25, 0.5, 0.75, 0.9, 1.0)]
    ts = np.linspace(0, e_.max(), 9)[1:-1]
    print("XCD %d: alive when a WG starts (10/25/50/75/90/100%% of starts): %s; alive over time %s" % (x, qs, [int(((s_ <= t) & (e_ > t)).sum()) for t in ts]))
