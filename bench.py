#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X.  Default = BASELINE.json's headline metric: Mpixels/s of
rasterize+interpolate forward+backward at 512^2, batch 64 per GPU (SURVEY.md 8(d) workload "CH").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ch|c2|c3|c4]

`--gpus N` with N > 1 and no launcher environment re-executes this script under `python -m torch.distributed.run`
with N ranks (one process per GPU, RCCL over xGMI); under a launcher (RANK / WORLD_SIZE set) it joins that job.

Workloads (BASELINE.json `configs`):
  ch  (default) the metric: 10k-triangle mesh, 64 items per GPU @512^2, A = 4           weak scaling
  c2  configs[1]: the same graph, 16 items @512^2 on one GPU                            weak scaling
  c3  configs[2]: + 2048^2 mipmapped texture() + antialias(), 32 items per GPU @1024^2  weak scaling
  c4  configs[3]: 256 items @512^2 in total, 256/N per GPU (32 per GPU at N = 8)        strong scaling
and the regimes the benchmark scene does not show: the same graph as ch on
  dense         the benchmark mesh with the camera pulled in: coverage 1.0, overdraw ~1, 64 items @512^2
  s10k          SURVEY 8(d)'s stress variant: 10 000 independent triangles of 2..64 px per item, 64 items @512^2
  t1m           a one-million-triangle lattice mesh in index order, 2 items @1024^2
  t1m_shuffled  the same mesh with the rows of `tri` permuted (a triangle soup as far as binning is concerned)

One step = one pass of the hot path over the rank's items, inputs resident in HBM:
    rast, rast_db = rasterize(ctx, pos, tri, (H, W));  out, _ = interpolate(attr, rast, tri)
    torch.autograd.backward(out, G)                    # upstream gradient G ~ N(0,1), fixed
(c3: interpolate(uv, diff_attrs='all') -> texture(trilinear) -> antialias in between.)
With N > 1 the step also contains the path's two exchanges (north_star): the all-gather of the per-item
output images to every rank -- issued per chunk of items on RCCL's stream while the next chunk is being rendered
(`--chunks`, `--no-gather-images`) -- and the all-reduce of the gradient of the SHARED vertex attributes.

Timing: after W warm-up steps, `--windows` (default 5) windows of EXACTLY K steps each, every window bracketed by
barrier + synchronize on both sides and timed with the host clock (max over ranks) and with a hipEvent pair;
`ms_per_step` is the MEDIAN window, `ms_per_step_min/max` the spread (median over event pairs).

Rank 0 prints ONE JSON line -- a COMPACT record, under 8 KB (the driver keeps 8 KB of stdout) -- and writes the complete
record, everything described below with all its prose, to `bench_detail.json` next to this file (the line names it).  Short
keys of the line: kernels {name: [avg ms per launch, alg frac, req frac]}; configs.<name> {ms, gpix, cov, k {kernel: ms},
dom [longest kernel, alg frac, req frac], path [alg frac, req frac], par {...}}.  Beside the headline number it carries
  roofline      the LONGEST kernel of the step (hipEvents recorded by the library on the launch stream):
                `frac` = ALGORITHMIC bytes / time / 8 TB/s (SURVEY 8(d)'s convention: compulsory tensor traffic, whether or
                not a kernel can skip part of it); `traffic` = the PMC-measured HBM bytes of that kernel per launch, taken
                from profiles/traffic.json (a builder-session rocprofv3 measurement, named in `traffic_source` -- NOT
                measured in this run); `hbm_frac_counter` = traffic / time / peak = the physically moved bytes' rate;
                `frac_required` = REQUIRED bytes / time / peak: writes for every pixel, reads only for what the kernel cannot
                avoid -- rast / rast_db / uv of tiles the rasterizer flagged non-empty, upstream gradients of covered pixels
                (from this run's coverage and tile flags) -- i.e. the bytes that do move; no bandwidth in the line exceeds the peak;
  path_hbm_frac the whole step's algorithmic bytes / step time against the same peak; path_hbm_frac_required the same with
                required bytes; path_hbm_frac_counter with the PMC-counted bytes of profiles/traffic.json (where they exist);
  cpu_baseline  the CPU oracle (a port of the reference's algorithm -- the reference has no CPU path) on this box's
                host cores, bounded sample; cpu_reference = the reference's own kernels under the CUDA-on-CPU shim
                of oracle/refshim (one thread), when oracle/_ref is present;
  parity        id mismatches / max-abs errors of this very workload against the reference itself (oracle/_ref) when
                present, else against the oracle;
  configs       (default single-GPU run only) BASELINE configs[1], [2] and the [4] stand-in measured in the same process:
                per config ms_per_step (median of windows), per-kernel times, roofline of its longest kernel and a parity
                block of that very workload at its full resolution (`--no-extra-configs` leaves them out).
`--dry-run-cpu` replaces the kernels by a stand-in and RCCL by gloo so that the launch / collective / timing
plumbing can be exercised on a machine without GPUs (tests/test_bench_spawn.py); its line says "dry_run": true.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
XGMI_LINK_GBS = 153.0          # per direction per link; 7 links per GPU, fully connected (task statement / SURVEY 8e)
HOST_MS_PER_CHUNK = 0.07       # measured host cost of issuing one chunk's six launches eagerly (compiled host layer: DESIGN 5; 0.20 through round 5)
COPY_GBS = 5400.0              # streaming copy rate of one MI355X (tools/write_bw.py: read + write), prices the image-packing pass
TRIANGLES = 10000

WORKLOADS = {
    "ch": dict(per_gpu=64, total=None, res=512, graph="ri", scaling="weak", attrs=4,
               metric="Mpixels/s rasterize+interpolate fwd+bwd @512^2 batch64"),
    "c2": dict(per_gpu=16, total=None, res=512, graph="ri", scaling="weak", attrs=4,
               metric="Mpixels/s rasterize+interpolate fwd+bwd @512^2 batch16 (BASELINE configs[1])"),
    "c3": dict(per_gpu=32, total=None, res=1024, graph="full", scaling="weak", attrs=2,
               metric="Mpixels/s rasterize+interpolate+texture(2048^2 mip)+antialias fwd+bwd @1024^2 batch32 (BASELINE configs[2])"),
    "c4": dict(per_gpu=None, total=256, res=512, graph="ri", scaling="strong", attrs=4,
               metric="Mpixels/s rasterize+interpolate fwd+bwd @512^2 batch256 sharded over the GPUs (BASELINE configs[3])"),
    "dense": dict(per_gpu=64, total=None, res=512, graph="ri", scaling="weak", attrs=4, scene="dense",
                  metric="Mpixels/s rasterize+interpolate fwd+bwd @512^2 batch64, camera pulled in (coverage 1.0)"),
    "s10k": dict(per_gpu=64, total=None, res=512, graph="ri", scaling="weak", attrs=4, scene="s10k",
                 metric="Mpixels/s rasterize+interpolate fwd+bwd @512^2 batch64, S10k stress triangles (SURVEY 8(d))"),
    "t1m": dict(per_gpu=2, total=None, res=1024, graph="ri", scaling="weak", attrs=4, scene="t1m",
                metric="Mpixels/s rasterize+interpolate fwd+bwd @1024^2 batch2, 1M-triangle mesh in index order"),
    "t1m_shuffled": dict(per_gpu=2, total=None, res=1024, graph="ri", scaling="weak", attrs=4, scene="t1m_shuffled",
                         metric="Mpixels/s rasterize+interpolate fwd+bwd @1024^2 batch2, 1M-triangle mesh, tri rows shuffled"),
}
PARITY_BAR = ("ids identical; forward floats 1e-5 abs; gradients 1e-5*max(1,|g|inf) -- every gradient is a SUM of per-pixel terms "
              "(attr: all items into one tensor), so the bar is relative to its magnitude (tests/conftest.py)")


def build_scene(name, N, A, dry=False, rank=0):
    """The workload's geometry (numpy): pos [N,V,4], tri [T,3], attr [1,V,A], uv [1,V,2] where the scene has one."""
    from nvdiffrast_amd.utils import m10k_batch, dense_batch, big_mesh_batch, stress_triangles
    if dry:
        return m10k_batch(N, seed=20240, attrs=A, nx=8, ny=4, pose_seed=20240 + 1000 * rank)
    if name == "dense":
        return dense_batch(N, attrs=A)
    if name == "s10k":
        b = stress_triangles(N, T=TRIANGLES, res=512)
        b["attr"] = np.random.default_rng(3).uniform(size=(1, b["pos"].shape[1], A)).astype(np.float32)
        return b
    if name in ("t1m", "t1m_shuffled"):
        return big_mesh_batch(N, attrs=A, shuffle=name.endswith("shuffled"))
    return m10k_batch(N, seed=20240, attrs=A, pose_seed=20240 + 1000 * rank)


KERNEL_PASSES = {"tex_grad": ("tex_grad_light", "tex_grad", "tex_grad_fold")}


def algorithmic_bytes(graph, P, A, T, N):
    """SURVEY.md 8(d) / DESIGN.md 5-6: compulsory tensor traffic per kernel launch (geometry is cache resident).
    P = pixels of the launch.  Returns ({kernel: bytes}, bytes of the whole step)."""
    if graph == "ri":
        per_kernel = {
            "raster_setup": N * T * (12 + 48 + 68),            # tri + 3 verts in, record + AABB out
            "raster_fine": 32 * P,                              # W rast 16 + W rast_db 16
            "interp_fwd": (16 + 4 * A) * P,                     # R rast, W out
            "interp_grad": (4 * A + 16 + 16) * P,               # R dy, R rast, W g_rast
            "raster_grad": 32 * P,                              # R g_rast 16 + R rast 16
            "interp_raster_grad": (4 * A + 16) * P,             # fused backward as the operator layer runs it: R dy, R rast (g_rast is not written: ops._LazyGrad)
        }
        return per_kernel, (112 + 8 * A) * P
    C = 3
    per_kernel = {
        "raster_setup": N * T * (12 + 48 + 68),
        "raster_fine": 32 * P,
        "interp_fwd_da": (32 + 4 * A + 8 * A) * P,              # R rast, rast_db; W uv, uv_da
        "tex_fwd": (4 * A + 8 * A + 4 * C) * P,                 # R uv, uv_da; W colour (texel taps are cache traffic)
        "aa_discontinuity": 16 * P,                             # R rast (ids)
        "tex_grad": (4 * C + 4 * A + 8 * A + 4 * A + 8 * A) * P,  # R dy, uv, uv_da; W g_uv, g_uv_da
        "interp_grad_da": (4 * A + 8 * A + 32 + 32) * P,        # R dy, dda, rast, rast_db; W g_rast, g_rast_db
        "interp_raster_grad_da": (4 * A + 8 * A + 32) * P,      # the fused pair as the operator layer runs it: R dy, dda, rast, rast_db (g_rast, g_rast_db are not written)
        "raster_grad_db": 48 * P,                               # R g_rast, g_rast_db, rast
    }
    return per_kernel, 384 * P


def required_bytes(graph, P, A, T, N, cov, tile_cov):
    """Bytes a kernel cannot avoid: writes for every pixel, reads only for non-empty 8x8 tiles (rast,
    rast_db, uv, uv_da: `tile_cov` of the pixels) or covered pixels (upstream gradients the kernels fetch per pixel: `cov`)."""
    if cov is None or tile_cov is None:
        return {}, None
    if graph == "ri":
        k = {
            "raster_setup": N * T * (12 + 48 + 68),
            "raster_fine": 32 * P,
            "interp_fwd": (16 * tile_cov + 4 * A) * P,
            "interp_grad": (4 * A * cov + 16 * tile_cov + 16) * P,
            "raster_grad": (16 * cov + 16 * tile_cov) * P,
            "interp_raster_grad": (4 * A * cov + 16 * tile_cov) * P,
        }
        step = k["raster_fine"] + k["interp_fwd"] + k["interp_raster_grad"]
        return k, step
    C = 3
    k = {
        "raster_setup": N * T * (12 + 48 + 68),
        "raster_fine": 32 * P,
        "interp_fwd_da": (32 * tile_cov + 12 * A) * P,
        "tex_fwd": (12 * A * tile_cov + 4 * C) * P,
        "aa_discontinuity": 16 * tile_cov * P,
        "tex_grad": (4 * C + 12 * A * tile_cov + 12 * A) * P,
        "interp_grad_da": (12 * A * cov + 32 * tile_cov + 32) * P,
        "interp_raster_grad_da": (12 * A * cov + 32 * tile_cov) * P,
        "raster_grad_db": (32 * cov + 16 * tile_cov) * P,
    }
    step = (k["raster_fine"] + k["interp_fwd_da"] + k["tex_fwd"] + k["aa_discontinuity"] + 2 * 4 * C * P      # + antialias: out = color fwd, g_color = dy bwd
            + k["tex_grad"] + k["interp_raster_grad_da"])
    return k, step


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def respawn_under_launcher(args):
    """`bench.py --gpus N` without a launcher: become `torch.distributed.run` with N ranks on this node."""
    if not args.dry_run_cpu:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            print(json.dumps({"error": "bench.py --gpus %d: only %d GPU(s) visible on this node" % (args.gpus, have)}))
            sys.exit(2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["NVDR_BENCH_SPAWNED"] = "1"
    sys.exit(subprocess.call(cmd, env=env))


def plan_chunks(n_items, item_bytes, compute_ms, links):
    """Chunk count of a rank's step when the per-item images are all-gathered inside it.

    The gather's duration does not depend on the chunking: every link carries one peer's images, G = n_items * item_bytes /
    link bandwidth.  What chunking changes is how early the first gather can START -- after the first chunk's forward
    kernels, C / (2c) into the step -- against the host's launch work, which bounds a chunk from below (a chunk whose
    kernels are shorter than the host's ~0.2 ms of launch calls makes the step host-bound).  Model of the step:
        t(c) = max(C, c * h, first_forward(c) + G),   first_forward(c) = max(C / (2 c), h / 2)
    and the smallest c within 2 % of the minimum is taken (c <= 8, c <= n_items)."""
    if links <= 0:
        return 1, None
    G = n_items * item_bytes / (XGMI_LINK_GBS * 1e9) * 1e3
    h = HOST_MS_PER_CHUNK
    best = None
    table = {}
    for c in range(1, max(1, min(8, n_items)) + 1):
        t = max(compute_ms, c * h, max(compute_ms / (2 * c), h / 2) + G)
        table[c] = round(t, 4)
        if best is None or t < best[1] * 0.98:
            best = (c, t)
    return best[0], {"gather_floor_ms": round(G, 4), "compute_ms_assumed": round(compute_ms, 4), "host_ms_per_chunk": h,
                     "predicted_step_ms_by_chunks": table}


def predicted_scaling(ms_per_item, scaling, per_gpu, total, item_bytes, allreduce_ms=0.03, gather_every=(4, 8), channels=4):
    """What the 1/2/4/8-GPU curve must look like from the link arithmetic alone: whole-job speed-up over
    one GPU with the image all-gather inside the step (every rank receives (N-1) x its own image bytes, one peer per
    link: floor = own bytes / 153 GB/s, whatever N) and without it (independent ranks + one small all-reduce)."""
    from nvdiffrast_amd.parallel import payload_bytes_per_pixel
    fmts = ("f32", "f16", "rgba8", "rgb8")
    out = {"with_image_gather": {}, "without_image_gather": {}, **{"gather_every_%d" % k: {} for k in gather_every},
           # the per-step gather as the default line runs it: packed on the producing rank (one more streaming pass over the image:
           # 4 + b bytes per value at the copy rate), collected one step later -- a step takes max(kernels + pack, link time)
           "pipelined_per_step_gather": {f: {} for f in fmts},
           "assumes": "compute = %.4f ms per item (this run), "
           "xGMI link %.0f GB/s at 100 %% efficiency, %.0f us for the shared-gradient all-reduce" % (ms_per_item, XGMI_LINK_GBS, allreduce_ms * 1e3)}
    items1 = per_gpu if scaling == "weak" else total
    t1 = items1 * ms_per_item
    for n in (1, 2, 4, 8):
        items = per_gpu if scaling == "weak" else -(-total // n)
        job_items = items * n if scaling == "weak" else total
        C = items * ms_per_item
        G = items * item_bytes / (XGMI_LINK_GBS * 1e9) * 1e3 if n > 1 else 0.0
        ar = allreduce_ms if n > 1 else 0.0
        c, _ = plan_chunks(items, item_bytes, C, n - 1)
        t_g = max(C, c * HOST_MS_PER_CHUNK if n > 1 else 0.0, (max(C / (2 * c), HOST_MS_PER_CHUNK / 2) + G) if n > 1 else C) + ar
        t_n = C + ar
        for f in fmts:
            frac = payload_bytes_per_pixel(f, channels) / (4.0 * channels)
            pack = 0.0 if f == "f32" else items * item_bytes * (1 + frac) / (COPY_GBS * 1e9) * 1e3
            t_p = (max(C + pack, G * frac) + ar) if n > 1 else C
            out["pipelined_per_step_gather"][f][str(n)] = round((job_items / t_p) / (items1 / t1), 2)
        out["with_image_gather"][str(n)] = round((job_items / t_g) / (items1 / t1), 2)
        out["without_image_gather"][str(n)] = round((job_items / t_n) / (items1 / t1), 2)
        for k in gather_every:
            # the images of every k-th step only, their gather running behind the kernels of the k steps until the next one:
            # k steps take max(k steps of kernels, one gather + the step it starts in)
            t_k = max(k * t_n, (G + C / 2 + ar) if n > 1 else 0.0) / k
            out["gather_every_%d" % k][str(n)] = round((job_items / t_k) / (items1 / t1), 2)
    return out


class DryKernels:
    """Stand-in for the HIP path in --dry-run-cpu: same tensor shapes, trivial arithmetic, autograd intact."""

    def __init__(self, dev):
        self.dev = dev

    def forward(self, pos, attr, res, A):
        n = pos.shape[0]
        base = pos[:, :1, :1].reshape(n, 1, 1, 1) + attr.mean()
        return base.expand(n, res, res, A).contiguous()


class Job:
    """One workload on this rank: its tensors resident in device memory and the step the timed region repeats."""

    def __init__(self, name, N, total_items, first, rank, world, dev, dry, res, distributed, chunks, gather, gather_every=1,
                 gather_format="f32", gather_pipelined=False):
        import torch.distributed as dist
        from nvdiffrast_amd.parallel import broadcast_shared
        self.dist = dist
        self.name, self.wl = name, WORKLOADS[name]
        self.N, self.total_items, self.rank, self.world, self.dev, self.dry = N, total_items, rank, world, dev, dry
        self.RES = res
        self.A = self.wl["attrs"]
        self.full = self.wl["graph"] == "full"
        self.distributed, self.gather_on = distributed, gather
        # Per-item poses differ across ranks; geometry every item shares (tri, attr/uv, texture) comes from rank 0 over
        # RCCL, as it would in a data-parallel job.
        self.scene_name = self.wl.get("scene", "m10k")
        self.scene = build_scene(self.scene_name, N, self.A, dry, rank)
        self.gather_every = max(1, int(gather_every))
        self.gather_format = gather_format
        self.last_batch = None
        self.set_gather_mode(gather_format, gather_pipelined)
        self.step_index = 0
        self.in_flight = []          # image gathers of earlier steps that may still be running (gather_every > 1)
        self.literal = False         # step variant: SURVEY 8(d)'s literal `(out*G).sum().backward()`
        sc = self.scene
        self.pos = torch.from_numpy(sc["pos"]).to(dev).requires_grad_(True)
        self.tri = torch.from_numpy(sc["tri"]).to(dev)
        self.shared = torch.from_numpy(sc["uv"] if self.full else sc["attr"]).to(dev)
        self.C_out = 3 if self.full else self.A
        self.tex = self.tex_np = None
        if self.full:
            tex_res = 2048 if not dry else 32
            self.tex_np = np.random.default_rng(5).uniform(size=(1, tex_res, tex_res, 3)).astype(np.float32)
            self.tex = torch.from_numpy(self.tex_np).to(dev)
        if distributed:
            broadcast_shared([self.tri, self.shared] + ([self.tex] if self.full else []), src=0)
        self.shared.requires_grad_(True)
        if self.full:
            self.tex.requires_grad_(True)
        gen = torch.Generator(device=dev)
        gen.manual_seed(77 + rank)
        self.G = torch.randn((N, res, res, self.C_out), generator=gen, device=dev, dtype=torch.float32)
        if dry:
            self.impl = DryKernels(dev)
            self.dr = self.ctx = self.topo = self.lib = None
        else:
            import nvdiffrast_amd.torch as dr
            from nvdiffrast_amd import _capi
            self.dr, self._capi = dr, _capi
            self.lib = _capi.load()
            self.ctx = dr.RasterizeCudaContext(device=dev)
            self.topo = dr.antialias_construct_topology_hash(self.tri) if self.full else None
        self.set_chunks(chunks)
        self.last_rast = None

    def set_gather_mode(self, fmt, pipelined):
        """What the image all-gather carries and when its result is waited for (nvdiffrast_amd/parallel.py ImageGather)."""
        from nvdiffrast_amd.parallel import ImageGather
        self.gather_format = fmt
        self.pipe = None
        if pipelined and self.gather_every == 1:
            n_items = self.total_items if self.wl["total"] is not None else None       # (strong scaling: the split may be ragged)
            self.pipe = ImageGather(fmt, n_items=n_items, pipelined=True)

    def set_chunks(self, chunks):
        """The rank's items are rendered in this many calls (chunks exist to overlap a chunk's image all-gather with the next
        chunk's kernels; without the gather one call is best: larger launches, and the rasterizer's work order needs 2048 bins)."""
        self.chunks = max(1, min(int(chunks), self.N))
        self.bounds = [(self.N * c // self.chunks, self.N * (c + 1) // self.chunks) for c in range(self.chunks)]
        if not hasattr(self, "_gathered"):
            self._gathered = {}
        self.gathered = self._gathered.setdefault(self.chunks, [None] * self.chunks)      # receive buffers, reused every step

    def render(self, p):
        """Forward of the op graph for the items `p` [n,V,4] -> (output image [n,H,W,C], rast)."""
        dr, res = self.dr, self.RES
        if self.dry:
            return self.impl.forward(p, self.shared, res, self.C_out), None
        rast, rast_db = dr.rasterize(self.ctx, p, self.tri, (res, res))
        if not self.full:
            out, _ = dr.interpolate(self.shared, rast, self.tri)
            return out, rast
        uv, uv_da = dr.interpolate(self.shared, rast, self.tri, rast_db=rast_db, diff_attrs="all")
        col = dr.texture(self.tex, uv, uv_da, filter_mode="linear-mipmap-linear")
        return dr.antialias(col, rast, p, self.tri, topology_hash=self.topo), rast

    def step(self):
        from nvdiffrast_amd.parallel import allreduce_shared_grads, start_image_gather
        self.pos.grad = None
        self.shared.grad = None
        if self.full:
            self.tex.grad = None
        self.last_rast = None             # (kept for coverage() after the LAST step only: a training loop does not hold the previous step's rast
        #                                   while it renders the next; holding it cost 17 us per headline step, profiles/r05q_keep_rast_ab.log)
        # gather_every = k: the consumer of the complete batch of images (a logger, a discriminator on another rank ...) wants
        # it every k-th step only; that step's all-gather then has k steps' worth of kernels to hide behind -- it is waited for
        # when its receive buffers are needed again, k steps later (or at the end of the timed window)
        gather_now = self.gather_on and self.step_index % self.gather_every == 0
        self.step_index += 1
        if gather_now:
            for w, _keep in self.in_flight:
                w.wait()
            self.in_flight = []
        pending = []
        for c, (a, b) in enumerate(self.bounds):
            p = self.pos if self.chunks == 1 else self.pos[a:b]
            out, rast = self.render(p)
            if gather_now and self.pipe is not None:
                # packed on this rank, all-gathered on RCCL's own stream; collected at the end of the NEXT step
                self.pipe.submit(out.detach())
            elif gather_now:
                # the collective runs on RCCL's own stream, ordered after this chunk's kernels; the next chunk's
                # kernels are issued right away and overlap it
                h = start_image_gather(out.detach(), self.gather_format, recv=self.gathered[c])
                self.gathered[c] = h.recv
                pending.append((h, None))
            if self.literal:
                (out * (self.G if self.chunks == 1 else self.G[a:b])).sum().backward()
            else:
                torch.autograd.backward(out, self.G if self.chunks == 1 else self.G[a:b])
            self.last_rast = rast
        if self.distributed:
            allreduce_shared_grads([self.shared] + ([self.tex] if self.full else []))
        if gather_now and self.pipe is not None:
            self.last_batch = self.pipe.collect()         # the previous step's complete batch (its link time ran behind this step's kernels)
        if self.gather_every > 1:
            self.in_flight += pending
        else:
            for w, _keep in pending:
                w.wait()

    def drain(self):
        for w, _keep in self.in_flight:
            w.wait()
        self.in_flight = []
        if self.pipe is not None:
            left = self.pipe.drain()
            if left:
                self.last_batch = left[-1]

    def fence(self):
        if not self.dry:
            torch.cuda.synchronize()
        if self.distributed:
            self.dist.barrier()
        if not self.dry:
            torch.cuda.synchronize()

    def timed_windows(self, run, warmup, steps, windows):
        """`windows` windows of exactly `steps` steps, each bracketed by barrier + synchronize on both sides.
        Returns (host seconds per window, MAX over ranks; hipEvent milliseconds per window of this rank)."""
        for _ in range(warmup):
            run()
        host, evt = [], []
        for _ in range(windows):
            self.fence()
            if not self.dry:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            t0 = time.perf_counter()
            for _ in range(steps):
                run()
            self.drain()                                      # gathers still in flight belong to this window
            if not self.dry:
                e1.record()
            self.fence()
            el = time.perf_counter() - t0
            if self.distributed:
                tt = torch.tensor([el], dtype=torch.float64, device=self.dev)
                self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
                el = float(tt.item())
            host.append(el)
            evt.append(float(e0.elapsed_time(e1)) if not self.dry else el * 1e3)
        return host, evt

    # ---- everything below runs on rank 0 after the timed region -----------------------------------------------------

    def _measure_coverage(self):
        """Both coverage figures of the last step's rast, once; the tensor is let go afterwards (a 268 MB block that stays
        allocated moves every later allocation: the kernels' times depend on where their tensors lie, see profile_kernels)."""
        if self.last_rast is not None:
            c = (self.last_rast[..., 3] > 0).float()
            self._coverage = (round(float(c.mean().item()), 4),
                              round(float(torch.nn.functional.max_pool2d(c[:, None], 8, ceil_mode=True).mean().item()), 4))
            self.last_rast = None
        return getattr(self, "_coverage", (None, None))

    def coverage(self):
        """Fraction of the pixels of the last step that a triangle covers (kernels may skip upstream gradients elsewhere)."""
        return self._measure_coverage()[0]

    def tile_coverage(self):
        """Fraction of the 8x8-pixel tiles of the last step with a covered pixel: what the consumers of rast cannot skip."""
        return self._measure_coverage()[1]

    def profile_kernels(self, prof_steps, ms_per_step):
        """Per-kernel hipEvent timing inside the library (one chunk per launch) -> (kernels, roofline, path_frac)."""
        lib, _capi = self.lib, self._capi
        self._measure_coverage()                         # (... and lets the timed loop's last rast go)
        lib.nvdr_profile_reset()
        lib.nvdr_profile_enable(1)
        for _ in range(prof_steps):
            # the same allocation pattern as step(): nothing of the previous iteration alive when the next one allocates.  The
            # caching allocator then hands every tensor the block it had the step before; with one more 268 MB block in
            # circulation (a rast or an image held across iterations) interpolate's forward kernel runs 74 instead of 56 us on
            # the same inputs (profiles/r05q_keep_rast_ab.log).
            self.pos.grad = None; self.shared.grad = None
            if self.full:
                self.tex.grad = None
            out, _ = self.render(self.pos)
            torch.autograd.backward(out, self.G)
            del out, _
        torch.cuda.synchronize()
        prof = _capi.profile_read()
        lib.nvdr_profile_enable(0)
        lib.nvdr_profile_reset()
        P_rank = self.N * self.RES * self.RES
        alg, path_bytes = algorithmic_bytes(self.wl["graph"], P_rank, self.A, int(self.tri.shape[0]), self.N)
        req, req_path = required_bytes(self.wl["graph"], P_rank, self.A, int(self.tri.shape[0]), self.N, self.coverage(), self.tile_coverage())
        frac = lambda b, ms: None if b is None else round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)    # noqa: E731
        kernels = {}
        for name, (total_ms, launches) in prof.items():
            avg_ms = total_ms / max(launches, 1)
            b, rb = alg.get(name), req.get(name)
            kernels[name] = {"avg_ms": round(avg_ms, 4), "launches_per_step": launches / prof_steps,
                             "alg_bytes": b, "gbs": None if b is None else round(b / (avg_ms * 1e-3) / 1e9, 1),
                             "required_bytes": None if rb is None else int(rb), "frac_alg": frac(b, avg_ms), "frac_required": frac(rb, avg_ms)}
        # One op spread over several launches is judged as ONE pass: its algorithmic bytes against the sum of its launches
        # (the texture gradient with caller scratch = k_tex_grad_light + k_tex_grad + k_tex_grad_fold).
        for pname, members in KERNEL_PASSES.items():
            have = [m for m in members if m in kernels]
            if len(have) > 1:
                t = sum(kernels[m]["avg_ms"] * kernels[m]["launches_per_step"] for m in have)
                b, rb = alg.get(pname), req.get(pname)
                for m in have:
                    kernels[m]["alg_bytes"] = kernels[m]["gbs"] = kernels[m]["required_bytes"] = kernels[m]["frac_alg"] = kernels[m]["frac_required"] = None
                    kernels[m]["part_of"] = pname + "_pass"
                kernels[pname + "_pass"] = {"avg_ms": round(t, 4), "launches_per_step": 1.0, "alg_bytes": b, "launches": have,
                                            "gbs": None if b is None else round(b / (t * 1e-3) / 1e9, 1),
                                            "required_bytes": None if rb is None else int(rb), "frac_alg": frac(b, t), "frac_required": frac(rb, t)}
        # Dominant kernel (or pass) = the longest one (time per step), full stop.
        dominant = max((k for k in kernels if "part_of" not in kernels[k]),
                       key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches_per_step"])
        dk = kernels[dominant]
        traffic = source = None
        path_counter = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")          # PMC-derived HBM bytes/launch, builder session
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                ent = tj.get(self.name, {})
                if self.N == self.wl["per_gpu"] and ent:
                    counted = sum(v * kernels[k]["launches_per_step"] for k, v in ent.items() if k in kernels)
                    path_counter = round(counted / (max(ms_per_step, 1e-9) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                if self.N == self.wl["per_gpu"]:
                    members = kernels[dominant].get("launches")
                    traffic = ent.get(dominant) if not members else (sum(ent[m] for m in members) if all(m in ent for m in members) else None)
                    source = "profiles/traffic.json (%s): rocprofv3 PMC passes of a builder session, not measured in this run" % tj.get("_source", "builder session")
            except Exception:  # noqa: BLE001
                traffic = None
        sec = dk["avg_ms"] * 1e-3
        roofline = {"bound": "hbm", "kernel": dominant, "achieved": dk["gbs"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": None if dk["gbs"] is None else round(dk["gbs"] / HBM_PEAK_GBS, 4),
                    "frac_is": "algorithmic bytes (SURVEY 8(d)) / time / peak",
                    "traffic": traffic, "traffic_source": source if traffic else None,
                    "required_bytes_per_launch": dk.get("required_bytes"), "frac_required": dk.get("frac_required"),
                    "traffic_ratio": None if not (traffic and dk["alg_bytes"]) else round(traffic / dk["alg_bytes"], 3),
                    "hbm_frac_counter": None if not traffic else round(traffic / sec / 1e9 / HBM_PEAK_GBS, 4),
                    "kernel_avg_ms": dk["avg_ms"], "alg_bytes_per_launch": dk["alg_bytes"]}
        path_frac = {"alg": round((path_bytes / (max(ms_per_step, 1e-9) * 1e-3) / 1e9) / HBM_PEAK_GBS, 4),
                     "required": None if req_path is None else round((req_path / (max(ms_per_step, 1e-9) * 1e-3) / 1e9) / HBM_PEAK_GBS, 4),
                     "counter": path_counter}
        return kernels, roofline, path_frac

    def checker(self):
        import oracle
        from oracle import ref as oref
        return (oref, "reference (oracle/_ref)") if oref.available() else (oracle, "oracle")

    def parity(self):
        """This very workload (its inputs, its resolution) against the reference itself, a bounded number of items."""
        chk, chk_name = self.checker()
        dr, sc, RES, dev = self.dr, self.scene, self.RES, self.dev
        err = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())   # noqa: E731
        mag = lambda a: float(np.abs(a).max())                                                            # noqa: E731
        if not self.full:
            import oracle
            big = self.scene_name.startswith("t1m")
            if big:
                # a million triangles: ids and sampled barycentrics against the REFERENCE's own image of this very item, kept as a
                # fixture (tests/golden/t1m_reference.npz, made by tests/golden/make_t1m_fixture.py from oracle/_ref: 20 s of
                # emulation per scene); gradients against the C oracle, whose image of this item equals the fixture's bit for bit
                # (tests/test_reference_fixture.py)
                chk, chk_name = oracle, "oracle"
            # (the reference under its CPU shim renders ~1.9 Mpix/s: eight items of the benchmark mesh are a few seconds)
            # the headline workload itself: EVERY item of its batch against the reference (64 items = 16.8 Mpix: half a minute of
            # emulation); config 4's 256 items and the regimes: a sample
            ns = 1 if big else min((64 if self.name == "ch" else 8) if self.scene_name == "m10k" else 2, self.N)
            ro, _ = chk.rasterize(sc["pos"][:ns], sc["tri"], (RES, RES))
            Gs = self.G[:ns].cpu().numpy()
            ga_o, gr_o, _ = chk.interpolate_grad(sc["attr"], ro, sc["tri"], Gs)
            gp_o = chk.rasterize_grad(sc["pos"][:ns], sc["tri"], ro, gr_o)
            pos_s = torch.from_numpy(sc["pos"][:ns]).to(dev).requires_grad_(True)
            attr_s = torch.from_numpy(sc["attr"]).to(dev).requires_grad_(True)
            r_s, _ = dr.rasterize(self.ctx, pos_s, self.tri, (RES, RES))
            o_s, _ = dr.interpolate(attr_s, r_s, self.tri)
            torch.autograd.backward(o_s, self.G[:ns])
            r_h = r_s.detach().cpu().numpy()
            res = {"against": chk_name, "items": ns, "resolution": [RES, RES], "bar": PARITY_BAR,
                   "tri_id_mismatches": int((r_h[..., 3] != ro[..., 3]).sum()),
                   "bary_max_abs_err": err(r_h[..., :3], ro[..., :3]),
                   "g_attr_max_abs_err": err(attr_s.grad.cpu().numpy(), ga_o), "g_attr_max_abs": mag(ga_o),
                   "g_pos_max_abs_err": err(pos_s.grad.cpu().numpy(), gp_o), "g_pos_max_abs": mag(gp_o)}
            if big:
                fx_path = os.path.join(ROOT, "tests", "golden", "t1m_reference.npz")
                if os.path.exists(fx_path) and RES == 1024:
                    import hashlib
                    fx = np.load(fx_path)
                    ids_h = np.ascontiguousarray(r_h[:1, ..., 3]).astype(np.uint32).astype("<u4")
                    yx = fx[self.scene_name + "/sample_yx"]
                    want = fx[self.scene_name + "/sample_rast"]
                    res["reference_fixture"] = {
                        "file": "tests/golden/t1m_reference.npz", "items": 1,
                        "ids_sha256_equal": hashlib.sha256(ids_h.tobytes()).hexdigest() == bytes(fx[self.scene_name + "/ids_sha256"]).decode(),
                        "covered_pixels": [int((ids_h > 0).sum()), int(fx[self.scene_name + "/covered"])],
                        "sampled_pixels": int(len(yx)), "sampled_bary_max_abs_err": err(r_h[0, yx[:, 0], yx[:, 1], :3], want[:, :3]),
                        "sampled_id_mismatches": int((r_h[0, yx[:, 0], yx[:, 1], 3] != want[:, 3]).sum())}
                    if res["reference_fixture"]["ids_sha256_equal"] and res["reference_fixture"]["sampled_id_mismatches"] == 0:
                        res["against"] = "reference fixture (ids: sha-256 of item 0's id image; 256 sampled barycentrics) + oracle (gradients)"
            if self.scene_name == "m10k" and self.N <= 64:
                # ALL items of the batch against the C oracle (the reference sums attr's gradient over every item: the shared-
                # attribute gradient of the full batch is the one number a two-item check cannot vouch for)
                ra, _ = oracle.rasterize(sc["pos"], sc["tri"], (RES, RES))
                ga_a, gr_a, _ = oracle.interpolate_grad(sc["attr"], ra, sc["tri"], self.G.cpu().numpy())
                gp_a = oracle.rasterize_grad(sc["pos"], sc["tri"], ra, gr_a)
                self.pos.grad = None; self.shared.grad = None
                out, rast = self.render(self.pos)
                torch.autograd.backward(out, self.G)
                res["all_items"] = {"against": "oracle", "items": self.N,
                                    "tri_id_mismatches": int((rast.detach()[..., 3].cpu().numpy() != ra[..., 3]).sum()),
                                    "g_attr_max_abs_err": err(self.shared.grad.cpu().numpy(), ga_a), "g_attr_max_abs": mag(ga_a),
                                    "g_pos_max_abs_err": err(self.pos.grad.cpu().numpy(), gp_a), "g_pos_max_abs": mag(gp_a)}
            return res
        # four-op chain, one item at the config's own resolution and texture size: every op against the reference ON THE INPUTS
        # THE HIP PATH GAVE IT (oracle/chain.py explains why a chain through a texture is not judged end to end)
        from oracle.chain import four_op_chain
        # (two items: g_tex's background texel collects 6e5 f32 terms PER ITEM in the reference's own atomics, whose rounding -- not
        # this library's fixed-point sums -- is what the comparison then measures: 1.9e-5 of |g| with two items, 3.8e-5 with four)
        nc = min(2, self.N)
        res = four_op_chain(dr, self.ctx, self.topo, chk, sc["pos"][:nc], sc["tri"], sc["uv"], self.tex_np, self.G[:nc], (RES, RES), dev=dev)
        res.update({"against": chk_name, "items": nc, "resolution": [RES, RES], "texture": list(self.tex_np.shape[1:3]), "bar": PARITY_BAR,
                    "compared": "each op on identical inputs (the HIP path's own intermediate tensors)"})
        return res

    def cpu_baselines(self, cpu_items):
        """The CPU oracle (and the reference under its CPU shim) on this host's cores, bounded samples of this batch."""
        import oracle
        from oracle import ref as oref
        sc, RES, N = self.scene, self.RES, self.N

        def chain(mod, n, Gc):
            pc, tc = sc["pos"][:n], sc["tri"]
            if not self.full:
                r_c, _ = mod.rasterize(pc, tc, (RES, RES))
                mod.interpolate(sc["attr"], r_c, tc)
                _ga, gr, _ = mod.interpolate_grad(sc["attr"], r_c, tc, Gc)
                mod.rasterize_grad(pc, tc, r_c, gr)
                return
            r, rdb = mod.rasterize(pc, tc, (RES, RES))
            uv_r, uvda_r = mod.interpolate(sc["uv"], r, tc, rdb, "all")
            col_r = mod.texture(self.tex_np, uv_r, uvda_r, filter_mode="linear-mipmap-linear")
            mod.antialias(col_r, r, pc, tc)
            g_col, _gp = mod.antialias_grad(col_r, r, pc, tc, Gc)
            g = mod.texture_grad(self.tex_np, uv_r, g_col, uvda_r, filter_mode="linear-mipmap-linear")
            _ga, g_rast, g_rdb = mod.interpolate_grad(sc["uv"], r, tc, g["uv"], rdb, g["uv_da"], "all")
            mod.rasterize_grad(pc, tc, r, g_rast, g_rdb)

        nc = max(1, min(cpu_items, N))
        Gc = self.G[:nc].cpu().numpy()
        reps = 5 if not self.full else 3
        times = []
        for _rep in range(reps):
            t1 = time.perf_counter()
            chain(oracle, nc, Gc)
            times.append(time.perf_counter() - t1)
        tmed = sorted(times)[len(times) // 2]
        try:
            affinity = len(os.sched_getaffinity(0))
        except Exception:  # noqa: BLE001
            affinity = None
        threads = oracle.num_threads()
        cpu = {"value": round(nc * RES * RES / tmed / 1e6, 2), "unit": "Mpixels/s", "cores": threads, "kind": "port", "cpu": cpu_model(),
               "sample": f"{nc} of {N} items, fwd+bwd, median of {reps}",
               "note": f"the reference has no CPU path; this is the repo's C/OpenMP restatement, pinned to the reference by the tests. "
                       f"cores = the OpenMP team it ran with, omp_get_max_threads() = {threads}: OpenMP sizes its default team from the "
                       f"CPUs this process may run on (affinity mask: {affinity}), not from os.cpu_count() = {os.cpu_count()} logical CPUs"}
        cpu_ref = None
        if oref.available():
            nr = min(4 if not self.full else 1, N)
            t1 = time.perf_counter()
            chain(oref, nr, self.G[:nr].cpu().numpy())
            tr = time.perf_counter() - t1
            cpu_ref = {"value": round(nr * RES * RES / tr / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": "reference", "cpu": cpu_model(),
                       "sample": f"{nr} item(s), fwd+bwd, one run",
                       "note": "the reference's own CUDA kernels and CudaRaster compiled for the host and executed by a fibre-based "
                               "CUDA-on-CPU shim (oracle/refshim) -- an emulation on one thread, not a tuned CPU implementation"}
        return cpu, cpu_ref


def window_stats(host_s, evt_ms, steps):
    per = sorted(h / steps * 1e3 for h in host_s)
    med = per[len(per) // 2] if len(per) % 2 else 0.5 * (per[len(per) // 2 - 1] + per[len(per) // 2])
    ev = sorted(e / steps for e in evt_ms)
    return med, {"windows": len(per), "steps_per_window": steps, "ms_per_step_median": round(med, 4),
                 "ms_per_step_min": round(per[0], 4), "ms_per_step_max": round(per[-1], 4),
                 "ms_per_step_windows": [round(h / steps * 1e3, 4) for h in host_s],
                 "hip_event_ms_per_step_median": round(ev[len(ev) // 2], 4),
                 "clock": "host perf_counter around barrier+synchronize brackets (max over ranks); hip_event_* = torch.cuda.Event pair "
                          "on the launch stream around the same steps"}


def graph_replay_in_child(argv, keys, script=None):
    """The same step replayed from ONE hipGraph (nothing on the path allocates at the C-ABI level or synchronises the host, so
    the whole step captures): what the kernels alone take when the host's launch work is out of the way.  Measured by a child
    process running this file's --graph path: a capture that goes wrong inside the runtime takes the process with it, and
    the bench line must not depend on that."""
    import subprocess
    cmd = [sys.executable, script or os.path.abspath(__file__)] + list(argv) + (["--graph"] if "--graph" not in argv else [])
    if script is None:
        cmd += ["--no-cpu-baseline", "--no-extra-configs"]
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
        if p.returncode != 0:
            return {"error": "child exited with %d: %s" % (p.returncode, p.stderr.decode(errors="replace").strip().splitlines()[-1:] or "")}
        doc = json.loads(p.stdout.decode().strip().splitlines()[-1])
        return {k: doc[k] for k in keys}
    except Exception as e:  # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, e)}


def extra_config(name, dev, steps, warmup, windows, with_cpu, graph_too=False):
    """One more BASELINE config measured in this process (single GPU): the block that goes under `configs`."""
    wl = WORKLOADS[name]
    N = wl["per_gpu"] or wl["total"]                    # (c4: configs[3]'s 256 items on ONE GPU, the N = 1 point of its strong-scaling curve)
    job = Job(name, N, N, 0, 0, 1, dev, False, wl["res"], False, 1, False)
    host, evt = job.timed_windows(job.step, warmup, steps, windows)
    ms, timing = window_stats(host, evt, steps)
    kernels, roofline, path_frac = job.profile_kernels(max(3, min(steps, 5)), ms)
    P = N * wl["res"] * wl["res"]
    block = {"metric": wl["metric"], "value": round(P / (ms * 1e-3) / 1e6, 1), "unit": "Mpixels/s", "ms_per_step": round(ms, 4),
             "steps": steps, "warmup": warmup, "timing": timing, "batch": N, "resolution": [wl["res"], wl["res"]],
             "triangles": int(job.tri.shape[0]), "coverage": job.coverage(), "tile_coverage": job.tile_coverage(), "launch": "eager",
             "roofline": roofline, "path_hbm_frac": path_frac["alg"], "path_hbm_frac_required": path_frac["required"],
             "path_hbm_frac_counter": path_frac["counter"], "kernels": kernels, "parity": job.parity()}
    # the rasterizer's scratch policy for this batch (include/nvdr_hip.h NVDR_OPT_SCRATCH_LIMIT_MB): worst case = no host
    # synchronisation anywhere in the step; adaptive = a clip pool that grows on demand, one counter read back per call
    worst = int(job.lib.nvdr_rasterize_scratch_bytes(N, int(job.tri.shape[0]), wl["res"], wl["res"]))
    limit = int(job.lib.nvdr_get_option(job._capi.OPT_SCRATCH_LIMIT_MB))
    block["rasterizer_scratch"] = {"worst_case_mb": worst >> 20, "limit_mb": limit, "adaptive_pool": worst > (limit << 20)}
    if with_cpu:
        cpu, cpu_ref = job.cpu_baselines(4 if job.full else 16)
        block["cpu_baseline"], block["cpu_reference"] = cpu, cpu_ref
    if graph_too:
        block["hipgraph_replay"] = graph_replay_in_child(["--workload", name, "--steps", str(steps), "--warmup", str(warmup),
                                                          "--windows", str(windows)], ("ms_per_step", "value"))
    del job
    torch.cuda.empty_cache()
    return block


def c5_standin():
    """BASELINE configs[4]: the reference's earth.py needs a release asset (earth.npz) that no checkout holds; the same op
    graph and sizes (2048^2 reference render, 512^2 candidate, 2048^2 texture, Adam, 200 iterations) on a procedural globe."""
    sys.path.insert(0, os.path.join(ROOT, "samples"))
    import fit_texture_synth
    fit_texture_synth.fit(iters=10, res=512, ref_res=2048, tex_size=2048)                       # warm-up: allocations, scratch
    r = fit_texture_synth.fit(iters=200, res=512, ref_res=2048, tex_size=2048)
    r["what"] = "samples/fit_texture_synth.py: all four ops fwd+bwd + Adam per iteration, eager launching; stands in for samples/torch/earth.py"
    r["hipgraph_replay"] = graph_replay_in_child(["--iters", "200", "--res", "512", "--ref-res", "2048", "--tex", "2048", "--graph"],
                                                 ("iters_per_s", "loss_last"), script=os.path.join(ROOT, "samples", "fit_texture_synth.py"))
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--windows", type=int, default=5, help="timing windows of --steps steps each; the median window is reported")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="ch")
    ap.add_argument("--batch", type=int, default=None, help="items per GPU (overrides the workload's)")
    ap.add_argument("--res", type=int, default=None, help="resolution (overrides the workload's; dry runs use a small one)")
    ap.add_argument("--chunks", type=int, default=None,
                    help="N > 1: the rank's items are rendered in this many chunks so that a chunk's image all-gather "
                         "overlaps the next chunk's kernels (default: from the link / host arithmetic, see plan_chunks; 1 at N = 1)")
    ap.add_argument("--no-gather-images", action="store_true", help="N > 1: keep the output images sharded (no all-gather in the step)")
    ap.add_argument("--gather-every", type=int, default=1,
                    help="N > 1: all-gather the output images of every k-th step only, overlapped with the following steps (default 1 = "
                         "every step, the number the metric is quoted on; k = 4 is the schedule the link arithmetic predicts to scale >= 6x)")
    ap.add_argument("--gather-format", choices=["f32", "f16", "rgba8", "rgb8"], default="rgb8",
                    help="N > 1: what the per-step image all-gather carries (nvdiffrast_amd/parallel.py): f32 as rendered, f16, rgba8 = every "
                         "channel as unorm8, rgb8 = the first three channels as unorm8 (default: the per-step gather the xGMI links can carry)")
    ap.add_argument("--no-gather-pipeline", action="store_true",
                    help="N > 1: wait for a step's images inside that step (default: collect them at the end of the NEXT step, so the "
                         "link time hides behind its kernels)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="default run: do not also measure BASELINE configs[1], [2], [4]")
    ap.add_argument("--cpu-items", type=int, default=64, help="items in the CPU-oracle sample")
    ap.add_argument("--graph", action="store_true",
                    help="capture the step into one hipGraph and time replays (single GPU only; the default, and the "
                         "number the driver records, is eager launching)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="run the N > 1 code path (process group, broadcast, chunked all-gather, all-reduce) even with one rank: "
                         "checks the RCCL calls on a 1-GPU box")
    ap.add_argument("--detail", default=None, help="where the complete record goes (default: bench_detail.json next to this file)")
    ap.add_argument("--dry-run-cpu", action="store_true", help="gloo + stand-in kernels: exercises the multi-rank plumbing without GPUs")
    args = ap.parse_args()

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        respawn_under_launcher(args)                             # does not return
    # The JSON line must be the only thing on stdout.  Native libraries write there too (RCCL prints a version banner from C
    # stdio, flushed at exit, i.e. AFTER the line): everything else this process writes to fd 1 goes to stderr instead.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if launched and world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    distributed = world > 1 or args.force_collectives
    dry = args.dry_run_cpu
    if distributed and not launched:                              # --force-collectives without a launcher: a group of one
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")

    import torch.distributed as dist
    if dry:
        dev = torch.device("cpu")
        if distributed:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo")
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if distributed:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="nccl", device_id=dev)     # RCCL over xGMI

    from nvdiffrast_amd import parallel
    from nvdiffrast_amd.parallel import shard_range
    parallel.force_collectives(args.force_collectives)

    wl = WORKLOADS[args.workload]
    RES = args.res or (32 if dry else wl["res"])
    if wl["total"] is not None:                                   # strong scaling: a fixed batch split over the ranks
        total_items = wl["total"] if args.batch is None else args.batch * world
        first, N = shard_range(total_items, world, rank)
    else:
        N = args.batch or wl["per_gpu"]
        total_items, first = N * world, N * rank
    gather = distributed and not args.no_gather_images
    C_out = 3 if wl["graph"] == "full" else wl["attrs"]
    item_bytes = RES * RES * C_out * 4
    # chunk policy: from the bytes one link has to carry against the rank's predicted compute and the host's launch cost
    # (plan_chunks); the per-item compute estimate is the single-GPU measurement of this tree, refined below
    est_ms_per_item = (0.0080 if wl["graph"] == "ri" else 0.105) * (RES * RES) / (wl["res"] * wl["res"])
    n_max = -(-total_items // world)
    chunks, chunk_plan = plan_chunks(n_max, item_bytes, n_max * est_ms_per_item, (world - 1) if gather else 0)
    chunk_plan_chunks = chunks                                     # (for the f32-in-step comparison leg)
    if args.chunks:
        chunks = args.chunks
    if not gather:
        chunks = args.chunks or 1
    # The default exchange: packed images (--gather-format), collected one step later -- the whole next step to hide behind, so
    # the rank's items are rendered in one call.  --no-gather-pipeline / --gather-every k keep the chunked overlap inside the step.
    # An explicit --chunks > 1 asks for the overlap inside the step, i.e. no pipelining.
    pipelined = gather and not args.no_gather_pipeline and args.gather_every == 1 and not (args.chunks and args.chunks > 1)
    if pipelined:
        chunks = 1

    job = Job(args.workload, N, total_items, first, rank, world, dev, dry, RES, distributed, chunks, gather, args.gather_every,
              gather_format=args.gather_format if gather else "f32", gather_pipelined=pipelined)

    run = job.step
    if args.graph:
        # No op of the path synchronises the host or allocates at the C-ABI level, so the whole step captures.
        assert not distributed and not dry, "--graph is a single-GPU measurement"
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                job.step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            job.step()
        run = graph.replay

    host, evt = job.timed_windows(run, args.warmup, args.steps, max(1, args.windows))
    ms_per_step, timing = window_stats(host, evt, args.steps)
    ranks_seen = world
    if distributed:
        ones = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)               # every rank took part in the timed region
        ranks_seen = int(round(float(ones.item())))
        assert ranks_seen == dist.get_world_size()
    # The image all-gather is pure xGMI traffic whose size does not depend on how fast the kernels are: report the
    # same job without it next to `value`, and the link arithmetic, so that the two effects can be told apart.
    collective = None
    if distributed:
        from nvdiffrast_amd.parallel import payload_bytes_per_pixel, payload_channels
        ms_ng = ms_per_step
        ms_f32 = None
        if gather:
            planned = job.chunks
            job.drain()
            job.gather_on = False
            job.set_chunks(1)                                     # (nothing to overlap: the rank's items in one call)
            host_ng, evt_ng = job.timed_windows(run, min(args.warmup, 2), args.steps, max(1, min(args.windows, 3)))
            job.gather_on = True
            ms_ng, _ = window_stats(host_ng, evt_ng, args.steps)
            if args.gather_format != "f32" or pipelined:
                # ... and the same job with the images gathered as rendered (f32), waited for inside the step, chunked to overlap:
                # what the default line of rounds 2-5 measured
                job.set_gather_mode("f32", False)
                job.set_chunks(chunk_plan_chunks)
                host_f, evt_f = job.timed_windows(run, min(args.warmup, 2), args.steps, max(1, min(args.windows, 3)))
                ms_f32, _ = window_stats(host_f, evt_f, args.steps)
                job.drain()
                job.set_gather_mode(args.gather_format, pipelined)
            job.set_chunks(planned)
        fmt = args.gather_format if gather else "f32"
        img_bytes = N * RES * RES * payload_bytes_per_pixel(fmt, C_out)
        links = world - 1                                         # fully connected xGMI: one link per peer (0: a forced group of one)
        collective = {
            "image_gather_in_step": bool(gather),
            "gather_format": fmt if gather else None,
            "gather_payload": ("%d of %d channels, %d byte(s) each" % (payload_channels(fmt, C_out), C_out, payload_bytes_per_pixel(fmt, C_out) // payload_channels(fmt, C_out))) if gather else None,
            "gather_pipelined": bool(pipelined),                  # a step's images are collected at the end of the next step
            "image_bytes_sent_per_rank_per_step": img_bytes * links if gather else 0,
            "image_bytes_received_per_rank_per_step": img_bytes * links if gather else 0,
            "ms_per_step_with_f32_gather_in_step": None if ms_f32 is None else round(ms_f32, 4),
            "value_with_f32_gather_in_step": None if ms_f32 is None else round(total_items * RES * RES / (ms_f32 * 1e-3) / 1e6, 1),
            "xgmi_links_per_gpu_used": links, "xgmi_link_gbs": XGMI_LINK_GBS,
            # every link carries one rank's images, all links in parallel: the floor does not shrink with more GPUs
            "xgmi_floor_ms": round(img_bytes / (XGMI_LINK_GBS * 1e9) * 1e3, 4) if links else None,
            "ms_per_step_without_image_gather": round(ms_ng, 4),
            "value_without_image_gather": round(total_items * RES * RES / (ms_ng * 1e-3) / 1e6, 1),
            "chunks": job.chunks, "chunk_plan": chunk_plan, "gather_every": job.gather_every,
            # the curve the link arithmetic predicts, from THIS run's compute time per item (gather-free step / items)
            "predicted_scaling": predicted_scaling(ms_ng / max(N, 1), wl["scaling"], wl["per_gpu"] or N, wl["total"] or total_items, item_bytes,
                                                   channels=C_out),
        }

    P_total = total_items * RES * RES
    value = P_total / (ms_per_step * 1e-3) / 1e6                  # whole-job Mpixels/s

    result = None
    if rank == 0:
        kernels = roofline = parity = cpu = cpu_ref = configs = literal = None
        path_frac = {"alg": None, "required": None, "counter": None}
        coverage = tile_cov = None
        if not dry:
            coverage, tile_cov = job.coverage(), job.tile_coverage()
            kernels, roofline, path_frac = job.profile_kernels(max(3, min(args.steps, 10)), ms_per_step)
            if world == 1:
                parity = job.parity()
                if not args.no_cpu_baseline:
                    cpu, cpu_ref = job.cpu_baselines(args.cpu_items if not job.full else min(args.cpu_items, 4))
            if world == 1 and not distributed and not args.graph:
                # SURVEY 8(d)'s literal timed region, `(out * G).sum().backward()`: the same library work plus torch's
                # element-wise multiply, reduction and their backward (three more passes over the output image)
                job.literal = True
                host_l, evt_l = job.timed_windows(job.step, 2, args.steps, max(1, min(args.windows, 3)))
                job.literal = False
                ms_l, _ = window_stats(host_l, evt_l, args.steps)
                literal = {"ms_per_step": round(ms_l, 4), "value": round(P_total / (ms_l * 1e-3) / 1e6, 1),
                           "what": "(out*G).sum().backward() instead of torch.autograd.backward(out, G)"}
        full = job.full
        scene_txt = {"m10k": "random-pose 10k-triangle lattice mesh", "dense": "10k-triangle lattice mesh, camera pulled in",
                     "s10k": "S10k stress triangles", "t1m": "1M-triangle lattice mesh, index order",
                     "t1m_shuffled": "1M-triangle lattice mesh, tri rows shuffled"}[job.scene_name]
        cfg = {"workload": "%s: %s (T=%d, V=%d), %d items on this GPU of %d in total @%dx%d, %s, "
                           "upstream grad fed to backward directly"
                           % (args.workload.upper(), scene_txt, int(job.tri.shape[0]), int(job.pos.shape[1]), N, total_items, RES, RES,
                              "rasterize+interpolate(uv,da)+texture(2048^2 trilinear)+antialias fwd+bwd" if full
                              else "A=%d attrs, rasterize+interpolate fwd+bwd" % job.A),
               "batch_per_gpu": N, "total_items": total_items, "resolution": [RES, RES], "triangles": int(job.tri.shape[0]),
               "coverage": coverage, "tile_coverage": tile_cov,
               "parallelism": "dp%d (items sharded; per step: %sall-reduce of the shared-input gradients)"
                              % (world, (("all-gather of the output images as %s, collected one step later, " % job.gather_format) if job.pipe is not None
                                         else "all-gather of the output images%s as %s in %d chunks overlapped with rendering, "
                                         % (" of every %d-th step" % job.gather_every if job.gather_every > 1 else "", job.gather_format, job.chunks)) if gather else ""),
               "chunks": job.chunks, "gather_images": bool(gather), "gather_every": job.gather_every,
               "gather_format": job.gather_format if gather else None, "gather_pipelined": job.pipe is not None,
               "launch": "hipGraph replay" if args.graph else "eager"}
        # ---- the other BASELINE configs and the regimes the benchmark scene hides: same process, single GPU, default run only ---
        if (not dry and world == 1 and not distributed and args.workload == "ch" and args.batch is None and args.res is None
                and not args.graph and not args.no_extra_configs):
            del job
            torch.cuda.empty_cache()
            configs = {}
            for name, st in (("c2", 20), ("c3", 6), ("c4", 6), ("dense", 10), ("s10k", 6), ("t1m", 6), ("t1m_shuffled", 6)):
                try:
                    configs[name] = extra_config(name, dev, st, 3, 5 if name in ("c2", "c3") else 3,
                                                 with_cpu=(name == "c3" and not args.no_cpu_baseline), graph_too=(name == "c2"))
                except Exception as e:  # noqa: BLE001
                    configs[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                configs["c5_standin"] = c5_standin()
            except Exception as e:  # noqa: BLE001
                configs["c5_standin"] = {"error": "%s: %s" % (type(e).__name__, e)}
        result = {
            "metric": wl["metric"],
            "value": round(value, 1), "unit": "Mpixels/s",
            "n_gpus": world, "rccl_ranks": ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "ms_per_step_min": timing["ms_per_step_min"], "ms_per_step_max": timing["ms_per_step_max"],
            "higher_is_better": True, "scaling": wl["scaling"],
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": cfg,
            "timing": timing,
            "roofline": roofline,
            "path_hbm_frac": path_frac["alg"], "path_hbm_frac_required": path_frac["required"], "path_hbm_frac_counter": path_frac["counter"],
            "literal_step": literal,
            "kernels": kernels,
            "cpu_baseline": cpu,
            "cpu_reference": cpu_ref,
            "parity": parity,
            "collective": collective,
            "configs": configs,
        }
        if dry:
            result["dry_run"] = True
            result["backend"] = "gloo" if distributed else "none"
            if gather and job.pipe is not None:
                result["gathered_rows_total"] = int(job.last_batch.shape[0])          # the last step's complete batch, as collected
                result["gathered_payload"] = [str(job.last_batch.dtype).replace("torch.", ""), int(job.last_batch.shape[-1])]
            elif gather:
                g0 = job.gathered[0]
                result["gathered_rows_chunk0"] = int(g0.shape[0])
                result["gathered_rows_total"] = int(sum(g.shape[0] for g in job.gathered))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # (a CPU dry run -- the test suite's -- leaves no record next to this file unless asked for one with --detail)
        detail = args.detail or (None if args.dry_run_cpu else os.path.join(ROOT, "bench_detail.json"))
        if detail:
            try:
                with open(detail, "w") as f:
                    json.dump(result, f, indent=1)
            except OSError:
                detail = None
        line = compact_line(result, detail)
        os.write(json_fd, (line + "\n").encode())
    os.close(json_fd)


LINE_LIMIT = 7600          # bytes; the driver keeps the last 8 KB of stdout


def _r(x, n=4):
    return None if x is None else (round(x, n) if isinstance(x, float) else x)


def _sig(x, n=3):
    return None if x is None else float("%.*g" % (n, x))


def compact_parity(p):
    if not p:
        return p
    if "error" in p:
        return p
    out = {"vs": "ref-fixture" if "fixture" in p.get("against", "") else "ref" if "reference" in p.get("against", "") else "oracle",
           "items": p.get("items"), "ids": p.get("tri_id_mismatches"), "bary": _sig(p.get("bary_max_abs_err"))}
    if "reference_fixture" in p:
        f = p["reference_fixture"]
        out["fx"] = [int(f["ids_sha256_equal"]), f["sampled_id_mismatches"], _sig(f["sampled_bary_max_abs_err"])]     # sha equal, sampled ids, sampled bary
    for k in ("g_attr", "g_pos"):
        if k + "_max_abs_err" in p:
            out[k] = [_sig(p[k + "_max_abs_err"]), _sig(p[k + "_max_abs"])]
    if "all_items" in p:
        a = p["all_items"]
        out["all"] = {"vs": "oracle", "items": a["items"], "ids": a["tri_id_mismatches"],
                      "g_attr": [_sig(a["g_attr_max_abs_err"]), _sig(a["g_attr_max_abs"])], "g_pos": [_sig(a["g_pos_max_abs_err"]), _sig(a["g_pos_max_abs"])]}
    chain = {}
    for k in ("uv", "col", "aa"):                                  # four-op chain: forward errors, then [err, magnitude] of each gradient
        if k + "_err" in p:
            chain[k] = _sig(p[k + "_err"])
    for k in ("rast_db", "uv_da", "g_col", "g_tex", "g_uv", "g_uv_da", "g_uvattr", "g_rast", "g_rast_db", "g_pos"):
        if k + "_err" in p:
            chain[k] = [_sig(p[k + "_err"]), _sig(p[k + "_max"])]
    if chain:
        out["ops"] = chain
    return out


def compact_kernels(kernels):
    """{name: [avg ms per launch, algorithmic fraction of peak, required fraction of peak]} (passes included, members without fractions)."""
    if not kernels:
        return kernels
    # [ms, algorithmic fraction of the HBM peak, required fraction].  A kernel that skips the reads of empty tiles can show an
    # ALGORITHMIC fraction above 1 (the section 8(d) convention credits it with bytes it never moved): such a figure is not a
    # bandwidth and is never printed -- the cell then carries null and the required fraction, which is the physical statement.
    def cell(v):
        fa, fr = v.get("frac_alg"), v.get("frac_required")
        if fa is not None and fa > 1.0:
            assert fr is not None, "an algorithmic fraction above 1 needs the required fraction next to it"
            fa = None
        return [_r(v["avg_ms"]), _r(fa, 3), _r(fr, 3)]
    return {k: cell(v) for k, v in kernels.items()}


def compact_config(c):
    if "error" in c:
        return {"error": str(c["error"])[:160]}
    if "iters_per_s" in c:                                          # the configs[4] stand-in
        out = {"it_s": _r(c["iters_per_s"], 1), "loss": [_sig(c.get("loss_first")), _sig(c.get("loss_last"))]}
        g = c.get("hipgraph_replay") or {}
        out["graph_it_s"] = _r(g.get("iters_per_s"), 1) if "error" not in g else "error"
        return out
    rf = c["roofline"]
    out = {"ms": c["ms_per_step"], "gpix": round(c["value"] / 1e3, 2), "cov": c.get("coverage"), "tcov": c.get("tile_coverage"),
           "k": {k: _r(v["avg_ms"]) for k, v in c["kernels"].items() if "part_of" not in v or True},
           "dom": [rf["kernel"], _r(rf["frac"], 3), _r(rf.get("frac_required"), 3)],
           "path": [_r(c.get("path_hbm_frac"), 3), _r(c.get("path_hbm_frac_required"), 3), _r(c.get("path_hbm_frac_counter"), 3)],
           "par": compact_parity(c.get("parity"))}
    if (c.get("rasterizer_scratch") or {}).get("adaptive_pool") is not None and c.get("batch", 0) >= 256:
        out["adaptive_pool"] = c["rasterizer_scratch"]["adaptive_pool"]          # config 4's anchor: false = no host sync in its step
    g = c.get("hipgraph_replay")
    if g:
        out["graph_ms"] = g.get("ms_per_step", "error")
    if c.get("cpu_baseline"):
        out["cpu"] = [c["cpu_baseline"]["value"], c["cpu_baseline"]["cores"]]
    return out


def compact_line(full, detail_path):
    """The record the driver keeps: every contract key, numbers instead of prose, under LINE_LIMIT bytes."""
    if full is None:
        return json.dumps(None)
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step", "ms_per_step_min",
                                "ms_per_step_max", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = full["config"]
    t = full["timing"]
    out["timing"] = {"windows": t["windows"], "ms_per_step_windows": t["ms_per_step_windows"], "hip_event_ms_per_step_median": t["hip_event_ms_per_step_median"]}
    rf = full["roofline"]
    if rf:
        rf = dict(rf)
        rf.pop("frac_is", None)
        if rf.get("traffic_source"):
            rf["traffic_source"] = "profiles/traffic.json (builder-session PMC passes, not this run)"
    out["roofline"] = rf
    for k in ("path_hbm_frac", "path_hbm_frac_required", "path_hbm_frac_counter"):
        out[k] = full.get(k)
    if full.get("literal_step"):
        out["value_literal_step"] = full["literal_step"]["value"]
        out["ms_literal_step"] = full["literal_step"]["ms_per_step"]
    out["kernels"] = compact_kernels(full["kernels"])
    for k in ("cpu_baseline", "cpu_reference"):
        c = full.get(k)
        out[k] = None if not c else {kk: c[kk] for kk in ("value", "unit", "cores", "kind", "cpu", "sample") if kk in c}
    p = full.get("parity")
    out["parity"] = None if not p else dict(compact_parity(p), bar=PARITY_BAR)
    out["collective"] = full.get("collective")
    cf = full.get("configs")
    out["configs"] = None if cf is None else {k: compact_config(v) for k, v in cf.items()}
    for k in ("dry_run", "backend", "gathered_rows_chunk0", "gathered_rows_total", "gathered_payload"):
        if k in full:
            out[k] = full[k]
    out["detail"] = None if not detail_path else os.path.basename(detail_path)
    line = json.dumps(out, separators=(",", ":"))
    # never over the limit: shed the least important parts first (they remain in the detail file)
    for shed in (("collective", "chunk_plan"), ("collective", "predicted_scaling", "assumes"), ("config", "workload"), ("timing",), ("parity", "bar")):
        if len(line) <= LINE_LIMIT:
            break
        d = out
        for k in shed[:-1]:
            d = d.get(k) or {}
        if isinstance(d, dict) and shed[-1] in d:
            d[shed[-1]] = None
            line = json.dumps(out, separators=(",", ":"))
    if len(line) > LINE_LIMIT and out.get("configs"):
        for c in out["configs"].values():
            c.pop("k", None)
        line = json.dumps(out, separators=(",", ":"))
    return line


if __name__ == "__main__":
    main()
