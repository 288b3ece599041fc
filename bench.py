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

One step = one pass of the hot path over the rank's items, inputs resident in HBM:
    rast, rast_db = rasterize(ctx, pos, tri, (H, W));  out, _ = interpolate(attr, rast, tri)
    torch.autograd.backward(out, G)                    # upstream gradient G ~ N(0,1), fixed
(c3: interpolate(uv, diff_attrs='all') -> texture(trilinear) -> antialias in between.)
With N > 1 the step also contains the path's two exchanges (north_star): the all-gather of the per-item
output images to every rank -- issued per chunk of items on RCCL's stream while the next chunk is being rendered
(`--chunks`, `--no-gather-images`) -- and the all-reduce of the gradient of the SHARED vertex attributes.

Rank 0 prints ONE JSON line.  Beside the headline number it carries
  roofline      the LONGEST kernel of the step (hipEvents recorded by the library on the launch stream): algorithmic
                bytes / time against the 8 TB/s HBM peak, the PMC-measured HBM traffic if profiles/traffic.json has it;
  path_hbm_frac the whole step's algorithmic bytes / step time against the same peak;
  cpu_baseline  the CPU oracle (a port of the reference's algorithm -- the reference has no CPU path) on this box's
                host cores, bounded sample; cpu_reference = the reference's own kernels under the CUDA-on-CPU shim
                of oracle/refshim (one thread), when oracle/_ref is present;
  parity        id mismatches / max-abs errors of this very workload against the reference itself (oracle/_ref) when
                present, else against the oracle.
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
TRIANGLES = 10000

WORKLOADS = {
    # name: (items per GPU or None, total items or None, resolution, graph, scaling, attrs)
    "ch": dict(per_gpu=64, total=None, res=512, graph="ri", scaling="weak", attrs=4,
               metric="Mpixels/s rasterize+interpolate fwd+bwd @512^2 batch64"),
    "c2": dict(per_gpu=16, total=None, res=512, graph="ri", scaling="weak", attrs=4,
               metric="Mpixels/s rasterize+interpolate fwd+bwd @512^2 batch16 (BASELINE configs[1])"),
    "c3": dict(per_gpu=32, total=None, res=1024, graph="full", scaling="weak", attrs=2,
               metric="Mpixels/s rasterize+interpolate+texture(2048^2 mip)+antialias fwd+bwd @1024^2 batch32 (BASELINE configs[2])"),
    "c4": dict(per_gpu=None, total=256, res=512, graph="ri", scaling="strong", attrs=4,
               metric="Mpixels/s rasterize+interpolate fwd+bwd @512^2 batch256 sharded over the GPUs (BASELINE configs[3])"),
}


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
        "raster_grad_db": 48 * P,                               # R g_rast, g_rast_db, rast
    }
    return per_kernel, 384 * P


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


class DryKernels:
    """Stand-in for the HIP path in --dry-run-cpu: same tensor shapes, trivial arithmetic, autograd intact."""

    def __init__(self, dev):
        self.dev = dev

    def forward(self, pos, attr, res, A):
        n = pos.shape[0]
        base = pos[:, :1, :1].reshape(n, 1, 1, 1) + attr.mean()
        return base.expand(n, res, res, A).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="ch")
    ap.add_argument("--batch", type=int, default=None, help="items per GPU (overrides the workload's)")
    ap.add_argument("--res", type=int, default=None, help="resolution (overrides the workload's; dry runs use a small one)")
    ap.add_argument("--chunks", type=int, default=None,
                    help="N > 1: the rank's items are rendered in this many chunks so that a chunk's image all-gather "
                         "overlaps the next chunk's kernels (default: chunks of >= 64 items, at most 4; 1 at N = 1)")
    ap.add_argument("--no-gather-images", action="store_true", help="N > 1: keep the output images sharded (no all-gather in the step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-items", type=int, default=64, help="items in the CPU-oracle sample")
    ap.add_argument("--graph", action="store_true",
                    help="capture the step into one hipGraph and time replays (single GPU only; the default, and the "
                         "number the driver records, is eager launching)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="run the N > 1 code path (process group, broadcast, chunked all-gather, all-reduce) even with one rank: "
                         "checks the RCCL calls on a 1-GPU box")
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
    from nvdiffrast_amd.parallel import broadcast_shared, allreduce_shared_grads, gather_items_async, shard_range
    parallel.force_collectives(args.force_collectives)
    from nvdiffrast_amd.utils import m10k_batch

    wl = WORKLOADS[args.workload]
    RES = args.res or (32 if dry else wl["res"])
    A = wl["attrs"]
    full = wl["graph"] == "full"
    if wl["total"] is not None:                                   # strong scaling: a fixed batch split over the ranks
        total_items = wl["total"] if args.batch is None else args.batch * world
        first, N = shard_range(total_items, world, rank)
    else:
        N = args.batch or wl["per_gpu"]
        total_items, first = N * world, N * rank
    # A chunk must carry enough pixels to cover the host's launch work for it (measured: 16-item chunks at 512^2 doubled the
    # step time, 64-item chunks cost nothing); with one chunk the gather still overlaps the backward kernels.
    chunks = args.chunks or (max(1, min(4, N // 64)) if distributed else 1)
    chunks = max(1, min(chunks, N))
    gather = distributed and not args.no_gather_images

    # Per-item poses differ across ranks; geometry every item shares (tri, attr/uv, texture) comes from rank 0 over
    # RCCL, as it would in a data-parallel job.
    if dry:
        scene = m10k_batch(N, seed=20240, attrs=A, nx=8, ny=4, pose_seed=20240 + 1000 * rank)
    else:
        scene = m10k_batch(N, seed=20240, attrs=A, pose_seed=20240 + 1000 * rank)
    pos = torch.from_numpy(scene["pos"]).to(dev).requires_grad_(True)
    tri = torch.from_numpy(scene["tri"]).to(dev)
    shared = torch.from_numpy(scene["uv"] if full else scene["attr"]).to(dev)
    C_out = 3 if full else A
    tex = None
    if full:
        tex_res = 2048 if not dry else 32
        tex = torch.from_numpy(np.random.default_rng(5).uniform(size=(1, tex_res, tex_res, 3)).astype(np.float32)).to(dev)
    if distributed:
        broadcast_shared([tri, shared] + ([tex] if full else []), src=0)
    shared.requires_grad_(True)
    if full:
        tex.requires_grad_(True)
    G = torch.from_numpy(np.random.default_rng(77 + rank).normal(size=(N, RES, RES, C_out)).astype(np.float32)).to(dev)

    if dry:
        kernels_impl = DryKernels(dev)
        dr = ctx = topo = None
    else:
        import nvdiffrast_amd.torch as dr
        from nvdiffrast_amd import _capi
        lib = _capi.load()
        ctx = dr.RasterizeCudaContext(device=dev)
        topo = dr.antialias_construct_topology_hash(tri) if full else None

    bounds = [(N * c // chunks, N * (c + 1) // chunks) for c in range(chunks)]
    mode = {"gather": gather}                                     # switched off for the second, gather-free timing below
    gathered = [None] * chunks                                    # receive buffers, reused every step

    def render(p):
        """Forward of the op graph for the items `p` [n,V,4] -> output image [n,H,W,C]."""
        if dry:
            return kernels_impl.forward(p, shared, RES, C_out), None
        rast, rast_db = dr.rasterize(ctx, p, tri, (RES, RES))
        if not full:
            out, _ = dr.interpolate(shared, rast, tri)
            return out, rast
        uv, uv_da = dr.interpolate(shared, rast, tri, rast_db=rast_db, diff_attrs="all")
        col = dr.texture(tex, uv, uv_da, filter_mode="linear-mipmap-linear")
        return dr.antialias(col, rast, p, tri, topology_hash=topo), rast

    def step():
        pos.grad = None
        shared.grad = None
        if full:
            tex.grad = None
        pending = []
        last = None
        for c, (a, b) in enumerate(bounds):
            p = pos if chunks == 1 else pos[a:b]
            out, rast = render(p)
            if mode["gather"]:
                # the collective runs on RCCL's own stream, ordered after this chunk's kernels; the next chunk's
                # kernels are issued right away and overlap it
                work, gathered[c] = gather_items_async(out.detach(), out=gathered[c])
                pending.append(work)
            torch.autograd.backward(out, G if chunks == 1 else G[a:b])
            last = (rast, out)
        if distributed:
            allreduce_shared_grads([shared] + ([tex] if full else []))
        for w in pending:
            w.wait()
        return last

    def fence():
        if not dry:
            torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    run = step
    if args.graph:
        # No op of the path synchronises the host or allocates at the C-ABI level, so the whole step captures.
        assert not distributed and not dry, "--graph is a single-GPU measurement"
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        run = graph.replay
    def timed(warmup, steps):
        """`steps` steps bracketed by barrier + synchronize on both sides; returns the MAX over ranks of the elapsed seconds."""
        for _ in range(warmup):
            run()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        fence()
        el = time.perf_counter() - t0
        if distributed:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el

    elapsed = timed(args.warmup, args.steps)
    ranks_seen = world
    if distributed:
        ones = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)               # every rank took part in the timed region
        ranks_seen = int(round(float(ones.item())))
        assert ranks_seen == dist.get_world_size()
    # The image all-gather is pure xGMI traffic whose size does not depend on how fast the kernels are: report the
    # same job without it next to `value`, and the link arithmetic, so that the two effects can be told apart.
    collective = None
    if gather:
        mode["gather"] = False
        elapsed_ng = timed(min(args.warmup, 2), args.steps)
        mode["gather"] = True
        img_bytes = N * RES * RES * C_out * 4
        links = world - 1                                         # fully connected xGMI: one link per peer (0: a forced group of one)
        collective = {
            "image_bytes_sent_per_rank_per_step": img_bytes * links, "image_bytes_received_per_rank_per_step": img_bytes * links,
            "xgmi_links_per_gpu_used": links, "xgmi_link_gbs": XGMI_LINK_GBS,
            # every link carries one rank's images, all links in parallel: the floor does not shrink with more GPUs
            "xgmi_floor_ms": round(img_bytes / (XGMI_LINK_GBS * 1e9) * 1e3, 4) if links else None,
            "ms_per_step_without_image_gather": round(elapsed_ng / args.steps * 1e3, 4),
            "value_without_image_gather": round(total_items * RES * RES / (elapsed_ng / args.steps) / 1e6, 1),
        }

    P_rank = N * RES * RES
    P_total = total_items * RES * RES
    ms_per_step = elapsed / args.steps * 1e3
    value = P_total / (elapsed / args.steps) / 1e6                # whole-job Mpixels/s

    result = None
    if rank == 0:
        kernels = roofline = parity = cpu = cpu_ref = None
        path_frac = None
        if not dry:
            # ---- per-kernel timing (hipEvents on the launch stream, inside the library), one chunk per launch ----
            lib.nvdr_profile_reset()
            lib.nvdr_profile_enable(1)
            prof_steps = max(3, min(args.steps, 10))
            for _ in range(prof_steps):
                pos.grad = None; shared.grad = None
                if full:
                    tex.grad = None
                out, _ = render(pos)
                torch.autograd.backward(out, G)
            torch.cuda.synchronize()
            prof = _capi.profile_read()
            lib.nvdr_profile_enable(0)
            lib.nvdr_profile_reset()
            alg, path_bytes = algorithmic_bytes(wl["graph"], P_rank, A, int(tri.shape[0]), N)
            kernels = {}
            for name, (total_ms, launches) in prof.items():
                avg_ms = total_ms / max(launches, 1)
                b = alg.get(name)
                kernels[name] = {"avg_ms": round(avg_ms, 4), "launches_per_step": launches / prof_steps,
                                 "alg_bytes": b, "gbs": None if b is None else round(b / (avg_ms * 1e-3) / 1e9, 1)}
            # Dominant kernel = the longest one (time per step), full stop.
            dominant = max(kernels, key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches_per_step"])
            dk = kernels[dominant]
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")      # PMC-derived HBM bytes/launch (CH), if collected
            if args.workload == "ch" and N == 64 and os.path.exists(tpath):
                try:
                    traffic = json.load(open(tpath)).get(dominant)
                except Exception:  # noqa: BLE001
                    traffic = None
            roofline = {"bound": "hbm", "kernel": dominant, "achieved": dk["gbs"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": None if dk["gbs"] is None else round(dk["gbs"] / HBM_PEAK_GBS, 4),
                        "traffic": traffic,
                        "traffic_ratio": None if not (traffic and dk["alg_bytes"]) else round(traffic / dk["alg_bytes"], 3),
                        "kernel_avg_ms": dk["avg_ms"], "alg_bytes_per_launch": dk["alg_bytes"]}
            step_ms_1gpu = sum(kernels[k]["avg_ms"] * kernels[k]["launches_per_step"] for k in kernels)
            path_frac = round((path_bytes / (max(ms_per_step, 1e-9) * 1e-3) / 1e9) / HBM_PEAK_GBS, 4)

            # ---- parity of this workload against the reference itself (oracle/_ref) or the oracle ----------------
            if world == 1 and not full:
                import oracle
                from oracle import ref as oref
                chk, chk_name = (oref, "reference (oracle/_ref)") if oref.available() else (oracle, "oracle")
                ns = 2
                ro, _ = chk.rasterize(scene["pos"][:ns], scene["tri"], (RES, RES))
                Gs = G[:ns].cpu().numpy()
                ga_o, gr_o, _ = chk.interpolate_grad(scene["attr"], ro, scene["tri"], Gs)
                gp_o = chk.rasterize_grad(scene["pos"][:ns], scene["tri"], ro, gr_o)
                pos_s = torch.from_numpy(scene["pos"][:ns]).to(dev).requires_grad_(True)
                attr_s = torch.from_numpy(scene["attr"]).to(dev).requires_grad_(True)
                r_s, _ = dr.rasterize(ctx, pos_s, tri, (RES, RES))
                o_s, _ = dr.interpolate(attr_s, r_s, tri)
                torch.autograd.backward(o_s, G[:ns])
                parity = {
                    "against": chk_name, "items": ns,
                    "tri_id_mismatches": int((r_s[..., 3].detach().cpu().numpy() != ro[..., 3]).sum()),
                    "bary_max_abs_err": float(np.abs(r_s[..., :3].detach().cpu().numpy() - ro[..., :3]).max()),
                    "g_attr_max_abs_err": float(np.abs(attr_s.grad.cpu().numpy() - ga_o).max()),
                    "g_attr_max_abs": float(np.abs(ga_o).max()),
                    "g_pos_max_abs_err": float(np.abs(pos_s.grad.cpu().numpy() - gp_o).max()),
                    "g_pos_max_abs": float(np.abs(gp_o).max()),
                }

                # ---- CPU baselines on this host's cores, bounded samples ---------------------------------------
                if not args.no_cpu_baseline:
                    def chain(mod, pc, tc, ac, Gc):
                        r_c, _ = mod.rasterize(pc, tc, (RES, RES))
                        mod.interpolate(ac, r_c, tc)
                        _ga, gr, _ = mod.interpolate_grad(ac, r_c, tc, Gc)
                        mod.rasterize_grad(pc, tc, r_c, gr)

                    nc = max(1, min(args.cpu_items, N))
                    Gc = G[:nc].cpu().numpy()
                    times = []
                    for _rep in range(5):
                        t1 = time.perf_counter()
                        chain(oracle, scene["pos"][:nc], scene["tri"], scene["attr"], Gc)
                        times.append(time.perf_counter() - t1)
                    tmed = sorted(times)[2]
                    cpu = {"value": round(nc * RES * RES / tmed / 1e6, 2), "unit": "Mpixels/s", "cores": oracle.num_threads(),
                           "kind": "port",
                           "sample": f"{nc} of the {N} items of the same batch, fwd+bwd, median of 5 (the reference has no CPU "
                                     f"path; this is the repo's C/OpenMP restatement, pinned to the reference by the tests), "
                                     f"host cpu_count={os.cpu_count()}"}
                    if oref.available():
                        nr = min(4, N)
                        t1 = time.perf_counter()
                        chain(oref, scene["pos"][:nr], scene["tri"], scene["attr"], G[:nr].cpu().numpy())
                        tr = time.perf_counter() - t1
                        cpu_ref = {"value": round(nr * RES * RES / tr / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": "reference",
                                   "sample": f"{nr} items of the same batch, fwd+bwd, one run: the reference's own CUDA kernels and "
                                             f"CudaRaster compiled for the host and executed by a fibre-based CUDA-on-CPU shim "
                                             f"(oracle/refshim) -- an emulation on one thread, not a tuned CPU implementation"}

        cfg = {"workload": "%s: random-pose 10k-triangle lattice mesh (T=%d, V=%d), %d items on this GPU of %d in total @%dx%d, %s, "
                           "upstream grad fed to backward directly"
                           % (args.workload.upper(), int(tri.shape[0]), int(pos.shape[1]), N, total_items, RES, RES,
                              "rasterize+interpolate(uv,da)+texture(2048^2 trilinear)+antialias fwd+bwd" if full
                              else "A=%d attrs, rasterize+interpolate fwd+bwd" % A),
               "batch_per_gpu": N, "total_items": total_items, "resolution": [RES, RES], "triangles": int(tri.shape[0]),
               "parallelism": "dp%d (items sharded; per step: %sall-reduce of the shared-input gradients)"
                              % (world, ("all-gather of the output images in %d chunks overlapped with rendering, " % chunks) if gather else ""),
               "chunks": chunks, "gather_images": bool(gather),
               "launch": "hipGraph replay" if args.graph else "eager"}
        result = {
            "metric": wl["metric"],
            "value": round(value, 1), "unit": "Mpixels/s",
            "n_gpus": world, "rccl_ranks": ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": wl["scaling"],
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": cfg,
            "roofline": roofline,
            "path_hbm_frac": path_frac,
            "kernels": kernels,
            "cpu_baseline": cpu,
            "cpu_reference": cpu_ref,
            "parity": parity,
            "collective": collective,
        }
        if dry:
            result["dry_run"] = True
            result["backend"] = "gloo" if distributed else "none"
            if gather:
                g0 = gathered[0]
                result["gathered_rows_chunk0"] = int(g0.shape[0])
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
