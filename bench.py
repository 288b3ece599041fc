#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of rasterize+interpolate forward+backward at 512^2, batch 64
per GPU (BASELINE.json `metric`; SURVEY.md 8(d) workload "CH").

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one batch that is already resident in HBM:
    rast, rast_db = rasterize(ctx, pos, tri, (512, 512))
    out, _        = interpolate(attr, rast, tri)
    torch.autograd.backward(out, G)        # upstream gradient G ~ N(0,1), fixed
(+ all-reduce of the shared attribute gradient when N > 1: `attr` is one [1,V,4] tensor
shared by every item, so its gradient is the path's only cross-rank exchange; per-item
`pos` gradients stay local.)  Rank 0 prints ONE JSON line.

Beside the headline number the line carries
  roofline      the dominant kernel's achieved HBM GB/s (algorithmic bytes / hipEvent time,
                events recorded by the library on the launch stream) against the 8 TB/s peak;
  cpu_baseline  the CPU oracle (a port of the reference's algorithm; the reference itself has
                no CPU path) timed on this box's host cores on a bounded sample;
  parity        id mismatches / max-abs errors of this very workload against the oracle.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RES = 512
BATCH = 64
ATTRS = 4
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_launch(P, A, T, V, N):
    """compulsory tensor traffic per kernel launch (geometry is cache resident)."""
    return {
        "raster_setup": N * T * (12 + 48 + 68),            # tri + 3 verts in, record + AABB out
        "raster_fine": 32 * P,                              # W rast 16 + W rast_db 16
        "interp_fwd": (16 + 4 * A) * P,                     # R rast, W out
        "interp_grad": (4 * A + 16 + 16) * P,               # R dy, R rast, W g_rast
        "raster_grad": 32 * P,                              # R g_rast 16 + R rast 16
        "raster_grad_db": 48 * P,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH, help="items per GPU (default: the metric's 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-items", type=int, default=64, help="items in the CPU-oracle sample")
    ap.add_argument("--gather-images", action="store_true",
                    help="also all-gather the per-item output images to every rank inside the step (off: items stay sharded)")
    ap.add_argument("--graph", action="store_true",
                    help="capture the step into one hipGraph and time replays (single GPU only; the default, and the "
                         "number the driver records, is eager launching)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)     # RCCL over xGMI

    import nvdiffrast_amd.torch as dr
    from nvdiffrast_amd import _capi
    from nvdiffrast_amd.parallel import broadcast_shared, allreduce_shared_grads, gather_items
    from nvdiffrast_amd.utils import m10k_batch

    lib = _capi.load()
    N = args.batch
    # Each rank renders its own 64 items (weak scaling); geometry that all items share
    # (tri, attr) comes from rank 0 over RCCL, as it would in a data-parallel job.
    scene = m10k_batch(N, seed=20240, attrs=ATTRS, pose_seed=20240 + 1000 * rank)
    pos = torch.from_numpy(scene["pos"]).to(dev).requires_grad_(True)
    tri = torch.from_numpy(scene["tri"]).to(dev)
    attr = torch.from_numpy(scene["attr"]).to(dev)
    if distributed:
        broadcast_shared([tri, attr], src=0)
    attr.requires_grad_(True)
    G = torch.from_numpy(np.random.default_rng(77 + rank).normal(size=(N, RES, RES, ATTRS)).astype(np.float32)).to(dev)
    ctx = dr.RasterizeCudaContext(device=dev)

    def step():
        pos.grad = None
        attr.grad = None
        rast, rast_db = dr.rasterize(ctx, pos, tri, (RES, RES))
        out, _ = dr.interpolate(attr, rast, tri)
        torch.autograd.backward(out, G)
        if distributed:
            allreduce_shared_grads([attr])
            if args.gather_images:
                gather_items(out.detach(), N * world)
        return rast, out

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    run = step
    if args.graph:
        # No op of the path synchronises the host or allocates at the C-ABI level, so the whole step captures.
        assert not distributed, "--graph is a single-GPU measurement"
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
    for _ in range(args.warmup):
        run()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    P = N * RES * RES
    ms_per_step = elapsed / args.steps * 1e3
    value = world * P / (elapsed / args.steps) / 1e6              # whole-job Mpixels/s

    result = None
    if rank == 0:
        # ---- per-kernel timing (hipEvents on the launch stream, inside the library) --------
        lib.nvdr_profile_reset()
        lib.nvdr_profile_enable(1)
        prof_steps = max(3, min(args.steps, 10))
        for _ in range(prof_steps):
            pos.grad = None; attr.grad = None
            rast, rast_db = dr.rasterize(ctx, pos, tri, (RES, RES))
            out, _ = dr.interpolate(attr, rast, tri)
            torch.autograd.backward(out, G)
        torch.cuda.synchronize()
        prof = _capi.profile_read()
        lib.nvdr_profile_enable(0)
        lib.nvdr_profile_reset()
        alg = algorithmic_bytes_per_launch(P, ATTRS, tri.shape[0], pos.shape[1], N)
        kernels = {}
        for name, (total_ms, launches) in prof.items():
            avg_ms = total_ms / max(launches, 1)
            b = alg.get(name)
            kernels[name] = {"avg_ms": round(avg_ms, 4), "launches_per_step": launches / prof_steps,
                             "alg_bytes": b, "gbs": None if b is None else round(b / (avg_ms * 1e-3) / 1e9, 1)}
        # Dominant kernel = largest share of the step.  k_fine and k_interp_grad take the same time to within
        # run-to-run noise, so kernels within 5 % of the longest are treated as tied and the tie goes to the one
        # that moves the most algorithmic bytes (every kernel's own numbers are in `kernels` either way).
        share = {k: kernels[k]["avg_ms"] * kernels[k]["launches_per_step"] for k in kernels}
        top = max(share.values())
        dominant = max((k for k in share if share[k] >= 0.95 * top), key=lambda k: (kernels[k]["alg_bytes"] or 0, share[k]))
        dk = kernels[dominant]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")      # PMC-derived HBM bytes/launch, if collected
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dominant)
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": dominant, "achieved": dk["gbs"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": None if dk["gbs"] is None else round(dk["gbs"] / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "kernel_avg_ms": dk["avg_ms"], "alg_bytes_per_launch": dk["alg_bytes"]}
        path_bytes = (112 + 8 * ATTRS) * P                           # 144 B/pixel at A = 4

        # ---- parity of this workload against the oracle (checker only; single-GPU runs) ----
        parity = cpu = None
        run_checks = (world == 1)
        if run_checks:
            import oracle
            ns = 2
            ro, _ = oracle.rasterize(scene["pos"][:ns], scene["tri"], (RES, RES))
            Gs = G[:ns].cpu().numpy()
            ga_o, gr_o, _ = oracle.interpolate_grad(scene["attr"], ro, scene["tri"], Gs)
            gp_o = oracle.rasterize_grad(scene["pos"][:ns], scene["tri"], ro, gr_o)
            # device gradients for the same two items
            pos_s = torch.from_numpy(scene["pos"][:ns]).to(dev).requires_grad_(True)
            attr_s = torch.from_numpy(scene["attr"]).to(dev).requires_grad_(True)
            r_s, _ = dr.rasterize(ctx, pos_s, tri, (RES, RES))
            o_s, _ = dr.interpolate(attr_s, r_s, tri)
            torch.autograd.backward(o_s, G[:ns])
            parity = {
                "items": ns,
                "tri_id_mismatches": int((r_s[..., 3].detach().cpu().numpy() != ro[..., 3]).sum()),
                "bary_max_abs_err": float(np.abs(r_s[..., :3].detach().cpu().numpy() - ro[..., :3]).max()),
                "g_attr_max_abs_err": float(np.abs(attr_s.grad.cpu().numpy() - ga_o).max()),
                "g_pos_max_abs_err": float(np.abs(pos_s.grad.cpu().numpy() - gp_o).max()),
                "g_pos_max_abs": float(np.abs(gp_o).max()),
            }

            # ---- CPU baseline: the oracle on this host's cores, bounded sample -----------------
            if not args.no_cpu_baseline:
                nc = max(1, min(args.cpu_items, N))
                pc, tc, ac = scene["pos"][:nc], scene["tri"], scene["attr"]
                Gc = G[:nc].cpu().numpy()
                times = []
                for rep in range(5):
                    t1 = time.perf_counter()
                    r_c, _ = oracle.rasterize(pc, tc, (RES, RES))
                    o_c, _ = oracle.interpolate(ac, r_c, tc)
                    ga, gr, _ = oracle.interpolate_grad(ac, r_c, tc, Gc)
                    gp = oracle.rasterize_grad(pc, tc, r_c, gr)
                    times.append(time.perf_counter() - t1)
                tmed = sorted(times)[2]
                cpu = {"value": round(nc * RES * RES / tmed / 1e6, 2), "unit": "Mpixels/s", "cores": oracle.num_threads(),
                       "kind": "port",
                       "sample": f"{nc} of the {N} items of the same batch, fwd+bwd, median of 5 (reference has no CPU path; "
                                 f"this is the repo's C/OpenMP restatement), host cpu_count={os.cpu_count()}"}

        result = {
            "metric": "Mpixels/s rasterize+interpolate fwd+bwd @512^2 batch64",
            "value": round(value, 1), "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "CH: random-pose 10k-triangle lattice mesh (T=10000, V=5151), batch %d per GPU @%dx%d, "
                                   "A=%d attrs, rasterize+interpolate fwd+bwd, upstream grad fed to backward directly"
                                   % (N, RES, RES, ATTRS),
                       "batch_per_gpu": N, "resolution": [RES, RES], "triangles": int(tri.shape[0]),
                       "parallelism": "dp%d (items sharded, shared-attr grad all-reduce)" % world,
                       "launch": "hipGraph replay" if args.graph else "eager"},
            "roofline": roofline,
            "path_hbm_frac": round((path_bytes / (ms_per_step * 1e-3) / 1e9) / HBM_PEAK_GBS, 4),
            "kernels": kernels,
            "cpu_baseline": cpu,
            "parity": parity,
        }
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
