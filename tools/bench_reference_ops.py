#!/usr/bin/env python3
"""The headline step driven by THE REFERENCE'S OWN nvdiffrast/torch/ops.py bound to nvdiffrast_amd.torch._plugin
(literally) next to the same step through this package's operator layer.  The reference's layer
passes no tile flags and never calls the fused backward: what it gets is what the plugin finds by itself (the flags travel with
the tensors rasterize_fwd_cuda / interpolate_fwd return), not what nvdiffrast_amd/torch/ops.py adds on top (the fused pair).
    NVDR_REFERENCE_OPS=/path/to/reference/nvdiffrast/torch/ops.py python tools/bench_reference_ops.py [ch|c2]
(tools/gpurun_reference_ops_bench.sh ships the file to the GPU box inside the command line.)"""
import importlib.util, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nvdiffrast_amd.torch as dr
from nvdiffrast_amd.torch import _plugin
from nvdiffrast_amd.utils import m10k_batch


def reference_layer():
    path = os.environ.get("NVDR_REFERENCE_OPS", "/root/reference/nvdiffrast/torch/ops.py")
    sys.modules["_nvdiffrast_c"] = _plugin
    spec = importlib.util.spec_from_file_location("nvdr_reference_ops_bench", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def measure(layer, N, steps=20, windows=5, profile=False):
    dev = torch.device("cuda", 0)
    b = m10k_batch(N)
    pos = torch.from_numpy(b["pos"]).to(dev).requires_grad_(True)
    tri = torch.from_numpy(b["tri"]).to(dev)
    attr = torch.from_numpy(b["attr"]).to(dev).requires_grad_(True)
    G = torch.randn(N, 512, 512, 4, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    ctx = layer.RasterizeCudaContext(device=dev)

    def step():
        pos.grad = None; attr.grad = None
        rast, _ = layer.rasterize(ctx, pos, tri, (512, 512))
        out, _ = layer.interpolate(attr, rast, tri)
        torch.autograd.backward(out, G)

    for _ in range(5): step()
    ms = []
    for _ in range(windows):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): step()
        torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) / steps * 1e3)
    kernels = None
    if profile:                                      # the library's own per-launch timing (hipEvents around every launch), 10 steps
        from nvdiffrast_amd import _capi
        lib = _capi.load()
        lib.nvdr_profile_reset(); lib.nvdr_profile_enable(1)
        for _ in range(10): step()
        torch.cuda.synchronize()
        kernels = {k: round(t / c * 1e3, 1) for k, (t, c) in sorted(_capi.profile_read().items())}     # us per launch
        lib.nvdr_profile_enable(0); lib.nvdr_profile_reset()
    return sorted(ms)[len(ms) // 2], (pos.grad.clone(), attr.grad.clone()), kernels


if __name__ == "__main__":
    N = 16 if (len(sys.argv) > 1 and sys.argv[1] == "c2") else 64
    ref = reference_layer()
    assert ref._nvdiffrast_c is _plugin
    t_ref, g_ref, k_ref = measure(ref, N, profile=True)
    t_own, g_own, k_own = measure(dr, N, profile=True)
    _plugin.set_tile_skipping(False)
    t_ref_noflags, _, _ = measure(ref, N)
    _plugin.set_tile_skipping(True)
    # gradient agreement, norm-wise (||a - b|| / ||b||: the position gradient sums terms of both signs per vertex, so the largest
    # single difference relative to the largest entry mostly shows the summation order of the atomics) and entry-wise; `again` is
    # the reference layer against ITSELF on a second run -- the floor the order of the atomics sets
    _, g_again, _ = measure(ref, N, steps=2, windows=1)
    # ... and with the ONE line INTEGRATION.md section 1 suggests adding to the reference's _rasterize_func.forward: gradients of
    # outputs nobody used arrive as None instead of 16 B/pixel of zeros that autograd writes and rasterize_grad_db reads back
    inner = ref._rasterize_func.forward

    def forward(ctx, *a):
        ctx.set_materialize_grads(False)
        return inner(ctx, *a)
    ref._rasterize_func.forward = staticmethod(forward)
    t_ref_line, g_line, k_line = measure(ref, N, profile=True)
    ref._rasterize_func.forward = staticmethod(inner)
    line = [float((a - b).double().norm() / b.double().norm()) for a, b in zip(g_line, g_own)]
    rel = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(g_ref, g_own)]
    nrm = [float((a - b).double().norm() / b.double().norm()) for a, b in zip(g_ref, g_own)]
    again = [float((a - b).double().norm() / b.double().norm()) for a, b in zip(g_ref, g_again)]
    print(json.dumps({"items": N, "ms_reference_ops_on_plugin": round(t_ref, 4), "ms_package_ops": round(t_own, 4),
                      "ms_reference_ops_without_tile_flags": round(t_ref_noflags, 4), "ratio": round(t_ref / t_own, 3),
                      "ms_reference_ops_no_materialized_zeros": round(t_ref_line, 4), "ratio_no_materialized_zeros": round(t_ref_line / t_own, 3),
                      "grad_norm_diff_no_materialized_zeros": [float("%.2e" % r) for r in line], "kernel_us_no_materialized_zeros": k_line,
                      "grad_rel_diff_pos_attr": [float("%.2e" % r) for r in rel], "grad_norm_diff_pos_attr": [float("%.2e" % r) for r in nrm],
                      "grad_norm_diff_run_to_run": [float("%.2e" % r) for r in again], "fused_backward": _plugin.fused_backward_count(),
                      "kernel_us_reference_ops": k_ref, "kernel_us_package_ops": k_own}))
