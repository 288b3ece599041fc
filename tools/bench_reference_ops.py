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


def measure(layer, N, steps=20, windows=5):
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
    return sorted(ms)[len(ms) // 2], (pos.grad.clone(), attr.grad.clone())


if __name__ == "__main__":
    N = 16 if (len(sys.argv) > 1 and sys.argv[1] == "c2") else 64
    ref = reference_layer()
    assert ref._nvdiffrast_c is _plugin
    t_ref, g_ref = measure(ref, N)
    t_own, g_own = measure(dr, N)
    _plugin.set_tile_skipping(False)
    t_ref_noflags, _ = measure(ref, N)
    _plugin.set_tile_skipping(True)
    rel = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(g_ref, g_own)]
    print(json.dumps({"items": N, "ms_reference_ops_on_plugin": round(t_ref, 4), "ms_package_ops": round(t_own, 4),
                      "ms_reference_ops_without_tile_flags": round(t_ref_noflags, 4), "ratio": round(t_ref / t_own, 3),
                      "grad_rel_diff_pos_attr": [float("%.2e" % r) for r in rel]}))
