"""Run the reference's OWN Python layer (``nvdiffrast/torch/ops.py``, unmodified, loaded from the reference
checkout) on a plugin module of our choice -- TEST INFRASTRUCTURE ONLY.

``load_reference_ops(plugin)`` executes the reference's ops.py with ``import _nvdiffrast_c`` resolved to
``plugin``:
  * ``plugin = nvdiffrast_amd.torch._plugin``  -> the reference's operator layer on the MI355X kernels, exactly
    the binding INTEGRATION.md section 1 describes (GPU tests);
  * ``plugin = cpu_plugin()``                  -> the reference's operator layer on the reference's C++/CUDA
    sources compiled for the host (oracle/_ref): the whole reference stack running on the CPU, which is how
    BASELINE config 1 ("triangle.py on the reference CPU context") and the golden fixtures of
    tests/golden/make_reference_fixture.py are produced.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

from . import ref as _ref

REFERENCE_OPS = os.environ.get("NVDR_REFERENCE_OPS", "/root/reference/nvdiffrast/torch/ops.py")


def reference_ops_available():
    return os.path.exists(REFERENCE_OPS)


def load_reference_ops(plugin, name="nvdr_reference_ops"):
    """The reference's ops.py as a module whose ``_nvdiffrast_c`` is ``plugin``."""
    saved = sys.modules.get("_nvdiffrast_c")
    sys.modules["_nvdiffrast_c"] = plugin
    try:
        spec = importlib.util.spec_from_file_location(name, REFERENCE_OPS)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        if saved is None:
            sys.modules.pop("_nvdiffrast_c", None)
        else:
            sys.modules["_nvdiffrast_c"] = saved
    return mod


# --------------------------------------------------------------------------- the CPU build as a torch plugin
def _np(t):
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        if t.numel() == 0 and t.dim() == 1:
            return None                     # ops.py passes absent optional tensors as torch.tensor([]) (ops.py:301-305)
        return t.detach().cpu().numpy()
    return t


def _t(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a))


class _TorchMip:
    def __init__(self, w):
        self.w = w


class CpuPlugin:
    """``_nvdiffrast_c`` on CPU torch tensors, executed by oracle/_ref (the reference's own code)."""

    def __init__(self, variant="fma"):
        self.P = _ref.plugin(variant)
        self.TextureMipWrapper = _ref.TextureMipWrapper
        self.TopologyHashWrapper = _ref.TopologyHashWrapper
        self._empty_mip = None

    def get_log_level(self):
        return self.P.get_log_level()

    def set_log_level(self, level):
        self.P.set_log_level(level)

    def RasterizeCRStateWrapper(self, idx):
        return self.P.RasterizeCRStateWrapper(idx)

    def rasterize_fwd_cuda(self, state, pos, tri, resolution, ranges, peeling_idx):
        return tuple(_t(a) for a in self.P.rasterize_fwd_cuda(state, _np(pos), _np(tri), resolution, ranges.numpy(), peeling_idx))

    def rasterize_grad(self, pos, tri, out, dy):
        return _t(self.P.rasterize_grad(_np(pos), _np(tri), _np(out), _np(dy.contiguous())))

    def rasterize_grad_db(self, pos, tri, out, dy, ddb):
        return _t(self.P.rasterize_grad_db(_np(pos), _np(tri), _np(out), _np(dy.contiguous()), _np(ddb.contiguous())))

    def interpolate_fwd(self, attr, rast, tri):
        return tuple(_t(a) for a in self.P.interpolate_fwd(_np(attr), _np(rast), _np(tri)))

    def interpolate_fwd_da(self, attr, rast, tri, rast_db, diff_all, diff_list):
        return tuple(_t(a) for a in self.P.interpolate_fwd_da(_np(attr), _np(rast), _np(tri), _np(rast_db), diff_all, diff_list))

    def interpolate_grad(self, attr, rast, tri, dy):
        return tuple(_t(a) for a in self.P.interpolate_grad(_np(attr), _np(rast), _np(tri), _np(dy.contiguous())))

    def interpolate_grad_da(self, attr, rast, tri, dy, rast_db, dda, diff_all, diff_list):
        return tuple(_t(a) for a in self.P.interpolate_grad_da(_np(attr), _np(rast), _np(tri), _np(dy.contiguous()), _np(rast_db),
                                                               _np(dda.contiguous()), diff_all, diff_list))

    def texture_construct_mip(self, tex, max_mip_level, cube_mode):
        return self.P.texture_construct_mip(_np(tex), max_mip_level, cube_mode)

    def texture_fwd(self, tex, uv, f, b):
        return _t(self.P.texture_fwd(_np(tex), _np(uv), f, b))

    def _wrapper(self, w):
        if isinstance(w, _ref.TextureMipWrapper):
            return w
        return self.P.TextureMipWrapper()

    def texture_fwd_mip(self, tex, uv, uv_da, bias, wrapper, stack, f, b):
        return _t(self.P.texture_fwd_mip(_np(tex), _np(uv), _np(uv_da), _np(bias), self._wrapper(wrapper), [_np(s) for s in stack], f, b))

    def texture_grad_nearest(self, tex, uv, dy, f, b):
        return _t(self.P.texture_grad_nearest(_np(tex), _np(uv), _np(dy.contiguous()), f, b))

    def texture_grad_linear(self, tex, uv, dy, f, b):
        return tuple(_t(a) for a in self.P.texture_grad_linear(_np(tex), _np(uv), _np(dy.contiguous()), f, b))

    def texture_grad_linear_mipmap_nearest(self, tex, uv, dy, uv_da, bias, wrapper, stack, f, b):
        g_tex, g_uv, g_stack = self.P.texture_grad_linear_mipmap_nearest(_np(tex), _np(uv), _np(dy.contiguous()), _np(uv_da), _np(bias),
                                                                         self._wrapper(wrapper), [_np(s) for s in stack], f, b)
        return _t(g_tex), _t(g_uv), [_t(g) for g in g_stack]

    def texture_grad_linear_mipmap_linear(self, tex, uv, dy, uv_da, bias, wrapper, stack, f, b):
        g_tex, g_uv, g_da, g_bias, g_stack = self.P.texture_grad_linear_mipmap_linear(_np(tex), _np(uv), _np(dy.contiguous()), _np(uv_da), _np(bias),
                                                                                      self._wrapper(wrapper), [_np(s) for s in stack], f, b)
        return _t(g_tex), _t(g_uv), _t(g_da), _t(g_bias), [_t(g) for g in g_stack]

    def antialias_construct_topology_hash(self, tri):
        return self.P.antialias_construct_topology_hash(_np(tri))

    def antialias_fwd(self, color, rast, pos, tri, topology_hash):
        out, work = self.P.antialias_fwd(_np(color), _np(rast), _np(pos), _np(tri), topology_hash)
        return _t(out), _t(work)

    def antialias_grad(self, color, rast, pos, tri, dy, work_buffer):
        return tuple(_t(a) for a in self.P.antialias_grad(_np(color), _np(rast), _np(pos), _np(tri), _np(dy.contiguous()), _np(work_buffer)))


def cpu_plugin(variant="fma"):
    return CpuPlugin(variant)


class _FakeCudaDevice:
    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def reference_on_cpu(variant="fma"):
    """The reference's ops module bound to the CPU build.  ``RasterizeCudaContext.__init__`` asks torch for the
    current CUDA device (ops.py:62-67); on a machine without one the returned module's context class is given a
    device index of 0 through a two-line shim around those two torch calls."""
    mod = load_reference_ops(cpu_plugin(variant), name="nvdr_reference_ops_cpu_" + variant)

    class _Cuda:
        device = _FakeCudaDevice

        @staticmethod
        def current_device():
            return 0

    class _TorchView:
        """`torch` as ops.py sees it: everything from torch, except torch.cuda.{current_device, device}."""
        cuda = _Cuda

        def __getattr__(self, name):
            return getattr(torch, name)

    mod.torch = _TorchView()
    return mod
