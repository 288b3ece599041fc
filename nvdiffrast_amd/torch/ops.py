"""Public API: same names, signatures, defaults and return arities as the reference's
``nvdiffrast/torch/ops.py`` (cited per function), running on the MI355X plugin.

Autograd wiring follows the reference's five ``torch.autograd.Function`` classes: the
same tensors are saved, the same backward entry points are chosen, and gradients are
returned for the same inputs.
"""
import warnings

import numpy as np
import torch

from . import _plugin

__all__ = [
    "RasterizeCudaContext", "RasterizeGLContext", "get_log_level", "set_log_level",
    "rasterize", "DepthPeeler", "interpolate",
]


# ----------------------------------------------------------------------------- logging
# reference ops.py:18-41

def get_log_level():
    """Current log level (0 info, 1 warning, 2 error, 3 fatal)."""
    return _plugin.get_log_level()


def set_log_level(level):
    """Set the log level; messages below it are silent.  Default is 1."""
    _plugin.set_log_level(level)


# ----------------------------------------------------------------------------- context
# reference ops.py:47-68

class RasterizeCudaContext:
    """Rasterizer context bound to one GPU.  Holds the rasterizer's scratch memory; it is
    released with the object.  Not thread-safe, like the reference's."""

    def __init__(self, device=None):
        if device is None:
            idx = torch.cuda.current_device()
        else:
            with torch.cuda.device(device):
                idx = torch.cuda.current_device()
        self.cpp_wrapper = _plugin.RasterizeCRStateWrapper(idx)
        self.active_depth_peeler = None


# ----------------------------------------------------------------------------- rasterize
# reference ops.py:75-135

class _rasterize_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raster_ctx, pos, tri, resolution, ranges, grad_db, peeling_idx):
        out, out_db = _plugin.rasterize_fwd_cuda(raster_ctx.cpp_wrapper, pos, tri, resolution, ranges, peeling_idx)
        ctx.save_for_backward(pos, tri, out)
        ctx.saved_grad_db = grad_db
        # An unused output's gradient arrives as None instead of a materialised zero tensor;
        # that is 32 B/pixel the reference writes and reads back for nothing.
        ctx.set_materialize_grads(False)
        return out, out_db

    @staticmethod
    def backward(ctx, dy, ddb):
        pos, tri, out = ctx.saved_tensors
        if dy is None and (ddb is None or not ctx.saved_grad_db):
            return None, None, None, None, None, None, None
        if dy is None:
            dy = torch.zeros_like(out)
        if ctx.saved_grad_db and ddb is not None:
            g_pos = _plugin.rasterize_grad_db(pos, tri, out, dy, ddb)
        else:
            g_pos = _plugin.rasterize_grad(pos, tri, out, dy)
        return None, g_pos, None, None, None, None, None


def _empty_ranges():
    return torch.empty(size=(0, 2), dtype=torch.int32, device="cpu")


def rasterize(glctx, pos, tri, resolution, ranges=None, grad_db=True):
    """Rasterize triangles (reference ops.py:93-135).

    pos: [N,V,4] float32 (instanced mode) or [V,4] (range mode, needs ``ranges``);
    tri: [T,3] int32; resolution: (height, width); ranges: CPU int32 [N,2] (start, count).
    Returns (rast [N,H,W,4] = (u, v, z/w, triangle_id + 1), rast_db [N,H,W,4] =
    (du/dX, du/dY, dv/dX, dv/dY)).  ``grad_db`` routes rast_db's gradients into ``pos``.
    """
    assert isinstance(glctx, RasterizeCudaContext)
    assert grad_db is True or grad_db is False
    assert isinstance(pos, torch.Tensor) and isinstance(tri, torch.Tensor)
    resolution = tuple(resolution)
    if ranges is None:
        ranges = _empty_ranges()
    else:
        assert isinstance(ranges, torch.Tensor)
    if glctx.active_depth_peeler is not None:
        # The reference returns (does not raise) this error object (ops.py:131-132).
        return RuntimeError("Cannot call rasterize() during depth peeling operation, use rasterize_next_layer() instead")
    return _rasterize_func.apply(glctx, pos, tri, resolution, ranges, grad_db, -1)


# ----------------------------------------------------------------------------- depth peeling
# reference ops.py:141-204

class DepthPeeler:
    """Context manager that rasterizes successive depth layers; arguments as ``rasterize()``."""

    def __init__(self, glctx, pos, tri, resolution, ranges=None, grad_db=True):
        assert isinstance(glctx, RasterizeCudaContext)
        assert grad_db is True or grad_db is False
        assert isinstance(pos, torch.Tensor) and isinstance(tri, torch.Tensor)
        resolution = tuple(resolution)
        if ranges is None:
            ranges = _empty_ranges()
        else:
            assert isinstance(ranges, torch.Tensor)
        self.raster_ctx = glctx
        self.pos = pos
        self.tri = tri
        self.resolution = resolution
        self.ranges = ranges
        self.grad_db = grad_db
        self.peeling_idx = None

    def __enter__(self):
        if self.raster_ctx is None:
            raise RuntimeError("Cannot re-enter a terminated depth peeling operation")
        if self.raster_ctx.active_depth_peeler is not None:
            raise RuntimeError("Cannot have multiple depth peelers active simultaneously in a rasterization context")
        self.raster_ctx.active_depth_peeler = self
        self.peeling_idx = 0
        return self

    def __exit__(self, *args):
        assert self.raster_ctx.active_depth_peeler is self
        self.raster_ctx.active_depth_peeler = None
        # Drop every reference to the inputs.
        self.raster_ctx = self.pos = self.tri = self.resolution = None
        self.ranges = self.grad_db = self.peeling_idx = None
        return None

    def rasterize_next_layer(self):
        """Like ``rasterize()`` but surface points reported by earlier layers are culled."""
        assert self.raster_ctx.active_depth_peeler is self
        assert self.peeling_idx >= 0
        result = _rasterize_func.apply(self.raster_ctx, self.pos, self.tri, self.resolution, self.ranges,
                                       self.grad_db, self.peeling_idx)
        self.peeling_idx += 1
        return result


# ----------------------------------------------------------------------------- interpolate
# reference ops.py:211-291

class _interpolate_func_da(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attr, rast, tri, rast_db, diff_attrs_all, diff_attrs_list):
        out, out_da = _plugin.interpolate_fwd_da(attr, rast, tri, rast_db, diff_attrs_all, diff_attrs_list)
        ctx.save_for_backward(attr, rast, tri, rast_db)
        ctx.saved_misc = diff_attrs_all, diff_attrs_list
        return out, out_da

    @staticmethod
    def backward(ctx, dy, dda):
        attr, rast, tri, rast_db = ctx.saved_tensors
        diff_attrs_all, diff_attrs_list = ctx.saved_misc
        g_attr, g_rast, g_rast_db = _plugin.interpolate_grad_da(attr, rast, tri, dy, rast_db, dda,
                                                               diff_attrs_all, diff_attrs_list)
        return g_attr, g_rast, None, g_rast_db, None, None


class _interpolate_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attr, rast, tri):
        out, out_da = _plugin.interpolate_fwd(attr, rast, tri)
        ctx.save_for_backward(attr, rast, tri)
        return out, out_da

    @staticmethod
    def backward(ctx, dy, _):
        attr, rast, tri = ctx.saved_tensors
        g_attr, g_rast = _plugin.interpolate_grad(attr, rast, tri, dy)
        return g_attr, g_rast, None


def interpolate(attr, rast, tri, rast_db=None, diff_attrs=None):
    """Interpolate vertex attributes (reference ops.py:241-291).

    attr: [V,A] (range mode) or [N,V,A] / [1,V,A] (instanced, broadcast allowed);
    diff_attrs: None, 'all' or a list of attribute indices whose image-space derivatives
    are wanted (needs ``rast_db``).  Returns (out [N,H,W,A], out_da [N,H,W,2*len(diff_attrs)]);
    out_da has a zero-length last axis when no derivatives are requested.
    """
    if diff_attrs is None:
        diff_attrs = []
    elif diff_attrs != 'all':
        diff_attrs = np.asarray(diff_attrs, np.int32)
        assert len(diff_attrs.shape) == 1
        diff_attrs = diff_attrs.tolist()
    diff_attrs_all = int(diff_attrs == 'all')
    diff_attrs_list = [] if diff_attrs_all else diff_attrs

    assert all(isinstance(x, torch.Tensor) for x in (attr, rast, tri))
    if diff_attrs:
        assert isinstance(rast_db, torch.Tensor)
        return _interpolate_func_da.apply(attr, rast, tri, rast_db, diff_attrs_all, diff_attrs_list)
    return _interpolate_func.apply(attr, rast, tri)


# ----------------------------------------------------------------------------- legacy GL stub
# reference ops.py:550-559

class RasterizeGLContext(RasterizeCudaContext):
    def __init__(self, output_db=True, mode='automatic', device=None):
        warnings.warn("RasterizeGLContext has been deprecated and uses RasterizeCudaContext internally",
                      DeprecationWarning, stacklevel=2)
        super().__init__(device=device)

    def set_context(self):
        pass

    def release_context(self):
        pass
