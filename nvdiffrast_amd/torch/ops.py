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
    "texture", "texture_construct_mip", "antialias", "antialias_construct_topology_hash",
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


# ----------------------------------------------------------------------------- texture
# reference ops.py:298-465

class _texture_func_mip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, filter_mode, tex, uv, uv_da, mip_level_bias, mip_wrapper, filter_mode_enum, boundary_mode_enum, *mip_stack):
        empty = torch.tensor([])
        if uv_da is None:
            uv_da = empty
        if mip_level_bias is None:
            mip_level_bias = empty
        if mip_wrapper is None:
            mip_wrapper = _plugin.TextureMipWrapper()
        out = _plugin.texture_fwd_mip(tex, uv, uv_da, mip_level_bias, mip_wrapper, mip_stack, filter_mode_enum, boundary_mode_enum)
        ctx.save_for_backward(tex, uv, uv_da, mip_level_bias, *mip_stack)
        ctx.saved_misc = filter_mode, mip_wrapper, filter_mode_enum, boundary_mode_enum
        return out

    @staticmethod
    def backward(ctx, dy):
        tex, uv, uv_da, mip_level_bias, *mip_stack = ctx.saved_tensors
        filter_mode, mip_wrapper, filter_mode_enum, boundary_mode_enum = ctx.saved_misc
        if filter_mode == 'linear-mipmap-linear':
            g_tex, g_uv, g_uv_da, g_mip_level_bias, g_mip_stack = _plugin.texture_grad_linear_mipmap_linear(
                tex, uv, dy, uv_da, mip_level_bias, mip_wrapper, mip_stack, filter_mode_enum, boundary_mode_enum)
            return (None, g_tex, g_uv, g_uv_da, g_mip_level_bias, None, None, None) + tuple(g_mip_stack)
        else:  # linear-mipmap-nearest
            g_tex, g_uv, g_mip_stack = _plugin.texture_grad_linear_mipmap_nearest(
                tex, uv, dy, uv_da, mip_level_bias, mip_wrapper, mip_stack, filter_mode_enum, boundary_mode_enum)
            return (None, g_tex, g_uv, None, None, None, None, None) + tuple(g_mip_stack)


class _texture_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, filter_mode, tex, uv, filter_mode_enum, boundary_mode_enum):
        out = _plugin.texture_fwd(tex, uv, filter_mode_enum, boundary_mode_enum)
        ctx.save_for_backward(tex, uv)
        ctx.saved_misc = filter_mode, filter_mode_enum, boundary_mode_enum
        return out

    @staticmethod
    def backward(ctx, dy):
        tex, uv = ctx.saved_tensors
        filter_mode, filter_mode_enum, boundary_mode_enum = ctx.saved_misc
        if filter_mode == 'linear':
            g_tex, g_uv = _plugin.texture_grad_linear(tex, uv, dy, filter_mode_enum, boundary_mode_enum)
            return None, g_tex, g_uv, None, None
        else:  # nearest
            g_tex = _plugin.texture_grad_nearest(tex, uv, dy, filter_mode_enum, boundary_mode_enum)
            return None, g_tex, None, None, None


def texture(tex, uv, uv_da=None, mip_level_bias=None, mip=None, filter_mode='auto', boundary_mode='wrap', max_mip_level=None):
    """Texture sampling (reference ops.py:345-439).

    tex: [N or 1, Ht, Wt, C] float32, or a cube map [N or 1, 6, S, S, C] with boundary_mode='cube';
    uv: [N,H,W,2] (cube: direction vectors [N,H,W,3]); uv_da: optional image-space derivatives of uv,
    [N,H,W,4] (cube: [N,H,W,6]);
    mip_level_bias: optional [N,H,W]; mip: a ``texture_construct_mip()`` result or a list of tensors
    (custom mip stack, levels 1..L, which then receive their own gradients); filter_mode: 'auto',
    'nearest', 'linear', 'linear-mipmap-nearest', 'linear-mipmap-linear' ('auto' = trilinear when
    uv_da or mip_level_bias is given, else 'linear'); boundary_mode: 'wrap', 'clamp', 'zero', 'cube';
    max_mip_level limits the mip chain.  Returns [N,H,W,C].
    """
    if filter_mode == 'auto':
        filter_mode = 'linear-mipmap-linear' if (uv_da is not None or mip_level_bias is not None) else 'linear'
    if max_mip_level is None:
        max_mip_level = -1
    else:
        max_mip_level = int(max_mip_level)
        assert max_mip_level >= 0
    assert isinstance(tex, torch.Tensor) and isinstance(uv, torch.Tensor)
    if 'mipmap' in filter_mode:
        assert isinstance(uv_da, torch.Tensor) or isinstance(mip_level_bias, torch.Tensor)
    if max_mip_level == 0 and filter_mode in ['linear-mipmap-nearest', 'linear-mipmap-linear']:
        filter_mode = 'linear'
    filter_mode_dict = {'nearest': 0, 'linear': 1, 'linear-mipmap-nearest': 2, 'linear-mipmap-linear': 3}
    filter_mode_enum = filter_mode_dict[filter_mode]
    boundary_mode_dict = {'cube': 0, 'wrap': 1, 'clamp': 2, 'zero': 3}
    boundary_mode_enum = boundary_mode_dict[boundary_mode]
    if 'mipmap' in filter_mode:
        mip_wrapper, mip_stack = None, []
        if mip is not None:
            assert isinstance(mip, (_plugin.TextureMipWrapper, list))
            if isinstance(mip, list):
                assert all(isinstance(x, torch.Tensor) for x in mip)
                mip_stack = mip
            else:
                mip_wrapper = mip
        else:
            mip_wrapper = _plugin.texture_construct_mip(tex, max_mip_level, boundary_mode == 'cube')
    if filter_mode == 'linear-mipmap-linear' or filter_mode == 'linear-mipmap-nearest':
        return _texture_func_mip.apply(filter_mode, tex, uv, uv_da, mip_level_bias, mip_wrapper,
                                       filter_mode_enum, boundary_mode_enum, *mip_stack)
    return _texture_func.apply(filter_mode, tex, uv, filter_mode_enum, boundary_mode_enum)


def texture_construct_mip(tex, max_mip_level=None, cube_mode=False):
    """Build the mip stack of a constant texture once (reference ops.py:442-465); pass the result as ``mip=``."""
    assert isinstance(tex, torch.Tensor)
    assert cube_mode is True or cube_mode is False
    if max_mip_level is None:
        max_mip_level = -1
    else:
        max_mip_level = int(max_mip_level)
        assert max_mip_level >= 0
    return _plugin.texture_construct_mip(tex, max_mip_level, cube_mode)


# ----------------------------------------------------------------------------- antialias
# reference ops.py:471-544

class _antialias_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, rast, pos, tri, topology_hash, pos_gradient_boost):
        out, work_buffer = _plugin.antialias_fwd(color, rast, pos, tri, topology_hash)
        ctx.save_for_backward(color, rast, pos, tri)
        ctx.saved_misc = pos_gradient_boost, work_buffer
        return out

    @staticmethod
    def backward(ctx, dy):
        color, rast, pos, tri = ctx.saved_tensors
        pos_gradient_boost, work_buffer = ctx.saved_misc
        g_color, g_pos = _plugin.antialias_grad(color, rast, pos, tri, dy, work_buffer)
        if pos_gradient_boost != 1.0:
            g_pos = g_pos * pos_gradient_boost
        return g_color, None, g_pos, None, None, None


def antialias(color, rast, pos, tri, topology_hash=None, pos_gradient_boost=1.0):
    """Silhouette antialiasing (reference ops.py:489-526).

    color: [N,H,W,C]; rast: main output of ``rasterize()``; pos, tri: as given to ``rasterize()``.
    A vertex shared by several triangles must use one index everywhere, otherwise its edges count as
    silhouettes.  ``topology_hash``: optional ``antialias_construct_topology_hash(tri)`` result;
    ``pos_gradient_boost`` scales the gradient that reaches ``pos``.  Returns [N,H,W,C].
    """
    assert all(isinstance(x, torch.Tensor) for x in (color, rast, pos, tri))
    if topology_hash is not None:
        assert isinstance(topology_hash, _plugin.TopologyHashWrapper)
    else:
        topology_hash = _plugin.antialias_construct_topology_hash(tri)
    return _antialias_func.apply(color, rast, pos, tri, topology_hash, pos_gradient_boost)


def antialias_construct_topology_hash(tri):
    """Build the topology hash of a constant triangle tensor once (reference ops.py:529-544)."""
    assert isinstance(tri, torch.Tensor)
    return _plugin.antialias_construct_topology_hash(tri)


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
