"""Operator layer of ``nvdiffrast_amd.torch``: the public functions and classes of the reference's
``nvdiffrast/torch/ops.py`` (names, argument order, defaults, return arity, error behaviour) on top of the
MI355X plugin (``_plugin``, the stand-in for the pybind module ``_nvdiffrast_c``).

The interface is the reference's; the construction is this package's own.  All four differentiable ops go
through ONE ``torch.autograd.Function`` (``_Dispatch``) driven by small stateless descriptors
(``_RasterizeOp``, ``_InterpolateOp``, ``_TextureOp``, ``_AntialiasOp``) that say which plugin entry points
compute the forward and the gradients and which inputs those gradients belong to; argument checking and mode
resolution are table driven (``_FILTER_MODES``, ``_BOUNDARY_MODES``) and shared (``_tensors``, ``_as_ranges``,
``_mip_limit``).  Behaviour that scripts written for the reference rely on is kept deliberately, including its
quirks -- they are listed where they occur, with the reference line they come from.

``tests/test_capi_exports.py`` compares every public signature with the reference's file, and
``tests/test_gpu_reference_ops.py`` runs the reference's own ``ops.py`` on ``_plugin`` next to this module.
"""
import warnings
import weakref

import numpy as np
import torch

from . import _plugin

__all__ = [
    "RasterizeCudaContext", "RasterizeGLContext", "DepthPeeler",
    "get_log_level", "set_log_level",
    "rasterize", "interpolate", "texture", "texture_construct_mip",
    "antialias", "antialias_construct_topology_hash",
]

# ------------------------------------------------------------------------------------------------
# Logging passthrough (reference ops.py:18-41): levels are c10's, 0 INFO .. 3 FATAL, default 1.

def get_log_level():
    """Current log level of the native library (0 = INFO, 1 = WARNING (default), 2 = ERROR, 3 = FATAL)."""
    return _plugin.get_log_level()


def set_log_level(level):
    """Set the log level; 0 makes the rasterizer report when its internal buffers grow."""
    _plugin.set_log_level(level)


# ------------------------------------------------------------------------------------------------
# Shared argument handling.

def _tensors(**named):
    """Every argument must be a torch.Tensor; AssertionError otherwise, as the reference's asserts."""
    for name, value in named.items():
        assert isinstance(value, torch.Tensor), f"{name} must be a torch.Tensor"


_NO_RANGES = None


def _as_ranges(ranges):
    """Range-mode table, or the empty CPU [0,2] int32 tensor the plugin expects in instanced mode (ops.py:125-126)."""
    if ranges is None:
        global _NO_RANGES
        if _NO_RANGES is None:
            _NO_RANGES = torch.empty(size=(0, 2), dtype=torch.int32, device="cpu")
        return _NO_RANGES
    _tensors(ranges=ranges)
    return ranges


def _mip_limit(max_mip_level):
    """None -> -1 (no limit); otherwise a non-negative int (ops.py:399-403, 458-462)."""
    if max_mip_level is None:
        return -1
    level = int(max_mip_level)
    assert level >= 0
    return level


_FILTER_MODES = {"nearest": 0, "linear": 1, "linear-mipmap-nearest": 2, "linear-mipmap-linear": 3}   # ops.py:415
_BOUNDARY_MODES = {"cube": 0, "wrap": 1, "clamp": 2, "zero": 3}                                        # ops.py:417
_MIPMAPPED = ("linear-mipmap-nearest", "linear-mipmap-linear")


# ------------------------------------------------------------------------------------------------
# One autograd node for every op.

class _Dispatch(torch.autograd.Function):
    """forward(op, *args) runs ``op.forward``; backward hands the upstream gradients to ``op.backward`` and
    returns one gradient slot per forward argument (plus None for the descriptor itself)."""

    @staticmethod
    def forward(ctx, op, *args):
        outputs, keep, state = op.forward(*args)
        ctx.op, ctx.state, ctx.arity = op, state, len(args)
        ctx.save_for_backward(*keep)
        # The gradient of an output nobody used arrives as None instead of a materialised zero tensor: for
        # rasterize that is 32 B/pixel of zeros the reference writes and reads back for nothing (its rast_db
        # gradient when only `rast` is consumed).  Descriptors treat None as "no contribution".
        ctx.set_materialize_grads(False)
        return outputs

    @staticmethod
    def backward(ctx, *upstream):
        if all(g is None for g in upstream):
            return (None,) * (ctx.arity + 1)
        grads = ctx.op.backward(ctx.state, ctx.saved_tensors, *upstream)
        assert len(grads) == ctx.arity
        return (None,) + tuple(grads)


_LazyGrad, _LazySource = _plugin._LazyGrad, _plugin._LazySource     # (the stand-in for a gradient nobody may look at: _plugin.py)


class _RasterOrigin:
    """What a `rast` tensor remembers about the rasterize call that produced it, for the fused backward pass.

    The metric's graph is rasterize -> interpolate.  Its backward is two kernels over the same pixels; the library has
    one kernel that does both (csrc/backward_fused.hip, `_plugin.interpolate_rasterize_grad`).  Whether the fused result
    may be USED is only known when autograd delivers rast's gradient to the rasterize node: if interpolate was the sole
    contributor, the object that arrives is the very g_rast interpolate returned; if anything else contributed (antialias does
    not, but user code such as a mask made from rast does), autograd has summed the contributions into another tensor.  So
    interpolate's backward computes g_attr AND the position gradient in one pass, returns as g_rast a stand-in that computes
    the real values only if somebody looks at them (_LazyGrad: the fused kernel does not write g_rast) -- always a correct
    gradient -- and leaves the position gradient here; rasterize's backward takes it when the object it receives is that
    stand-in, and otherwise computes the gradient from what it did receive, as if nothing had been prepared (the context then
    stops preparing: `fused_disabled`).  `pending` refers to the stand-ins weakly: a backward pass that never reaches the
    rasterize node (autograd.grad(..., inputs=[attr])) leaves nothing behind that pins the upstream gradient."""
    __slots__ = ("pos", "tri", "state", "rast_ptr", "rast_version", "rast_shape", "db_ptr", "db_version", "grad_db",
                 "interpolations", "pending", "flags")

    def __init__(self, pos, tri, state, rast, flags, rast_db, grad_db):
        self.pos, self.tri, self.state = pos, tri, state
        self.rast_ptr, self.rast_version, self.rast_shape = rast.data_ptr(), rast._version, tuple(rast.shape)
        self.db_ptr, self.db_version, self.grad_db = rast_db.data_ptr(), rast_db._version, bool(grad_db)
        self.interpolations = 0            # interpolate() calls that took this rast
        self.pending = None                # (weak ref to the g_rast stand-in, g_pos, weak ref to the g_rast_db stand-in or None) between the two backward nodes
        self.flags = flags                 # tile occupancy of this rast (one byte per 8x8 tile), written by the rasterizer

    def flags_for(self, rast):
        """The occupancy flags, if `rast` still is what rasterize() returned (same storage, never written to since):
        the kernels that read rast then skip the tiles without any triangle (include/nvdr_hip.h `tile_flags`)."""
        if rast.data_ptr() == self.rast_ptr and rast._version == self.rast_version and tuple(rast.shape) == self.rast_shape:
            return self.flags
        return None

    def usable_by(self, attr, rast, tri, rast_db=None):
        """interpolate(attr, rast, tri[, rast_db]) may prepare the position gradient: this very rast (and rast_db),
        untouched, the same triangle tensor (pose-style scripts interpolate with another index buffer), one vertex set,
        nobody else doing the same."""
        st = self.state
        return (_plugin.fused_backward_mode() == "auto" and st.fused_disabled != _plugin.fused_backward_epoch() and self.pending is None
                and self.interpolations == 1 and rast.requires_grad and self.pos.requires_grad
                and rast.data_ptr() == self.rast_ptr and rast._version == self.rast_version and tuple(rast.shape) == self.rast_shape
                and (rast_db is None or (rast_db.data_ptr() == self.db_ptr and rast_db._version == self.db_version))
                and tri.data_ptr() == self.tri.data_ptr() and tri.shape == self.tri.shape
                and attr.shape[-2] == self.pos.shape[-2])


class _ZeroTiles:
    """Carried by interpolate()'s outputs: `flags` (the rasterizer's tile occupancy) marks 8x8 tiles in which this tensor is
    zero.  Valid while the tensor is untouched (same storage, same version counter)."""
    __slots__ = ("flags", "ptr", "version", "shape")

    def __init__(self, flags, t):
        self.flags, self.ptr, self.version, self.shape = flags, t.data_ptr(), t._version, tuple(t.shape[:3])

    def still(self, t):
        """`t` is the tensor this record was made for, unchanged."""
        return t.data_ptr() == self.ptr and t._version == self.version and tuple(t.shape[:3]) == self.shape

    @staticmethod
    def records(uv, uv_da):
        """The records behind of(uv, uv_da), for the backward pass to check against its saved tensors (which need not be
        the same Python objects any more)."""
        zd = getattr(uv_da, "_nvdr_zero_tiles", None) if uv_da is not None and uv_da.numel() else None
        return getattr(uv, "_nvdr_zero_tiles", None), zd

    @staticmethod
    def from_registry(uv, uv_da):
        """The same answer from the records kept by storage (interpolate() served by the compiled host layer leaves no
        attribute on its outputs): _plugin.flags_of."""
        f = _plugin.flags_of(uv, "zero")
        if f is not None and uv_da is not None and uv_da.numel():
            fd = _plugin.flags_of(uv_da, "zero")
            if fd is None or fd.data_ptr() != f.data_ptr():
                return None
        return f

    @staticmethod
    def of(uv, uv_da):
        """The flags that texture(uv, uv_da) may use: both tensors (uv_da may be absent) zero on the same empty tiles."""
        z = getattr(uv, "_nvdr_zero_tiles", None)
        if z is None:
            return _ZeroTiles.from_registry(uv, uv_da)
        if uv.data_ptr() != z.ptr or uv._version != z.version or tuple(uv.shape[:3]) != z.shape:
            return None
        if uv_da is not None and uv_da.numel():
            zd = getattr(uv_da, "_nvdr_zero_tiles", None)
            if (zd is None or zd.flags is not z.flags or uv_da.data_ptr() != zd.ptr or uv_da._version != zd.version
                    or tuple(uv_da.shape[:3]) != zd.shape):
                return None
        return z.flags


class _RasterizeOp:
    """args: context, pos, tri, resolution, ranges, grad_db, peeling_idx -> (rast, rast_db); gradient to pos only."""

    @staticmethod
    def forward(raster_ctx, pos, tri, resolution, ranges, grad_db, peeling_idx):
        state = raster_ctx.cpp_wrapper
        rast, rast_db = _plugin.rasterize_fwd_cuda(state, pos, tri, resolution, ranges, peeling_idx)
        origin = rast._nvdr_origin = _RasterOrigin(pos, tri, state, rast, state.last_flags, rast_db, grad_db)
        return (rast, rast_db), (pos, tri, rast), (bool(grad_db), origin)

    @staticmethod
    def backward(state, saved, d_rast, d_rast_db):
        grad_db, origin = state
        pos, tri, rast = saved
        if origin is not None and origin.pending is not None:
            (lz_rast, g_pos, lz_db), origin.pending = origin.pending, None
            lz_rast, lz_db = lz_rast(), (None if lz_db is None else lz_db())
            # the prepared gradient stands if what arrives for rast is interpolate's own (unwritten) g_rast -- the very object:
            # nobody added to it, no hook replaced it -- and what arrives for rast_db is nothing (plain interpolation),
            # irrelevant (grad_db=False) or interpolate's own g_rast_db
            db_ok = (d_rast_db is None and lz_db is None) or not grad_db or (lz_db is not None and d_rast_db is lz_db)
            # ... and nobody has EDITED it on the way: a hook may have changed the gradient in place -- the object stays the same --
            # so the prepared gradient stands only while the stand-in's version counter is where it was when it was made
            untouched = lz_rast is not None and lz_rast.unedited() and (lz_db is None or lz_db.unedited())
            if d_rast is not None and d_rast is lz_rast and db_ok and untouched:
                _plugin.fused_backward_count("used")
                return None, g_pos, None, None, None, None, None
            # rast's gradient has other contributors in this program: what was prepared is void, and preparing it again
            # on this context would be wasted work every step
            _plugin.fused_backward_count("discarded")
            if origin.state.fused_disabled != _plugin.fused_backward_epoch():
                _plugin._log_info("fused rasterize/interpolate backward switched off on this context: rast's gradient has other "
                                  "contributors (set_fused_backward('auto') re-arms it)")
            origin.state.fused_disabled = _plugin.fused_backward_epoch()
        if isinstance(d_rast, _LazyGrad):
            d_rast = d_rast.materialize()
        if isinstance(d_rast_db, _LazyGrad):
            d_rast_db = d_rast_db.materialize()
        if d_rast is None:
            if not grad_db:
                return (None,) * 7
            d_rast = torch.zeros_like(rast)
        flags = origin.flags_for(rast)
        if grad_db and d_rast_db is not None:
            g_pos = _plugin.rasterize_grad_db(pos, tri, rast, d_rast, d_rast_db, tile_flags=flags)
        else:
            g_pos = _plugin.rasterize_grad(pos, tri, rast, d_rast, tile_flags=flags)
        return None, g_pos, None, None, None, None, None


class _InterpolateOp:
    """args: attr, rast, tri, rast_db or None, diff_all, diff_list -> (out, out_da)."""

    @staticmethod
    def forward(attr, rast, tri, rast_db, diff_all, diff_list):
        with_da = rast_db is not None
        origin = getattr(rast, "_nvdr_origin", None)     # set by rasterize() on its own output (fused backward, tile flags)
        flags = None if origin is None else origin.flags_for(rast)
        if with_da:
            outs = _plugin.interpolate_fwd_da(attr, rast, tri, rast_db, diff_all, diff_list, tile_flags=flags)
            keep = (attr, rast, tri, rast_db)
        else:
            outs = _plugin.interpolate_fwd(attr, rast, tri, tile_flags=flags)
            keep = (attr, rast, tri)
        if origin is not None:
            origin.interpolations += 1
        if flags is not None:
            # interpolate writes zeros where no triangle is visible: the rasterizer's empty tiles are tiles of zeros in both
            # outputs, and texture() may skip reading them there (as long as the tensors stay what they are now)
            for t in outs:
                if t.numel():
                    t._nvdr_zero_tiles = _ZeroTiles(flags, t)
        return tuple(outs), keep, (with_da, diff_all, diff_list, origin)

    @staticmethod
    def _plain_grad(attr, rast, tri, d_out, origin):
        """Gradient without pixel differentials; with the position gradient prepared in the same pass when the rast
        came straight from rasterize() and this is its only interpolation (see _RasterOrigin)."""
        flags = None if origin is None else origin.flags_for(rast)
        if origin is not None and origin.pending is not None:
            origin.pending = None          # left over from a backward pass that never reached the rasterize node
            #                                (autograd.grad(inputs=[attr]), an exception): void, and must not block or pin memory
        if origin is not None and origin.usable_by(attr, rast, tri):
            g_attr, _, _, g_pos = _plugin.interpolate_rasterize_grad(attr, rast, tri, origin.pos, d_out, with_g_rast=False, tile_flags=flags)
            # g_rast itself is not written: autograd gets a stand-in that computes it if anybody looks (_LazyGrad)
            source = _LazySource(lambda: (_plugin.interpolate_grad(attr, rast, tri, d_out, tile_flags=flags, fuse=False)[1],), (attr, rast, tri, d_out))
            g_rast = _LazyGrad(rast, source, 0)
            origin.pending = (weakref.ref(g_rast), g_pos, None)
            return g_attr, g_rast
        return _plugin.interpolate_grad(attr, rast, tri, d_out, tile_flags=flags, fuse=False)

    @staticmethod
    def backward(state, saved, d_out, d_out_da):
        with_da, diff_all, diff_list, origin = state
        if d_out is None:
            d_out = torch.zeros(tuple(saved[1].shape[:3]) + (saved[0].shape[-1],), dtype=saved[0].dtype, device=saved[0].device)
        if with_da and d_out_da is None:                 # differentials computed but unused: the plain gradient is the same
            attr, rast, tri, _rast_db = saved
            g_attr, g_rast = _InterpolateOp._plain_grad(attr, rast, tri, d_out, origin)
            return g_attr, g_rast, None, None, None, None
        if with_da:
            attr, rast, tri, rast_db = saved
            flags = None if origin is None else origin.flags_for(rast)
            if origin is not None and origin.pending is not None:
                origin.pending = None      # (stale, as in _plain_grad)
            if origin is not None and origin.usable_by(attr, rast, tri, rast_db):
                # config 3's pair: interpolate_grad_da + rasterize_grad_db in one pass (see _RasterOrigin)
                g_attr, _, _, g_pos = _plugin.interpolate_rasterize_grad(
                    attr, rast, tri, origin.pos, d_out, with_g_rast=False, tile_flags=flags, rast_db=rast_db, dda=d_out_da,
                    diff_attrs_all=diff_all, diff_attrs_vec=diff_list, db_to_pos=origin.grad_db)
                source = _LazySource(lambda: _plugin.interpolate_grad_da(attr, rast, tri, d_out, rast_db, d_out_da, diff_all, diff_list,
                                                                         tile_flags=flags, fuse=False)[1:], (attr, rast, tri, d_out, rast_db, d_out_da))
                g_rast, g_rast_db = _LazyGrad(rast, source, 0), _LazyGrad(rast_db, source, 1)
                origin.pending = (weakref.ref(g_rast), g_pos, weakref.ref(g_rast_db))
                return g_attr, g_rast, None, g_rast_db, None, None
            g_attr, g_rast, g_rast_db = _plugin.interpolate_grad_da(attr, rast, tri, d_out, rast_db, d_out_da, diff_all, diff_list,
                                                                    tile_flags=flags, fuse=False)
            return g_attr, g_rast, None, g_rast_db, None, None
        attr, rast, tri = saved
        g_attr, g_rast = _InterpolateOp._plain_grad(attr, rast, tri, d_out, origin)
        return g_attr, g_rast, None, None, None, None


class _TextureOp:
    """args: filter_mode, boundary id, tex, uv, uv_da, mip_level_bias, mip wrapper, *custom mip levels -> out.
    The four gradient entry points of the plugin are selected by filter mode; custom mip levels receive their own
    gradients (trailing slots)."""

    @staticmethod
    def forward(filter_mode, boundary, tex, uv, uv_da, mip_level_bias, mip_wrapper, *mip_stack):
        f = _FILTER_MODES[filter_mode]
        zf = None if boundary == _BOUNDARY_MODES["cube"] else _ZeroTiles.of(uv, uv_da)      # tiles of known-zero uv / uv_da
        zrec = _ZeroTiles.records(uv, uv_da) if zf is not None else None
        if zrec is not None and zrec[0] is None:
            zrec = None                                      # (flags from the registry: the backward pass asks it again)
        if filter_mode in _MIPMAPPED:
            # absent optional tensors travel as empty tensors, an absent wrapper as an empty one (ops.py:301-307)
            placeholder = torch.tensor([])
            uv_da = placeholder if uv_da is None else uv_da
            mip_level_bias = placeholder if mip_level_bias is None else mip_level_bias
            mip_wrapper = _plugin.TextureMipWrapper() if mip_wrapper is None else mip_wrapper
            out = _plugin.texture_fwd_mip(tex, uv, uv_da, mip_level_bias, mip_wrapper, mip_stack, f, boundary, tile_flags=zf)
            keep = (tex, uv, uv_da, mip_level_bias) + tuple(mip_stack)
        else:
            out = _plugin.texture_fwd(tex, uv, f, boundary, tile_flags=zf)
            keep = (tex, uv)
        return out, keep, (filter_mode, f, boundary, mip_wrapper, len(mip_stack), zf, zrec)

    @staticmethod
    def backward(state, saved, d_out):
        filter_mode, f, boundary, mip_wrapper, n_custom, zf, zrec = state
        g_uv = g_uv_da = g_bias = None
        g_levels = (None,) * n_custom
        if zf is not None and zrec is None:
            now = _ZeroTiles.from_registry(saved[1], saved[2] if filter_mode in _MIPMAPPED else None)
            if now is None or now.data_ptr() != zf.data_ptr():
                zf = None                                     # uv / uv_da were written to since the forward pass
        elif zf is not None and not (zrec[0].still(saved[1])
                                     and (zrec[1] is None or filter_mode not in _MIPMAPPED or zrec[1].still(saved[2]))):
            zf = None                                         # uv / uv_da were written to since the forward pass
        if filter_mode in _MIPMAPPED:
            tex, uv, uv_da, bias = saved[:4]
            stack = list(saved[4:])
            if filter_mode == "linear-mipmap-linear":
                g_tex, g_uv, g_uv_da, g_bias, g_stack = _plugin.texture_grad_linear_mipmap_linear(
                    tex, uv, d_out, uv_da, bias, mip_wrapper, stack, f, boundary, tile_flags=zf)
            else:
                g_tex, g_uv, g_stack = _plugin.texture_grad_linear_mipmap_nearest(
                    tex, uv, d_out, uv_da, bias, mip_wrapper, stack, f, boundary, tile_flags=zf)
            g_levels = tuple(g_stack)
        else:
            tex, uv = saved
            if filter_mode == "linear":
                g_tex, g_uv = _plugin.texture_grad_linear(tex, uv, d_out, f, boundary, tile_flags=zf)
            else:
                g_tex = _plugin.texture_grad_nearest(tex, uv, d_out, f, boundary, tile_flags=zf)
        return (None, None, g_tex, g_uv, g_uv_da, g_bias, None) + g_levels


class _AntialiasOp:
    """args: color, rast, pos, tri, topology hash, pos_gradient_boost -> out.  The work buffer written by the
    forward pass is replayed by the gradient pass (ops.py:476, torch_antialias.cpp:223)."""

    @staticmethod
    def forward(color, rast, pos, tri, topology_hash, pos_gradient_boost):
        origin = getattr(rast, "_nvdr_origin", None)
        out, work_buffer = _plugin.antialias_fwd(color, rast, pos, tri, topology_hash,
                                                 tile_flags=None if origin is None else origin.flags_for(rast))
        return out, (color, rast, pos, tri), (pos_gradient_boost, work_buffer)

    @staticmethod
    def backward(state, saved, d_out):
        boost, work_buffer = state
        color, rast, pos, tri = saved
        g_color, g_pos = _plugin.antialias_grad(color, rast, pos, tri, d_out, work_buffer)
        if boost != 1.0:
            g_pos = g_pos * boost
        return g_color, None, g_pos, None, None, None


# ------------------------------------------------------------------------------------------------
# Rasterizer contexts and rasterize().

class RasterizeCudaContext:
    """Rasterizer state for one GPU (reference ops.py:47-68).

    Owns the native rasterizer's scratch memory.  A context belongs to the device it was created for
    (``device`` or, when None, the current one) and must not be used from two threads at once."""

    def __init__(self, device=None):
        if device is None:
            index = torch.cuda.current_device()
        else:
            with torch.cuda.device(device):
                index = torch.cuda.current_device()
        self.cpp_wrapper = _plugin.RasterizeCRStateWrapper(index)
        self.active_depth_peeler = None


class RasterizeGLContext(RasterizeCudaContext):
    """Deprecated alias kept for old scripts (reference ops.py:550-559): an OpenGL context never existed here
    either; ``output_db`` and ``mode`` are accepted and ignored, ``set_context`` / ``release_context`` do nothing."""

    def __init__(self, output_db=True, mode="automatic", device=None):
        warnings.warn("RasterizeGLContext has been deprecated and uses RasterizeCudaContext internally", DeprecationWarning, stacklevel=2)
        super().__init__(device=device)

    def set_context(self):
        pass

    def release_context(self):
        pass


def _raster_request(glctx, pos, tri, resolution, ranges, grad_db):
    """Validation shared by rasterize() and DepthPeeler (ops.py:118-126, 150-158)."""
    assert isinstance(glctx, RasterizeCudaContext)
    assert grad_db is True or grad_db is False
    _tensors(pos=pos, tri=tri)
    return tuple(resolution), _as_ranges(ranges)


def rasterize(glctx, pos, tri, resolution, ranges=None, grad_db=True):
    """Rasterize triangles.

    Args:
        glctx: a ``RasterizeCudaContext``.
        pos: clip-space vertex positions, float32 on the GPU: ``[minibatch, num_vertices, 4]`` (instanced mode)
            or ``[num_vertices, 4]`` (range mode).
        tri: ``[num_triangles, 3]`` int32 vertex indices on the GPU.
        resolution: ``(height, width)`` of the output.
        ranges: range mode only -- int32 CPU tensor ``[minibatch, 2]`` of (first triangle, triangle count).
        grad_db: propagate gradients of the second output to ``pos``.

    Returns:
        ``(rast, rast_db)``, both ``[minibatch, height, width, 4]`` float32: ``rast`` = (u, v, z/w, triangle_id + 1
        or 0 for background), ``rast_db`` = (du/dX, du/dY, dv/dX, dv/dY).  Row 0 is the bottom scan line.

    Reference quirk kept: while a ``DepthPeeler`` is active on ``glctx`` this RETURNS (does not raise) a
    ``RuntimeError`` instance (ops.py:131-132).
    """
    resolution, ranges = _raster_request(glctx, pos, tri, resolution, ranges, grad_db)
    if glctx.active_depth_peeler is not None:
        return RuntimeError("Cannot call rasterize() during depth peeling operation, use rasterize_next_layer() instead")
    return _rasterize_layer(glctx, pos, tri, resolution, ranges, grad_db, -1)


def _rasterize_layer(glctx, pos, tri, resolution, ranges, grad_db, peeling_idx):
    """One rasterizer pass: the compiled host layer (csrc_host/nvdr_torch_host.cpp: validation, allocation, launch and the
    autograd node in C++) when it is there and takes the call, else the Python one (_plugin: rare modes, every error message)."""
    host = _plugin.host_layer()
    if host is not None and len(resolution) == 2:
        state = glctx.cpp_wrapper.host_state(host)
        served = host.rasterize(state, pos, tri, int(resolution[0]), int(resolution[1]), ranges, grad_db, peeling_idx)
        if served is not None:
            if peeling_idx >= 0:
                glctx.cpp_wrapper.depth, glctx.cpp_wrapper.peel = state.depth, state.peel      # (where tools and tests look for them)
            return served
    return _Dispatch.apply(_RasterizeOp, glctx, pos, tri, resolution, ranges, grad_db, peeling_idx)


class DepthPeeler:
    """Context manager that peels depth layers front to back (reference ops.py:141-204)::

        with DepthPeeler(glctx, pos, tri, resolution) as peeler:
            for _ in range(num_layers):
                rast, rast_db = peeler.rasterize_next_layer()

    Only one peeler can be active per context, and a peeler cannot be re-entered after it has exited."""

    _FIELDS = ("raster_ctx", "pos", "tri", "resolution", "ranges", "grad_db", "peeling_idx")

    def __init__(self, glctx, pos, tri, resolution, ranges=None, grad_db=True):
        resolution, ranges = _raster_request(glctx, pos, tri, resolution, ranges, grad_db)
        self.raster_ctx, self.pos, self.tri = glctx, pos, tri
        self.resolution, self.ranges, self.grad_db = resolution, ranges, grad_db
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
        for field in self._FIELDS:                       # drop every reference to the inputs (ops.py:183-190)
            setattr(self, field, None)
        return None

    def rasterize_next_layer(self):
        """Next depth layer: same outputs as ``rasterize()``; pixels no deeper than the previous layer are hidden."""
        assert self.raster_ctx.active_depth_peeler is self
        assert self.peeling_idx >= 0
        layer = self.peeling_idx
        self.peeling_idx += 1
        return _rasterize_layer(self.raster_ctx, self.pos, self.tri, self.resolution, self.ranges, self.grad_db, layer)


# ------------------------------------------------------------------------------------------------
# interpolate()

def interpolate(attr, rast, tri, rast_db=None, diff_attrs=None):
    """Interpolate vertex attributes over the pixels of a rasterized image.

    Args:
        attr: ``[num_vertices, num_attributes]`` (range mode) or ``[minibatch, num_vertices, num_attributes]``
            (instanced; a minibatch of 1 is broadcast) float32 on the GPU.
        rast: first output of ``rasterize()``.
        tri: the triangle tensor used for rasterization.
        rast_db: second output of ``rasterize()``; needed for attribute pixel differentials.
        diff_attrs: attribute indices to differentiate with respect to the pixel position, or ``'all'``.

    Returns:
        ``(out, out_da)``: ``[minibatch, height, width, num_attributes]`` and
        ``[minibatch, height, width, 2 * len(diff_attrs)]`` as (dA/dX, dA/dY) pairs -- last dimension 0 when no
        differentials were requested.
    """
    want_all = isinstance(diff_attrs, str) and diff_attrs == "all"
    if diff_attrs is None:
        selected = []
    elif want_all:
        selected = []
    else:
        arr = np.asarray(diff_attrs, np.int32)
        assert len(arr.shape) == 1
        selected = arr.tolist()
    _tensors(attr=attr, rast=rast, tri=tri)
    with_da = bool(want_all or selected)
    if with_da:
        _tensors(rast_db=rast_db)
    host = _plugin.host_layer()
    if host is not None:                                    # (see _rasterize_layer)
        served = host.interpolate(attr, rast, tri, rast_db if with_da else None, want_all, selected)
        if served is not None:
            return served
    if with_da:
        return _Dispatch.apply(_InterpolateOp, attr, rast, tri, rast_db, int(want_all), selected)
    return _Dispatch.apply(_InterpolateOp, attr, rast, tri, None, 0, [])


# ------------------------------------------------------------------------------------------------
# texture()

def _resolve_filter_mode(filter_mode, uv_da, mip_level_bias, max_mip_level):
    """'auto' picks the best mode the inputs allow (ops.py:395-396); a mip chain limited to level 0 is plain
    bilinear filtering (ops.py:411-412)."""
    if filter_mode == "auto":
        filter_mode = "linear-mipmap-linear" if (uv_da is not None or mip_level_bias is not None) else "linear"
    if "mipmap" in filter_mode:
        assert isinstance(uv_da, torch.Tensor) or isinstance(mip_level_bias, torch.Tensor)
    if max_mip_level == 0 and filter_mode in _MIPMAPPED:
        filter_mode = "linear"
    return filter_mode


def texture(tex, uv, uv_da=None, mip_level_bias=None, mip=None, filter_mode="auto", boundary_mode="wrap", max_mip_level=None):
    """Sample a texture.

    Args:
        tex: ``[minibatch, tex_height, tex_width, channels]`` or, for cube maps (``boundary_mode='cube'``),
            ``[minibatch, 6, size, size, channels]``; float32 on the GPU; a minibatch of 1 is broadcast.
        uv: ``[minibatch, height, width, 2]`` texture coordinates (cube maps: ``[..., 3]`` direction vectors).
        uv_da: pixel differentials of ``uv`` (last dimension twice that of ``uv``), selects the mip level.
        mip_level_bias: ``[minibatch, height, width]`` added to the mip level; alone, it IS the level.
        mip: a stack from ``texture_construct_mip()``, or a list of tensors (levels 1.. of a custom stack, which
            then receive their own gradients instead of passing them on to ``tex``).  Built internally when a
            mipmapped mode needs one and none is given.
        filter_mode: ``'auto'``, ``'nearest'``, ``'linear'``, ``'linear-mipmap-nearest'`` or ``'linear-mipmap-linear'``.
        boundary_mode: ``'wrap'``, ``'clamp'``, ``'zero'`` or ``'cube'``.
        max_mip_level: limit on the number of mip levels built and used.

    Returns:
        ``[minibatch, height, width, channels]``.  Invalid cube-map directions (e.g. zero vectors) give zeros
        and no gradients.
    """
    limit = _mip_limit(max_mip_level)
    _tensors(tex=tex, uv=uv)
    filter_mode = _resolve_filter_mode(filter_mode, uv_da, mip_level_bias, limit)
    _FILTER_MODES[filter_mode]                              # KeyError for an unknown mode, as the reference's dict lookup
    boundary = _BOUNDARY_MODES[boundary_mode]
    host = _plugin.host_layer()                             # (see _rasterize_layer)
    if filter_mode not in _MIPMAPPED:
        if host is not None:
            served = host.texture(tex, uv, None, None, None, 0, [], False, _FILTER_MODES[filter_mode], boundary, _plugin._TEX_GRAD_SCRATCH)
            if served is not None:
                return served
        return _Dispatch.apply(_TextureOp, filter_mode, boundary, tex, uv, None, None, None)
    wrapper, levels = None, []
    if mip is None:
        wrapper = _plugin.texture_construct_mip(tex, limit, boundary_mode == "cube")
    else:
        assert isinstance(mip, (_plugin.TextureMipWrapper, list))
        if isinstance(mip, list):
            assert all(isinstance(level, torch.Tensor) for level in mip)
            levels = mip
        else:
            wrapper = mip
    if host is not None and not levels and wrapper.mip is not None:
        served = host.texture(tex, uv, uv_da, mip_level_bias, wrapper.mip, wrapper.max_mip_level, wrapper.texture_size, wrapper.cube_mode,
                              _FILTER_MODES[filter_mode], boundary, _plugin._TEX_GRAD_SCRATCH)
        if served is not None:
            return served
    return _Dispatch.apply(_TextureOp, filter_mode, boundary, tex, uv, uv_da, mip_level_bias, wrapper, *levels)


def texture_construct_mip(tex, max_mip_level=None, cube_mode=False):
    """Build the mip stack of a texture once, for reuse through ``texture(..., mip=...)`` while the texture
    stays constant.  ``cube_mode`` must be True for cube maps.  Returns an opaque object."""
    _tensors(tex=tex)
    assert cube_mode is True or cube_mode is False
    return _plugin.texture_construct_mip(tex, _mip_limit(max_mip_level), cube_mode)


# ------------------------------------------------------------------------------------------------
# antialias()

def antialias(color, rast, pos, tri, topology_hash=None, pos_gradient_boost=1.0):
    """Blend silhouette pixels according to the coverage of the edge that crosses them, which is what gives
    vertex positions a gradient through visibility.

    Silhouettes are found through shared vertex INDICES in ``tri``: a vertex used by several triangles must be
    referenced by the same index everywhere, otherwise every edge around it counts as a silhouette.

    Args:
        color: ``[minibatch, height, width, channels]`` image to antialias.
        rast: first output of ``rasterize()``; pos, tri: the tensors that were rasterized.
        topology_hash: result of ``antialias_construct_topology_hash(tri)`` (built internally when omitted).
        pos_gradient_boost: multiplier for the gradient that reaches ``pos``.

    Returns:
        The antialiased image, same shape as ``color``.
    """
    _tensors(color=color, rast=rast, pos=pos, tri=tri)
    if topology_hash is None:
        topology_hash = _plugin.antialias_construct_topology_hash(tri)
    else:
        assert isinstance(topology_hash, _plugin.TopologyHashWrapper)
    host = _plugin.host_layer()                             # (see _rasterize_layer)
    if host is not None and topology_hash.ev_hash is not None:
        served = host.antialias(color, rast, pos, tri, topology_hash.ev_hash, float(pos_gradient_boost))
        if served is not None:
            return served
    return _Dispatch.apply(_AntialiasOp, color, rast, pos, tri, topology_hash, pos_gradient_boost)


def antialias_construct_topology_hash(tri):
    """Build the edge -> opposite-vertex table of a triangle tensor once, for reuse through
    ``antialias(..., topology_hash=...)`` while the topology stays constant.  Returns an opaque object."""
    _tensors(tri=tri)
    return _plugin.antialias_construct_topology_hash(tri)
