"""Host glue: the MI355X stand-in for the reference's pybind11 module ``_nvdiffrast_c``.

Every public function here has the name, argument order, return arity and error
behaviour of the matching ``m.def`` in the reference's
``csrc/torch/torch_bindings.cpp:43-71``; the bodies do what the reference's
``csrc/torch/torch_*.cpp`` glue does (validate tensors, allocate outputs with torch,
launch on torch's current stream) but hand the work to ``libnvdr_hip.so`` through the C
ABI of ``include/nvdr_hip.h``.  The reference's own ``nvdiffrast/torch/ops.py`` runs
unchanged on top of this module.

There is no CPU path: non-GPU tensors are rejected with the reference's messages and a
missing native library raises at first use.
"""
import torch

from .. import _capi

_log_level = 1  # torch_bindings.cpp:50-51 forwards to FLAGS_caffe2_log_level (default 1 per ops.py:34)


def get_log_level():
    return _log_level


def set_log_level(level):
    global _log_level
    _log_level = int(level)


# ----------------------------------------------------------------------------- checks
# Same conditions and wording as NVDR_CHECK_* (csrc/torch/torch_common.inl:20-28).

def _fail(func, msg):
    raise RuntimeError(f"{func}(): {msg}")


def _names(named):
    return ", ".join(n for n, _ in named)


def _check_device(func, **named):
    items = [(n, t) for n, t in named.items()]
    dev = None
    for _, t in items:
        if not t.is_cuda or (dev is not None and t.device != dev):
            _fail(func, f"Inputs {_names(items)} must reside on the same GPU device")
        dev = t.device
    return dev


def _check_cpu(func, **named):
    for n, t in named.items():
        if t.device.type != "cpu":
            _fail(func, f"Inputs {_names(list(named.items()))} must reside on CPU")


def _check_contiguous(func, **named):
    for n, t in named.items():
        if not t.is_contiguous():
            _fail(func, f"Inputs {_names(list(named.items()))} must be contiguous tensors")


def _check_f32(func, **named):
    for n, t in named.items():
        if t.dtype != torch.float32:
            _fail(func, f"Inputs {_names(list(named.items()))} must be float32 tensors")


def _check_i32(func, **named):
    for n, t in named.items():
        if t.dtype != torch.int32:
            _fail(func, f"Inputs {_names(list(named.items()))} must be int32 tensors")


def _require(cond, func, msg):
    if not cond:
        _fail(func, msg)


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def _pad8(x):
    return (x + 7) & ~7


# ----------------------------------------------------------------------------- rasterize

class RasterizeCRStateWrapper:
    """Per-context state (reference: csrc/torch/torch_types.h:15-23, torch_rasterize.cpp:27-38).

    Owns the scratch memory of the rasterizer (triangle records, AABBs, pool counters)
    and, while a DepthPeeler is active, the two depth surfaces.  All of it is torch
    memory (caching allocator), grown on demand and kept for the context's lifetime."""

    def __init__(self, cuda_device_idx):
        self.cuda_device_idx = int(cuda_device_idx)
        self.scratch = None
        self.depth = None        # current depth surface  [N,Hp,Wp] int32 (u32 bits)
        self.peel = None         # previous layer's depth surface

    def get_scratch(self, nbytes, device):
        if self.scratch is None or self.scratch.numel() < nbytes or self.scratch.device != device:
            self.scratch = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        return self.scratch

    def depth_surfaces(self, shape, device, swap):
        """Returns (peel_in or None, depth_out); mirrors swapDepthAndPeel (RasterImpl.cpp:123-130)."""
        if swap:
            self.depth, self.peel = self.peel, self.depth
        if self.depth is None or tuple(self.depth.shape) != tuple(shape) or self.depth.device != device:
            self.depth = torch.empty(shape, dtype=torch.int32, device=device)
        return (self.peel if swap else None), self.depth


def rasterize_fwd_cuda(state, pos, tri, resolution, ranges, peeling_idx):
    """torch_rasterize.cpp:43-166."""
    fn = "rasterize_fwd_cuda"
    dev = _check_device(fn, pos=pos, tri=tri)
    _check_cpu(fn, ranges=ranges)
    _check_contiguous(fn, pos=pos, tri=tri, ranges=ranges)
    _check_f32(fn, pos=pos)
    _check_i32(fn, tri=tri, ranges=ranges)
    _require(pos.get_device() == state.cuda_device_idx, fn,
             "CudaRaster context must must reside on the same device as input tensors")

    instance_mode = pos.dim() > 2
    if instance_mode:
        _require(pos.dim() == 3 and pos.size(0) > 0 and pos.size(1) > 0 and pos.size(2) == 4, fn,
                 "instance mode - pos must have shape [>0, >0, 4]")
    else:
        _require(pos.dim() == 2 and pos.size(0) > 0 and pos.size(1) == 4, fn, "range mode - pos must have shape [>0, 4]")
        _require(ranges.dim() == 2 and ranges.size(0) > 0 and ranges.size(1) == 2, fn,
                 "range mode - ranges must have shape [>0, 2]")
    _require(tri.dim() == 2 and tri.size(0) > 0 and tri.size(1) == 3, fn, "tri must have shape [>0, 3]")

    height, width = int(resolution[0]), int(resolution[1])
    depth = pos.size(0) if instance_mode else ranges.size(0)
    _require(height > 0 and width > 0, fn, "resolution must be [>0, >0]")

    V = pos.size(1) if instance_mode else pos.size(0)
    T = tri.size(0)
    if instance_mode:
        max_tri, ranges_dev = T, None
    else:
        max_tri = max(int(ranges[:, 1].max().item()), 1)
        ranges_dev = ranges.to(dev)

    lib = _capi.load()
    with torch.cuda.device(dev):
        out = torch.empty((depth, height, width, 4), dtype=torch.float32, device=dev)
        out_db = torch.empty((depth, height, width, 4), dtype=torch.float32, device=dev)
        nbytes = lib.nvdr_rasterize_scratch_bytes(depth, max_tri, height, width)
        scratch = state.get_scratch(nbytes, dev)

        # Depth surfaces exist only while peeling (peeling_idx >= 0); layer k > 0 reads layer k-1's.
        peel_in = depth_out = None
        if peeling_idx >= 0:
            peel_in, depth_out = state.depth_surfaces((depth, _pad8(height), _pad8(width)), dev, swap=peeling_idx > 0)

        rc = lib.nvdr_rasterize_fwd(pos.data_ptr(), tri.data_ptr(), _capi.ptr(ranges_dev),
                                    int(instance_mode), depth, V, T, max_tri, height, width,
                                    _capi.ptr(peel_in), _capi.ptr(depth_out),
                                    scratch.data_ptr(), scratch.numel(),
                                    out.data_ptr(), out_db.data_ptr(), _stream(dev))
    _capi.check(rc, fn)
    return out, out_db


def rasterize_grad_db(pos, tri, out, dy, ddb):
    """torch_rasterize.cpp:171-256.  ``ddb`` may be None (== rasterize_grad)."""
    fn = "rasterize_grad_db"
    enable_db = ddb is not None
    if enable_db:
        dev = _check_device(fn, pos=pos, tri=tri, out=out, dy=dy, ddb=ddb)
        _check_contiguous(fn, pos=pos, tri=tri, out=out)
        _check_f32(fn, pos=pos, out=out, dy=dy, ddb=ddb)
    else:
        dev = _check_device(fn, pos=pos, tri=tri, out=out, dy=dy)
        _check_contiguous(fn, pos=pos, tri=tri, out=out)
        _check_f32(fn, pos=pos, out=out, dy=dy)
    _check_i32(fn, tri=tri)

    instance_mode = pos.dim() > 2
    _require(out.dim() == 4, fn, "tensor out must be rank-4")
    depth, height, width = out.size(0), out.size(1), out.size(2)
    _require(depth > 0 and height > 0 and width > 0, fn, "resolution must be [>0, >0, >0]")
    if instance_mode:
        _require(pos.dim() == 3 and pos.size(0) == depth and pos.size(1) > 0 and pos.size(2) == 4, fn,
                 "pos must have shape [depth, >0, 4]")
    else:
        _require(pos.dim() == 2 and pos.size(0) > 0 and pos.size(1) == 4, fn, "pos must have shape [>0, 4]")
    _require(tri.dim() == 2 and tri.size(0) > 0 and tri.size(1) == 3, fn, "tri must have shape [>0, 3]")
    _require(tuple(out.shape) == (depth, height, width, 4), fn, "out must have shape [depth, height, width, 4]")
    _require(tuple(dy.shape) == (depth, height, width, 4), fn, "dy must have shape [depth, height, width, 4]")
    if enable_db:
        _require(tuple(ddb.shape) == (depth, height, width, 4), fn, "ddb must have shape [depth, height, width, 4]")

    dy_ = dy.contiguous()
    ddb_ = ddb.contiguous() if enable_db else None
    V = pos.size(1) if instance_mode else pos.size(0)
    with torch.cuda.device(dev):
        grad = torch.zeros_like(pos)
        rc = _capi.load().nvdr_rasterize_grad(pos.data_ptr(), tri.data_ptr(), out.data_ptr(), dy_.data_ptr(),
                                              _capi.ptr(ddb_), int(instance_mode), depth, V, tri.size(0),
                                              height, width, grad.data_ptr(), _stream(dev))
    _capi.check(rc, fn)
    return grad


def rasterize_grad(pos, tri, out, dy):
    """torch_rasterize.cpp:259-263."""
    return rasterize_grad_db(pos, tri, out, dy, None)


# ----------------------------------------------------------------------------- interpolate

_IP_MAX_DIFF_ATTRS = 32   # csrc/common/interpolate.h:18


def _diff_list(diff_attrs_vec):
    import ctypes
    n = len(diff_attrs_vec)
    arr = (ctypes.c_int32 * max(n, 1))(*[int(x) for x in diff_attrs_vec])
    return arr, n


def interpolate_fwd_da(attr, rast, tri, rast_db, diff_attrs_all, diff_attrs_vec):
    """torch_interpolate.cpp:42-124."""
    fn = "interpolate_fwd_da"
    enable_da = (rast_db is not None) and (bool(diff_attrs_all) or len(diff_attrs_vec) > 0)
    instance_mode = attr.dim() > 2
    if enable_da:
        dev = _check_device(fn, attr=attr, rast=rast, tri=tri, rast_db=rast_db)
        _check_contiguous(fn, attr=attr, rast=rast, tri=tri, rast_db=rast_db)
        _check_f32(fn, attr=attr, rast=rast, rast_db=rast_db)
    else:
        dev = _check_device(fn, attr=attr, rast=rast, tri=tri)
        _check_contiguous(fn, attr=attr, rast=rast, tri=tri)
        _check_f32(fn, attr=attr, rast=rast)
    _check_i32(fn, tri=tri)

    _require(rast.dim() == 4 and rast.size(0) > 0 and rast.size(1) > 0 and rast.size(2) > 0 and rast.size(3) == 4, fn,
             "rast must have shape[>0, >0, >0, 4]")
    _require(tri.dim() == 2 and tri.size(0) > 0 and tri.size(1) == 3, fn, "tri must have shape [>0, 3]")
    _require(attr.dim() in (2, 3) and attr.size(0) > 0 and attr.size(1) > 0 and (attr.dim() == 2 or attr.size(2) > 0), fn,
             "attr must have shape [>0, >0, >0] or [>0, >0]")
    if instance_mode:
        _require(attr.size(0) == rast.size(0) or attr.size(0) == 1, fn, "minibatch size mismatch between inputs rast, attr")
    if enable_da:
        _require(rast_db.dim() == 4 and rast_db.size(0) > 0 and rast_db.size(1) > 0 and rast_db.size(2) > 0 and rast_db.size(3) == 4,
                 fn, "rast_db must have shape[>0, >0, >0, 4]")
        _require(rast_db.size(1) == rast.size(1) and rast_db.size(2) == rast.size(2), fn,
                 "spatial size mismatch between inputs rast and rast_db")
        _require(rast_db.size(0) == rast.size(0), fn, "minibatch size mismatch between inputs rast, rast_db")
        if not diff_attrs_all:
            _require(len(diff_attrs_vec) <= _IP_MAX_DIFF_ATTRS, fn,
                     "too many entries in diff_attrs list (increase IP_MAX_DIFF_ATTRS)")

    V = attr.size(1 if instance_mode else 0)
    A = attr.size(2 if instance_mode else 1)
    N, H, W = rast.size(0), rast.size(1), rast.size(2)
    D = (A if diff_attrs_all else len(diff_attrs_vec)) if enable_da else 0
    lst, nlst = _diff_list([] if diff_attrs_all else diff_attrs_vec)
    with torch.cuda.device(dev):
        out = torch.empty((N, H, W, A), dtype=torch.float32, device=dev)
        out_da = torch.empty((N, H, W, 2 * D), dtype=torch.float32, device=dev)
        rc = _capi.load().nvdr_interpolate_fwd(attr.data_ptr(), rast.data_ptr(), tri.data_ptr(),
                                               rast_db.data_ptr() if enable_da else None,
                                               int(instance_mode), attr.size(0) if instance_mode else 1,
                                               N, V, A, tri.size(0), H, W,
                                               int(bool(diff_attrs_all)), lst, nlst,
                                               out.data_ptr(), out_da.data_ptr() if enable_da else None, _stream(dev))
    _capi.check(rc, fn)
    return out, out_da


def interpolate_fwd(attr, rast, tri):
    """torch_interpolate.cpp:127-132."""
    return interpolate_fwd_da(attr, rast, tri, None, False, [])


def interpolate_grad_da(attr, rast, tri, dy, rast_db, dda, diff_attrs_all, diff_attrs_vec):
    """torch_interpolate.cpp:137-239."""
    fn = "interpolate_grad_da"
    enable_da = (rast_db is not None) and (bool(diff_attrs_all) or len(diff_attrs_vec) > 0)
    instance_mode = attr.dim() > 2
    if enable_da:
        dev = _check_device(fn, attr=attr, rast=rast, tri=tri, dy=dy, rast_db=rast_db, dda=dda)
        _check_contiguous(fn, attr=attr, rast=rast, tri=tri, rast_db=rast_db)
        _check_f32(fn, attr=attr, rast=rast, dy=dy, rast_db=rast_db, dda=dda)
    else:
        dev = _check_device(fn, attr=attr, rast=rast, tri=tri, dy=dy)
        _check_contiguous(fn, attr=attr, rast=rast, tri=tri)
        _check_f32(fn, attr=attr, rast=rast, dy=dy)
    _check_i32(fn, tri=tri)

    attr_depth = attr.size(0) if instance_mode else 1
    _require(rast.dim() == 4 and rast.size(0) > 0 and rast.size(1) > 0 and rast.size(2) > 0 and rast.size(3) == 4, fn,
             "rast must have shape[>0, >0, >0, 4]")
    _require(tri.dim() == 2 and tri.size(0) > 0 and tri.size(1) == 3, fn, "tri must have shape [>0, 3]")
    _require(attr.dim() in (2, 3) and attr.size(0) > 0 and attr.size(1) > 0 and (attr.dim() == 2 or attr.size(2) > 0), fn,
             "attr must have shape [>0, >0, >0] or [>0, >0]")
    _require(dy.dim() == 4 and dy.size(0) > 0 and dy.size(1) == rast.size(1) and dy.size(2) == rast.size(2) and dy.size(3) > 0,
             fn, "dy must have shape [>0, height, width, >0]")
    _require(dy.size(3) == attr.size(attr.dim() - 1), fn, "argument count mismatch between inputs dy, attr")
    _require((attr_depth == rast.size(0) or attr_depth == 1) and dy.size(0) == rast.size(0), fn,
             "minibatch size mismatch between inputs rast, dy, attr")
    if enable_da:
        _require(dda.dim() == 4 and dda.size(0) > 0 and dda.size(1) == rast.size(1) and dda.size(2) == rast.size(2), fn,
                 "dda must have shape [>0, height, width, ?]")
        _require(dda.size(0) == rast.size(0), fn, "minibatch size mismatch between rast, dda")
        _require(rast_db.dim() == 4 and rast_db.size(0) > 0 and rast_db.size(1) > 0 and rast_db.size(2) > 0 and rast_db.size(3) == 4,
                 fn, "rast_db must have shape[>0, >0, >0, 4]")
        _require(rast_db.size(1) == rast.size(1) and rast_db.size(2) == rast.size(2), fn,
                 "spatial size mismatch between inputs rast and rast_db")
        _require(rast_db.size(0) == rast.size(0), fn, "minibatch size mismatch between inputs rast, rast_db")
        if not diff_attrs_all:
            _require(len(diff_attrs_vec) <= _IP_MAX_DIFF_ATTRS, fn,
                     "too many entries in diff_attrs list (increase IP_MAX_DIFF_ATTRS)")

    V = attr.size(1 if instance_mode else 0)
    A = attr.size(2 if instance_mode else 1)
    N, H, W = rast.size(0), rast.size(1), rast.size(2)
    dy_ = dy.contiguous()
    dda_ = dda.contiguous() if enable_da else None
    lst, nlst = _diff_list([] if diff_attrs_all else diff_attrs_vec)
    with torch.cuda.device(dev):
        g_attr = torch.zeros_like(attr)
        g_rast = torch.empty_like(rast)
        g_rast_db = torch.empty_like(rast_db) if enable_da else None
        rc = _capi.load().nvdr_interpolate_grad(attr.data_ptr(), rast.data_ptr(), tri.data_ptr(), dy_.data_ptr(),
                                                rast_db.data_ptr() if enable_da else None, _capi.ptr(dda_),
                                                int(instance_mode), attr_depth, N, V, A, tri.size(0), H, W,
                                                int(bool(diff_attrs_all)), lst, nlst,
                                                g_attr.data_ptr(), g_rast.data_ptr(), _capi.ptr(g_rast_db), _stream(dev))
    _capi.check(rc, fn)
    return g_attr, g_rast, g_rast_db


def interpolate_grad(attr, rast, tri, dy):
    """torch_interpolate.cpp:242-248."""
    g_attr, g_rast, _ = interpolate_grad_da(attr, rast, tri, dy, None, None, False, [])
    return g_attr, g_rast
