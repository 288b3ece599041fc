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
import os
import weakref

import torch
from torch.utils._pytree import tree_map as _tree_map

from .. import _capi

# torch_bindings.cpp:50-51 forward to c10's FLAGS_caffe2_log_level (0 INFO, 1 WARNING = default, 2 ERROR, 3 FATAL);
# here the level lives in the native library (NVDR_OPT_LOG_LEVEL) and gates nvdr_log().

def get_log_level():
    return int(_capi.load().nvdr_get_option(_capi.OPT_LOG_LEVEL))


def set_log_level(level):
    _capi.check(_capi.load().nvdr_set_option(_capi.OPT_LOG_LEVEL, int(level)), "set_log_level")


def set_cube_corner_fix(enable):
    """Not in the reference.  False (default): cube-map corner texels are sampled exactly as the reference does,
    including its loss of the corner flag for texture slices >= 1 (texture_kernel.cu:85-88,431-432).  True: the
    texel missing at a cube corner is the average of the other three for every slice."""
    _capi.check(_capi.load().nvdr_set_option(_capi.OPT_CUBE_CORNER_FIX, int(bool(enable))), "set_cube_corner_fix")


# Fused backward of rasterize -> interpolate (ops.py `_RasterOrigin`): "auto" = prepare the position gradient inside
# interpolate's backward kernel and use it when autograd shows that nothing else contributed to rast's gradient;
# "off" = always the two separate kernels of the reference's structure.
_fused = {"mode": "auto", "used": 0, "discarded": 0, "materialized": 0}


def set_fused_backward(mode):
    """Not in the reference.  "auto" (default) or "off"."""
    assert mode in ("auto", "off")
    _fused["mode"] = mode
    if mode == "auto":
        _fused_epoch[0] += 1
    if _host_state["mod"]:
        _host_state["mod"].set_fused(mode == "auto")


def fused_backward_mode():
    return _fused["mode"]


_fused_epoch = [0]


def fused_backward_epoch():
    """Contexts remember WHEN a prepared gradient was discarded on them; set_fused_backward("auto") starts a new epoch, which
    re-arms them all (one debugging step with a mask built from rast need not cost the fused kernel for the rest of the run)."""
    return _fused_epoch[0]


def fused_backward_count(what=None):
    """Counts how often a prepared position gradient was used / discarded, and how often the gradient of rast that the fused
    kernel did not write had to be computed after all (tests); without argument returns all three."""
    if what is None:
        r = {"used": _fused["used"], "discarded": _fused["discarded"], "materialized": _fused["materialized"]}
        if _host_state["mod"]:
            # the compiled layer never discards: what other consumers of rast contribute is ADDED to the prepared share
            # ("fused_plus"; rasterize_grad is linear in dy) -- both of its outcomes count as used
            c = _host_state["mod"].counters()
            r["used"] += c["fused_alone"] + c["fused_plus"]
            r.update(c)
        return r
    _fused[what] += 1


# ---- the compiled host layer ------------------------------------------------------------------------------------------------
# csrc_host/nvdr_torch_host.cpp does for rasterize() and interpolate() what this module does -- validation, allocation, launch --
# and carries their autograd nodes, in C++ (the reference's glue is C++ too: csrc/torch/torch_rasterize.cpp, torch_interpolate.cpp).
# ops.py asks it first; it declines (None) anything but the ordinary case, and the call then comes here: this module stays the
# path of the rare modes and of every error message.  NVDR_HOST=0, set_host_layer("python") or an unbuilt module: all calls come here.
_host_state = {"mod": False, "enabled": True}


def host_layer():
    """The compiled host module, or None."""
    m = _host_state["mod"]
    if m is False:
        m = _host_state["mod"] = _capi.host()
        if m is not None:
            m.set_fused(_fused["mode"] == "auto")
            m.set_skip(_tiles["skip"])
            m.set_verify(_tiles["verify"])
    return m if _host_state["enabled"] else None


def set_host_layer(mode):
    """Not in the reference.  "compiled" (default where the module is built) or "python"."""
    assert mode in ("compiled", "python")
    _host_state["enabled"] = mode == "compiled"


def host_layer_name():
    return "compiled" if host_layer() is not None else "python"


def _host_raw(fn, *args):
    """A function of the compiled layer as a PLUGIN-level entry point: plain tensors out, no autograd node on them (the caller --
    the reference's ops.py, a test -- brings its own autograd function; inside its forward() grad mode is off already)."""
    if torch.is_grad_enabled():
        with torch.no_grad():
            return fn(*args)
    return fn(*args)


def _is_capturing(device):
    return torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing()


def _log_info(msg):
    _capi.load().nvdr_log(0, msg.encode())


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


# Host time matters: with small batches the step time IS the host's (BASELINE config 2 at batch 16 is host-bound on an
# MI355X).  The public torch.cuda helpers (current_stream(), the `device` context manager) cost ~6 us per use in index
# normalisation and object construction; these go to the same C entry points directly.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream(device):
    """The current hipStream_t of `device` as an integer."""
    if _raw_stream is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


class _on_device:
    """`with torch.cuda.device(dev)` that does nothing in the usual case of `dev` being the current device already."""
    __slots__ = ("idx", "prev")

    def __init__(self, device):
        self.idx = device.index
        self.prev = -1

    def __enter__(self):
        cur = _raw_device() if _raw_device is not None else torch.cuda.current_device()
        if cur != self.idx:
            self.prev = cur
            torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch.cuda.set_device(self.prev)
        return False


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
        self.clean_layout = None
        self.fused_disabled = -1      # ops.py: the fused-backward epoch in which a prepared gradient was discarded on this context (-> stop preparing)
        self.captured = False    # some call of this context was recorded into a hipGraph
        self.retired = []        # scratch buffers that recorded graphs still point to
        self.reported_bytes = 0
        self.pools = {}          # (N, max_tri) -> clip-pool slots per image that the last call in growing mode needed
        self.sizes = {}          # (N, max_tri, H, W, pool) -> scratch bytes
        self.last_flags = None   # tile occupancy of the most recent rasterize_fwd_cuda output (ops.py attaches it to that rast)
        self.depth = None        # current depth surface  [N,Hp,Wp] int32 (u32 bits)
        self.peel = None         # previous layer's depth surface
        self._host = None        # the compiled host layer's state of this context (its own scratch and depth surfaces)

    def host_state(self, host):
        if self._host is None:
            self._host = host.RasterState(self.cuda_device_idx)
        return self._host

    # -- what tests and tools look at, whichever host layer serves the context
    def poison_scratch(self, value=0):
        """Overwrite the rasterizer's scratch (both layers' buffers) and forget that it was left clean."""
        if self.scratch is not None:
            self.scratch.fill_(value)
        self.clean_layout = None
        if self._host is not None:
            self._host.poison_scratch(value)

    def set_pool_hint(self, n, max_tri, slots):
        self.pools[(n, max_tri)] = slots
        h = host_layer()
        if h is not None:
            self.host_state(h).set_pool(n, max_tri, slots)

    def pool_slots(self, n, max_tri):
        """Clip-pool slots per image remembered for this shape by the layer that serves the context (-1: none yet)."""
        h = host_layer()
        if h is not None:
            return self.host_state(h).get_pool(n, max_tri)
        return self.pools.get((n, max_tri), -1)

    def get_scratch(self, nbytes, device, layout):
        """Returns (buffer, clean): `clean` tells the library that the buffer's control block is as this
        context's previous successful call with the same layout left it (include/nvdr_hip.h).

        hipGraph capture freezes both the buffer address and the flag into the recorded launches, while the
        graph may be replayed after calls with other layouts have used the buffer.  So a call recorded during
        capture never claims a clean buffer (its memset becomes part of the graph and every replay is
        self-contained), a context that has been captured never claims one again in eager mode either
        (a replay may have run in between), and a buffer that a graph may point to is kept alive when the
        context outgrows it."""
        capturing = _is_capturing(device)
        if self.scratch is None or self.scratch.numel() < nbytes or self.scratch.device != device:
            if self.captured and self.scratch is not None:
                self.retired.append(self.scratch)
            self.scratch = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self.clean_layout = None
            if nbytes > self.reported_bytes:
                # RasterImpl.cpp:189-197: report growth at 10 MB granularity, INFO severity
                mb = ((((int(nbytes) - 1) >> 20) + 1 + 9) // 10) * 10
                _log_info("Internal buffers grown to %d MB" % mb)
                self.reported_bytes = mb << 20
        if capturing:
            self.captured = True
        clean = (self.clean_layout == layout) and not self.captured
        self.clean_layout = None                  # re-armed by mark_clean() once the call has succeeded
        return self.scratch, clean

    def mark_clean(self, layout):
        self.clean_layout = layout

    def scratch_bytes(self, lib, n, max_tri, h, w, pool):
        """nvdr_rasterize_scratch_bytes[_pool], remembered per shape (a pure function of its arguments)."""
        key = (n, max_tri, h, w, pool)
        v = self.sizes.get(key)
        if v is None:
            if len(self.sizes) > 256:
                self.sizes.clear()
            v = self.sizes[key] = int(lib.nvdr_rasterize_scratch_bytes_pool(n, max_tri, h, w, pool))
        return v

    def pool_hint(self, n, max_tri):
        """Clip-pool slots per image to start with in growing mode: what this shape needed last time, else the worst
        case of 4096 triangles (small meshes are covered completely) or a quarter slot per triangle."""
        return self.pools.get((n, max_tri), min(6 * max_tri, max(6 * 4096, max_tri // 4)))

    def grow_pool(self, n, max_tri, need):
        self.pools[(n, max_tri)] = min(6 * max_tri, need + need // 4 + 1024)
        return self.pools[(n, max_tri)]

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
    h = host_layer()
    if h is not None and len(resolution) == 2:
        # the ordinary case, served by the compiled layer (csrc_host/nvdr_torch_host.cpp); None: not ordinary -> everything below
        hs = state.host_state(h)
        served = _host_raw(h.rasterize, hs, pos, tri, int(resolution[0]), int(resolution[1]), ranges, True, int(peeling_idx))
        if served is not None:
            out, out_db = served
            if peeling_idx >= 0:
                state.depth, state.peel = hs.depth, hs.peel
            state.last_flags = flags = hs.last_flags
            _attach_tiles(out, flags, "rast")
            out._nvdr_tiles.origin = _FwdOrigin(pos, tri, state, out, out_db)
            return out, out_db
    ps, ts = pos.shape, tri.shape
    dev = pos.device
    instance_mode = len(ps) > 2
    # The common case -- everything in order -- is decided by ONE expression (host time is the step time of small batches);
    # anything else goes through the reference's checks one by one, for the reference's message (torch_rasterize.cpp:47-77).
    if not (pos.is_cuda and tri.device == dev and dev.index == state.cuda_device_idx and ranges.device.type == "cpu"
            and pos.dtype is torch.float32 and tri.dtype is torch.int32 and ranges.dtype is torch.int32
            and pos.is_contiguous() and tri.is_contiguous() and ranges.is_contiguous()
            and len(ps) == 3 and ps[0] > 0 and ps[1] > 0 and ps[2] == 4 and len(ts) == 2 and ts[0] > 0 and ts[1] == 3):
        dev = _check_device(fn, pos=pos, tri=tri)
        _check_cpu(fn, ranges=ranges)
        _check_contiguous(fn, pos=pos, tri=tri, ranges=ranges)
        _check_f32(fn, pos=pos)
        _check_i32(fn, tri=tri, ranges=ranges)
        _require(pos.get_device() == state.cuda_device_idx, fn,
                 "CudaRaster context must must reside on the same device as input tensors")
        if instance_mode:
            _require(pos.dim() == 3 and pos.size(0) > 0 and pos.size(1) > 0 and pos.size(2) == 4, fn,
                     "instance mode - pos must have shape [>0, >0, 4]")
        else:
            _require(pos.dim() == 2 and pos.size(0) > 0 and pos.size(1) == 4, fn, "range mode - pos must have shape [>0, 4]")
            _require(ranges.dim() == 2 and ranges.size(0) > 0 and ranges.size(1) == 2, fn,
                     "range mode - ranges must have shape [>0, 2]")
        _require(tri.dim() == 2 and tri.size(0) > 0 and tri.size(1) == 3, fn, "tri must have shape [>0, 3]")

    height, width = int(resolution[0]), int(resolution[1])
    depth = ps[0] if instance_mode else ranges.size(0)
    _require(height > 0 and width > 0, fn, "resolution must be [>0, >0]")

    V = ps[1] if instance_mode else ps[0]
    T = ts[0]
    if instance_mode:
        max_tri, ranges_dev = T, None
    else:
        max_tri = max(int(ranges[:, 1].max().item()), 1)
        ranges_dev = ranges.to(dev)

    lib = _capi.load()
    with _on_device(dev):
        out = torch.empty((depth, height, width, 4), dtype=torch.float32, device=dev)
        out_db = torch.empty((depth, height, width, 4), dtype=torch.float32, device=dev)
        # one byte per 8x8 tile: does any pixel show a triangle?  Consumers of `out` skip the empty tiles (include/nvdr_hip.h)
        # ... and, behind those, the order in which the consumers' launches walk the image (bins with triangles first)
        flags = torch.empty((tile_flags_bytes(depth, height, width),), dtype=torch.uint8, device=dev)
        # Depth surfaces exist only while peeling (peeling_idx >= 0); layer k > 0 reads layer k-1's.
        peel_in = depth_out = None
        if peeling_idx >= 0:
            peel_in, depth_out = state.depth_surfaces((depth, _pad8(height), _pad8(width)), dev, swap=peeling_idx > 0)

        # Scratch policy (include/nvdr_hip.h, NVDR_OPT_SCRATCH_LIMIT_MB): the worst case while it is affordable -- no
        # overflow possible, no host synchronisation, hipGraph-capturable -- otherwise a clip pool that grows on demand.
        worst = state.scratch_bytes(lib, depth, max_tri, height, width, -1)
        adaptive = worst > (int(lib.nvdr_get_option(_capi.OPT_SCRATCH_LIMIT_MB)) << 20)
        pool = state.pool_hint(depth, max_tri) if adaptive else -1
        worst_pool = 6 * max_tri                 # the clipper's worst case: a pool this large cannot overflow and the library
        #                                          neither clears nor writes the demand counter for it (include/nvdr_hip.h)
        if adaptive and _is_capturing(dev):
            _fail(fn, "this call needs %d MB of worst-case rasterizer scratch, above the NVDR_OPT_SCRATCH_LIMIT_MB limit; the "
                      "growing clip pool used instead reads a counter back after each call and cannot be captured into a "
                      "graph (raise the limit to capture)" % (worst >> 20))
        while True:
            nbytes = state.scratch_bytes(lib, depth, max_tri, height, width, pool)
            layout = (depth, max_tri, height, width, pool)
            scratch, clean = state.get_scratch(nbytes, dev, layout)
            rc = lib.nvdr_rasterize_fwd(pos.data_ptr(), tri.data_ptr(), _capi.ptr(ranges_dev),
                                        int(instance_mode), depth, V, T, max_tri, height, width,
                                        _capi.ptr(peel_in), _capi.ptr(depth_out),
                                        scratch.data_ptr(), scratch.numel(), int(clean), pool,
                                        out.data_ptr(), out_db.data_ptr(), flags.data_ptr(), _stream(dev))
            if rc != 0 or not adaptive or pool < 0 or pool >= worst_pool:
                break                                                        # (worst-case pool: nothing to read back)
            off = lib.nvdr_rasterize_pool_peak_offset(depth, max_tri, height, width, pool)
            need = int(scratch[off:off + 4].view(torch.int32).item())        # the one host synchronisation of this mode
            if need <= pool:
                break
            grown = state.grow_pool(depth, max_tri, need)
            if grown <= pool:                                                # cannot happen (need <= 6 * max_tri): never spin
                _fail(fn, "clip pool demand %d exceeds the worst case of %d slots per image" % (need, worst_pool))
            pool = grown
            _log_info("Clip pool grown to %d sub-triangle slots per image" % pool)
    _capi.check(rc, fn)
    state.mark_clean(layout)
    state.last_flags = flags
    _attach_tiles(out, flags, "rast")
    out._nvdr_tiles.origin = _FwdOrigin(pos, tri, state, out, out_db)
    return out, out_db


# ---- tile occupancy records -----------------------------------------------------------------------------------------------
# rasterize_fwd_cuda leaves with its `rast` a record of the tile flags the rasterizer wrote for it; interpolate_fwd* leave with
# their outputs a record of the same flags as "tiles of zeros".  Every entry point that reads such a tensor looks the record up
# ITSELF when the caller passes no `tile_flags` -- so a caller that binds this module exactly like the reference's ops.py binds
# _nvdiffrast_c, with no extra arguments, gets the empty-tile skipping too.  A record is honoured only
# while the tensor is what it was: same storage, same version counter, same shape.  Deviation from the reference that this
# cannot see: writes that do not bump the version counter (`rast.data[...] = x`, an external kernel writing through data_ptr).
# `set_tile_skipping(False)` switches the whole mechanism off; `set_tile_flag_verification(True)` (or NVDR_VERIFY_TILE_FLAGS=1)
# re-derives the flags from the tensor actually passed on every use and raises when they disagree.
_tiles = {"skip": True, "verify": os.environ.get("NVDR_VERIFY_TILE_FLAGS", "0") not in ("", "0"), "verified": 0}
_tile_registry = {}          # data_ptr -> weakref of the tensor that carries the record (for tensors autograd hands back as copies)


def set_tile_skipping(enable):
    """Not in the reference.  True (default): kernels that read a rast / rast_db / uv / uv_da tensor skip the 8x8 tiles the
    rasterizer found empty while the tensor is untouched (version counter).  False: every pixel is read, as in the reference."""
    _tiles["skip"] = bool(enable)
    if _host_state["mod"]:
        _host_state["mod"].set_skip(bool(enable))


def set_tile_flag_verification(enable):
    """Not in the reference.  Debug mode: before every use of tile flags, recompute them from the tensor that was passed (one
    host synchronisation per use) and raise RuntimeError if a tile flagged empty is not.  Also: NVDR_VERIFY_TILE_FLAGS=1."""
    _tiles["verify"] = bool(enable)
    if _host_state["mod"]:
        _host_state["mod"].set_verify(bool(enable))        # (the checking mode is served by this module)


def tile_flag_verifications():
    return _tiles["verified"]


class _TileRecord:
    """kind "rast": flag 0 = no pixel of the tile shows a triangle; kind "zero": flag 0 = every element of the tile is zero."""
    __slots__ = ("flags", "ptr", "version", "shape", "kind", "origin", "__weakref__")

    def __init__(self, flags, t, kind):
        # (full shape AND strides: an alias of the same storage with another last dimension or layout does not inherit the record)
        self.flags, self.ptr, self.version, self.shape, self.kind = flags, t.data_ptr(), t._version, (tuple(t.shape), t.stride()), kind
        self.origin = None                 # kind "rast": _FwdOrigin, set by rasterize_fwd_cuda

    def still(self, t):
        return t.data_ptr() == self.ptr and t._version == self.version and (tuple(t.shape), t.stride()) == self.shape


_KIND = {"rast": 0, "zero": 1}


def _attach_tiles(t, flags, kind):
    t._nvdr_tiles = _TileRecord(flags, t, kind)
    ptr = t.data_ptr()

    def _forget(_ref, ptr=ptr):
        if _tile_registry.get(ptr) is _ref:
            del _tile_registry[ptr]
    _tile_registry[ptr] = weakref.ref(t, _forget)
    h = host_layer()
    if h is not None and flags is not None:
        h.attach(t, flags, _KIND[kind])          # the compiled layer's registry (by storage): its interpolate() finds the flags too


def _record_of(t, kind):
    """The valid record of `t`, found on the tensor itself or -- for the detached copies autograd returns for a Function's saved
    OUTPUTS -- on the tensor that carries it, if that one is still alive (its storage then cannot have been recycled)."""
    rec = getattr(t, "_nvdr_tiles", None)
    if rec is None:
        ref = _tile_registry.get(t.data_ptr())
        owner = ref() if ref is not None else None
        if owner is not None and owner.data_ptr() == t.data_ptr():
            rec = getattr(owner, "_nvdr_tiles", None)
    if rec is None or rec.kind != kind or not rec.still(t):
        return None
    return rec


def flags_of(t, kind="rast"):
    """The tile flags that are known to describe `t` as it is now -- from the record on the tensor (this module) or from the
    compiled layer's registry -- or None."""
    rec = _record_of(t, kind)
    if rec is not None:
        return rec.flags
    h = host_layer()
    return None if h is None else h.flags_of(t, _KIND[kind])


def _verify_tiles(fn, t, flags, kind):
    if torch.cuda.is_current_stream_capturing():
        return
    n, h, w = (int(x) for x in t.shape[:3])
    grid = tile_flags_grid(flags, n, h, w)
    live = (t[..., 3] > 0) if kind == "rast" else (t != 0).any(-1)
    occ = torch.nn.functional.max_pool2d(live.float()[:, None], 8, ceil_mode=True)[:, 0] > 0
    bad = int(((grid == 0) & occ).sum().item())
    _tiles["verified"] += 1
    if bad:
        _fail(fn, "tile flags disagree with the tensor they are used with: %d tile(s) flagged empty are not (was the tensor "
                  "written to without bumping its version counter?)" % bad)


def _auto_flags(fn, tile_flags, kind, *tensors):
    """What an entry point uses as tile flags: the caller's (a tensor), none (False, or skipping switched off), or -- None -- the
    flags of the records of ALL `tensors` (they must agree)."""
    if tile_flags is False or not _tiles["skip"]:
        return None
    if tile_flags is None:
        for t in tensors:
            f = flags_of(t, kind)
            if f is None or (tile_flags is not None and f.data_ptr() != tile_flags.data_ptr()):
                return None
            tile_flags = f
    if tile_flags is not None and _tiles["verify"]:
        for t in tensors:
            _verify_tiles(fn, t, tile_flags, kind)
    return tile_flags


def tile_flags_bytes(n, h, w):
    """nvdr_tile_flags_bytes(N, H, W) without the call (it is on every consumer's path; tests/test_capi_exports.py keeps the
    two in step): occupancy bytes, padding to 16, and for 2048 .. 65536 bins of 64x64 pixels of images up to 2048 pixels a side the
    work order (nBins + 1 ints) and what k_flag_order builds it from."""
    flags = n * ((h + 7) >> 3) * ((w + 7) >> 3)
    bins = n * ((h + 63) >> 6) * ((w + 63) >> 6)
    off = (flags + 15) // 16 * 16
    if not (2048 <= bins <= 65536 and h <= 2048 and w <= 2048):
        return off
    return (off + 4 * (bins + 1) + 7) // 8 * 8 + 8 * bins          # order + count, then the rasterizer's byte per bin and tile row


def tile_flags_grid(tile_flags, n, h, w):
    """The [N, ceil(H/8), ceil(W/8)] occupancy bytes at the front of a tile_flags buffer (tests, tools)."""
    th, tw = (h + 7) >> 3, (w + 7) >> 3
    return tile_flags[:n * th * tw].view(n, th, tw)


def _flags_ok(fn, tile_flags, n, h, w, dev):
    """tile_flags (not a reference argument): None, or the uint8 buffer rasterize_fwd_cuda made for exactly the rast tensor
    being passed (include/nvdr_hip.h: occupancy bytes of its 8x8 tiles, then the consumers' work order)."""
    if tile_flags is None:
        return None
    _require(tile_flags.dtype == torch.uint8 and tile_flags.dim() == 1 and tile_flags.is_contiguous() and tile_flags.device == dev
             and tile_flags.numel() == tile_flags_bytes(int(n), int(h), int(w)), fn,
             "tile_flags do not belong to this rast tensor")
    return tile_flags.data_ptr()


# ---- fused backward of rasterize -> interpolate ------------------------------------------------------------------------------
# The library has one kernel for interpolate_grad[_da] + rasterize_grad[_db] (interpolate_rasterize_grad below).  The operator
# layer of this package drives it with its own bookkeeping (ops.py _RasterOrigin); a caller that binds this module exactly like
# the reference's ops.py binds _nvdiffrast_c calls interpolate_grad[_da] and, later in the same
# backward pass, rasterize_grad[_db] -- so the same exchange is done HERE between those two entry points:
#   interpolate_grad[_da](attr, rast, ...)   rast is what rasterize_fwd_cuda returned (its record says so: _FwdOrigin), untouched,
#       interpolated once, and pos requires a gradient: the fused kernel computes g_attr AND the position gradient; g_rast
#       (and g_rast_db) are not written -- the caller gets stand-ins that compute them if anybody looks (_LazyGrad);
#   rasterize_grad[_db](pos, tri, out, dy, ddb)   dy is that very stand-in, unedited (autograd delivers the object itself when
#       nothing else contributed to rast's gradient): the prepared gradient is returned.  A `ddb` that is an ordinary tensor --
#       the zeros the reference's ops.py lets autograd materialise for an unused rast_db (its backward, ops.py:84-90), or a real
#       gradient from another consumer -- adds its share in a pass that reads ddb first and rast only where ddb is non-zero
#       (nvdr_rasterize_grad with dy == NULL).  Anything else arriving: the stand-in is materialised and the two-kernel path
#       runs as if nothing had been prepared (and the context stops preparing, as in ops.py).
# Calls that pass `fuse=False` (this package's ops.py: it does the same one level up) never take part.

class _LazyGrad(torch.Tensor):
    """The gradient of `rast` (or `rast_db`) that interpolate's backward returns when the fused kernel has already turned it into
    the position gradient: a tensor whose VALUES are computed only if somebody looks at them.

    In the graph rasterize -> interpolate the only reader of rast's gradient is rasterize's backward, and that one does not need
    it any more (ops.py _RasterOrigin, _FwdOrigin below) -- yet it is 16 B/pixel of stores (268 MB of the 297 MB the fused kernel wrote at the
    headline batch), zeros for three quarters of them.  So the kernel does not write it, and autograd gets this stand-in: shape,
    dtype and device of the real thing (a stride-0 view of one zero, so it has storage for the engine's stream bookkeeping), and
    a __torch_dispatch__ that replaces it by the real gradient -- computed then by the reference's own two-kernel formulation,
    interpolate_grad[_da] -- in front of ANY operation that touches it: autograd's summation when rast's gradient has other
    contributors, the clone of retain_grad(), a hook, arithmetic on what autograd.grad(..., inputs=[rast]) hands out.  Whoever
    looks sees the reference's values; the price is paid by those who look.  (What no dispatch can see: raw `.data_ptr()` access
    to this object from user code -- it points at a single zero.)"""

    @staticmethod
    def __new__(cls, like, source, index):
        key = (like.device, like.dtype)
        zero = _LazyGrad._zeros.get(key)
        if zero is None:
            zero = torch.zeros((), dtype=like.dtype, device=like.device)
            if not _is_capturing(like.device):           # (memory allocated while a hipGraph is recorded belongs to the graph's pool)
                _LazyGrad._zeros[key] = zero
        r = torch.Tensor._make_subclass(cls, zero.expand(like.shape), False)
        r._source, r._index = source, index
        # An in-place operation on the stand-in (a hook doing g.mul_(2)) is carried out on the materialised values by
        # __torch_dispatch__, below the level that counts versions -- the counter that moves is the STAND-IN's (shared by all
        # stand-ins of this device and dtype: they are views of one zero).  unedited() compares it with what it was here.
        r._v0 = r._version
        return r

    def unedited(self):
        return self._version == self._v0

    def materialize(self):
        return self._source.get()[self._index]

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        real = lambda t: t.materialize() if isinstance(t, _LazyGrad) else t          # noqa: E731
        return func(*_tree_map(real, args), **_tree_map(real, kwargs or {}))


_LazyGrad._zeros = {}


class _LazySource:
    """Computes the real gradients behind one or two _LazyGrad objects, once, on first use; drops its inputs afterwards.
    `inputs` are the tensors the thunk reads: autograd's own saved-tensor check ran when the backward node unpacked them, so an
    in-place change made AFTER that (an optimizer step between autograd.grad(loss, [rast]) and the first look at the result)
    would go unnoticed -- their version counters are recorded here and compared at materialisation, with autograd's message."""
    __slots__ = ("thunk", "values", "inputs")

    def __init__(self, thunk, inputs=()):
        self.thunk, self.values = thunk, None
        self.inputs = tuple((t, t._version) for t in inputs if t is not None)

    def get(self):
        if self.values is None:
            for t, version in self.inputs:
                if t._version != version:
                    raise RuntimeError("one of the variables needed for gradient computation has been modified by an inplace operation: "
                                       "a tensor read by the deferred gradient of rast is at version %d; expected version %d (the "
                                       "gradient was requested before the change and first looked at after it)" % (t._version, version))
            self.values, self.thunk, self.inputs = self.thunk(), None, ()
            fused_backward_count("materialized")
        return self.values


class _FwdOrigin:
    """What the record of a `rast` tensor remembers of the rasterize_fwd_cuda call that wrote it (see the section comment)."""
    __slots__ = ("pos", "pos_version", "tri", "tri_version", "state", "rast_ptr", "rast_version", "rast_shape",
                 "db_ptr", "db_version", "interpolations", "pending")

    def __init__(self, pos, tri, state, rast, rast_db):
        self.pos, self.pos_version, self.tri, self.tri_version, self.state = pos, pos._version, tri, tri._version, state
        self.rast_ptr, self.rast_version, self.rast_shape = rast.data_ptr(), rast._version, tuple(rast.shape)
        self.db_ptr, self.db_version = rast_db.data_ptr(), rast_db._version
        self.interpolations = 0            # interpolate_fwd[_da] calls that read this rast
        self.pending = None                # (weak ref to the g_rast stand-in, g_pos, weak ref to the g_rast_db stand-in or None)

    def is_rast(self, t):
        return t.data_ptr() == self.rast_ptr and t._version == self.rast_version and tuple(t.shape) == self.rast_shape

    def usable_by(self, attr, rast, tri, rast_db):
        return (_fused["mode"] == "auto" and self.state.fused_disabled != _fused_epoch[0] and self.interpolations == 1
                and self.pos.requires_grad and rast.requires_grad and self.is_rast(rast)
                and self.pos._version == self.pos_version and self.tri._version == self.tri_version
                and (rast_db is None or (rast_db.data_ptr() == self.db_ptr and rast_db._version == self.db_version))
                and tri.data_ptr() == self.tri.data_ptr() and tri.shape == self.tri.shape
                and attr.shape[-2] == self.pos.shape[-2])


def _origin_of(rast):
    rec = _record_of(rast, "rast")
    return None if rec is None else rec.origin


def _fused_interpolate_grad(org, attr, rast, tri, dy, rast_db, dda, diff_attrs_all, diff_attrs_vec, tile_flags):
    """interpolate_grad[_da] of a rast straight from rasterize_fwd_cuda: position gradient prepared in the same pass."""
    g_attr, _, _, g_pos = interpolate_rasterize_grad(attr, rast, tri, org.pos, dy, with_g_rast=False, tile_flags=tile_flags,
                                                     rast_db=rast_db, dda=dda, diff_attrs_all=diff_attrs_all,
                                                     diff_attrs_vec=diff_attrs_vec, db_to_pos=True)
    if rast_db is None:
        source = _LazySource(lambda: interpolate_grad_da(attr, rast, tri, dy, None, None, False, [], tile_flags, fuse=False)[1:2],
                             (attr, rast, tri, dy))
        g_rast, g_rast_db = _LazyGrad(rast, source, 0), None
    else:
        source = _LazySource(lambda: interpolate_grad_da(attr, rast, tri, dy, rast_db, dda, diff_attrs_all, diff_attrs_vec, tile_flags,
                                                         fuse=False)[1:], (attr, rast, tri, dy, rast_db, dda))
        g_rast, g_rast_db = _LazyGrad(rast, source, 0), _LazyGrad(rast_db, source, 1)
        g_rast_db._origin = org
    g_rast._origin = org                   # (the stand-in keeps the exchange alive: `rast` itself may be gone when rasterize's node runs)
    org.pending = (weakref.ref(g_rast), g_pos, None if g_rast_db is None else weakref.ref(g_rast_db))
    return g_attr, g_rast, g_rast_db


def _take_prepared(fn, pos, tri, out, dy, ddb, call):
    """rasterize_grad[_db] side of the exchange: the prepared position gradient if (dy, ddb) are what interpolate_grad[_da] handed
    out, else None (after which the caller materialises any stand-in and computes as usual).  `call(dy, ddb, grad)` launches
    nvdr_rasterize_grad into an existing gradient buffer."""
    lazy = isinstance(dy, _LazyGrad)
    # the stand-in carries the exchange (the caller's `rast` may be gone by now); an ordinary tensor arriving while something is
    # prepared for this rast -- autograd summed rast's gradient with somebody else's -- voids it
    org = getattr(dy, "_origin", None) if lazy else _origin_of(out)
    if org is None or org.pending is None:
        return None
    (w_rast, g_pos, w_db), org.pending = org.pending, None
    lz_rast, lz_db = w_rast(), (None if w_db is None else w_db())
    ok = (lazy and lz_rast is dy and dy.unedited() and org.is_rast(out)
          and pos.data_ptr() == org.pos.data_ptr() and pos._version == org.pos_version and pos.shape == org.pos.shape
          and tri.data_ptr() == org.tri.data_ptr() and tri._version == org.tri_version and tri.shape == org.tri.shape)
    if ok and w_db is not None:
        # prepared WITH rast_db's gradient folded in: stands only if that is what arrives for ddb (a caller whose rasterize ran
        # with grad_db=False calls rasterize_grad: ddb None)
        ok = ddb is not None and ddb is lz_db and lz_db.unedited()
    elif ok and ddb is not None:
        # prepared from dy alone; ddb's share is added now (linearity) -- unless it is a stand-in of some other exchange
        ok = not isinstance(ddb, _LazyGrad)
        if ok:
            call(None, ddb, g_pos)
    if ok:
        fused_backward_count("used")
        return g_pos
    fused_backward_count("discarded")
    if org.state.fused_disabled != _fused_epoch[0]:
        _log_info("fused rasterize/interpolate backward switched off on this context: rast's gradient has other contributors "
                  "(set_fused_backward('auto') re-arms it)")
    org.state.fused_disabled = _fused_epoch[0]
    return None


_graph_task_id = getattr(torch._C, "_current_graph_task_id", None)


def _in_backward():
    """True while autograd's engine is running a backward pass on this thread."""
    return _graph_task_id is not None and _graph_task_id() >= 0


def rasterize_grad_db(pos, tri, out, dy, ddb, tile_flags=None):
    """torch_rasterize.cpp:171-256.  ``ddb`` may be None (== rasterize_grad).  ``dy`` may be None as well -- what a caller whose
    autograd function runs with set_materialize_grads(False) passes for an unused `rast`: only ddb's
    share is computed; with both None the gradient is zero."""
    fn = "rasterize_grad_db"
    enable_db = ddb is not None
    given = {"dy": dy} if dy is not None else {}             # the upstream gradients that are there
    if enable_db:
        given["ddb"] = ddb
    dev = _check_device(fn, pos=pos, tri=tri, out=out, **given)
    _check_contiguous(fn, pos=pos, tri=tri, out=out)
    _check_f32(fn, pos=pos, out=out, **given)
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
    if dy is not None:
        _require(tuple(dy.shape) == (depth, height, width, 4), fn, "dy must have shape [depth, height, width, 4]")
    if enable_db:
        _require(tuple(ddb.shape) == (depth, height, width, 4), fn, "ddb must have shape [depth, height, width, 4]")

    V = pos.size(1) if instance_mode else pos.size(0)
    tile_flags = _auto_flags(fn, tile_flags, "rast", out)

    def call(dy, ddb, grad):
        dy_ = None if dy is None else dy.contiguous()
        ddb_ = None if ddb is None else ddb.contiguous()
        with _on_device(dev):
            rc = _capi.load().nvdr_rasterize_grad(pos.data_ptr(), tri.data_ptr(), out.data_ptr(), _capi.ptr(dy_),
                                                  _capi.ptr(ddb_), int(instance_mode), depth, V, tri.size(0),
                                                  height, width, grad.data_ptr(), _flags_ok(fn, tile_flags, depth, height, width, dev),
                                                  _stream(dev))
        _capi.check(rc, fn)

    # gradients handed out by interpolate_grad[_da] of this module (section "fused backward" above)
    grad = _take_prepared(fn, pos, tri, out, dy, ddb, call)
    if grad is not None:
        return grad
    dy = dy.materialize() if isinstance(dy, _LazyGrad) else dy
    ddb = ddb.materialize() if isinstance(ddb, _LazyGrad) else ddb
    with _on_device(dev):
        grad = torch.zeros_like(pos)
    if dy is not None or ddb is not None:
        call(dy, ddb, grad)
    return grad


def rasterize_grad(pos, tri, out, dy, tile_flags=None):
    """torch_rasterize.cpp:259-263."""
    return rasterize_grad_db(pos, tri, out, dy, None, tile_flags)


# ----------------------------------------------------------------------------- interpolate

_IP_MAX_DIFF_ATTRS = 32   # csrc/common/interpolate.h:18


def _diff_list(diff_attrs_vec):
    import ctypes
    n = len(diff_attrs_vec)
    arr = (ctypes.c_int32 * max(n, 1))(*[int(x) for x in diff_attrs_vec])
    return arr, n


def interpolate_fwd_da(attr, rast, tri, rast_db, diff_attrs_all, diff_attrs_vec, tile_flags=None):
    """torch_interpolate.cpp:42-124."""
    fn = "interpolate_fwd_da"
    enable_da = (rast_db is not None) and (bool(diff_attrs_all) or len(diff_attrs_vec) > 0)
    h = host_layer()
    if h is not None and tile_flags is None:                 # (the caller leaves the flags to the records: so does the compiled layer)
        served = _host_raw(h.interpolate, attr, rast, tri, rast_db if enable_da else None, bool(diff_attrs_all), [int(x) for x in diff_attrs_vec])
        if served is not None:
            org = _origin_of(rast)
            if org is not None:
                org.interpolations += 1
            return served
    instance_mode = attr.dim() > 2
    dev = attr.device
    f32 = torch.float32
    if (attr.is_cuda and rast.device == dev and tri.device == dev and attr.dtype is f32 and rast.dtype is f32 and tri.dtype is torch.int32
            and attr.is_contiguous() and rast.is_contiguous() and tri.is_contiguous()
            and (not enable_da or (rast_db.device == dev and rast_db.dtype is f32 and rast_db.is_contiguous()))):
        pass                                   # (the common case in one expression; otherwise the reference's checks and messages)
    elif enable_da:
        dev = _check_device(fn, attr=attr, rast=rast, tri=tri, rast_db=rast_db)
        _check_contiguous(fn, attr=attr, rast=rast, tri=tri, rast_db=rast_db)
        _check_f32(fn, attr=attr, rast=rast, rast_db=rast_db)
        _check_i32(fn, tri=tri)
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
    tile_flags = _auto_flags(fn, tile_flags, "rast", rast)
    org = _origin_of(rast)
    if org is not None:
        org.interpolations += 1
    with _on_device(dev):
        out = torch.empty((N, H, W, A), dtype=torch.float32, device=dev)
        out_da = torch.empty((N, H, W, 2 * D), dtype=torch.float32, device=dev)
        rc = _capi.load().nvdr_interpolate_fwd(attr.data_ptr(), rast.data_ptr(), tri.data_ptr(),
                                               rast_db.data_ptr() if enable_da else None,
                                               int(instance_mode), attr.size(0) if instance_mode else 1,
                                               N, V, A, tri.size(0), H, W,
                                               int(bool(diff_attrs_all)), lst, nlst,
                                               out.data_ptr(), out_da.data_ptr() if enable_da else None,
                                               _flags_ok(fn, tile_flags, N, H, W, dev), _stream(dev))
    _capi.check(rc, fn)
    if tile_flags is not None:
        # zeros are written where no triangle is visible: the rasterizer's empty tiles are tiles of zeros in both outputs, and
        # texture() need not read them there (while the tensors stay what they are now)
        _attach_tiles(out, tile_flags, "zero")
        if out_da.numel():
            _attach_tiles(out_da, tile_flags, "zero")
    return out, out_da


def interpolate_fwd(attr, rast, tri, tile_flags=None):
    """torch_interpolate.cpp:127-132."""
    return interpolate_fwd_da(attr, rast, tri, None, False, [], tile_flags)


def interpolate_grad_da(attr, rast, tri, dy, rast_db, dda, diff_attrs_all, diff_attrs_vec, tile_flags=None, fuse=None):
    """torch_interpolate.cpp:137-239.  `fuse` (not in the reference): None = prepare rasterize_grad's result in the same pass when
    `rast` came straight from rasterize_fwd_cuda (section "fused backward" above); False = never."""
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
    tile_flags = _auto_flags(fn, tile_flags, "rast", rast)
    # The exchange is armed only INSIDE an autograd backward pass: a direct call of this entry point -- a caller that
    # wants g_rast's memory, a custom op reading data_ptr() -- gets the reference's three ordinary tensors.
    if fuse is None and _fused["mode"] == "auto" and _in_backward():
        org = _origin_of(rast)
        if org is not None:
            org.pending = None             # left over from a backward pass that never reached rasterize_grad: void
            if org.usable_by(attr, rast, tri, rast_db if enable_da else None):
                return _fused_interpolate_grad(org, attr, rast, tri, dy, rast_db if enable_da else None, dda if enable_da else None,
                                               diff_attrs_all, diff_attrs_vec, tile_flags)
    with _on_device(dev):
        g_attr = torch.zeros_like(attr)
        g_rast = torch.empty_like(rast)
        g_rast_db = torch.empty_like(rast_db) if enable_da else None
        rc = _capi.load().nvdr_interpolate_grad(attr.data_ptr(), rast.data_ptr(), tri.data_ptr(), dy_.data_ptr(),
                                                rast_db.data_ptr() if enable_da else None, _capi.ptr(dda_),
                                                int(instance_mode), attr_depth, N, V, A, tri.size(0), H, W,
                                                int(bool(diff_attrs_all)), lst, nlst,
                                                g_attr.data_ptr(), g_rast.data_ptr(), _capi.ptr(g_rast_db),
                                                _flags_ok(fn, tile_flags, N, H, W, dev), _stream(dev))
    _capi.check(rc, fn)
    return g_attr, g_rast, g_rast_db


def interpolate_grad(attr, rast, tri, dy, tile_flags=None, fuse=None):
    """torch_interpolate.cpp:242-248."""
    g_attr, g_rast, _ = interpolate_grad_da(attr, rast, tri, dy, None, None, False, [], tile_flags, fuse)
    return g_attr, g_rast


def interpolate_rasterize_grad(attr, rast, tri, pos, dy, with_g_rast=True, tile_flags=None,
                               rast_db=None, dda=None, diff_attrs_all=False, diff_attrs_vec=(), db_to_pos=True):
    """Not in the reference's module: interpolate_grad[_da] (torch_interpolate.cpp:137-248) and rasterize_grad[_db]
    (torch_rasterize.cpp:171-263) of the graph rasterize -> interpolate in ONE kernel (csrc/backward_fused.hip).
    -> (g_attr, g_rast or None, g_rast_db or None, g_pos), equal to
        g_attr, g_rast[, g_rast_db] = interpolate_grad[_da](attr, rast, tri, dy[, rast_db, dda, ...])
        g_pos = rasterize_grad[_db](pos, tri, rast, g_rast[, g_rast_db])
    up to the summation order of the atomics.  `with_g_rast=False` skips writing g_rast / g_rast_db (only legal when nothing
    else consumes the gradient of rast).  `db_to_pos=False`: rast_db's gradient is not propagated to pos (grad_db=False).
    attr and pos must index the same vertices with the same `tri`."""
    fn = "interpolate_rasterize_grad"
    enable_da = (rast_db is not None) and (dda is not None) and (bool(diff_attrs_all) or len(diff_attrs_vec) > 0)
    dev = attr.device
    f32 = torch.float32
    if (attr.is_cuda and rast.device == dev and tri.device == dev and pos.device == dev and dy.device == dev
            and attr.dtype is f32 and rast.dtype is f32 and pos.dtype is f32 and dy.dtype is f32 and tri.dtype is torch.int32
            and attr.is_contiguous() and rast.is_contiguous() and tri.is_contiguous() and pos.is_contiguous()
            and (not enable_da or (rast_db.device == dev and dda.device == dev and rast_db.dtype is f32 and dda.dtype is f32
                                   and rast_db.is_contiguous()))):
        pass                                   # (the common case in one expression; otherwise check by check, for the messages)
    elif enable_da:
        dev = _check_device(fn, attr=attr, rast=rast, tri=tri, pos=pos, dy=dy, rast_db=rast_db, dda=dda)
        _check_contiguous(fn, attr=attr, rast=rast, tri=tri, pos=pos, rast_db=rast_db)
        _check_f32(fn, attr=attr, rast=rast, pos=pos, dy=dy, rast_db=rast_db, dda=dda)
        _check_i32(fn, tri=tri)
    else:
        dev = _check_device(fn, attr=attr, rast=rast, tri=tri, pos=pos, dy=dy)
        _check_contiguous(fn, attr=attr, rast=rast, tri=tri, pos=pos)
        _check_f32(fn, attr=attr, rast=rast, pos=pos, dy=dy)
        _check_i32(fn, tri=tri)
    attr_instance = attr.dim() > 2
    pos_instance = pos.dim() > 2
    _require(rast.dim() == 4 and rast.size(0) > 0 and rast.size(1) > 0 and rast.size(2) > 0 and rast.size(3) == 4, fn,
             "rast must have shape[>0, >0, >0, 4]")
    _require(tri.dim() == 2 and tri.size(0) > 0 and tri.size(1) == 3, fn, "tri must have shape [>0, 3]")
    _require(attr.dim() in (2, 3) and attr.size(0) > 0 and attr.size(1) > 0 and (attr.dim() == 2 or attr.size(2) > 0), fn,
             "attr must have shape [>0, >0, >0] or [>0, >0]")
    N, H, W = rast.size(0), rast.size(1), rast.size(2)
    if pos_instance:
        _require(pos.dim() == 3 and pos.size(0) == N and pos.size(1) > 0 and pos.size(2) == 4, fn, "pos must have shape [depth, >0, 4]")
    else:
        _require(pos.dim() == 2 and pos.size(0) > 0 and pos.size(1) == 4, fn, "pos must have shape [>0, 4]")
    attr_depth = attr.size(0) if attr_instance else 1
    V = attr.size(1 if attr_instance else 0)
    A = attr.size(2 if attr_instance else 1)
    _require(V == pos.size(1 if pos_instance else 0), fn, "attr and pos must have the same number of vertices")
    _require(dy.dim() == 4 and tuple(dy.shape) == (N, H, W, A), fn, "dy must have shape [depth, height, width, attributes]")
    _require(attr_depth == N or attr_depth == 1, fn, "minibatch size mismatch between inputs rast, dy, attr")
    D = 0
    if enable_da:
        D = A if diff_attrs_all else len(diff_attrs_vec)
        _require(tuple(rast_db.shape) == (N, H, W, 4), fn, "rast_db must have shape[>0, >0, >0, 4]")
        _require(dda.dim() == 4 and tuple(dda.shape) == (N, H, W, 2 * D), fn, "dda must have shape [>0, height, width, ?]")
        if not diff_attrs_all:
            _require(len(diff_attrs_vec) <= _IP_MAX_DIFF_ATTRS, fn, "too many entries in diff_attrs list (increase IP_MAX_DIFF_ATTRS)")
    dy_ = dy.contiguous()
    dda_ = dda.contiguous() if enable_da else None
    lst, nlst = _diff_list([] if (diff_attrs_all or not enable_da) else diff_attrs_vec)
    with _on_device(dev):
        # both zero-initialised gradients from ONE buffer: one fill launch instead of two (small batches are launch-bound).
        # Consequence: when autograd adopts them, attr.grad and pos.grad are views into one allocation -- pos.grad has
        # a non-zero storage_offset and each keeps the other's memory alive, unlike the separate tensors of the two-kernel path.
        # Optimizers and in-place gradient arithmetic do not notice; code that works on .grad's untyped_storage() would.
        tile_flags = _auto_flags(fn, tile_flags, "rast", rast)
        na = (attr.numel() + 3) & ~3                       # keeps g_pos 16-byte aligned
        zeros = torch.zeros((na + pos.numel(),), dtype=torch.float32, device=dev)
        g_attr = zeros[:attr.numel()].view(attr.shape)
        g_pos = zeros[na:].view(pos.shape)
        g_rast = torch.empty_like(rast) if with_g_rast else None
        g_rast_db = torch.empty_like(rast_db) if (with_g_rast and enable_da) else None
        rc = _capi.load().nvdr_interpolate_rasterize_grad(attr.data_ptr(), rast.data_ptr(), tri.data_ptr(), pos.data_ptr(), dy_.data_ptr(),
                                                          int(attr_instance), attr_depth, int(pos_instance), N, V, A, tri.size(0), H, W,
                                                          rast_db.data_ptr() if enable_da else None, _capi.ptr(dda_),
                                                          int(bool(diff_attrs_all)), lst, nlst, int(bool(db_to_pos)),
                                                          g_attr.data_ptr(), g_pos.data_ptr(), _capi.ptr(g_rast), _capi.ptr(g_rast_db),
                                                          _flags_ok(fn, tile_flags, N, H, W, dev), _stream(dev))
    _capi.check(rc, fn)
    return g_attr, g_rast, g_rast_db, g_pos


# ----------------------------------------------------------------------------- texture

_TEX_MAX_LEVELS = 17
_TEX_GRAD_SCRATCH = True      # texture_grad_*: bring scratch for the two-level reduction of constant-uv regions (tests switch it off)
_FILTER_NEAREST, _FILTER_LINEAR, _FILTER_LMN, _FILTER_LML = 0, 1, 2, 3
_BOUNDARY_CUBE = 0


class TextureMipWrapper:
    """Opaque mip stack (reference: csrc/torch/torch_types.h:28-35): one flat f32 tensor holding
    levels 1..L plus the metadata the consistency checks need (torch_texture.cpp:309-310)."""

    def __init__(self):
        self.mip = None
        self.max_mip_level = 0
        self.texture_size = []
        self.cube_mode = False


def _mip_info(tex_shape, cube_mode, max_mip_level, fn):
    import ctypes
    lw = (ctypes.c_int * _TEX_MAX_LEVELS)(); lh = (ctypes.c_int * _TEX_MAX_LEVELS)()
    off = (ctypes.c_int64 * _TEX_MAX_LEVELS)(); total = ctypes.c_int64(0)
    n, h, w, c = (tex_shape[0], tex_shape[2], tex_shape[3], tex_shape[4]) if cube_mode else tuple(tex_shape)
    L = _capi.load().nvdr_texture_mip_info(int(n), int(h), int(w), int(c), int(cube_mode), int(max_mip_level), lw, lh, off, ctypes.byref(total))
    if L < 0:
        # texture.cpp:15-60 raiseMipSizeError
        _fail(fn, f"unsupported texture size {w}x{h}: every mip level must have even (or unit) extents; "
                  f"use power-of-two sizes or limit max_mip_level")
    return L, list(lw[:L + 1]), list(lh[:L + 1]), list(off[:L + 1]), int(total.value)


def _check_tex_shape(fn, tex, cube_mode):
    if not cube_mode:
        _require(tex.dim() == 4 and all(s > 0 for s in tex.shape), fn, "tex must have shape[>0, >0, >0, >0]")
    else:
        _require(tex.dim() == 5 and tex.size(0) > 0 and tex.size(1) == 6 and tex.size(2) > 0 and tex.size(3) > 0 and tex.size(4) > 0,
                 fn, "tex must have shape[>0, 6, >0, >0, >0] in cube map mode")
        _require(tex.size(2) == tex.size(3), fn, "texture shape must be square in cube map mode")


def _tex_dims(tex, cube_mode):
    """(depth, height, width, channels) of a 2D texture [n,h,w,c] or a cube map [n,6,s,s,c]."""
    if cube_mode:
        return tex.size(0), tex.size(2), tex.size(3), tex.size(4)
    return tex.size(0), tex.size(1), tex.size(2), tex.size(3)


def texture_construct_mip(tex, max_mip_level, cube_mode):
    """torch_texture.cpp:98-169."""
    fn = "texture_construct_mip"
    h = host_layer()
    if h is not None:
        mip = h.construct_mip(tex, int(max_mip_level), bool(cube_mode))      # None: not the ordinary case -> the checks below word it
        if mip is not None:
            w = TextureMipWrapper()
            w.mip, w.max_mip_level, w.texture_size, w.cube_mode = mip, int(max_mip_level), list(tex.shape), bool(cube_mode)
            return w
    _require(max_mip_level >= -1, fn, "invalid max_mip_level")
    dev = _check_device(fn, tex=tex)
    _check_contiguous(fn, tex=tex)
    _check_f32(fn, tex=tex)
    _check_tex_shape(fn, tex, cube_mode)
    L, lw, lh, off, total = _mip_info(tex.shape, cube_mode, max_mip_level, fn)
    with _on_device(dev):
        mip = torch.empty((total,), dtype=torch.float32, device=dev)
        tn, th, tw, tc = _tex_dims(tex, cube_mode)
        rc = _capi.load().nvdr_texture_construct_mip(tex.data_ptr(), tn, th, tw, tc,
                                                     int(cube_mode), int(max_mip_level), mip.data_ptr(), _stream(dev))
    _capi.check(rc, fn)
    w = TextureMipWrapper()
    w.mip = mip
    w.max_mip_level = int(max_mip_level)
    w.texture_size = list(tex.shape)
    w.cube_mode = bool(cube_mode)
    return w


def _has(t):
    return t is not None and t.numel() > 0


def _texture_common(fn, tex, uv, uv_da, mip_level_bias, mip_wrapper, mip_stack, filter_mode, boundary_mode, grad_mips):
    """Validation shared by forward and backward (torch_texture.cpp:174-343, 421-604).
    Returns (dev, enable_mip, has_uv_da, has_bias, level tensors/views, grad level tensors, grad_mip flat)."""
    _require(0 <= filter_mode < 4, fn, "filter_mode unsupported")
    _require(0 <= boundary_mode < 4, fn, "boundary_mode unsupported")
    enable_mip = filter_mode in (_FILTER_LMN, _FILTER_LML)
    has_stack = len(mip_stack) > 0
    mip_w = mip_wrapper.mip if mip_wrapper is not None else None
    max_mip_level = len(mip_stack) if has_stack else (mip_wrapper.max_mip_level if mip_wrapper is not None else 0)
    has_uv_da, has_bias = _has(uv_da), _has(mip_level_bias)
    if enable_mip:
        _require(max_mip_level >= -1, fn, "invalid max_mip_level")
        _require(has_uv_da or has_bias, fn, "mipmapping filter mode requires uv_da and/or mip_level_bias input")
        _require(has_stack or mip_w is not None, fn, "mipmapping filter mode requires mip wrapper or mip stack input")
    dev = _check_device(fn, tex=tex, uv=uv)
    _check_contiguous(fn, tex=tex, uv=uv)
    _check_f32(fn, tex=tex, uv=uv)
    if enable_mip:
        if has_stack:
            for t in mip_stack:
                if not t.is_cuda or t.device != dev:
                    _fail(fn, "Mip stack inputs must reside on the correct GPU device")
                if not t.is_contiguous():
                    _fail(fn, "Mip stack inputs must be contiguous tensors")
                if t.dtype != torch.float32:
                    _fail(fn, "Mip stack inputs must be float32 tensors")
        else:
            _check_device(fn, mip_w=mip_w); _check_contiguous(fn, mip_w=mip_w); _check_f32(fn, mip_w=mip_w)
        if has_uv_da:
            _check_device(fn, uv_da=uv_da); _check_contiguous(fn, uv_da=uv_da); _check_f32(fn, uv_da=uv_da)
        if has_bias:
            _check_device(fn, mip_level_bias=mip_level_bias); _check_contiguous(fn, mip_level_bias=mip_level_bias)
            _check_f32(fn, mip_level_bias=mip_level_bias)

    cube_mode = boundary_mode == _BOUNDARY_CUBE
    _check_tex_shape(fn, tex, cube_mode)
    if not cube_mode:
        _require(uv.dim() == 4 and uv.size(0) > 0 and uv.size(1) > 0 and uv.size(2) > 0 and uv.size(3) == 2, fn,
                 "uv must have shape [>0, >0, >0, 2]")
    else:
        _require(uv.dim() == 4 and uv.size(0) > 0 and uv.size(1) > 0 and uv.size(2) > 0 and uv.size(3) == 3, fn,
                 "uv must have shape [>0, >0, >0, 3] in cube map mode")
    tn, th, tw, tc = _tex_dims(tex, cube_mode)
    _require(tn == 1 or tn == uv.size(0), fn, "minibatch size mismatch between inputs tex, uv")
    _require(tw <= (1 << 16) and th <= (1 << 16), fn, "texture size too large")
    n, H, W = uv.size(0), uv.size(1), uv.size(2)
    if enable_mip:
        if has_uv_da:
            if not cube_mode:
                _require(uv_da.dim() == 4 and tuple(uv_da.shape) == (n, H, W, 4), fn,
                         "uv_da must have shape [minibatch_size, height, width, 4]")
            else:
                _require(uv_da.dim() == 4 and tuple(uv_da.shape) == (n, H, W, 6), fn,
                         "uv_da must have shape [minibatch_size, height, width, 6] in cube map mode")
        if has_bias:
            _require(mip_level_bias.dim() == 3 and tuple(mip_level_bias.shape) == (n, H, W), fn,
                     "mip_level_bias must have shape [minibatch_size, height, width]")

    faces = 6 if cube_mode else 1
    levels, g_levels, g_flat = [], [], None
    if enable_mip:
        if has_stack:
            for i, t in enumerate(mip_stack, start=1):
                sw, sh = max(tw >> i, 1), max(th >> i, 1)
                if not cube_mode:
                    _require(t.dim() == 4 and t.size(0) == tn and t.size(1) == sh and t.size(2) == sw and t.size(3) == tc,
                             fn, "mip level size mismatch in custom mip stack")
                else:
                    _require(t.dim() == 5 and t.size(0) == tn and t.size(1) == 6 and t.size(2) == sh and t.size(3) == sw and t.size(4) == tc,
                             fn, "mip level size mismatch in mip stack")
                if sw == 1 and sh == 1:
                    _require(i == len(mip_stack), fn, "mip level size mismatch in mip stack")
                levels.append(t)
            _require(len(levels) < _TEX_MAX_LEVELS, fn, "too many levels in custom mip stack")
            if grad_mips:
                g_levels = [torch.zeros_like(t) for t in mip_stack]
        else:
            L, lw, lh, off, total = _mip_info(tex.shape, cube_mode, max_mip_level, fn)
            _require(list(tex.shape) == list(mip_wrapper.texture_size) and cube_mode == mip_wrapper.cube_mode, fn,
                     "mip does not match texture size")
            _require(mip_w.dim() == 1 and mip_w.size(0) == total, fn, "wrapped mip tensor size mismatch")
            cnt = [tn * faces * lh[i] * lw[i] * tc for i in range(L + 1)]
            levels = [mip_w[off[i]:off[i] + cnt[i]] for i in range(1, L + 1)]
            if grad_mips:
                g_flat = torch.zeros_like(mip_w)
                g_levels = [g_flat[off[i]:off[i] + cnt[i]] for i in range(1, L + 1)]
    return dev, enable_mip, has_uv_da, has_bias, levels, g_levels, g_flat


def texture_fwd_mip(tex, uv, uv_da, mip_level_bias, mip_wrapper, mip_stack, filter_mode, boundary_mode, tile_flags=None):
    """torch_texture.cpp:174-407."""
    fn = "texture_fwd_mip"
    h = host_layer()
    if h is not None and tile_flags is None and len(mip_stack) == 0:
        mw = mip_wrapper if (mip_wrapper is not None and mip_wrapper.mip is not None) else None
        served = _host_raw(h.texture, tex, uv, uv_da if _has(uv_da) else None, mip_level_bias if _has(mip_level_bias) else None,
                           None if mw is None else mw.mip, 0 if mw is None else int(mw.max_mip_level),
                           [] if mw is None else [int(x) for x in mw.texture_size], False if mw is None else bool(mw.cube_mode),
                           int(filter_mode), int(boundary_mode), _TEX_GRAD_SCRATCH)
        if served is not None:
            return served
    dev, enable_mip, has_uv_da, has_bias, levels, _, _ = _texture_common(
        fn, tex, uv, uv_da, mip_level_bias, mip_wrapper, list(mip_stack), filter_mode, boundary_mode, False)
    tn, th, tw, C = _tex_dims(tex, boundary_mode == _BOUNDARY_CUBE)
    n, H, W = uv.size(0), uv.size(1), uv.size(2)
    ptrs, L = _capi.ptr_array(levels)
    tile_flags = None if boundary_mode == _BOUNDARY_CUBE else \
        _auto_flags(fn, tile_flags, "zero", *([uv, uv_da] if (enable_mip and has_uv_da) else [uv]))
    with _on_device(dev):
        out = torch.empty((n, H, W, C), dtype=torch.float32, device=dev)
        rc = _capi.load().nvdr_texture_fwd(tex.data_ptr(), ptrs, L, uv.data_ptr(),
                                           uv_da.data_ptr() if (enable_mip and has_uv_da) else None,
                                           mip_level_bias.data_ptr() if (enable_mip and has_bias) else None,
                                           tn, th, tw, C, n, H, W,
                                           int(filter_mode), int(boundary_mode), out.data_ptr(),
                                           _flags_ok(fn, tile_flags, n, H, W, dev), _stream(dev))
    _capi.check(rc, fn)
    return out


def texture_fwd(tex, uv, filter_mode, boundary_mode, tile_flags=None):
    """torch_texture.cpp:411-416."""
    return texture_fwd_mip(tex, uv, None, None, None, [], filter_mode, boundary_mode, tile_flags)


def texture_grad_linear_mipmap_linear(tex, uv, dy, uv_da, mip_level_bias, mip_wrapper, mip_stack, filter_mode, boundary_mode, tile_flags=None):
    """torch_texture.cpp:421-690 -> (g_tex, g_uv, g_uv_da, g_mip_level_bias, [g_mip...])."""
    fn = "texture_grad_linear_mipmap_linear"
    mip_stack = list(mip_stack)
    dev, enable_mip, has_uv_da, has_bias, levels, g_levels, g_flat = _texture_common(
        fn, tex, uv, uv_da, mip_level_bias, mip_wrapper, mip_stack, filter_mode, boundary_mode, True)
    _check_device(fn, dy=dy)
    _check_f32(fn, dy=dy)
    tn, th, tw, C = _tex_dims(tex, boundary_mode == _BOUNDARY_CUBE)
    n, H, W = uv.size(0), uv.size(1), uv.size(2)
    _require(dy.dim() == 4 and tuple(dy.shape) == (n, H, W, C), fn, "dy must have shape [minibatch_size, height, width, channels]")
    dy_ = dy.contiguous()
    has_stack = len(mip_stack) > 0
    ptrs, L = _capi.ptr_array(levels)
    gptrs, _ = _capi.ptr_array(g_levels)
    with _on_device(dev):
        g_tex = torch.zeros_like(tex)
        g_uv = g_uv_da = g_bias = None
        if filter_mode != _FILTER_NEAREST:
            g_uv = torch.empty_like(uv)
            if filter_mode == _FILTER_LML:
                if has_uv_da:
                    g_uv_da = torch.empty_like(uv_da)
                if has_bias:
                    g_bias = torch.empty_like(mip_level_bias)
        # scratch for the two-level reduction of constant-uv regions (include/nvdr_hip.h); not worth a second launch for
        # images of a few blocks
        lib = _capi.load()
        tile_flags = None if boundary_mode == _BOUNDARY_CUBE else \
            _auto_flags(fn, tile_flags, "zero", *([uv, uv_da] if (enable_mip and has_uv_da) else [uv]))
        scratch = None
        if _TEX_GRAD_SCRATCH and filter_mode != _FILTER_NEAREST and boundary_mode != _BOUNDARY_CUBE and n * H * W >= 4096:
            scratch = torch.empty((lib.nvdr_texture_grad_scratch_bytes(n, H, W, C) // 4,), dtype=torch.int32, device=dev)
        rc = _capi.load().nvdr_texture_grad(tex.data_ptr(), ptrs, L, uv.data_ptr(),
                                            uv_da.data_ptr() if (enable_mip and has_uv_da) else None,
                                            mip_level_bias.data_ptr() if (enable_mip and has_bias) else None,
                                            dy_.data_ptr(), tn, th, tw, C, n, H, W,
                                            int(filter_mode), int(boundary_mode), int(enable_mip and not has_stack),
                                            g_tex.data_ptr(), gptrs, _capi.ptr(g_uv), _capi.ptr(g_uv_da), _capi.ptr(g_bias),
                                            _capi.ptr(scratch), 0 if scratch is None else scratch.numel() * 4,
                                            _flags_ok(fn, tile_flags, n, H, W, dev), _stream(dev))
    _capi.check(rc, fn)
    return g_tex, g_uv, g_uv_da, g_bias, (g_levels if has_stack else [])


def texture_grad_nearest(tex, uv, dy, filter_mode, boundary_mode, tile_flags=None):
    """torch_texture.cpp:692-698."""
    return texture_grad_linear_mipmap_linear(tex, uv, dy, None, None, None, [], filter_mode, boundary_mode, tile_flags)[0]


def texture_grad_linear(tex, uv, dy, filter_mode, boundary_mode, tile_flags=None):
    """torch_texture.cpp:700-706."""
    r = texture_grad_linear_mipmap_linear(tex, uv, dy, None, None, None, [], filter_mode, boundary_mode, tile_flags)
    return r[0], r[1]


def texture_grad_linear_mipmap_nearest(tex, uv, dy, uv_da, mip_level_bias, mip_wrapper, mip_stack, filter_mode, boundary_mode, tile_flags=None):
    """torch_texture.cpp:708-713."""
    r = texture_grad_linear_mipmap_linear(tex, uv, dy, uv_da, mip_level_bias, mip_wrapper, mip_stack, filter_mode, boundary_mode, tile_flags)
    return r[0], r[1], r[4]


# ----------------------------------------------------------------------------- antialias

class TopologyHashWrapper:
    """Opaque edge -> opposite-vertex table (reference: csrc/torch/torch_types.h:37-45)."""

    def __init__(self):
        self.ev_hash = None


def antialias_construct_topology_hash(tri):
    """torch_antialias.cpp:25-63."""
    fn = "antialias_construct_topology_hash"
    dev = _check_device(fn, tri=tri)
    _check_contiguous(fn, tri=tri)
    _check_i32(fn, tri=tri)
    _require(tri.dim() == 2 and tri.size(0) > 0 and tri.size(1) == 3, fn, "tri must have shape [>0, 3]")
    lib = _capi.load()
    nbytes = lib.nvdr_antialias_hash_bytes(tri.size(0))
    with _on_device(dev):
        ev_hash = torch.empty((nbytes // 4,), dtype=torch.int32, device=dev)        # cleared by the library
        rc = lib.nvdr_antialias_construct_topology_hash(tri.data_ptr(), tri.size(0), ev_hash.data_ptr(), nbytes, _stream(dev))
    _capi.check(rc, fn)
    w = TopologyHashWrapper()
    w.ev_hash = ev_hash
    return w


def _aa_checks(fn, color, rast, pos, tri, dy=None):
    instance_mode = pos.dim() > 2
    _require(color.dim() == 4 and all(s > 0 for s in color.shape), fn, "color must have shape[>0, >0, >0, >0]")
    _require(rast.dim() == 4 and rast.size(0) > 0 and rast.size(1) > 0 and rast.size(2) > 0 and rast.size(3) == 4, fn,
             "rast must have shape[>0, >0, >0, 4]")
    _require(tri.dim() == 2 and tri.size(0) > 0 and tri.size(1) == 3, fn, "tri must have shape [>0, 3]")
    _require(color.size(1) == rast.size(1) and color.size(2) == rast.size(2), fn,
             "color and rast inputs must have same spatial dimensions")
    if dy is not None:
        _require(dy.dim() == 4 and all(s > 0 for s in dy.shape), fn, "dy must have shape[>0, >0, >0, >0]")
        _require(color.size(1) == dy.size(1) and color.size(2) == dy.size(2) and color.size(3) == dy.size(3), fn,
                 "color and dy inputs must have same dimensions")
    if instance_mode:
        _require(pos.dim() == 3 and pos.size(0) > 0 and pos.size(1) > 0 and pos.size(2) == 4, fn,
                 "pos must have shape [>0, >0, 4] or [>0, 4]")
        _require(rast.size(0) == color.size(0) and pos.size(0) == color.size(0), fn,
                 "minibatch size mismatch between inputs color, rast, pos")
    else:
        _require(pos.dim() == 2 and pos.size(0) > 0 and pos.size(1) == 4, fn, "pos must have shape [>0, >0, 4] or [>0, 4]")
        _require(rast.size(0) == color.size(0), fn, "minibatch size mismatch between inputs color, rast")
    if dy is not None:
        _require(dy.size(0) == color.size(0), fn, "minibatch size mismatch between inputs dy, color, raster_out")
    return instance_mode


def antialias_fwd(color, rast, pos, tri, topology_hash_wrap, tile_flags=None):
    """torch_antialias.cpp:68-155 -> (out, work_buffer)."""
    fn = "antialias_fwd"
    topology_hash = topology_hash_wrap.ev_hash
    h = host_layer()
    if h is not None and tile_flags is None and topology_hash is not None:
        served = h.antialias_fwd_raw(color, rast, pos, tri, topology_hash)
        if served is not None:
            return served                                        # (out, work_buffer)
    dev = _check_device(fn, color=color, rast=rast, pos=pos, tri=tri, topology_hash=topology_hash)
    _check_contiguous(fn, color=color, rast=rast, pos=pos, tri=tri, topology_hash=topology_hash)
    _check_f32(fn, color=color, rast=rast, pos=pos)
    _check_i32(fn, tri=tri, topology_hash=topology_hash)
    instance_mode = _aa_checks(fn, color, rast, pos, tri)
    N, H, W, C = color.shape
    V = pos.size(1 if instance_mode else 0)
    lib = _capi.load()
    tile_flags = _auto_flags(fn, tile_flags, "rast", rast)
    with _on_device(dev):
        out = torch.empty_like(color)                                                   # the library copies color into it
        work_buffer = torch.empty((N * W * H * 8 + 4,), dtype=torch.float32, device=dev)
        rc = lib.nvdr_antialias_fwd(color.data_ptr(), rast.data_ptr(), pos.data_ptr(), tri.data_ptr(),
                                    topology_hash.data_ptr(), topology_hash.numel() * 4,
                                    int(instance_mode), N, V, tri.size(0), H, W, C,
                                    out.data_ptr(), work_buffer.data_ptr(), work_buffer.numel() * 4,
                                    _flags_ok(fn, tile_flags, N, H, W, dev), _stream(dev))
    _capi.check(rc, fn)
    return out, work_buffer


def antialias_grad(color, rast, pos, tri, dy, work_buffer):
    """torch_antialias.cpp:160-241 -> (g_color, g_pos)."""
    fn = "antialias_grad"
    dev = _check_device(fn, color=color, rast=rast, pos=pos, tri=tri, dy=dy, work_buffer=work_buffer)
    _check_contiguous(fn, color=color, rast=rast, pos=pos, tri=tri, work_buffer=work_buffer)
    _check_f32(fn, color=color, rast=rast, pos=pos, dy=dy, work_buffer=work_buffer)
    _check_i32(fn, tri=tri)
    instance_mode = _aa_checks(fn, color, rast, pos, tri, dy)
    N, H, W, C = color.shape
    V = pos.size(1 if instance_mode else 0)
    dy_ = dy.contiguous()
    with _on_device(dev):
        g_color = torch.empty_like(dy_)                                                  # the library copies dy into it
        g_pos = torch.zeros_like(pos)
        rc = _capi.load().nvdr_antialias_grad(color.data_ptr(), rast.data_ptr(), pos.data_ptr(), tri.data_ptr(),
                                              dy_.data_ptr(), work_buffer.data_ptr(), work_buffer.numel() * 4,
                                              int(instance_mode), N, V, tri.size(0), H, W, C,
                                              g_color.data_ptr(), g_pos.data_ptr(), _stream(dev))
    _capi.check(rc, fn)
    return g_color, g_pos
