"""The REFERENCE ITSELF on the CPU -- TEST INFRASTRUCTURE ONLY.

numpy front end over ``oracle/_ref/libnvdr_ref.so``: the reference's own C++/CUDA sources (glue, kernels and
the CudaRaster rasterizer) compiled unmodified for the host by ``oracle/refshim/build.py`` on top of a
CUDA-on-CPU shim.  This is what pins the repo's oracle (``oracle/*.c``) -- and through it the HIP kernels --
to the reference: tests compare ``oracle.X(...)`` with ``oracle.ref.X(...)`` on the same inputs.

Three layers:
  * ``Plugin``      -- the functions of the pybind module ``_nvdiffrast_c`` (csrc/torch/torch_bindings.cpp:43-71),
                       same names / argument order / return arity, on numpy arrays;
  * module level    -- ``rasterize``, ``interpolate``, ``texture``, ``antialias`` and their ``*_grad`` with the
                       argument conventions of the ``oracle`` package, for one-line comparisons in tests;
  * ``torch_plugin``-- ``Plugin`` on CPU torch tensors, good enough to run the reference's own
                       ``nvdiffrast/torch/ops.py`` on (see tests/golden/make_reference_fixture.py).

Only ``tests/``, ``__graft_entry__`` (build + smoke) and fixture generators may import this module.
The library is built where /root/reference exists and travels to the GPU box as a binary (oracle/_ref/ is
git-ignored but not gpurun-ignored); ``available()`` says whether it can be used.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_HERE, "_ref")
VARIANTS = {"fma": "libnvdr_ref.so", "nofma": "libnvdr_ref_nofma.so"}
_libs = {}


class _TensorDesc(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("dtype", ctypes.c_int), ("device", ctypes.c_int),
                ("ndim", ctypes.c_int), ("shape", ctypes.c_int64 * 8)]


def build(force=False, reference="/root/reference"):
    """Compile oracle/_ref from the reference checkout (needs /root/reference; ~40 s)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("nvdr_refshim_build", os.path.join(_HERE, "refshim", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build(reference=reference, force=force, verbose=True)


def available(variant="fma"):
    return os.path.exists(os.path.join(_DIR, VARIANTS[variant]))


def lib(variant="fma"):
    if variant not in _libs:
        path = os.path.join(_DIR, VARIANTS[variant])
        if not os.path.exists(path):
            raise FileNotFoundError("%s not built: run `python oracle/refshim/build.py` where /root/reference exists" % path)
        L = ctypes.CDLL(path)
        L.nvdr_ref_last_error.restype = ctypes.c_char_p
        L.nvdr_ref_log_text.restype = ctypes.c_char_p
        L.nvdr_ref_reference_root.restype = ctypes.c_char_p
        L.nvdr_ref_ctx_create.restype = ctypes.c_void_p
        L.nvdr_ref_mip_wrapper_empty.restype = ctypes.c_void_p
        L.nvdr_ref_result_size.restype = ctypes.c_int64
        _libs[variant] = L
    return _libs[variant]


def _desc(a, device=1):
    """numpy array (or None) -> descriptor; keeps the contiguous copy alive through the returned tuple."""
    d = _TensorDesc()
    if a is None:
        d.ndim = -1
        return d, None
    if a.dtype == np.int32 or a.dtype == np.int64 or a.dtype == np.uint32:
        a = np.ascontiguousarray(a, np.int32); d.dtype = 1
    else:
        a = np.ascontiguousarray(a, np.float32); d.dtype = 0
    d.data = a.ctypes.data
    d.device = device
    d.ndim = a.ndim
    for i, s in enumerate(a.shape):
        d.shape[i] = s
    return d, a


class _Call:
    """Collects descriptors (keeping their arrays alive), runs one entry point, unpacks the result list."""
    def __init__(self, L):
        self.L = L
        self.keep = []

    def t(self, a, device=1):
        d, k = _desc(a, device)
        self.keep.append((d, k))
        return ctypes.byref(d)

    def tlist(self, arrs):
        n = len(arrs)
        arr = (_TensorDesc * max(n, 1))()
        for i, a in enumerate(arrs):
            d, k = _desc(a)
            arr[i] = d
            self.keep.append(k)
        self.keep.append(arr)
        return arr, n

    def run(self, fn, *args):
        res = ctypes.c_void_p()
        rc = getattr(self.L, fn)(*args, ctypes.byref(res))
        if rc != 0:
            raise RuntimeError(self.L.nvdr_ref_last_error().decode())
        out = []
        try:
            for i in range(self.L.nvdr_ref_result_count(res)):
                if not self.L.nvdr_ref_result_defined(res, i):
                    out.append(None)
                    continue
                shape = tuple(int(self.L.nvdr_ref_result_size(res, i, d)) for d in range(self.L.nvdr_ref_result_ndim(res, i)))
                a = np.empty(shape, np.int32 if self.L.nvdr_ref_result_dtype(res, i) else np.float32)
                self.L.nvdr_ref_result_copy(res, i, ctypes.c_void_p(a.ctypes.data))
                out.append(a)
        finally:
            self.L.nvdr_ref_result_free(res)
        return out


class RasterizeCRStateWrapper:
    def __init__(self, L):
        self.L = L
        self.h = ctypes.c_void_p(L.nvdr_ref_ctx_create())
        if not self.h:
            raise RuntimeError(L.nvdr_ref_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None):
            self.L.nvdr_ref_ctx_destroy(self.h)
            self.h = None


class TextureMipWrapper:
    def __init__(self, L=None, h=None, mip=None):
        L = L or lib()
        self.L = L
        self.h = ctypes.c_void_p(L.nvdr_ref_mip_wrapper_empty()) if h is None else h
        self.mip = mip

    def __del__(self):
        if getattr(self, "h", None):
            self.L.nvdr_ref_mip_wrapper_free(self.h)
            self.h = None


class TopologyHashWrapper:
    def __init__(self, L, h, ev_hash):
        self.L, self.h, self.ev_hash = L, h, ev_hash

    def __del__(self):
        if getattr(self, "h", None):
            self.L.nvdr_ref_hash_free(self.h)
            self.h = None


class Plugin:
    """``_nvdiffrast_c`` (torch_bindings.cpp:43-71) on numpy arrays, executed by the reference's own code."""

    def __init__(self, variant="fma"):
        self.variant = variant
        self.L = lib(variant)

    # logging passthrough (torch_bindings.cpp:50-51)
    def get_log_level(self):
        return int(self.L.nvdr_ref_get_log_level())

    def set_log_level(self, level):
        self.L.nvdr_ref_set_log_level(int(level))

    def log_text(self):
        return self.L.nvdr_ref_log_text().decode()

    def RasterizeCRStateWrapper(self, device_idx=0):
        return RasterizeCRStateWrapper(self.L)

    def TextureMipWrapper(self):
        return TextureMipWrapper(self.L)

    def rasterize_fwd_cuda(self, state, pos, tri, resolution, ranges, peeling_idx):
        c = _Call(self.L)
        return tuple(c.run("nvdr_ref_rasterize_fwd_cuda", state.h, c.t(pos), c.t(tri), int(resolution[0]), int(resolution[1]),
                           c.t(ranges, device=0), int(peeling_idx)))

    def rasterize_grad(self, pos, tri, out, dy):
        c = _Call(self.L)
        return c.run("nvdr_ref_rasterize_grad", c.t(pos), c.t(tri), c.t(out), c.t(dy))[0]

    def rasterize_grad_db(self, pos, tri, out, dy, ddb):
        c = _Call(self.L)
        return c.run("nvdr_ref_rasterize_grad_db", c.t(pos), c.t(tri), c.t(out), c.t(dy), c.t(ddb))[0]

    def interpolate_fwd(self, attr, rast, tri):
        c = _Call(self.L)
        return tuple(c.run("nvdr_ref_interpolate_fwd", c.t(attr), c.t(rast), c.t(tri)))

    def interpolate_fwd_da(self, attr, rast, tri, rast_db, diff_attrs_all, diff_attrs_list):
        c = _Call(self.L)
        lst = (ctypes.c_int * max(len(diff_attrs_list), 1))(*[int(x) for x in diff_attrs_list])
        return tuple(c.run("nvdr_ref_interpolate_fwd_da", c.t(attr), c.t(rast), c.t(tri), c.t(rast_db),
                           int(bool(diff_attrs_all)), lst, len(diff_attrs_list)))

    def interpolate_grad(self, attr, rast, tri, dy):
        c = _Call(self.L)
        return tuple(c.run("nvdr_ref_interpolate_grad", c.t(attr), c.t(rast), c.t(tri), c.t(dy)))

    def interpolate_grad_da(self, attr, rast, tri, dy, rast_db, dda, diff_attrs_all, diff_attrs_list):
        c = _Call(self.L)
        lst = (ctypes.c_int * max(len(diff_attrs_list), 1))(*[int(x) for x in diff_attrs_list])
        return tuple(c.run("nvdr_ref_interpolate_grad_da", c.t(attr), c.t(rast), c.t(tri), c.t(dy), c.t(rast_db), c.t(dda),
                           int(bool(diff_attrs_all)), lst, len(diff_attrs_list)))

    def texture_construct_mip(self, tex, max_mip_level, cube_mode):
        c = _Call(self.L)
        h = ctypes.c_void_p()
        mip = c.run("nvdr_ref_texture_construct_mip", c.t(tex), int(max_mip_level), int(bool(cube_mode)), ctypes.byref(h))[0]
        return TextureMipWrapper(self.L, h, mip)

    def texture_fwd(self, tex, uv, filter_mode, boundary_mode):
        c = _Call(self.L)
        return c.run("nvdr_ref_texture_fwd", c.t(tex), c.t(uv), int(filter_mode), int(boundary_mode))[0]

    def texture_fwd_mip(self, tex, uv, uv_da, mip_level_bias, mip_wrapper, mip_stack, filter_mode, boundary_mode):
        c = _Call(self.L)
        st, n = c.tlist(mip_stack)
        return c.run("nvdr_ref_texture_fwd_mip", c.t(tex), c.t(uv), c.t(uv_da), c.t(mip_level_bias), mip_wrapper.h, st, n,
                     int(filter_mode), int(boundary_mode))[0]

    def texture_grad_nearest(self, tex, uv, dy, filter_mode, boundary_mode):
        c = _Call(self.L)
        return c.run("nvdr_ref_texture_grad_nearest", c.t(tex), c.t(uv), c.t(dy), int(filter_mode), int(boundary_mode))[0]

    def texture_grad_linear(self, tex, uv, dy, filter_mode, boundary_mode):
        c = _Call(self.L)
        return tuple(c.run("nvdr_ref_texture_grad_linear", c.t(tex), c.t(uv), c.t(dy), int(filter_mode), int(boundary_mode)))

    def texture_grad_linear_mipmap_nearest(self, tex, uv, dy, uv_da, mip_level_bias, mip_wrapper, mip_stack, filter_mode, boundary_mode):
        c = _Call(self.L)
        st, n = c.tlist(mip_stack)
        r = c.run("nvdr_ref_texture_grad_linear_mipmap_nearest", c.t(tex), c.t(uv), c.t(dy), c.t(uv_da), c.t(mip_level_bias),
                  mip_wrapper.h, st, n, int(filter_mode), int(boundary_mode))
        return r[0], r[1], r[2:]

    def texture_grad_linear_mipmap_linear(self, tex, uv, dy, uv_da, mip_level_bias, mip_wrapper, mip_stack, filter_mode, boundary_mode):
        c = _Call(self.L)
        st, n = c.tlist(mip_stack)
        r = c.run("nvdr_ref_texture_grad_linear_mipmap_linear", c.t(tex), c.t(uv), c.t(dy), c.t(uv_da), c.t(mip_level_bias),
                  mip_wrapper.h, st, n, int(filter_mode), int(boundary_mode))
        return r[0], r[1], r[2], r[3], r[4:]

    def antialias_construct_topology_hash(self, tri):
        c = _Call(self.L)
        h = ctypes.c_void_p()
        ev = c.run("nvdr_ref_antialias_construct_topology_hash", c.t(tri), ctypes.byref(h))[0]
        return TopologyHashWrapper(self.L, h, ev)

    def antialias_fwd(self, color, rast, pos, tri, topology_hash):
        c = _Call(self.L)
        return tuple(c.run("nvdr_ref_antialias_fwd", c.t(color), c.t(rast), c.t(pos), c.t(tri), topology_hash.h))

    def antialias_grad(self, color, rast, pos, tri, dy, work_buffer):
        c = _Call(self.L)
        return tuple(c.run("nvdr_ref_antialias_grad", c.t(color), c.t(rast), c.t(pos), c.t(tri), c.t(dy), c.t(work_buffer)))


_plugins = {}


def plugin(variant="fma"):
    if variant not in _plugins:
        _plugins[variant] = Plugin(variant)
    return _plugins[variant]


# --------------------------------------------------------------------------- oracle-style convenience layer
_FILTER = {"nearest": 0, "linear": 1, "linear-mipmap-nearest": 2, "linear-mipmap-linear": 3}
_BOUNDARY = {"cube": 0, "wrap": 1, "clamp": 2, "zero": 3}
_EMPTY_RANGES = np.zeros((0, 2), np.int32)


def rasterize(pos, tri, resolution, ranges=None, variant="fma", ctx=None):
    """-> (rast, rast_db), as nvdiffrast.torch.rasterize (ops.py:93-135 passes an empty [0,2] ranges tensor
    in instanced mode and peeling_idx -1)."""
    P = plugin(variant)
    ctx = ctx or P.RasterizeCRStateWrapper()
    return P.rasterize_fwd_cuda(ctx, pos, tri, resolution, _EMPTY_RANGES if ranges is None else ranges, -1)


def rasterize_surfaces(pos, tri, resolution, ranges=None, variant="fma"):
    """-> (ids u32 [N,Hp,Wp] = triangle id + 1, depth u32 [N,Hp,Wp]): the rasterizer's own integer surfaces
    (what oracle.rasterize_ids restates), read back from the CudaRaster context after a forward call."""
    P = plugin(variant)
    ctx = P.RasterizeCRStateWrapper()
    r, _ = P.rasterize_fwd_cuda(ctx, pos, tri, resolution, _EMPTY_RANGES if ranges is None else ranges, -1)
    N, H, W = r.shape[:3]
    Hp, Wp = (H + 7) & ~7, (W + 7) & ~7
    ids = np.empty((N, Hp, Wp), np.uint32)
    depth = np.empty((N, Hp, Wp), np.uint32)
    P.L.nvdr_ref_ctx_surfaces(ctx.h, ctypes.c_void_p(ids.ctypes.data), ctypes.c_void_p(depth.ctypes.data), ctypes.c_size_t(ids.size))
    return ids, depth


def rasterize_layers(pos, tri, resolution, num_layers, ranges=None, variant="fma"):
    """Depth peeling as DepthPeeler drives it (ops.py:141-204): layer k is a forward call with peeling_idx=k
    on ONE context.  -> list of (rast, rast_db)."""
    P = plugin(variant)
    ctx = P.RasterizeCRStateWrapper()
    rng = _EMPTY_RANGES if ranges is None else ranges
    return [P.rasterize_fwd_cuda(ctx, pos, tri, resolution, rng, k) for k in range(num_layers)]


def rasterize_grad(pos, tri, rast, dy, ddb=None, variant="fma"):
    P = plugin(variant)
    return P.rasterize_grad(pos, tri, rast, dy) if ddb is None else P.rasterize_grad_db(pos, tri, rast, dy, ddb)


def _diff(diff_attrs):
    if diff_attrs is None:
        return False, []
    if isinstance(diff_attrs, str):
        assert diff_attrs == "all"
        return True, []
    return False, [int(x) for x in np.asarray(diff_attrs).reshape(-1)]


def interpolate(attr, rast, tri, rast_db=None, diff_attrs=None, variant="fma"):
    P = plugin(variant)
    da_all, da_list = _diff(diff_attrs)
    if da_all or da_list:
        return P.interpolate_fwd_da(attr, rast, tri, rast_db, da_all, da_list)
    return P.interpolate_fwd(attr, rast, tri)


def interpolate_grad(attr, rast, tri, dy, rast_db=None, dda=None, diff_attrs=None, variant="fma"):
    """-> (g_attr, g_rast, g_rast_db or None)"""
    P = plugin(variant)
    da_all, da_list = _diff(diff_attrs)
    if da_all or da_list:
        return P.interpolate_grad_da(attr, rast, tri, dy, rast_db, dda, da_all, da_list)
    ga, gr = P.interpolate_grad(attr, rast, tri, dy)
    return ga, gr, None


def _texture_setup(P, tex, uv_da, mip_level_bias, mip, filter_mode, boundary_mode, max_mip_level):
    """The Python-side decisions of nvdiffrast.torch.texture (ops.py:395-433)."""
    if filter_mode == "auto":
        filter_mode = "linear-mipmap-linear" if (uv_da is not None or mip_level_bias is not None) else "linear"
    mml = -1 if max_mip_level is None else int(max_mip_level)
    if mml == 0 and "mipmap" in filter_mode:
        filter_mode = "linear"
    wrapper, stack = None, []
    if "mipmap" in filter_mode:
        if mip is not None:
            wrapper, stack = P.TextureMipWrapper(), list(mip)
        else:
            wrapper = P.texture_construct_mip(tex, mml, boundary_mode == "cube")
    return filter_mode, wrapper, stack


def texture_build_mip(tex, max_mip_level=-1, cube=False, variant="fma"):
    """The flat mip tensor (levels 1..L back to back) texture_construct_mip produces."""
    return plugin(variant).texture_construct_mip(tex, max_mip_level, cube).mip


def texture(tex, uv, uv_da=None, mip_level_bias=None, mip=None, filter_mode="auto", boundary_mode="wrap", max_mip_level=None, variant="fma"):
    P = plugin(variant)
    fm, wrapper, stack = _texture_setup(P, tex, uv_da, mip_level_bias, mip, filter_mode, boundary_mode, max_mip_level)
    if "mipmap" in fm:
        return P.texture_fwd_mip(tex, uv, uv_da, mip_level_bias, wrapper, stack, _FILTER[fm], _BOUNDARY[boundary_mode])
    return P.texture_fwd(tex, uv, _FILTER[fm], _BOUNDARY[boundary_mode])


def texture_grad(tex, uv, dy, uv_da=None, mip_level_bias=None, mip=None, filter_mode="auto", boundary_mode="wrap", max_mip_level=None, variant="fma"):
    """-> dict(tex, uv, uv_da, mip_level_bias, mip) like oracle.texture_grad."""
    P = plugin(variant)
    fm, wrapper, stack = _texture_setup(P, tex, uv_da, mip_level_bias, mip, filter_mode, boundary_mode, max_mip_level)
    f, b = _FILTER[fm], _BOUNDARY[boundary_mode]
    out = dict(tex=None, uv=None, uv_da=None, mip_level_bias=None, mip=None)
    if fm == "nearest":
        out["tex"] = P.texture_grad_nearest(tex, uv, dy, f, b)
    elif fm == "linear":
        out["tex"], out["uv"] = P.texture_grad_linear(tex, uv, dy, f, b)
    elif fm == "linear-mipmap-nearest":
        out["tex"], out["uv"], gm = P.texture_grad_linear_mipmap_nearest(tex, uv, dy, uv_da, mip_level_bias, wrapper, stack, f, b)
        out["mip"] = gm if stack else None
    else:
        out["tex"], out["uv"], g_da, g_bias, gm = P.texture_grad_linear_mipmap_linear(tex, uv, dy, uv_da, mip_level_bias, wrapper, stack, f, b)
        out["uv_da"] = g_da if uv_da is not None else None
        out["mip_level_bias"] = g_bias if mip_level_bias is not None else None
        out["mip"] = gm if stack else None
    return out


def antialias(color, rast, pos, tri, variant="fma", return_work=False):
    P = plugin(variant)
    h = P.antialias_construct_topology_hash(tri)
    out, work = P.antialias_fwd(color, rast, pos, tri, h)
    return (out, work) if return_work else out


def antialias_grad(color, rast, pos, tri, dy, variant="fma"):
    """-> (g_color, g_pos)"""
    P = plugin(variant)
    h = P.antialias_construct_topology_hash(tri)
    _out, work = P.antialias_fwd(color, rast, pos, tri, h)
    return P.antialias_grad(color, rast, pos, tri, dy, work)
