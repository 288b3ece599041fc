"""CPU oracle for the nvdiffrast hot path -- TEST INFRASTRUCTURE ONLY.

numpy front-end over ``libnvdr_oracle.so`` (C restatement of the reference's
algorithms, see ``nvdr_oracle.h`` for the file:line citations).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product path (``nvdiffrast_amd``) never does.

Parity status: PINNED.  Tests use this package through ``oracle.pinned.PinnedOracle``, which runs every call
also through the reference itself (``oracle/_ref``: the reference's own sources compiled for the host, see
``oracle/ref.py``) and requires agreement; ``docs/img/tri.png`` and ``tests/golden/reference_pipeline.npz``
(vectors produced by the reference) are reproduced as well.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnvdr_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u32p = ctypes.POINTER(ctypes.c_uint32)


def build(force=False):
    """Compile the C oracle with gcc (a few seconds)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if not force and os.path.exists(_LIB_PATH):
        if os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(s) for s in srcs):
            return _LIB_PATH
    subprocess.check_call(["make", "-s", "-B", "-C", _HERE])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def num_threads():
    return int(lib().nvdro_num_threads())


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _pad8(x):
    return (x + 7) & ~7


# --------------------------------------------------------------------------- rasterize

def _raster_args(pos, tri, ranges):
    pos = _f32(pos)
    tri = _i32(tri)
    instance = pos.ndim == 3
    if instance:
        N, V = pos.shape[0], pos.shape[1]
        rng = None
    else:
        assert ranges is not None, "range mode needs ranges"
        rng = _i32(ranges)
        N, V = rng.shape[0], pos.shape[0]
    return pos, tri, rng, instance, N, V, tri.shape[0]


def rasterize(pos, tri, resolution, ranges=None, peel_depth=None, return_depth=False):
    """-> (rast [N,H,W,4], rast_db [N,H,W,4][, depth u32 [N,Hp,Wp]]).

    ``peel_depth``: previous layer's depth surface (enables the peel test)."""
    pos, tri, rng, inst, N, V, T = _raster_args(pos, tri, ranges)
    H, W = int(resolution[0]), int(resolution[1])
    out = np.empty((N, H, W, 4), np.float32)
    out_db = np.empty((N, H, W, 4), np.float32)
    depth = np.empty((N, _pad8(H), _pad8(W)), np.uint32)
    peel = None if peel_depth is None else np.ascontiguousarray(peel_depth, np.uint32)
    rc = lib().nvdro_rasterize_fwd(_p(pos, _f32p), _p(tri, _i32p), _p(rng, _i32p), int(inst),
                                   N, V, T, H, W, int(peel is not None), _p(peel, _u32p),
                                   _p(depth, _u32p), _p(out, _f32p), _p(out_db, _f32p))
    assert rc == 0
    return (out, out_db, depth) if return_depth else (out, out_db)


def rasterize_ids(pos, tri, resolution, ranges=None, peel_depth=None):
    """Integer stage only -> (ids u32 [N,Hp,Wp], depth u32 [N,Hp,Wp])."""
    pos, tri, rng, inst, N, V, T = _raster_args(pos, tri, ranges)
    H, W = int(resolution[0]), int(resolution[1])
    ids = np.empty((N, _pad8(H), _pad8(W)), np.uint32)
    depth = np.empty((N, _pad8(H), _pad8(W)), np.uint32)
    peel = None if peel_depth is None else np.ascontiguousarray(peel_depth, np.uint32)
    rc = lib().nvdro_rasterize_ids(_p(pos, _f32p), _p(tri, _i32p), _p(rng, _i32p), int(inst),
                                   N, V, T, H, W, int(peel is not None), _p(peel, _u32p),
                                   _p(depth, _u32p), _p(ids, _u32p))
    assert rc == 0
    return ids, depth


def rasterize_grad(pos, tri, rast, dy, ddb=None):
    pos = _f32(pos); tri = _i32(tri); rast = _f32(rast); dy = _f32(dy)
    ddb = None if ddb is None else _f32(ddb)
    inst = pos.ndim == 3
    N, H, W = rast.shape[0], rast.shape[1], rast.shape[2]
    V = pos.shape[1] if inst else pos.shape[0]
    g = np.empty_like(pos)
    rc = lib().nvdro_rasterize_grad(_p(pos, _f32p), _p(tri, _i32p), _p(rast, _f32p), _p(dy, _f32p),
                                    _p(ddb, _f32p), int(inst), N, V, tri.shape[0], H, W, _p(g, _f32p))
    assert rc == 0
    return g


# --------------------------------------------------------------------------- interpolate

def _diff_args(diff_attrs, A):
    if diff_attrs is None or (not isinstance(diff_attrs, str) and len(diff_attrs) == 0):
        return 0, None, 0
    if isinstance(diff_attrs, str):
        assert diff_attrs == "all"
        return 1, None, A
    lst = _i32(np.asarray(diff_attrs).reshape(-1))
    return 0, lst, int(lst.shape[0])


def interpolate(attr, rast, tri, rast_db=None, diff_attrs=None):
    attr = _f32(attr); rast = _f32(rast); tri = _i32(tri)
    inst = attr.ndim == 3
    Nattr = attr.shape[0] if inst else 1
    V, A = attr.shape[-2], attr.shape[-1]
    N, H, W = rast.shape[:3]
    diff_all, lst, D = _diff_args(diff_attrs, A)
    rdb = _f32(rast_db) if D > 0 else None
    out = np.empty((N, H, W, A), np.float32)
    out_da = np.empty((N, H, W, 2 * D), np.float32)
    rc = lib().nvdro_interpolate_fwd(_p(attr, _f32p), _p(rast, _f32p), _p(tri, _i32p), _p(rdb, _f32p),
                                     int(inst), Nattr, N, V, A, tri.shape[0], H, W,
                                     diff_all, _p(lst, _i32p), 0 if lst is None else int(lst.shape[0]),
                                     _p(out, _f32p), _p(out_da, _f32p) if D > 0 else None)
    assert rc == 0
    return out, out_da


def interpolate_grad(attr, rast, tri, dy, rast_db=None, dda=None, diff_attrs=None):
    """-> (g_attr, g_rast, g_rast_db or None)"""
    attr = _f32(attr); rast = _f32(rast); tri = _i32(tri); dy = _f32(dy)
    inst = attr.ndim == 3
    Nattr = attr.shape[0] if inst else 1
    V, A = attr.shape[-2], attr.shape[-1]
    N, H, W = rast.shape[:3]
    diff_all, lst, D = _diff_args(diff_attrs, A)
    rdb = _f32(rast_db) if D > 0 else None
    dda_ = _f32(dda) if D > 0 else None
    g_attr = np.empty_like(attr)
    g_rast = np.empty_like(rast)
    g_rdb = np.empty_like(rast) if D > 0 else None
    rc = lib().nvdro_interpolate_grad(_p(attr, _f32p), _p(rast, _f32p), _p(tri, _i32p), _p(dy, _f32p),
                                      _p(rdb, _f32p), _p(dda_, _f32p), int(inst), Nattr,
                                      N, V, A, tri.shape[0], H, W,
                                      diff_all, _p(lst, _i32p), 0 if lst is None else int(lst.shape[0]),
                                      _p(g_attr, _f32p), _p(g_rast, _f32p), _p(g_rdb, _f32p))
    assert rc == 0
    return g_attr, g_rast, g_rdb


# --------------------------------------------------------------------------- texture

_FILTER = {"nearest": 0, "linear": 1, "linear-mipmap-nearest": 2, "linear-mipmap-linear": 3}
_BOUNDARY = {"cube": 0, "wrap": 1, "clamp": 2, "zero": 3}
_i64p = ctypes.POINTER(ctypes.c_int64)


def set_cube_corner_fix(on):
    """False (default) = the reference's cube-corner behaviour, lost corner flag for slices >= 1 included."""
    lib().nvdro_set_cube_corner_fix(int(bool(on)))


def _tex_dims(tex_shape):
    """(n, h, w, c, cube) of a [n,h,w,c] texture or a [n,6,s,s,c] cube map."""
    if len(tex_shape) == 5:
        n, six, h, w, c = [int(x) for x in tex_shape]
        assert six == 6 and h == w
        return n, h, w, c, 1
    n, h, w, c = [int(x) for x in tex_shape]
    return n, h, w, c, 0


def texture_mip_info(tex_shape, max_mip_level=-1):
    """-> (L, widths, heights, offsets_in_floats, total_floats); raises on odd extents (texture.cpp:85-86)."""
    n, h, w, c, cube = _tex_dims(tex_shape)
    lw = (ctypes.c_int * 17)(); lh = (ctypes.c_int * 17)(); off = (ctypes.c_int64 * 17)()
    total = ctypes.c_int64(0)
    L = lib().nvdro_texture_mip_info(n, h, w, c, cube, int(max_mip_level), lw, lh, off, ctypes.byref(total))
    if L < 0:
        raise ValueError("texture extents must be divisible by two at every mip level")
    return L, list(lw[:L + 1]), list(lh[:L + 1]), list(off[:L + 1]), int(total.value)


def texture_build_mip(tex, max_mip_level=-1):
    """2x2 box mip chain -> list of arrays for levels 1..L, each [n,h,w,c] (views of one flat buffer)."""
    tex = _f32(tex)
    n, h, w, c, cube = _tex_dims(tex.shape)
    L, lw, lh, off, total = texture_mip_info(tex.shape, max_mip_level)
    flat = np.zeros(max(total, 1), np.float32)
    rc = lib().nvdro_texture_build_mip(_p(tex, _f32p), n, h, w, c, cube, int(max_mip_level), _p(flat, _f32p))
    assert rc == 0
    f = 6 if cube else 1
    shp = (lambda i: (n, 6, lh[i], lw[i], c)) if cube else (lambda i: (n, lh[i], lw[i], c))
    return [flat[off[i]:off[i] + n * f * lh[i] * lw[i] * c].reshape(shp(i)) for i in range(1, L + 1)]


def _ptr_array(arrs):
    n = max(len(arrs), 1)
    pa = (_f32p * n)()
    for i, a in enumerate(arrs):
        pa[i] = a.ctypes.data_as(_f32p)
    return pa


def _texture_setup(tex, uv, uv_da, mip_level_bias, mip, filter_mode, boundary_mode, max_mip_level):
    tex = _f32(tex); uv = _f32(uv)
    if filter_mode == "auto":
        filter_mode = "linear-mipmap-linear" if (uv_da is not None or mip_level_bias is not None) else "linear"
    mml = -1 if max_mip_level is None else int(max_mip_level)
    if mml == 0 and "mipmap" in filter_mode:
        filter_mode = "linear"
    levels = []
    if "mipmap" in filter_mode:
        assert uv_da is not None or mip_level_bias is not None
        levels = [_f32(m) for m in mip] if mip is not None else texture_build_mip(tex, mml)
    else:
        uv_da = mip_level_bias = None
    uv_da = None if uv_da is None else _f32(uv_da)
    mip_level_bias = None if mip_level_bias is None else _f32(mip_level_bias)
    return tex, uv, uv_da, mip_level_bias, levels, filter_mode, _FILTER[filter_mode], _BOUNDARY[boundary_mode]


def texture(tex, uv, uv_da=None, mip_level_bias=None, mip=None, filter_mode="auto", boundary_mode="wrap", max_mip_level=None):
    """Arguments as nvdiffrast.torch.texture (2D textures); ``mip`` = optional list of level arrays."""
    tex, uv, uv_da, bias, levels, _, f, b = _texture_setup(tex, uv, uv_da, mip_level_bias, mip, filter_mode, boundary_mode, max_mip_level)
    N, H, W = uv.shape[:3]
    tn, th, tw, tc, cube = _tex_dims(tex.shape)
    assert cube == (b == 0), "cube map textures need boundary_mode='cube' (and vice versa)"
    out = np.empty((N, H, W, tc), np.float32)
    pa = _ptr_array(levels)
    rc = lib().nvdro_texture_fwd(_p(tex, _f32p), pa, len(levels), _p(uv, _f32p), _p(uv_da, _f32p), _p(bias, _f32p),
                                 tn, th, tw, tc, N, H, W, f, b, _p(out, _f32p))
    assert rc == 0, rc
    return out


def texture_grad(tex, uv, dy, uv_da=None, mip_level_bias=None, mip=None, filter_mode="auto", boundary_mode="wrap", max_mip_level=None):
    """-> dict(tex, uv, uv_da, mip_level_bias, mip).  With ``mip`` given (custom stack) the levels get their
    own gradients (dict['mip']); otherwise mip gradients are folded into dict['tex'] like MipGradKernel."""
    tex, uv, uv_da, bias, levels, fname, f, b = _texture_setup(tex, uv, uv_da, mip_level_bias, mip, filter_mode, boundary_mode, max_mip_level)
    dy = _f32(dy)
    N, H, W = uv.shape[:3]
    custom = mip is not None and len(levels) > 0
    g_tex = np.zeros_like(tex)
    g_levels = [np.zeros_like(l) for l in levels]
    g_uv = np.zeros_like(uv) if f != 0 else None
    g_uv_da = np.zeros_like(uv_da) if (f == 3 and uv_da is not None) else None
    g_bias = np.zeros_like(bias) if (f == 3 and bias is not None) else None
    pa = _ptr_array(levels); ga = _ptr_array(g_levels)
    tn, th, tw, tc, cube = _tex_dims(tex.shape)
    assert cube == (b == 0), "cube map textures need boundary_mode='cube' (and vice versa)"
    rc = lib().nvdro_texture_grad(_p(tex, _f32p), pa, len(levels), _p(uv, _f32p), _p(uv_da, _f32p), _p(bias, _f32p),
                                  _p(dy, _f32p), tn, th, tw, tc, N, H, W, f, b,
                                  int(not custom), _p(g_tex, _f32p), ga, _p(g_uv, _f32p), _p(g_uv_da, _f32p), _p(g_bias, _f32p))
    assert rc == 0, rc
    return dict(tex=g_tex, uv=g_uv, uv_da=g_uv_da, mip_level_bias=g_bias, mip=g_levels if custom else None)


# --------------------------------------------------------------------------- antialias

def antialias(color, rast, pos, tri):
    color = _f32(color); rast = _f32(rast); pos = _f32(pos); tri = _i32(tri)
    inst = pos.ndim == 3
    N, H, W, C = color.shape
    V = pos.shape[1] if inst else pos.shape[0]
    out = np.empty_like(color)
    rc = lib().nvdro_antialias_fwd(_p(color, _f32p), _p(rast, _f32p), _p(pos, _f32p), _p(tri, _i32p), int(inst),
                                   N, V, tri.shape[0], H, W, C, _p(out, _f32p))
    assert rc == 0, rc
    return out


def antialias_grad(color, rast, pos, tri, dy):
    """-> (g_color, g_pos)"""
    color = _f32(color); rast = _f32(rast); pos = _f32(pos); tri = _i32(tri); dy = _f32(dy)
    inst = pos.ndim == 3
    N, H, W, C = color.shape
    V = pos.shape[1] if inst else pos.shape[0]
    g_color = np.empty_like(color)
    g_pos = np.empty_like(pos)
    rc = lib().nvdro_antialias_grad(_p(color, _f32p), _p(rast, _f32p), _p(pos, _f32p), _p(tri, _i32p), _p(dy, _f32p),
                                    int(inst), N, V, tri.shape[0], H, W, C, _p(g_color, _f32p), _p(g_pos, _f32p))
    assert rc == 0, rc
    return g_color, g_pos
