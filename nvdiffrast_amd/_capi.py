"""ctypes binding of the C ABI declared in include/nvdr_hip.h.

The HIP library is the ONLY compute backend of this package.  If it is missing or a
call fails this module raises; there is no CPU or eager fallback anywhere.
"""
import ctypes
import os

from . import _build

_lib = None

c_void_p, c_int, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
_i32p = ctypes.POINTER(ctypes.c_int32)
_intp = ctypes.POINTER(ctypes.c_int)
_i64p = ctypes.POINTER(ctypes.c_int64)
_vpp = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes); kept in one table so tests can check it against the header.
ABI_VERSION = 8            # include/nvdr_hip.h; 2: scratch_clean; 3: options + log; 4: caller-chosen clip pool; 5: fused backward, texture_grad scratch; 6: tile flags; 7: work order behind the flags; 8: image pack / unpack
IMAGE_F32, IMAGE_F16, IMAGE_UNORM8 = 0, 1, 2
OPT_LOG_LEVEL, OPT_CUBE_CORNER_FIX, OPT_SCRATCH_LIMIT_MB = 0, 1, 2
c_longlong = ctypes.c_longlong

SIGNATURES = {
    "nvdr_last_error": (ctypes.c_char_p, []),
    "nvdr_abi_version": (c_int, []),
    "nvdr_set_option": (c_int, [c_int, c_int]),
    "nvdr_get_option": (c_int, [c_int]),
    "nvdr_log": (c_int, [c_int, ctypes.c_char_p]),
    "nvdr_profile_enable": (None, [c_int]),
    "nvdr_profile_reset": (None, []),
    "nvdr_profile_read": (c_int, [ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int), c_int]),
    "nvdr_rasterize_scratch_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "nvdr_rasterize_scratch_bytes_pool": (c_size_t, [c_int, c_int, c_int, c_int, c_longlong]),
    "nvdr_rasterize_pool_peak_offset": (c_size_t, [c_int, c_int, c_int, c_int, c_longlong]),
    "nvdr_rasterize_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_longlong, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nvdr_rasterize_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "nvdr_interpolate_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                     c_int, c_int, c_int, c_int, c_int, c_int,
                                     c_int, _i32p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nvdr_interpolate_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                      c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_int, _i32p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nvdr_interpolate_rasterize_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                                c_int, c_int, c_int, c_int, c_int, c_int,
                                                c_void_p, c_void_p, c_int, _i32p, c_int, c_int,
                                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nvdr_texture_mip_info": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, _intp, _intp, _i64p, _i64p]),
    "nvdr_texture_construct_mip": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "nvdr_texture_fwd": (c_int, [c_void_p, _vpp, c_int, c_void_p, c_void_p, c_void_p,
                                 c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "nvdr_texture_grad_scratch_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "nvdr_tile_flags_bytes": (c_size_t, [c_int, c_int, c_int]),
    "nvdr_texture_grad": (c_int, [c_void_p, _vpp, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                  c_void_p, _vpp, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "nvdr_image_packed_bytes": (c_size_t, [c_size_t, c_int]),
    "nvdr_image_pack": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_void_p]),
    "nvdr_image_unpack": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "nvdr_antialias_hash_bytes": (c_size_t, [c_int]),
    "nvdr_antialias_work_bytes": (c_size_t, [c_int, c_int, c_int]),
    "nvdr_antialias_construct_topology_hash": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "nvdr_antialias_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                   c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "nvdr_antialias_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                    c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
}


def ptr_array(tensors):
    """HOST array of device pointers (ctypes void*[]) for a list of torch tensors; (array, n)."""
    n = len(tensors)
    arr = (c_void_p * max(n, 1))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr, n


def lib_path():
    """The in-tree library; NVDR_LIB_PATH (development: A/B runs of two builds on the GPU box) names another one."""
    return os.environ.get("NVDR_LIB_PATH") or _build.LIB_PATH


def load():
    """Load libnvdr_hip.so (never builds implicitly on a GPU box: the .so ships in-tree)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"nvdiffrast_amd: native library {path} is missing. Build it with "
            f"`python -m nvdiffrast_amd._build` (needs hipcc); there is no fallback path.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.nvdr_abi_version() != ABI_VERSION:
        raise RuntimeError(f"nvdiffrast_amd: {path} has ABI version {lib.nvdr_abi_version()}, this package needs "
                           f"{ABI_VERSION}; rebuild it with `python -m nvdiffrast_amd._build`.")
    _lib = _compiled_binding(lib)
    return _lib


class _BoundLib:
    """The library's entry points behind the compiled call layer (csrc_host/nvdr_ffi.c): every name of SIGNATURES whose
    parameters are plain pointers and integers is a `_nvdr_ffi.bind` callable (same arguments as the ctypes function: ints,
    None for NULL, ctypes arrays for host arrays); anything else -- `nvdr_last_error`, symbols outside the table -- is the
    ctypes function.  `ffi` tells which layer is in use."""

    def __init__(self, cdll, bound):
        self.__dict__["_cdll"] = cdll
        self.__dict__["ffi"] = bool(bound)
        self.__dict__.update(bound)

    def __getattr__(self, name):
        return getattr(self.__dict__["_cdll"], name)


def _compiled_binding(cdll):
    """ctypes spends 1.5-5 us per call converting twenty arguments; the compiled layer 0.15 us (the reference's pybind11 module,
    csrc/torch/torch_bindings.cpp:43-71, is compiled too).  NVDR_FFI=0, or a module that has not been built: plain ctypes."""
    if os.environ.get("NVDR_FFI", "1") in ("0", ""):
        return _BoundLib(cdll, {})
    try:
        from . import _nvdr_ffi
    except ImportError:
        return _BoundLib(cdll, {})
    # (host arrays -- the diff-attribute list, the mip pointer arrays -- arrive as ctypes arrays: buffers; entry points that take
    # ctypes.byref() objects, nvdr_texture_mip_info and nvdr_profile_read, stay with ctypes: neither is on a hot path)
    code = {c_void_p: "p", c_int: "i", c_size_t: "n", c_longlong: "L", _i32p: "p", _vpp: "p", ctypes.c_char_p: "p"}
    res = {c_int: "i", c_size_t: "n", None: "v"}
    bound = {}
    for name, (restype, argtypes) in SIGNATURES.items():
        if restype not in res or any(a not in code for a in argtypes) or len(argtypes) > 28:
            continue
        addr = ctypes.cast(getattr(cdll, name), c_void_p).value
        bound[name] = _nvdr_ffi.bind(addr, res[restype] + "".join(code[a] for a in argtypes))
    return _BoundLib(cdll, bound)


_host = None
_host_tried = False
_HOST_SYMBOLS = ("nvdr_last_error", "nvdr_get_option", "nvdr_log", "nvdr_rasterize_scratch_bytes_pool", "nvdr_rasterize_pool_peak_offset",
                 "nvdr_tile_flags_bytes", "nvdr_rasterize_fwd", "nvdr_rasterize_grad", "nvdr_interpolate_fwd", "nvdr_interpolate_grad",
                 "nvdr_interpolate_rasterize_grad", "nvdr_texture_mip_info", "nvdr_texture_construct_mip", "nvdr_texture_fwd",
                 "nvdr_texture_grad", "nvdr_texture_grad_scratch_bytes", "nvdr_antialias_fwd", "nvdr_antialias_grad")


def host():
    """The compiled host layer of rasterize / interpolate (csrc_host/nvdr_torch_host.cpp: validation, allocation, launch and the
    autograd nodes in C++), bound to the entry points of THIS library instance -- or None: NVDR_HOST=0, or the module has not
    been built (the Python host layer, torch/_plugin.py, then serves every call)."""
    global _host, _host_tried
    if _host_tried:
        return _host
    _host_tried = True
    if os.environ.get("NVDR_HOST", "1") in ("0", ""):
        return None
    try:
        import torch  # noqa: F401  (the module links against libtorch)
        from . import _nvdr_host
    except ImportError:
        return None
    cdll = load()._cdll
    _nvdr_host.init({name: ctypes.cast(getattr(cdll, name), c_void_p).value for name in _HOST_SYMBOLS})
    _host = _nvdr_host
    return _host


def check(rc, what):
    if rc != 0:
        msg = load().nvdr_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what}(): {msg}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def profile_read(cap=64):
    lib = load()
    names = (ctypes.c_char_p * cap)()
    total = (ctypes.c_double * cap)()
    cnt = (c_int * cap)()
    n = lib.nvdr_profile_read(names, total, cnt, cap)
    return {names[i].decode(): (total[i], cnt[i]) for i in range(n)}
