"""Build libnvdr_hip.so for gfx950 with hipcc (in-tree, so it travels to the GPU box)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libnvdr_hip.so")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-fhip-fp32-correctly-rounded-divide-sqrt",     # IEEE 1/x and sqrt: the integer raster rules depend on it
    "-Wall", "-Wno-unused-function",
]


# Per-file flags.  The SLP vectorizer packs neighbouring scalar f32 operations into v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32;
# the packed forms need their operands in aligned register pairs, and in these kernels -- long straight-line per-pixel
# arithmetic at 64 registers -- the moves that arrange the pairs cost more than the packing saves (r05, interleaved A/B on
# one box: fused backward with differentials 301 -> 273 us, k_tex_fwd 287 -> 261 us, dense k_fine 148 -> 141 us, fused
# backward 254 -> 244 us).  interpolate.hip and antialias.hip are 2-3 % faster WITH it and keep it.
EXTRA_FLAGS = {
    "raster.hip": ["-fno-slp-vectorize"],
    "backward_fused.hip": ["-fno-slp-vectorize"],
    "texture.hip": ["-fno-slp-vectorize"],
}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    inc = os.path.join(_HERE, "..", "include", "nvdr_hip.h")
    return sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")] + [inc]


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in _deps())


def find_hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found; cannot build libnvdr_hip.so")


def build(force=False, verbose=False):
    """One object per source file, compiled in parallel (each .hip is a self-contained translation unit: no relocatable
    device code), then linked; objects of unchanged sources are reused unless a header changed or `force` is set."""
    if not force and not is_stale():
        _build_host_layers(False, verbose)
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    hipcc = find_hipcc()
    objdir = os.path.join(_HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [d for d in _deps() if not d.endswith(".hip")]
    hdr_time = max(os.path.getmtime(h) for h in headers)
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_time):
            return obj
        cmd = [hipcc] + flags + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources()))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    _build_host_layers(force, verbose)
    return LIB_PATH


def _build_host_layers(force, verbose):
    """The two compiled host layers are optional at run time (_capi falls back to ctypes, ops.py to _plugin): a box without gcc
    or the Python / torch headers still gets the library it has just built.  An explicit build_ffi() / build_host() raises."""
    for step in (build_ffi, build_host):
        try:
            step(force=force, verbose=verbose)
        except Exception as e:                                                  # noqa: BLE001
            import warnings
            warnings.warn("nvdiffrast_amd: %s failed (%s); the package runs without it" % (step.__name__, e))


FFI_SRC = os.path.join(_HERE, "csrc_host", "nvdr_ffi.c")
FFI_PATH = os.path.join(_HERE, "_nvdr_ffi.so")


def build_ffi(force=False, verbose=False):
    """The compiled call layer between Python and the C ABI (csrc_host/nvdr_ffi.c): plain C against the CPython headers, gcc."""
    if not force and os.path.exists(FFI_PATH) and os.path.getmtime(FFI_PATH) > os.path.getmtime(FFI_SRC):
        return FFI_PATH
    import sysconfig
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        raise RuntimeError("gcc not found; cannot build _nvdr_ffi.so (the package falls back to ctypes without it)")
    cmd = [cc, "-O2", "-Wall", "-shared", "-fPIC", "-I" + sysconfig.get_paths()["include"], FFI_SRC, "-o", FFI_PATH]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return FFI_PATH


HOST_SRC = os.path.join(_HERE, "csrc_host", "nvdr_torch_host.cpp")
HOST_PATH = os.path.join(_HERE, "_nvdr_host.so")


def build_host(force=False, verbose=False):
    """The compiled host layer of rasterize / interpolate (csrc_host/nvdr_torch_host.cpp): HOST code only, plain g++ against
    torch's headers -- no device code, nothing goes through torch's hipify build path.  The kernels are reached through the
    C ABI, whose addresses the module receives at run time."""
    deps = [HOST_SRC, os.path.join(_HERE, "..", "include", "nvdr_hip.h")]
    if not force and os.path.exists(HOST_PATH) and os.path.getmtime(HOST_PATH) > max(os.path.getmtime(d) for d in deps):
        return HOST_PATH
    import sysconfig
    import torch
    from torch.utils import cpp_extension
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        raise RuntimeError("g++ not found; cannot build _nvdr_host.so (the package falls back to the Python host layer without it)")
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_nvdr_host", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    cmd += ["-I" + p for p in cpp_extension.include_paths()] + ["-I" + os.path.join(rocm, "include"), "-I" + sysconfig.get_paths()["include"]]
    cmd += [HOST_SRC, "-o", HOST_PATH, "-L" + tlib, "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_python", "-lamdhip64",
            "-Wl,-rpath," + tlib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return HOST_PATH


if __name__ == "__main__":
    build_ffi(force=True, verbose=True)
    build_host(force=True, verbose=True)
    build(force=True, verbose=True)
    print(LIB_PATH)
