#!/usr/bin/env python3
"""Recipe for oracle/_ref: compiles the reference's OWN sources, where they lie under /root/reference, for
the host CPU and links them with the CUDA-on-CPU shim of this directory.  TEST INFRASTRUCTURE ONLY.

What is compiled, unmodified, straight from the reference checkout:
  csrc/common/{common,texture}.cpp, csrc/common/{rasterize,interpolate,texture_kernel,antialias}.cu,
  csrc/common/cudaraster/impl/{Buffer,CudaRaster,RasterImpl}.cpp, RasterImpl_kernel.cu (+ the four .inl stages),
  csrc/torch/torch_{rasterize,interpolate,texture,antialias}.cpp (the glue, against include/torch/extension.h).
The only source that cannot be fed to g++ as it is is cudaraster/impl/Util.inl, whose lines 23-79 are PTX inline
asm: a temporary copy (deleted after the build, never written into the repo) has each
`asm("<ptx>" : "=c"(out) : "c"(in), ...)` statement rewritten, by the regular expression below, into
`out = ptx_<sanitised ptx>(in, ...)`; include/nvdr_ptx_emu.h holds those functions.  Nothing else is touched.

Two libraries are produced, differing only in floating-point contraction:
  oracle/_ref/libnvdr_ref.so        -mfma -ffp-contract=fast   (a*b+c fused where the compiler sees it: nvcc's default -fmad=true)
  oracle/_ref/libnvdr_ref_nofma.so  -ffp-contract=off          (no fusion)
nvcc's and gcc's choice of WHICH products to fuse need not agree, so tests treat the pair as the bracket
of the reference's result where a contraction decides an integer outcome.

Usage: python oracle/refshim/build.py [--reference /root/reference] [--force]
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(os.path.dirname(HERE), "_ref")
DEFAULT_REF = "/root/reference"

HOST_SOURCES = [
    "csrc/common/common.cpp",
    "csrc/common/texture.cpp",
    "csrc/common/cudaraster/impl/Buffer.cpp",
    "csrc/common/cudaraster/impl/CudaRaster.cpp",
    "csrc/common/cudaraster/impl/RasterImpl.cpp",
    "csrc/torch/torch_rasterize.cpp",
    "csrc/torch/torch_interpolate.cpp",
    "csrc/torch/torch_texture.cpp",
    "csrc/torch/torch_antialias.cpp",
]
DEVICE_SOURCES = [
    "csrc/common/rasterize.cu",
    "csrc/common/interpolate.cu",
    "csrc/common/texture_kernel.cu",
    "csrc/common/antialias.cu",
]
CUDARASTER_KERNEL = "csrc/common/cudaraster/impl/RasterImpl_kernel.cu"
UTIL_INL = "csrc/common/cudaraster/impl/Util.inl"
SHIM_SOURCES = ["shim_runtime.cpp", "ref_capi.cpp"]

ASM_RE = re.compile(r'asm\(\s*"([^"]*)"\s*:\s*"=([rlfd])"\((\w+)\)\s*(?::\s*([^;]*?))?\)\s*;')
# nvcc pre-includes cuda_runtime.h into every .cu translation unit and defines __CUDACC__.
DEVICE_FLAGS = ["-x", "c++", "-D__CUDACC__", "-include", "cuda_runtime.h"]
CAST = {"r": "uint32_t", "l": "int64_t", "f": "float", "d": "double"}


def patch_util_inl(text):
    """asm("<ptx>" : "=c"(out) : "c"(a), "c"(b)) -> out = (decltype(out))ptx_<name>((T)(a), (T)(b));"""
    def sub(m):
        ptx, _oc, out, ins = m.groups()
        name = "ptx_" + re.sub(r"[^A-Za-z0-9]", "_", ptx.strip())
        args = ", ".join("(%s)(%s)" % (CAST[c], v) for c, v in re.findall(r'"([rlfd])"\((\w+)\)', ins or ""))
        return "%s = (decltype(%s))%s(%s);" % (out, out, name, args)
    patched, n = ASM_RE.subn(sub, text)
    if "asm(" in patched:
        raise RuntimeError("an asm statement of Util.inl was not recognised")
    # Second rewrite, same temporary copy: the three C casts `(U32)values.{x,y,z}` of setupPleq (Util.inl:188-190) convert
    # a float depth to an unsigned integer.  CUDA compiles them to cvt.rzi.u32.f32, which SATURATES (a vertex depth
    # that float rounding pushed just above 2^32 -- clipped vertices on the far plane -- becomes 0xFFFFFFFF); in C++
    # the same cast is undefined out of range and x86 wraps it.  nvdr_ptx_emu.h supplies the CUDA behaviour.
    patched, m = re.subn(r"\(U32\)values\.([xyz])", r"nvdr_cuda_cvt_rzi_u32_f32(values.\1)", patched)
    if m != 3:
        raise RuntimeError("expected the three float->U32 casts of setupPleq, found %d" % m)
    return '#include "nvdr_ptx_emu.h"\n' + patched, n


def newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(reference=DEFAULT_REF, force=False, verbose=True):
    """Returns the list of built libraries; raises if the reference checkout is missing."""
    if not os.path.isdir(os.path.join(reference, "csrc")):
        raise FileNotFoundError("reference checkout not found at %s" % reference)
    os.makedirs(OUT_DIR, exist_ok=True)
    ref_files = [os.path.join(reference, s) for s in HOST_SOURCES + DEVICE_SOURCES + [CUDARASTER_KERNEL, UTIL_INL]]
    own_files = [os.path.join(HERE, s) for s in SHIM_SOURCES + ["cover_probe.cpp"]] + [os.path.abspath(__file__)]
    for root, _d, files in os.walk(os.path.join(HERE, "include")):
        own_files += [os.path.join(root, f) for f in files]
    variants = [("libnvdr_ref.so", ["-mfma", "-ffp-contract=fast"]),
                ("libnvdr_ref_nofma.so", ["-ffp-contract=off"])]
    outs = [os.path.join(OUT_DIR, v[0]) for v in variants]
    if not force and all(os.path.exists(o) for o in outs) and min(os.path.getmtime(o) for o in outs) >= newest(ref_files + own_files):
        return outs

    tmp = tempfile.mkdtemp(prefix="nvdr_ref_build_")
    try:
        # Temporary, patched copy of Util.inl beside an unmodified copy of the 15-line kernel file that includes it.
        patched, n = patch_util_inl(open(os.path.join(reference, UTIL_INL)).read())
        gen = os.path.join(tmp, "gen")
        os.makedirs(gen)
        open(os.path.join(gen, "Util.inl"), "w").write(patched)
        shutil.copy(os.path.join(reference, CUDARASTER_KERNEL), os.path.join(gen, "RasterImpl_kernel.cu"))
        if verbose:
            print("refshim: %d PTX asm statements of Util.inl routed to nvdr_ptx_emu.h" % n)

        inc = ["-I", os.path.join(HERE, "include"),
               "-I", os.path.join(reference, "csrc/common"),
               "-I", os.path.join(reference, "csrc/common/cudaraster/impl"),
               "-I", os.path.join(reference, "csrc/torch")]
        base = ["g++", "-std=c++17", "-O2", "-fPIC", "-fno-fast-math", "-fno-strict-aliasing", "-fno-extern-tls-init", "-g1", "-w", "-DNVDR_TORCH",
                "-DNVDR_REFERENCE_ROOT=\"%s\"" % reference]

        def compile_one(job):
            src, obj, extra = job
            cmd = base + extra + inc + ["-c", src, "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("refshim: compile failed: %s\n%s" % (" ".join(cmd), r.stderr[-6000:]))
            return obj

        for lib_name, fp_flags in variants:
            odir = os.path.join(tmp, lib_name + ".o")
            os.makedirs(odir)
            jobs = []
            for s in HOST_SOURCES:
                jobs.append((os.path.join(reference, s), os.path.join(odir, s.replace("/", "_") + ".o"), fp_flags))
            for s in DEVICE_SOURCES:
                jobs.append((os.path.join(reference, s), os.path.join(odir, s.replace("/", "_") + ".o"), fp_flags + DEVICE_FLAGS))
            jobs.append((os.path.join(gen, "RasterImpl_kernel.cu"), os.path.join(odir, "RasterImpl_kernel.o"), fp_flags + DEVICE_FLAGS))
            for s in SHIM_SOURCES:
                jobs.append((os.path.join(HERE, s), os.path.join(odir, s + ".o"), fp_flags))
            # probe into Util.inl's coverage functions: device code, resolves "Util.inl" to the patched copy
            jobs.append((os.path.join(HERE, "cover_probe.cpp"), os.path.join(odir, "cover_probe.o"), fp_flags + DEVICE_FLAGS + ["-I", gen]))
            with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
                objs = list(ex.map(compile_one, jobs))
            out = os.path.join(OUT_DIR, lib_name)
            r = subprocess.run(["g++", "-shared", "-o", out] + objs + ["-lm"], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("refshim: link failed\n" + r.stderr[-4000:])
            if verbose:
                print("refshim: built", out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return outs


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=DEFAULT_REF)
    ap.add_argument("--force", action="store_true")
    a = ap.parse_args()
    try:
        build(a.reference, a.force)
    except Exception as e:  # noqa: BLE001
        print(e, file=sys.stderr)
        sys.exit(1)
