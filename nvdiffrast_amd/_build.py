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
    if not force and not is_stale():
        return LIB_PATH
    cmd = [find_hipcc()] + HIPCC_FLAGS + sources() + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    build(force=True, verbose=True)
    print(LIB_PATH)
