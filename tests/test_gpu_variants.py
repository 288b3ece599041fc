"""The rasterizer has two instantiations of its fine stage: one for launches too small to fill the chip (bins with very
many triangles are shared by several workgroups) and one for large launches.  The test scenes are small, so the rest of
the suite runs the first; here the rasterizer tests run once more with the sharing switched off (a development switch
read once per process, hence the subprocess) so that the other instantiation sees the same scenes."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rasterizer_tests_with_bin_sharing_disabled():
    env = dict(os.environ, NVDR_DEBUG="1048576")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_raster_interp.py"),
                        os.path.join(ROOT, "tests", "test_gpu_edge_cases.py"), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_rasterizer_tests_with_a_triangle_list_for_every_bin():
    """Large meshes get per-bin triangle lists (k_binscan / k_binfill / k_fine<..., LIST>); the development switch makes
    every mesh count as large and every bin with triangles take its list, so the rasterizer's own scenes -- peeling, range
    mode, clipped triangles in the pool, viewport tiling, shared bins, the fuzz seeds -- run through that path as well."""
    env = dict(os.environ, NVDR_DEBUG="268435456")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_raster_interp.py"),
                        os.path.join(ROOT, "tests", "test_gpu_edge_cases.py"), os.path.join(ROOT, "tests", "test_gpu_fuzz.py"),
                        os.path.join(ROOT, "tests", "test_gpu_bin_lists.py"),
                        "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=1800, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_gpu_suite_on_the_python_host_layer():
    """NVDR_HOST=0: rasterize() and interpolate() served by torch/_plugin.py + the Python autograd node instead of the compiled
    host layer (csrc_host/nvdr_torch_host.cpp) -- what a box without the built module runs."""
    env = dict(os.environ, NVDR_HOST="0")
    files = ["test_gpu_raster_interp.py", "test_gpu_end_to_end.py", "test_gpu_edge_cases.py", "test_gpu_fuzz.py", "test_gpu_work_order.py"]
    r = subprocess.run([sys.executable, "-m", "pytest"] + [os.path.join(ROOT, "tests", f) for f in files] +
                       ["-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=1800, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_gpu_suite_with_tile_flag_verification():
    """NVDR_VERIFY_TILE_FLAGS=1: every use of tile flags anywhere in these files re-derives them from the tensor actually passed
    and fails on a mismatch (legality rests on pointer / version / shape; this run checks the claim itself)."""
    env = dict(os.environ, NVDR_VERIFY_TILE_FLAGS="1")
    files = ["test_gpu_tile_flags.py", "test_gpu_raster_interp.py", "test_gpu_texture_aa.py", "test_gpu_fused_backward.py",
             "test_gpu_end_to_end.py", "test_gpu_edge_cases.py", "test_gpu_reference_ops.py",       # (the reference's own call lists)
             "test_gpu_plugin_fused_backward.py"]                                                  # (flags found by the plugin itself, inside autograd)
    r = subprocess.run([sys.executable, "-m", "pytest"] + [os.path.join(ROOT, "tests", f) for f in files] +
                       ["-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=2400, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
