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
