"""Register / scratch budgets of the hot kernels, read from the compiler's own metadata (hipcc -S, no GPU needed).

Scratch is ordinary memory behind a write-through L2 on this part: a 4-byte spill executed by every lane of a launch is
megabytes of HBM writes (round 2: 48 B/lane of spills cost k_fine 12 us of 138 and 80 MB of traffic), and
the occupancy each kernel was tuned for needs its VGPR count to stay under the corresponding limit."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nvdiffrast_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _metadata(tmp_path, src):
    out = tmp_path / (src + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fhip-fp32-correctly-rounded-divide-sqrt", "-S", "--cuda-device-only",
           "-I", os.path.join(ROOT, "include"), "-o", str(out), os.path.join(CSRC, src)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text()
    kernels = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", text):
        kernels[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    return kernels


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_rasterizer_kernels_stay_within_their_budgets(tmp_path):
    k = _metadata(tmp_path, "raster.hip")
    fine = k["_ZN4nvdr6k_fineILb0ELb0ELb0ELb0EEEvNS_10FineParamsE"]            # production: no peel, no depth surface, no debug, no sharing
    assert fine == (0, fine[1]) and fine[1] <= 64, fine                         # 8 waves/SIMD, 4 workgroups/CU; not one spilled register
    shared = k["_ZN4nvdr6k_fineILb0ELb0ELb0ELb1EEEvNS_10FineParamsE"]
    assert shared[1] <= 64 and shared[0] <= 16, shared
    grad = k["_ZN4nvdr13k_raster_gradILb0EEEvNS_10GradParamsEii"]
    assert grad[0] == 0 and grad[1] <= 80, grad                                 # 6 waves/SIMD
    setup = k["_ZN4nvdr7k_setupENS_11SetupParamsEi"]
    assert setup[1] <= 102, setup                                               # 5 workgroups/CU (its scratch belongs to the clipper's rare path)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_interpolate_kernels_stay_within_their_budgets(tmp_path):
    k = _metadata(tmp_path, "interpolate.hip")
    for name, (scratch, vgprs) in k.items():
        if "k_interp_grad" in name or "k_interp_fwd" in name:
            assert scratch == 0, (name, scratch)
    g4 = [v for n, v in k.items() if "k_interp_gradILi4ELb0E" in n]
    assert g4 and g4[0][1] <= 64, g4                                            # 8 waves/SIMD, 8 workgroups/CU
