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


_ASM = {}


def _assembly(tmp_path, src):
    if src not in _ASM:
        _metadata(tmp_path, src)
    return _ASM[src]


def _kernel_body(text, mangled):
    i = text.index(mangled + ":")
    return text[i:text.index("s_endpgm", i)].split("\n")


def _metadata(tmp_path, src):
    out = tmp_path / (src + ".s")
    from nvdiffrast_amd import _build
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fhip-fp32-correctly-rounded-divide-sqrt"] + _build.EXTRA_FLAGS.get(src, []) + \
          ["-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"), "-o", str(out), os.path.join(CSRC, src)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text()
    _ASM[src] = text
    kernels = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", text):
        kernels[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    return kernels


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_rasterizer_kernels_stay_within_their_budgets(tmp_path):
    k = _metadata(tmp_path, "raster.hip")
    fine = k["_ZN4nvdr6k_fineILb0ELb0ELb0ELb0ELb0ELb0EEEvNS_10FineParamsE"]       # production: no peel, no depth surface, no debug, no sharing, no lists
    # 8 waves/SIMD, 4 workgroups/CU, no scratch (r04 parked the packed thread id and two hoisted values: 12 B/lane; r05 takes the
    # thread id apart at the kernel's first instruction, replaces __syncthreads_and -- which needs the flat id -- by one LDS word,
    # and scans with DPP moves instead of __shfl_up, whose per-lane source addresses were hoisted out of the pass loop)
    assert fine[0] == 0 and fine[1] <= 64, fine
    shared = k["_ZN4nvdr6k_fineILb0ELb0ELb0ELb1ELb0ELb0EEEvNS_10FineParamsE"]
    assert shared[1] <= 64 and shared[0] == 0, shared
    for name in ("_ZN4nvdr6k_fineILb0ELb0ELb0ELb0ELb1ELb0EEEvNS_10FineParamsE", "_ZN4nvdr6k_fineILb0ELb0ELb0ELb1ELb1ELb0EEEvNS_10FineParamsE",
                 "_ZN4nvdr6k_fineILb0ELb0ELb0ELb1ELb1ELb1EEEvNS_10FineParamsE"):
        lists = k[name]                                                         # large meshes: bins with triangle lists
        assert lists[1] <= 64 and lists[0] == 0, (name, lists)
    for name, v in k.items():                                                   # depth peeling / depth surface: every instantiation at 8 waves/SIMD
        if re.search(r"k_fineILb[01]ELb[01]ELb0E", name):                         # (all but the debug instantiations)
            assert v[1] <= 64 and v[0] <= 16, (name, v)
    grad = k["_ZN4nvdr13k_raster_gradILb0ELb0EEEvNS_10GradParamsEii"]
    assert grad[0] == 0 and grad[1] <= 80, grad                                 # 6 waves/SIMD
    db_only = k["_ZN4nvdr13k_raster_gradILb1ELb1EEEvNS_10GradParamsEii"]          # rast_db's share alone (dy == NULL): the plugin-level fused backward
    assert db_only[0] == 0 and db_only[1] <= 128, db_only
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


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_fused_backward_and_texture_gradient_kernels_stay_within_their_budgets(tmp_path):
    """The fused backward without pixel differentials runs at 8 waves/SIMD (64 VGPRs) without a spilled register in every
    instantiation (the ordered decode, the early exit of empty blocks and the opaque parameter reads were all added under
    that constraint); with differentials 6 waves/SIMD and the spills DESIGN.md 4.4 states.  The texture gradient's light
    kernel keeps 8 workgroups per CU, the heavy one its 96 registers with no more than 20 B/lane of scratch."""
    k = _metadata(tmp_path, "backward_fused.hip")
    plain = {n: v for n, v in k.items() if "k_interp_raster_grad" in n and n.endswith("ELb0EEEvNS_11FusedParamsEiii")}
    da = {n: v for n, v in k.items() if "k_interp_raster_grad" in n and n.endswith("ELb1EEEvNS_11FusedParamsEiii")}
    assert len(plain) == 6 and len(da) == 6, sorted(k)
    for n, (scratch, vgprs) in plain.items():
        assert scratch == 0 and vgprs <= 64, (n, scratch, vgprs)
    for n, (scratch, vgprs) in da.items():
        assert scratch <= 56 and vgprs <= 80, (n, scratch, vgprs)
    t = _metadata(tmp_path, "texture.hip")
    light = {n: v for n, v in t.items() if "k_tex_grad_light" in n}
    assert light and all(s == 0 and v <= 64 for s, v in light.values()), light
    heavy = t["_ZN4nvdr10k_tex_gradILi3ELb0ELb0ELi3EEEvNS_9TexParamsEi"]          # the general kernel: trilinear, 2-D, three channels
    assert heavy[0] <= 20 and heavy[1] <= 96, heavy
    lean = {n: v for n, v in t.items() if "k_tex_grad_lean" in n}                  # config 3's heavy blocks since r05: no scratch, 6+ waves/SIMD
    assert len(lean) == 8 and all(s == 0 and v <= 80 for s, v in lean.values()), lean
    assert t["_ZN4nvdr15k_tex_grad_leanILi3ELi3EEEvNS_9TexParamsEi"][1] <= 72, lean


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_shared_bin_handoff_waits_for_its_exchanges_before_the_barrier(tmp_path):
    """a part of a shared bin publishes its keys with returning exchanges and may count itself in only when
    the old values have come back in EVERY wave.  The compiler had sunk that wait below the workgroup barrier (the
    values are consumed after it).  Every k_fine<..., SPLIT> instantiation must have `s_waitcnt vmcnt(0)` between its
    last global_atomic_swap and the barrier that follows, and the arrival counter must be a release/acquire pair."""
    text = _assembly(tmp_path, "raster.hip")
    names = re.findall(r"^(_ZN4nvdr6k_fineILb[01]ELb[01]ELb0ELb1ELb0ELb0EEEvNS_10FineParamsE):", text, flags=re.M)
    assert len(names) == 4, names
    for name in names:
        body = _kernel_body(text, name)
        swaps = [i for i, l in enumerate(body) if "global_atomic_swap_x2" in l]
        assert len(swaps) == 8, (name, len(swaps))
        barrier = next(i for i in range(swaps[-1], len(body)) if "s_barrier" in body[i])
        between = [l.strip() for l in body[swaps[-1] + 1:barrier]]
        assert "s_waitcnt vmcnt(0)" in between, (name, between)
        # the arrival counter: a release / acquire pair at agent scope (write-back, atomic, invalidate) -- the default among the
        # forms the timing switches select (raster.hip)
        windows = [" ".join(l.strip() for l in body[i:i + 5]) for i in range(barrier, len(body)) if "buffer_wbl2" in body[i]]
        assert any("global_atomic_add" in w and "buffer_inv" in w for w in windows), (name, windows)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_wave_uniform_texel_loads_are_scalar_loads_the_compiler_tracks(tmp_path):
    """the uniform-tile sample of k_tex_fwd and the flags / constant quad of k_tex_grad_light_w were fetched by
    hand-written `s_load_dword` asm statements with the `s_waitcnt` in a SEPARATE statement -- nothing kept the register
    allocator from touching the destination SGPRs in between (SMEM returns are not interlocked).  They are now ordinary
    loads through a constant-address-space pointer (nvdr_device.hpp scalar_load): the compiler selects the scalar loads and
    places the waits itself.  No source may contain an s_load asm statement, and the two kernels must still fetch those
    values through the scalar cache (s_load from a base that is not the kernarg pointer s[0:1])."""
    for f in os.listdir(CSRC):
        text = open(os.path.join(CSRC, f)).read()
        assert not re.search(r'asm[^;]*"[^"]*s_load', text), f
        assert not re.search(r'asm[^;]*"[^"]*s_waitcnt lgkmcnt', text), f
    text = _assembly(tmp_path, "texture.hip")
    for name, least in (("_ZN4nvdr9k_tex_fwdILi3ELb0ELi3EEEvNS_9TexParamsE", 8), ("_ZN4nvdr18k_tex_grad_light_wILi3ELi3EEEvNS_9TexParamsEii", 8)):
        i = text.index("\n" + name + ":")
        body = text[i:text.index(".Lfunc_end", i)]
        loads = [l for l in body.split("\n") if re.search(r"s_load_dword(x[234])?\s", l) and "s[0:1]" not in l]
        assert len(loads) >= least, (name, len(loads))
