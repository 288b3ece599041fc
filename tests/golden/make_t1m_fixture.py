"""Pins the million-triangle path to the reference itself (VERDICT r5 "missing" 4).

Runs the reference's own rasterizer -- oracle/_ref: its CudaRaster sources compiled for the host by oracle/refshim/build.py --
ONCE over item 0 of bench.py's `t1m` and `t1m_shuffled` scenes (1 M triangles, 1024^2: per-bin triangle lists, SHADE launches)
and over one S10k stress item at 1024^2, and stores what a test on the GPU box can compare against without the reference:

    ids_sha256        SHA-256 of the triangle-id image (float32 channel 3 of rast as little-endian uint32 ids, row major)
    covered           number of covered pixels
    sample_yx, sample_rast   256 pixels (seeded choice among the covered ones) with the reference's (u, v, z/w, id)

Minutes of CPU per scene (the emulation runs CUDA threads as fibres): run offline where /root/reference exists,
    python tests/golden/make_t1m_fixture.py            -> tests/golden/t1m_reference.npz
tests/test_gpu_full_size.py and bench.py's t1m parity blocks read the file ("vs": "ref-fixture")."""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "t1m_reference.npz")


def scenes():
    """(name, pos [1,V,4], tri [T,3], resolution): exactly what bench.py builds (item 0)."""
    from nvdiffrast_amd.utils import big_mesh_batch, stress_triangles
    for name in ("t1m", "t1m_shuffled"):
        b = big_mesh_batch(1, attrs=4, shuffle=name.endswith("shuffled"))
        yield name, b["pos"][:1], b["tri"], (1024, 1024)
    b = stress_triangles(1, T=10000, res=1024)
    yield "s10k_1024", b["pos"][:1], b["tri"], (1024, 1024)


def digest(rast):
    ids = np.ascontiguousarray(rast[..., 3]).astype(np.uint32)
    return hashlib.sha256(ids.astype("<u4").tobytes()).hexdigest()


def main(only=None):
    from oracle import ref
    ref.build()
    assert ref.available(), "oracle/_ref is not built (needs /root/reference)"
    out = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    for name, pos, tri, res in scenes():
        if only and name not in only:
            continue
        t0 = time.time()
        rast, _ = ref.rasterize(pos, tri, res)
        cov = np.argwhere(rast[0, ..., 3] > 0)
        pick = cov[np.random.default_rng(1234).choice(len(cov), size=256, replace=False)]
        out[name + "/ids_sha256"] = np.frombuffer(digest(rast).encode(), dtype=np.uint8)
        out[name + "/covered"] = np.int64(len(cov))
        out[name + "/sample_yx"] = pick.astype(np.int32)
        out[name + "/sample_rast"] = rast[0, pick[:, 0], pick[:, 1]].astype(np.float32)
        print("%s: %d covered pixels, sha %s, %.0f s" % (name, len(cov), digest(rast)[:16], time.time() - t0), flush=True)
        np.savez_compressed(OUT, **out)


if __name__ == "__main__":
    main(sys.argv[1:])
