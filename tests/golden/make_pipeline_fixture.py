#!/usr/bin/env python3
"""Generates tests/golden/pipeline_small.npz: inputs and oracle outputs of the whole path on one small scene
(rasterize -> interpolate(diff_attrs='all') -> trilinear texture -> antialias, forward and backward).

The reference cannot run in this environment (CUDA only), so these vectors come from the repo's own
oracle: they pin the ORACLE (tests/test_oracle_texture_aa.py::test_pipeline_fixture_is_reproduced) and give the GPU
suite a committed target (tests/test_gpu_end_to_end.py::test_pipeline_matches_committed_fixture); they do not
upgrade the parity status beyond what DESIGN.md section 2 states.

    python tests/golden/make_pipeline_fixture.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle                                                      # noqa: E402
from nvdiffrast_amd.utils import m10k_batch                        # noqa: E402

RES = (40, 56)


def inputs():
    b = m10k_batch(2, seed=77, nx=14, ny=7, attrs=3)
    rng = np.random.default_rng(78)
    tex = rng.uniform(size=(1, 32, 32, 3)).astype(np.float32)
    g_out = rng.normal(size=(2,) + RES + (3,)).astype(np.float32)
    return dict(pos=b["pos"], tri=b["tri"], uv=b["uv"], tex=tex, g_out=g_out)


def run_oracle(i):
    """Forward and backward of the op chain with the oracle; returns a dict of arrays."""
    rast, rast_db = oracle.rasterize(i["pos"], i["tri"], RES)
    uv, uv_da = oracle.interpolate(i["uv"], rast, i["tri"], rast_db=rast_db, diff_attrs="all")
    col = oracle.texture(i["tex"], uv, uv_da, filter_mode="linear-mipmap-linear")
    out = oracle.antialias(col, rast, i["pos"], i["tri"])
    g_col, g_pos_aa = oracle.antialias_grad(col, rast, i["pos"], i["tri"], i["g_out"])
    gt = oracle.texture_grad(i["tex"], uv, g_col, uv_da, filter_mode="linear-mipmap-linear")
    g_uvattr, g_rast, g_rast_db = oracle.interpolate_grad(i["uv"], rast, i["tri"], gt["uv"], rast_db=rast_db, dda=gt["uv_da"],
                                                          diff_attrs="all")
    g_pos_r = oracle.rasterize_grad(i["pos"], i["tri"], rast, g_rast, g_rast_db)
    return dict(rast=rast, rast_db=rast_db, uv=uv, uv_da=uv_da, col=col, out=out,
                g_tex=gt["tex"], g_uvattr=g_uvattr, g_pos=(g_pos_aa + g_pos_r).astype(np.float32))


if __name__ == "__main__":
    oracle.build()
    i = inputs()
    o = run_oracle(i)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pipeline_small.npz")
    np.savez_compressed(path, **{"in_" + k: v for k, v in i.items()}, **{"out_" + k: v for k, v in o.items()})
    print(path, os.path.getsize(path), "bytes;", "coverage %.2f" % float((o["rast"][..., 3] > 0).mean()))
