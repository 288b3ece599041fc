"""Parity of the four-op chain (rasterize -> interpolate(diff_attrs='all') -> trilinear texture -> antialias, forward and
backward) at any size -- TEST INFRASTRUCTURE ONLY (tests/ and bench.py's parity leg).

A chain through a texture cannot be judged end to end at full size: the uv gradient of bilinear sampling is DISCONTINUOUS
at texel boundaries, so of a million pixels sampling a 2048^2 white-noise texture a few hundred lie within one ulp of uv
(1.2e-7 x 2048 texels) of a boundary, take the neighbouring texel pair in one implementation and not in the other, and
move the summed gradients by O(100) -- in the reference's own two builds as much as here.  What is well defined is every
OP ON IDENTICAL INPUTS: the checker (the reference itself, oracle/_ref, or the oracle) is run op by op on the tensors the
path under test actually produced and consumed -- its own rast, uv, colours and upstream gradients -- so each comparison
carries the single-op bars of tests/conftest.py, and together they cover the whole chain.  The end-to-end differences are
returned as well, labelled as what they are (conditioning)."""
import numpy as np
import torch


def _err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def _mag(a):
    return float(np.abs(a).max())


def four_op_chain(dr, ctx, topo, chk, pos_np, tri_np, uv_np, tex_np, dy, res, dev="cuda", end_to_end=True):
    """Runs the chain on `dr` (an nvdiffrast.torch-compatible module) and the checker `chk` op by op on its tensors.
    dy: upstream gradient of the antialiased image (torch tensor on `dev`).  Returns a flat dict of max-abs errors
    (`*_err`), magnitudes (`*_max`) and the id mismatch count."""
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)          # noqa: E731
    N = lambda t: t.detach().cpu().numpy()                                   # noqa: E731
    pos = T(pos_np).requires_grad_(True)
    uvattr = T(uv_np).requires_grad_(True)
    tex = T(tex_np).requires_grad_(True)
    tri = T(tri_np)
    rast, rast_db = dr.rasterize(ctx, pos, tri, res)
    uv, uv_da = dr.interpolate(uvattr, rast, tri, rast_db=rast_db, diff_attrs="all")
    col = dr.texture(tex, uv, uv_da, filter_mode="linear-mipmap-linear")
    aa = dr.antialias(col, rast, pos, tri, topology_hash=topo)
    for t in (rast, rast_db, uv, uv_da, col):
        t.retain_grad()
    torch.autograd.backward(aa, dy)
    h = {k: N(v) for k, v in dict(rast=rast, rast_db=rast_db, uv=uv, uv_da=uv_da, col=col, aa=aa).items()}
    g = {k: N(v.grad) for k, v in dict(rast=rast, rast_db=rast_db, uv=uv, uv_da=uv_da, col=col, tex=tex, uvattr=uvattr, pos=pos).items()}
    dy_np = N(dy)
    out = {}
    kw = dict(filter_mode="linear-mipmap-linear")

    # ---- every op on the inputs the path under test gave it ------------------------------------------------------
    r, rdb = chk.rasterize(pos_np, tri_np, res)
    out["tri_id_mismatches"] = int((h["rast"][..., 3] != r[..., 3]).sum())
    out["coverage"] = float((r[..., 3] > 0).mean())
    out["bary_max_abs_err"] = _err(h["rast"][..., :3], r[..., :3])
    out["rast_db_err"], out["rast_db_max"] = _err(h["rast_db"], rdb), _mag(rdb)
    uv_c, uvda_c = chk.interpolate(uv_np, h["rast"], tri_np, h["rast_db"], "all")
    out["uv_err"] = _err(h["uv"], uv_c)
    out["uv_da_err"], out["uv_da_max"] = _err(h["uv_da"], uvda_c), _mag(uvda_c)
    col_c = chk.texture(tex_np, h["uv"], h["uv_da"], **kw)
    out["col_err"] = _err(h["col"], col_c)
    aa_c = chk.antialias(h["col"], h["rast"], pos_np, tri_np)
    out["aa_err"] = _err(h["aa"], aa_c)
    g_col_c, g_pos_aa_c = chk.antialias_grad(h["col"], h["rast"], pos_np, tri_np, dy_np)
    out["g_col_err"], out["g_col_max"] = _err(g["col"], g_col_c), _mag(g_col_c)
    tg = chk.texture_grad(tex_np, h["uv"], g["col"], h["uv_da"], **kw)
    out["g_tex_err"], out["g_tex_max"] = _err(g["tex"], tg["tex"]), _mag(tg["tex"])
    out["g_uv_err"], out["g_uv_max"] = _err(g["uv"], tg["uv"]), _mag(tg["uv"])
    out["g_uv_da_err"], out["g_uv_da_max"] = _err(g["uv_da"], tg["uv_da"]), _mag(tg["uv_da"])
    ga_c, g_rast_c, g_rdb_c = chk.interpolate_grad(uv_np, h["rast"], tri_np, g["uv"], h["rast_db"], g["uv_da"], "all")
    out["g_uvattr_err"], out["g_uvattr_max"] = _err(g["uvattr"], ga_c), _mag(ga_c)
    out["g_rast_err"], out["g_rast_max"] = _err(g["rast"], g_rast_c), _mag(g_rast_c)
    out["g_rast_db_err"], out["g_rast_db_max"] = _err(g["rast_db"], g_rdb_c), _mag(g_rdb_c)
    # pos receives the sum of two ops' gradients (rasterize and antialias), each on the path's own inputs
    g_pos_r = chk.rasterize_grad(pos_np, tri_np, h["rast"], g["rast"], g["rast_db"])
    out["g_pos_err"] = _err(g["pos"], g_pos_r + g_pos_aa_c)
    out["g_pos_max"] = max(_mag(g_pos_r), _mag(g_pos_aa_c), _mag(g_pos_r + g_pos_aa_c))

    # ---- the same chain end to end in the checker: conditioning of the chain, not parity of the kernels -------------
    if end_to_end:
        uv_e, uvda_e = chk.interpolate(uv_np, r, tri_np, rdb, "all")
        col_e = chk.texture(tex_np, uv_e, uvda_e, **kw)
        aa_e = chk.antialias(col_e, r, pos_np, tri_np)
        g_col_e, g_pos_aa_e = chk.antialias_grad(col_e, r, pos_np, tri_np, dy_np)
        tge = chk.texture_grad(tex_np, uv_e, g_col_e, uvda_e, **kw)
        ga_e, g_rast_e, g_rdb_e = chk.interpolate_grad(uv_np, r, tri_np, tge["uv"], rdb, tge["uv_da"], "all")
        g_pos_e = chk.rasterize_grad(pos_np, tri_np, r, g_rast_e, g_rdb_e) + g_pos_aa_e
        out["end_to_end"] = {
            "note": "whole chain in the checker vs whole chain in the path under test: dominated by pixels whose uv lies within an ulp "
                    "of a texel boundary (discontinuous uv gradient of bilinear sampling); not a parity figure",
            "col_err": _err(h["col"], col_e), "aa_err": _err(h["aa"], aa_e),
            "g_tex_err": _err(g["tex"], tge["tex"]), "g_tex_max": _mag(tge["tex"]),
            "g_uvattr_err": _err(g["uvattr"], ga_e), "g_uvattr_max": _mag(ga_e),
            "g_pos_err": _err(g["pos"], g_pos_e), "g_pos_max": _mag(g_pos_e)}
    return out
