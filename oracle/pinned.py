"""Oracle pinned to the reference -- TEST INFRASTRUCTURE ONLY.

``PinnedOracle`` is what the ``oracle`` fixture of tests/conftest.py hands to every test: it forwards each call to
the C oracle (``oracle/*.c``) and, when ``oracle/_ref`` is present (the reference's own sources compiled for
the host, see oracle/ref.py), runs THE SAME CALL through the reference and asserts that both agree before
returning the oracle's result.  So every scene of every test -- CPU tests here, GPU parity tests on the GPU
box -- is checked against the reference itself, not only against this repo's restatement of it.

Agreement bars (oracle vs reference):
  * triangle-id channel: identical.  Where the f32 depth/cull arithmetic's FMA contraction decides an integer
    outcome the reference is bracketed by its two builds (``fma`` / ``nofma``, oracle/refshim/build.py):
    a pixel may disagree with the ``fma`` build only if it agrees with the ``nofma`` build;
  * forward floats: 1e-5 abs (barycentrics, z/w, attributes, texels, antialiased colours), pixel
    differentials 1e-5 relative to the largest magnitude;
  * gradients: 2e-5 relative to the largest magnitude of the tensor -- the reference sums with f32 atomics in
    launch order, the oracle in f64, so a sum of magnitude M cannot agree better than ~M * 2^-23 * sqrt(n).
"""
import os

import numpy as np

from . import ref as _ref

FWD_ATOL = 1e-5
GRAD_RTOL = 2e-5


class PinMismatch(AssertionError):
    pass


def _maxabs(a):
    a = np.asarray(a)
    if a.size == 0:
        return 0.0
    a = np.abs(a[np.isfinite(a)])
    return float(a.max()) if a.size else 0.0


class PinnedOracle:
    def __init__(self, oracle_module, max_pixels=8 << 20):
        self._o = oracle_module
        self.enabled = _ref.available("fma") and os.environ.get("NVDR_PIN", "1") != "0"
        self.have_nofma = _ref.available("nofma")
        self.max_pixels = max_pixels
        self.stats = {}                    # op -> [calls pinned, worst normalised deviation]

    def __getattr__(self, name):           # everything not overridden below goes straight to the oracle
        return getattr(self._o, name)

    # ------------------------------------------------------------------ helpers
    def _note(self, op, dev):
        s = self.stats.setdefault(op, [0, 0.0])
        s[0] += 1
        s[1] = max(s[1], dev)

    def _close(self, op, what, a, b, atol=None, rtol=None):
        if a is None and b is None:
            return
        if (a is None) != (b is None):
            raise PinMismatch("%s: %s is %s in the oracle but %s in the reference" % (op, what, type(a).__name__, type(b).__name__))
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
        if a.shape != b.shape:
            raise PinMismatch("%s: %s has shape %s in the oracle, %s in the reference" % (op, what, a.shape, b.shape))
        if a.size == 0:
            return
        tol = atol if atol is not None else rtol * max(1.0, _maxabs(b))
        bad = ~np.isclose(a, b, rtol=0.0, atol=tol, equal_nan=True)
        self._note(op, float(np.abs(a - b)[np.isfinite(a) & np.isfinite(b)].max(initial=0.0)) / tol)
        if bad.any():
            i = np.unravel_index(np.argmax(np.where(bad, np.abs(np.nan_to_num(a - b)), -1.0)), a.shape)
            raise PinMismatch("%s: oracle disagrees with the reference on %s: %d of %d values beyond %.3g (worst at %s: oracle %r, reference %r)"
                              % (op, what, int(bad.sum()), a.size, tol, i, a[i], b[i]))

    def _small_enough(self, *arrs):
        return all(int(np.prod(np.shape(a)[:3])) <= self.max_pixels for a in arrs)

    # ------------------------------------------------------------------ rasterize
    def rasterize(self, pos, tri, resolution, ranges=None, peel_depth=None, return_depth=False):
        out = self._o.rasterize(pos, tri, resolution, ranges=ranges, peel_depth=peel_depth, return_depth=return_depth)
        if self.enabled and peel_depth is None and self._small_enough(out[0]):
            self._pin_raster("rasterize", out[0], out[1], pos, tri, resolution, ranges)
        return out

    def _pin_raster(self, op, ro, dbo, pos, tri, resolution, ranges):
        r, db = _ref.rasterize(pos, tri, resolution, ranges)
        self._pin_raster_pair(op, ro, dbo, r, db, lambda: _ref.rasterize(pos, tri, resolution, ranges, variant="nofma"))

    def _pin_raster_pair(self, op, ro, dbo, r, db, nofma_fn):
        if r.shape != ro.shape:
            raise PinMismatch("%s: shape %s vs reference %s" % (op, ro.shape, r.shape))
        diff = ro[..., 3] != r[..., 3]
        same = ~diff
        if diff.any():
            ok = False
            if self.have_nofma:
                rn, _ = nofma_fn()
                ok = bool((ro[..., 3][diff] == rn[..., 3][diff]).all())
            if not ok:
                raise PinMismatch("%s: %d triangle ids differ from the reference (first at %s: oracle %r, reference %r)"
                                  % (op, int(diff.sum()), tuple(np.argwhere(diff)[0]), ro[..., 3][diff][0], r[..., 3][diff][0]))
        self._close(op, "rast(u,v,z/w)", ro[same][:, :3], r[same][:, :3], atol=FWD_ATOL)
        self._close(op, "rast_db", dbo[same], db[same], rtol=FWD_ATOL)

    def rasterize_ids(self, pos, tri, resolution, ranges=None, peel_depth=None):
        """Integer surfaces (ids, U32 depth): identical to the surfaces of the reference's CudaRaster context --
        every covered pixel's depth VALUE, not only the winning triangle."""
        ids, depth = self._o.rasterize_ids(pos, tri, resolution, ranges=ranges, peel_depth=peel_depth)
        if self.enabled and peel_depth is None and self._small_enough(ids):
            rid, rdepth = _ref.rasterize_surfaces(pos, tri, resolution, ranges)
            bad = ids != rid
            if bad.any() and self.have_nofma:
                nid, ndepth = _ref.rasterize_surfaces(pos, tri, resolution, ranges, variant="nofma")
                rid = np.where(bad & (ids == nid), nid, rid); rdepth = np.where(bad & (ids == nid), ndepth, rdepth)
                bad = ids != rid
            if bad.any():
                raise PinMismatch("rasterize_ids: %d triangle ids differ from the reference" % int(bad.sum()))
            cov = ids > 0
            dbad = cov & (depth != rdepth)
            self._note("rasterize_ids", 0.0)
            if dbad.any():
                i = tuple(np.argwhere(dbad)[0])
                raise PinMismatch("rasterize_ids: %d of %d covered pixels differ in their U32 depth (first at %s: oracle %d, reference %d)"
                                  % (int(dbad.sum()), int(cov.sum()), i, int(depth[i]), int(rdepth[i])))
        return ids, depth

    def rasterize_layers(self, pos, tri, resolution, num_layers, ranges=None):
        """Depth peeling: list of (rast, rast_db, depth) from the oracle's explicit-peel-surface form, each layer
        pinned to what the reference's context produces for peeling_idx = layer (ops.py:141-204)."""
        layers, peel = [], None
        for _k in range(num_layers):
            r, db, depth = self._o.rasterize(pos, tri, resolution, ranges=ranges, peel_depth=peel, return_depth=True)
            layers.append((r, db, depth))
            peel = depth
        if self.enabled and self._small_enough(layers[0][0]):
            ref_layers = _ref.rasterize_layers(pos, tri, resolution, num_layers, ranges)
            nofma = []
            for k in range(num_layers):
                def nf(k=k):
                    if not nofma:
                        nofma.extend(_ref.rasterize_layers(pos, tri, resolution, num_layers, ranges, variant="nofma"))
                    return nofma[k]
                self._pin_raster_pair("rasterize[layer %d]" % k, layers[k][0], layers[k][1], ref_layers[k][0], ref_layers[k][1], nf)
        return layers

    def rasterize_grad(self, pos, tri, rast, dy, ddb=None):
        g = self._o.rasterize_grad(pos, tri, rast, dy, ddb=ddb)
        if self.enabled and self._small_enough(rast):
            self._close("rasterize_grad", "g_pos", g, _ref.rasterize_grad(pos, tri, rast, dy, ddb), rtol=GRAD_RTOL)
        return g

    # ------------------------------------------------------------------ interpolate
    def interpolate(self, attr, rast, tri, rast_db=None, diff_attrs=None):
        out, da = self._o.interpolate(attr, rast, tri, rast_db=rast_db, diff_attrs=diff_attrs)
        if self.enabled and self._small_enough(rast):
            ro, rda = _ref.interpolate(attr, rast, tri, rast_db, diff_attrs)
            self._close("interpolate", "out", out, ro, rtol=FWD_ATOL)
            self._close("interpolate", "out_da", da, rda, rtol=FWD_ATOL)
        return out, da

    def interpolate_grad(self, attr, rast, tri, dy, rast_db=None, dda=None, diff_attrs=None):
        g = self._o.interpolate_grad(attr, rast, tri, dy, rast_db=rast_db, dda=dda, diff_attrs=diff_attrs)
        if self.enabled and self._small_enough(rast):
            rg = _ref.interpolate_grad(attr, rast, tri, dy, rast_db, dda, diff_attrs)
            self._close("interpolate_grad", "g_attr", g[0], rg[0], rtol=GRAD_RTOL)
            self._close("interpolate_grad", "g_rast", g[1][..., :2], rg[1][..., :2], rtol=GRAD_RTOL)
            if g[2] is not None:
                self._close("interpolate_grad", "g_rast_db", g[2], rg[2], rtol=GRAD_RTOL)
        return g

    # ------------------------------------------------------------------ texture
    def texture_build_mip(self, tex, max_mip_level=-1):
        levels = self._o.texture_build_mip(tex, max_mip_level)
        if self.enabled and levels:
            flat = _ref.texture_build_mip(tex, max_mip_level, cube=(np.ndim(tex) == 5))
            mine = np.concatenate([l.reshape(-1) for l in levels])
            self._close("texture_construct_mip", "mip levels", mine, flat[:mine.size], atol=FWD_ATOL)
        return levels

    def texture(self, tex, uv, uv_da=None, mip_level_bias=None, mip=None, filter_mode="auto", boundary_mode="wrap", max_mip_level=None):
        kw = dict(uv_da=uv_da, mip_level_bias=mip_level_bias, mip=mip, filter_mode=filter_mode, boundary_mode=boundary_mode, max_mip_level=max_mip_level)
        out = self._o.texture(tex, uv, **kw)
        if self.enabled and self._small_enough(uv):
            self._close("texture[%s,%s]" % (filter_mode, boundary_mode), "out", out, _ref.texture(tex, uv, **kw), rtol=FWD_ATOL)
        return out

    def texture_grad(self, tex, uv, dy, uv_da=None, mip_level_bias=None, mip=None, filter_mode="auto", boundary_mode="wrap", max_mip_level=None):
        kw = dict(uv_da=uv_da, mip_level_bias=mip_level_bias, mip=mip, filter_mode=filter_mode, boundary_mode=boundary_mode, max_mip_level=max_mip_level)
        g = self._o.texture_grad(tex, uv, dy, **kw)
        if self.enabled and self._small_enough(uv):
            rg = _ref.texture_grad(tex, uv, dy, **kw)
            op = "texture_grad[%s,%s]" % (filter_mode, boundary_mode)
            for k in ("tex", "uv", "uv_da", "mip_level_bias"):
                if g[k] is not None:
                    self._close(op, "g_" + k, g[k], rg[k], rtol=GRAD_RTOL)
            if g["mip"] is not None:
                for i, (a, b) in enumerate(zip(g["mip"], rg["mip"])):
                    self._close(op, "g_mip[%d]" % (i + 1), a, b, rtol=GRAD_RTOL)
        return g

    # ------------------------------------------------------------------ antialias
    def antialias(self, color, rast, pos, tri):
        out = self._o.antialias(color, rast, pos, tri)
        # (a table with indices >= V makes the reference read pos[] out of bounds -- see oracle/antialias.c analyze(): not pinned)
        if self.enabled and self._small_enough(rast) and int(np.max(tri, initial=0)) < np.shape(pos)[-2]:
            r = _ref.antialias(color, rast, pos, tri)
            try:
                self._close("antialias", "out", out, r, rtol=FWD_ATOL)
            except PinMismatch:
                d = os.environ.get("NVDR_PIN_DUMP")          # development: keep the case for a replay on the CPU
                if d:
                    np.savez(os.path.join(d, "antialias_case.npz"), color=color, rast=rast, pos=pos, tri=tri, oracle=out, ref=r,
                             oracle2=self._o.antialias(color, rast, pos, tri), ref2=_ref.antialias(color, rast, pos, tri))
                raise
        return out

    def antialias_grad(self, color, rast, pos, tri, dy):
        gc, gp = self._o.antialias_grad(color, rast, pos, tri, dy)
        if self.enabled and self._small_enough(rast) and int(np.max(tri, initial=0)) < np.shape(pos)[-2]:
            rc, rp = _ref.antialias_grad(color, rast, pos, tri, dy)
            self._close("antialias_grad", "g_color", gc, rc, rtol=GRAD_RTOL)
            self._close("antialias_grad", "g_pos", gp, rp, rtol=GRAD_RTOL)
        return gc, gp
