/*
 * oracle/raster.c -- CPU restatement of the reference rasterizer.  TEST INFRASTRUCTURE ONLY
 * (see nvdr_oracle.h).  Build with -ffp-contract=off: every fused multiply-add below
 * is written explicitly (fmaf) at the sites where nvcc's default -fmad=true would
 * contract the reference expression; all other float ops round separately.
 *
 * Parity: triangle ids identical to the reference's own CudaRaster (oracle/_ref, FMA build) on every test
 * scene -- snapping, culls, clipper, depth ties, peeling, range mode, viewport tiling -- and docs/img/tri.png
 * bit for bit.  Where contraction decides a depth comparison the reference's two builds bracket the result.
 */
#include "nvdr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SP_LOG2      4            /* Constants.hpp:14  CR_SUBPIXEL_LOG2          */
#define TILE         8            /* Constants.hpp:18  CR_TILE_LOG2 = 3          */
#define MAX_VIEWPORT 2048         /* Constants.hpp:13  CR_MAXVIEWPORT_LOG2 = 11  */
#define LERP_ERR0    2200u        /* Constants.hpp:69  CR_LERP_ERROR(0)          */
#define DEPTH_MIN    17600u       /* Constants.hpp:70  CR_LERP_ERROR(3)          */
#define DEPTH_MAX    (0xFFFFFFFFu - 17600u) /* Constants.hpp:71 */

int nvdro_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- PTX conversion semantics (Util.inl:31-35) ------------ */

static int32_t cvt_rni_sat_s32(float a)
{
    if (isnan(a)) return 0;
    if (a >= 2147483648.0f) return INT32_MAX;
    if (a <= -2147483648.0f) return INT32_MIN;
    return (int32_t)lrintf(a);                  /* default rounding mode = nearest-even */
}

/* C-style (U32)float cast as nvcc compiles it: cvt.rzi.u32.f32, which saturates. */
static uint32_t cvt_rzi_u32(float a)
{
    if (isnan(a) || a <= 0.0f) return 0u;
    if (a >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)a;
}

static int32_t f2bits(float f) { int32_t i; memcpy(&i, &f, 4); return i; }

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
static int imin3(int a, int b, int c) { return imin(imin(a, b), c); }
static int imax3(int a, int b, int c) { return imax(imax(a, b), c); }

/* ---- sub-triangle record ----------------------------------------------------- */

typedef struct {
    int32_t  px[3], py[3];      /* snapped vertices, subpixels rel. viewport centre, CCW */
    uint32_t zx, zy, zb;        /* U32 depth plane: depth = zx*X + zy*Y + zb            */
    int32_t  id;                /* parent triangle index + 1                            */
} SubTri;

typedef struct {
    int   vpw, vph;             /* viewport size in pixels (unpadded)                   */
    float xs, ys, xo, yo;       /* viewport-tile transform, RasterImpl.cpp:295-298      */
} Viewport;

/* Depth plane: Util.inl:184-210 (setupPleq). zv = per-vertex depth values (f32),
 * (v0x,v0y) = vertex 0 in subpixels relative to the bottom-left sample. */
static void setup_depth_plane(const float zv[3], int v0x, int v0y,
                              int d1x, int d1y, int d2x, int d2y, float area_rcp,
                              uint32_t* zx, uint32_t* zy, uint32_t* zb)
{
    float mx = fmaxf(fmaxf(zv[0], zv[1]), zv[2]);
    int sh = imin(imax((f2bits(mx) >> 23) - (127 + 22), 0), 8);
    int32_t t0 = (int32_t)(cvt_rzi_u32(zv[0]) >> sh);
    int32_t t1 = (int32_t)(cvt_rzi_u32(zv[1]) >> sh) - t0;
    int32_t t2 = (int32_t)(cvt_rzi_u32(zv[2]) >> sh) - t0;

    uint32_t rcp_mant = ((uint32_t)f2bits(area_rcp) & 0x007FFFFFu) | 0x00800000u;
    int rcp_shift = (23 + 127) - (f2bits(area_rcp) >> 23);

    int64_t xc = ((int64_t)t1 * d2y - (int64_t)t2 * d1y) * (int64_t)rcp_mant;
    int64_t yc = ((int64_t)t2 * d1x - (int64_t)t1 * d2x) * (int64_t)rcp_mant;
    uint32_t px = (uint32_t)(xc >> (rcp_shift - (sh + SP_LOG2)));
    uint32_t py = (uint32_t)(yc >> (rcp_shift - (sh + SP_LOG2)));

    int32_t cx = (v0x * 2 + imin3(d1x, d2x, 0) + imax3(d1x, d2x, 0)) >> (SP_LOG2 + 1);
    int32_t cy = (v0y * 2 + imin3(d1y, d2y, 0) + imax3(d1y, d2y, 0)) >> (SP_LOG2 + 1);
    int32_t vcx = v0x - cx * (1 << SP_LOG2);
    int32_t vcy = v0y - cy * (1 << SP_LOG2);

    uint32_t pz = (uint32_t)t0 << sh;
    pz -= (uint32_t)(((xc >> 13) * vcx + (yc >> 13) * vcy) >> (rcp_shift - (sh + 13)));
    pz -= px * (uint32_t)cx + py * (uint32_t)cy;
    *zx = px; *zy = py; *zb = pz;
}

/* Snap + cull + setup of one (sub)triangle: TriangleSetup.inl:11-24 (snapTriangle),
 * :42-116 (prepareTriangle), :120-177 (setupTriangle).  v[k] = {x,y,z,w} in the
 * viewport tile's clip space.  Returns 1 and fills *st if the triangle survives. */
static int snap_cull_setup(const Viewport* vp, const float v[3][4], int id, SubTri* st)
{
    float vsx = (float)(vp->vpw << (SP_LOG2 - 1));
    float vsy = (float)(vp->vph << (SP_LOG2 - 1));
    float rw[3];
    int px[3], py[3];
    for (int k = 0; k < 3; k++) {
        rw[k] = 1.0f / v[k][3];
        px[k] = cvt_rni_sat_s32(v[k][0] * rw[k] * vsx);     /* two separate multiplies */
        py[k] = cvt_rni_sat_s32(v[k][1] * rw[k] * vsy);
    }
    int lox = imin3(px[0], px[1], px[2]), loy = imin3(py[0], py[1], py[2]);
    int hix = imax3(px[0], px[1], px[2]), hiy = imax3(py[0], py[1], py[2]);

    /* prepareTriangle: degenerate => cull (no backface culling, torch_rasterize.cpp:94). */
    int d1x = px[1] - px[0], d1y = py[1] - py[0];
    int d2x = px[2] - px[0], d2y = py[2] - py[0];
    int32_t area = (int32_t)((uint32_t)d1x * (uint32_t)d2y - (uint32_t)d1y * (uint32_t)d2x);
    if (area == 0) return 0;

    /* AABB falls between sample points => cull (TriangleSetup.inl:59-70). */
    int ss = 1 << SP_LOG2;
    int bx = (vp->vpw << (SP_LOG2 - 1)) - (ss >> 1);
    int by = (vp->vph << (SP_LOG2 - 1)) - (ss >> 1);
    int alox = (lox + ss - 1 + bx) & -ss, aloy = (loy + ss - 1 + by) & -ss;
    int ahix = (hix + bx) & -ss,          ahiy = (hiy + by) & -ss;
    if (alox > ahix || aloy > ahiy) return 0;

    /* AABB holds one or two samples => cull unless one is covered (:72-111). */
    int diff = ahix + ahiy - alox - aloy;
    if (diff <= ss) {
        int ok = 0;
        for (int pass = 0; pass < 2 && !ok; pass++) {
            int sx = pass ? ahix : alox, sy = pass ? ahiy : aloy;
            if (pass && diff == 0) break;
            int tx[3], ty[3];
            for (int k = 0; k < 3; k++) { tx[k] = px[k] + bx - sx; ty[k] = py[k] + by - sy; }
            int e0 = tx[0] * ty[1] - ty[0] * tx[1];
            int e1 = tx[1] * ty[2] - ty[1] * tx[2];
            int e2 = tx[2] * ty[0] - ty[2] * tx[0];
            if (area < 0) { e0 = -e0; e1 = -e1; e2 = -e2; }
            ok = !(e0 < 0 || e1 < 0 || e2 < 0);
        }
        if (!ok) return 0;
    }

    /* setupTriangle: make CCW by swapping vertices 1,2 (:130-137). */
    float vz[3] = { v[0][2], v[1][2], v[2][2] };
    if (area < 0) {
        int t;
        t = d1x; d1x = d2x; d2x = t;   t = d1y; d1y = d2y; d2y = t;
        t = px[1]; px[1] = px[2]; px[2] = t;   t = py[1]; py[1] = py[2]; py[2] = t;
        float f = vz[1]; vz[1] = vz[2]; vz[2] = f;
        f = rw[1]; rw[1] = rw[2]; rw[2] = f;
        area = -area;
    }

    /* Depth values per vertex (:145-151).  nvcc contracts (z*zcoef)*rcpW + zbias into
     * fma(z*zcoef, rcpW, zbias); restated explicitly. */
    float zcoef = (float)(DEPTH_MAX - DEPTH_MIN) * 0.5f;
    float zbias = (float)(DEPTH_MAX + DEPTH_MIN) * 0.5f;   /* U32 sum wraps to 2^32-1 */
    float zv[3];
    for (int k = 0; k < 3; k++) zv[k] = fmaf(vz[k] * zcoef, rw[k], zbias);

    int wv0x = px[0] + (vp->vpw << (SP_LOG2 - 1));
    int wv0y = py[0] + (vp->vph << (SP_LOG2 - 1));
    setup_depth_plane(zv, wv0x - (1 << (SP_LOG2 - 1)), wv0y - (1 << (SP_LOG2 - 1)),
                      d1x, d1y, d2x, d2y, 1.0f / (float)area, &st->zx, &st->zy, &st->zb);
    for (int k = 0; k < 3; k++) { st->px[k] = px[k]; st->py[k] = py[k]; }
    st->id = id;
    return 1;
}

/* Sutherland-Hodgman in barycentric space: Util.inl:101-130 (clipPolygonWithPlane).
 * Plane value at barycentric (u,v) is f0 + f1*u + f2*v, inside iff >= 0.
 * FMA sites follow nvcc's left-to-right contraction of the reference expressions. */
static int clip_poly_plane(float* out, const float* in, int n_in, float f0, float f1, float f2)
{
    int n_out = 0;
    if (n_in >= 3) {
        int ai = (n_in - 1) * 2;
        float av = fmaf(f2, in[ai + 1], fmaf(f1, in[ai + 0], f0));
        for (int bi = 0; bi < n_in * 2; bi += 2) {
            float bv = fmaf(f2, in[bi + 1], fmaf(f1, in[bi + 0], f0));
            if (av * bv < 0.0f) {
                float bc = av / (av - bv);
                float ac = 1.0f - bc;
                out[n_out + 0] = fmaf(in[ai + 0], ac, in[bi + 0] * bc);
                out[n_out + 1] = fmaf(in[ai + 1], ac, in[bi + 1] * bc);
                n_out += 2;
            }
            if (bv >= 0.0f) {
                out[n_out + 0] = in[bi + 0];
                out[n_out + 1] = in[bi + 1];
                n_out += 2;
            }
            ai = bi;
            av = bv;
        }
    }
    return n_out >> 1;
}

/* Util.inl:134-160 (clipTriangleWithFrustum). bary holds up to 9 (u,v) pairs. */
static int clip_triangle_frustum(float* bary, const float v0[4], const float v1[4],
                                 const float v2[4], const float d1[4], const float d2[4])
{
    int num = 3;
    bary[0] = 0.f; bary[1] = 0.f; bary[2] = 1.f; bary[3] = 0.f; bary[4] = 0.f; bary[5] = 1.f;
    for (int ax = 0; ax < 3; ax++) {
        if ((v0[3] < fabsf(v0[ax])) | (v1[3] < fabsf(v1[ax])) | (v2[3] < fabsf(v2[ax]))) {
            float tmp[18];
            num = clip_poly_plane(tmp, bary, num, v0[3] + v0[ax], d1[3] + d1[ax], d2[3] + d2[ax]);
            num = clip_poly_plane(bary, tmp, num, v0[3] - v0[ax], d1[3] - d1[ax], d2[3] - d2[ax]);
        }
    }
    return num;
}

/* Per-triangle setup: TriangleSetup.inl:181-435.  vin = the three clip-space vertices of
 * the triangle (already fetched, indices validated by the caller).  Emits up to 7
 * sub-triangles in fan order; returns the count. */
static int setup_triangle(const Viewport* vp, const float vin[3][4], int id, SubTri out[7])
{
    float v[3][4];
    for (int k = 0; k < 3; k++) {
        /* Viewport-tile transform (:262-267); a*b + c*d -> fma(a, b, c*d). */
        v[k][0] = fmaf(vin[k][0], vp->xs, vin[k][3] * vp->xo);
        v[k][1] = fmaf(vin[k][1], vp->ys, vin[k][3] * vp->yo);
        v[k][2] = vin[k][2];
        v[k][3] = vin[k][3];
    }

    /* All three vertices outside one frustum plane => cull (:271-283). */
    if ((v[0][3] < fabsf(v[0][0])) | (v[0][3] < fabsf(v[0][1])) | (v[0][3] < fabsf(v[0][2]))) {
        for (int ax = 0; ax < 3; ax++) {
            int pos_out = (v[0][3] < +v[0][ax]) & (v[1][3] < +v[1][ax]) & (v[2][3] < +v[2][ax]);
            int neg_out = (v[0][3] < -v[0][ax]) & (v[1][3] < -v[1][ax]) & (v[2][3] < -v[2][ax]);
            if (pos_out | neg_out) return 0;
        }
    }

    /* Entirely inside the frustum => single triangle, no clipper (:329-352). */
    int inside = 1;
    for (int k = 0; k < 3; k++)
        inside = inside && (v[k][3] >= fmaxf(fmaxf(fabsf(v[k][0]), fabsf(v[k][1])), fabsf(v[k][2])));
    if (inside)
        return snap_cull_setup(vp, (const float (*)[4])v, id, &out[0]);

    /* Clip (:355-434). */
    float d1[4], d2[4], bary[18];
    for (int c = 0; c < 4; c++) { d1[c] = v[1][c] - v[0][c]; d2[c] = v[2][c] - v[0][c]; }
    int nv = clip_triangle_frustum(bary, v[0], v[1], v[2], d1, d2);

    float cv[9][4];
    for (int i = 0; i < nv; i++)
        for (int c = 0; c < 4; c++)
            cv[i][c] = fmaf(d2[c], bary[i * 2 + 1], fmaf(d1[c], bary[i * 2 + 0], v[0][c]));

    int n = 0;
    for (int i = 2; i < nv; i++) {
        float t[3][4];
        memcpy(t[0], cv[0], 16); memcpy(t[1], cv[i - 1], 16); memcpy(t[2], cv[i], 16);
        if (snap_cull_setup(vp, (const float (*)[4])t, id, &out[n])) n++;
    }
    return n;
}

/* ---- coverage + ROP ------------------------------------------------------------ */

/* Rasterize one sub-triangle into a viewport-tile's id/depth surface.
 * Samples: FineRaster.inl:77-78 (pixel i sits at 16*i - (vpw-1)*8 subpixels).
 * Fill rule: Util.inl:304-309 (edge exclusive iff dy>0 || (dy==0 && dx<=0)).
 * Depth + ROP: FineRaster.inl:345-361, 152-172 (fragment survives iff depth <= stored,
 * and when peeling iff depth > peel; fragments arrive in ascending triangle order). */
static void raster_subtri(const SubTri* s, const Viewport* vp, int stride,
                          uint32_t* idb, uint32_t* depb, const uint32_t* peelb)
{
    int bx = (vp->vpw - 1) << (SP_LOG2 - 1);
    int by = (vp->vph - 1) << (SP_LOG2 - 1);
    int lox = imin3(s->px[0], s->px[1], s->px[2]), hix = imax3(s->px[0], s->px[1], s->px[2]);
    int loy = imin3(s->py[0], s->py[1], s->py[2]), hiy = imax3(s->py[0], s->py[1], s->py[2]);
    /* pixel index range whose sample lies inside the AABB */
    int x0 = (lox + bx + 15) >> 4, x1 = (hix + bx) >> 4;
    int y0 = (loy + by + 15) >> 4, y1 = (hiy + by) >> 4;
    x0 = imax(x0, 0); y0 = imax(y0, 0);
    x1 = imin(x1, vp->vpw - 1); y1 = imin(y1, vp->vph - 1);

    int ex[3], ey[3], excl[3];
    for (int e = 0; e < 3; e++) {
        int a = e, b = (e + 1) % 3;
        ex[e] = s->px[b] - s->px[a];
        ey[e] = s->py[b] - s->py[a];
        excl[e] = (ey[e] > 0 || (ey[e] == 0 && ex[e] <= 0));
    }
    for (int y = y0; y <= y1; y++) {
        int sy = y * 16 - by;
        for (int x = x0; x <= x1; x++) {
            int sx = x * 16 - bx;
            int in = 1;
            for (int e = 0; e < 3 && in; e++) {
                /* E = (a - s) x d */
                int32_t E = (s->px[e] - sx) * ey[e] - (s->py[e] - sy) * ex[e];
                in = excl[e] ? (E > 0) : (E >= 0);
            }
            if (!in) continue;
            uint32_t depth = s->zx * (uint32_t)x + s->zy * (uint32_t)y + s->zb;
            size_t pi = (size_t)x + (size_t)stride * (size_t)y;
            if (peelb && depth <= peelb[pi]) continue;
            if (depth > depb[pi]) continue;
            depb[pi] = depth;
            idb[pi] = (uint32_t)s->id;
        }
    }
}

/* Full integer stage for all images: torch_rasterize.cpp:76-124 + RasterImpl.cpp. */
static int raster_ids(const float* pos, const int32_t* tri, const int32_t* ranges,
                      int instance_mode, int N, int V, int T, int H, int W,
                      int peel, const uint32_t* peel_buf, uint32_t* depth_buf, uint32_t* id_buf)
{
    int Hp = (H + TILE - 1) & -TILE, Wp = (W + TILE - 1) & -TILE;
    int tcx = (Wp + MAX_VIEWPORT - 1) / MAX_VIEWPORT, tcy = (Hp + MAX_VIEWPORT - 1) / MAX_VIEWPORT;
    int tsx = ((Wp + tcx - 1) / tcx + TILE - 1) & -TILE;
    int tsy = ((Hp + tcy - 1) / tcy + TILE - 1) & -TILE;

#pragma omp parallel for schedule(dynamic, 1)
    for (int n = 0; n < N; n++) {
        uint32_t* idb = id_buf + (size_t)n * Hp * Wp;
        uint32_t* depb = depth_buf + (size_t)n * Hp * Wp;
        const uint32_t* peelb = (peel && peel_buf) ? peel_buf + (size_t)n * Hp * Wp : NULL;
        const float* vb = instance_mode ? pos + (size_t)n * V * 4 : pos;
        int t_off = instance_mode ? 0 : ranges[n * 2 + 0];
        int t_cnt = instance_mode ? T : ranges[n * 2 + 1];

        /* deferredClear(0): colour 0, depth CR_DEPTH_MAX (RasterImpl.cpp:306-307). */
        for (size_t i = 0; i < (size_t)Hp * Wp; i++) { idb[i] = 0u; depb[i] = DEPTH_MAX; }

        for (int ty = 0; ty < tcy; ty++)
        for (int tx = 0; tx < tcx; tx++) {
            int offx = tx * tsx, offy = ty * tsy;
            Viewport vp;
            vp.vpw = (W - offx) < tsx ? (W - offx) : tsx;
            vp.vph = (H - offy) < tsy ? (H - offy) : tsy;
            if (vp.vpw <= 0 || vp.vph <= 0) continue;
            vp.xs = (float)W / (float)vp.vpw;
            vp.ys = (float)H / (float)vp.vph;
            vp.xo = (float)(W - vp.vpw - 2 * offx) / (float)vp.vpw;
            vp.yo = (float)(H - vp.vph - 2 * offy) / (float)vp.vph;
            size_t boff = (size_t)offx + (size_t)offy * Wp;

            for (int i = 0; i < t_cnt; i++) {
                int t = t_off + i;
                if ((uint32_t)t >= (uint32_t)T) continue;                 /* :228-233 */
                uint32_t i0 = (uint32_t)tri[t * 3 + 0], i1 = (uint32_t)tri[t * 3 + 1], i2 = (uint32_t)tri[t * 3 + 2];
                if (i0 >= (uint32_t)V || i1 >= (uint32_t)V || i2 >= (uint32_t)V) continue; /* :241-248 */
                float vin[3][4];
                memcpy(vin[0], vb + (size_t)i0 * 4, 16);
                memcpy(vin[1], vb + (size_t)i1 * 4, 16);
                memcpy(vin[2], vb + (size_t)i2 * 4, 16);
                SubTri st[7];
                int ns = setup_triangle(&vp, (const float (*)[4])vin, t + 1, st);
                for (int k = 0; k < ns; k++)
                    raster_subtri(&st[k], &vp, Wp, idb + boff, depb + boff, peelb ? peelb + boff : NULL);
            }
        }
    }
    return 0;
}

int nvdro_rasterize_ids(const float* pos, const int32_t* tri, const int32_t* ranges,
                        int instance_mode, int N, int V, int T, int H, int W,
                        int peel, const uint32_t* peel_buf, uint32_t* depth_buf,
                        uint32_t* id_buf)
{
    return raster_ids(pos, tri, ranges, instance_mode, N, V, T, H, W, peel, peel_buf, depth_buf, id_buf);
}

/* ---- pixel shader: rasterize.cu:15-114 ------------------------------------------- */

static float tri_id_to_float(int x)      /* common.h:193 */
{
    if (x <= 0x01000000) return (float)x;
    int32_t b = 0x4a800000 + x; float f; memcpy(&f, &b, 4); return f;
}

static int float_to_tri_id(float x)      /* common.h:192 */
{
    if (x <= 16777216.f) return (int)x;
    return f2bits(x) - 0x4a800000;
}

static float satf(float x) { return (x != x) ? 0.0f : fminf(fmaxf(x, 0.0f), 1.0f); } /* __saturatef: NaN -> +0 */

int nvdro_rasterize_fwd(const float* pos, const int32_t* tri, const int32_t* ranges,
                        int instance_mode, int N, int V, int T, int H, int W,
                        int peel, const uint32_t* peel_buf, uint32_t* depth_buf,
                        float* out, float* out_db)
{
    int Hp = (H + TILE - 1) & -TILE, Wp = (W + TILE - 1) & -TILE;
    uint32_t* idb = (uint32_t*)malloc((size_t)N * Hp * Wp * 4);
    if (!idb) return 1;
    raster_ids(pos, tri, ranges, instance_mode, N, V, T, H, W, peel, peel_buf, depth_buf, idb);

    float xs = 2.f / (float)W, xo = 1.f / (float)W - 1.f;
    float ys = 2.f / (float)H, yo = 1.f / (float)H - 1.f;

#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; n++)
    for (int py = 0; py < H; py++)
    for (int px = 0; px < W; px++) {
        size_t po = ((size_t)n * H + py) * W + px;
        float* o = out + po * 4;
        float* odb = out_db + po * 4;
        int ti = (int)idb[((size_t)n * Hp + py) * Wp + px] - 1;
        if (ti < 0 || ti >= T) { for (int c = 0; c < 4; c++) { o[c] = 0.f; odb[c] = 0.f; } continue; }
        int vi0 = tri[ti * 3 + 0], vi1 = tri[ti * 3 + 1], vi2 = tri[ti * 3 + 2];
        if (vi0 < 0 || vi0 >= V || vi1 < 0 || vi1 >= V || vi2 < 0 || vi2 >= V) {
            /* reference leaves torch::empty memory untouched here; oracle writes zeros */
            for (int c = 0; c < 4; c++) { o[c] = 0.f; odb[c] = 0.f; }
            continue;
        }
        const float* vb = instance_mode ? pos + (size_t)n * V * 4 : pos;
        const float* p0 = vb + (size_t)vi0 * 4; const float* p1 = vb + (size_t)vi1 * 4; const float* p2 = vb + (size_t)vi2 * 4;

        float fx = fmaf(xs, (float)px, xo);
        float fy = fmaf(ys, (float)py, yo);
        float p0x = fmaf(-fx, p0[3], p0[0]), p0y = fmaf(-fy, p0[3], p0[1]);
        float p1x = fmaf(-fx, p1[3], p1[0]), p1y = fmaf(-fy, p1[3], p1[1]);
        float p2x = fmaf(-fx, p2[3], p2[0]), p2y = fmaf(-fy, p2[3], p2[1]);
        float a0 = fmaf(p1x, p2y, -(p1y * p2x));
        float a1 = fmaf(p2x, p0y, -(p2y * p0x));
        float a2 = fmaf(p0x, p1y, -(p0y * p1x));

        float iw = 1.f / (a0 + a1 + a2);
        float b0 = a0 * iw, b1 = a1 * iw;
        float z = fmaf(p2[2], a2, fmaf(p1[2], a1, p0[2] * a0));
        float w = fmaf(p2[3], a2, fmaf(p1[3], a1, p0[3] * a0));
        float zw = z / w;

        b0 = satf(b0); b1 = satf(b1);
        float bs = 1.f / fmaxf(b0 + b1, 1.f);
        b0 *= bs; b1 *= bs;
        zw = fmaxf(fminf(zw, 1.f), -1.f);
        o[0] = b0; o[1] = b1; o[2] = zw; o[3] = tri_id_to_float(ti + 1);

        float dfxdx = xs * iw, dfydy = ys * iw;
        float da0dx = fmaf(p2[1], p1[3], -(p1[1] * p2[3])), da0dy = fmaf(p1[0], p2[3], -(p2[0] * p1[3]));
        float da1dx = fmaf(p0[1], p2[3], -(p2[1] * p0[3])), da1dy = fmaf(p2[0], p0[3], -(p0[0] * p2[3]));
        float da2dx = fmaf(p1[1], p0[3], -(p0[1] * p1[3])), da2dy = fmaf(p0[0], p1[3], -(p1[0] * p0[3]));
        float datdx = da0dx + da1dx + da2dx, datdy = da0dy + da1dy + da2dy;
        odb[0] = dfxdx * fmaf(b0, datdx, -da0dx);
        odb[1] = dfydy * fmaf(b0, datdy, -da0dy);
        odb[2] = dfxdx * fmaf(b1, datdx, -da1dx);
        odb[3] = dfydy * fmaf(b1, datdy, -da1dy);
    }
    free(idb);
    return 0;
}

/* ---- backward: rasterize.cu:119-277 ------------------------------------------------- */

int nvdro_rasterize_grad(const float* pos, const int32_t* tri, const float* out,
                         const float* dy, const float* ddb,
                         int instance_mode, int N, int V, int T, int H, int W,
                         float* grad_pos)
{
    float xs = 2.f / (float)W, xo = 1.f / (float)W - 1.f;
    float ys = 2.f / (float)H, yo = 1.f / (float)H - 1.f;
    size_t gsz = (size_t)(instance_mode ? N : 1) * V * 4;
    double* acc = (double*)calloc(gsz, sizeof(double));
    if (!acc) return 1;

    /* Instanced: images own disjoint gradient slices -> parallel over images.
     * Range mode: one shared slice -> serial, fixed pixel order. */
#pragma omp parallel for schedule(dynamic, 1) if (instance_mode)
    for (int n = 0; n < N; n++)
    for (int py = 0; py < H; py++)
    for (int px = 0; px < W; px++) {
        size_t pidx = ((size_t)n * H + py) * W + px;
        float dyx = dy[pidx * 4 + 0], dyy = dy[pidx * 4 + 1];
        float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
        if (ddb) { d0 = ddb[pidx * 4 + 0]; d1 = ddb[pidx * 4 + 1]; d2 = ddb[pidx * 4 + 2]; d3 = ddb[pidx * 4 + 3]; }
        int ti = float_to_tri_id(out[pidx * 4 + 3]) - 1;
        if (ti < 0 || ti >= T) continue;
        int all_dy = f2bits(dyx) | f2bits(dyy);
        int all_ddb = ddb ? (f2bits(d0) | f2bits(d1) | f2bits(d2) | f2bits(d3)) : 0;
        if ((uint32_t)((uint32_t)(all_dy | all_ddb) << 1) == 0u) continue;

        int vi0 = tri[ti * 3 + 0], vi1 = tri[ti * 3 + 1], vi2 = tri[ti * 3 + 2];
        if (vi0 < 0 || vi0 >= V || vi1 < 0 || vi1 >= V || vi2 < 0 || vi2 >= V) continue;
        size_t vo = instance_mode ? (size_t)n * V : 0;
        const float* p0 = pos + (vo + vi0) * 4; const float* p1 = pos + (vo + vi1) * 4; const float* p2 = pos + (vo + vi2) * 4;

        float fx = fmaf(xs, (float)px, xo), fy = fmaf(ys, (float)py, yo);
        float p0x = fmaf(-fx, p0[3], p0[0]), p0y = fmaf(-fy, p0[3], p0[1]);
        float p1x = fmaf(-fx, p1[3], p1[0]), p1y = fmaf(-fy, p1[3], p1[1]);
        float p2x = fmaf(-fx, p2[3], p2[0]), p2y = fmaf(-fy, p2[3], p2[1]);
        float a0 = fmaf(p1x, p2y, -(p1y * p2x));
        float a1 = fmaf(p2x, p0y, -(p2y * p0x));
        float a2 = fmaf(p0x, p1y, -(p0y * p1x));

        float at = a0 + a1 + a2;
        float ep = copysignf(1e-6f, at);
        float iw = 1.f / (at + ep);
        float b0 = a0 * iw, b1 = a1 * iw;

        float gb0 = dyx * iw, gb1 = dyy * iw;
        float gbb = gb0 * b0 + gb1 * b1;
        float gp0x = gbb * (p2y - p1y) - gb1 * p2y;
        float gp1x = gbb * (p0y - p2y) + gb0 * p2y;
        float gp2x = gbb * (p1y - p0y) - gb0 * p1y + gb1 * p0y;
        float gp0y = gbb * (p1x - p2x) + gb1 * p2x;
        float gp1y = gbb * (p2x - p0x) - gb0 * p2x;
        float gp2y = gbb * (p0x - p1x) + gb0 * p1x - gb1 * p0x;
        float gp0w = -fx * gp0x - fy * gp0y;
        float gp1w = -fx * gp1x - fy * gp1y;
        float gp2w = -fx * gp2x - fy * gp2y;

        if (ddb && (uint32_t)((uint32_t)all_ddb << 1) != 0u) {
            float dfxdX = xs * iw, dfydY = ys * iw;
            d0 *= dfxdX; d1 *= dfydY; d2 *= dfxdX; d3 *= dfydY;

            float da0dX = p1[1] * p2[3] - p2[1] * p1[3];
            float da1dX = p2[1] * p0[3] - p0[1] * p2[3];
            float da2dX = p0[1] * p1[3] - p1[1] * p0[3];
            float da0dY = p2[0] * p1[3] - p1[0] * p2[3];
            float da1dY = p0[0] * p2[3] - p2[0] * p0[3];
            float da2dY = p1[0] * p0[3] - p0[0] * p1[3];
            float datdX = da0dX + da1dX + da2dX;
            float datdY = da0dY + da1dY + da2dY;

            float x01 = p0[0] - p1[0], x12 = p1[0] - p2[0], x20 = p2[0] - p0[0];
            float y01 = p0[1] - p1[1], y12 = p1[1] - p2[1], y20 = p2[1] - p0[1];
            float w01 = p0[3] - p1[3], w12 = p1[3] - p2[3], w20 = p2[3] - p0[3];

            float a0p1 = fy * p2[0] - fx * p2[1];
            float a0p2 = fx * p1[1] - fy * p1[0];
            float a1p0 = fx * p2[1] - fy * p2[0];
            float a1p2 = fy * p0[0] - fx * p0[1];

            float wdudX = 2.f * b0 * datdX - da0dX;
            float wdudY = 2.f * b0 * datdY - da0dY;
            float wdvdX = 2.f * b1 * datdX - da1dX;
            float wdvdY = 2.f * b1 * datdY - da1dY;

            float c0  = iw * (d0 * wdudX + d1 * wdudY + d2 * wdvdX + d3 * wdvdY);
            float cx  = c0 * fx - d0 * b0 - d2 * b1;
            float cy  = c0 * fy - d1 * b0 - d3 * b1;
            float cxy = iw * (d0 * datdX + d1 * datdY);
            float czw = iw * (d2 * datdX + d3 * datdY);

            gp0x += c0 * y12 - cy * w12 + czw * p2y + d3 * p2[3];
            gp1x += c0 * y20 - cy * w20 - cxy * p2y - d1 * p2[3];
            gp2x += c0 * y01 - cy * w01 + cxy * p1y - czw * p0y + d1 * p1[3] - d3 * p0[3];
            gp0y += cx * w12 - c0 * x12 - czw * p2x - d2 * p2[3];
            gp1y += cx * w20 - c0 * x20 + cxy * p2x + d0 * p2[3];
            gp2y += cx * w01 - c0 * x01 - cxy * p1x + czw * p0x - d0 * p1[3] + d2 * p0[3];
            gp0w += cy * x12 - cx * y12 - czw * a1p0 + d2 * p2[1] - d3 * p2[0];
            gp1w += cy * x20 - cx * y20 - cxy * a0p1 - d0 * p2[1] + d1 * p2[0];
            gp2w += cy * x01 - cx * y01 - cxy * a0p2 - czw * a1p2 + d0 * p1[1] - d1 * p1[0] - d2 * p0[1] + d3 * p0[0];
        }

        double* g0 = acc + (vo + vi0) * 4; double* g1 = acc + (vo + vi1) * 4; double* g2 = acc + (vo + vi2) * 4;
        g0[0] += gp0x; g0[1] += gp0y; g0[3] += gp0w;
        g1[0] += gp1x; g1[1] += gp1y; g1[3] += gp1w;
        g2[0] += gp2x; g2[1] += gp2y; g2[3] += gp2w;
    }
    for (size_t i = 0; i < gsz; i++) grad_pos[i] = (float)acc[i];
    free(acc);
    return 0;
}
