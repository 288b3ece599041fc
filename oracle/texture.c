/*
 * oracle/texture.c -- CPU restatement of the reference texture op (2D textures).
 * TEST INFRASTRUCTURE ONLY (see nvdr_oracle.h).  Follows csrc/common/texture.cpp:62-102 (mip
 * geometry), csrc/common/texture_kernel.cu:322-585 (texel indexing, mip level selection),
 * :644-699 (mip build), :709-800 (forward), :843-895 (mip gradient pull), :905-1140 (backward)
 * and the glue semantics of csrc/torch/torch_texture.cpp.
 *
 * Pinned to the reference itself: tests run every call also through oracle/_ref (the reference's own
 * texture_kernel.cu / texture.cpp / torch_texture.cpp compiled for the host) and require agreement
 * (oracle/pinned.py, tests/test_ref_pins_oracle.py).  Gradient sums are
 * accumulated in f64 in pixel order.  The one fused multiply-add written out below
 * (texel-space coordinate u*w - 0.5) is where nvcc contracts by default; the HIP kernels use the
 * same explicit fma so both sides agree to the last bit on texel weights.
 * Cube maps (texture_kernel.cu:31-317) are restated from the geometry rather than from the
 * reference's bit tables: the face table below is the OpenGL convention the reference implements
 * (s = sa*ss/(2|c|) + 1/2, t = ta*ts/(2|c|) + 1/2), its gradient functions are the derivatives of that
 * map, and texels beyond a face edge are folded onto the neighbouring face with integer geometry.
 * The texel missing at a cube corner: the reference flags it with index -1 and then adds 6*tz*w*h to all four
 * indices (:431-432).  wrapCubeMap gives the missing texel face -1 and x = y = 0, i.e. index -w*w (:85-88), so for
 * texture slices tz >= 1 the "flag" becomes the valid index 6*tz*w*w - w*w (texel (0,0) of face 5 of the previous
 * slice), which is then sampled with its bilinear weight, and the corner average is lost.  That is reproduced here by default;
 * nvdro_set_cube_corner_fix(1) keeps the flag for every slice instead (opt-in, not reference behaviour).
 * Reciprocals of the major axis are rounded towards zero like the reference's __frcp_rz (:110,136,163,206,264).
 */
#include "nvdr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { F_NEAREST = 0, F_LINEAR = 1, F_LMN = 2, F_LML = 3 };
enum { B_CUBE = 0, B_WRAP = 1, B_CLAMP = 2, B_ZERO = 3 };
#define MAX_LEVELS 17

static int g_cube_corner_fix = 0;
void nvdro_set_cube_corner_fix(int on) { g_cube_corner_fix = on ? 1 : 0; }

/* __frcp_rz: 1/x rounded towards zero.  The double quotient is exact to 53 bits and a float reciprocal that
 * is not exactly representable lies further than that from any float, so truncating the double is exact. */
static float frcp_rz(float x)
{
    double d = 1.0 / (double)x;
    float f = (float)d;
    if (isnan(f)) return f;
    if (isinf(f)) return isinf(d) ? f : copysignf(3.402823466e+38f, f);
    if (fabs((double)f) > fabs(d)) f = nextafterf(f, 0.f);
    return f;
}

static int level_dim(int d, int level) { int v = d >> level; return v > 1 ? v : 1; }   /* texture.h mipLevelSize */

/* texture.cpp:62-102.  Offsets are in floats from the start of the mip buffer (levels >= 1). */
int nvdro_texture_mip_info(int tex_n, int tex_h, int tex_w, int C, int cube, int max_mip_level,
                           int* lvl_w, int* lvl_h, int64_t* lvl_off, int64_t* total_floats)
{
    int w = tex_w, h = tex_h, level = 0;
    int64_t total = 0;
    int c = cube ? C * 6 : C;
    if (lvl_w) lvl_w[0] = w;
    if (lvl_h) lvl_h[0] = h;
    if (lvl_off) lvl_off[0] = -1;
    if (max_mip_level != 0) {
        while ((w | h) > 1) {
            level += 1;
            if ((w > 1 && (w & 1)) || (h > 1 && (h & 1))) return -1;       /* raiseMipSizeError */
            if (w > 1) w >>= 1;
            if (h > 1) h >>= 1;
            if (lvl_w) lvl_w[level] = w;
            if (lvl_h) lvl_h[level] = h;
            if (lvl_off) lvl_off[level] = total;
            total += (int64_t)w * h * tex_n * c;
            if (max_mip_level >= 0 && level == max_mip_level) break;
        }
    }
    if (total_floats) *total_floats = total;
    return level;
}

/* texture_kernel.cu:644-699: 2x2 box filter, 1x2 when one extent of the source level is 1. */
int nvdro_texture_build_mip(const float* tex, int tex_n, int tex_h, int tex_w, int C,
                            int cube, int L, float* mip)
{
    int lw[MAX_LEVELS], lh[MAX_LEVELS];
    int64_t off[MAX_LEVELS], total;
    if (cube) tex_n *= 6;                                                   /* six faces per slice */
    int levels = nvdro_texture_mip_info(tex_n, tex_h, tex_w, C, 0, L, lw, lh, off, &total);
    if (levels < 0) return -1;
    for (int l = 1; l <= levels; l++) {
        const float* in = (l == 1) ? tex : mip + off[l - 1];
        float* out = mip + off[l];
        int wi = lw[l - 1], hi = lh[l - 1], wo = lw[l], ho = lh[l];
        for (int z = 0; z < tex_n; z++)
        for (int y = 0; y < ho; y++)
        for (int x = 0; x < wo; x++)
        for (int c = 0; c < C; c++) {
            float* o = out + (((size_t)z * ho + y) * wo + x) * C + c;
            if (wi == 1 || hi == 1) {
                /* one of the extents is already 1: average the two remaining texels */
                size_t i0 = (hi == 1) ? ((size_t)z * hi * wi + 2 * (size_t)x) : ((size_t)z * hi * wi + 2 * (size_t)y * wi);
                size_t i1 = (hi == 1) ? i0 + 1 : i0 + wi;
                *o = .5f * (in[i0 * C + c] + in[i1 * C + c]);
            } else {
                size_t i0 = ((size_t)z * hi + 2 * (size_t)y) * wi + 2 * (size_t)x;
                float v0 = in[i0 * C + c], v1 = in[(i0 + 1) * C + c];
                float v2 = in[(i0 + wi) * C + c], v3 = in[(i0 + wi + 1) * C + c];
                *o = .25f * (((v0 + v1) + v2) + v3);
            }
        }
    }
    return 0;
}

typedef struct {
    int tex_n, tex_h, tex_w, C, filter, boundary, level_max;
} TexCfg;

/* texture_kernel.cu:322-366 */
static int64_t index_nearest(const TexCfg* t, float u, float v, int tz)
{
    int w = t->tex_w, h = t->tex_h;
    if (t->boundary == B_WRAP) { u = u - floorf(u); v = v - floorf(v); }
    u = u * (float)w;
    v = v * (float)h;
    int iu = (int)floorf(u), iv = (int)floorf(v);
    if (t->boundary == B_ZERO && (iu < 0 || iu >= w || iv < 0 || iv >= h)) return -1;
    iu = iu < 0 ? 0 : (iu > w - 1 ? w - 1 : iu);
    iv = iv < 0 ? 0 : (iv > h - 1 ? h - 1 : iv);
    return (int64_t)iu + (int64_t)w * (iv + (int64_t)tz * h);
}

/* texture_kernel.cu:368-472; tc = texel indices (x0y0, x1y0, x0y1, x1y1) or -1; returns the weights. */
static void index_linear(const TexCfg* t, float u, float v, int tz, int level, int64_t tc[4], float* fu, float* fv)
{
    int w = level_dim(t->tex_w, level), h = level_dim(t->tex_h, level);
    int clampU = 0, clampV = 0;
    if (t->boundary == B_WRAP) { u = u - floorf(u); v = v - floorf(v); }
    u = fmaf(u, (float)w, -0.5f);
    v = fmaf(v, (float)h, -0.5f);
    if (t->boundary == B_CLAMP) {
        u = fminf(fmaxf(u, 0.f), (float)w - 1.f);
        v = fminf(fmaxf(v, 0.f), (float)h - 1.f);
        clampU = (u == 0.f || u == (float)w - 1.f);
        clampV = (v == 0.f || v == (float)h - 1.f);
    }
    int iu0 = (int)floorf(u), iv0 = (int)floorf(v);
    int iu1 = iu0 + (clampU ? 0 : 1), iv1 = iv0 + (clampV ? 0 : 1);
    u -= (float)iu0;
    v -= (float)iv0;
    if (t->boundary == B_WRAP) {
        if (iu0 < 0) iu0 += w;
        if (iv0 < 0) iv0 += h;
        if (iu1 >= w) iu1 -= w;
        if (iv1 >= h) iv1 -= h;
    }
    int64_t base = (int64_t)tz * w * h;
    tc[0] = base + iu0 + (int64_t)w * iv0;
    tc[1] = base + iu1 + (int64_t)w * iv0;
    tc[2] = base + iu0 + (int64_t)w * iv1;
    tc[3] = base + iu1 + (int64_t)w * iv1;
    if (t->boundary == B_ZERO) {
        int u0o = (iu0 < 0 || iu0 >= w), u1o = (iu1 < 0 || iu1 >= w);
        int v0o = (iv0 < 0 || iv0 >= h), v1o = (iv1 < 0 || iv1 >= h);
        if (u0o || v0o) tc[0] = -1;
        if (u1o || v0o) tc[1] = -1;
        if (u0o || v1o) tc[2] = -1;
        if (u1o || v1o) tc[3] = -1;
    }
    *fu = u; *fv = v;
}

static int finite4(const float* a) { return isfinite(a[0]) && isfinite(a[1]) && isfinite(a[2]) && isfinite(a[3]); }

/* ---- cube maps ---------------------------------------------------------------------------- */

/* Face f: major axis ma (sign msgn), s = ss * v[sa] / (2|c|) + 1/2, t = ts * v[ta] / (2|c|) + 1/2.
 * Faces +x -x +y -y +z -z as in texture_kernel.cu:87-110. */
typedef struct { int ma, msgn, sa, ss, ta, ts; } CubeFace;
static const CubeFace kFace[6] = {
    {0, +1, 2, -1, 1, -1}, {0, -1, 2, +1, 1, -1},
    {1, +1, 0, +1, 2, +1}, {1, -1, 0, +1, 2, -1},
    {2, +1, 0, +1, 1, -1}, {2, -1, 0, -1, 1, -1},
};

static int cube_face_of(const float v[3])
{
    float ax = fabsf(v[0]), ay = fabsf(v[1]), az = fabsf(v[2]);
    int f;
    if (az > fmaxf(ax, ay)) f = 4; else if (ay > ax) f = 2; else f = 0;
    if (v[kFace[f].ma] < 0.f) f += 1;
    return f;
}

/* texture_kernel.cu:87-110: (s,t) in [0,1] and the face, or -1 for an invalid direction. */
static int cube_index(const float v[3], float* s, float* t)
{
    int f = cube_face_of(v);
    const CubeFace* F = &kFace[f];
    float m = frcp_rz(fabsf(v[F->ma])) * .5f;
    float x = fmaf(v[F->sa], (float)F->ss * m, .5f);
    float y = fmaf(v[F->ta], (float)F->ts * m, .5f);
    if (!isfinite(x) || !isfinite(y)) return -1;
    *s = fminf(fmaxf(x, 0.f), 1.f);
    *t = fminf(fmaxf(y, 0.f), 1.f);
    return f;
}

/* Texel (ix,iy) of face f at size w, possibly one step outside the face, -> linear texel index
 * x + w*(y + w*face) on the face it really belongs to, or -1 for the texel that does not exist at a
 * cube corner.  Integer geometry in units of half texels: the cube is [-w,w]^3, texel centres sit at
 * odd coordinates, and a texel beyond an edge folds onto the neighbouring face one half-texel inside it. */
static int64_t cube_texel(int f, int ix, int iy, int w)
{
    int ox = (ix < 0 || ix >= w), oy = (iy < 0 || iy >= w);
    if (ox && oy) return -1;
    if (!ox && !oy) return (int64_t)ix + (int64_t)w * (iy + (int64_t)w * f);
    const CubeFace* F = &kFace[f];
    int p[3];
    p[F->ma] = F->msgn * w;
    p[F->sa] = F->ss * (2 * ix + 1 - w);
    p[F->ta] = F->ts * (2 * iy + 1 - w);
    int oa = ox ? F->sa : F->ta;                       /* axis along which we left the face */
    int nsgn = p[oa] > 0 ? 1 : -1;
    p[oa] = nsgn * w;                                  /* new major axis */
    p[F->ma] = F->msgn * (w - 1);                      /* one half-texel inside the new face */
    int nf = oa * 2 + (nsgn < 0 ? 1 : 0);
    const CubeFace* G = &kFace[nf];
    int x = (G->ss * p[G->sa] + w - 1) / 2, y = (G->ts * p[G->ta] + w - 1) / 2;
    return (int64_t)x + (int64_t)w * (y + (int64_t)w * nf);
}

/* Bilinear footprint on a cube level (texture_kernel.cu:382-434): no clamp, no wrap; *corner is set
 * when one of the four texels is the missing corner texel.  Returns 0 for an invalid direction. */
static int index_linear_cube(const TexCfg* t, const float v3[3], int tz, int level, int64_t tc[4], float* fu, float* fv, int* corner)
{
    int w = level_dim(t->tex_w, level);
    float s, tt;
    int f = cube_index(v3, &s, &tt);
    *corner = 0;
    if (f < 0) { tc[0] = tc[1] = tc[2] = tc[3] = -1; *fu = 0.f; *fv = 0.f; return 0; }
    float u = fmaf(s, (float)w, -0.5f), v = fmaf(tt, (float)w, -0.5f);
    int iu0 = (int)floorf(u), iv0 = (int)floorf(v);
    int iu1 = iu0 + 1, iv1 = iv0 + 1;
    *fu = u - (float)iu0; *fv = v - (float)iv0;
    int64_t base = (int64_t)6 * tz * w * w;
    int xs[4] = {iu0, iu1, iu0, iu1}, ys[4] = {iv0, iv0, iv1, iv1};
    for (int k = 0; k < 4; k++) {
        int64_t c = cube_texel(f, xs[k], ys[k], w);
        if (c >= 0) tc[k] = base + c;
        else if (tz > 0 && !g_cube_corner_fix) tc[k] = base - (int64_t)w * w;   /* texture_kernel.cu:85-88,431-432: flag lost */
        else { tc[k] = -1; *corner = 1; }
    }
    return 1;
}

/* dA/d(s,t) -> dA/d(x,y,z) (texture_kernel.cu:113-140). */
static void cube_grad(const float v[3], float gu, float gv, float g[3])
{
    const CubeFace* F = &kFace[cube_face_of(v)];
    float c = v[F->ma];
    float m = frcp_rz(fabsf(c)), h = m * .5f;
    float su = (float)F->ss * gu, sv = (float)F->ts * gv;
    float sg = (c < 0.f) ? 1.f : -1.f;                                     /* -sign(c) */
    g[F->sa] = su * h;
    g[F->ta] = sv * h;
    g[F->ma] = sg * (su * v[F->sa] + sv * v[F->ta]) * m * h;
    if (!isfinite(g[0]) || !isfinite(g[1]) || !isfinite(g[2])) g[0] = g[1] = g[2] = 0.f;
}

/* d(x,y,z)/d(X,Y) -> (ds/dX, ds/dY, dt/dX, dt/dY) (texture_kernel.cu:184-233). */
static void cube_grad_st(const float v[3], const float dX[3], const float dY[3], float r[4])
{
    const CubeFace* F = &kFace[cube_face_of(v)];
    float c = v[F->ma];
    float m = frcp_rz(fabsf(c)), h = m * .5f;
    float k = ((c < 0.f) ? -1.f : 1.f) * m * h;                            /* sign(c) / (2 c^2) */
    float ss = (float)F->ss, ts = (float)F->ts, a = v[F->sa], b = v[F->ta];
    r[0] = ss * (h * dX[F->sa] - k * a * dX[F->ma]);
    r[1] = ss * (h * dY[F->sa] - k * a * dY[F->ma]);
    r[2] = ts * (h * dX[F->ta] - k * b * dX[F->ma]);
    r[3] = ts * (h * dY[F->ta] - k * b * dY[F->ma]);
    if (!finite4(r)) r[0] = r[1] = r[2] = r[3] = 0.f;
}

/* d(ds/dX, ds/dY, dt/dX, dt/dY)/d(x,y,z): J[axis][component] (texture_kernel.cu:235-317). */
static void cube_grad2(const float v[3], const float dX[3], const float dY[3], float J[3][4])
{
    const CubeFace* F = &kFace[cube_face_of(v)];
    float c = v[F->ma];
    float m = frcp_rz(fabsf(c)), h = m * .5f;
    float k = ((c < 0.f) ? -1.f : 1.f) * m * h;
    float k2 = 2.f * k / c;                                                /* -dk/dc */
    float ss = (float)F->ss, ts = (float)F->ts, a = v[F->sa], b = v[F->ta];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) J[i][j] = 0.f;
    /* d/da of the s components, d/db of the t components */
    J[F->sa][0] = -ss * k * dX[F->ma];  J[F->sa][1] = -ss * k * dY[F->ma];
    J[F->ta][2] = -ts * k * dX[F->ma];  J[F->ta][3] = -ts * k * dY[F->ma];
    /* d/dc */
    J[F->ma][0] = ss * (-k * dX[F->sa] + k2 * a * dX[F->ma]);
    J[F->ma][1] = ss * (-k * dY[F->sa] + k2 * a * dY[F->ma]);
    J[F->ma][2] = ts * (-k * dX[F->ta] + k2 * b * dX[F->ma]);
    J[F->ma][3] = ts * (-k * dY[F->ta] + k2 * b * dY[F->ma]);
}

/* dL/d(ds/dX, ds/dY, dt/dX, dt/dY) -> dL/d(d(x,y,z)/dX), dL/d(d(x,y,z)/dY) (texture_kernel.cu:142-182). */
static void cube_grad4(const float v[3], const float dw[4], float g0[3], float g1[3])
{
    const CubeFace* F = &kFace[cube_face_of(v)];
    float c = v[F->ma];
    float m = frcp_rz(fabsf(c)), h = m * .5f;
    float k = ((c < 0.f) ? -1.f : 1.f) * m * h;
    float ss = (float)F->ss, ts = (float)F->ts, a = v[F->sa], b = v[F->ta];
    g0[F->sa] = dw[0] * ss * h;  g0[F->ta] = dw[2] * ts * h;  g0[F->ma] = -k * (dw[0] * ss * a + dw[2] * ts * b);
    g1[F->sa] = dw[1] * ss * h;  g1[F->ta] = dw[3] * ts * h;  g1[F->ma] = -k * (dw[1] * ss * a + dw[3] * ts * b);
    int ok = isfinite(g0[0]) && isfinite(g0[1]) && isfinite(g0[2]) && isfinite(g1[0]) && isfinite(g1[1]) && isfinite(g1[2]);
    if (!ok) { g0[0] = g0[1] = g0[2] = 0.f; g1[0] = g1[1] = g1[2] = 0.f; }
}

/* texture_kernel.cu:477-585.  dw (optional) = d flevel / d(ds/dX, ds/dY, dt/dX, dt/dY); dfdv (cube only)
 * = d flevel / d(x,y,z) through the direction-dependent face mapping. */
static void mip_level(const TexCfg* t, const float* uv3, const float* uv_da, const float* bias, size_t pidx,
                      int* level0, int* level1, float* flevel_out, float dw[4], float dfdv[3])
{
    float flevel = 0.f;
    int cube = (t->boundary == B_CUBE);
    *level0 = 0; *level1 = 0;
    if (uv_da) {
        float d[4];
        float dvdX[3] = {0.f, 0.f, 0.f}, dvdY[3] = {0.f, 0.f, 0.f};
        if (cube) {
            const float* q = uv_da + pidx * 6;
            dvdX[0] = q[0]; dvdY[0] = q[1]; dvdX[1] = q[2]; dvdY[1] = q[3]; dvdX[2] = q[4]; dvdY[2] = q[5];
            cube_grad_st(uv3, dvdX, dvdY, d);
        } else {
            for (int i = 0; i < 4; i++) d[i] = uv_da[pidx * 4 + i];
        }
        float uscl = (float)t->tex_w, vscl = (float)t->tex_h;
        float dsdx = d[0] * uscl, dsdy = d[1] * uscl, dtdx = d[2] * vscl, dtdy = d[3] * vscl;
        float A = dsdx * dsdx + dtdx * dtdx;
        float B = dsdy * dsdy + dtdy * dtdy;
        float Cc = dsdx * dsdy + dtdx * dtdy;
        float l2b = 0.5f * (A + B);
        float l2n = 0.25f * (A - B) * (A - B) + Cc * Cc;
        float l2a = sqrtf(l2n);
        float lenMajorSqr = l2b + l2a;
        if (dw && t->filter == F_LML) {
            float k = 0.72134752f / (l2n + l2a * l2b);
            float AB = k * .5f * (A - B);
            float Cw = k * Cc;
            float l2aw = k * l2a;
            float g[4];
            g[0] = uscl * (dsdx * (l2aw + AB) + dsdy * Cw);
            g[1] = uscl * (dsdy * (l2aw - AB) + dsdx * Cw);
            g[2] = vscl * (dtdx * (l2aw + AB) + dtdy * Cw);
            g[3] = vscl * (dtdy * (l2aw - AB) + dtdx * Cw);
            int ok = finite4(g);
            if (cube) {
                float J[3][4], fv[3];
                cube_grad2(uv3, dvdX, dvdY, J);
                for (int ax = 0; ax < 3; ax++) fv[ax] = ((J[ax][0] * g[0] + J[ax][1] * g[1]) + J[ax][2] * g[2]) + J[ax][3] * g[3];
                ok = ok && isfinite(fv[0]) && isfinite(fv[1]) && isfinite(fv[2]);
                for (int ax = 0; ax < 3; ax++) dfdv[ax] = ok ? fv[ax] : 0.f;
            }
            for (int i = 0; i < 4; i++) dw[i] = ok ? g[i] : 0.f;
        }
        flevel = .5f * log2f(lenMajorSqr);           /* reference: __log2f; may be inf/NaN, the clamp fixes it */
    }
    if (bias) flevel += bias[pidx];
    flevel = fminf(fmaxf(flevel, 0.f), (float)t->level_max);
    *level0 = (int)floorf(flevel);
    if (t->filter == F_LML && flevel > 0.f) {
        *level1 = (*level0 + 1 < t->level_max) ? *level0 + 1 : t->level_max;
        flevel -= (float)*level0;
    }
    *flevel_out = flevel;
}

static float lerpf(float a, float b, float c) { return a + c * (b - a); }
static float bilerpf(float a, float b, float c, float d, float fu, float fv) { return lerpf(lerpf(a, b, fu), lerpf(c, d, fu), fv); }

/* texture_kernel.cu:590-614: the four texels of channel c; at a cube corner the missing texel takes
 * the average of the other three. */
static void fetch_quad(const float* p, const int64_t tc[4], int corner, int C, int c, float a[4])
{
    float sum = 0.f;
    for (int k = 0; k < 4; k++) { a[k] = tc[k] >= 0 ? p[tc[k] * C + c] : 0.f; if (corner && tc[k] >= 0) sum += a[k]; }
    if (corner) { float avg = sum * 0.33333333f; for (int k = 0; k < 4; k++) if (tc[k] < 0) a[k] = avg; }
}

/* texture_kernel.cu:616-639: scatter of the four weights; at a cube corner the missing texel's weight
 * is shared by the other three. */
static void accum_quad(double* acc, const int64_t tc[4], int corner, int C, int c, const float wgt[4])
{
    float cb = 0.f;
    if (corner) { for (int k = 0; k < 4; k++) if (tc[k] < 0) cb = wgt[k]; cb *= 0.33333333f; }
    for (int k = 0; k < 4; k++) if (tc[k] >= 0) acc[tc[k] * C + c] += (double)(corner ? wgt[k] + cb : wgt[k]);
}

static int check_cfg(TexCfg* t, int L, int tex_n, int tex_h, int tex_w, int C, int filter, int boundary)
{
    if (filter < 0 || filter > 3 || boundary < 0 || boundary > 3) return -1;
    if (boundary == B_CUBE && tex_h != tex_w) return -1;
    t->tex_n = tex_n; t->tex_h = tex_h; t->tex_w = tex_w; t->C = C; t->filter = filter; t->boundary = boundary;
    t->level_max = (filter == F_LMN || filter == F_LML) ? L : 0;
    return 0;
}

/* Footprint of one level in either addressing mode; `ok` = 0 only for an invalid cube direction. */
static int footprint(const TexCfg* t, const float* uvp, int tz, int level, int64_t tc[4], float* fu, float* fv, int* corner)
{
    if (t->boundary == B_CUBE) return index_linear_cube(t, uvp, tz, level, tc, fu, fv, corner);
    *corner = 0;
    index_linear(t, uvp[0], uvp[1], tz, level, tc, fu, fv);
    return 1;
}

static int64_t nearest_texel(const TexCfg* t, const float* uvp, int tz)
{
    if (t->boundary != B_CUBE) return index_nearest(t, uvp[0], uvp[1], tz);
    float s, tt;
    int f = cube_index(uvp, &s, &tt);               /* :331-338: no wrap, face folded into tz */
    if (f < 0) return -1;
    int w = t->tex_w;
    int iu = (int)floorf(s * (float)w), iv = (int)floorf(tt * (float)w);
    iu = iu < 0 ? 0 : (iu > w - 1 ? w - 1 : iu);
    iv = iv < 0 ? 0 : (iv > w - 1 ? w - 1 : iv);
    return (int64_t)iu + (int64_t)w * (iv + (int64_t)w * (6 * tz + f));
}

/* texture_kernel.cu:709-800.  uv has 2 components per pixel (3 for cube maps), uv_da 4 (6). */
int nvdro_texture_fwd(const float* tex, const float* const* mip_ptrs, int L,
                      const float* uv, const float* uv_da, const float* mip_level_bias,
                      int tex_n, int tex_h, int tex_w, int C,
                      int N, int H, int W, int filter, int boundary, float* out)
{
    TexCfg t;
    int rc = check_cfg(&t, L, tex_n, tex_h, tex_w, C, filter, boundary);
    if (rc) return rc;
    const float* lv[MAX_LEVELS];
    lv[0] = tex;
    for (int i = 1; i <= t.level_max; i++) lv[i] = mip_ptrs[i - 1];
    int mips = (filter == F_LMN || filter == F_LML);
    int uvs = (boundary == B_CUBE) ? 3 : 2;
    size_t HW = (size_t)H * W, P = (size_t)N * HW;

#pragma omp parallel for schedule(static)
    for (long long pi = 0; pi < (long long)P; pi++) {
        size_t pidx = (size_t)pi;
        int pz = (int)(pidx / HW);
        int tz = (tex_n == 1) ? 0 : pz;
        const float* uvp = uv + pidx * uvs;
        float* o = out + pidx * C;
        if (filter == F_NEAREST) {
            int64_t tc = nearest_texel(&t, uvp, tz);
            for (int c = 0; c < C; c++) o[c] = tc >= 0 ? tex[tc * C + c] : 0.f;
            continue;
        }
        int level0 = 0, level1 = 0; float flevel = 0.f;
        if (mips) mip_level(&t, uvp, uv_da, mip_level_bias, pidx, &level0, &level1, &flevel, NULL, NULL);
        int64_t tc0[4], tc1[4]; float fu0, fv0, fu1 = 0.f, fv1 = 0.f; int corner0 = 0, corner1 = 0;
        footprint(&t, uvp, tz, level0, tc0, &fu0, &fv0, &corner0);
        int second = (filter == F_LML && flevel > 0.f);
        if (second) footprint(&t, uvp, tz, level1, tc1, &fu1, &fv1, &corner1);
        for (int c = 0; c < C; c++) {
            float q[4];
            fetch_quad(lv[level0], tc0, corner0, C, c, q);
            float a = bilerpf(q[0], q[1], q[2], q[3], fu0, fv0);
            if (second) {
                fetch_quad(lv[level1], tc1, corner1, C, c, q);
                a = lerpf(a, bilerpf(q[0], q[1], q[2], q[3], fu1, fv1), flevel);
            }
            o[c] = a;
        }
    }
    return 0;
}

/* texture_kernel.cu:905-1140 (+ :843-895 when pull_mip_grads).  g_tex / g_mip_ptrs are fully
 * overwritten.  Outputs that the mode does not produce may be NULL. */
int nvdro_texture_grad(const float* tex, const float* const* mip_ptrs, int L,
                       const float* uv, const float* uv_da, const float* mip_level_bias,
                       const float* dy,
                       int tex_n, int tex_h, int tex_w, int C,
                       int N, int H, int W, int filter, int boundary,
                       int pull_mip_grads,
                       float* g_tex, float* const* g_mip_ptrs,
                       float* g_uv, float* g_uv_da, float* g_mip_level_bias)
{
    TexCfg t;
    int rc = check_cfg(&t, L, tex_n, tex_h, tex_w, C, filter, boundary);
    if (rc) return rc;
    int mips = (filter == F_LMN || filter == F_LML);
    int cube = (boundary == B_CUBE);
    int uvs = cube ? 3 : 2, das = cube ? 6 : 4;
    int slices = cube ? tex_n * 6 : tex_n;
    const float* lv[MAX_LEVELS];
    double* acc[MAX_LEVELS];
    size_t cnt[MAX_LEVELS];
    lv[0] = tex;
    for (int i = 0; i <= t.level_max; i++) {
        if (i > 0) lv[i] = mip_ptrs[i - 1];
        cnt[i] = (size_t)slices * level_dim(tex_h, i) * level_dim(tex_w, i) * C;
        acc[i] = (double*)calloc(cnt[i], sizeof(double));
        if (!acc[i]) return -3;
    }
    size_t HW = (size_t)H * W, P = (size_t)N * HW;

    for (size_t pidx = 0; pidx < P; pidx++) {
        int pz = (int)(pidx / HW);
        int tz = (tex_n == 1) ? 0 : pz;
        const float* pdy = dy + pidx * C;
        const float* uvp = uv + pidx * uvs;
        uint32_t dmax = 0;
        for (int c = 0; c < C; c++) { uint32_t b; memcpy(&b, &pdy[c], 4); dmax |= b; }
        float dm; memcpy(&dm, &dmax, 4);
        if (dm == 0.f) {                                            /* :922-971 */
            if (filter != F_NEAREST && g_uv) for (int i = 0; i < uvs; i++) g_uv[pidx * uvs + i] = 0.f;
            if (filter == F_LML) {
                if (g_uv_da) for (int i = 0; i < das; i++) g_uv_da[pidx * das + i] = 0.f;
                if (g_mip_level_bias) g_mip_level_bias[pidx] = 0.f;
            }
            continue;
        }
        if (filter == F_NEAREST) {
            int64_t tc = nearest_texel(&t, uvp, tz);
            if (tc >= 0) for (int c = 0; c < C; c++) acc[0][tc * C + c] += (double)pdy[c];
            continue;
        }
        int level0 = 0, level1 = 0; float flevel = 0.f;
        float dw[4] = {0.f, 0.f, 0.f, 0.f}, dfdv[3] = {0.f, 0.f, 0.f};
        if (mips) mip_level(&t, uvp, uv_da, mip_level_bias, pidx, &level0, &level1, &flevel, dw, dfdv);
        int64_t tc0[4], tc1[4]; float fu0, fv0, fu1 = 0.f, fv1 = 0.f; int corner0 = 0, corner1 = 0;
        footprint(&t, uvp, tz, level0, tc0, &fu0, &fv0, &corner0);
        float w011 = fu0 * fv0, w010 = fu0 - w011, w001 = fv0 - w011, w000 = 1.f - fu0 - w001;
        float tw0[4] = {w000, w010, w001, w011};
        float sclu0 = (float)level_dim(tex_w, level0), sclv0 = (float)level_dim(tex_h, level0);
        float gu = 0.f, gv = 0.f, df = 0.f;
        int trilinear = (filter == F_LML);
        float tw1[4] = {0.f, 0.f, 0.f, 0.f}, sclu1 = 0.f, sclv1 = 0.f;
        if (trilinear) {
            footprint(&t, uvp, tz, level1, tc1, &fu1, &fv1, &corner1);
            float w111 = fu1 * fv1, w110 = fu1 - w111, w101 = fv1 - w111, w100 = 1.f - fu1 - w101;
            tw1[0] = w100; tw1[1] = w110; tw1[2] = w101; tw1[3] = w111;
            sclu1 = (float)level_dim(tex_w, level1); sclv1 = (float)level_dim(tex_h, level1);
        }
        for (int c = 0; c < C; c++) {
            float d = pdy[c];
            float d0 = trilinear ? (1.f - flevel) * d : d;
            float wq[4] = {tw0[0] * d0, tw0[1] * d0, tw0[2] * d0, tw0[3] * d0};
            accum_quad(acc[level0], tc0, corner0, C, c, wq);
            float a[4];
            fetch_quad(lv[level0], tc0, corner0, C, c, a);
            float ad = (a[3] + a[0] - a[1] - a[2]);
            gu += d0 * ((a[1] - a[0]) + fv0 * ad) * sclu0;
            gv += d0 * ((a[2] - a[0]) + fu0 * ad) * sclv0;
            if (trilinear && flevel > 0.f) {
                float d1 = flevel * d;
                float wq1[4] = {tw1[0] * d1, tw1[1] * d1, tw1[2] * d1, tw1[3] * d1};
                accum_quad(acc[level1], tc1, corner1, C, c, wq1);
                float b[4];
                fetch_quad(lv[level1], tc1, corner1, C, c, b);
                float bd = (b[3] + b[0] - b[1] - b[2]);
                gu += d1 * ((b[1] - b[0]) + fv1 * bd) * sclu1;
                gv += d1 * ((b[2] - b[0]) + fu1 * bd) * sclv1;
                df += (bilerpf(b[0], b[1], b[2], b[3], fu1, fv1) - bilerpf(a[0], a[1], a[2], a[3], fu0, fv0)) * d;
            }
        }
        if (g_uv) {
            if (cube) {
                float g3[3];
                cube_grad(uvp, gu, gv, g3);
                for (int i = 0; i < 3; i++) g_uv[pidx * 3 + i] = trilinear ? g3[i] + dfdv[i] * df : g3[i];
            } else { g_uv[pidx * 2] = gu; g_uv[pidx * 2 + 1] = gv; }
        }
        if (trilinear) {
            if (g_mip_level_bias) g_mip_level_bias[pidx] = df;
            if (uv_da && g_uv_da) {
                float dwf[4] = {dw[0] * df, dw[1] * df, dw[2] * df, dw[3] * df};
                if (cube) {
                    float g0[3], g1[3];
                    cube_grad4(uvp, dwf, g0, g1);
                    for (int i = 0; i < 3; i++) { g_uv_da[pidx * 6 + 2 * i] = g0[i]; g_uv_da[pidx * 6 + 2 * i + 1] = g1[i]; }
                } else for (int i = 0; i < 4; i++) g_uv_da[pidx * 4 + i] = dwf[i];
            }
        }
    }

    /* MipGradKernel (:843-895): every base texel pulls its ancestors' gradients, weight 1/4 per
     * level (1/2 when the level below had an extent of 1). */
    if (pull_mip_grads && t.level_max > 0) {
        for (int z = 0; z < slices; z++)
        for (int y = 0; y < tex_h; y++)
        for (int x = 0; x < tex_w; x++) {
            int xx = x, yy = y; double wgt = 1.0;
            int pw = tex_w, ph = tex_h;
            for (int l = 1; l <= t.level_max; l++) {
                if (pw > 1) wgt *= .5;
                if (ph > 1) wgt *= .5;
                pw = level_dim(tex_w, l); ph = level_dim(tex_h, l);
                xx >>= 1; yy >>= 1;
                const double* src = acc[l] + (((size_t)z * ph + yy) * pw + xx) * C;
                double* dst = acc[0] + (((size_t)z * tex_h + y) * tex_w + x) * C;
                for (int c = 0; c < C; c++) dst[c] += src[c] * wgt;
            }
        }
    }
    for (size_t i = 0; i < cnt[0]; i++) g_tex[i] = (float)acc[0][i];
    for (int l = 1; l <= t.level_max; l++)
        if (g_mip_ptrs && g_mip_ptrs[l - 1]) for (size_t i = 0; i < cnt[l]; i++) g_mip_ptrs[l - 1][i] = (float)acc[l][i];
    for (int i = 0; i <= t.level_max; i++) free(acc[i]);
    return 0;
}

/* Test hooks for the cube-map helpers (tests/test_oracle_texture_aa.py). */
long long nvdro_cube_texel(int face, int ix, int iy, int w) { return (long long)cube_texel(face, ix, iy, w); }
int nvdro_cube_index(const float* v, float* s, float* t) { return cube_index(v, s, t); }
