/*
 * oracle/texture.c -- CPU restatement of the reference texture op (2D textures).
 * TEST INFRASTRUCTURE ONLY (see nvdr_oracle.h).  Follows csrc/common/texture.cpp:62-102 (mip
 * geometry), csrc/common/texture_kernel.cu:322-585 (texel indexing, mip level selection),
 * :644-699 (mip build), :709-800 (forward), :843-895 (mip gradient pull), :905-1140 (backward)
 * and the glue semantics of csrc/torch/torch_texture.cpp.
 *
 * Parity unpinned: the reference holds no golden vectors for this op.  Gradient sums are
 * accumulated in f64 in pixel order.  The one fused multiply-add written out below
 * (texel-space coordinate u*w - 0.5) is where nvcc contracts by default; the HIP kernels use the
 * same explicit fma so both sides agree to the last bit on texel weights.
 * Cube maps (boundary mode 0) are not restated yet: every entry point returns -2 for them.
 */
#include "nvdr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { F_NEAREST = 0, F_LINEAR = 1, F_LMN = 2, F_LML = 3 };
enum { B_CUBE = 0, B_WRAP = 1, B_CLAMP = 2, B_ZERO = 3 };
#define MAX_LEVELS 17

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
    if (cube) return -2;
    int lw[MAX_LEVELS], lh[MAX_LEVELS];
    int64_t off[MAX_LEVELS], total;
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

/* texture_kernel.cu:477-585.  dw (optional) = d flevel / d uv_da. */
static void mip_level(const TexCfg* t, const float* uv_da, const float* bias, size_t pidx,
                      int* level0, int* level1, float* flevel_out, float dw[4])
{
    float flevel = 0.f;
    *level0 = 0; *level1 = 0;
    if (uv_da) {
        const float* d = uv_da + pidx * 4;
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
static float texel(const float* p, int64_t tc, int C, int c) { return tc >= 0 ? p[tc * C + c] : 0.f; }

static int check_cfg(TexCfg* t, int L, int tex_n, int tex_h, int tex_w, int C, int filter, int boundary)
{
    if (boundary == B_CUBE) return -2;
    if (filter < 0 || filter > 3 || boundary < 0 || boundary > 3) return -1;
    t->tex_n = tex_n; t->tex_h = tex_h; t->tex_w = tex_w; t->C = C; t->filter = filter; t->boundary = boundary;
    t->level_max = (filter == F_LMN || filter == F_LML) ? L : 0;
    return 0;
}

/* texture_kernel.cu:709-800 */
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
    size_t HW = (size_t)H * W, P = (size_t)N * HW;

#pragma omp parallel for schedule(static)
    for (long long pi = 0; pi < (long long)P; pi++) {
        size_t pidx = (size_t)pi;
        int pz = (int)(pidx / HW);
        int tz = (tex_n == 1) ? 0 : pz;
        float u = uv[pidx * 2], v = uv[pidx * 2 + 1];
        float* o = out + pidx * C;
        if (filter == F_NEAREST) {
            int64_t tc = index_nearest(&t, u, v, tz);
            for (int c = 0; c < C; c++) o[c] = texel(tex, tc, C, c);
            continue;
        }
        int level0 = 0, level1 = 0; float flevel = 0.f;
        if (mips) mip_level(&t, uv_da, mip_level_bias, pidx, &level0, &level1, &flevel, NULL);
        int64_t tc0[4], tc1[4]; float fu0, fv0, fu1 = 0.f, fv1 = 0.f;
        index_linear(&t, u, v, tz, level0, tc0, &fu0, &fv0);
        int second = (filter == F_LML && flevel > 0.f);
        if (second) index_linear(&t, u, v, tz, level1, tc1, &fu1, &fv1);
        for (int c = 0; c < C; c++) {
            const float* p0 = lv[level0];
            float a = bilerpf(texel(p0, tc0[0], C, c), texel(p0, tc0[1], C, c), texel(p0, tc0[2], C, c), texel(p0, tc0[3], C, c), fu0, fv0);
            if (second) {
                const float* p1 = lv[level1];
                float b = bilerpf(texel(p1, tc1[0], C, c), texel(p1, tc1[1], C, c), texel(p1, tc1[2], C, c), texel(p1, tc1[3], C, c), fu1, fv1);
                a = lerpf(a, b, flevel);
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
    const float* lv[MAX_LEVELS];
    double* acc[MAX_LEVELS];
    size_t cnt[MAX_LEVELS];
    lv[0] = tex;
    for (int i = 0; i <= t.level_max; i++) {
        if (i > 0) lv[i] = mip_ptrs[i - 1];
        cnt[i] = (size_t)tex_n * level_dim(tex_h, i) * level_dim(tex_w, i) * C;
        acc[i] = (double*)calloc(cnt[i], sizeof(double));
        if (!acc[i]) return -3;
    }
    size_t HW = (size_t)H * W, P = (size_t)N * HW;

    for (size_t pidx = 0; pidx < P; pidx++) {
        int pz = (int)(pidx / HW);
        int tz = (tex_n == 1) ? 0 : pz;
        const float* pdy = dy + pidx * C;
        uint32_t dmax = 0;
        for (int c = 0; c < C; c++) { uint32_t b; memcpy(&b, &pdy[c], 4); dmax |= b; }
        float dm; memcpy(&dm, &dmax, 4);
        if (dm == 0.f) {                                            /* :922-971 */
            if (filter != F_NEAREST && g_uv) { g_uv[pidx * 2] = 0.f; g_uv[pidx * 2 + 1] = 0.f; }
            if (filter == F_LML) {
                if (g_uv_da) for (int i = 0; i < 4; i++) g_uv_da[pidx * 4 + i] = 0.f;
                if (g_mip_level_bias) g_mip_level_bias[pidx] = 0.f;
            }
            continue;
        }
        float u = uv[pidx * 2], v = uv[pidx * 2 + 1];
        if (filter == F_NEAREST) {
            int64_t tc = index_nearest(&t, u, v, tz);
            if (tc >= 0) for (int c = 0; c < C; c++) acc[0][tc * C + c] += (double)pdy[c];
            continue;
        }
        int level0 = 0, level1 = 0; float flevel = 0.f;
        float dw[4] = {0.f, 0.f, 0.f, 0.f};
        if (mips) mip_level(&t, uv_da, mip_level_bias, pidx, &level0, &level1, &flevel, dw);
        int64_t tc0[4], tc1[4]; float fu0, fv0, fu1 = 0.f, fv1 = 0.f;
        index_linear(&t, u, v, tz, level0, tc0, &fu0, &fv0);
        float w011 = fu0 * fv0, w010 = fu0 - w011, w001 = fv0 - w011, w000 = 1.f - fu0 - w001;
        float tw0[4] = {w000, w010, w001, w011};
        float sclu0 = (float)level_dim(tex_w, level0), sclv0 = (float)level_dim(tex_h, level0);
        float gu = 0.f, gv = 0.f, df = 0.f;

        if (filter == F_LINEAR || filter == F_LMN) {
            for (int c = 0; c < C; c++) {
                float d = pdy[c];
                for (int k = 0; k < 4; k++) if (tc0[k] >= 0) acc[level0][tc0[k] * C + c] += (double)(tw0[k] * d);
                const float* p0 = lv[level0];
                float a00 = texel(p0, tc0[0], C, c), a10 = texel(p0, tc0[1], C, c), a01 = texel(p0, tc0[2], C, c), a11 = texel(p0, tc0[3], C, c);
                float ad = (a11 + a00 - a10 - a01);
                gu += d * ((a10 - a00) + fv0 * ad) * sclu0;
                gv += d * ((a01 - a00) + fu0 * ad) * sclv0;
            }
            if (g_uv) { g_uv[pidx * 2] = gu; g_uv[pidx * 2 + 1] = gv; }
            continue;
        }

        /* trilinear */
        index_linear(&t, u, v, tz, level1, tc1, &fu1, &fv1);
        float w111 = fu1 * fv1, w110 = fu1 - w111, w101 = fv1 - w111, w100 = 1.f - fu1 - w101;
        float tw1[4] = {w100, w110, w101, w111};
        float sclu1 = (float)level_dim(tex_w, level1), sclv1 = (float)level_dim(tex_h, level1);
        for (int c = 0; c < C; c++) {
            float d = pdy[c];
            float d0 = (1.f - flevel) * d;
            for (int k = 0; k < 4; k++) if (tc0[k] >= 0) acc[level0][tc0[k] * C + c] += (double)(tw0[k] * d0);
            const float* p0 = lv[level0];
            float a00 = texel(p0, tc0[0], C, c), a10 = texel(p0, tc0[1], C, c), a01 = texel(p0, tc0[2], C, c), a11 = texel(p0, tc0[3], C, c);
            float ad = (a11 + a00 - a10 - a01);
            gu += d0 * ((a10 - a00) + fv0 * ad) * sclu0;
            gv += d0 * ((a01 - a00) + fu0 * ad) * sclv0;
            if (flevel > 0.f) {
                float d1 = flevel * d;
                for (int k = 0; k < 4; k++) if (tc1[k] >= 0) acc[level1][tc1[k] * C + c] += (double)(tw1[k] * d1);
                const float* p1 = lv[level1];
                float b00 = texel(p1, tc1[0], C, c), b10 = texel(p1, tc1[1], C, c), b01 = texel(p1, tc1[2], C, c), b11 = texel(p1, tc1[3], C, c);
                float bd = (b11 + b00 - b10 - b01);
                gu += d1 * ((b10 - b00) + fv1 * bd) * sclu1;
                gv += d1 * ((b01 - b00) + fu1 * bd) * sclv1;
                float a = bilerpf(a00, a10, a01, a11, fu0, fv0);
                float b = bilerpf(b00, b10, b01, b11, fu1, fv1);
                df += (b - a) * d;
            }
        }
        if (g_uv) { g_uv[pidx * 2] = gu; g_uv[pidx * 2 + 1] = gv; }
        if (g_mip_level_bias) g_mip_level_bias[pidx] = df;
        if (uv_da && g_uv_da) for (int i = 0; i < 4; i++) g_uv_da[pidx * 4 + i] = dw[i] * df;
    }

    /* MipGradKernel (:843-895): every base texel pulls its ancestors' gradients, weight 1/4 per
     * level (1/2 when the level below had an extent of 1). */
    if (pull_mip_grads && t.level_max > 0) {
        for (int z = 0; z < tex_n; z++)
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
