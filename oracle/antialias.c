/*
 * oracle/antialias.c -- CPU restatement of the reference antialias op.
 * TEST INFRASTRUCTURE ONLY (see nvdr_oracle.h).  Follows csrc/common/antialias.cu:15-25
 * (rational compare helpers), :111-134,139-160 (edge -> opposite-vertex topology),
 * :165-214 (discontinuity finder), :219-382 (analysis + blend), :387-556 (gradients) and
 * csrc/torch/torch_antialias.cpp:68-241 (xh = W/2, yh = H/2, out = color.clone(), g_color = dy.clone()).
 *
 * Pinned to the reference itself (oracle/_ref, oracle/pinned.py; edges in general position, see DESIGN.md
 * "knife-edge silhouettes").  The reference blends with f32 atomics in
 * a scheduling-dependent order; the oracle sums every pixel's / vertex's contributions in f64 in
 * work-item order (pixel-major, "right" item before "down" item) and rounds once.
 * The topology map is an exact edge -> (first, second distinct opposite vertex) table filled in
 * triangle order, which is what the reference's hash holds for any mesh whose edges are shared by
 * at most two triangles (antialias.cu:82-96).
 */
#include "nvdr_oracle.h"

/* Contraction as nvcc's default (-fmad=true) performs it on the reference's expressions. */
#define PROJ(c, w, half, f)   fmaf((c) * (w), (half), -(f))          /* c * w * half - f        (antialias.cu:307-318) */
#define CROSS(a, b, c, d)     fmaf((a), (b), -((c) * (d)))           /* a * b - c * d           (:321-325,346-348,517) */

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define F32_MAX 3.402823466e+38f

static int f2i(float x) { int32_t i; memcpy(&i, &x, 4); return i; }
static int same_sign(float a, float b) { return (f2i(a) ^ f2i(b)) >= 0; }
static int rational_gt(float n0, float n1, float d0, float d1) { return (n0 * d1 > n1 * d0) == same_sign(d0, d1); }
static int max_idx3(float n0, float n1, float n2, float d0, float d1, float d2)
{
    int g10 = rational_gt(n1, n0, d1, d0);
    int g20 = rational_gt(n2, n0, d2, d0);
    int g21 = rational_gt(n2, n1, d2, d1);
    if (g20 && g21) return 2;
    if (g10) return 1;
    return 0;
}
static int tri_id_of(float x)
{
    if (x <= 16777216.f) return (int)x;
    return f2i(x) - 0x4a800000;
}

/* ---- topology: open-addressing table keyed by the (min,max) vertex pair ------------------ */

typedef struct { uint64_t key; int a, b; } EdgeSlot;     /* a, b = opposite vertex + 1 (0 = none) */
typedef struct { EdgeSlot* s; size_t mask; } EdgeMap;

static size_t edge_hash(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; return (size_t)k; }

static EdgeSlot* edge_slot(EdgeMap* m, uint64_t key, int insert)
{
    size_t i = edge_hash(key) & m->mask;
    for (;;) {
        if (m->s[i].key == key) return &m->s[i];
        if (m->s[i].key == 0) { if (!insert) return NULL; m->s[i].key = key; return &m->s[i]; }
        i = (i + 1) & m->mask;
    }
}

static uint64_t edge_key(int va, int vb)
{
    uint64_t v0 = (uint32_t)(va < vb ? va : vb) + 1u, v1 = (uint32_t)(va < vb ? vb : va) + 1u;
    return v0 | (v1 << 32);
}

static void edge_insert(EdgeMap* m, int va, int vb, int vn)        /* :111-120 + :82-96 */
{
    if (va == vb) return;
    EdgeSlot* e = edge_slot(m, edge_key(va, vb), 1);
    int v = vn + 1;
    if (e->a == 0) e->a = v;
    else if (e->a != v && e->b == 0) e->b = v;
}

static int edge_find(const EdgeMap* m, int va, int vb, int vr)     /* :122-134 */
{
    if (va == vb) return -1;
    EdgeSlot* e = edge_slot((EdgeMap*)m, edge_key(va, vb), 0);
    int x = (e ? e->a : 0) - 1, y = (e ? e->b : 0) - 1;
    if (x == vr) return y;
    if (y == vr) return x;
    return -1;
}

static int build_edge_map(EdgeMap* m, const int32_t* tri, int T)
{
    size_t cap = 64;
    while (cap < (size_t)T * 6) cap <<= 1;
    m->s = (EdgeSlot*)calloc(cap, sizeof(EdgeSlot));
    m->mask = cap - 1;
    if (!m->s) return -1;
    for (int i = 0; i < T; i++) {                                   /* :139-160 (numVertices unchecked: 0x7fffffff) */
        int v0 = tri[i * 3], v1 = tri[i * 3 + 1], v2 = tri[i * 3 + 2];
        if (v0 < 0 || v1 < 0 || v2 < 0) continue;
        if (v0 == v1 || v1 == v2 || v2 == v0) continue;
        edge_insert(m, v1, v2, v0);
        edge_insert(m, v2, v0, v1);
        edge_insert(m, v0, v1, v2);
    }
    return 0;
}

/* ---- analysis of one candidate (pixel, direction) pair: antialias.cu:236-379 -------------- */

typedef struct {
    int hit;            /* an edge was found and alpha computed */
    float alpha;
    int di, tri1, tri;  /* edge index, "edge belongs to the neighbour's triangle", triangle id */
    size_t pixel0, pixel1;
    int px, py;         /* pixel the edge distance is measured from */
} AAItem;

typedef struct {
    const float* rast; const float* pos; const int32_t* tri; const EdgeMap* map;
    int instance, N, V, T, H, W;
    float xh, yh;
} AACfg;

static void swapf(float* a, float* b) { float t = *a; *a = *b; *b = t; }

static void analyze(const AACfg* p, int px, int py, int pz, int d, AAItem* it)
{
    it->hit = 0; it->alpha = 0.f;
    size_t pixel0 = (size_t)px + (size_t)p->W * (py + (size_t)p->H * pz);
    size_t pixel1 = pixel0 + (d ? (size_t)p->W : 1);
    it->pixel0 = pixel0; it->pixel1 = pixel1;
    float z0 = p->rast[pixel0 * 4 + 2], t0 = p->rast[pixel0 * 4 + 3];
    float z1 = p->rast[pixel1 * 4 + 2], t1 = p->rast[pixel1 * 4 + 3];
    int tri0 = tri_id_of(t0) - 1, tri1 = tri_id_of(t1) - 1;
    int tri = (tri0 >= 0) ? tri0 : tri1;
    if (tri0 >= 0 && tri1 >= 0) tri = (z0 < z1) ? tri0 : tri1;
    if (tri == tri1) { px += 1 - d; py += d; }
    if (tri < 0 || tri >= p->T) return;
    int vi0 = p->tri[tri * 3], vi1 = p->tri[tri * 3 + 1], vi2 = p->tri[tri * 3 + 2];
    if (vi0 < 0 || vi0 >= p->V || vi1 < 0 || vi1 >= p->V || vi2 < 0 || vi2 >= p->V) return;
    int op0 = edge_find(p->map, vi2, vi1, vi0);
    int op1 = edge_find(p->map, vi0, vi2, vi1);
    int op2 = edge_find(p->map, vi1, vi0, vi2);
    /* The table is built without knowing V (torch_antialias.cpp:40 sets numVertices = 0x7fffffff), so a triangle with an index
     * >= V puts that index into it as an opposite vertex, and the reference then reads pos[] out of bounds here (:282-300):
     * undefined.  Defined here as "no opposite vertex" (what nvdiffrast_amd does); oracle/pinned.py does not pin such tables. */
    if (op0 >= p->V) op0 = -1;
    if (op1 >= p->V) op1 = -1;
    if (op2 >= p->V) op2 = -1;
    size_t vb = p->instance ? (size_t)pz * p->V : 0;
    const float* P0 = p->pos + (vb + vi0) * 4;
    const float* P1 = p->pos + (vb + vi1) * 4;
    const float* P2 = p->pos + (vb + vi2) * 4;
    const float* O0 = (op0 < 0) ? P0 : p->pos + (vb + op0) * 4;
    const float* O1 = (op1 < 0) ? P1 : p->pos + (vb + op1) * 4;
    const float* O2 = (op2 < 0) ? P2 : p->pos + (vb + op2) * 4;

    float w0 = 1.f / P0[3], w1 = 1.f / P1[3], w2 = 1.f / P2[3];
    float ow0 = 1.f / O0[3], ow1 = 1.f / O1[3], ow2 = 1.f / O2[3];
    float fx = (float)px + .5f - p->xh;
    float fy = (float)py + .5f - p->yh;
    /* a*b - c and a*b - c*d below are written as the fused operations nvcc's default -fmad=true makes of them
     * (fma(a, b, -c), fma(a, b, -(c*d))): whether an edge lying exactly on a pixel boundary gives |alpha| = 0.5
     * (position gradient killed, :541-546) or 0.49999997 depends on it.  Pinned by oracle/_ref's fma build. */
    float x0 = PROJ(P0[0], w0, p->xh, fx), y0 = PROJ(P0[1], w0, p->yh, fy);
    float x1 = PROJ(P1[0], w1, p->xh, fx), y1 = PROJ(P1[1], w1, p->yh, fy);
    float x2 = PROJ(P2[0], w2, p->xh, fx), y2 = PROJ(P2[1], w2, p->yh, fy);
    float ox0 = PROJ(O0[0], ow0, p->xh, fx), oy0 = PROJ(O0[1], ow0, p->yh, fy);
    float ox1 = PROJ(O1[0], ow1, p->xh, fx), oy1 = PROJ(O1[1], ow1, p->yh, fy);
    float ox2 = PROJ(O2[0], ow2, p->xh, fx), oy2 = PROJ(O2[1], ow2, p->yh, fy);

    float bb = CROSS(x1 - x0, y2 - y0, x2 - x0, y1 - y0);
    float a0 = CROSS(x1 - ox0, y2 - oy0, x2 - ox0, y1 - oy0);
    float a1 = CROSS(x2 - ox1, y0 - oy1, x0 - ox1, y2 - oy1);
    float a2 = CROSS(x0 - ox2, y1 - oy2, x1 - ox2, y0 - oy2);
    if (!(same_sign(a0, bb) || same_sign(a1, bb) || same_sign(a2, bb))) return;

    if (d) { swapf(&x0, &y0); swapf(&x1, &y1); swapf(&x2, &y2); }
    float dx0 = x2 - x1, dx1 = x0 - x2, dx2 = x1 - x0;
    float dy0 = y2 - y1, dy1 = y0 - y2, dy2 = y1 - y0;
    float dc = -F32_MAX;
    float ds = (tri == tri0) ? 1.f : -1.f;
    float d0 = ds * CROSS(x1, dy0, y1, dx0);
    float d1 = ds * CROSS(x2, dy1, y2, dx1);
    float d2 = ds * CROSS(x0, dy2, y0, dx2);
    if (same_sign(y1, y2)) { d0 = -F32_MAX; dy0 = 1.f; }
    if (same_sign(y2, y0)) { d1 = -F32_MAX; dy1 = 1.f; }
    if (same_sign(y0, y1)) { d2 = -F32_MAX; dy2 = 1.f; }
    int di = max_idx3(d0, d1, d2, dy0, dy1, dy2);
    if (di == 0 && same_sign(a0, bb) && fabsf(dy0) >= fabsf(dx0)) dc = d0 / dy0;
    if (di == 1 && same_sign(a1, bb) && fabsf(dy1) >= fabsf(dx1)) dc = d1 / dy1;
    if (di == 2 && same_sign(a2, bb) && fabsf(dy2) >= fabsf(dx2)) dc = d2 / dy2;
    const float eps = .0625f;
    if (dc > -eps && dc < 1.f + eps) {
        dc = fminf(fmaxf(dc, 0.f), 1.f);
        it->hit = 1;
        it->alpha = ds * (.5f - dc);
        it->di = di;
        it->tri1 = (ds < 0.f);
        it->tri = tri;
        it->px = px; it->py = py;
    }
}

static void fill_cfg(AACfg* c, const float* rast, const float* pos, const int32_t* tri, const EdgeMap* m,
                     int instance, int N, int V, int T, int H, int W)
{
    c->rast = rast; c->pos = pos; c->tri = tri; c->map = m;
    c->instance = instance; c->N = N; c->V = V; c->T = T; c->H = H; c->W = W;
    c->xh = .5f * (float)W; c->yh = .5f * (float)H;
}

/* Is (px,py)->(right|down) a work item?  antialias.cu:176-195: ids compared as floats, clamped at the border. */
static int is_candidate(const AACfg* p, int px, int py, int pz, int d)
{
    if (d ? (py >= p->H - 1) : (px >= p->W - 1)) return 0;
    size_t pixel0 = (size_t)px + (size_t)p->W * (py + (size_t)p->H * pz);
    size_t pixel1 = pixel0 + (d ? (size_t)p->W : 1);
    return p->rast[pixel0 * 4 + 3] != p->rast[pixel1 * 4 + 3];
}

int nvdro_antialias_fwd(const float* color, const float* rast, const float* pos,
                        const int32_t* tri, int instance_mode,
                        int N, int V, int T, int H, int W, int C, float* out)
{
    EdgeMap m;
    if (build_edge_map(&m, tri, T)) return -3;
    AACfg cfg; fill_cfg(&cfg, rast, pos, tri, &m, instance_mode, N, V, T, H, W);
    size_t P = (size_t)N * H * W;
    double* acc = (double*)calloc(P * C, sizeof(double));
    if (!acc) { free(m.s); return -3; }
    for (int pz = 0; pz < N; pz++)
    for (int py = 0; py < H; py++)
    for (int px = 0; px < W; px++)
    for (int d = 0; d < 2; d++) {
        if (!is_candidate(&cfg, px, py, pz, d)) continue;
        AAItem it; analyze(&cfg, px, py, pz, d, &it);
        if (!it.hit) continue;
        const float* c0 = color + it.pixel0 * C;
        const float* c1 = color + it.pixel1 * C;
        double* o = acc + (it.alpha > 0.f ? it.pixel0 : it.pixel1) * C;
        for (int i = 0; i < C; i++) o[i] += (double)(it.alpha * (c1[i] - c0[i]));          /* :363-371 */
    }
    for (size_t i = 0; i < P * C; i++) out[i] = (float)((double)color[i] + acc[i]);
    free(acc); free(m.s);
    return 0;
}

int nvdro_antialias_grad(const float* color, const float* rast, const float* pos,
                         const int32_t* tri, const float* dy, int instance_mode,
                         int N, int V, int T, int H, int W, int C,
                         float* g_color, float* g_pos)
{
    EdgeMap m;
    if (build_edge_map(&m, tri, T)) return -3;
    AACfg cfg; fill_cfg(&cfg, rast, pos, tri, &m, instance_mode, N, V, T, H, W);
    size_t P = (size_t)N * H * W;
    size_t NV = (size_t)(instance_mode ? N : 1) * V;
    double* gc = (double*)calloc(P * C, sizeof(double));
    double* gp = (double*)calloc(NV * 4, sizeof(double));
    if (!gc || !gp) { free(gc); free(gp); free(m.s); return -3; }

    for (int pz = 0; pz < N; pz++)
    for (int py0 = 0; py0 < H; py0++)
    for (int px0 = 0; px0 < W; px0++)
    for (int d = 0; d < 2; d++) {
        if (!is_candidate(&cfg, px0, py0, pz, d)) continue;
        AAItem it; analyze(&cfg, px0, py0, pz, d, &it);
        if (!it.hit || f2i(it.alpha) == 0) continue;                                         /* :409 tests the BITS of alpha (-0.0 passes) */
        float alpha = it.alpha;
        int di = it.di;
        int px = it.px, py = it.py;
        int t = it.tri;                                                                      /* :423 re-reads the same id */
        const float* pDy = dy + (alpha > 0.f ? it.pixel0 : it.pixel1) * C;
        const float* c0 = color + it.pixel0 * C;
        const float* c1 = color + it.pixel1 * C;
        float dd = 0.f;
        for (int i = 0; i < C; i++) {
            float g = pDy[i];
            if (g != 0.f) {
                dd += g * (c1[i] - c0[i]);
                float v = alpha * g;
                gc[it.pixel0 * C + i] += (double)(-v);
                gc[it.pixel1 * C + i] += (double)v;
            }
        }
        if (dd == 0.f) continue;
        int i1 = (di < 2) ? (di + 1) : 0;
        int i2 = (i1 < 2) ? (i1 + 1) : 0;
        int vi1 = tri[3 * t + i1], vi2 = tri[3 * t + i2];
        if (vi1 < 0 || vi1 >= V || vi2 < 0 || vi2 >= V) continue;
        size_t vb = instance_mode ? (size_t)pz * V : 0;
        float p1x = pos[(vb + vi1) * 4], p1y = pos[(vb + vi1) * 4 + 1], p1w = pos[(vb + vi1) * 4 + 3];
        float p2x = pos[(vb + vi2) * 4], p2y = pos[(vb + vi2) * 4 + 1], p2w = pos[(vb + vi2) * 4 + 3];
        float pxh = cfg.xh, pyh = cfg.yh;
        float fx = (float)px + .5f - pxh, fy = (float)py + .5f - pyh;
        if (d) { swapf(&p1x, &p1y); swapf(&p2x, &p2y); swapf(&pxh, &pyh); swapf(&fx, &fy); }
        float w1 = 1.f / p1w, w2 = 1.f / p2w;
        float x1 = PROJ(p1x, w1, pxh, fx), y1 = PROJ(p1y, w1, pyh, fy);
        float x2 = PROJ(p2x, w2, pxh, fx), y2 = PROJ(p2y, w2, pyh, fy);
        float dx = x2 - x1, dyy = y2 - y1;
        float db = CROSS(x1, dyy, y1, dx);
        float ep = copysignf(1e-3f, dyy);
        float iy = 1.f / (dyy + ep);
        float dby = db * iy;
        float iw1 = -w1 * iy * dd, iw2 = w2 * iy * dd;
        float gp1x = iw1 * pxh * y2, gp2x = iw2 * pxh * y1;
        float gp1y = iw1 * pyh * (dby - x2), gp2y = iw2 * pyh * (dby - x1);
        float gp1w = -fmaf(p1x, gp1x, p1y * gp1y) * w1;
        float gp2w = -fmaf(p2x, gp2x, p2y * gp2y) * w2;
        if (d) { swapf(&gp1x, &gp1y); swapf(&gp2x, &gp2y); }
        if (fabsf(alpha) >= 0.5f) { gp1x = gp1y = gp1w = 0.f; gp2x = gp2y = gp2w = 0.f; }
        double* q1 = gp + (vb + vi1) * 4; q1[0] += gp1x; q1[1] += gp1y; q1[3] += gp1w;
        double* q2 = gp + (vb + vi2) * 4; q2[0] += gp2x; q2[1] += gp2y; q2[3] += gp2w;
    }
    for (size_t i = 0; i < P * C; i++) g_color[i] = (float)((double)dy[i] + gc[i]);
    for (size_t i = 0; i < NV * 4; i++) g_pos[i] = (float)gp[i];
    free(gc); free(gp); free(m.s);
    return 0;
}
