/*
 * oracle/interp.c -- CPU restatement of the reference attribute interpolation.
 * TEST INFRASTRUCTURE ONLY (see nvdr_oracle.h).  Follows csrc/common/interpolate.cu.
 * Gradient sums are accumulated in f64 in a fixed pixel order.
 */
#include "nvdr_oracle.h"

#include <stdlib.h>
#include <string.h>

static int tri_id_of(float x)            /* common.h:192 float_to_triidx */
{
    if (x <= 16777216.f) return (int)x;
    int32_t i; memcpy(&i, &x, 4); return i - 0x4a800000;
}

static int resolve_diff_attr(int i, int diff_all, const int32_t* list, int A)
{
    int j = diff_all ? i : list[i];
    if (j < 0) j += A;                    /* interpolate.cu:102-103 python-style wrap */
    return (j >= 0 && j < A) ? j : -1;
}

/* interpolate.cu:15-126 */
int nvdro_interpolate_fwd(const float* attr, const float* rast, const int32_t* tri,
                          const float* rast_db, int attr_instance, int Nattr,
                          int N, int V, int A, int T, int H, int W,
                          int diff_all, const int32_t* diff_attrs, int num_diff,
                          float* out, float* out_da)
{
    int D = diff_all ? A : num_diff;
    int bc = attr_instance && Nattr == 1;                 /* torch_interpolate.cpp:98 */
    size_t P = (size_t)N * H * W;
    size_t HW = (size_t)H * W;

#pragma omp parallel for schedule(static)
    for (long long pi = 0; pi < (long long)P; pi++) {
        int n = (int)((size_t)pi / HW);
        const float* r = rast + (size_t)pi * 4;
        float* o = out + (size_t)pi * A;
        float* oda = (D > 0 && out_da) ? out_da + (size_t)pi * D * 2 : NULL;
        int ti = tri_id_of(r[3]) - 1;
        int valid = (ti >= 0 && ti < T);
        int vi0 = 0, vi1 = 0, vi2 = 0;
        if (valid) { vi0 = tri[ti * 3 + 0]; vi1 = tri[ti * 3 + 1]; vi2 = tri[ti * 3 + 2]; }
        int bad = (vi0 < 0 || vi0 >= V || vi1 < 0 || vi1 >= V || vi2 < 0 || vi2 >= V);
        if (!valid || bad) {
            /* The reference zero-fills when the whole warp is empty and otherwise multiplies
             * vertex-0 attributes by zero barycentrics (:39-80); identical unless attributes
             * hold inf/NaN.  Corrupt indices leave torch::empty memory; oracle writes zeros. */
            for (int i = 0; i < A; i++) o[i] = 0.f;
            if (oda) for (int i = 0; i < D * 2; i++) oda[i] = 0.f;
            continue;
        }
        size_t vo = (attr_instance && !bc) ? (size_t)n * V : 0;
        const float* a0 = attr + (vo + vi0) * A;
        const float* a1 = attr + (vo + vi1) * A;
        const float* a2 = attr + (vo + vi2) * A;
        float b0 = r[0], b1 = r[1], b2 = 1.f - r[0] - r[1];
        for (int i = 0; i < A; i++)
            o[i] = b0 * a0[i] + b1 * a1[i] + b2 * a2[i];
        if (!oda) continue;
        const float* db = rast_db + (size_t)pi * 4;
        float dudx = db[0], dudy = db[1], dvdx = db[2], dvdy = db[3];
        for (int i = 0; i < D; i++) {
            int j = resolve_diff_attr(i, diff_all, diff_attrs, A);
            float dsdx = 0.f, dsdy = 0.f;
            if (j >= 0) {
                float dsdu = a0[j] - a2[j], dsdv = a1[j] - a2[j];
                dsdx = dudx * dsdu + dvdx * dsdv;
                dsdy = dudy * dsdu + dvdy * dsdv;
            }
            oda[i * 2 + 0] = dsdx; oda[i * 2 + 1] = dsdy;
        }
    }
    return 0;
}

/* interpolate.cu:131-274 */
int nvdro_interpolate_grad(const float* attr, const float* rast, const int32_t* tri,
                           const float* dy, const float* rast_db, const float* dda,
                           int attr_instance, int Nattr,
                           int N, int V, int A, int T, int H, int W,
                           int diff_all, const int32_t* diff_attrs, int num_diff,
                           float* g_attr, float* g_rast, float* g_rast_db)
{
    int D = (rast_db && dda) ? (diff_all ? A : num_diff) : 0;
    int na = attr_instance ? Nattr : 1;
    int bc = attr_instance && Nattr < N;                  /* torch_interpolate.cpp:207 */
    int per_image = attr_instance && !bc;
    size_t gsz = (size_t)na * V * A;
    double* acc = (double*)calloc(gsz, sizeof(double));
    if (!acc) return 1;

#pragma omp parallel for schedule(dynamic, 1) if (per_image)
    for (int n = 0; n < N; n++)
    for (size_t q = 0; q < (size_t)H * W; q++) {
        size_t pi = (size_t)n * H * W + q;
        const float* r = rast + pi * 4;
        float* gr = g_rast + pi * 4;
        float* grdb = (D > 0 && g_rast_db) ? g_rast_db + pi * 4 : NULL;
        int ti = tri_id_of(r[3]) - 1;
        if (ti < 0 || ti >= T) {
            gr[0] = gr[1] = gr[2] = gr[3] = 0.f;
            if (grdb) grdb[0] = grdb[1] = grdb[2] = grdb[3] = 0.f;
            continue;
        }
        int vi0 = tri[ti * 3 + 0], vi1 = tri[ti * 3 + 1], vi2 = tri[ti * 3 + 2];
        if (vi0 < 0 || vi0 >= V || vi1 < 0 || vi1 >= V || vi2 < 0 || vi2 >= V) {
            gr[0] = gr[1] = gr[2] = gr[3] = 0.f;        /* reference: untouched torch::empty */
            if (grdb) grdb[0] = grdb[1] = grdb[2] = grdb[3] = 0.f;
            continue;
        }
        size_t vo = per_image ? (size_t)n * V : 0;
        const float* a0 = attr + (vo + vi0) * A; const float* a1 = attr + (vo + vi1) * A; const float* a2 = attr + (vo + vi2) * A;
        double* ga0 = acc + (vo + vi0) * A; double* ga1 = acc + (vo + vi1) * A; double* ga2 = acc + (vo + vi2) * A;
        const float* pdy = dy + pi * A;
        float b0 = r[0], b1 = r[1], b2 = 1.f - r[0] - r[1];
        float gb0 = 0.f, gb1 = 0.f;
        for (int i = 0; i < A; i++) {
            float y = pdy[i];
            gb0 += y * (a0[i] - a2[i]);
            gb1 += y * (a1[i] - a2[i]);
            ga0[i] += (double)(b0 * y); ga1[i] += (double)(b1 * y); ga2[i] += (double)(b2 * y);
        }
        gr[0] = gb0; gr[1] = gb1; gr[2] = 0.f; gr[3] = 0.f;
        if (D <= 0) continue;

        const float* pdda = dda + pi * D * 2;
        const float* db = rast_db + pi * 4;
        float dudx = db[0], dudy = db[1], dvdx = db[2], dvdy = db[3];
        float gdudx = 0.f, gdudy = 0.f, gdvdx = 0.f, gdvdy = 0.f;
        for (int i = 0; i < D; i++) {
            int j = resolve_diff_attr(i, diff_all, diff_attrs, A);
            if (j < 0) continue;
            float dsdx = pdda[i * 2 + 0], dsdy = pdda[i * 2 + 1];
            float dsdu = a0[j] - a2[j], dsdv = a1[j] - a2[j];
            gdudx += dsdu * dsdx; gdudy += dsdu * dsdy;
            gdvdx += dsdv * dsdx; gdvdy += dsdv * dsdy;
            float du = dsdx * dudx + dsdy * dudy;
            float dv = dsdx * dvdx + dsdy * dvdy;
            ga0[j] += (double)du; ga1[j] += (double)dv; ga2[j] += (double)(-du - dv);
        }
        if (grdb) { grdb[0] = gdudx; grdb[1] = gdudy; grdb[2] = gdvdx; grdb[3] = gdvdy; }
    }
    for (size_t i = 0; i < gsz; i++) g_attr[i] = (float)acc[i];
    free(acc);
    return 0;
}
