// nvdr_raster_tape.hpp -- per-pixel gradient of the rasterizer's pixel shader w.r.t. the clip-space positions of the
// pixel's three vertices (rasterize.cu:119-277), shared by k_raster_grad (raster.hip) and the fused backward kernel
// (backward_fused.hip).
#pragma once

#include "nvdr_device.hpp"

namespace nvdr {

// Reverse-mode differentiation of the pixel shader (k_fine: barycentrics from the edge functions of the pixel-relative
// vertices, rasterize.cu:63-113), written as a tape: forward values first, then adjoints propagated output -> input.
// Semantics of rasterize.cu:119-277: the clamps of the forward pass are ignored and 1/at is regularised with a signed 1e-6.
//   P        the three vertices (x, y, z, w);  (fx, fy) the pixel centre in NDC;  (xs, ys) = (2/W, 2/H)
//   dyx, dyy upstream gradients of the barycentrics (u, v);  ddb of (du/dX, du/dY, dv/dX, dv/dY), used when use_db
//   g        out: d/d(x, y, w) of vertex 0, 1, 2
template <bool ENABLE_DB>
__device__ __forceinline__ void raster_tape(const float4 (&P)[3], float fx, float fy, float xs, float ys,
                                            float dyx, float dyy, float4 ddb, bool use_db, float (&g)[9])
{
    float X[3], Y[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { X[k] = P[k].x - fx * P[k].w; Y[k] = P[k].y - fy * P[k].w; }
    float a[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { const int i = (k + 1) % 3, j = (k + 2) % 3; a[k] = X[i] * Y[j] - Y[i] * X[j]; }
    const float at = a[0] + a[1] + a[2];
    const float iw = 1.f / (at + copysignf(1e-6f, at));
    const float b0 = a[0] * iw, b1 = a[1] * iw;

    float gb0 = dyx, gb1 = dyy;            // adjoints of the barycentrics
    float giw = 0.f;                       // adjoint of iw from everything except b0, b1
    float gx[3] = {0.f, 0.f, 0.f}, gyv[3] = {0.f, 0.f, 0.f}, gw[3] = {0.f, 0.f, 0.f};    // adjoints of the raw x, y, w
    if (ENABLE_DB && use_db) {
        // rast_db = (sx*(b0*DtX - D0X), sy*(b0*DtY - D0Y), sx*(b1*DtX - D1X), sy*(b1*DtY - D1Y)) with
        // sx = xs*iw, sy = ys*iw, D_kX = y_j w_i - y_i w_j, D_kY = x_i w_j - x_j w_i (i = k+1, j = k+2), Dt = sum.
        float DX[3], DY[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int i = (k + 1) % 3, j = (k + 2) % 3;
            DX[k] = P[j].y * P[i].w - P[i].y * P[j].w;
            DY[k] = P[i].x * P[j].w - P[j].x * P[i].w;
        }
        const float DtX = DX[0] + DX[1] + DX[2], DtY = DY[0] + DY[1] + DY[2];
        const float sx = xs * iw, sy = ys * iw;
        // adjoints of T_kX = b_k*DtX - D_kX (k = 0, 1) and of sx, sy
        const float t0x = ddb.x * sx, t0y = ddb.y * sy, t1x = ddb.z * sx, t1y = ddb.w * sy;
        giw = xs * (ddb.x * (b0 * DtX - DX[0]) + ddb.z * (b1 * DtX - DX[1]))
            + ys * (ddb.y * (b0 * DtY - DY[0]) + ddb.w * (b1 * DtY - DY[1]));
        gb0 += t0x * DtX + t0y * DtY;
        gb1 += t1x * DtX + t1y * DtY;
        const float gDtX = t0x * b0 + t1x * b1, gDtY = t0y * b0 + t1y * b1;
        const float gDX[3] = {gDtX - t0x, gDtX - t1x, gDtX};
        const float gDY[3] = {gDtY - t0y, gDtY - t1y, gDtY};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int i = (k + 1) % 3, j = (k + 2) % 3;
            // D_kX = y_j w_i - y_i w_j
            gyv[j] += gDX[k] * P[i].w; gw[i] += gDX[k] * P[j].y;
            gyv[i] -= gDX[k] * P[j].w; gw[j] -= gDX[k] * P[i].y;
            // D_kY = x_i w_j - x_j w_i
            gx[i] += gDY[k] * P[j].w; gw[j] += gDY[k] * P[i].x;
            gx[j] -= gDY[k] * P[i].w; gw[i] -= gDY[k] * P[j].x;
        }
    }
    // b_k = a_k * iw, iw = 1 / at', at = a0 + a1 + a2
    giw += gb0 * a[0] + gb1 * a[1];
    const float gat = -giw * iw * iw;
    const float ga[3] = {gb0 * iw + gat, gb1 * iw + gat, gat};
    // a_k = X_i Y_j - Y_i X_j
    float gX[3] = {0.f, 0.f, 0.f}, gY[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int i = (k + 1) % 3, j = (k + 2) % 3;
        gX[i] += ga[k] * Y[j]; gY[j] += ga[k] * X[i];
        gY[i] -= ga[k] * X[j]; gX[j] -= ga[k] * Y[i];
    }
    // X_k = x_k - fx w_k, Y_k = y_k - fy w_k
#pragma unroll
    for (int k = 0; k < 3; k++) {
        g[k * 3 + 0] = gx[k] + gX[k];
        g[k * 3 + 1] = gyv[k] + gY[k];
        g[k * 3 + 2] = gw[k] - fx * gX[k] - fy * gY[k];
    }
}

}  // namespace nvdr
