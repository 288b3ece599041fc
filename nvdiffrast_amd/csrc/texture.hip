// texture.hip -- texture sampling forward / backward, mip construction for gfx950.
//
// Replaces csrc/common/texture_kernel.cu + csrc/common/texture.cpp + csrc/torch/torch_texture.cpp
// behind the C ABI (2D textures and cube maps).
//
//  k_mip_build   one lane per output texel, 2x2 box filter (texture_kernel.cu:644-699); the small levels at the end
//                of the chain in one launch of one workgroup (k_mip_build_tail).
//  k_tex_fwd     one lane per pixel, a wave = one 8x8 pixel tile so that the 4..8 texel taps of
//                a wave land in a compact texture footprint (L1/L2 hits); uv / uv_da / out are
//                read and written as whole float2 / float4 per lane.  Tiles are handed to the
//                XCDs in contiguous chunks so neighbouring tiles share an L2.
//  k_tex_grad    same mapping; texel weights are accumulated in an LDS table of 8x2-texel patches
//                (32-bit fixed point, 32-bit keys), pixels with identical footprints are merged in
//                registers first, and only the claimed patches are flushed, with line-coalesced
//                hardware f32 atomics; uv / uv_da / bias gradients are written per pixel
//                (texture_kernel.cu:905-1140).
//  k_mip_grad    one lane per 4x4 block of base texels pulls the block's ancestors' gradients (:843-895); rows of the
//                block are accessed as float4 vectors when the channel count allows (k_mip_grad_vec).
#include "nvdr_device.hpp"
#include "nvdr_host.hpp"

namespace nvdr {

constexpr int kTexMaxLevels = 17;                 // texture.h:24 TEX_MAX_MIP_LEVEL (16) + base level
enum { TEX_NEAREST = 0, TEX_LINEAR = 1, TEX_LMN = 2, TEX_LML = 3 };          // ops.py:415
enum { TEX_B_CUBE = 0, TEX_B_WRAP = 1, TEX_B_CLAMP = 2, TEX_B_ZERO = 3 };    // ops.py:417

struct TexParams {
    const float* tex[kTexMaxLevels];
    float*       gradTex[kTexMaxLevels];
    const float* uv; const float* uvDA; const float* bias; const float* dy;
    float* out; float* gradUV; float* gradUVDA; float* gradBias;
    int boundary, channels, imgW, imgH, n, texW, texH, texDepth, levelMax;
    int tilesX, tilesY, dbg;
    int cornerFix;                  // NVDR_OPT_CUBE_CORNER_FIX: keep the cube-corner flag for texture slices >= 1
    // Gradient pass, caller-provided scratch (NULL = none): one record per WAVE of pixels that all sample the same texel
    // quad of level 0 with a zero footprint (k_tex_grad's uniform-wave path): rec[0..3][r] = texel index of each tap
    // (-1: none; rec[0][r] = -1 also marks "no record", the state the host puts the array in before the launch),
    // rec[4..7][r] = the taps' bilinear weights, rec[8..8+C)[r] = the wave's summed upstream gradient per channel.
    // k_tex_grad_fold merges the records (a constant-uv background produces ONE texel quad for hundreds of thousands of
    // waves) and adds the totals to the gradient texture.
    int* rec; int nrec;
    // Tiles (8x8 pixels) in which uv and uv_da are KNOWN to be zero: the rasterizer's occupancy flags, handed on by the
    // operator layer when uv / uv_da are interpolate()'s own, untouched outputs for that rast (nvdr_device.hpp TileFlags;
    // f == nullptr: nothing known).  Pixels of such tiles take uv = 0, uv_da = 0 without reading them.
    TileFlags zflags;
    // Two-kernel gradient pass (k_tex_grad_light first): one byte per 16x16-pixel block, 1 = the block is left to k_tex_grad,
    // 0 = k_tex_grad_light has done all of it (k_tex_grad's workgroup leaves at once).  NULL = single-kernel pass.
    uint8_t* heavy;
};

constexpr int kTexRecHeader = 8;                  // words per record in front of the channel totals

__device__ __forceinline__ int level_dim(int d, int level) { int v = d >> level; return v > 1 ? v : 1; }

// texture_kernel.cu:322-366
__device__ __forceinline__ int tex_index_nearest(const TexParams& p, float u, float v, int tz)
{
#pragma clang fp contract(off)
    const int w = p.texW, h = p.texH;
    if (p.boundary == TEX_B_WRAP) { u = u - floorf(u); v = v - floorf(v); }
    u = u * (float)w;
    v = v * (float)h;
    int iu = __float2int_rd(u), iv = __float2int_rd(v);
    if (p.boundary == TEX_B_ZERO && (iu < 0 || iu >= w || iv < 0 || iv >= h)) return -1;
    iu = min(max(iu, 0), w - 1);
    iv = min(max(iv, 0), h - 1);
    return iu + w * (iv + tz * h);
}

// Bilinear footprint: texel indices of taps x0y0, x1y0, x0y1, x1y1 (-1 = no texel), weights, and each
// tap's texel column / row inside its slice (cube maps: row = y + w * face).  `corner` marks a cube
// corner footprint, whose missing texel stands for the average of the other three.
struct Quad { int tc[4]; float fu, fv; int tx[4], ty[4]; bool corner; };

// texture_kernel.cu:368-472.  The one explicit fma is where the reference's compiler contracts.
__device__ __forceinline__ Quad tex_index_linear(const TexParams& p, float u, float v, int tz, int level)
{
#pragma clang fp contract(off)
    const int w = level_dim(p.texW, level), h = level_dim(p.texH, level);
    bool clampU = false, clampV = false;
    if (p.boundary == TEX_B_WRAP) { u = u - floorf(u); v = v - floorf(v); }
    u = __fmaf_rn(u, (float)w, -0.5f);
    v = __fmaf_rn(v, (float)h, -0.5f);
    if (p.boundary == TEX_B_CLAMP) {
        u = fminf(fmaxf(u, 0.f), (float)w - 1.f);
        v = fminf(fmaxf(v, 0.f), (float)h - 1.f);
        clampU = (u == 0.f || u == (float)w - 1.f);
        clampV = (v == 0.f || v == (float)h - 1.f);
    }
    int iu0 = __float2int_rd(u), iv0 = __float2int_rd(v);
    int iu1 = iu0 + (clampU ? 0 : 1), iv1 = iv0 + (clampV ? 0 : 1);
    Quad q;
    q.fu = u - (float)iu0;
    q.fv = v - (float)iv0;
    if (p.boundary == TEX_B_WRAP) {
        if (iu0 < 0) iu0 += w;
        if (iv0 < 0) iv0 += h;
        if (iu1 >= w) iu1 -= w;
        if (iv1 >= h) iv1 -= h;
    }
    q.tx[0] = iu0; q.tx[1] = iu1; q.tx[2] = iu0; q.tx[3] = iu1;
    q.ty[0] = iv0; q.ty[1] = iv0; q.ty[2] = iv1; q.ty[3] = iv1;
    q.corner = false;
    const int base = tz * w * h;
    q.tc[0] = base + iu0 + w * iv0;
    q.tc[1] = base + iu1 + w * iv0;
    q.tc[2] = base + iu0 + w * iv1;
    q.tc[3] = base + iu1 + w * iv1;
    if (p.boundary == TEX_B_ZERO) {
        const bool u0o = (iu0 < 0 || iu0 >= w), u1o = (iu1 < 0 || iu1 >= w);
        const bool v0o = (iv0 < 0 || iv0 >= h), v1o = (iv1 < 0 || iv1 >= h);
        if (u0o || v0o) q.tc[0] = -1;
        if (u1o || v0o) q.tc[1] = -1;
        if (u0o || v1o) q.tc[2] = -1;
        if (u1o || v1o) q.tc[3] = -1;
    }
    return q;
}

__device__ __forceinline__ bool finite4(float4 a) { return isfinite(a.x) && isfinite(a.y) && isfinite(a.z) && isfinite(a.w); }

// ---- cube maps (texture_kernel.cu:31-317) ----------------------------------------------------
// Stated from the geometry, not from the reference's bit tables: face f has major axis ma (sign
// msgn) and s = ss * v[sa] / (2|c|) + 1/2, t = ts * v[ta] / (2|c|) + 1/2 (the OpenGL convention the
// reference implements, :87-110); the gradient helpers are the derivatives of that map and texels
// beyond a face edge are folded onto the neighbouring face with integer geometry.

// 1/x rounded towards zero for x >= 0, as the reference's __frcp_rz (:110,136,163,206,264): the correctly
// rounded quotient, stepped one ulp down when it lies above the exact value (sign of the exact residual
// r*x - 1 from one fma).  +inf from a denormal x steps down to FLT_MAX, as round-towards-zero overflow does.
__device__ __forceinline__ float rcp_rz(float x)
{
    float r = 1.f / x;
    if (__fmaf_rn(r, x, -1.f) > 0.f) r = __uint_as_float(__float_as_uint(r) - 1u);
    return r;
}

struct CubeFace { int ma, msgn, sa, ss, ta, ts; };
__device__ __forceinline__ CubeFace cube_face(int f)
{
    // +x -x +y -y +z -z
    const int ma = f >> 1;
    const int msgn = (f & 1) ? -1 : 1;
    const int sa = (ma == 0) ? 2 : 0;
    const int ta = (ma == 1) ? 2 : 1;
    const int ss = (f == 0 || f == 5) ? -1 : 1;
    const int ts = (f == 2) ? 1 : -1;
    return CubeFace{ma, msgn, sa, ss, ta, ts};
}

__device__ __forceinline__ float comp3(float3 v, int i) { return i == 0 ? v.x : i == 1 ? v.y : v.z; }
__device__ __forceinline__ void set3(float3& v, int i, float x) { if (i == 0) v.x = x; else if (i == 1) v.y = x; else v.z = x; }

__device__ __forceinline__ int compi3(int3 v, int i) { return i == 0 ? v.x : i == 1 ? v.y : v.z; }
__device__ __forceinline__ void seti3(int3& v, int i, int x) { if (i == 0) v.x = x; else if (i == 1) v.y = x; else v.z = x; }

__device__ __forceinline__ int cube_face_of(float3 v)
{
    const float ax = fabsf(v.x), ay = fabsf(v.y), az = fabsf(v.z);
    int f;
    if (az > fmaxf(ax, ay)) f = 4; else if (ay > ax) f = 2; else f = 0;
    if (comp3(v, f >> 1) < 0.f) f += 1;
    return f;
}

// (s,t) in [0,1] and the face, or -1 for an invalid direction (:87-110).
__device__ __forceinline__ int cube_index(float3 v, float& s, float& t)
{
#pragma clang fp contract(off)
    const int f = cube_face_of(v);
    const CubeFace F = cube_face(f);
    const float m = rcp_rz(fabsf(comp3(v, F.ma))) * .5f;
    const float x = __fmaf_rn(comp3(v, F.sa), (float)F.ss * m, .5f);
    const float y = __fmaf_rn(comp3(v, F.ta), (float)F.ts * m, .5f);
    if (!isfinite(x) || !isfinite(y)) return -1;
    s = fminf(fmaxf(x, 0.f), 1.f);
    t = fminf(fmaxf(y, 0.f), 1.f);
    return f;
}

// Texel (ix,iy) of face f at size w, possibly one step outside the face -> (column, row + w * face) on
// the face it belongs to; false for the texel that does not exist at a cube corner.  Half-texel integer
// geometry: cube [-w,w]^3, texel centres at odd coordinates, a texel beyond an edge folds one
// half-texel inside the neighbouring face.
__device__ __forceinline__ bool cube_texel(int f, int ix, int iy, int w, int& cx, int& cy)
{
    const bool ox = (ix < 0 || ix >= w), oy = (iy < 0 || iy >= w);
    if (ox && oy) return false;
    if (!ox && !oy) { cx = ix; cy = iy + w * f; return true; }
    const CubeFace F = cube_face(f);
    int3 q = make_int3(0, 0, 0);
    seti3(q, F.ma, F.msgn * w);
    seti3(q, F.sa, F.ss * (2 * ix + 1 - w));
    seti3(q, F.ta, F.ts * (2 * iy + 1 - w));
    const int oa = ox ? F.sa : F.ta;
    const int nsgn = compi3(q, oa) > 0 ? 1 : -1;
    seti3(q, oa, nsgn * w);
    seti3(q, F.ma, F.msgn * (w - 1));
    const int nf = oa * 2 + (nsgn < 0 ? 1 : 0);
    const CubeFace G = cube_face(nf);
    cx = (G.ss * compi3(q, G.sa) + w - 1) >> 1;
    cy = ((G.ts * compi3(q, G.ta) + w - 1) >> 1) + w * nf;
    return true;
}

// Footprint on a cube level (:382-434): no clamp, no wrap.  All taps -1 for an invalid direction.
__device__ __forceinline__ Quad tex_index_linear_cube(const TexParams& p, float3 v3, int tz, int level)
{
#pragma clang fp contract(off)
    Quad q;
    q.corner = false;
    q.fu = 0.f; q.fv = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) { q.tc[k] = -1; q.tx[k] = 0; q.ty[k] = 0; }
    const int w = level_dim(p.texW, level);
    float s, t;
    const int f = cube_index(v3, s, t);
    if (f < 0) return q;
    const float u = __fmaf_rn(s, (float)w, -0.5f), v = __fmaf_rn(t, (float)w, -0.5f);
    const int iu0 = __float2int_rd(u), iv0 = __float2int_rd(v);
    q.fu = u - (float)iu0; q.fv = v - (float)iv0;
    const int base = 6 * tz * w * w;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int cx, cy;
        if (cube_texel(f, iu0 + (k & 1), iv0 + (k >> 1), w, cx, cy)) { q.tx[k] = cx; q.ty[k] = cy; q.tc[k] = base + cx + w * cy; }
        else if (tz > 0 && !p.cornerFix) {
            // The reference marks the missing corner texel with face -1, x = y = 0, i.e. index -w*w (:85-88), and then
            // adds 6*tz*w*w to all four indices (:431-432): for slices >= 1 the mark becomes texel (0,0) of face 5 of the
            // PREVIOUS slice, sampled with its bilinear weight and without the corner average.  Reproduced by default;
            // tx = -1 keeps the tap out of the per-slice LDS patch table of the gradient kernel (direct atomic instead).
            q.tc[k] = base - w * w; q.tx[k] = -1; q.ty[k] = 0;
        }
        else q.corner = true;
    }
    return q;
}

// dA/d(s,t) -> dA/d(x,y,z) (:113-140).
__device__ __forceinline__ float3 cube_grad(float3 v, float gu, float gv)
{
#pragma clang fp contract(off)
    const CubeFace F = cube_face(cube_face_of(v));
    const float c = comp3(v, F.ma);
    const float m = rcp_rz(fabsf(c)), h = m * .5f;
    const float su = (float)F.ss * gu, sv = (float)F.ts * gv;
    const float sg = (c < 0.f) ? 1.f : -1.f;
    float3 g = make_float3(0.f, 0.f, 0.f);
    set3(g, F.sa, su * h);
    set3(g, F.ta, sv * h);
    set3(g, F.ma, sg * (su * comp3(v, F.sa) + sv * comp3(v, F.ta)) * m * h);
    if (!isfinite(g.x) || !isfinite(g.y) || !isfinite(g.z)) g = make_float3(0.f, 0.f, 0.f);
    return g;
}

// d(x,y,z)/d(X,Y) -> (ds/dX, ds/dY, dt/dX, dt/dY) (:184-233).
__device__ __forceinline__ float4 cube_grad_st(float3 v, float3 dX, float3 dY)
{
#pragma clang fp contract(off)
    const CubeFace F = cube_face(cube_face_of(v));
    const float c = comp3(v, F.ma);
    const float m = rcp_rz(fabsf(c)), h = m * .5f;
    const float k = ((c < 0.f) ? -1.f : 1.f) * m * h;
    const float ss = (float)F.ss, ts = (float)F.ts, a = comp3(v, F.sa), b = comp3(v, F.ta);
    const float4 r = make_float4(ss * (h * comp3(dX, F.sa) - k * a * comp3(dX, F.ma)), ss * (h * comp3(dY, F.sa) - k * a * comp3(dY, F.ma)),
                                 ts * (h * comp3(dX, F.ta) - k * b * comp3(dX, F.ma)), ts * (h * comp3(dY, F.ta) - k * b * comp3(dY, F.ma)));
    return finite4(r) ? r : make_float4(0.f, 0.f, 0.f, 0.f);
}

// sum_j g_j * d(component j of (ds/dX, ds/dY, dt/dX, dt/dY))/d(x,y,z) (:235-317 contracted with g).
__device__ __forceinline__ float3 cube_grad2_dot(float3 v, float3 dX, float3 dY, float4 g)
{
#pragma clang fp contract(off)
    const CubeFace F = cube_face(cube_face_of(v));
    const float c = comp3(v, F.ma);
    const float m = rcp_rz(fabsf(c)), h = m * .5f;
    const float k = ((c < 0.f) ? -1.f : 1.f) * m * h;
    const float k2 = 2.f * k / c;
    const float ss = (float)F.ss, ts = (float)F.ts, a = comp3(v, F.sa), b = comp3(v, F.ta);
    const float dXc = comp3(dX, F.ma), dYc = comp3(dY, F.ma);
    // rows of the Jacobian, in the oracle's summation order ((J0*g0 + J1*g1) + J2*g2) + J3*g3
    const float ja0 = -ss * k * dXc, ja1 = -ss * k * dYc;
    const float jb2 = -ts * k * dXc, jb3 = -ts * k * dYc;
    const float jc0 = ss * (-k * comp3(dX, F.sa) + k2 * a * dXc), jc1 = ss * (-k * comp3(dY, F.sa) + k2 * a * dYc);
    const float jc2 = ts * (-k * comp3(dX, F.ta) + k2 * b * dXc), jc3 = ts * (-k * comp3(dY, F.ta) + k2 * b * dYc);
    float3 r = make_float3(0.f, 0.f, 0.f);
    set3(r, F.sa, ((ja0 * g.x + ja1 * g.y) + 0.f * g.z) + 0.f * g.w);
    set3(r, F.ta, ((0.f * g.x + 0.f * g.y) + jb2 * g.z) + jb3 * g.w);
    set3(r, F.ma, ((jc0 * g.x + jc1 * g.y) + jc2 * g.z) + jc3 * g.w);
    return r;
}

// dL/d(ds/dX, ds/dY, dt/dX, dt/dY) -> dL/d(d(x,y,z)/dX), dL/d(d(x,y,z)/dY) (:142-182).
__device__ __forceinline__ void cube_grad4(float3 v, float4 dw, float3& g0, float3& g1)
{
#pragma clang fp contract(off)
    const CubeFace F = cube_face(cube_face_of(v));
    const float c = comp3(v, F.ma);
    const float m = rcp_rz(fabsf(c)), h = m * .5f;
    const float k = ((c < 0.f) ? -1.f : 1.f) * m * h;
    const float ss = (float)F.ss, ts = (float)F.ts, a = comp3(v, F.sa), b = comp3(v, F.ta);
    g0 = make_float3(0.f, 0.f, 0.f); g1 = g0;
    set3(g0, F.sa, dw.x * ss * h); set3(g0, F.ta, dw.z * ts * h); set3(g0, F.ma, -k * (dw.x * ss * a + dw.z * ts * b));
    set3(g1, F.sa, dw.y * ss * h); set3(g1, F.ta, dw.w * ts * h); set3(g1, F.ma, -k * (dw.y * ss * a + dw.w * ts * b));
    const bool ok = isfinite(g0.x) && isfinite(g0.y) && isfinite(g0.z) && isfinite(g1.x) && isfinite(g1.y) && isfinite(g1.z);
    if (!ok) { g0 = make_float3(0.f, 0.f, 0.f); g1 = g0; }
}

// texture_kernel.cu:477-585
template <int FILTER, bool BIAS_ONLY, bool CUBE = false>
__device__ __forceinline__ void tex_mip_level(const TexParams& p, size_t pidx, int& level0, int& level1, float& flevel, float4* dw,
                                              float3 uv3 = make_float3(0.f, 0.f, 0.f), float3* dfdv = nullptr, bool zeroDA = false,
                                              const float4* preDA = nullptr)
{
#pragma clang fp contract(off)
    level0 = 0; level1 = 0; flevel = 0.f;
    if (FILTER == TEX_NEAREST || FILTER == TEX_LINEAR) return;
    if (!BIAS_ONLY) {
        float4 d;
        float3 dvdX = make_float3(0.f, 0.f, 0.f), dvdY = dvdX;
        if (CUBE) {
            const float2* q = (const float2*)p.uvDA + pidx * 3;                 // (d/dX, d/dY) of x, y, z
            const float2 d0 = q[0], d1 = q[1], d2 = q[2];
            dvdX = make_float3(d0.x, d1.x, d2.x); dvdY = make_float3(d0.y, d1.y, d2.y);
            d = cube_grad_st(uv3, dvdX, dvdY);
        } else {
            d = zeroDA ? make_float4(0.f, 0.f, 0.f, 0.f) : preDA ? *preDA : ((const float4*)p.uvDA)[pidx];
        }
        const float uscl = (float)p.texW, vscl = (float)p.texH;
        const float dsdx = d.x * uscl, dsdy = d.y * uscl, dtdx = d.z * vscl, dtdy = d.w * vscl;
        const float A = dsdx * dsdx + dtdx * dtdx;
        const float B = dsdy * dsdy + dtdy * dtdy;
        const float C = dsdx * dsdy + dtdx * dtdy;
        const float l2b = 0.5f * (A + B);
        const float l2n = 0.25f * (A - B) * (A - B) + C * C;
        const float l2a = sqrtf(l2n);
        const float lenMajorSqr = l2b + l2a;
        if (dw && FILTER == TEX_LML) {
            const float k = 0.72134752f / (l2n + l2a * l2b);                  // 0.5 / ln 2
            const float AB = k * .5f * (A - B);
            const float Cw = k * C;
            const float l2aw = k * l2a;
            const float4 g = make_float4(uscl * (dsdx * (l2aw + AB) + dsdy * Cw), uscl * (dsdy * (l2aw - AB) + dsdx * Cw),
                                         vscl * (dtdx * (l2aw + AB) + dtdy * Cw), vscl * (dtdy * (l2aw - AB) + dtdx * Cw));
            bool ok = finite4(g);
            if (CUBE) {
                const float3 fv = cube_grad2_dot(uv3, dvdX, dvdY, g);
                ok = ok && isfinite(fv.x) && isfinite(fv.y) && isfinite(fv.z);
                *dfdv = ok ? fv : make_float3(0.f, 0.f, 0.f);
            }
            *dw = ok ? g : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        flevel = .5f * log2f(lenMajorSqr);                                    // inf/NaN are fixed by the clamp
    }
    if (p.bias) flevel += p.bias[pidx];
    flevel = fminf(fmaxf(flevel, 0.f), (float)p.levelMax);
    level0 = __float2int_rd(flevel);
    if (FILTER == TEX_LML && flevel > 0.f) {
        level1 = min(level0 + 1, p.levelMax);
        flevel -= (float)level0;
    }
}

// The four texels of channel c; at a cube corner the missing texel takes the average of the other
// three (texture_kernel.cu:590-614).
__device__ __forceinline__ void fetch_quad(const float* base, const Quad& q, int C, int c, float a[4])
{
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) { a[k] = q.tc[k] >= 0 ? base[q.tc[k] * C + c] : 0.f; sum += a[k]; }
    if (q.corner) {
        const float avg = sum * 0.33333333f;
#pragma unroll
        for (int k = 0; k < 4; k++) if (q.tc[k] < 0) a[k] = avg;
    }
}

// Scatter weights of the four taps; at a cube corner the missing texel's weight is shared by the other
// three (texture_kernel.cu:616-639).
__device__ __forceinline__ void corner_weights(const Quad& q, float w[4])
{
    if (!q.corner) return;
    float cb = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) if (q.tc[k] < 0) cb = w[k];
    cb *= 0.33333333f;
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] += cb;
}

// Nearest texel of a cube map (:331-338): no wrap, the face is folded into the slice index.
__device__ __forceinline__ int tex_index_nearest_cube(const TexParams& p, float3 v, int tz, int& x, int& y)
{
#pragma clang fp contract(off)
    float s, t;
    const int f = cube_index(v, s, t);
    if (f < 0) return -1;
    const int w = p.texW;
    int iu = __float2int_rd(s * (float)w), iv = __float2int_rd(t * (float)w);
    iu = min(max(iu, 0), w - 1);
    iv = min(max(iv, 0), w - 1);
    x = iu; y = iv + w * f;
    return iu + w * (y + 6 * tz * w);
}

__device__ __forceinline__ float lerp1(float a, float b, float c) { return a + c * (b - a); }
__device__ __forceinline__ float bilerp1(float a, float b, float c, float d, float fu, float fv) { return lerp1(lerp1(a, b, fu), lerp1(c, d, fu), fv); }

// A texel's C_CT channels as one vector load when the channel count allows it.
template <int C_CT> struct TexelVec { float v[C_CT > 0 ? C_CT : 1]; };

template <int C_CT>
__device__ __forceinline__ void load_texel(float* dst, const float* base, int tc, int C)
{
    if (tc < 0) { for (int c = 0; c < (C_CT > 0 ? C_CT : C); c++) dst[c] = 0.f; return; }
    const float* s = base + tc * (C_CT > 0 ? C_CT : C);
    if (C_CT == 4) { const float4 t = *(const float4*)s; dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w; }
    else if (C_CT == 2) { const float2 t = *(const float2*)s; dst[0] = t.x; dst[1] = t.y; }
    else { for (int c = 0; c < (C_CT > 0 ? C_CT : C); c++) dst[c] = s[c]; }
}

// Pixel of this lane: a workgroup owns a 16x16 pixel block (four waves = 2x2 tiles of 8x8), blocks
// are dealt to the XCDs in contiguous chunks.  Returns false when the whole workgroup has no block;
// `inside` tells whether this lane's pixel exists.
template <bool ORDERED = true>
__device__ __forceinline__ bool tex_pixel(const TexParams& p, int& px, int& py, int& pz, bool& inside, int* blkOut = nullptr)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // 32-bit on purpose (the host checks the block count): a 64-bit division here is ~140 scalar instructions per wave
    const int blocksPerImage = p.tilesX * p.tilesY;                           // tilesX/Y count 16x16 blocks here
    inside = false;
    int bx, by;
    if (ORDERED && p.zflags.order) {
        // the rasterizer's work order travels with the flags: bins with triangles first, an eighth of them per XCD
        // (nvdr_device.hpp TileFlags)
        if (!decode_block_ordered(p.zflags, p.tilesX, p.tilesY, 16, 16, bx, by, pz)) return false;
    } else {
        const int total = blocksPerImage * p.n;
        const int perXcd = (total + 7) >> 3;
        const int j = (int)(blockIdx.x >> 3);
        const int blk = (int)(blockIdx.x & 7) * perXcd + j;
        if (j >= perXcd || blk >= total) return false;
        pz = blk / blocksPerImage;
        const int rem = blk - pz * blocksPerImage;
        by = rem / p.tilesX; bx = rem - by * p.tilesX;
    }
    if (blkOut) *blkOut = pz * blocksPerImage + by * p.tilesX + bx;          // the block's number whatever the order of the launch
    px = bx * 16 + (wave & 1) * 8 + (lane & 7);
    py = by * 16 + (wave >> 1) * 8 + (lane >> 3);
    inside = px < p.imgW && py < p.imgH;
    return true;
}

// ---- forward (texture_kernel.cu:709-800) ---------------------------------------------------

template <int FILTER, bool BIAS_ONLY, int C_CT>
__global__ __launch_bounds__(256) void k_tex_fwd(const TexParams p)
{
    int px, py, pz; bool inside;
    constexpr int CMAX = C_CT > 0 ? C_CT : 1;
    const int C = C_CT > 0 ? C_CT : p.channels;
    // The sample of a tile of known-zero uv / uv_da (a wave is one 8x8 tile: three quarters of a rendered image): level 0 (the
    // footprint is zero: flevel = clamp(log2 0) = 0) at uv = (0, 0), ONE place -- computed once per wave, the quad's texels through
    // scalar loads, and stored 64 times, instead of walking the whole path per pixel.  Not with a per-pixel bias, which moves the
    // level pixel by pixel.
    const bool kUniform = C_CT > 0 && FILTER != TEX_NEAREST && !p.bias;
    float ur[CMAX];
    bool haveU = false;
    auto uniform_sample = [&](int tz_) {
        const Quad q0 = tex_index_linear(p, 0.f, 0.f, tz_, 0);
        float a[4][CMAX];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int tc = __builtin_amdgcn_readfirstlane(q0.tc[k]);
            // wave-uniform address: scalar loads (selected and waited for by the compiler, nvdr_device.hpp scalar_load); a tap
            // without a texel reads texel 0 and is zeroed afterwards, so that all loads are in flight together
            const float* tp = p.tex[0] + (size_t)max(tc, 0) * CMAX;
#pragma unroll
            for (int c = 0; c < CMAX; c++) { const float t = scalar_load(tp + c); a[k][c] = tc >= 0 ? t : 0.f; }
        }
#pragma unroll
        for (int c = 0; c < CMAX; c++) ur[c] = bilerp1(a[0][c], a[1][c], a[2][c], a[3][c], q0.fu, q0.fv);
        haveU = true;
    };
    auto store_texel = [&](float* o, const float* r) {
        if (C_CT == 4) *(float4*)o = make_float4(r[0], r[1 % CMAX], r[2 % CMAX], r[3 % CMAX]);
        else if (C_CT == 2) *(float2*)o = make_float2(r[0], r[1 % CMAX]);
        else for (int c = 0; c < CMAX; c++) o[c] = r[c];
    };
    if (kUniform && p.zflags.order) {
        // Along the work order a covered bin's workgroup also stores the samples of one EMPTY bin of its XCD's share (the same
        // 16x16 block of it), whose own workgroup leaves at once: the stores of the bins that only need that one sample go out while
        // the waves with real footprints wait for their gathers (nvdr_device.hpp ordered_list_pair, as in k_interp_fwd_cols).
        const TileFlags& t = p.zflags;
        const int xcd = (int)(blockIdx.x & 7), j = (int)(blockIdx.x >> 3);
        const int slot = j >> 4, sub = j & 15;
        int own, partner; bool skip;
        ordered_list_pair(t.nBins, t.order[t.nBins], xcd, slot, own, partner, skip);
        if (own < 0 || skip) return;
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        auto place = [&](int idx, int& x, int& y, int& z) {
            const int bin = __builtin_amdgcn_readfirstlane(t.order[idx]);
            z = bin / (t.binsX * t.binsY);
            const int rem = bin - z * (t.binsX * t.binsY);
            const int binY = rem / t.binsX, binX = rem - binY * t.binsX;
            x = (binX * 4 + (sub & 3)) * 16 + (wave & 1) * 8 + (lane & 7);
            y = (binY * 4 + (sub >> 2)) * 16 + (wave >> 1) * 8 + (lane >> 3);
        };
        if (partner >= 0) {
            int qx, qy, qz;
            place(partner, qx, qy, qz);
            const int qtz = (p.texDepth == 1) ? 0 : qz;
            uniform_sample(qtz);
            if (qx < p.imgW && qy < p.imgH) store_texel(p.out + ((size_t)qx + (size_t)p.imgW * (qy + (size_t)p.imgH * qz)) * C, ur);
            if (p.texDepth != 1) haveU = false;                    // (another image may sample another texture slice)
        }
        place(own, px, py, pz);
        inside = px < p.imgW && py < p.imgH;
        if (!inside) return;
    } else if (!tex_pixel(p, px, py, pz, inside) || !inside) return;
    const int tz = (p.texDepth == 1) ? 0 : pz;
    const size_t pidx = (size_t)px + (size_t)p.imgW * (py + (size_t)p.imgH * pz);
    const bool zt = p.zflags.empty(pz, py, px);               // uv = uv_da = 0 known for this tile: not read
    const float2 uv = zt ? make_float2(0.f, 0.f) : ((const float2*)p.uv)[pidx];
    float* pOut = p.out + pidx * C;

    if (FILTER == TEX_NEAREST) {
        const int tc = tex_index_nearest(p, uv.x, uv.y, tz);
        if (C_CT > 0) {
            float t[CMAX];
            load_texel<C_CT>(t, p.tex[0], tc, C);
            if (C_CT == 4) *(float4*)pOut = make_float4(t[0], t[1 % CMAX], t[2 % CMAX], t[3 % CMAX]);
            else if (C_CT == 2) *(float2*)pOut = make_float2(t[0], t[1 % CMAX]);
            else for (int c = 0; c < C_CT; c++) pOut[c] = t[c];
        } else {
            for (int c = 0; c < C; c++) pOut[c] = tc >= 0 ? p.tex[0][tc * C + c] : 0.f;
        }
        return;
    }

    if (kUniform && zt) {
        if (!haveU) uniform_sample(tz);
        store_texel(pOut, ur);
        return;
    }

    int level0, level1; float flevel;
    tex_mip_level<FILTER, BIAS_ONLY>(p, pidx, level0, level1, flevel, nullptr, make_float3(0.f, 0.f, 0.f), nullptr, zt);
    const Quad q0 = tex_index_linear(p, uv.x, uv.y, tz, level0);
    const float* pIn0 = p.tex[level0];
    const bool second = (FILTER == TEX_LML) && flevel > 0.f;
    Quad q1 = q0;
    const float* pIn1 = pIn0;
    if (second) { q1 = tex_index_linear(p, uv.x, uv.y, tz, level1); pIn1 = p.tex[level1]; }

    if (C_CT > 0) {
        float a00[CMAX], a10[CMAX], a01[CMAX], a11[CMAX], r[CMAX];
        load_texel<C_CT>(a00, pIn0, q0.tc[0], C); load_texel<C_CT>(a10, pIn0, q0.tc[1], C);
        load_texel<C_CT>(a01, pIn0, q0.tc[2], C); load_texel<C_CT>(a11, pIn0, q0.tc[3], C);
#pragma unroll
        for (int c = 0; c < CMAX; c++) r[c] = bilerp1(a00[c], a10[c], a01[c], a11[c], q0.fu, q0.fv);
        if (second) {
            float b00[CMAX], b10[CMAX], b01[CMAX], b11[CMAX];
            load_texel<C_CT>(b00, pIn1, q1.tc[0], C); load_texel<C_CT>(b10, pIn1, q1.tc[1], C);
            load_texel<C_CT>(b01, pIn1, q1.tc[2], C); load_texel<C_CT>(b11, pIn1, q1.tc[3], C);
#pragma unroll
            for (int c = 0; c < CMAX; c++) r[c] = lerp1(r[c], bilerp1(b00[c], b10[c], b01[c], b11[c], q1.fu, q1.fv), flevel);
        }
        if (C_CT == 4) *(float4*)pOut = make_float4(r[0], r[1 % CMAX], r[2 % CMAX], r[3 % CMAX]);
        else if (C_CT == 2) *(float2*)pOut = make_float2(r[0], r[1 % CMAX]);
        else for (int c = 0; c < CMAX; c++) pOut[c] = r[c];
    } else {
        for (int c = 0; c < C; c++) {
            float a = bilerp1(q0.tc[0] >= 0 ? pIn0[q0.tc[0] * C + c] : 0.f, q0.tc[1] >= 0 ? pIn0[q0.tc[1] * C + c] : 0.f,
                              q0.tc[2] >= 0 ? pIn0[q0.tc[2] * C + c] : 0.f, q0.tc[3] >= 0 ? pIn0[q0.tc[3] * C + c] : 0.f, q0.fu, q0.fv);
            if (second) {
                const float b = bilerp1(q1.tc[0] >= 0 ? pIn1[q1.tc[0] * C + c] : 0.f, q1.tc[1] >= 0 ? pIn1[q1.tc[1] * C + c] : 0.f,
                                        q1.tc[2] >= 0 ? pIn1[q1.tc[2] * C + c] : 0.f, q1.tc[3] >= 0 ? pIn1[q1.tc[3] * C + c] : 0.f, q1.fu, q1.fv);
                a = lerp1(a, b, flevel);
            }
            pOut[c] = a;
        }
    }
}

// The four texels of a cube footprint, all channels at once (vector loads for C_CT = 2 / 4), with the corner rule of
// fetch_quad applied per channel (same summation order, so both routes give identical bits).
template <int C_CT>
__device__ __forceinline__ void fetch_quad_vec(const float* base, const Quad& q, int C, float a[4][C_CT > 0 ? C_CT : 1])
{
#pragma unroll
    for (int k = 0; k < 4; k++) load_texel<C_CT>(a[k], base, q.tc[k], C);
    if (q.corner) {
#pragma unroll
        for (int c = 0; c < C_CT; c++) {
            const float avg = (((a[0][c] + a[1][c]) + a[2][c]) + a[3][c]) * 0.33333333f;
#pragma unroll
            for (int k = 0; k < 4; k++) if (q.tc[k] < 0) a[k][c] = avg;
        }
    }
}

// Cube-map forward: same pixel mapping, direction vectors instead of (u,v), footprints that may cross
// a face edge (three faces at a corner).  C_CT = 1..4: channel count known at compile time (texels of 2 and 4 channels
// are single 8 / 16-byte loads, the output one store); C_CT = 0: generic channel loop.
template <int FILTER, bool BIAS_ONLY, int C_CT>
__global__ __launch_bounds__(256) void k_tex_fwd_cube(const TexParams p)
{
    int px, py, pz; bool inside;
    if (!tex_pixel(p, px, py, pz, inside) || !inside) return;
    constexpr int CMAX = C_CT > 0 ? C_CT : 1;
    const int C = C_CT > 0 ? C_CT : p.channels;
    const int tz = (p.texDepth == 1) ? 0 : pz;
    const size_t pidx = (size_t)px + (size_t)p.imgW * (py + (size_t)p.imgH * pz);
    const float* puv = p.uv + pidx * 3;
    const float3 uv3 = make_float3(puv[0], puv[1], puv[2]);
    float* pOut = p.out + pidx * C;

    if (FILTER == TEX_NEAREST) {
        int x, y;
        const int tc = tex_index_nearest_cube(p, uv3, tz, x, y);
        if (C_CT > 0) {
            float t[CMAX];
            load_texel<C_CT>(t, p.tex[0], tc, C);
            if (C_CT == 4) *(float4*)pOut = make_float4(t[0], t[1 % CMAX], t[2 % CMAX], t[3 % CMAX]);
            else if (C_CT == 2) *(float2*)pOut = make_float2(t[0], t[1 % CMAX]);
            else for (int c = 0; c < C_CT; c++) pOut[c] = t[c];
        } else {
            for (int c = 0; c < C; c++) pOut[c] = tc >= 0 ? p.tex[0][tc * C + c] : 0.f;
        }
        return;
    }
    int level0, level1; float flevel;
    tex_mip_level<FILTER, BIAS_ONLY, true>(p, pidx, level0, level1, flevel, nullptr, uv3, nullptr);
    const Quad q0 = tex_index_linear_cube(p, uv3, tz, level0);
    const bool second = (FILTER == TEX_LML) && flevel > 0.f;
    Quad q1 = q0;
    if (second) q1 = tex_index_linear_cube(p, uv3, tz, level1);
    if (C_CT > 0) {
        float a[4][CMAX], r[CMAX];
        fetch_quad_vec<C_CT>(p.tex[level0], q0, C, a);
#pragma unroll
        for (int c = 0; c < CMAX; c++) r[c] = bilerp1(a[0][c], a[1][c], a[2][c], a[3][c], q0.fu, q0.fv);
        if (second) {
            fetch_quad_vec<C_CT>(p.tex[level1], q1, C, a);
#pragma unroll
            for (int c = 0; c < CMAX; c++) r[c] = lerp1(r[c], bilerp1(a[0][c], a[1][c], a[2][c], a[3][c], q1.fu, q1.fv), flevel);
        }
        if (C_CT == 4) *(float4*)pOut = make_float4(r[0], r[1 % CMAX], r[2 % CMAX], r[3 % CMAX]);
        else if (C_CT == 2) *(float2*)pOut = make_float2(r[0], r[1 % CMAX]);
        else for (int c = 0; c < CMAX; c++) pOut[c] = r[c];
        return;
    }
    for (int c = 0; c < C; c++) {
        float a[4];
        fetch_quad(p.tex[level0], q0, C, c, a);
        float r = bilerp1(a[0], a[1], a[2], a[3], q0.fu, q0.fv);
        if (second) {
            fetch_quad(p.tex[level1], q1, C, c, a);
            r = lerp1(r, bilerp1(a[0], a[1], a[2], a[3], q1.fu, q1.fv), flevel);
        }
        pOut[c] = r;
    }
}

// ---- backward (texture_kernel.cu:905-1140) -------------------------------------------------

// Texel-gradient accumulator of one workgroup: an LDS open-addressing table of 8x2-texel patches
// keyed by (level, patch x, patch y), each patch holding 16 texels x C channels of 32-bit fixed-point
// sums (nvdr_device.hpp: LDS integer atomics are ~30x cheaper than ds_add_f32, and a global atomic
// costs one memory transaction per touched cache line).  32 bits (resolution 2^-22 of the block's
// largest |dy|, i.e. the ulp of an f32 sum of that size) instead of the vertex tables' 64 keep the
// table at 25 KB so that six workgroups fit a CU.  Every tap of the workgroup's 16x16 pixels is
// added here; at the end each patch row is flushed by consecutive lanes, so one atomic instruction
// covers a few whole lines instead of 64 scattered ones, and pixels that hit the same texel (e.g. a
// constant-uv background) cost one global atomic per workgroup instead of one per pixel.
// The fixed-point scale comes from the block's largest |dy| (every tap weight is in [0,1]).
struct PatchTable {
    uint32_t* keys;               // [groups]  0 = empty
    int* vals;                    // [groups * 16 * C] 32-bit fixed-point sums
    int groups, C;

    __device__ __forceinline__ void clear(int tid, int nthreads) {
        // vals then keys, contiguous and 16-byte aligned (groups is a power of two >= 16): 16-byte stores
        uint4* q = (uint4*)vals;
        const int n16 = (groups * 64 * C + groups * 4) >> 4;
        for (int i = tid; i < n16; i += nthreads) q[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    // (level + 1) << 27 | patch row << 12 | patch column: textures up to 32768 x 65536 texels (the host falls
    // back to direct atomics beyond that).  32-bit keys: one ds_cmpst_b32 and a 32-bit hash per probe.
    static __device__ __forceinline__ uint32_t key_of(int level, int x, int y) {
        return ((uint32_t)(level + 1) << 27) | ((uint32_t)(y >> 1) << 12) | (uint32_t)(x >> 3);
    }
    // Index of texel (x, y) of `level` in vals (in texels, multiply by C) or -1 when the table is full.
    __device__ __forceinline__ int find(int level, int x, int y) const {
        const uint32_t key = key_of(level, x, y);
        uint32_t h = key * 0x9E3779B1u;
        h ^= h >> 15;
#pragma unroll 1
        for (int probe = 0; probe < 8; probe++) {
            h &= (uint32_t)(groups - 1);
            const uint32_t old = atomicCAS(&keys[h], 0u, key);
            if (old == 0u || old == key) return (int)h * 16 + (y & 1) * 8 + (x & 7);
            h++;
        }
        return -1;
    }
};

template <int FILTER, bool BIAS_ONLY, bool CUBE, int C_CT>
__global__ __launch_bounds__(256, 5) void k_tex_grad(const TexParams p, int groups)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_mem[];
    const int C = C_CT > 0 ? C_CT : p.channels;          // compile-time channel count for the common cases: the loops unroll
    // LDS layout: vals [groups*16*C] | keys [groups] | {block max, used count, -, -} | used-patch list [groups]
    PatchTable tab{(uint32_t*)((int*)s_mem + (size_t)groups * 16 * C), (int*)s_mem, groups, C};
    uint32_t* s_max = tab.keys + groups;                                    // [0] block max, [1] number of used patches
    int* s_used = (int*)(s_max + 4);                                        // [groups] indices of the used patches (flush)
    int px = 0, py = 0, pz = 0, blk = 0; bool inside;
    if (!tex_pixel(p, px, py, pz, inside, &blk)) return;
    if (p.heavy && !p.heavy[blk]) return;                        // two-kernel pass: blocks that k_tex_grad_light has finished
    if (groups > 0 && !(p.dbg & 2048)) tab.clear(threadIdx.x, 256);
    if (threadIdx.x == 0) { s_max[0] = 0u; s_max[1] = 0u; }
    __syncthreads();

    const int tz = (p.texDepth == 1) ? 0 : pz;
    const size_t pidx = (size_t)px + (size_t)p.imgW * (py + (size_t)p.imgH * pz);
    const float* pDy = p.dy + pidx * C;
    const bool zt = !CUBE && inside && p.zflags.empty(pz, py, px);   // uv = uv_da = 0 known for this tile: not read

    // uv and uv_da (2-D textures) are fetched TOGETHER WITH the upstream gradient, before the block's barrier: their round trip
    // to memory overlaps dy's and the wait for the block's other waves instead of following both (the kernel is bound by its chain
    // of dependent loads, section 6: 0.790 -> 0.758 ms at config 3 with the fetch behind dy, same registers).  A pixel whose
    // upstream gradient turns out to be zero has then fetched them for nothing; where dy is masked to the covered pixels the
    // background's tiles are flagged and read nothing.
    constexpr bool kPreUV = !CUBE;
    float2 preUV = make_float2(0.f, 0.f);
    float4 preDA = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kPreUV && inside && !zt) {
        preUV = ((const float2*)p.uv)[pidx];
        if ((FILTER == TEX_LMN || FILTER == TEX_LML) && !BIAS_ONLY) preDA = ((const float4*)p.uvDA)[pidx];
    }
    // ---- phase A: all-zero upstream gradients take the early-out (explicit zero stores, :922-971);
    //      the rest publish the block's largest |dy|.
    bool active = false;
    float m = 0.f;
    if (inside) {
        uint32_t dmax = 0u;
        for (int c = 0; c < C; c++) { const float d = pDy[c]; dmax |= (uint32_t)__float_as_int(d); m = max_abs_keep_nan(m, d); }
        active = !(__int_as_float((int)dmax) == 0.f);
        if (!active) {
            m = 0.f;
            if (FILTER != TEX_NEAREST) {
                if (CUBE) { float* g = p.gradUV + pidx * 3; g[0] = 0.f; g[1] = 0.f; g[2] = 0.f; }
                else ((float2*)p.gradUV)[pidx] = make_float2(0.f, 0.f);
            }
            if (FILTER == TEX_LML) {
                if (p.gradUVDA) {
                    if (CUBE) { float2* g = (float2*)p.gradUVDA + pidx * 3; g[0] = g[1] = g[2] = make_float2(0.f, 0.f); }
                    else ((float4*)p.gradUVDA)[pidx] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                if (p.gradBias) p.gradBias[pidx] = 0.f;
            }
        }
    }
    block_max_update(s_max, m);
    __syncthreads();
    const uint32_t maxBits = *s_max;
    if (maxBits == 0u) return;                                   // nobody has anything to scatter
    const bool direct = (groups == 0) || maxBits >= 0x7F800000u; // no table / inf or NaN present: plain f32 atomics
    const FixedScale32 fs(direct ? 0x3F800000u : maxBits);

    // One tap's contribution goes to the LDS table when the tap has a slot; taps without one (table full, no
    // table, inf/NaN present) are sent to memory directly under a wave-uniform test, off the common path.
    auto scatter = [&](int slot, int c, float v) {
        if (slot >= 0) atomicAdd(&tab.vals[slot * C + c], fs.to_fixed(v));
    };
    auto spill = [&](int slot, int level, int tc, int c, float v) {
        // (the level number is laundered through an empty asm: otherwise the 64-bit ADDRESS of the table entry is kept
        //  live across the whole scatter -- and, there being no register for it, parked in scratch -- for this rare path)
        if (tc >= 0 && slot < 0) { int lv = level; asm volatile("" : "+v"(lv)); atomic_add_f32(p.gradTex[lv] + tc * C + c, v); }
    };
    auto slots_of = [&](const Quad& q, int level, int* sl) {
        // The four taps of a bilinear footprint share patches most of the time: look each patch up once.
        sl[0] = sl[1] = sl[2] = sl[3] = -1;
        if (direct || (p.dbg & (1024 | 131072))) return;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (q.tc[k] < 0 || q.tx[k] < 0) continue;
            bool reused = false;
#pragma unroll
            for (int j = 0; j < k; j++) {
                if (!reused && q.tc[j] >= 0 && sl[j] >= 0 && (q.tx[k] >> 3) == (q.tx[j] >> 3) && (q.ty[k] >> 1) == (q.ty[j] >> 1)) {
                    sl[k] = (sl[j] & ~15) + (q.ty[k] & 1) * 8 + (q.tx[k] & 7);
                    reused = true;
                }
            }
            if (!reused) sl[k] = tab.find(level, q.tx[k], q.ty[k]);
        }
    };

    // Pixels that land on exactly the same texels (constant uv: backgrounds, flat regions under
    // magnification) would serialise on one LDS address; consecutive lanes with identical footprints
    // are summed in registers first (RunScan) and only the last lane of a run scatters.
    auto same_quad = [&](const Quad& q) {
        const int xa = q.tx[0] | (q.tx[1] << 16), xb = q.tx[2] | (q.tx[3] << 16);
        int same = (int)(RunScan::prev_lane(xa, -1) == xa) & (int)(RunScan::prev_lane(xb, -1) == xb);
#pragma unroll
        for (int k = 0; k < 4; k++) same &= (int)(RunScan::prev_lane(q.ty[k], -1) == q.ty[k]);
        const int valid = (q.tc[0] >= 0 ? 1 : 0) | (q.tc[1] >= 0 ? 2 : 0) | (q.tc[2] >= 0 ? 4 : 0) | (q.tc[3] >= 0 ? 8 : 0);
        same &= (int)(RunScan::prev_lane(valid, -1) == valid);
        return same;
    };
    auto run_of = [&](const Quad& a, int la, const Quad& b, int lb, bool second) {
        const int f = la | (lb << 8) | ((second ? 1 : 0) << 16);
        int same = RunScan::prev_lane(f, -1) == f;
        same &= same_quad(a);
        same &= same_quad(b);
        return RunScan(RunScan::FromHead{}, !same, true);
    };

    // ---- phase B, uniform waves -------------------------------------------------------------------
    // A wave whose 64 pixels all carry the SAME texture coordinate with a ZERO pixel footprint -- the background of a
    // rendered image, which interpolate() fills with uv = 0, uv_da = 0 -- samples level 0 bilinearly at one place:
    // flevel = clamp(log2(0) [+ bias]) = 0 and the level gradient vanishes (:477-585), so g_uv_da = g_bias = 0 and
    // every lane's contribution to the four texels is weight * dy.  Such a wave sums dy over its lanes and lets ONE
    // lane claim the patch and add the four totals, instead of 64 lanes probing the same hash key (a 64-way
    // same-address LDS atomic per probe) and carrying the whole general path.  Exact up to the summation order.
    bool uniformWave = false;
    if (!CUBE && !direct && !(p.dbg & (1024 | 4096)) && (FILTER == TEX_LINEAR || ((FILTER == TEX_LMN || FILTER == TEX_LML) && !BIAS_ONLY))) {
        if (__ballot(active) == ~0ull) {
            const float2 t = zt ? make_float2(0.f, 0.f) : kPreUV ? preUV : ((const float2*)p.uv)[pidx];
            const int ux = __float_as_int(t.x), uy = __float_as_int(t.y);
            bool same = (ux == __builtin_amdgcn_readfirstlane(ux)) & (uy == __builtin_amdgcn_readfirstlane(uy));
            if (FILTER != TEX_LINEAR) {
                const float4 d = zt ? make_float4(0.f, 0.f, 0.f, 0.f) : kPreUV ? preDA : ((const float4*)p.uvDA)[pidx];
                same &= (d.x == 0.f) & (d.y == 0.f) & (d.z == 0.f) & (d.w == 0.f);
                if (p.bias) same &= !(fabsf(p.bias[pidx]) == INFINITY);          // -inf + inf would be NaN, not -inf
            }
            uniformWave = __ballot(same) == ~0ull;
        }
    }
    if (uniformWave) {
        const int lane = threadIdx.x & 63;
        const float2 t = zt ? make_float2(0.f, 0.f) : kPreUV ? preUV : ((const float2*)p.uv)[pidx];
        const Quad q0 = tex_index_linear(p, t.x, t.y, tz, 0);
        const float* pIn0 = p.tex[0];
        const float w011 = q0.fu * q0.fv, w010 = q0.fu - w011, w001 = q0.fv - w011, w000 = 1.f - q0.fu - w001;
        const float tw0[4] = {w000, w010, w001, w011};
        const float sclu0 = (float)p.texW, sclv0 = (float)p.texH;
        // With scratch from the caller the wave's totals leave as ONE record (merged by k_tex_grad_fold after this kernel):
        // a background of constant uv would otherwise send one flush per 16x16-pixel block -- 100 k workgroups at config 3 --
        // to the same four texels, where same-address f32 atomics execute one after another (half of that case's time).
        const bool toRecord = p.rec != nullptr;
        int* recBase = nullptr;
        if (toRecord) {
            recBase = p.rec + (blk * 4 + (int)(threadIdx.x >> 6));
            if (lane == 63) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    recBase[(size_t)(4 + k) * p.nrec] = __float_as_int(tw0[k]);
                    if (k > 0) recBase[(size_t)k * p.nrec] = q0.tc[k];
                }
            }
        }
        int sl0[4] = {-1, -1, -1, -1};
        if (lane == 63 && !toRecord) slots_of(q0, 0, sl0);
        float gu = 0.f, gv = 0.f;
        for (int c = 0; c < C; c++) {
            const float d = pDy[c];
            const float tot = wave_sum_to_last(d);                       // valid in lane 63
            if (lane == 63) {
                if (toRecord) recBase[(size_t)(kTexRecHeader + c) * p.nrec] = __float_as_int(tot);
                else {
#pragma unroll
                    for (int k = 0; k < 4; k++) { const float v = tw0[k] * tot; scatter(sl0[k], c, v); spill(sl0[k], 0, q0.tc[k], c, v); }
                }
            }
            float a[4];
            fetch_quad(pIn0, q0, C, c, a);
            const float ad = (a[3] + a[0] - a[1] - a[2]);
            gu += d * ((a[1] - a[0]) + q0.fv * ad) * sclu0;
            gv += d * ((a[2] - a[0]) + q0.fu * ad) * sclv0;
        }
        // tap 0's index makes the record valid (taps without a texel -- boundary mode zero -- carry -1 and weight anything;
        // a record whose FIRST tap has no texel is stored with the first valid tap moved to the front)
        if (toRecord && lane == 63) {
            int first = q0.tc[0];
            if (first < 0) {
#pragma unroll
                for (int k = 1; k < 4; k++) {
                    if (first < 0 && q0.tc[k] >= 0) {
                        first = q0.tc[k];
                        recBase[(size_t)4 * p.nrec] = __float_as_int(tw0[k]);
                        recBase[(size_t)k * p.nrec] = -1;
                    }
                }
            }
            recBase[0] = first;                                           // stays -1 when no tap has a texel
        }
        ((float2*)p.gradUV)[pidx] = make_float2(gu, gv);
        if (FILTER == TEX_LML) {
            if (p.gradBias) p.gradBias[pidx] = 0.f;
            if (p.gradUVDA) ((float4*)p.gradUVDA)[pidx] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }

    // ---- phase B --------------------------------------------------------------------------------
    if (active && !uniformWave) {
        float3 uv3 = make_float3(0.f, 0.f, 0.f);
        if (CUBE) { const float* q = p.uv + pidx * 3; uv3 = make_float3(q[0], q[1], q[2]); }
        else { const float2 t = zt ? make_float2(0.f, 0.f) : kPreUV ? preUV : ((const float2*)p.uv)[pidx]; uv3 = make_float3(t.x, t.y, 0.f); }
        auto footprint = [&](int level) { return CUBE ? tex_index_linear_cube(p, uv3, tz, level) : tex_index_linear(p, uv3.x, uv3.y, tz, level); };

        if (FILTER == TEX_NEAREST) {
            int x = 0, y = 0;
            int tc;
            if (CUBE) tc = tex_index_nearest_cube(p, uv3, tz, x, y);
            else {
                tc = tex_index_nearest(p, uv3.x, uv3.y, tz);
                if (tc >= 0) { const int t = tc - tz * p.texW * p.texH; y = t / p.texW; x = t - y * p.texW; }
            }
            if (tc >= 0) {
                const int sl = direct ? -1 : tab.find(0, x, y);
                for (int c = 0; c < C; c++) { scatter(sl, c, pDy[c]); spill(sl, 0, tc, c, pDy[c]); }
            }
        } else {
            float4 dw = make_float4(0.f, 0.f, 0.f, 0.f);
            float3 dfdv = make_float3(0.f, 0.f, 0.f);
            int level0, level1; float flevel;
            tex_mip_level<FILTER, BIAS_ONLY, CUBE>(p, pidx, level0, level1, flevel, &dw, uv3, &dfdv, zt, (kPreUV && !BIAS_ONLY) ? &preDA : nullptr);

            const Quad q0 = footprint(level0);
            const float* pIn0 = p.tex[level0];
            const float w011 = q0.fu * q0.fv, w010 = q0.fu - w011, w001 = q0.fv - w011, w000 = 1.f - q0.fu - w001;
            const float tw0[4] = {w000, w010, w001, w011};
            const float sclu0 = (float)level_dim(p.texW, level0), sclv0 = (float)level_dim(p.texH, level0);
            int sl0[4];
            slots_of(q0, level0, sl0);
            float gu = 0.f, gv = 0.f, df = 0.f;

            constexpr bool kTri = (FILTER == TEX_LML);
            const bool second = kTri && flevel > 0.f;
            Quad q1 = q0;
            const float* pIn1 = pIn0;
            float tw1[4] = {0.f, 0.f, 0.f, 0.f}, sclu1 = 0.f, sclv1 = 0.f;
            int sl1[4] = {-1, -1, -1, -1};
            if (kTri) {
                q1 = footprint(level1);
                pIn1 = p.tex[level1];
                const float w111 = q1.fu * q1.fv, w110 = q1.fu - w111, w101 = q1.fv - w111, w100 = 1.f - q1.fu - w101;
                tw1[0] = w100; tw1[1] = w110; tw1[2] = w101; tw1[3] = w111;
                sclu1 = (float)level_dim(p.texW, level1); sclv1 = (float)level_dim(p.texH, level1);
                if (second) slots_of(q1, level1, sl1);
            }
            // Texel values for the uv gradients: with a compile-time channel count all eight taps are fetched
            // here as whole texels (one vector load each, all in flight across the table work below) instead
            // of channel by channel inside the loop (24 dependent scalar gathers for RGB).
            constexpr bool kPrefetch = (C_CT > 0) && !CUBE;
            constexpr int CMAX = C_CT > 0 ? C_CT : 1;
            float ta[4][CMAX], tb[4][CMAX];
            if (kPrefetch) {
#pragma unroll
                for (int k = 0; k < 4; k++) load_texel<C_CT>(ta[k], pIn0, q0.tc[k], C);
                if (second) {
#pragma unroll
                    for (int k = 0; k < 4; k++) load_texel<C_CT>(tb[k], pIn1, q1.tc[k], C);
                }
            }
            const RunScan rs = run_of(q0, level0, q1, level1, second);
            int lost = 0;                                        // sign bit: some valid tap (tc >= 0) of this lane has no slot (< 0)
#pragma unroll
            for (int k = 0; k < 4; k++) lost |= (~q0.tc[k] & sl0[k]) | (second ? (~q1.tc[k] & sl1[k]) : 0);
            const bool anyLost = !(p.dbg & 131072) && __ballot(lost < 0) != 0ull;     // (131072: timing experiment without lookups and scatter)
            for (int c = 0; c < C; c++) {
                const float d = pDy[c];
                const float d0 = kTri ? (1.f - flevel) * d : d;
                const float d1 = second ? flevel * d : 0.f;
                float v0[4] = {tw0[0] * d0, tw0[1] * d0, tw0[2] * d0, tw0[3] * d0};
                float v1[4] = {tw1[0] * d1, tw1[1] * d1, tw1[2] * d1, tw1[3] * d1};
                if (CUBE) { corner_weights(q0, v0); if (kTri) corner_weights(q1, v1); }
                float z = 0.f;
                if (rs.any_merge()) {
                    if (kTri) { rs.scan3(v0[0], v0[1], v0[2]); rs.scan3(v0[3], v1[0], v1[1]); rs.scan3(v1[2], v1[3], z); }
                    else      { rs.scan3(v0[0], v0[1], v0[2]); rs.scan3(v0[3], z, z); }
                }
                if (p.dbg & 65536) {                                 // timing experiment: everything but the LDS adds
                } else if (rs.tail) {
#pragma unroll
                    for (int k = 0; k < 4; k++) scatter(sl0[k], c, v0[k]);
                    if (second) {
#pragma unroll
                        for (int k = 0; k < 4; k++) scatter(sl1[k], c, v1[k]);
                    }
                    if (anyLost) {
#pragma unroll
                        for (int k = 0; k < 4; k++) spill(sl0[k], level0, q0.tc[k], c, v0[k]);
                        if (second) {
#pragma unroll
                            for (int k = 0; k < 4; k++) spill(sl1[k], level1, q1.tc[k], c, v1[k]);
                        }
                    }
                }
                float a[4];
                if (kPrefetch) { a[0] = ta[0][c % CMAX]; a[1] = ta[1][c % CMAX]; a[2] = ta[2][c % CMAX]; a[3] = ta[3][c % CMAX]; }
                else fetch_quad(pIn0, q0, C, c, a);
                const float ad = (a[3] + a[0] - a[1] - a[2]);
                gu += d0 * ((a[1] - a[0]) + q0.fv * ad) * sclu0;
                gv += d0 * ((a[2] - a[0]) + q0.fu * ad) * sclv0;
                if (second) {
                    const float dd1 = flevel * d;
                    float b[4];
                    if (kPrefetch) { b[0] = tb[0][c % CMAX]; b[1] = tb[1][c % CMAX]; b[2] = tb[2][c % CMAX]; b[3] = tb[3][c % CMAX]; }
                    else fetch_quad(pIn1, q1, C, c, b);
                    const float bd = (b[3] + b[0] - b[1] - b[2]);
                    gu += dd1 * ((b[1] - b[0]) + q1.fv * bd) * sclu1;
                    gv += dd1 * ((b[2] - b[0]) + q1.fu * bd) * sclv1;
                    df += (bilerp1(b[0], b[1], b[2], b[3], q1.fu, q1.fv) - bilerp1(a[0], a[1], a[2], a[3], q0.fu, q0.fv)) * d;
                }
            }
            if (CUBE) {
                const float3 g3 = cube_grad(uv3, gu, gv);
                float* g = p.gradUV + pidx * 3;
                g[0] = kTri ? g3.x + dfdv.x * df : g3.x;
                g[1] = kTri ? g3.y + dfdv.y * df : g3.y;
                g[2] = kTri ? g3.z + dfdv.z * df : g3.z;
            } else {
                ((float2*)p.gradUV)[pidx] = make_float2(gu, gv);
            }
            if (kTri) {
                if (p.gradBias) p.gradBias[pidx] = df;
                if (!BIAS_ONLY && p.gradUVDA) {
                    const float4 dwf = make_float4(dw.x * df, dw.y * df, dw.z * df, dw.w * df);
                    if (CUBE) {
                        float3 g0, g1;
                        cube_grad4(uv3, dwf, g0, g1);
                        float2* g = (float2*)p.gradUVDA + pidx * 3;
                        g[0] = make_float2(g0.x, g1.x); g[1] = make_float2(g0.y, g1.y); g[2] = make_float2(g0.z, g1.z);
                    } else ((float4*)p.gradUVDA)[pidx] = dwf;
                }
            }
        }
    }
    if (direct || (p.dbg & 2048)) return;

    // ---- flush: consecutive lanes take consecutive (texel, channel) entries of a patch row ---------
    // Only the patches that were claimed are visited (a 16x16-pixel block touches a few dozen of the table's
    // patches, a background block one or two): scanning the whole table cost more instructions than the
    // rest of the kernel.
    __syncthreads();
    for (int g = threadIdx.x; g < groups; g += 256)
        if (tab.keys[g] != 0u) s_used[atomicAdd(&s_max[1], 1u)] = g;
    __syncthreads();
    const int perGroup = 16 * C;
    const int n = (int)s_max[1] * perGroup;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int u = i / perGroup;
        const int g = s_used[u];
        const uint32_t key = tab.keys[g];
        const int r = i - u * perGroup;
        const int t = tab.vals[g * perGroup + r];
        if (t == 0) continue;
        const int tx = r / C, c = r - tx * C;
        const int level = (int)(key >> 27) - 1;
        const int x = (int)(key & 0xFFFu) * 8 + (tx & 7);
        const int y = (int)((key >> 12) & 0x7FFFu) * 2 + (tx >> 3);
        const int w = level_dim(p.texW, level), h = level_dim(p.texH, level) * (CUBE ? 6 : 1);
        if (x >= w || y >= h) continue;                          // cannot happen: only valid texels are inserted
        atomic_add_f32(p.gradTex[level] + ((tz * h + y) * w + x) * C + c, fs.to_float(t));
    }
}

// ---- the heavy blocks of the common case, leaner (round 5) -------------------------------------------------------------
// k_tex_grad above is the general kernel: cube maps, every filter, any channel count -- 96 VGPRs with 20 B/lane of scratch, five
// waves per SIMD, a third of its static code exec-mask bookkeeping (0.73 ms = 0.19 of the HBM peak on config 3's heavy blocks).
// This one does the common case only -- 2-D texture, bilinear or trilinear filter with the level from uv_da, 1..4 channels, the
// blocks the light kernel left over -- around the same LDS patch table, and is written for registers:
//   * one mip level at a time (footprint, slots, four texels in flight, scatter, that level's share of the uv gradient) instead
//     of both levels' eight texels and two footprints live together;
//   * the level's adjoint with respect to uv_da is recomputed at the end from uv_da fetched again, so that neither lives across
//     the taps;
//   * pixels with IDENTICAL footprints (a constant-uv background inside a silhouette block) are found by comparing with the
//     wave's first and last active lane and summed with DPP -- not by a segmented scan over all eight taps' coordinates; a wave
//     that is one such group altogether leaves a record for k_tex_grad_fold like the light kernel's.
// Round 5 also tried what the review asked for, a DENSE texel window per level instead of the table (cell = (y - y0) * W + x - x0,
// no keys, no probes): 0.46 ms with the taps that fit, but on config 3's mesh -- jittered vertices, so the mip level changes from
// triangle to triangle -- a 16x16-pixel block holds 4.7 distinct levels on average (up to 11), and with 34x34 texels for the
// finest level, 18x18 and 10x10 for the next two, 45-64 % of the pixels had a tap outside (per-level corners and a 34x34 window
// for EVERY level still lose 7-15 %, at 14 KB per level); those taps went to memory as scattered f32 atomics, 3.7 ms.  The
// table adapts to where the taps are; a window does not (tools/exp_texwin_sim.py reproduces the count on the CPU).
// Where this kernel's 0.71 ms go on config 3 (stages switched off one at a time, r05): 0.11 ms loads / level selection / stores,
// 0.20 ms the eight 12-byte texel gathers and the uv gradient, 0.12 ms slot lookups and LDS adds, 0.27 ms the flush -- ~90 M
// f32 atomics (every block writes each texel it touched once: ~700 texels x 3 channels x 40 k blocks) at the rate coalesced
// runs of 24 lanes sustain.  What is left to gain is in the flush: larger blocks per table (fewer texels shared by neighbours).
__device__ __forceinline__ uint32_t wave_min_u32_to_last(uint32_t v) {                 // result valid in lane 63
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0xb1, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x4e, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x118, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x142, 0xa, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x143, 0xc, 0xf, false));
    return v;
}

template <int FILTER, int C>
__global__ __launch_bounds__(256, 6) void k_tex_grad_lean(const TexParams p, int groups)
{
    constexpr bool kTri = (FILTER == TEX_LML);
    extern __shared__ __attribute__((aligned(16))) unsigned char s_mem[];
    // LDS layout as in k_tex_grad: vals [groups*16*C] | keys [groups] | {block max, used count, -, -} | used-patch list [groups]
    PatchTable tab{(uint32_t*)((int*)s_mem + (size_t)groups * 16 * C), (int*)s_mem, groups, C};
    uint32_t* s_max = tab.keys + groups;
    int* s_used = (int*)(s_max + 4);
    int px = 0, py = 0, pz = 0, blk = 0; bool inside;
    if (!tex_pixel(p, px, py, pz, inside, &blk)) return;
    if (p.heavy && !p.heavy[blk]) return;                                    // finished by the light kernel
    tab.clear(threadIdx.x, 256);
    if (threadIdx.x == 0) { s_max[0] = 0u; s_max[1] = 0u; }
    const int lane = threadIdx.x & 63;
    const int tz = (p.texDepth == 1) ? 0 : pz;
    const size_t pidx = (size_t)px + (size_t)p.imgW * (py + (size_t)p.imgH * pz);
    const bool zt = inside && p.zflags.empty(pz, py, px);                    // uv = uv_da = 0 known for this tile: not read
    float2 uv = make_float2(0.f, 0.f);
    float4 da = make_float4(0.f, 0.f, 0.f, 0.f);
    float d[C];
#pragma unroll
    for (int c = 0; c < C; c++) d[c] = 0.f;
    if (inside) {
        const float* pDy = p.dy + pidx * C;
#pragma unroll
        for (int c = 0; c < C; c++) d[c] = pDy[c];
        if (!zt) {
            uv = ((const float2*)p.uv)[pidx];
            if (kTri) da = ((const float4*)p.uvDA)[pidx];
        }
    }
    __syncthreads();
    // ---- phase A: zero stores for pixels without an upstream gradient (:922-971); the block's largest |dy| ---------------
    bool active = false;
    float m = 0.f;
    {
        uint32_t dmax = 0u;
#pragma unroll
        for (int c = 0; c < C; c++) { dmax |= (uint32_t)__float_as_int(d[c]); m = max_abs_keep_nan(m, d[c]); }
        active = inside && !(__int_as_float((int)dmax) == 0.f);
        if (inside && !active) {
            ((float2*)p.gradUV)[pidx] = make_float2(0.f, 0.f);
            if (kTri) {
                if (p.gradUVDA) ((float4*)p.gradUVDA)[pidx] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.gradBias) p.gradBias[pidx] = 0.f;
            }
        }
        if (!active) m = 0.f;
    }
    block_max_update(s_max, m);
    int level0 = 0, level1 = 0; float flevel = 0.f;
    if (kTri && active) tex_mip_level<FILTER, false, false>(p, pidx, level0, level1, flevel, nullptr, make_float3(0.f, 0.f, 0.f), nullptr, zt, &da);
    __syncthreads();
    const uint32_t maxBits = s_max[0];
    if (maxBits == 0u) return;                                   // nobody has anything to scatter
    const bool direct = maxBits >= 0x7F800000u;                  // inf or NaN present: plain f32 atomics
    const FixedScale32 fs(direct ? 0x3F800000u : maxBits);

    // ---- identical footprints inside the wave ------------------------------------------------------------------------------
    // dsc[c] = what this lane scatters for channel c: its own upstream gradient; the group's total for a group's first lane;
    // nothing for the group's other lanes.
    const uint64_t act = __ballot(active);
    float dsc[C];
#pragma unroll
    for (int c = 0; c < C; c++) dsc[c] = d[c];
    bool scatters = active;
    bool uniformWave = false;
    {
        const float bias = (kTri && p.bias && inside) ? p.bias[pidx] : 0.f;
        auto group = [&](int leader, uint64_t& members) {
            bool same = active;
            same &= __float_as_int(uv.x) == __builtin_amdgcn_readlane(__float_as_int(uv.x), leader);
            same &= __float_as_int(uv.y) == __builtin_amdgcn_readlane(__float_as_int(uv.y), leader);
            if (kTri) {
                same &= __float_as_int(da.x) == __builtin_amdgcn_readlane(__float_as_int(da.x), leader);
                same &= __float_as_int(da.y) == __builtin_amdgcn_readlane(__float_as_int(da.y), leader);
                same &= __float_as_int(da.z) == __builtin_amdgcn_readlane(__float_as_int(da.z), leader);
                same &= __float_as_int(da.w) == __builtin_amdgcn_readlane(__float_as_int(da.w), leader);
                if (p.bias) same &= __float_as_int(bias) == __builtin_amdgcn_readlane(__float_as_int(bias), leader);
            }
            members = __ballot(same);
            if (__popcll(members) > 1) {
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const float tot = wave_sum_to_last(same ? d[c] : 0.f);
                    const float t = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), 63));
                    if (lane == leader) dsc[c] = t;
                }
                if (same && lane != leader) scatters = false;
            }
        };
        uint64_t mA = 0ull, mB = 0ull;
        if (act) {
            group(__builtin_ctzll(act), mA);
            const uint64_t rest = act & ~mA;
            if (rest) group(63 - __builtin_clzll(rest), mB);
            // a wave that is ONE group of pixels without a footprint: level 0, one quad -- a record for k_tex_grad_fold (a background
            // sends one total per wave to the same four texels from thousands of blocks: same-address f32 atomics execute one by one)
            if (mA == ~0ull && p.rec) {
                bool flat = !kTri || (da.x == 0.f && da.y == 0.f && da.z == 0.f && da.w == 0.f && !(fabsf(bias) == INFINITY));
                uniformWave = __ballot(flat) == ~0ull;
            }
        }
    }

    if (active) {
        const float* const* texp = p.tex;
        float gu = 0.f, gv = 0.f, s0 = 0.f, s1 = 0.f;
        const bool second = kTri && flevel > 0.f;
#pragma unroll
        for (int l = 0; l < (kTri ? 2 : 1); l++) {
            if (l == 1 && !second) break;
            const int level = l == 0 ? level0 : level1;
            const Quad q = tex_index_linear(p, uv.x, uv.y, tz, level);
            const float w11 = q.fu * q.fv, w10 = q.fu - w11, w01 = q.fv - w11, w00 = 1.f - q.fu - w01;
            const float tw[4] = {w00, w10, w01, w11};
            const float* pIn = texp[level];
            float ta[4][C];
#pragma unroll
            for (int k = 0; k < 4; k++) load_texel<C>(ta[k], pIn, q.tc[k], C);
            // table slots of the four taps (-1: to memory): the taps of a footprint share patches most of the time
            int cell[4] = {-1, -1, -1, -1};
            int lost = 0;                                        // sign bit: some valid tap of this lane has no slot
            if (scatters && !direct && !uniformWave) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (q.tc[k] < 0) continue;
                    bool reused = false;
#pragma unroll
                    for (int j = 0; j < k; j++) {
                        if (!reused && q.tc[j] >= 0 && cell[j] >= 0 && (q.tx[k] >> 3) == (q.tx[j] >> 3) && (q.ty[k] >> 1) == (q.ty[j] >> 1)) {
                            cell[k] = (cell[j] & ~15) + (q.ty[k] & 1) * 8 + (q.tx[k] & 7);
                            reused = true;
                        }
                    }
                    if (!reused) cell[k] = tab.find(level, q.tx[k], q.ty[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) lost |= (~q.tc[k] & cell[k]);
            const float lw = !kTri ? 1.f : l == 0 ? 1.f - flevel : flevel;       // this level's share
            const bool anyLost = !uniformWave && __ballot(scatters && lost < 0) != 0ull;
            if (scatters && !uniformWave) {
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const float dl = dsc[c] * lw;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (cell[k] >= 0) atomicAdd(&tab.vals[cell[k] * C + c], fs.to_fixed(tw[k] * dl));
                }
            }
            if (anyLost && scatters) {
                float* g = p.gradTex[level];
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (q.tc[k] >= 0 && cell[k] < 0) {
#pragma unroll
                        for (int c = 0; c < C; c++) atomic_add_f32(g + q.tc[k] * C + c, tw[k] * (dsc[c] * lw));
                    }
            }
            // uv gradient of this level (texture_kernel.cu:1046-1094), with the pixel's OWN upstream gradient
            const float sclu = (float)level_dim(p.texW, level), sclv = (float)level_dim(p.texH, level);
#pragma unroll
            for (int c = 0; c < C; c++) {
                const float a0 = ta[0][c], a1 = ta[1][c], a2 = ta[2][c], a3 = ta[3][c];
                const float ad = (a3 + a0 - a1 - a2);
                const float dl = kTri ? (l == 0 ? (1.f - flevel) * d[c] : flevel * d[c]) : d[c];
                gu += dl * ((a1 - a0) + q.fv * ad) * sclu;
                gv += dl * ((a2 - a0) + q.fu * ad) * sclv;
                if (kTri) { const float b = bilerp1(a0, a1, a2, a3, q.fu, q.fv) * d[c]; if (l == 0) s0 += b; else s1 += b; }
            }
        }
        ((float2*)p.gradUV)[pidx] = make_float2(gu, gv);
        if (kTri) {
            const float df = second ? s1 - s0 : 0.f;
            if (p.gradBias) p.gradBias[pidx] = df;
            if (p.gradUVDA) {
                float4 dw = make_float4(0.f, 0.f, 0.f, 0.f);
                int l0, l1; float fl;
                const float4 da2 = zt ? make_float4(0.f, 0.f, 0.f, 0.f) : ((const float4*)p.uvDA)[pidx];
                tex_mip_level<FILTER, false, false>(p, pidx, l0, l1, fl, &dw, make_float3(0.f, 0.f, 0.f), nullptr, zt, &da2);
                ((float4*)p.gradUVDA)[pidx] = make_float4(dw.x * df, dw.y * df, dw.z * df, dw.w * df);
            }
        }
    }
    if (uniformWave) {
        // one record for the wave (the totals are with the group's first lane, lane 0)
        const Quad q0 = tex_index_linear(p, uv.x, uv.y, tz, 0);
        const float w011 = q0.fu * q0.fv, w010 = q0.fu - w011, w001 = q0.fv - w011, w000 = 1.f - q0.fu - w001;
        const float tw0[4] = {w000, w010, w001, w011};
        int* recBase = p.rec + (blk * 4 + (int)(threadIdx.x >> 6));
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int tot = __builtin_amdgcn_readfirstlane(__float_as_int(dsc[c]));
            if (lane == 63) recBase[(size_t)(kTexRecHeader + c) * p.nrec] = tot;
        }
        if (lane == 63) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                recBase[(size_t)(4 + k) * p.nrec] = __float_as_int(tw0[k]);
                if (k > 0) recBase[(size_t)k * p.nrec] = q0.tc[k];
            }
            // tap 0's index makes the record valid; a record whose FIRST tap has no texel carries the first valid tap in front
            int first = q0.tc[0];
            if (first < 0) {
#pragma unroll
                for (int k = 1; k < 4; k++) {
                    if (first < 0 && q0.tc[k] >= 0) {
                        first = q0.tc[k];
                        recBase[(size_t)4 * p.nrec] = __float_as_int(tw0[k]);
                        recBase[(size_t)k * p.nrec] = -1;
                    }
                }
            }
            recBase[0] = first;
        }
    }
    if (direct) return;

    // ---- flush: consecutive lanes take consecutive (texel, channel) entries of a patch row (as k_tex_grad) -----------------
    __syncthreads();
    for (int g = threadIdx.x; g < groups; g += 256)
        if (tab.keys[g] != 0u) s_used[atomicAdd(&s_max[1], 1u)] = g;
    __syncthreads();
    constexpr int perGroup = 16 * C;
    const int n = (int)s_max[1] * perGroup;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int u = i / perGroup;
        const int g = s_used[u];
        const uint32_t key = tab.keys[g];
        const int r = i - u * perGroup;
        const int t = tab.vals[g * perGroup + r];
        if (t == 0) continue;
        const int tx = r / C, c = r - tx * C;
        const int level = (int)(key >> 27) - 1;
        const int x = (int)(key & 0xFFFu) * 8 + (tx & 7);
        const int y = (int)((key >> 12) & 0x7FFFu) * 2 + (tx >> 3);
        const int w = level_dim(p.texW, level), h = level_dim(p.texH, level);
        if (x >= w || y >= h) continue;                          // cannot happen: only valid texels are inserted
        atomic_add_f32(p.gradTex[level] + ((tz * h + y) * w + x) * C + c, fs.to_float(t));
    }
}

// First kernel of the two-kernel gradient pass (2-D textures, bilinear footprints, caller scratch): everything that needs no
// scatter machinery, at the occupancy of a streaming kernel.  k_tex_grad carries 96 VGPRs and a 26 KB table for its general
// path, so five waves per SIMD is all a background pixel gets there -- and three quarters of a rendered image's pixels are
// background (one texel quad, zero footprint: the uniform-wave path) or carry no upstream gradient at all (zero stores).
// Here a 16x16-pixel block whose four waves are ALL of those two kinds is finished -- zero stores, the uv gradient of the
// constant quad, one record per uniform wave for k_tex_grad_fold -- and marked 0 in `heavy`; a block with any other wave
// (real footprints, partly active waves, inf / NaN gradients) is marked 1 and left entirely to k_tex_grad, untouched.
template <int FILTER, int C_CT>
__global__ __launch_bounds__(256, 8) void k_tex_grad_light(const TexParams p)
{
    __shared__ int s_heavy;
    int px = 0, py = 0, pz = 0, blk = 0; bool inside;
    if (!tex_pixel<false>(p, px, py, pz, inside, &blk)) return;      // every block costs the same here: image order (longer contiguous rows)
    if (threadIdx.x == 0) s_heavy = 0;
    __syncthreads();
    const int C = C_CT > 0 ? C_CT : p.channels;
    constexpr int CMAX = C_CT > 0 ? C_CT : 1;
    const int lane = threadIdx.x & 63;
    const int tz = (p.texDepth == 1) ? 0 : pz;
    const size_t pidx = (size_t)px + (size_t)p.imgW * (py + (size_t)p.imgH * pz);
    const float* pDy = p.dy + pidx * C;
    const bool zt = inside && p.zflags.empty(pz, py, px);

    bool active = false, finite = true;
    float dreg[CMAX];
    if (inside) {
        uint32_t dmax = 0u;
        for (int c = 0; c < C; c++) {
            const float d = pDy[c];
            if (C_CT > 0) dreg[c % CMAX] = d;
            dmax |= (uint32_t)__float_as_int(d);
            finite = finite && (fabsf(d) < INFINITY);                       // false for inf and NaN
        }
        active = !(__int_as_float((int)dmax) == 0.f);
    }
    const uint64_t am = __ballot(active);
    bool light = (am == 0ull);                                               // nothing to scatter in this wave
    float2 t = make_float2(0.f, 0.f);
    if (am == ~0ull) {                                                       // every pixel active: the uniform-wave test of k_tex_grad
        t = zt ? make_float2(0.f, 0.f) : ((const float2*)p.uv)[pidx];
        const int ux = __float_as_int(t.x), uy = __float_as_int(t.y);
        bool same = (ux == __builtin_amdgcn_readfirstlane(ux)) & (uy == __builtin_amdgcn_readfirstlane(uy));
        if (FILTER != TEX_LINEAR) {
            const float4 d = zt ? make_float4(0.f, 0.f, 0.f, 0.f) : ((const float4*)p.uvDA)[pidx];
            same &= (d.x == 0.f) & (d.y == 0.f) & (d.z == 0.f) & (d.w == 0.f);
            if (p.bias) same &= !(fabsf(p.bias[pidx]) == INFINITY);
        }
        light = __ballot(same) == ~0ull;
    }
    if (__ballot(!finite) != 0ull) light = false;
    if (!light && lane == 0) s_heavy = 1;
    __syncthreads();
    const bool heavy = s_heavy != 0;
    if (threadIdx.x == 0) p.heavy[blk] = heavy ? 1 : 0;
    if (heavy) return;

    if (am == 0ull) {                                                        // explicit zeros (texture_kernel.cu:922-971)
        if (inside) {
            ((float2*)p.gradUV)[pidx] = make_float2(0.f, 0.f);
            if (FILTER == TEX_LML) {
                if (p.gradUVDA) ((float4*)p.gradUVDA)[pidx] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.gradBias) p.gradBias[pidx] = 0.f;
            }
        }
        return;
    }
    // uniform wave (all 64 lanes active, hence inside): as in k_tex_grad, with the totals leaving as a record
    const Quad q0 = tex_index_linear(p, t.x, t.y, tz, 0);
    const float* pIn0 = p.tex[0];
    const float w011 = q0.fu * q0.fv, w010 = q0.fu - w011, w001 = q0.fv - w011, w000 = 1.f - q0.fu - w001;
    const float tw0[4] = {w000, w010, w001, w011};
    const float sclu0 = (float)p.texW, sclv0 = (float)p.texH;
    int* recBase = p.rec + (blk * 4 + (int)(threadIdx.x >> 6));
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            recBase[(size_t)(4 + k) * p.nrec] = __float_as_int(tw0[k]);
            if (k > 0) recBase[(size_t)k * p.nrec] = q0.tc[k];
        }
    }
    float gu = 0.f, gv = 0.f;
    for (int c = 0; c < C; c++) {
        const float d = C_CT > 0 ? dreg[c % CMAX] : pDy[c];
        const float tot = wave_sum_to_last(d);                               // valid in lane 63
        if (lane == 63) recBase[(size_t)(kTexRecHeader + c) * p.nrec] = __float_as_int(tot);
        float a[4];
        fetch_quad(pIn0, q0, C, c, a);
        const float ad = (a[3] + a[0] - a[1] - a[2]);
        gu += d * ((a[1] - a[0]) + q0.fv * ad) * sclu0;
        gv += d * ((a[2] - a[0]) + q0.fu * ad) * sclv0;
    }
    if (lane == 63) {
        int first = q0.tc[0];
        if (first < 0) {
#pragma unroll
            for (int k = 1; k < 4; k++) {
                if (first < 0 && q0.tc[k] >= 0) {
                    first = q0.tc[k];
                    recBase[(size_t)4 * p.nrec] = __float_as_int(tw0[k]);
                    recBase[(size_t)k * p.nrec] = -1;
                }
            }
        }
        recBase[0] = first;
    }
    ((float2*)p.gradUV)[pidx] = make_float2(gu, gv);
    if (FILTER == TEX_LML) {
        if (p.gradBias) p.gradBias[pidx] = 0.f;
        if (p.gradUVDA) ((float4*)p.gradUVDA)[pidx] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// The same pass with ONE WAVE per 16x16-pixel block (compile-time channel counts): the wave walks the block's four 8x8 tiles in
// stages -- four flags, then the four tiles' upstream gradients (4 x C loads in flight per lane instead of C), then, only
// for tiles that need them, uv / uv_da -- and votes on the block by itself: no LDS, no barrier.  The kernel above is bound by
// its chain of dependent loads with one tile per wave (0.33 ms for 1.05 GB at config 3, 3 TB/s); this form has four times the
// bytes in flight per wave at the same occupancy.  A workgroup is four waves = four blocks side by side (64 x 16 pixels).
// Records, zero stores and uv gradients are those of the kernel above, tile for tile (record number = block * 4 + tile).
template <int FILTER, int C>
__global__ __launch_bounds__(256, 8) void k_tex_grad_light_w(const TexParams p, int gx4, int exp)
{
    int bx4, by, pz;
    if (!decode_block(gx4, p.tilesY, p.n, bx4, by, pz)) return;
    const int lane = threadIdx.x & 63;
    const int bx = bx4 * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (bx >= p.tilesX) return;
    const int blk = (pz * p.tilesY + by) * p.tilesX + bx;
    const int tz = (p.texDepth == 1) ? 0 : pz;
    const int px0 = bx * 16 + (lane & 7), py0 = by * 16 + (lane >> 3);
    // Pixel numbers fit 32 bits here and so do their byte offsets into every per-pixel tensor (the host checks: 16 bytes per
    // pixel at most), so every access is base pointer + one 32-bit lane offset.  This kernel is bound by its instruction count
    // -- a streaming pass over 33 M pixels pays 0.85 us per vector instruction per wave of pixels at config 3 -- and 64-bit
    // index arithmetic, per-tile wave reductions and per-lane copies of wave-uniform values were most of the 1000 it had.
    const uint32_t pidx0 = (uint32_t)px0 + (uint32_t)p.imgW * ((uint32_t)py0 + (uint32_t)p.imgH * (uint32_t)pz);
    auto pix = [&](int t) { return pidx0 + (uint32_t)((t & 1) * 8) + (uint32_t)((t >> 1) * 8) * (uint32_t)p.imgW; };
    auto at = [](const void* base, uint32_t bytes) { return (const char*)base + bytes; };

    bool inside[4], ztile[4];
#pragma unroll
    for (int t = 0; t < 4; t++) inside[t] = px0 + (t & 1) * 8 < p.imgW && py0 + (t >> 1) * 8 < p.imgH;
    float d[4][C];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const float* pDy = (const float*)at(p.dy, (inside[t] ? pix(t) : 0u) * (uint32_t)(4 * C));   // (a pixel beyond the image reads pixel 0's and ignores it)
#pragma unroll
        for (int c = 0; c < C; c++) d[t][c] = pDy[c];
    }
    {
        // the block's four flags: two bytes in each of two rows of the flag array, each pair inside one dword (the block's first
        // tile column is even; the array is 16-byte aligned and padded) -- two scalar loads in flight together
        const int tx8 = bx * 2, ty8 = by * 2;
        uint32_t w0 = 0x01010101u, w1 = 0x01010101u;                               // (no flags, or beyond the array: "not empty")
        const bool haveF = p.zflags.f != nullptr && tx8 < p.zflags.w;
        const size_t f0 = ((size_t)pz * p.zflags.h + ty8) * p.zflags.w + tx8, f1 = f0 + p.zflags.w;
        const bool row0 = haveF && ty8 < p.zflags.h, row1 = haveF && ty8 + 1 < p.zflags.h;
        const uint32_t* wp0 = (const uint32_t*)p.zflags.f + (f0 >> 2);
        const uint32_t* wp1 = (const uint32_t*)p.zflags.f + (f1 >> 2);
        if (row0) w0 = scalar_load(wp0);
        if (row1) w1 = scalar_load(wp1);
        const uint32_t s0 = (uint32_t)(f0 & 3) * 8u, s1 = (uint32_t)(f1 & 3) * 8u;
        const bool col1 = tx8 + 1 < p.zflags.w;
        ztile[0] = row0 && ((w0 >> s0) & 0xFFu) == 0u;
        ztile[1] = row0 && col1 && ((w0 >> (s0 + 8u)) & 0xFFu) == 0u;
        ztile[2] = row1 && ((w1 >> s1) & 0xFFu) == 0u;
        ztile[3] = row1 && col1 && ((w1 >> (s1 + 8u)) & 0xFFu) == 0u;
        if (haveF && (p.zflags.w & 1)) {                                           // odd row length: a pair may straddle two dwords -- byte by byte
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int x8 = tx8 + (t & 1), y8 = ty8 + (t >> 1);
                ztile[t] = x8 < p.zflags.w && y8 < p.zflags.h && p.zflags.f[((size_t)pz * p.zflags.h + y8) * p.zflags.w + x8] == 0;
            }
        }
    }
    uint64_t am[4];
    uint32_t fin = 0u;                                                             // OR of the magnitudes' exponent tests
#pragma unroll
    for (int t = 0; t < 4; t++) {
        uint32_t dmax = 0u;
#pragma unroll
        for (int c = 0; c < C; c++) {
            if (!inside[t]) d[t][c] = 0.f;
            dmax |= (uint32_t)__float_as_int(d[t][c]);
            fin = max(fin, (uint32_t)__float_as_int(d[t][c]) & 0x7FFFFFFFu);       // >= 0x7F800000: inf or NaN
        }
        am[t] = __ballot((dmax << 1) != 0u);                                      // some channel is not +-0
    }
    bool light = __ballot(fin >= 0x7F800000u) == 0ull;
    // tiles whose pixels are all active: the uniform-wave test of k_tex_grad (uv / uv_da only where the flags do not already say zero)
    float2 uvt[4];
    float4 dat[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        uvt[t] = make_float2(0.f, 0.f); dat[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (am[t] == ~0ull && !ztile[t]) {
            uvt[t] = *(const float2*)at(p.uv, pix(t) * 8u);
            if (FILTER != TEX_LINEAR) dat[t] = *(const float4*)at(p.uvDA, pix(t) * 16u);
        }
    }
    int su[4], sv[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        su[t] = __builtin_amdgcn_readfirstlane(__float_as_int(uvt[t].x)); sv[t] = __builtin_amdgcn_readfirstlane(__float_as_int(uvt[t].y));
        if (am[t] == 0ull || ztile[t]) continue;                                  // (a flagged tile's uv is zero by construction)
        if (am[t] != ~0ull) { light = false; continue; }
        bool same = (__float_as_int(uvt[t].x) == su[t]) && (__float_as_int(uvt[t].y) == sv[t]);
        if (FILTER != TEX_LINEAR) {
            same &= (dat[t].x == 0.f) & (dat[t].y == 0.f) & (dat[t].z == 0.f) & (dat[t].w == 0.f);
            if (p.bias) same &= !(fabsf(*(const float*)at(p.bias, pix(t) * 4u)) == INFINITY);
        }
        if (__ballot(same) != ~0ull) light = false;
    }
#pragma unroll
    for (int t = 0; t < 4; t++) if (am[t] != 0ull && am[t] != ~0ull) light = false;    // (a partly active tile, flagged or not)
    if (FILTER != TEX_LINEAR && p.bias) {
#pragma unroll
        for (int t = 0; t < 4; t++)                                               // flagged tiles: the bias may still be infinite (-inf + inf)
            if (light && am[t] == ~0ull && ztile[t] && __ballot(fabsf(*(const float*)at(p.bias, pix(t) * 4u)) == INFINITY) != 0ull) light = false;
    }
    if (lane == 0) p.heavy[blk] = light ? 0 : 1;
    if (!light) return;

    auto store_out = [&](int t, float2 g) {
        if ((exp & 1) && g.x != 12345.f) return;
        *(float2*)at(p.gradUV, pix(t) * 8u) = g;
        if (FILTER == TEX_LML) {
            if (p.gradBias) *(float*)at(p.gradBias, pix(t) * 4u) = 0.f;
            if (p.gradUVDA) *(float4*)at(p.gradUVDA, pix(t) * 16u) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if ((am[0] | am[1] | am[2] | am[3]) == 0ull) {
        // no upstream gradient anywhere in the block: explicit zeros (texture_kernel.cu:922-971)
#pragma unroll
        for (int t = 0; t < 4; t++) if (inside[t]) store_out(t, make_float2(0.f, 0.f));
        return;
    }

    // The constant quad: texel indices, weights, and per channel the two texel differences the uv gradient needs.  Wave-uniform
    // throughout: the texels come through scalar loads, and the quad is recomputed only when a tile's uv differs from its
    // predecessor's (a background's tiles all sample one place).
    const float* pIn0 = p.tex[0];
    const float sclu0 = (float)p.texW, sclv0 = (float)p.texH;
    int qu = 0, qv = 0, qtc[4] = {-1, -1, -1, -1};
    float qw[4] = {0.f, 0.f, 0.f, 0.f}, xu[C], xv[C];
    bool have = false;
    auto quad_for = [&](int u, int v) {
        if (have && u == qu && v == qv) return;
        have = true; qu = u; qv = v;
        const Quad q0 = tex_index_linear(p, __int_as_float(u), __int_as_float(v), tz, 0);
        const float w011 = q0.fu * q0.fv, w010 = q0.fu - w011, w001 = q0.fv - w011, w000 = 1.f - q0.fu - w001;
        qw[0] = w000; qw[1] = w010; qw[2] = w001; qw[3] = w011;
        float qa[4][C];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            qtc[k] = __builtin_amdgcn_readfirstlane(q0.tc[k]);
            // wave-uniform address: scalar loads (nvdr_device.hpp scalar_load); a tap without a texel reads texel 0 and is zeroed
            const float* tp = pIn0 + (size_t)max(qtc[k], 0) * C;
#pragma unroll
            for (int c = 0; c < C; c++) { const float t = scalar_load(tp + c); qa[k][c] = qtc[k] >= 0 ? t : 0.f; }
        }
#pragma unroll
        for (int c = 0; c < C; c++) {
            const float ad = (qa[3][c] + qa[0][c] - qa[1][c] - qa[2][c]);
            xu[c] = (qa[1][c] - qa[0][c]) + q0.fv * ad;
            xv[c] = (qa[2][c] - qa[0][c]) + q0.fu * ad;
        }
    };
    // one record, one word per lane, in ONE store: words 0..3 texel indices (a record whose FIRST tap has no texel -- boundary mode
    // zero -- is stored with the first valid tap moved to the front; -1 in word 0 = no tap has one), 4..7 weights, 8.. the summed
    // upstream gradient per channel
    auto put_record = [&](int slot, const float* tot) {
        int first = qtc[0], fk = 0;
#pragma unroll
        for (int k = 1; k < 4; k++) if (first < 0 && qtc[k] >= 0) { first = qtc[k]; fk = k; }
        int word = first;
#pragma unroll
        for (int k = 1; k < 4; k++) if (lane == k) word = (fk == k) ? -1 : qtc[k];
        if (lane == 4) word = __float_as_int(fk == 0 ? qw[0] : fk == 1 ? qw[1] : fk == 2 ? qw[2] : qw[3]);
#pragma unroll
        for (int k = 1; k < 4; k++) if (lane == 4 + k) word = __float_as_int(qw[k]);
#pragma unroll
        for (int c = 0; c < C; c++) if (lane == kTexRecHeader + c) word = __float_as_int(tot[c]);
        if (lane < kTexRecHeader + C) p.rec[(size_t)lane * p.nrec + slot] = word;
    };
    auto grad_of = [&](int t) {
        float2 g = make_float2(0.f, 0.f);
#pragma unroll
        for (int c = 0; c < C; c++) { g.x += d[t][c] * xu[c] * sclu0; g.y += d[t][c] * xv[c] * sclv0; }
        return g;
    };

    const bool oneQuad = (am[0] & am[1] & am[2] & am[3]) == ~0ull && su[1] == su[0] && su[2] == su[0] && su[3] == su[0]
                                                                  && sv[1] == sv[0] && sv[2] == sv[0] && sv[3] == sv[0];
    if (oneQuad && !(exp & 4)) {
        // the whole block samples one place (the rule in a background): ONE record for its 256 pixels, in the first tile's slot
        // (the other three stay "no record", as the host initialised them)
        quad_for(su[0], sv[0]);
        float tot[C];
#pragma unroll
        for (int c = 0; c < C; c++)
            tot[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_to_last((d[0][c] + d[1][c]) + (d[2][c] + d[3][c]))), 63));
        put_record(blk * 4, tot);
#pragma unroll
        for (int t = 0; t < 4; t++) store_out(t, grad_of(t));
        return;
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if (!inside[t]) continue;
        float2 g = make_float2(0.f, 0.f);
        if (am[t] != 0ull && !(exp & 4)) {
            // uniform tile (all 64 pixels active, hence inside): as in k_tex_grad, with the totals leaving as a record
            quad_for(su[t], sv[t]);
            float tot[C];
#pragma unroll
            for (int c = 0; c < C; c++) tot[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_to_last(d[t][c])), 63));
            put_record(blk * 4 + t, tot);
            g = grad_of(t);
        }
        store_out(t, g);
    }
}

// Second level of the gradient reduction for constant-uv regions: merges the per-wave records of k_tex_grad (TexParams::rec)
// and adds the totals to level 0 of the gradient texture.  One wave takes kFoldPerWave consecutive records, each lane
// kFoldPerLane of them (strided by 64, so every load instruction is coalesced; all loads of a lane are independent and in
// flight together).  A lane adds weight x total of the records that carry the same texel quad as its first one into
// registers -- the rule: neighbouring waves of a background all do -- and sends the others (region borders) to memory one
// by one; if then all lanes of the wave hold the same quad, their sums are combined over the lanes (DPP) and ONE lane
// issues the 4 x C atomics, else every lane issues its own.  Order of the f32 sums: fixed by the launch geometry.
constexpr int kFoldPerLane = 8;
constexpr int kFoldPerWave = 64 * kFoldPerLane;

template <int C_CT>
__global__ __launch_bounds__(256) void k_tex_grad_fold(const int* __restrict__ rec, int nrec, int channels, float* __restrict__ gradTex)
{
    const int C = C_CT > 0 ? C_CT : channels;
    constexpr int CMAX = C_CT > 0 ? C_CT : 8;                 // generic instantiation: up to 8 channels in registers, more record by record
    // third level, inside the workgroup: waves that end up with the same quad hand their sums to wave 0 through LDS, so a
    // background costs 4 x C atomics per 2048 records (same-address f32 atomics execute one after another at the memory
    // side: with one set per wave they were the whole run time of this kernel)
    __shared__ int s_quad[4][4];
    __shared__ float s_sum[4][4][CMAX];
    const int lane = threadIdx.x & 63;
    const int wid = (int)(threadIdx.x >> 6);
    const int wave = (int)blockIdx.x * 4 + wid;
    const int begin = wave * kFoldPerWave;
    const bool regs = C <= CMAX;
    int mine[4] = {-1, -1, -1, -1};                           // this lane's quad (its first valid record's)
    float acc[4][CMAX];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int c = 0; c < CMAX; c++) acc[k][c] = 0.f;
    int t[kFoldPerLane][4];
    float w[kFoldPerLane][4];
#pragma unroll
    for (int i = 0; i < kFoldPerLane; i++) {
        const int r = begin + i * 64 + lane;
        const bool in = r < nrec;
        t[i][0] = in ? rec[r] : -1;
#pragma unroll
        for (int k = 1; k < 4; k++) t[i][k] = in ? rec[(size_t)k * nrec + r] : -1;
#pragma unroll
        for (int k = 0; k < 4; k++) w[i][k] = in ? __int_as_float(rec[(size_t)(4 + k) * nrec + r]) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < kFoldPerLane; i++) {
        if (t[i][0] < 0) continue;                            // no record (the wave was not uniform) / no texel at all
        const int r = begin + i * 64 + lane;
        if (mine[0] < 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) mine[k] = t[i][k];
        }
        const bool same = regs & (t[i][0] == mine[0]) & (t[i][1] == mine[1]) & (t[i][2] == mine[2]) & (t[i][3] == mine[3]);
        for (int c = 0; c < C; c++) {
            const float tot = __int_as_float(rec[(size_t)(kTexRecHeader + c) * nrec + r]);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float v = w[i][k] * tot;
                if (same) { if (c < CMAX) acc[k][c < CMAX ? c : 0] += v; }
                else if (t[i][k] >= 0) atomic_add_f32(gradTex + (size_t)t[i][k] * C + c, v);
            }
        }
    }
    // combine over the lanes when the whole wave holds one quad (lanes without any record count as agreeing)
    const bool have = mine[0] >= 0;
    const uint64_t hm = __ballot(have);
    const int src = hm ? __builtin_ctzll(hm) : 0;
    int f[4];
#pragma unroll
    for (int k = 0; k < 4; k++) f[k] = hm ? __builtin_amdgcn_readlane(mine[k], src) : -1;
    const bool agree = !have || ((mine[0] == f[0]) & (mine[1] == f[1]) & (mine[2] == f[2]) & (mine[3] == f[3]));
    const bool waveUniform = hm != 0ull && __ballot(agree) == ~0ull;
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < 4; k++) s_quad[wid][k] = waveUniform ? f[k] : -2;        // -2: nothing to hand over
    }
    if (waveUniform) {
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int c = 0; c < CMAX; c++) {
                const float sum = wave_sum_to_last((have && c < C) ? acc[k][c] : 0.f);       // valid in lane 63
                if (lane == 63) s_sum[wid][k][c] = sum;
            }
    } else if (have) {
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int c = 0; c < CMAX; c++)
                if (c < C && mine[k] >= 0 && acc[k][c] != 0.f) atomic_add_f32(gradTex + (size_t)mine[k] * C + c, acc[k][c]);
    }
    __syncthreads();
    // wave 0: one lane per (tap, channel); a wave's sums join those of the first wave with the same quad
    if (wid == 0 && lane < 4 * CMAX) {
        const int k = lane / CMAX, c = lane - k * CMAX;
        bool done[4] = {false, false, false, false};
#pragma unroll
        for (int a = 0; a < 4; a++) {
            if (done[a] || s_quad[a][0] == -2) continue;
            float sum = s_sum[a][k][c];
#pragma unroll
            for (int b = a + 1; b < 4; b++) {
                if (!done[b] && s_quad[b][0] == s_quad[a][0] && s_quad[b][1] == s_quad[a][1] && s_quad[b][2] == s_quad[a][2] && s_quad[b][3] == s_quad[a][3]) {
                    sum += s_sum[b][k][c];
                    done[b] = true;
                }
            }
            const int texel = s_quad[a][k];
            if (c < C && texel >= 0 && sum != 0.f) atomic_add_f32(gradTex + (size_t)texel * C + c, sum);
        }
    }
}

// ---- mip construction / mip gradient pull ------------------------------------------------------

struct MipParams { const float* in; float* out; int wi, hi, wo, ho, depth, C; };

// One output element (texel x channel) of a mip level: the 2x2 box average of the level above, 1x2 / 2x1 once an
// extent has reached 1 (texture_kernel.cu:644-704).
// I = index type: 64-bit for the large levels, 32-bit where the level above is known to be small (64-bit divisions
// are ~100 instructions each).
template <typename I>
__device__ __forceinline__ float mip_element(const MipParams& p, I i)
{
    const int c = (int)(i % (I)p.C);
    I t = i / (I)p.C;
    const int x = (int)(t % (I)p.wo); t /= (I)p.wo;
    const int y = (int)(t % (I)p.ho);
    const int z = (int)(t / (I)p.ho);
    const float* in = p.in;
    if (p.wi == 1 || p.hi == 1) {                          // one extent already 1: average the two remaining texels
        const I i0 = (p.hi == 1) ? ((I)z * p.hi * p.wi + (I)2 * x) : ((I)z * p.hi * p.wi + (I)2 * y * p.wi);
        const I i1 = (p.hi == 1) ? i0 + 1 : i0 + p.wi;
        return .5f * (in[i0 * p.C + c] + in[i1 * p.C + c]);
    }
    const I i0 = ((I)z * p.hi + (I)2 * y) * p.wi + (I)2 * x;
    const float v0 = in[i0 * p.C + c], v1 = in[(i0 + 1) * p.C + c];
    const float v2 = in[(i0 + p.wi) * p.C + c], v3 = in[(i0 + p.wi + 1) * p.C + c];
    return .25f * (((v0 + v1) + v2) + v3);
}

// One lane per output TEXEL (all its channels), addressed by the launch grid: (64 texels, 4 rows, slice) per workgroup.
__global__ __launch_bounds__(256) void k_mip_build(const MipParams p)
{
    const int x = (int)blockIdx.x * 64 + (int)(threadIdx.x & 63);
    const int y = (int)blockIdx.y * 4 + (int)(threadIdx.x >> 6);
    const int z = (int)blockIdx.z;
    if (x >= p.wo || y >= p.ho) return;
    float* out = p.out + (((size_t)z * p.ho + y) * p.wo + x) * p.C;
    if (p.wi == 1 || p.hi == 1) {                          // one extent already 1: average the two remaining texels
        const size_t i0 = (p.hi == 1) ? ((size_t)z * p.hi * p.wi + 2 * (size_t)x) : ((size_t)z * p.hi * p.wi + 2 * (size_t)y * p.wi);
        const size_t i1 = (p.hi == 1) ? i0 + 1 : i0 + p.wi;
        for (int c = 0; c < p.C; c++) out[c] = .5f * (p.in[i0 * p.C + c] + p.in[i1 * p.C + c]);
        return;
    }
    const size_t i0 = ((size_t)z * p.hi + 2 * (size_t)y) * p.wi + 2 * (size_t)x;
    const float* a = p.in + i0 * p.C;
    const float* b = a + (size_t)p.wi * p.C;
    for (int c = 0; c < p.C; c++) out[c] = .25f * (((a[c] + a[p.C + c]) + b[c]) + b[p.C + c]);
}

// The small levels at the end of the chain (a few thousand elements and fewer, each depending on the one before) in
// ONE launch of one workgroup: a launch per level is 6-8 us of latency for microseconds of work.
struct MipTailParams { const float* in; float* mip; long long off[kTexMaxLevels]; int w[kTexMaxLevels], h[kTexMaxLevels]; int first, last, depth, C; };

constexpr int kMipTailElems  = 12288;                      // capacity of the tail's first level (elements = texels x channels x slices)
constexpr int kMipTailElems2 = kMipTailElems / 4;          // ... and of its second level

// First level of the tail: the earliest l such that level l fits the first LDS buffer and level l + 1 (if any) the
// second; L + 1 = no tail.  Thin textures (one extent already 1) halve per level instead of quartering, so their tail
// starts later than that of a square texture with the same element count.
static int mip_tail_start(const int* lw, const int* lh, int L, long long per_texel)
{
    auto elems = [&](int l) { return (long long)lw[l] * lh[l] * per_texel; };
    int tail = 1;
    while (tail <= L && !(elems(tail) <= kMipTailElems && (tail == L || elems(tail + 1) <= kMipTailElems2))) tail++;
    if (tail >= L) tail = L + 1;                              // a tail of one level gains nothing
    return tail;
}

__global__ __launch_bounds__(1024) void k_mip_build_tail(const MipTailParams q)
{
    // Every level is written to memory AND kept in LDS for the next one, two buffers alternating: no memory round trip
    // between the levels.  A level is a quarter of the one before while both extents shrink but only HALF of it once one
    // extent has reached 1 (texture.cpp:77-98), so the host admits a chain into the tail only if its first level fits
    // s_a and its second fits s_b (tail_start); every later level is at most half of the level two steps before it.
    __shared__ float s_a[kMipTailElems];
    __shared__ float s_b[kMipTailElems2];
    for (int l = q.first; l <= q.last; l++) {
        float* keep = ((l - q.first) & 1) ? s_b : s_a;
        const float* prev = ((l - q.first) & 1) ? s_a : s_b;
        MipParams p;
        p.in = (l == q.first) ? q.in : prev;
        p.out = q.mip + q.off[l];
        p.wi = q.w[l - 1]; p.hi = q.h[l - 1]; p.wo = q.w[l]; p.ho = q.h[l]; p.depth = q.depth; p.C = q.C;
        const int total = p.wo * p.ho * p.depth * p.C;
        for (int i = threadIdx.x; i < total; i += 1024) { const float v = mip_element<int>(p, i); p.out[i] = v; keep[i] = v; }      // the level above has <= 4 * kMipTailElems elements
        __syncthreads();
    }
}

struct MipGradParams { float* gradTex[kTexMaxLevels]; int texW, texH, depth, C, levelMax; };

// One lane per 4x4 block of base texels (all channels, four at a time), addressed by the launch grid -- no divisions,
// and the block's ancestors are fetched once: four level-1 texels, one texel of every further level, all loads issued
// before the first sum.  (As one lane per element with 64-bit div/mod chains, every element fetching all of its
// ancestors itself, the kernel took 141 us on a 2048^2 RGB texture; its traffic is 115 MB.)  Every base texel still
// adds its ancestors in level order with the reference's weights (:873-889).
__global__ __launch_bounds__(256) void k_mip_grad(const MipGradParams p)
{
    const int X0 = ((int)blockIdx.x * 64 + (int)(threadIdx.x & 63)) * 4;
    const int Y0 = ((int)blockIdx.y * 4 + (int)(threadIdx.x >> 6)) * 4;
    const int z = (int)blockIdx.z;
    if (X0 >= p.texW || Y0 >= p.texH) return;
    for (int c0 = 0; c0 < p.C; c0 += 4) {
        const int nc = min(4, p.C - c0);
        float v1[2][2][4];                                  // level 1: the block's 2x2 parents
        float vl[kTexMaxLevels][4];                         // levels >= 2: one ancestor each
        {
            const int lv = min(1, p.levelMax);
            const int pw = level_dim(p.texW, lv), ph = level_dim(p.texH, lv);
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int ax = min((X0 >> lv) + i, pw - 1), ay = min((Y0 >> lv) + j, ph - 1);
                    const float* g = p.gradTex[lv] + (((size_t)z * ph + ay) * pw + ax) * p.C + c0;
#pragma unroll
                    for (int k = 0; k < 4; k++) v1[j][i][k] = (k < nc) ? g[k] : 0.f;
                }
        }
#pragma unroll
        for (int level = 2; level < kTexMaxLevels; level++) {
            const int lv = min(level, p.levelMax);          // levels beyond the last re-read the last one and are dropped by a select
            const int pw = level_dim(p.texW, lv), ph = level_dim(p.texH, lv);
            const float* g = p.gradTex[lv] + (((size_t)z * ph + (Y0 >> lv)) * pw + (X0 >> lv)) * p.C + c0;
#pragma unroll
            for (int k = 0; k < 4; k++) vl[level][k] = (k < nc) ? g[k] : 0.f;
        }
#pragma unroll
        for (int dy = 0; dy < 4; dy++) {
            if (Y0 + dy >= p.texH) break;
#pragma unroll
            for (int dx = 0; dx < 4; dx++) {
                if (X0 + dx >= p.texW) break;
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                float w = 1.f;
#pragma unroll
                for (int level = 1; level < kTexMaxLevels; level++) {
                    if (level_dim(p.texW, level - 1) > 1) w *= .5f;
                    if (level_dim(p.texH, level - 1) > 1) w *= .5f;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float g = (level == 1) ? v1[dy >> 1][dx >> 1][k] : vl[level][k];
                        acc[k] = (level <= p.levelMax) ? acc[k] + g * w : acc[k];
                    }
                }
                float* g0 = p.gradTex[0] + (((size_t)z * p.texH + (Y0 + dy)) * p.texW + (X0 + dx)) * p.C + c0;
#pragma unroll
                for (int k = 0; k < 4; k++) if (k < nc) g0[k] += acc[k];
            }
        }
    }
}

// The same for 1..4 channels known at compile time, texture width a multiple of 4 and 16-byte aligned levels: a lane's
// four texels of one row are 4*C contiguous floats = C float4 vectors, read, updated and written as such (scalar
// accesses at a 12-byte stride touched every cache line of a row twelve times).
template <int C>
__global__ __launch_bounds__(256) void k_mip_grad_vec(const MipGradParams p)
{
    const int X0 = ((int)blockIdx.x * 64 + (int)(threadIdx.x & 63)) * 4;
    const int Y0 = ((int)blockIdx.y * 4 + (int)(threadIdx.x >> 6)) * 4;
    const int z = (int)blockIdx.z;
    if (X0 >= p.texW || Y0 >= p.texH) return;
    float v1[2][2][C];
    float vl[kTexMaxLevels][C];
    {
        const int pw = level_dim(p.texW, 1), ph = level_dim(p.texH, 1);
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int ax = min((X0 >> 1) + i, pw - 1), ay = min((Y0 >> 1) + j, ph - 1);
                const float* g = p.gradTex[1] + (((size_t)z * ph + ay) * pw + ax) * C;
#pragma unroll
                for (int k = 0; k < C; k++) v1[j][i][k] = g[k];
            }
    }
#pragma unroll
    for (int level = 2; level < kTexMaxLevels; level++) {
        const int lv = min(level, p.levelMax);
        const int pw = level_dim(p.texW, lv), ph = level_dim(p.texH, lv);
        const float* g = p.gradTex[lv] + (((size_t)z * ph + (Y0 >> lv)) * pw + (X0 >> lv)) * C;
#pragma unroll
        for (int k = 0; k < C; k++) vl[level][k] = g[k];
    }
#pragma unroll
    for (int dy = 0; dy < 4; dy++) {
        if (Y0 + dy >= p.texH) break;
        float4* row = (float4*)(p.gradTex[0] + (((size_t)z * p.texH + (Y0 + dy)) * p.texW + X0) * C);
        float r[4 * C];
#pragma unroll
        for (int q = 0; q < C; q++) { const float4 t = row[q]; r[4 * q] = t.x; r[4 * q + 1] = t.y; r[4 * q + 2] = t.z; r[4 * q + 3] = t.w; }
#pragma unroll
        for (int f = 0; f < 4 * C; f++) {
            const int dx = f / C, k = f % C;
            float acc = 0.f, w = 1.f;
#pragma unroll
            for (int level = 1; level < kTexMaxLevels; level++) {
                if (level_dim(p.texW, level - 1) > 1) w *= .5f;
                if (level_dim(p.texH, level - 1) > 1) w *= .5f;
                const float g = (level == 1) ? v1[dy >> 1][dx >> 1][k] : vl[level][k];
                acc = (level <= p.levelMax) ? acc + g * w : acc;
            }
            r[f] += acc;
        }
#pragma unroll
        for (int q = 0; q < C; q++) row[q] = make_float4(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]);
    }
}

static int mip_info(int tex_n, int tex_h, int tex_w, int C, int cube, int max_mip_level,
                    int* lw, int* lh, int64_t* off, int64_t* total)
{
    int w = tex_w, h = tex_h, level = 0;
    int64_t tot = 0;
    const int c = cube ? C * 6 : C;
    if (lw) lw[0] = w;
    if (lh) lh[0] = h;
    if (off) off[0] = -1;
    if (max_mip_level != 0) {
        while ((w | h) > 1) {                                               // texture.cpp:77-98
            level += 1;
            if ((w > 1 && (w & 1)) || (h > 1 && (h & 1))) return -1;
            if (w > 1) w >>= 1;
            if (h > 1) h >>= 1;
            if (level < kTexMaxLevels) { if (lw) lw[level] = w; if (lh) lh[level] = h; if (off) off[level] = tot; }
            tot += (int64_t)w * h * tex_n * c;
            if (max_mip_level >= 0 && level == max_mip_level) break;
        }
    }
    if (total) *total = tot;
    return level;
}

static int fill_tex_params(TexParams& p, const char* who, const float* tex, const float* const* mip_ptrs_host, int L,
                           const float* uv, const float* uv_da, const float* bias,
                           int tex_n, int tex_h, int tex_w, int C, int N, int H, int W, int filter, int boundary)
{
    NVDR_REQUIRE(filter >= 0 && filter < 4, "filter_mode unsupported");
    NVDR_REQUIRE(boundary >= 0 && boundary < 4, "boundary_mode unsupported");
    const bool cube = (boundary == TEX_B_CUBE);
    NVDR_REQUIRE(tex && uv, "%s: null pointer", who);
    if (cube) {
        NVDR_REQUIRE(tex_n > 0 && tex_h > 0 && tex_w > 0 && C > 0, "tex must have shape[>0, 6, >0, >0, >0] in cube map mode");
        NVDR_REQUIRE(tex_h == tex_w, "texture shape must be square in cube map mode");
        NVDR_REQUIRE(N > 0 && H > 0 && W > 0, "uv must have shape [>0, >0, >0, 3] in cube map mode");
    } else {
        NVDR_REQUIRE(tex_n > 0 && tex_h > 0 && tex_w > 0 && C > 0, "tex must have shape[>0, >0, >0, >0]");
        NVDR_REQUIRE(N > 0 && H > 0 && W > 0, "uv must have shape [>0, >0, >0, 2]");
    }
    NVDR_REQUIRE(tex_n == 1 || tex_n == N, "minibatch size mismatch between inputs tex, uv");
    NVDR_REQUIRE(tex_w <= (1 << 16) && tex_h <= (1 << 16), "texture size too large");
    // 32-bit element indices, like the reference (texture.h:24: "a texture cannot be larger than 2 GB").
    NVDR_REQUIRE((long long)tex_n * (cube ? 6 : 1) * tex_h * tex_w * C < (1ll << 31), "texture size too large (more than 2^31 elements)");
    const bool mips = (filter == TEX_LMN || filter == TEX_LML);
    if (mips) {
        NVDR_REQUIRE(uv_da || bias, "mipmapping filter mode requires uv_da and/or mip_level_bias input");
        NVDR_REQUIRE(L >= 0 && L < kTexMaxLevels, "%s: bad mip level count %d", who, L);
        NVDR_REQUIRE(L == 0 || mip_ptrs_host, "mipmapping filter mode requires mip wrapper or mip stack input");
    }
    if (!cube) {
        NVDR_REQUIRE(!((uintptr_t)uv & 7), "uv input tensor not aligned to float2");
        NVDR_REQUIRE(!((uintptr_t)uv_da & 15), "uv_da input tensor not aligned to float4");
    } else {
        NVDR_REQUIRE(!((uintptr_t)uv_da & 7), "uv_da input tensor not aligned to float2");
    }
    p = TexParams{};
    p.tex[0] = tex;
    p.levelMax = mips ? L : 0;
    for (int i = 1; i <= p.levelMax; i++) {
        NVDR_REQUIRE(mip_ptrs_host[i - 1], "%s: mip level %d missing", who, i);
        p.tex[i] = mip_ptrs_host[i - 1];
    }
    p.uv = uv; p.uvDA = mips ? uv_da : nullptr; p.bias = mips ? bias : nullptr;
    p.boundary = boundary; p.channels = C; p.imgW = W; p.imgH = H; p.n = N;
    p.texW = tex_w; p.texH = tex_h; p.texDepth = tex_n;
    p.tilesX = (W + 15) / 16; p.tilesY = (H + 15) / 16;
    NVDR_REQUIRE((long long)p.tilesX * p.tilesY * N < (1ll << 30), "%s: too many pixel blocks", who);
    p.dbg = debug_flags();
    p.cornerFix = get_option(NVDR_OPT_CUBE_CORNER_FIX);
    return NVDR_OK;
}

static dim3 tex_grid(const TexParams& p, bool ordered = true)
{
    const long long blocks = (ordered && p.zflags.order) ? tile_flags_ordered_grid(p.zflags, 16)   // 16 blocks of 16x16 pixels per bin (tex_pixel)
                                                         : (long long)p.tilesX * p.tilesY * p.n;   // < 2^30: checked in fill_tex_params
    return dim3((unsigned)(((blocks + 7) / 8) * 8));
}

}  // namespace nvdr

using namespace nvdr;

extern "C" int nvdr_texture_mip_info(int tex_n, int tex_h, int tex_w, int C, int cube, int max_mip_level,
                                     int* lvl_w, int* lvl_h, int64_t* lvl_off, int64_t* total_floats)
{
    return mip_info(tex_n, tex_h, tex_w, C, cube, max_mip_level, lvl_w, lvl_h, lvl_off, total_floats);
}

extern "C" int nvdr_texture_construct_mip(const float* tex, int tex_n, int tex_h, int tex_w, int C, int cube,
                                          int max_mip_level, float* mip, nvdrStream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    NVDR_REQUIRE(max_mip_level >= -1, "invalid max_mip_level");
    NVDR_REQUIRE(!cube || tex_h == tex_w, "texture shape must be square in cube map mode");
    NVDR_REQUIRE(tex && tex_n > 0 && tex_h > 0 && tex_w > 0 && C > 0, "tex must have shape[>0, >0, >0, >0]");
    NVDR_REQUIRE(tex_w <= (1 << 16) && tex_h <= (1 << 16), "texture size too large");
    NVDR_REQUIRE((long long)tex_n * (cube ? 6 : 1) <= 65535, "texture_construct_mip: too many texture slices");
    int lw[kTexMaxLevels], lh[kTexMaxLevels]; int64_t off[kTexMaxLevels], total;
    const int depth = cube ? tex_n * 6 : tex_n;                             // six faces per slice
    const int L = mip_info(depth, tex_h, tex_w, C, 0, max_mip_level, lw, lh, off, &total);
    if (L < 0) {                                                             // texture.cpp:15-60 raiseMipSizeError
        set_error("texture_construct_mip: texture extents %d x %d cannot be halved down to 1 (odd size at a mip level); "
                  "use a power-of-two size, or limit max_mip_level", tex_w, tex_h);
        return NVDR_ERR_ARG;
    }
    NVDR_REQUIRE(L == 0 || mip, "texture_construct_mip: null mip buffer");
    // the small levels at the end of the chain form the tail (one launch); the larger ones get a launch each
    const int tail = mip_tail_start(lw, lh, L, (long long)depth * C);
    for (int l = 1; l < tail && l <= L; l++) {
        MipParams mp;
        mp.in = (l == 1) ? tex : mip + off[l - 1];
        mp.out = mip + off[l];
        mp.wi = lw[l - 1]; mp.hi = lh[l - 1]; mp.wo = lw[l]; mp.ho = lh[l]; mp.depth = depth; mp.C = C;
        ProfileScope ps("tex_mip_build", stream);
        hipLaunchKernelGGL(k_mip_build, dim3((unsigned)((mp.wo + 63) / 64), (unsigned)((mp.ho + 3) / 4), (unsigned)depth), dim3(256), 0, stream, mp);
    }
    if (tail <= L) {
        MipTailParams tp;
        tp.in = (tail == 1) ? tex : mip + off[tail - 1];
        tp.mip = mip; tp.first = tail; tp.last = L; tp.depth = depth; tp.C = C;
        for (int l = 0; l <= L; l++) { tp.off[l] = off[l]; tp.w[l] = lw[l]; tp.h[l] = lh[l]; }
        ProfileScope ps("tex_mip_build", stream);
        hipLaunchKernelGGL(k_mip_build_tail, dim3(1), dim3(1024), 0, stream, tp);
    }
    NVDR_LAUNCH_CHECK();
    return NVDR_OK;
}

#define NVDR_TEX_FWD_C(FILTER, BO)                                                                        \
    do {                                                                                                   \
        if (vec4)        hipLaunchKernelGGL((k_tex_fwd<FILTER, BO, 4>), grid, dim3(256), 0, stream, p);    \
        else if (C == 3) hipLaunchKernelGGL((k_tex_fwd<FILTER, BO, 3>), grid, dim3(256), 0, stream, p);    \
        else if (vec2)   hipLaunchKernelGGL((k_tex_fwd<FILTER, BO, 2>), grid, dim3(256), 0, stream, p);    \
        else if (C == 1) hipLaunchKernelGGL((k_tex_fwd<FILTER, BO, 1>), grid, dim3(256), 0, stream, p);    \
        else             hipLaunchKernelGGL((k_tex_fwd<FILTER, BO, 0>), grid, dim3(256), 0, stream, p);    \
    } while (0)

#define NVDR_TEX_FWD_CUBE_C(FILTER, BO)                                                                       \
    do {                                                                                                   \
        if (vec4)        hipLaunchKernelGGL((k_tex_fwd_cube<FILTER, BO, 4>), grid, dim3(256), 0, stream, p); \
        else if (C == 3) hipLaunchKernelGGL((k_tex_fwd_cube<FILTER, BO, 3>), grid, dim3(256), 0, stream, p); \
        else if (vec2)   hipLaunchKernelGGL((k_tex_fwd_cube<FILTER, BO, 2>), grid, dim3(256), 0, stream, p); \
        else if (C == 1) hipLaunchKernelGGL((k_tex_fwd_cube<FILTER, BO, 1>), grid, dim3(256), 0, stream, p); \
        else             hipLaunchKernelGGL((k_tex_fwd_cube<FILTER, BO, 0>), grid, dim3(256), 0, stream, p); \
    } while (0)

extern "C" int nvdr_texture_fwd(const float* tex, const float* const* mip_ptrs_host, int L,
                                const float* uv, const float* uv_da, const float* mip_level_bias,
                                int tex_n, int tex_h, int tex_w, int C, int N, int H, int W,
                                int filter_mode, int boundary_mode, float* out, const uint8_t* tile_flags, nvdrStream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    TexParams p;
    int rc = fill_tex_params(p, "texture_fwd", tex, mip_ptrs_host, L, uv, uv_da, mip_level_bias,
                             tex_n, tex_h, tex_w, C, N, H, W, filter_mode, boundary_mode);
    if (rc) return rc;
    NVDR_REQUIRE(out, "texture_fwd: null output");
    p.out = out;
    if (boundary_mode != TEX_B_CUBE && !(debug_flags() & 33554432)) p.zflags = tile_flags_view(tile_flags, N, H, W, !(debug_flags() & 134217728));
    bool vec4 = (C == 4) && !((uintptr_t)out & 15), vec2 = (C == 2) && !((uintptr_t)out & 7);
    for (int i = 0; i <= p.levelMax; i++) { vec4 = vec4 && !((uintptr_t)p.tex[i] & 15); vec2 = vec2 && !((uintptr_t)p.tex[i] & 7); }
    const dim3 grid = tex_grid(p);
    const bool bo = (p.levelMax >= 0) && (filter_mode >= TEX_LMN) && !p.uvDA;
    if (boundary_mode == TEX_B_CUBE) {
        ProfileScope ps("tex_fwd_cube", stream);
        switch (filter_mode) {
        case TEX_NEAREST: NVDR_TEX_FWD_CUBE_C(TEX_NEAREST, false); break;
        case TEX_LINEAR:  NVDR_TEX_FWD_CUBE_C(TEX_LINEAR, false); break;
        case TEX_LMN:     if (bo) NVDR_TEX_FWD_CUBE_C(TEX_LMN, true); else NVDR_TEX_FWD_CUBE_C(TEX_LMN, false); break;
        default:          if (bo) NVDR_TEX_FWD_CUBE_C(TEX_LML, true); else NVDR_TEX_FWD_CUBE_C(TEX_LML, false); break;
        }
    } else {
        ProfileScope ps("tex_fwd", stream);
        switch (filter_mode) {
        case TEX_NEAREST: NVDR_TEX_FWD_C(TEX_NEAREST, false); break;
        case TEX_LINEAR:  NVDR_TEX_FWD_C(TEX_LINEAR, false); break;
        case TEX_LMN:     if (bo) NVDR_TEX_FWD_C(TEX_LMN, true); else NVDR_TEX_FWD_C(TEX_LMN, false); break;
        default:          if (bo) NVDR_TEX_FWD_C(TEX_LML, true); else NVDR_TEX_FWD_C(TEX_LML, false); break;
        }
    }
    NVDR_LAUNCH_CHECK();
    return NVDR_OK;
}

// Records of the two-level reduction: one per wave of pixels = four per 16x16-pixel block, 8 + C words each.
static long long tex_grad_records(int N, int H, int W) { return 4ll * ((W + 15) / 16) * ((H + 15) / 16) * N; }

extern "C" size_t nvdr_texture_grad_scratch_bytes(int N, int H, int W, int C)
{
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    const size_t rec = (size_t)tex_grad_records(N, H, W) * (size_t)(kTexRecHeader + C) * 4;
    return align_up(rec, 256) + align_up((size_t)tex_grad_records(N, H, W) / 4, 256);        // records + one byte per block
}

extern "C" int nvdr_texture_grad(const float* tex, const float* const* mip_ptrs_host, int L,
                                 const float* uv, const float* uv_da, const float* mip_level_bias, const float* dy,
                                 int tex_n, int tex_h, int tex_w, int C, int N, int H, int W,
                                 int filter_mode, int boundary_mode, int pull_mip_grads,
                                 float* g_tex, float* const* g_mip_ptrs_host,
                                 float* g_uv, float* g_uv_da, float* g_mip_level_bias,
                                 void* scratch, size_t scratch_bytes, const uint8_t* tile_flags, nvdrStream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    TexParams p;
    int rc = fill_tex_params(p, "texture_grad", tex, mip_ptrs_host, L, uv, uv_da, mip_level_bias,
                             tex_n, tex_h, tex_w, C, N, H, W, filter_mode, boundary_mode);
    if (rc) return rc;
    NVDR_REQUIRE(dy && g_tex, "texture_grad: null pointer");
    NVDR_REQUIRE(filter_mode == TEX_NEAREST || g_uv, "texture_grad: g_uv missing");
    const bool cube = (boundary_mode == TEX_B_CUBE);
    if (!cube) {
        NVDR_REQUIRE(!((uintptr_t)g_uv & 7), "grad_uv output tensor not aligned to float2");
        NVDR_REQUIRE(!((uintptr_t)g_uv_da & 15), "grad_uv_da output tensor not aligned to float4");
    } else {
        NVDR_REQUIRE(!((uintptr_t)g_uv_da & 7), "grad_uv_da output tensor not aligned to float2");
    }
    p.dy = dy;
    if (!cube && !(debug_flags() & 33554432)) p.zflags = tile_flags_view(tile_flags, N, H, W, !(debug_flags() & 134217728));
    p.gradTex[0] = g_tex;
    for (int i = 1; i <= p.levelMax; i++) {
        NVDR_REQUIRE(g_mip_ptrs_host && g_mip_ptrs_host[i - 1], "texture_grad: gradient buffer of mip level %d missing", i);
        p.gradTex[i] = g_mip_ptrs_host[i - 1];
    }
    p.gradUV = g_uv;
    p.gradUVDA = (filter_mode == TEX_LML && p.uvDA) ? g_uv_da : nullptr;
    p.gradBias = (filter_mode == TEX_LML && p.bias) ? g_mip_level_bias : nullptr;
    NVDR_REQUIRE(!(filter_mode == TEX_LML && p.uvDA) || g_uv_da, "texture_grad: g_uv_da missing");
    NVDR_REQUIRE(!(filter_mode == TEX_LML && p.bias) || g_mip_level_bias, "texture_grad: g_mip_level_bias missing");
    const dim3 grid = tex_grid(p);
    const bool bo = (filter_mode >= TEX_LMN) && !p.uvDA;
    // LDS patch table: as many power-of-two patches of 16 texels as fit in 26 KiB (at most 512), so that
    // six workgroups share a CU (the kernel is latency bound: occupancy matters more than table size);
    // none (direct atomics) when even 16 patches do not fit.
    int groups = 512;
    while (groups >= 16 && (size_t)groups * (8 + 64 * (size_t)C) + 16 > 26 * 1024) groups >>= 1;
    if (groups < 16 || tex_w > 32768 || (long long)tex_h * (cube ? 6 : 1) > 65536 || (debug_flags() & 256)) groups = 0;   // key format
    const size_t lds = (size_t)groups * (8 + 64 * (size_t)C) + 16;         // 16 texels x C sums + key 4 B + used-list entry 4 B per patch
    // Second reduction level for constant-uv regions (TexParams::rec), when the caller brought scratch and the kernel has
    // a uniform-wave path for this mode (2-D, bilinear footprint, level from uv_da) and a table to fall back on.
    const long long nrec = tex_grad_records(N, H, W);
    const bool records = scratch && groups > 0 && !cube && filter_mode != TEX_NEAREST && !bo && !(debug_flags() & 16384)
                         && scratch_bytes >= nvdr_texture_grad_scratch_bytes(N, H, W, C) && nrec < (1ll << 31) / (kTexRecHeader + C);
    if (records) {
        NVDR_REQUIRE(!((uintptr_t)scratch & 3), "texture_grad: scratch must be 4-byte aligned");
        p.rec = (int*)scratch; p.nrec = (int)nrec;
        NVDR_HIP_CHECK(hipMemsetAsync(scratch, 0xFF, (size_t)nrec * 4, stream));           // "no record" in every first-tap slot
        // two-kernel pass: the blocks without real footprints first, at full occupancy (k_tex_grad_light)
        if (!(debug_flags() & 268435456)) {
            p.heavy = (uint8_t*)scratch + align_up((size_t)nrec * (size_t)(kTexRecHeader + C) * 4, 256);
            ProfileScope ps("tex_grad_light", stream);
            const dim3 gridL = tex_grid(p, false);
#define NVDR_TEX_LIGHT(FILTER)                                                                                      \
    do {                                                                                                            \
        if (C == 1)      hipLaunchKernelGGL((k_tex_grad_light<FILTER, 1>), gridL, dim3(256), 0, stream, p);          \
        else if (C == 2) hipLaunchKernelGGL((k_tex_grad_light<FILTER, 2>), gridL, dim3(256), 0, stream, p);          \
        else if (C == 3) hipLaunchKernelGGL((k_tex_grad_light<FILTER, 3>), gridL, dim3(256), 0, stream, p);          \
        else if (C == 4) hipLaunchKernelGGL((k_tex_grad_light<FILTER, 4>), gridL, dim3(256), 0, stream, p);          \
        else             hipLaunchKernelGGL((k_tex_grad_light<FILTER, 0>), gridL, dim3(256), 0, stream, p);          \
    } while (0)
            const int gx4 = (p.tilesX + 3) / 4;
            const int lexp = tune_int("NVDR_TUNE_TEX_LIGHT_EXP", 0);
            const dim3 gridW((unsigned)((((long long)gx4 * p.tilesY * p.n + 7) / 8) * 8));
#define NVDR_TEX_LIGHT_W(FILTER)                                                                                    \
    do {                                                                                                            \
        if (C == 1)      hipLaunchKernelGGL((k_tex_grad_light_w<FILTER, 1>), gridW, dim3(256), 0, stream, p, gx4, lexp);   \
        else if (C == 2) hipLaunchKernelGGL((k_tex_grad_light_w<FILTER, 2>), gridW, dim3(256), 0, stream, p, gx4, lexp);   \
        else if (C == 3) hipLaunchKernelGGL((k_tex_grad_light_w<FILTER, 3>), gridW, dim3(256), 0, stream, p, gx4, lexp);   \
        else             hipLaunchKernelGGL((k_tex_grad_light_w<FILTER, 4>), gridW, dim3(256), 0, stream, p, gx4, lexp);   \
    } while (0)
            if (C >= 1 && C <= 4 && (long long)N * H * W * 16 < (1ll << 32) && tune_int("NVDR_TUNE_TEX_LIGHT_W", 1)) {     // one wave per block (k_tex_grad_light_w; 32-bit byte offsets)
                if (filter_mode == TEX_LINEAR) NVDR_TEX_LIGHT_W(TEX_LINEAR);
                else if (filter_mode == TEX_LMN) NVDR_TEX_LIGHT_W(TEX_LMN);
                else NVDR_TEX_LIGHT_W(TEX_LML);
            }
            else if (filter_mode == TEX_LINEAR) NVDR_TEX_LIGHT(TEX_LINEAR);
            else if (filter_mode == TEX_LMN) NVDR_TEX_LIGHT(TEX_LMN);
            else NVDR_TEX_LIGHT(TEX_LML);
            NVDR_LAUNCH_CHECK();
        }
    }
    // the heavy blocks through the lean kernel where it applies (2-D, level from uv_da, bilinear footprints, 1..4 channels,
    // two-kernel pass, a table), else through the general one
    const bool lean = p.heavy && groups > 0 && !cube && !bo && (filter_mode == TEX_LINEAR || filter_mode == TEX_LML) && C >= 1 && C <= 4
                      && tune_int("NVDR_TUNE_TEX_LEAN", 1);
    if (lean) {
        ProfileScope ps("tex_grad", stream);
#define NVDR_TEX_LEAN(FILTER)                                                                                        \
    do {                                                                                                             \
        if (C == 1)      hipLaunchKernelGGL((k_tex_grad_lean<FILTER, 1>), grid, dim3(256), lds, stream, p, groups);  \
        else if (C == 2) hipLaunchKernelGGL((k_tex_grad_lean<FILTER, 2>), grid, dim3(256), lds, stream, p, groups);  \
        else if (C == 3) hipLaunchKernelGGL((k_tex_grad_lean<FILTER, 3>), grid, dim3(256), lds, stream, p, groups);  \
        else             hipLaunchKernelGGL((k_tex_grad_lean<FILTER, 4>), grid, dim3(256), lds, stream, p, groups);  \
    } while (0)
        if (filter_mode == TEX_LINEAR) NVDR_TEX_LEAN(TEX_LINEAR); else NVDR_TEX_LEAN(TEX_LML);
    } else {
        ProfileScope ps("tex_grad", stream);
#define NVDR_TEX_GRAD_C(FILTER, BO, CUBE, CC) hipLaunchKernelGGL((k_tex_grad<FILTER, BO, CUBE, CC>), grid, dim3(256), lds, stream, p, groups)
#define NVDR_TEX_GRAD(FILTER, BO)                                                                              \
    do {                                                                                                       \
        if (cube) NVDR_TEX_GRAD_C(FILTER, BO, true, 0);                                                        \
        else if (!BO && (FILTER == TEX_LINEAR || FILTER == TEX_LML) && C >= 1 && C <= 4) {                     \
            if (C == 1) NVDR_TEX_GRAD_C(FILTER, false, false, 1); else if (C == 2) NVDR_TEX_GRAD_C(FILTER, false, false, 2); \
            else if (C == 3) NVDR_TEX_GRAD_C(FILTER, false, false, 3); else NVDR_TEX_GRAD_C(FILTER, false, false, 4);        \
        } else NVDR_TEX_GRAD_C(FILTER, BO, false, 0);                                                          \
    } while (0)
        switch (filter_mode) {
        case TEX_NEAREST: NVDR_TEX_GRAD(TEX_NEAREST, false); break;
        case TEX_LINEAR:  NVDR_TEX_GRAD(TEX_LINEAR, false); break;
        case TEX_LMN:     if (bo) NVDR_TEX_GRAD(TEX_LMN, true); else NVDR_TEX_GRAD(TEX_LMN, false); break;
        default:          if (bo) NVDR_TEX_GRAD(TEX_LML, true); else NVDR_TEX_GRAD(TEX_LML, false); break;
        }
    }
    NVDR_LAUNCH_CHECK();
    if (records) {
        ProfileScope ps("tex_grad_fold", stream);
        const dim3 fgrid((unsigned)((nrec + 4ll * kFoldPerWave - 1) / (4ll * kFoldPerWave)));
        if (C == 1)      hipLaunchKernelGGL(k_tex_grad_fold<1>, fgrid, dim3(256), 0, stream, p.rec, p.nrec, C, g_tex);
        else if (C == 2) hipLaunchKernelGGL(k_tex_grad_fold<2>, fgrid, dim3(256), 0, stream, p.rec, p.nrec, C, g_tex);
        else if (C == 3) hipLaunchKernelGGL(k_tex_grad_fold<3>, fgrid, dim3(256), 0, stream, p.rec, p.nrec, C, g_tex);
        else if (C == 4) hipLaunchKernelGGL(k_tex_grad_fold<4>, fgrid, dim3(256), 0, stream, p.rec, p.nrec, C, g_tex);
        else             hipLaunchKernelGGL(k_tex_grad_fold<0>, fgrid, dim3(256), 0, stream, p.rec, p.nrec, C, g_tex);
        NVDR_LAUNCH_CHECK();
    }
    if (pull_mip_grads && p.levelMax > 0) {                                  // torch_texture.cpp:679-687
        MipGradParams mg;
        for (int i = 0; i <= p.levelMax; i++) mg.gradTex[i] = p.gradTex[i];
        mg.texW = tex_w; mg.texH = tex_h; mg.depth = cube ? tex_n * 6 : tex_n; mg.C = C; mg.levelMax = p.levelMax;
        NVDR_REQUIRE(mg.depth <= 65535, "texture_grad: too many texture slices for the mip gradient pass");
        ProfileScope ps("tex_mip_grad", stream);
        const dim3 mgrid((unsigned)((tex_w + 255) / 256), (unsigned)((tex_h + 15) / 16), (unsigned)mg.depth);
        const bool vec = C <= 4 && (tex_w & 3) == 0 && !((uintptr_t)mg.gradTex[0] & 15);     // only the base level is accessed as vectors
        if (vec && C == 1)      hipLaunchKernelGGL(k_mip_grad_vec<1>, mgrid, dim3(256), 0, stream, mg);
        else if (vec && C == 2) hipLaunchKernelGGL(k_mip_grad_vec<2>, mgrid, dim3(256), 0, stream, mg);
        else if (vec && C == 3) hipLaunchKernelGGL(k_mip_grad_vec<3>, mgrid, dim3(256), 0, stream, mg);
        else if (vec && C == 4) hipLaunchKernelGGL(k_mip_grad_vec<4>, mgrid, dim3(256), 0, stream, mg);
        else                    hipLaunchKernelGGL(k_mip_grad, mgrid, dim3(256), 0, stream, mg);
        NVDR_LAUNCH_CHECK();
    }
    return NVDR_OK;
}
