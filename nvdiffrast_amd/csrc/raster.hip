// raster.hip -- rasterize forward / backward for gfx950 (MI355X).
//
// Replaces, behind the C ABI of include/nvdr_hip.h, the reference's CudaRaster pipeline
// (csrc/common/cudaraster/impl/{TriangleSetup,BinRaster,CoarseRaster,FineRaster}.inl),
// its pixel shader and its gradient kernels (csrc/common/rasterize.cu).  The integer
// rules (snap, cull, fill rule, U32 depth plane, visibility) are the reference's; the
// machinery is not:
//
//  k_setup   one lane per triangle: snap / cull / clip / depth-plane setup.  Emits a 64 B
//            record per surviving (sub)triangle: three edge functions in "pixel form"
//            E(X,Y) = C + X*A + Y*B (fill rule folded into C), the depth plane, the id
//            and a packed tile AABB; records and AABBs are staged in LDS and leave as whole
//            rows.  Triangles crossing a frustum plane are queued and clipped densely in a second
//            pass (once each, polygon buffers in LDS); their sub-triangles take slots from a
//            per-image pool -- sized for the worst case (7 per triangle) while that is affordable,
//            so nothing can overflow and the host never synchronises (the reference retries after
//            a D2H copy, RasterImpl.cpp:174-231,367), else grown on demand by the caller.
//            Per-bin triangle counts and slot ranges are accumulated in an LDS histogram.
//  k_order   heavy-first work order of the (image, bin) items inside each XCD's chunk; last
//            reader of k_setup's counters, which it leaves zeroed for the next call.  In small
//            launches it also gives the bins with the most triangles extra work items (parts).
//  k_fine    one workgroup (8 waves) per 64x64-pixel bin.  (1) The waves scan the bin's range
//            of packed AABBs and compact the triangles that touch the bin into an LDS list
//            (ballot/mbcnt prefix sums).  (2) The list's (triangle, 8x8 tile) pairs are numbered
//            by a prefix sum and dealt to the waves 64 at a time, ONE LANE PER PAIR: the lane
//            walks its triangle's three edge functions over the tile's 64 pixel centres with
//            integer adds (two pixels per instruction, in 16-bit halves) into a 64-bit coverage mask, then pops the set bits (masks with many
//            fragments: eight masks at a time, eight lanes per mask, when the batch holds several; else the whole wave,
//            one lane per pixel, the mask as execution mask) and merges
//            depth << 32 | ~id into the tile's per-pixel keys with an LDS 64-bit atomic min:
//            minimum depth wins, ties go to the highest triangle id, which is exactly what
//            the reference's in-order LEQUAL ROP produces (FineRaster.inl:152-172,349-361)
//            without needing any ordering.  (3) One lane per pixel column, one image row of the bin per store: the winning id
//            goes straight into the pixel shader (rasterize.cu:15-114) in the same kernel, so the
//            id/depth surfaces never touch HBM (the depth surface is stored only for depth
//            peeling).  A bin shared by several workgroups: every part rasterises its share of the
//            bin's slot range, the parts' key arrays meet in memory (returning device-scope atomics
//            only) and the part that arrives last shades.
//  k_raster_grad  rasterize.cu:119-277; per-pixel gradients (a reverse-mode tape of the pixel
//            shader) are summed over triangle runs, accumulated per vertex in an LDS fixed-point
//            hash table per 64x16-pixel block and flushed with one hardware atomic per
//            (vertex, component).
#include "nvdr_device.hpp"
#include "nvdr_host.hpp"
#include "nvdr_raster_tape.hpp"

namespace nvdr {

constexpr int      kSpLog2    = 4;                       // Constants.hpp:14
constexpr uint32_t kDepthMin  = 17600u;                  // Constants.hpp:70
constexpr uint32_t kDepthMax  = 0xFFFFFFFFu - 17600u;    // Constants.hpp:71
constexpr int      kMaxViewport = 2048;                  // Constants.hpp:13
constexpr uint32_t kEmptyBox  = 0x000000FFu;             // txlo=255 > txhi=0
constexpr int      kSubPerTri = 7;                       // clipper output: <= 9 verts -> <= 7 tris

constexpr int kBinTiles   = 8;                           // bin = 8x8 tiles = 64x64 px
// (at most (2048 / 64)^2 = 1024 bins per viewport tile: scratch_layout sizes the per-bin arrays from the actual count)
constexpr int kFineWaves  = 8;                           // one wave per row of eight 8x8 tiles
constexpr int kFineThreads = kFineWaves * 64;
constexpr int kSplitTris  = 768;                         // bins with at least this many triangles are shared by several workgroups
constexpr int kSplitPart  = 384;                         // ... of about this many triangles each
constexpr int kSplitMaxParts = 4;
constexpr int kSplitsPerChunk = 32;                      // at most this many shared bins per XCD chunk of the work order
constexpr int kHelpersPerChunk = kSplitsPerChunk * (kSplitMaxParts - 1);
// Launches over large meshes (bins with triangle lists, below): a bin may hold tens of thousands of triangles, so it is shared
// by many more workgroups, of about kListSplitPart triangles each, as long as the chunk's helper slots last.
constexpr int kListSplitTris = 2048, kListSplitPart = 1024, kListSplitMaxParts = 127, kListHelpersPerChunk = 480;
// split descriptor: parts | part << 7 | split number << 14
constexpr int kPartBits = 7, kPartMask = (1 << kPartBits) - 1;
constexpr int kListCap    = 448;                         // LDS triangle list capacity (< 512: 9-bit entry numbers in a pair's code)
constexpr int kListMinTris = 32768;                      // meshes of at least this many triangles get per-bin triangle lists (k_binscan / k_binfill)
constexpr int kListScanFactor = 8, kListScanBias = 2048; // a bin takes its list instead of scanning its slot range when range > factor * triangles + bias
constexpr int kFillThreads = 1024, kFillSlots = kFillThreads * 4;   // k_binfill: slots per workgroup
constexpr int kWavesPerRow = kFineWaves / kBinTiles;        // waves sharing one row of eight 8x8 tiles
constexpr int kTilesPerWave = kBinTiles / kWavesPerRow;

struct Viewport {
    int   vpw, vph;            // viewport size in pixels (unpadded)
    int   offx, offy;          // viewport offset inside the image (multiples of 8)
    float xs, ys, xo, yo;      // clip-space transform into the viewport tile (RasterImpl.cpp:295-298)
};

struct SetupParams {
    const float* pos; const int* tri; const int* ranges;
    int instance, N, V, T, maxTri, poolBase, slots;
    Viewport vp;
    uint4* rec; uint32_t* bbox; int* poolCount;
    int* binCount; int binsX, binsY;       // per (image, 64x64 bin) triangle counts, filled here
    int* binHi; int* binLoInv;             // per bin: (max direct slot) + 1 and INT_MAX - (min direct slot); 0 = none
};

// ---------------------------------------------------------------------------------
// Triangle setup
// ---------------------------------------------------------------------------------

__device__ __forceinline__ int cvt_rni_sat_s32(float a) {
    // cvt.rni.sat.s32.f32 (Util.inl:31): nearest-even, clamp, NaN -> 0.
    if (a != a) return 0;
    a = fminf(fmaxf(a, -2147483648.0f), 2147483520.0f);
    return (int)rintf(a);
}

__device__ __forceinline__ uint32_t cvt_rzi_u32(float a) {
    if (!(a > 0.0f)) return 0u;
    if (a >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)a;
}

__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

struct SubTri {
    int px[3], py[3];
    uint32_t zx, zy, zb;
    uint32_t zmin;             // lower bound of the triangle's depths (TriangleSetup.inl:158,176), for k_fine's per-tile depth cull
};

// U32 fixed-point depth plane depth(X, Y) = zb + zx * X + zy * Y (wrapping arithmetic), bit-compatible
// with the reference's plane equation (Util.inl:184-210) because the visibility test compares these
// integers.  The scheme, in its own words:
//   * the three vertex depths (floats in [kDepthMin, kDepthMax]) are truncated to integers after dropping
//     `drop` low bits, drop = clamp(exponent(max depth) - 22, 0, 8), so that depth differences fit 24+ bits;
//   * 1/area arrives as a float and is taken apart into a 24-bit mantissa and a right shift, which turns the
//     division by the area into a 64-bit multiply and an arithmetic shift;
//   * the plane gradients are (dz x edge) * mantissa >> shift, rescaled from sub-pixel to pixel units;
//   * the constant is evaluated around an integer pixel near the middle of the triangle (cx, cy), so that
//     the products that get truncated stay small, and is then moved to the origin with exact U32 wraps.
__device__ void setup_depth_plane(const float zv[3], int v0x, int v0y, int d1x, int d1y, int d2x, int d2y,
                                  float area_rcp, uint32_t& zx, uint32_t& zy, uint32_t& zb)
{
    const float zmax = fmaxf(fmaxf(zv[0], zv[1]), zv[2]);
    const int drop = min(max(((__float_as_int(zmax) >> 23) - 127) - 22, 0), 8);
    const int z0  = (int)(cvt_rzi_u32(zv[0]) >> drop);
    const int dz1 = (int)(cvt_rzi_u32(zv[1]) >> drop) - z0;
    const int dz2 = (int)(cvt_rzi_u32(zv[2]) >> drop) - z0;

    const int rbits = __float_as_int(area_rcp);
    const long long mant = (long long)(((uint32_t)rbits & 0x007FFFFFu) | 0x00800000u);
    const int shift = (127 + 23) - (rbits >> 23);            // area_rcp = mant * 2^-shift

    const long long gx = ((long long)dz1 * d2y - (long long)dz2 * d1y) * mant;      // dz/dx * area * mant
    const long long gy = ((long long)dz2 * d1x - (long long)dz1 * d2x) * mant;
    const int toPixels = shift - (drop + kSpLog2);
    zx = (uint32_t)(gx >> toPixels);
    zy = (uint32_t)(gy >> toPixels);

    // reference pixel: centre of the triangle's bounding box, in pixels
    const int cx = (2 * v0x + min3i(d1x, d2x, 0) + max3i(d1x, d2x, 0)) >> (kSpLog2 + 1);
    const int cy = (2 * v0y + min3i(d1y, d2y, 0) + max3i(d1y, d2y, 0)) >> (kSpLog2 + 1);
    const int offx = v0x - (cx << kSpLog2), offy = v0y - (cy << kSpLog2);           // vertex 0 relative to it, sub-pixels

    uint32_t c = (uint32_t)z0 << drop;                                              // depth at vertex 0
    c -= (uint32_t)((((gx >> 13) * offx) + ((gy >> 13) * offy)) >> (shift - (drop + 13)));   // -> depth at (cx, cy)
    c -= zx * (uint32_t)cx + zy * (uint32_t)cy;                                     // -> depth at the origin
    zb = c;
}

// TriangleSetup.inl:11-24, 42-116, 120-177.  Explicit fmaf = the sites nvcc contracts.
__device__ bool snap_cull_setup(const Viewport& vp, const float (*v)[4], SubTri& st)
{
#pragma clang fp contract(off)
    float vsx = (float)(vp.vpw << (kSpLog2 - 1));
    float vsy = (float)(vp.vph << (kSpLog2 - 1));
    float rw[3]; int px[3], py[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        rw[k] = 1.0f / v[k][3];
        float sx = v[k][0] * rw[k]; sx = sx * vsx;
        float sy = v[k][1] * rw[k]; sy = sy * vsy;
        px[k] = cvt_rni_sat_s32(sx);
        py[k] = cvt_rni_sat_s32(sy);
    }
    int lox = min3i(px[0], px[1], px[2]), loy = min3i(py[0], py[1], py[2]);
    int hix = max3i(px[0], px[1], px[2]), hiy = max3i(py[0], py[1], py[2]);

    int d1x = px[1] - px[0], d1y = py[1] - py[0];
    int d2x = px[2] - px[0], d2y = py[2] - py[0];
    int area = (int)((uint32_t)d1x * (uint32_t)d2y - (uint32_t)d1y * (uint32_t)d2x);
    if (area == 0) return false;

    const int ss = 1 << kSpLog2;
    int bx = (vp.vpw << (kSpLog2 - 1)) - (ss >> 1);
    int by = (vp.vph << (kSpLog2 - 1)) - (ss >> 1);
    int alox = (lox + ss - 1 + bx) & -ss, aloy = (loy + ss - 1 + by) & -ss;
    int ahix = (hix + bx) & -ss,          ahiy = (hiy + by) & -ss;
    if (alox > ahix || aloy > ahiy) return false;

    int diff = ahix + ahiy - alox - aloy;
    if (diff <= ss) {
        bool ok = false;
        for (int pass = 0; pass < 2 && !ok; pass++) {
            if (pass && diff == 0) break;
            int sx = pass ? ahix : alox, sy = pass ? ahiy : aloy;
            int t0x = px[0] + bx - sx, t0y = py[0] + by - sy;
            int t1x = px[1] + bx - sx, t1y = py[1] + by - sy;
            int t2x = px[2] + bx - sx, t2y = py[2] + by - sy;
            int e0 = t0x * t1y - t0y * t1x;
            int e1 = t1x * t2y - t1y * t2x;
            int e2 = t2x * t0y - t2y * t0x;
            if (area < 0) { e0 = -e0; e1 = -e1; e2 = -e2; }
            ok = !(e0 < 0 || e1 < 0 || e2 < 0);
        }
        if (!ok) return false;
    }

    float vz0 = v[0][2], vz1 = v[1][2], vz2 = v[2][2];
    float rw0 = rw[0], rw1 = rw[1], rw2 = rw[2];
    if (area < 0) {
        int t;
        t = d1x; d1x = d2x; d2x = t;   t = d1y; d1y = d2y; d2y = t;
        t = px[1]; px[1] = px[2]; px[2] = t;   t = py[1]; py[1] = py[2]; py[2] = t;
        float f = vz1; vz1 = vz2; vz2 = f;
        f = rw1; rw1 = rw2; rw2 = f;
        area = -area;
    }

    const float zcoef = (float)(kDepthMax - kDepthMin) * 0.5f;
    const float zbias = (float)(kDepthMax + kDepthMin) * 0.5f;
    float zv[3];
    zv[0] = __fmaf_rn(vz0 * zcoef, rw0, zbias);
    zv[1] = __fmaf_rn(vz1 * zcoef, rw1, zbias);
    zv[2] = __fmaf_rn(vz2 * zcoef, rw2, zbias);

    int wv0x = px[0] + (vp.vpw << (kSpLog2 - 1));
    int wv0y = py[0] + (vp.vph << (kSpLog2 - 1));
    setup_depth_plane(zv, wv0x - (1 << (kSpLog2 - 1)), wv0y - (1 << (kSpLog2 - 1)),
                      d1x, d1y, d2x, d2y, 1.0f / (float)area, st.zx, st.zy, st.zb);
    {   // zmin = cvt.rni.sat.u32(min vertex depth - CR_LERP_ERROR(0)) & 0xfffff000 (TriangleSetup.inl:158,176; Constants.hpp:69)
        const float zlo = fminf(fminf(zv[0], zv[1]), zv[2]) - 2200.0f;
        st.zmin = (!(zlo > 0.0f) ? 0u : zlo >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)rintf(zlo)) & 0xFFFFF000u;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { st.px[k] = px[k]; st.py[k] = py[k]; }
    return true;
}

// Record layout (4 x uint4):
//   q0 = {A0, B0, C0, A1}   q1 = {B1, C1, A2, B2}   q2 = {C2, zx, zy, zb}   q3 = {id, aabb, zmin, 0}
// with E_e(X,Y) = C_e + X*A_e + Y*B_e >= 0  <=>  pixel (X,Y) is inside edge e.
__device__ void emit_record(const SetupParams& p, int n, int slot, const SubTri& s, int id, int* s_hist, uint4* stage, uint32_t* boxStage)
{
    const Viewport& vp = p.vp;
    int bx = (vp.vpw - 1) << (kSpLog2 - 1);
    int by = (vp.vph - 1) << (kSpLog2 - 1);
    int lox = min3i(s.px[0], s.px[1], s.px[2]), hix = max3i(s.px[0], s.px[1], s.px[2]);
    int loy = min3i(s.py[0], s.py[1], s.py[2]), hiy = max3i(s.py[0], s.py[1], s.py[2]);
    int x0 = max((lox + bx + 15) >> 4, 0), x1 = min((hix + bx) >> 4, vp.vpw - 1);
    int y0 = max((loy + by + 15) >> 4, 0), y1 = min((hiy + by) >> 4, vp.vph - 1);
    size_t so = (size_t)n * p.slots + slot;
    if (x0 > x1 || y0 > y1) { if (boxStage) *boxStage = kEmptyBox; else p.bbox[so] = kEmptyBox; return; }
    uint32_t box = (uint32_t)(x0 >> 3) | ((uint32_t)(y0 >> 3) << 8) | ((uint32_t)(x1 >> 3) << 16) | ((uint32_t)(y1 >> 3) << 24);
    // Per-bin histogram: here only for the clipper's output (stage == nullptr, rare); the common path's triangles are counted
    // by their wave together, after the pass (hist_add_wave in k_setup).
    if (!stage)
    for (int by_ = y0 >> 6; by_ <= (y1 >> 6); by_++)
        for (int bx_ = x0 >> 6; bx_ <= (x1 >> 6); bx_++)
        {
            const int nb = p.binsX * p.binsY;                         // s_hist = [3][nb]
            const int b = by_ * p.binsX + bx_;
            atomicAdd(&s_hist[b], 1);
            if (slot < p.poolBase) {                                  // index range of the bin's direct slots
                atomicMax(&s_hist[nb + b], slot + 1);
                atomicMax(&s_hist[2 * nb + b], 0x7FFFFFFF - slot);
            }
        }

    uint32_t A[3], B[3], C[3];
#pragma unroll
    for (int e = 0; e < 3; e++) {
        int a = e, b = (e + 1) % 3;
        int dx = s.px[b] - s.px[a], dy = s.py[b] - s.py[a];
        // E(s) = (a - s) x d with s = (16X - bx, 16Y - by); exclusive edge needs E > 0 (Util.inl:304-309).
        uint32_t excl = (dy > 0 || (dy == 0 && dx <= 0)) ? 1u : 0u;
        A[e] = (uint32_t)(-16 * dy);
        B[e] = (uint32_t)(16 * dx);
        C[e] = (uint32_t)(s.px[a] + bx) * (uint32_t)dy - (uint32_t)(s.py[a] + by) * (uint32_t)dx - excl;
    }
    // Direct slots go through the block's LDS stage and leave as whole 1 KiB rows (k_setup); pool slots
    // (clipper output, rare) are written in place.
    uint4* r = stage ? stage : p.rec + so * 4;
    r[0] = make_uint4(A[0], B[0], C[0], A[1]);
    r[1] = make_uint4(B[1], C[1], A[2], B[2]);
    r[2] = make_uint4(C[2], s.zx, s.zy, s.zb);
    r[3] = make_uint4((uint32_t)id, box, s.zmin, 0u);
    if (boxStage) *boxStage = box; else p.bbox[so] = box;         // direct slots: AABBs leave as whole rows too (k_setup)
}

// Workspace of the clipper: the polygon's vertices as barycentric pairs, two buffers of 9 x 2 floats per lane
// (Sutherland-Hodgman ping-pong), in LDS and word-major -- word w of a lane at ws[w * kClipLanes] -- so that
// lanes hit distinct banks.  (As private arrays these were dynamically indexed scratch: ~100 dependent memory
// round trips per clipped triangle.)  112 lanes x 36 words = 16128 B: fits the 16 KiB record stage it reuses.
constexpr int kClipLanes = 112;
#define NVDR_CW(off, i) ws[((off) + (i)) * kClipLanes]

// Util.inl:101-130: clips the polygon at word offset inOff against f0 + f1*b0 + f2*b1 >= 0 into outOff.
__device__ __forceinline__ int clip_poly_plane(float* ws, int outOff, int inOff, int n_in, float f0, float f1, float f2)
{
#pragma clang fp contract(off)
    int n_out = 0;
    if (n_in >= 3) {
        const int ai = (n_in - 1) * 2;
        float ax = NVDR_CW(inOff, ai), ay = NVDR_CW(inOff, ai + 1);
        float av = __fmaf_rn(f2, ay, __fmaf_rn(f1, ax, f0));
#pragma unroll 1
        for (int bi = 0; bi < n_in * 2; bi += 2) {
            const float bx = NVDR_CW(inOff, bi), by = NVDR_CW(inOff, bi + 1);
            const float bv = __fmaf_rn(f2, by, __fmaf_rn(f1, bx, f0));
            if (av * bv < 0.0f) {
                const float bc = av / (av - bv);
                const float ac = 1.0f - bc;
                NVDR_CW(outOff, n_out + 0) = __fmaf_rn(ax, ac, bx * bc);
                NVDR_CW(outOff, n_out + 1) = __fmaf_rn(ay, ac, by * bc);
                n_out += 2;
            }
            if (bv >= 0.0f) {
                NVDR_CW(outOff, n_out + 0) = bx;
                NVDR_CW(outOff, n_out + 1) = by;
                n_out += 2;
            }
            ax = bx; ay = by; av = bv;
        }
    }
    return n_out >> 1;
}

// Slow path, TriangleSetup.inl:355-434 + Util.inl:134-160: clip against the frustum in barycentric space, fan the
// polygon, set up every sub-triangle.  Returns the number of survivors (at most 7), left in st[].
__device__ __forceinline__ int clip_prepare(const SetupParams& p, const float (*v)[4], float* ws, SubTri* st)
{
#pragma clang fp contract(off)
    float d1[4], d2[4];
#pragma unroll
    for (int c = 0; c < 4; c++) { d1[c] = v[1][c] - v[0][c]; d2[c] = v[2][c] - v[0][c]; }
    int num = 3;
    NVDR_CW(0, 0) = 0.f; NVDR_CW(0, 1) = 0.f; NVDR_CW(0, 2) = 1.f; NVDR_CW(0, 3) = 0.f; NVDR_CW(0, 4) = 0.f; NVDR_CW(0, 5) = 1.f;
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
        if ((v[0][3] < fabsf(v[0][ax])) || (v[1][3] < fabsf(v[1][ax])) || (v[2][3] < fabsf(v[2][ax]))) {
            num = clip_poly_plane(ws, 18, 0, num, v[0][3] + v[0][ax], d1[3] + d1[ax], d2[3] + d2[ax]);
            num = clip_poly_plane(ws, 0, 18, num, v[0][3] - v[0][ax], d1[3] - d1[ax], d2[3] - d2[ax]);
        }
    }
    int ns = 0;
    float c0[4], cp[4], cc[4];
#pragma unroll 1
    for (int i = 0; i < num; i++) {
        const float b0 = NVDR_CW(0, i * 2), b1 = NVDR_CW(0, i * 2 + 1);
#pragma unroll
        for (int c = 0; c < 4; c++)
            cc[c] = __fmaf_rn(d2[c], b1, __fmaf_rn(d1[c], b0, v[0][c]));
        if (i == 0) { for (int c = 0; c < 4; c++) c0[c] = cc[c]; }
        else if (i >= 2) {
            float t[3][4];
            for (int c = 0; c < 4; c++) { t[0][c] = c0[c]; t[1][c] = cp[c]; t[2][c] = cc[c]; }
            if (snap_cull_setup(p.vp, t, st[ns])) ns++;
        }
        for (int c = 0; c < 4; c++) cp[c] = cc[c];
    }
    return ns;
}

// Emits the survivors: the first into the triangle's own slot, the others into pool slots pool_slot, pool_slot + 1,
// ... (reserved by the caller), all in place.
__device__ __forceinline__ void clip_emit(const SetupParams& p, int n, int slot0, const SubTri* st, int ns, int id, int* s_hist, int pool_slot)
{
    if (ns == 0) { p.bbox[(size_t)n * p.slots + slot0] = kEmptyBox; return; }
    emit_record(p, n, slot0, st[0], id, s_hist, nullptr, nullptr);
    // With the worst-case pool (6 per triangle) the slot can never exceed the pool; with a smaller pool chosen by
    // the caller, sub-triangles beyond it are dropped HERE and the call is reported as short through poolPeak
    // (the caller grows the pool and repeats the call, cf. the reference's retry, RasterImpl.cpp:174-231).
    for (int k = 1; k < ns; k++)
        if (pool_slot + k - 1 < p.slots - p.poolBase)
            emit_record(p, n, p.poolBase + pool_slot + k - 1, st[k], id, s_hist, nullptr, nullptr);
}

// Vertices of triangle slot i of image n after the viewport-tile transform (TriangleSetup.inl:196-267).
// Returns the triangle id + 1, or 0 when the slot holds no valid triangle.
__device__ __forceinline__ int fetch_tri(const SetupParams& p, int n, int i, float (*v)[4])
{
#pragma clang fp contract(off)
    const int cnt = p.instance ? p.T : p.ranges[2 * n + 1];
    if (i >= cnt) return 0;
    const int t = i + (p.instance ? 0 : p.ranges[2 * n]);
    if ((uint32_t)t >= (uint32_t)p.T) return 0;                                                        // :228-233
    const uint32_t i0 = (uint32_t)p.tri[t * 3 + 0], i1 = (uint32_t)p.tri[t * 3 + 1], i2 = (uint32_t)p.tri[t * 3 + 2];
    if (i0 >= (uint32_t)p.V || i1 >= (uint32_t)p.V || i2 >= (uint32_t)p.V) return 0;                  // :241-248
    const float4* vb = (const float4*)p.pos + (p.instance ? (size_t)n * p.V : 0);
    const float4 q0 = vb[i0], q1 = vb[i1], q2 = vb[i2];
    // Viewport-tile transform (:262-267); identity when the image is one viewport.
    v[0][0] = __fmaf_rn(q0.x, p.vp.xs, q0.w * p.vp.xo); v[0][1] = __fmaf_rn(q0.y, p.vp.ys, q0.w * p.vp.yo); v[0][2] = q0.z; v[0][3] = q0.w;
    v[1][0] = __fmaf_rn(q1.x, p.vp.xs, q1.w * p.vp.xo); v[1][1] = __fmaf_rn(q1.y, p.vp.ys, q1.w * p.vp.yo); v[1][2] = q1.z; v[1][3] = q1.w;
    v[2][0] = __fmaf_rn(q2.x, p.vp.xs, q2.w * p.vp.xo); v[2][1] = __fmaf_rn(q2.y, p.vp.ys, q2.w * p.vp.yo); v[2][2] = q2.z; v[2][3] = q2.w;
    return t + 1;
}

constexpr uint32_t kClipBox = 0x0000FFFEu;               // LDS stage only: the slot's triangle waits for the clipper

// Common path of one triangle: its record and AABB go to the block's LDS stage; a triangle that needs the
// clipper is queued instead (its block processes the queue densely afterwards, see k_setup).
__device__ __forceinline__ void setup_one(const SetupParams& p, int n, int i, int* s_hist, uint4* stage, uint32_t* boxStage, int* clipq, int* clipn)
{
#pragma clang fp contract(off)
    float v[3][4];
    const int id = fetch_tri(p, n, i, v);
    if (!id) { *boxStage = kEmptyBox; return; }

    // Trivial reject: all vertices beyond one frustum plane (:271-283).
    if ((v[0][3] < fabsf(v[0][0])) || (v[0][3] < fabsf(v[0][1])) || (v[0][3] < fabsf(v[0][2]))) {
        bool out = false;
#pragma unroll
        for (int ax = 0; ax < 3; ax++) {
            out |= (v[0][3] < +v[0][ax]) & (v[1][3] < +v[1][ax]) & (v[2][3] < +v[2][ax]);
            out |= (v[0][3] < -v[0][ax]) & (v[1][3] < -v[1][ax]) & (v[2][3] < -v[2][ax]);
        }
        if (out) { *boxStage = kEmptyBox; return; }
    }

    bool inside = true;
#pragma unroll
    for (int k = 0; k < 3; k++)
        inside = inside && (v[k][3] >= fmaxf(fmaxf(fabsf(v[k][0]), fabsf(v[k][1])), fabsf(v[k][2])));

    if (inside) {                                                                    // :329-352
        SubTri st;
        if (snap_cull_setup(p.vp, v, st)) emit_record(p, n, i, st, id, s_hist, stage, boxStage);
        else *boxStage = kEmptyBox;
    } else {
        clipq[atomicAdd(clipn, 1)] = i;
        *boxStage = kClipBox;
    }
}

// The wave's 64 direct slots (slot0 + lane, AABBs in `box`, tile units) into the block's bin histogram s_hist = [3][nb]: count,
// largest slot + 1, INT_MAX - smallest slot per bin.  Neighbouring triangles lie in the same bin, so one LDS atomic per lane
// means up to 64 serialised same-address atomics, three times over (r04: a quarter of k_setup's wave-cycles waited on LDS).  The
// lanes of the wave's first two distinct first-bins are counted by ONE lane each -- their count is a popcount, and since the
// slots ascend with the lane number their largest / smallest slots are those of the highest / lowest lane -- the rest, and the
// further bins of triangles that span several, on their own.  Every lane of the wave must call it.
__device__ __forceinline__ void hist_add_wave(int* s_hist, int nb, int binsX, uint32_t box, int slot0)
{
    const int lane = lane_id();
    const bool live = (box & 255u) <= ((box >> 16) & 255u);              // (kEmptyBox / kClipBox: txlo > txhi)
    const int x0 = (int)(box & 255u) >> 3, y0 = (int)((box >> 8) & 255u) >> 3, x1 = (int)((box >> 16) & 255u) >> 3, y1 = (int)(box >> 24) >> 3;
    const int bf = y0 * binsX + x0;
    uint64_t todo = __ballot(live);
#pragma unroll 1
    for (int it = 0; it < 2 && todo; it++) {
        const int lead = __builtin_ctzll(todo);
        const int b0 = __builtin_amdgcn_readlane(bf, lead);
        const uint64_t m = __ballot(live && bf == b0) & todo;
        if (lane == lead) {
            atomicAdd(&s_hist[b0], (int)__popcll(m));
            atomicMax(&s_hist[nb + b0], slot0 + (63 - __builtin_clzll(m)) + 1);
            atomicMax(&s_hist[2 * nb + b0], 0x7FFFFFFF - (slot0 + lead));
        }
        todo &= ~m;
    }
    const int slot = slot0 + lane;
    if ((todo >> lane) & 1ull) {
        atomicAdd(&s_hist[bf], 1);
        atomicMax(&s_hist[nb + bf], slot + 1);
        atomicMax(&s_hist[2 * nb + bf], 0x7FFFFFFF - slot);
    }
    if (live && (x1 > x0 || y1 > y0)) {
        for (int by = y0; by <= y1; by++)
            for (int bx = (by == y0 ? x0 + 1 : x0); bx <= x1; bx++) {
                const int b = by * binsX + bx;
                atomicAdd(&s_hist[b], 1);
                atomicMax(&s_hist[nb + b], slot + 1);
                atomicMax(&s_hist[2 * nb + b], 0x7FFFFFFF - slot);
            }
    }
}

// Dynamic LDS: int s_hist[3 * binsX * binsY].
__global__ __launch_bounds__(256, 8) void k_setup(const SetupParams p_arg, int blocksPerImage)
{
    // The parameter block is read where it lies, in the kernarg segment (a by-value struct handed on by
    // reference is copied into scratch by every thread: measured 94 MB of HBM writes in an earlier version).
    (void)p_arg;
    const SetupParams& p = *(const SetupParams*)__builtin_amdgcn_kernarg_segment_ptr();
    // image-major work list cut into one contiguous chunk per XCD (an image's vertices are fetched into one L2)
    int bxi, byi, n;
    if (!decode_block(blocksPerImage, 1, p.N, bxi, byi, n)) return;
    extern __shared__ int s_hist[];                   // [3][bins]: count, max slot + 1, INT_MAX - min slot per bin
    const int nb = p.binsX * p.binsY;
    for (int b = threadIdx.x; b < 3 * nb; b += 256) s_hist[b] = 0;
    // Records are staged in LDS (one 64-byte record per thread) and written out as contiguous 1 KiB
    // rows: the L2 is write-through, so four 16-byte stores at a 64-byte stride per lane would reach
    // memory as four partial-line writes each (measured: 4x the bytes).  Slots of culled triangles
    // receive whatever the stage holds; their AABB marks them empty.
    __shared__ uint4 s_rec[256 * 4];
    // The clipper path (triangles crossing a frustum plane: up to seven sub-triangle setups) is an
    // order of magnitude longer than the common path; run inside the per-triangle pass it would keep
    // whole waves waiting for their few clipped lanes.  Pass 1 queues those triangles, pass 2 runs
    // the clipper on the queue with consecutive lanes.
    __shared__ int s_clipq[256];
    __shared__ int s_clipn, s_poolNeed, s_poolBase;
    if (threadIdx.x == 0) s_clipn = 0;
    __shared__ uint32_t s_box[256];                          // AABBs of the block's slots: written out as one 1 KiB row
    __syncthreads();
    const int i0 = bxi * 256;
    setup_one(p, n, i0 + threadIdx.x, s_hist, s_rec + threadIdx.x * 4, &s_box[threadIdx.x], s_clipq, &s_clipn);
    hist_add_wave(s_hist, nb, p.binsX, s_box[threadIdx.x], i0 + (int)(threadIdx.x & ~63u));      // (each thread reads the AABB it staged itself)
    __syncthreads();
    {
        // Slots waiting for the clipper are left out (pass 2 writes them in place).
        const int cnt = p.instance ? p.T : p.ranges[2 * n + 1];
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int slot0 = bxi * 256 + wave * 64;
        uint4* dst = p.rec + ((size_t)n * p.slots + slot0) * 4;
        const uint4* src = s_rec + wave * 64 * 4;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int q = k * 64 + lane;                       // 16-byte chunk inside the wave's 4 KiB
            if (slot0 + (q >> 2) < cnt && s_box[wave * 64 + (q >> 2)] != kClipBox) dst[q] = src[q];
        }
        if (slot0 + lane < cnt && s_box[threadIdx.x] != kClipBox) p.bbox[(size_t)n * p.slots + slot0 + lane] = s_box[threadIdx.x];
    }
    // Pass 2, rounds of up to kClipLanes queued triangles (uniform: blocks without any skip all of it).  Every
    // triangle is clipped ONCE, its sub-triangles wait in private memory while the round's pool slots are reserved
    // with one global atomic (a returning atomic per clipped triangle on the image's counter serialises: measured
    // 0.5 ms on the stress scene), then they are written in place.
    if (s_clipn > 0) {
        __syncthreads();                                     // the record stage has left: it becomes the clipper's workspace
        float* wsAll = (float*)s_rec;
        for (int base = 0; base < s_clipn; base += kClipLanes) {
            if (threadIdx.x == 0) { s_poolNeed = 0; s_poolBase = 0; }
            __syncthreads();
            const bool act = (int)threadIdx.x < kClipLanes && base + (int)threadIdx.x < s_clipn;
            SubTri st[kSubPerTri];
            int ns = 0, ci = 0, id = 0, poolOff = 0;
            if (act) {
                ci = s_clipq[base + threadIdx.x];
                float v[3][4];
                id = fetch_tri(p, n, ci, v);
                ns = clip_prepare(p, v, wsAll + threadIdx.x, st);
                if (ns > 1) poolOff = atomicAdd(&s_poolNeed, ns - 1);
            }
            __syncthreads();
            if (threadIdx.x == 0 && s_poolNeed > 0) s_poolBase = atomicAdd(&p.poolCount[n], s_poolNeed);
            __syncthreads();
            if (act) clip_emit(p, n, ci, st, ns, id, s_hist, s_poolBase + poolOff);
            __syncthreads();                                 // every lane has read s_poolBase before the next round resets it
        }
    }
    __syncthreads();
    // One global atomic per non-empty bin per block (instead of one per triangle).
    for (int b = threadIdx.x; b < nb; b += 256) {
        const int c = s_hist[b];
        if (c) {
            const size_t o = (size_t)n * nb + b;
            atomicAdd(&p.binCount[o], c);
            if (s_hist[nb + b]) { atomicMax(&p.binHi[o], s_hist[nb + b]); atomicMax(&p.binLoInv[o], s_hist[2 * nb + b]); }
        }
    }
}

// ---------------------------------------------------------------------------------
// Per-bin triangle lists for large meshes (the reference's bin stage, BinRaster.inl:60-170,319-377 + CoarseRaster.inl:149-218,
// does this for every mesh: per-CTA batches of triangles compacted into 512-entry segments per bin, merged in triangle order).
// k_fine's filter walks a bin's RANGE of triangle slots, which is short when the index order follows the surface and the
// mesh is small, and is the whole mesh for a soup -- 256 bins x 1 M AABBs per image at 1024^2.  From kListMinTris triangles on,
// three steps after k_setup turn the per-bin counts it already made into exact lists:
//   k_binscan  one workgroup: which bins are better off with a list (range > kListScanFactor * triangles + kListScanBias),
//              exclusive prefix sum of their counts = where each list starts; lists beyond the buffer's capacity stay
//              range-scanned, so nothing can overflow and the host is not asked anything;
//   k_binfill  a workgroup per kFillSlots slots: LDS histogram of its (triangle, bin) pairs, ONE returning global atomic per
//              touched bin to reserve the block's stretch of that bin's list, then the slots are written -- in no particular
//              order: visibility is an order-free minimum (k_fine), so the lists need no sorting and no merge;
//   k_fine     reads list entries and gathers their AABBs instead of filtering.
// ---------------------------------------------------------------------------------
struct ListParams {
    const uint32_t* bbox; const int* ranges; const int* poolFinal;
    int* binCursor; int* anyList; uint32_t* binList;
    int instance, N, T, poolBase, slots, binsX, binsY;
    long long cap;                                  // list entries the buffer holds
};

__global__ __launch_bounds__(1024) void k_binscan(const int* __restrict__ binCount, const int* __restrict__ binHi, const int* __restrict__ binLoInv,
                                                  const int* __restrict__ poolCount, int* __restrict__ binCursor, int* __restrict__ anyList,
                                                  int totalBins, int binsPerImage, int N, int poolMax, long long cap, int factor, int bias)
{
    __shared__ long long s_wave[16];
    const int per = (totalBins + 1023) / 1024;
    const int b0 = min((int)threadIdx.x * per, totalBins), b1 = min(b0 + per, totalBins);
    for (int n = threadIdx.x; n < N; n += 1024) anyList[n] = 0;
    __syncthreads();
    auto wants = [&](int i) -> int {                // the bin's triangle count when a list beats the range scan, else 0
        const int c = binCount[i];
        if (c <= 0) return 0;
        const int hiSlot = binHi[i];
        const int scanLo = hiSlot ? ((0x7FFFFFFF - binLoInv[i]) & ~3) : 0;
        const int dlen = hiSlot ? (((hiSlot + 3) & ~3) - scanLo) : 0;
        const long long range = (long long)dlen + min(poolCount[i / binsPerImage], poolMax);
        return range > (long long)factor * c + bias ? c : 0;
    };
    long long sum = 0;
    for (int i = b0; i < b1; i++) sum += wants(i);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const long long v = __shfl_up(incl, d); if (lane >= d) incl += v; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    long long off = incl - sum;
    for (int w = 0; w < wave; w++) off += s_wave[w];
    for (int i = b0; i < b1; i++) {
        const int c = wants(i);
        int cur = -1;
        if (c > 0 && off + c <= cap) { cur = (int)off; anyList[i / binsPerImage] = 1; }
        if (c > 0) off += c;                            // (a list that does not fit leaves a hole: later, shorter ones may still fit)
        binCursor[i] = cur;
    }
}

// Wave-level increment of an LDS counter per lane's bin: the lanes of the first two distinct bins are counted with one atomic
// each (the common case, neighbouring triangles in one bin, would otherwise be 64 serialised same-address atomics), the rest
// on their own.  Returns the lane's rank in its bin.  Every lane of the wave must call it.
__device__ __forceinline__ int bin_rank(int* cnt, int b, bool valid)
{
    const int lane = lane_id();
    uint64_t todo = __ballot(valid);
    int rank = 0;
#pragma unroll 1
    for (int it = 0; it < 2 && todo; it++) {
        const int lead = __builtin_ctzll(todo);
        const int b0 = __builtin_amdgcn_readlane(b, lead);
        const uint64_t m = __ballot(valid && b == b0) & todo;
        int base = 0;
        if (lane == lead) base = atomicAdd(&cnt[b0], (int)__popcll(m));
        base = __builtin_amdgcn_readlane(base, lead);
        if ((m >> lane) & 1ull) rank = base + mask_rank(m);
        todo &= ~m;
    }
    if ((todo >> lane) & 1ull) rank = atomicAdd(&cnt[b], 1);
    return rank;
}

__global__ __launch_bounds__(kFillThreads) void k_binfill(const ListParams p, int blocksPerImage)
{
    __shared__ int s_cnt[1024], s_base[1024];
    const int n = blockIdx.x / blocksPerImage, blk = blockIdx.x - n * blocksPerImage;
    if (!p.anyList[n]) return;
    const int nb = p.binsX * p.binsY;
    const int direct = p.instance ? p.T : min(p.ranges[2 * n + 1], p.poolBase);
    const int poolEnd = p.poolBase + min(p.poolFinal[n], p.slots - p.poolBase);
    const int blockLo = blk * kFillSlots, blockHi = blockLo + kFillSlots;
    if (blockLo >= direct && (blockHi <= p.poolBase || blockLo >= poolEnd)) return;      // no live slot in this stretch
    for (int b = threadIdx.x; b < nb; b += kFillThreads) { s_cnt[b] = 0; s_base[b] = p.binCursor[(size_t)n * nb + b]; }   // < 0: no list for this bin
    __syncthreads();
    const int slot0 = blockLo + (int)threadIdx.x * 4;
    uint32_t box[4] = {kEmptyBox, kEmptyBox, kEmptyBox, kEmptyBox};
    if (slot0 < p.slots) {
        const uint4 b4 = *(const uint4*)(p.bbox + (size_t)n * p.slots + slot0);
        box[0] = b4.x; box[1] = b4.y; box[2] = b4.z; box[3] = b4.w;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int s = slot0 + k;
            if (!(s < direct || (s >= p.poolBase && s < poolEnd))) box[k] = kEmptyBox;
        }
    }
    // One pass counts, one writes: pass 0 leaves the block's pairs per bin in s_cnt, the bins' stretches are reserved, pass 1
    // hands out ranks again (any order will do) and stores.
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
#pragma unroll 1
        for (int k = 0; k < 4; k++) {
            const uint32_t bx = box[k];
            const int x0 = (int)(bx & 255u) >> 3, y0 = (int)((bx >> 8) & 255u) >> 3, x1 = (int)((bx >> 16) & 255u) >> 3, y1 = (int)(bx >> 24) >> 3;
            const bool live = (bx & 255u) <= ((bx >> 16) & 255u);                  // kEmptyBox: txlo > txhi
            // first bin of every lane together (wave-aggregated), further bins of large triangles on their own
            const int bf = y0 * p.binsX + x0;
            const bool vf = live && s_base[live ? bf : 0] >= 0;
            const int rf = bin_rank(s_cnt, bf, vf);
            if (pass && vf) p.binList[(size_t)s_base[bf] + rf] = (uint32_t)(slot0 + k);
            if (live && (x1 > x0 || y1 > y0)) {
                for (int by = y0; by <= y1; by++)
                    for (int bxx = (by == y0 ? x0 + 1 : x0); bxx <= x1; bxx++) {
                        const int b = by * p.binsX + bxx;
                        if (s_base[b] < 0) continue;
                        const int r = atomicAdd(&s_cnt[b], 1);
                        if (pass) p.binList[(size_t)s_base[b] + r] = (uint32_t)(slot0 + k);
                    }
            }
        }
        __syncthreads();
        if (pass == 0) {
            for (int b = threadIdx.x; b < nb; b += kFillThreads) {
                const int c = s_cnt[b];
                if (c > 0) s_base[b] = atomicAdd(&p.binCursor[(size_t)n * nb + b], c);
                s_cnt[b] = 0;
            }
            __syncthreads();
        }
    }
}

// Heavy-first work order.  Work items (image, bin) are partitioned into 8 contiguous chunks,
// one per XCD (block b of k_fine runs on XCD b % 8, so an image's bins share one L2); inside
// its chunk each XCD visits the bins with the most triangles first, which keeps the long
// bins off the tail of the launch.  Counting sort by log2 bucket, one block per chunk.
// Each item is written as {work, triangle count, first scanned slot, scanned length} so that k_fine starts
// from ONE load instead of a chain of dependent ones (order -> count / slot range -> AABBs).
// The kernel is the last reader of k_setup's counters and zeroes them on the way out: the control
// block is left as the next rasterize call needs it (no memset launch in front of every call).
__global__ __launch_bounds__(1024) void k_order(int* __restrict__ binCount, int* __restrict__ binHi, int* __restrict__ binLoInv,
                                                int* __restrict__ poolCount, int* __restrict__ poolFinal, int* __restrict__ poolPeak, int N,
                                                int4* __restrict__ order, int totalBins,
                                                int* __restrict__ splitInfo, int4* __restrict__ helpers, int* __restrict__ splitDone,
                                                int splitTris, int splitPart, int* __restrict__ chunkNz, int binsX, int binsY,
                                                const int* __restrict__ binCursor, int maxParts, int helpersPerChunk, int* __restrict__ splitKeyBase,
                                                int interleave, int* __restrict__ splitBin)
{
    __shared__ int s_bucket[32];
    __shared__ int4 s_split[kSplitsPerChunk];
    __shared__ int s_hbase[kSplitsPerChunk];
    if (threadIdx.x < kSplitsPerChunk) s_split[threadIdx.x] = make_int4(0, 0, 0, 0);
    if (splitTris != 0x7FFFFFFF)
        for (int h = threadIdx.x; h < helpersPerChunk; h += 1024) helpers[blockIdx.x * helpersPerChunk + h] = make_int4(-1, 0, 0, 0);
    const int perXcd = (totalBins + 7) >> 3;
    // The chunk's bins: a contiguous eighth of the (image-major) list -- an image's records in one L2 -- or, `interleave`, every
    // eighth bin: launches over fewer images than XCDs, where contiguous chunks would put all of an image's triangles, however
    // many workgroups share its bins, on one or two XCDs (a million-triangle mesh at batch 2: k_fine 1.3 -> 0.8 ms shared, all on
    // two XCDs).  Unused slots at the end of an interleaved chunk are marked for k_fine.
    const int lo = blockIdx.x * perXcd;
    // j-th bin of this chunk, or -1: contiguous, or one of every eight consecutive bins -- WHICH one rotates pseudo-randomly
    // from group to group, so that no line through the image (an edge-on mesh is one) lands on a single XCD
    auto chunk_bin = [&](int j) -> int {
        const int i = interleave ? 8 * j + (int)(((uint32_t)blockIdx.x + (((uint32_t)j * 0x9E3779B1u) >> 29)) & 7u) : lo + j;
        return i < totalBins ? i : -1;
    };
    for (int j = threadIdx.x; j < perXcd; j += 1024) order[lo + j] = make_int4(-1, 0, 0, 0);      // (slots that stay unused: k_fine leaves)
    if (threadIdx.x < 32) s_bucket[threadIdx.x] = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < perXcd; j += 1024) {
        const int i = chunk_bin(j);
        if (i < 0) continue;
        int c = binCount[i];
        int bk = c > 0 ? 32 - __clz(c) : 0;              // 0, 1, 2-3, 4-7, ...
        atomicAdd(&s_bucket[31 - bk], 1);                 // descending
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int k = 0; k < 32; k++) { int c = s_bucket[k]; s_bucket[k] = acc; acc += c; }
        chunkNz[blockIdx.x] = s_bucket[31];               // bins with triangles in this chunk: the empty ones follow them in the order
    }
    __syncthreads();
    for (int j = threadIdx.x; j < perXcd; j += 1024) {
        const int i = chunk_bin(j);
        if (i < 0) continue;
        int c = binCount[i];
        int bk = c > 0 ? 32 - __clz(c) : 0;
        int pos = atomicAdd(&s_bucket[31 - bk], 1);
        const int hiSlot = binHi[i];
        const int scanLo = hiSlot ? ((0x7FFFFFFF - binLoInv[i]) & ~3) : 0;
        const int dlen = hiSlot ? (((hiSlot + 3) & ~3) - scanLo) : 0;
        int4 ent = make_int4(i, c, scanLo, dlen);
        if (binCursor && c > 0) {                          // the bin has a triangle list (k_binscan): where it starts, and minus its length
            const int off = binCursor[i];
            if (off >= 0) { ent.z = off; ent.w = -c; }       // (helper items carry no count of their own: the list's length travels in w)
        }
        if (c == 0) {                                      // an empty bin: its coordinates, for the workgroup that clears it (k_fine)
            const int n = i / (binsX * binsY), b = i - n * (binsX * binsY), by = b / binsX;
            ent.z = (n << 10) | (by << 5) | (b - by * binsX);
        }
        order[lo + pos] = ent;
        // Bins with very many triangles (edge-on meshes) would each keep one workgroup busy for as long as the whole
        // launch takes: their slot range is shared by up to kSplitMaxParts workgroups (k_fine merges the parts' keys).
        // The kSplitsPerChunk heaviest bins of the chunk qualify: their position in the heavy-first order is their split number.
        if (c >= splitTris && pos < kSplitsPerChunk)
            s_split[pos] = make_int4(i, c, ent.z, ent.w);      // .y: its triangles, for now
        splitInfo[lo + pos] = 0;
        binCount[i] = 0; binHi[i] = 0; binLoInv[i] = 0;   // this thread was the bin's only reader in this pass
    }
    // Helper items of the shared bins, packed at the front of the chunk's helper slots (heaviest bin first): the parts a bin
    // gets are what it wants while the chunk's helper slots last (a bin left with one part is not shared); its parts' key
    // arrays are consecutive in the exchange buffer, from splitKeyBase on.
    __syncthreads();
    if (splitTris != 0x7FFFFFFF) {                          // (uniform: launches that share no bin skip all of this)
        __shared__ int s_kbase[kSplitsPerChunk];
        if (threadIdx.x == 0) {
            // triangles per part: as asked, or more when the chunk's helper slots would not last -- then every shared bin gets
            // parts of the same size instead of the heaviest ones taking all slots (sum(ceil(c / size) - 1) < sum(c) / size <= slots)
            long long sumC = 0;
            for (int q = 0; q < kSplitsPerChunk; q++) sumC += s_split[q].y;
            const int partSize = max(splitPart, (int)((sumC + helpersPerChunk - 1) / helpersPerChunk));
            int usedH = 0, usedK = 0;
            for (int q = 0; q < kSplitsPerChunk; q++) {
                const int c = s_split[q].y;
                int parts = c ? max(2, min(maxParts, (c + partSize - 1) / partSize)) : 0;
                if (parts) parts = min(parts, 1 + helpersPerChunk - usedH);
                if (parts < 2) { s_split[q].y = 0; continue; }
                s_split[q].y = ((blockIdx.x * kSplitsPerChunk + q) << (2 * kPartBits)) | parts;      // split number of this call, parts
                s_hbase[q] = usedH; s_kbase[q] = usedK;
                usedH += parts - 1; usedK += parts;
            }
        }
        __syncthreads();
        if (threadIdx.x < kSplitsPerChunk) {
            const int4 e = s_split[threadIdx.x];
            const int gs = blockIdx.x * kSplitsPerChunk + threadIdx.x;
            splitBin[gs] = e.y ? e.x : -1;
            if (e.y) {
                splitKeyBase[gs] = blockIdx.x * (helpersPerChunk + kSplitsPerChunk) + s_kbase[threadIdx.x];
                splitDone[gs] = 0;
                splitInfo[lo + threadIdx.x] = e.y;                      // (the bin's position in the heavy-first order is its split's)
                for (int k = 1; k < (e.y & kPartMask); k++)
                    helpers[blockIdx.x * helpersPerChunk + s_hbase[threadIdx.x] + k - 1] = make_int4(e.x, e.y | (k << kPartBits), e.z, e.w);
            }
        }
    }
    if (blockIdx.x == 0)
        for (int n = threadIdx.x; n < N; n += 1024) {
            const int need = poolCount[n];
            poolFinal[n] = need; poolCount[n] = 0;
            if (poolPeak) atomicMax(poolPeak, need);             // largest per-image demand of this call (all viewport tiles)
        }
}

// ---------------------------------------------------------------------------------
// Work order of the kernels that consume the image (nvdr_device.hpp TileFlags): the 64x64-pixel bins with a covered
// tile first, in image-major order, then the others; [nBins] = how many of the former.  One workgroup (the list needs a
// prefix sum over all bins), every thread a run of consecutive bins; runs after k_fine, whose waves -- each shades one
// row of eight tiles of a bin -- left one byte per bin and tile row: one 8-byte word per bin here, where the flags
// themselves are 64 bytes per bin in eight places (14 -> 7 us at the headline batch).
// ---------------------------------------------------------------------------------
constexpr int kFlagOrderThreads = 1024;
constexpr int kFlagOrderSingle = 4096;                     // up to this many bins one workgroup does it all in one launch
constexpr int kFlagOrderChunk = 2048;                      // beyond: bins per workgroup of the two-launch form (count, then place)
constexpr int kFlagOrderMaxGroups = kOrderMaxBins / kFlagOrderChunk;      // 32 partial counts (raster scratch, behind chunkNz)
// PHASE 0: the whole list by one workgroup (batches up to kFlagOrderSingle bins: one launch, 7 us at the headline batch).
// PHASE 1 / 2: larger batches, where one workgroup's walk grows linearly with the batch (81 us at 16384 bins): every workgroup
// counts the covered bins of its stretch into partial[] (1), then -- next launch -- places its stretch behind the stretches before
// it (2).  The list is the same in both forms: covered bins in image-major order, then the others.
template <int PHASE>
__global__ __launch_bounds__(kFlagOrderThreads) void k_flag_order(TileFlags t, const uint8_t* __restrict__ rowCov, int* __restrict__ order,
                                                                  int* __restrict__ partial, int groups)
{
    __shared__ int s_wave[kFlagOrderThreads / 64];
    const int per = (t.nBins + groups * kFlagOrderThreads - 1) / (groups * kFlagOrderThreads);         // <= 64 (kOrderMaxBins)
    const int b0 = min(((int)blockIdx.x * kFlagOrderThreads + (int)threadIdx.x) * per, t.nBins), b1 = min(b0 + per, t.nBins);
    unsigned long long mask = 0ull;                         // bit i: bin b0 + i has a covered tile
    {
        // k_fine left one byte per bin and tile row (FineParams::rowCov): one 8-byte word per bin; rows beyond the image were
        // not written
        const int bpi = t.binsX * t.binsY;
        int by = (b0 % bpi) / t.binsX, bx = b0 % t.binsX;
#pragma unroll 4
        for (int b = b0; b < b1; b++) {
            const int rows = min(8, t.h - by * 8);
            const unsigned long long v = *(const unsigned long long*)(rowCov + (size_t)b * 8) & (rows >= 8 ? ~0ull : ((1ull << (rows * 8)) - 1ull));
            if (v) mask |= 1ull << (b - b0);
            if (++bx == t.binsX) { bx = 0; if (++by == t.binsY) by = 0; }
        }
    }
    const int c = __popcll(mask);
    // inclusive scan of c over the workgroup: within the wave by shuffles, across the 16 waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int before = 0, nCov = 0;
    for (int w = 0; w < kFlagOrderThreads / 64; w++) { const int v = s_wave[w]; if (w < wave) before += v; nCov += v; }
    if (PHASE == 1) { if (threadIdx.x == 0) partial[blockIdx.x] = nCov; return; }
    int groupBase = 0;                                      // covered bins in the stretches before this workgroup's
    if (PHASE == 2) {
        int all = 0;
        for (int g = 0; g < groups; g++) { const int v = partial[g]; if (g < (int)blockIdx.x) groupBase += v; all += v; }
        nCov = all;
    }
    int co = groupBase + before + incl - c;                 // covered bins in front of this thread's run
    int eo = nCov + (b0 - co);                              // empty ones, behind all the covered
    for (int b = b0; b < b1; b++) {
        if ((mask >> (b - b0)) & 1ull) order[co++] = b; else order[eo++] = b;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) order[t.nBins] = nCov;
}

// ---------------------------------------------------------------------------------
// Fine raster + pixel shader
// ---------------------------------------------------------------------------------

struct FineParams {
    const uint4* rec; const uint32_t* bbox; const int* poolFinal; const int* ranges;
    const int4* order;
    const int* splitInfo; const int4* helpers;          // bins shared by several workgroups (k_order)
    unsigned long long* splitKeys; int* splitDone;      // their parts' key arrays [64 tiles][64 px] and arrival counters
    const int* splitKeyBase;                            // per split: number of its first key array
    const int* splitBin;                                // per split: its bin (work item), -1 = unused (the shading launch of LIST launches)
    const int* chunkNz;                                 // per XCD chunk: bins with triangles (in the plain instantiation each also clears one empty bin)
    const uint32_t* binList;                            // triangle lists of the bins that have one (k_binfill), else nullptr
    uint8_t* tileFlags; int tfW, tfH;                   // out, optional: per 8x8 tile of the image, 1 = some pixel shows a triangle (nvdr_device.hpp TileFlags)
    uint8_t* rowCov;                                    // out, optional: per bin and tile row, 1 = some tile of that row of the bin is flagged (k_flag_order reads these)
    const float* pos; const int* tri;
    int instance, N, V, T, maxTri, poolBase, slots;
    int W, H, Wp, Hp;              // image size and padded surface size
    Viewport vp;
    int binsX, binsY, totalBins;   // bins per viewport tile, N*binsX*binsY
    const uint32_t* peel; uint32_t* depth;
    float* out; float* out_db;
    float xs, xo, ys, yo;          // pixel -> clip transform of the whole image (torch_rasterize.cpp:146-149)
    int dbg;
    unsigned long long* dbgbuf;    // development: per-workgroup phase timestamps
};


constexpr int kEarlyZTiles = 4;                          // tile bounds a wave refreshes per batch of 64 pairs once covered tiles see more pairs (k_fine)

// Refresh cadence of the per-tile depth bounds (k_fine): kEarlyZTiles tiles per batch of 64 pairs once covered tiles see more
// pairs; one after a batch that held a mask with more than kCoop fragments; otherwise one every kColdRefresh batches.
constexpr int kColdRefresh = 8;

struct FineShared {
    uint32_t slot[kListCap];                               // bin triangle list: record slot of each entry
    uint32_t box[kListCap];                                // packed tile AABBs of the list entries
    unsigned long long key[kBinTiles][kBinTiles][64];      // per-pixel visibility keys of the bin [tileY][tileX][pixel]
    uint16_t pfx[kListCap + 64];                           // exclusive prefix of the entries' (triangle, tile) pair counts (<= 448 x 64)
    uint32_t tileZ[kBinTiles * kBinTiles];                 // per tile: an upper bound (upper half + 1) of what every pixel's depth will end up at most
    uint16_t bstart[kListCap];                             // per 64 pairs: the list entry that holds pair 64 k (at most kListCap x 64 pairs)
    int count;
    int pending;                                           // some wave has not scanned its share of the bin's slot range to the end
    int totalPairs;
    int ticket;                                            // rows of a partner bin handed to the waves as they finish (clear_bin)
};

// Rasterise up to 64 (triangle, tile) pairs, one pair per lane.
//  1. coverage: each lane walks its triangle's three edge functions over the 8x8 pixel
//     centres of its tile (integer adds only) and packs the signs into a 64-bit mask;
//  2. fragments: each lane pops the set bits of its mask, evaluates the U32 depth plane
//     and merges depth<<32|~id into the tile's key array with an LDS 64-bit atomic min.
// Bit b of the mask is pixel (x, y) = (b & 7, b >> 3) of the tile.
// Returns bit 0: some pair of the batch belongs to a tile with a finite depth bound (the cull ran); bit 1: the batch held a mask
// with more than kCoop fragments (large triangles: the scenes in which tiles get covered and bounds pay).
template <bool PEEL, bool DBG = false>
__device__ __forceinline__ int raster_pairs(FineShared& sh, const FineParams& p, const uint4* __restrict__ grec,
                                             int lane, int n, bool act, uint32_t q, int btx0, int bty0,
                                             unsigned long long* dbgNonEmpty = nullptr, bool ezOn = true)
{
    // q = list entry | tile column << 9 | tile row << 12 (inside the bin) of this lane's pair
    const int e = act ? (int)(q & 511u) : 0;
    const int tx = act ? (int)((q >> 9) & 7u) : 0;
    const int tyl = act ? (int)((q >> 12) & 7u) : 0;
    const int X0 = (btx0 + tx) * 8, Y0 = (bty0 + tyl) * 8;

    // Each lane gathers its triangle's 64-byte record from L2 (the image's records are L2 resident).
    const uint4* r = grec + (size_t)(act ? sh.slot[e] : sh.slot[0]) * 4;
    const uint4 q0 = r[0], q1 = r[1], q2 = r[2], q3 = r[3];
    int A0 = (int)q0.x, B0 = (int)q0.y;
    int A1 = (int)q0.w, B1 = (int)q1.x;
    int A2 = (int)q1.z, B2 = (int)q1.w;
    // Edge values at the tile's first pixel; wrapping 32-bit arithmetic, 24-bit operands.
    uint32_t e0 = q0.z + (uint32_t)__mul24(A0, X0) + (uint32_t)__mul24(B0, Y0);
    uint32_t e1 = q1.y + (uint32_t)__mul24(A1, X0) + (uint32_t)__mul24(B1, Y0);
    uint32_t e2 = q2.x + (uint32_t)__mul24(A2, X0) + (uint32_t)__mul24(B2, Y0);

    // Per-tile depth cull (the reference's early-Z, FineRaster.inl:13-34,67-71,282, with the triangle's depth PLANE over this tile
    // instead of its zmin): sh.tileZ bounds from above what every pixel of the tile can end up with (refresh_tile_bound); a pair
    // whose plane lies wholly behind that over the tile's 8x8 pixels cannot win one.  lb = a lower bound of the depth at any
    // covered pixel: the triangle's zmin (record q3.z), raised to the plane's smallest corner value D0 + min(0, 7 zx) + min(0, 7 zy)
    // when the slopes are small enough for the wrapped U32 arithmetic to be undone (D0 = depth at the tile origin, known modulo
    // 2^32; every covered pixel's depth D lies in [zmin, 2^32) and D0 = D - x zx - y zy with x, y < 8, so D0 - zmin lies in
    // [-S, 2^32 + S) with S = 7 (|zx| + |zy|) < 2^31: t = (d0 - zmin) mod 2^32 below 2^32 - S IS that difference or, when the true
    // value is 2^32 larger still, below it).  A culled pair gets an edge function that is negative everywhere: empty mask.
    const uint32_t zx = q2.y, zy = q2.z;
    const uint32_t d0 = q2.w + zx * (uint32_t)X0 + zy * (uint32_t)Y0;   // depth at the tile origin
    // (All of it under a wave-uniform test: a pair is a candidate only once its tile is covered completely -- in a scene without
    // overdraw no more pairs arrive for such a tile, and the wave skips the arithmetic.)
    bool hot = false;
    {
        const uint32_t bound = ezOn ? sh.tileZ[tyl * kBinTiles + tx] : 0xFFFFFFFFu;
        const bool cand = act && bound != 0xFFFFFFFFu;
        if (ezOn && __ballot(cand)) {
            hot = true;
            const int sx = (int)zx, sy = (int)zy;
            uint32_t lb = q3.z;
            if ((uint32_t)(sx + (1 << 27)) < (1u << 28) && (uint32_t)(sy + (1 << 27)) < (1u << 28)) {
                const uint32_t S = 7u * (uint32_t)(abs(sx) + abs(sy));
                const uint32_t nn = 7u * (uint32_t)(max(-sx, 0) + max(-sy, 0));
                const uint32_t t = d0 - q3.z;
                if (t < 0u - S && t >= nn) { const uint32_t r = q3.z + (t - nn); lb = r < q3.z ? 0xFFFFFFFFu : r; }
            }
            if (cand && (lb >> 16) >= bound) { e0 = 0x80000000u; A0 = 0; B0 = 0; }
        }
    }
    const int kHot = hot ? 1 : 0;
    if (DBG && (p.dbg & 256)) { if (q0.x == 0x12345u && q3.w == 77u) sh.key[0][0][lane] = q1.x; return 0; }      // experiment: no coverage, no fragments
    // Coverage.  Bit y * 8 + x of the mask is pixel (x, y) of the tile.
    uint64_t m;
    // Two pixels per instruction, in 16-bit halves.  A, B are multiples of 16 (emit_record), so with C = 16 c1 + c0, 0 <= c0 < 16,
    // E >= 0  <=>  E' = c1 + (A/16) X + (B/16) Y >= 0: four bits less.  Only the SIGN of E' matters, and over the tile's pixels E'
    // moves by at most 7 (|A'| + |B'|) from its value at the origin, so that value can be clamped to +-16000 as long as that
    // excursion stays below 14000 (|A'| + |B'| <= 2000: edges up to 125 px): a clamped start stays 2000 away from zero on its side,
    // an unclamped one is exact, and nothing leaves the 16-bit range.  Low half = rows 0..3, high half = rows 4..7 (own start values,
    // clamped separately); a wave holding a longer edge walks in 32 bits as before.
    const int a0 = A0 >> 4, b0 = B0 >> 4, a1 = A1 >> 4, b1 = B1 >> 4, a2 = A2 >> 4, b2 = B2 >> 4;
    const bool smallEdges = !act || (abs(a0) + abs(b0) <= 2000 && abs(a1) + abs(b1) <= 2000 && abs(a2) + abs(b2) <= 2000);
    if (__ballot(!smallEdges) == 0ull) {
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        auto pack = [](int lo, int hi) -> uint32_t { return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u); };
        auto start = [&](uint32_t e, int b) -> uint32_t {
            const int v = (int)e >> 4;
            return pack(min(max(v, -16000), 16000), min(max(v + 4 * b, -16000), 16000));
        };
        auto add2 = [](uint32_t x, uint32_t y) -> uint32_t { return __builtin_bit_cast(uint32_t, (us2)(__builtin_bit_cast(us2, x) + __builtin_bit_cast(us2, y))); };
        auto shr2 = [](uint32_t x) -> uint32_t { return __builtin_bit_cast(uint32_t, (us2)(__builtin_bit_cast(us2, x) >> (us2)(1))); };
        uint32_t r0 = start(e0, b0), r1 = start(e1, b1), r2 = start(e2, b2);
        const uint32_t sx0 = pack(a0, a0), sx1 = pack(a1, a1), sx2 = pack(a2, a2);
        const uint32_t sy0 = pack(b0 - 7 * a0, b0 - 7 * a0), sy1 = pack(b1 - 7 * a1, b1 - 7 * a1), sy2 = pack(b2 - 7 * a2, b2 - 7 * a2);
        uint32_t accA = 0u, accB = 0u;              // rows {0,1 | 4,5} and {2,3 | 6,7}: sign bits (= outside) enter at the top of each half
#pragma unroll
        for (int rp = 0; rp < 4; rp++) {
            uint32_t acc = rp < 2 ? accA : accB;
#pragma unroll
            for (int x = 0; x < 8; x++) {
                acc = shr2(acc) | ((r0 | r1 | r2) & 0x80008000u);
                if (x < 7) { r0 = add2(r0, sx0); r1 = add2(r1, sx1); r2 = add2(r2, sx2); }
            }
            if (rp < 2) accA = acc; else accB = acc;
            if (rp < 3) { r0 = add2(r0, sy0); r1 = add2(r1, sy1); r2 = add2(r2, sy2); }
        }
        const uint32_t outLo = __builtin_amdgcn_perm(accB, accA, 0x05040100u);      // rows 0..3
        const uint32_t outHi = __builtin_amdgcn_perm(accB, accA, 0x07060302u);      // rows 4..7
        m = act ? ~(((uint64_t)outHi << 32) | outLo) : 0ull;
    } else {
        // one pixel per step in 32 bits (the rare wave that holds an edge longer than 125 px -- rolled up, so that
        // it costs the common path neither registers nor code)
        uint32_t half[2] = {0u, 0u};             // rows 0..3, rows 4..7 (sign bits = outside), first pixel in the top bit
#pragma unroll
        for (int h = 0; h < 2; h++) {
            uint32_t bits = 0u;
#pragma unroll 1
            for (int y = 0; y < 4; y++) {
                uint32_t r0 = e0, r1 = e1, r2 = e2;
#pragma unroll
                for (int x = 0; x < 8; x++) {
                    bits = __builtin_amdgcn_alignbit(bits, r0 | r1 | r2, 31);     // (bits << 1) | sign
                    r0 += (uint32_t)A0; r1 += (uint32_t)A1; r2 += (uint32_t)A2;
                }
                e0 += (uint32_t)B0; e1 += (uint32_t)B1; e2 += (uint32_t)B2;
            }
            half[h] = bits;
        }
        // (this walk leaves pixel (x, y) in bit 63 - (y * 8 + x): reversed into the mask's order)
        m = act ? ~(((uint64_t)__builtin_bitreverse32(half[1]) << 32) | __builtin_bitreverse32(half[0])) : 0ull;
    }
    if (DBG && dbgNonEmpty) *dbgNonEmpty += __popcll(__ballot(m != 0));
    if (__ballot(m != 0) == 0) return kHot;
    if (DBG && (p.dbg & 128)) { if (m == 0x123456789ull) sh.key[0][0][lane] = m; return kHot; }                       // experiment: no fragment loop

    const uint32_t idk = ~q3.x;
    const int tile = tyl * kBinTiles + tx;


    // Fragments.  Triangles are small: a wave's 64 masks hold 4 fragments on average but 28 at the maximum
    // (measured on the headline batch), so letting every lane pop its own bits keeps 63 lanes waiting for the
    // fullest one.  Masks above kCoop fragments are therefore rasterised by the WHOLE wave, one lane per pixel
    // (the mask is the execution mask: no bit scanning, one conflict-free LDS atomic per mask) -- or, when the batch holds
    // kOctMin of them or more, eight masks at a time by eight lanes each (below); the others are popped by their own
    // lanes, at most kCoop rounds.
    // (A threshold that follows the batch -- sixteen while few masks exceed eight fragments -- was measured in round 5: within 1 %
    // on every scene; a fixed 12..16 is 2-3 % faster on meshes of small triangles and 3-5 % slower on the stress scene.)
#ifndef NVDR_KCOOP
#define NVDR_KCOOP 4       // (round 6: 8 -> 4 with the eight-at-a-time path behind it: S10k -2.6 %, the other scenes within noise; tools/build_ab.sh sweeps it)
#endif
    constexpr int kCoop = NVDR_KCOOP;
    constexpr int kOctMin = 5;                               // big masks in a batch from which they are taken eight at a time
    const bool big = __popcll(m) > kCoop;
    uint64_t heavy = __ballot(big);
    const int kBig = heavy ? 2 : 0;
    if (__ballot(!big && m != 0) && !(DBG && (p.dbg & 8192))) {
        const uint32_t zxl = zx & 0xFFFFFFu, zxh = zx >> 24, zyl = zy & 0xFFFFFFu, zyh = zy >> 24;
        unsigned long long* keys = sh.key[tyl][tx];
        // (the lanes used to start at different bits, so that equal masks would not all hit one LDS address: masks of one batch
        // are neighbouring triangles' -- different pixels -- and the rotation cost more than the conflicts it avoided, r05n)
        uint64_t ml = big ? 0ull : m;
        while (__ballot(ml != 0)) {
            if (ml != 0) {
                int b = __builtin_ctzll(ml);
                ml &= ml - 1;
                uint32_t x = (uint32_t)(b & 7), y = (uint32_t)(b >> 3);
                // zx*x + zy*y with x,y < 8 via 24-bit multiplies (FineRaster.inl:348 depth, U32 wrap).
                uint32_t depth = d0 + __umul24(zxl, x) + (__umul24(zxh, x) << 24) + __umul24(zyl, y) + (__umul24(zyh, y) << 24);
                bool live = true;
                if (PEEL) {
                    uint32_t pz = p.peel[((size_t)n * p.Hp + (Y0 + (int)y + p.vp.offy)) * p.Wp + (X0 + (int)x + p.vp.offx)];
                    live = depth > pz;                                               // FineRaster.inl:349
                }
                if (DBG && (p.dbg & 2048)) { if (depth == 0x12345u) keys[b] = depth; }            // experiment: the loop without its atomics
                else if (live) atomicMin(&keys[b], ((unsigned long long)depth << 32) | idk);      // (b = y * 8 + x)
            }
        }
    }
    const uint32_t m0lo = (uint32_t)m, m0hi = (uint32_t)(m >> 32);
    if (heavy && !(DBG && (p.dbg & 4096))) {
        const uint32_t xl = (uint32_t)(lane & 7), yl = (uint32_t)(lane >> 3);                 // this lane's pixel: bit `lane` of a mask
        unsigned long long* keys0 = &sh.key[0][0][0];
        // A batch with many big masks (the stress scenes: triangles of tens of pixels, several deep) takes them EIGHT at a time:
        // eight lanes per mask, lane k of a group walking column k of that mask's tile downwards.  The operands of a group come
        // from seven ds_bpermute (per eight masks) instead of seven v_readlane per mask, and a step is an add, a bit test and the
        // LDS minimum (the key's address is the column's plus a constant): ~12 instead of ~20 vector instructions per mask.  The
        // minima of a step touch eight tiles, 512 B apart: two lanes per bank pair per pass, hidden behind the vector work
        // (NOTES 9.9).  Not while peeling (a step would need the previous layer's depth from memory).
        if (!PEEL) {
            while (__popcll(heavy) >= kOctMin) {
                int srcLane = -1;                                                             // this group's mask is that lane's
#pragma unroll
                for (int g = 0; g < 8; g++) {
                    const int sl = heavy ? (int)__builtin_ctzll(heavy) : -1;
                    heavy &= heavy - 1;                                                       // (0 stays 0)
                    if (__builtin_amdgcn_inverse_ballot_w64(0xFFull << (8 * g))) srcLane = sl;
                }
                const int a4 = srcLane << 2;
                uint32_t smlo = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)m0lo), smhi = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)m0hi);
                const uint32_t szx = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)zx), szy = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)zy);
                const uint32_t sd0 = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)d0), sidk = (uint32_t)__builtin_amdgcn_ds_bpermute(a4, (int)idk);
                const int stile = __builtin_amdgcn_ds_bpermute(a4, tile);
                if (srcLane < 0) { smlo = 0u; smhi = 0u; }                                    // fewer than eight left: idle groups
                if (ezOn && xl == 0u && (smlo & smhi) == 0xFFFFFFFFu) {                       // full tile: its bound at once (as below)
                    const uint32_t cmax = sd0 + 7u * (uint32_t)max((int)szx, 0) + 7u * (uint32_t)max((int)szy, 0);
                    atomicMin(&sh.tileZ[stile], (cmax >> 16) + 1u);
                }
                const uint32_t wlo = smlo >> xl, whi = smhi >> xl;                            // column xl: bit 8 * (y & 3) of the half that holds row y
                uint32_t depth = sd0 + szx * xl;                                              // (U32 wrap, as the 24-bit pieces give it)
                unsigned long long* kp = keys0 + stile * 64 + (int)xl;
#pragma unroll
                for (int y = 0; y < 8; y++) {
                    if ((y < 4 ? wlo : whi) & (1u << (8 * (y & 3)))) atomicMin(&kp[y * 8], ((unsigned long long)depth << 32) | sidk);
                    depth += szy;
                }
            }
        }
        if (heavy) do {
            const int src = __builtin_ctzll(heavy);
            heavy &= heavy - 1;
            const uint64_t sm = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)m0hi, src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)m0lo, src);
            const uint32_t szx = (uint32_t)__builtin_amdgcn_readlane((int)zx, src), szy = (uint32_t)__builtin_amdgcn_readlane((int)zy, src);
            const uint32_t sd0 = (uint32_t)__builtin_amdgcn_readlane((int)d0, src), sidk = (uint32_t)__builtin_amdgcn_readlane((int)idk, src);
            const int stile = __builtin_amdgcn_readlane(tile, src);
            if (!PEEL && ezOn && sm == ~0ull) {
                // The triangle covers all 64 pixels of the tile: every pixel's key ends up at most at this triangle's depth there,
                // so the plane's largest corner value bounds the tile at once (the reference's updateTileZMax, FineRaster.inl:19-34,
                // without its scan; not while peeling, where a fragment may be rejected by the previous layer).  All four corners
                // are covered pixels, so their depths are true plane values (no wrap) and 7 |slope| < 2^32 makes the signed reading
                // of the slopes the right one.  Scalar arithmetic; one lane lowers the bound.
                const uint32_t cmax = sd0 + 7u * (uint32_t)max((int)szx, 0) + 7u * (uint32_t)max((int)szy, 0);
                if (lane == 0) atomicMin(&sh.tileZ[stile], (cmax >> 16) + 1u);
            }
            if (__builtin_amdgcn_inverse_ballot_w64(sm)) {                 // the mask IS the execution mask
                const uint32_t depth = sd0 + __umul24(szx & 0xFFFFFFu, xl) + (__umul24(szx >> 24, xl) << 24)
                                           + __umul24(szy & 0xFFFFFFu, yl) + (__umul24(szy >> 24, yl) << 24);
                bool live = true;
                if (PEEL) {
                    const int sX0 = __builtin_amdgcn_readlane(X0, src), sY0 = __builtin_amdgcn_readlane(Y0, src);
                    const uint32_t pz = p.peel[((size_t)n * p.Hp + (sY0 + (int)yl + p.vp.offy)) * p.Wp + (sX0 + (int)xl + p.vp.offx)];
                    live = depth > pz;
                }
                if (live) atomicMin(&keys0[stile * 64 + lane], ((unsigned long long)depth << 32) | sidk);      // (lane = yl * 8 + xl)
            }
        } while (heavy);
    }
    return kHot | kBig;
}

// One tile's depth bound for k_fine's per-tile depth cull, refreshed from the tile's keys: the largest depth any of its 64 pixels
// holds NOW is an upper bound of what each will hold in the end (keys only go down), and an untouched pixel holds the far plane, so
// the bound bites once the tile is covered -- by one triangle or by several together (the reference keeps the same maximum per
// tile, FineRaster.inl:13-34).  Waves take turns: after every batch of 64 pairs a wave refreshes one tile -- kEarlyZTiles of them once
// covered tiles keep receiving pairs (raster_pairs).  Racing plain stores are fine: every value ever stored is a valid bound.
__device__ __forceinline__ void refresh_tile_bound(FineShared& sh, int tile)
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    uint32_t d = ((const uint32_t*)&sh.key[0][0][0])[(tile * 64 + l) * 2 + 1];      // the keys' upper halves are the depths
    d = max(d, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d, 0xb1, 0xf, 0xf, false));      // quad_perm [1,0,3,2]
    d = max(d, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d, 0x4e, 0xf, 0xf, false));      // quad_perm [2,3,0,1]
    d = max(d, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d, 0x114, 0xf, 0xf, false));     // row_shr:4
    d = max(d, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d, 0x118, 0xf, 0xf, false));     // row_shr:8
    d = max(d, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d, 0x142, 0xa, 0xf, false));     // row_bcast:15 -> rows 1, 3
    d = max(d, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d, 0x143, 0xc, 0xf, false));     // row_bcast:31 -> rows 2, 3
    if (l == 63) atomicMin(&sh.tileZ[tile], (d >> 16) + 1u);                                  // (upper half, rounded up; a minimum: a full-tile bound may be lower already)
}

// DBG = development instrumentation (per-workgroup phase timestamps and experiment switches); the
// production instantiation carries none of it (the kernel sits at the 64-VGPR limit of 4 workgroups/CU).
// LIST = the launch may contain bins with a triangle list (large meshes); a separate instantiation, so that the others keep their
// register budget.
// A shared bin of a LIST launch has ONE key array in memory: its parts merge their keys into it with (non-returning) 64-bit
// atomic minima and leave; a second launch of this kernel, SHADE, one workgroup per shared bin, reads the array, leaves it
// as it found it (all ones) and shades.  The kernel boundary is the hand-off -- no arrival counter, no fence, no waiting for
// exchanges, which is what made many small parts expensive in the one-launch scheme of the small launches (k_fine over a
// million-triangle mesh, parts of 1024 triangles: 2.0 ms with the exchange).
template <bool PEEL, bool WRITE_DEPTH, bool DBG, bool SPLIT, bool LIST = false, bool SHADE = false>
__global__ __launch_bounds__(kFineThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_fine(const FineParams p)
{
    __shared__ FineShared sh;
    // The thread id is used HERE and nowhere below: the wave number lives in a scalar register, lane numbers come from v_mbcnt
    // wherever they are needed, and "thread 0" is lane 0 of wave 0 -- kept as a VGPR across the raster stage the packed id register
    // was parked in scratch (r04: 12 B/lane; tests/test_kernel_resources.py).
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // Work assignment.  Blocks b, b+8, b+16, ... run on one XCD (observed b % 8 placement)
    // and walk that XCD's chunk of the heavy-first order produced by k_order, so an image's
    // records / AABBs / vertices stay in one L2 and long bins start early.  Placement and
    // order only affect speed.
    // SPLIT (the launch has too few bins to fill the chip, so its length is the lifetime of the bin with the most
    // triangles): the first kHelpersPerChunk block numbers of every XCD are helper slots -- extra workgroups for the
    // heaviest bins, each taking a part of the bin's slot range (they start together with those bins).
    const int perXcd = (p.totalBins + 7) >> 3;
    constexpr int kHelpers = LIST ? kListHelpersPerChunk : kHelpersPerChunk;
    const int xcd = (int)(blockIdx.x & 7), jj = (int)(blockIdx.x >> 3) - (SPLIT ? kHelpers : 0);
    int4 it4;
    int item = -1, part = 0, parts = 1, split = 0;
    if (SHADE) {
        split = (int)blockIdx.x;
        const int b = p.splitBin[split];
        if (b < 0) return;
        it4 = make_int4(b, 0, 0, 0); parts = 2;
    } else if (SPLIT && jj < 0) {
        it4 = p.helpers[xcd * kHelpers + jj + kHelpers];
        if (it4.x < 0) return;
        split = it4.y >> (2 * kPartBits); part = (it4.y >> kPartBits) & kPartMask; parts = it4.y & kPartMask;
    } else {
        item = xcd * perXcd + jj;
        if (jj >= perXcd || (!SPLIT && item >= p.totalBins)) return;
        it4 = p.order[item];
        if (it4.x < 0) return;                               // (unused slot at the end of a chunk, k_order)
        if (SPLIT) {
            const int info = p.splitInfo[item];
            if (info) { split = info >> (2 * kPartBits); parts = info & kPartMask; }
        }
    }
    const int binsPerImage = p.binsX * p.binsY;
    // Zeros for every pixel of an empty bin (what the shader stores for a pixel without a triangle).  The parameters are
    // read afresh from the kernarg segment through a pointer the compiler cannot see through: hoisted to the kernel's
    // start they would occupy scalar registers across the raster stage, which has none to spare.
    auto clear_bin = [](int packed, int wave) {                // packed = image << 10 | bin row << 5 | bin column (k_order)
        const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        const FineParams& q = *(const FineParams*)ka;
        int lane;                                              // taken afresh (opaque): nothing of this stays live across the raster stage
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
        const int n2 = packed >> 10, by2 = (packed >> 5) & 31, bx2 = packed & 31;
        // a wave's eight pixel rows of the bin, ONE ROW PER STORE: 64 lanes x 16 B = 1 KiB contiguous (tile by tile a store was
        // eight 128-byte pieces 8 KB apart)
        const int X = bx2 * kBinTiles * 8 + lane;
        const int Y0 = (by2 * kBinTiles + wave / kWavesPerRow) * 8;
        if (Y0 >= q.vp.vph) return;
        if (X < q.vp.vpw) {
#pragma unroll 1
            for (int r = 0; r < 8; r++) {
                if (Y0 + r >= q.vp.vph) break;
                const size_t pidx = ((size_t)n2 * q.H + (Y0 + r + q.vp.offy)) * q.W + (X + q.vp.offx);
                ((float4*)q.out)[pidx] = make_float4(0.f, 0.f, 0.f, 0.f);
                store_streaming((float4*)q.out_db + pidx, make_float4(0.f, 0.f, 0.f, 0.f));
            }
        }
        if (q.tileFlags && lane < kBinTiles && (bx2 * kBinTiles + lane) * 8 < q.vp.vpw)        // lane t: the row's tile t
            q.tileFlags[((size_t)n2 * q.tfH + ((Y0 + q.vp.offy) >> 3)) * q.tfW + (((bx2 * kBinTiles + lane) * 8 + q.vp.offx) >> 3)] = 0;
        if (q.rowCov && lane == 0) q.rowCov[((size_t)(n2 * q.binsY + by2) * q.binsX + bx2) * kBinTiles + wave / kWavesPerRow] = 0;
    };
    // Empty bins are pure stores, and in the heavy-first order they all come last: 300 MB of zeros at the headline batch
    // that nothing overlaps (the launch was store floor + the bins' compute).  In the plain instantiation each
    // bin WITH triangles therefore also clears one empty bin of its chunk after shading its own -- the stores go out while
    // other workgroups of the CU rasterise -- and the empty bin's own workgroup leaves at once.
    constexpr bool kPlain = !PEEL && !WRITE_DEPTH && !DBG && !SPLIT && !LIST;
    const int work = it4.x;
    if (kPlain && it4.y == 0) {
        const int nz = p.chunkNz[xcd];
        if (jj - nz < nz) return;                                          // cleared by the chunk's (jj - nz)-th bin
        // an empty bin without a partner (more empty bins than others in the chunk): zeros straight away -- no key arrays, no
        // shader; its waves take their tile rows from a ticket like the partner-clearing epilogue does
        if (wave == 0 && lane_id() == 0) sh.ticket = 0;
        __syncthreads();
        int row = 0;
        if (lane_id() == 0) row = atomicAdd(&sh.ticket, 1);
        clear_bin(it4.z, __builtin_amdgcn_readfirstlane(row));
        return;
    }
    const int n   = work / binsPerImage;
    const int bin = work - n * binsPerImage;
    const int binY = bin / p.binsX, binX = bin - binY * p.binsX;
    const int btx0 = binX * kBinTiles, bty0 = binY * kBinTiles;
    const int binTris = SHADE ? 0 : (SPLIT && parts > 1) ? 0x7FFFFFFF : it4.y;   // triangles whose AABB touches this bin (a part does not know its share: it scans its whole range)

    const int lane = lane_id();
    const bool first = wave == 0 && lane == 0;
    const int vpwPad = (p.vp.vpw + 7) & ~7, vphPad = (p.vp.vph + 7) & ~7;

    unsigned long long tstamp[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long dbgSurv = 0;
    if (DBG && p.dbgbuf) tstamp[0] = wall_clock64();
    const unsigned long long kInit = ((unsigned long long)kDepthMax << 32) | 0xFFFFFFFFull;
#pragma unroll
    for (int t = 0; t < kTilesPerWave; t++) sh.key[wave / kWavesPerRow][(wave % kWavesPerRow) * kTilesPerWave + t][lane] = kInit;
    if (first) { sh.count = 0; sh.slot[0] = 0; sh.ticket = 0; sh.pending = 0; }
    if (wave == 0) sh.tileZ[lane] = 0xFFFFFFFFu;
    __syncthreads();

    // DBG experiment: bins at or above a triangle threshold (NVDR_DEBUG bits 20..31, x16) skip their raster stage
    if (binTris > 0 && !(DBG && (p.dbg >> 20) && binTris >= ((p.dbg >> 20) & 0xFFF) * 16)) {
        const int direct = p.instance ? p.T : p.ranges[2 * n + 1];
        const int pool   = min(p.poolFinal[n], p.slots - p.poolBase);
        // Index space scanned by the filter: [0, dlen) = the bin's range of direct slots (k_setup
        // recorded the smallest and largest slot that touches the bin; meshes are spatially coherent
        // in index order, so this is a small part of the image's triangles), [dlen, dlen + pool) =
        // the clipper's pool slots.  Four consecutive slots per lane per step.
        // A bin with a triangle list (large meshes, k_binscan / k_binfill): it4.z = where the list starts, it4.w = minus its
        // length; the "index space" is then the list itself and nothing is filtered.
        const bool listMode = LIST && it4.w < 0;
        const int scanLo = it4.z, dlen = it4.w;
        int total = listMode ? -it4.w : dlen + pool, scanBeg = 0;
        if (SPLIT && parts > 1) {                   // this workgroup's part of the index space (multiples of 4 slots)
            const int per = ((total + parts - 1) / parts + 3) & ~3;
            scanBeg = min(part * per, total);
            total = min(scanBeg + per, total);
        }
        const uint32_t* gbox = p.bbox + (size_t)n * p.slots;
        const uint4*    grec = p.rec + (size_t)n * p.slots * 4;

        constexpr int kStep = kFineThreads * 4;     // slots per workgroup step
        int scan = scanBeg + (listMode ? 0 : wave * 256);   // this wave's position in the scanned index space (list: the workgroup's)
        int sub = 0, skip = 0;                      // resume point inside the current group of 4x64 slots
        bool done = (scan >= total);
        int found = 0;                              // list entries consumed by earlier rounds

        auto load_boxes = [&](int pos) -> uint4 {
            int idx = pos + lane * 4;
            uint4 b = make_uint4(kEmptyBox, kEmptyBox, kEmptyBox, kEmptyBox);
            if (idx < total) {
                if (idx < dlen) {                   // direct slots; mask the padding beyond `direct`
                    const int slot = scanLo + idx;
                    b = *(const uint4*)(gbox + slot);
                    if (slot + 1 >= direct) b.y = kEmptyBox;
                    if (slot + 2 >= direct) b.z = kEmptyBox;
                    if (slot + 3 >= direct) b.w = kEmptyBox;
                    if (slot >= direct) b.x = kEmptyBox;
                } else {
                    b = *(const uint4*)(gbox + p.poolBase + (idx - dlen));
                    int rem = total - idx;
                    if (rem < 2) b.y = kEmptyBox;
                    if (rem < 3) b.z = kEmptyBox;
                    if (rem < 4) b.w = kEmptyBox;
                }
            }
            return b;
        };

        for (;;) {
            // (re)loaded at the start of every pass instead of being carried over the raster stage: four registers
            // less there, where the kernel sits at its 64-VGPR budget
            uint4 cur = (done || listMode) ? make_uint4(kEmptyBox, kEmptyBox, kEmptyBox, kEmptyBox) : load_boxes(scan);
            if (listMode && !done) {
                // ---- the next kListCap entries of the bin's list, with their AABBs ----------
                const int take = min(kListCap, total - scan);
                const int tid = wave * 64 + lane;
                if (tid < take) {
                    const uint32_t slot = p.binList[(size_t)(uint32_t)scanLo + scan + tid];
                    sh.slot[tid] = slot;
                    sh.box[tid] = gbox[slot];
                }
                if (first) sh.count = take;
                scan += take;
                done = (scan >= total);
            }
            // ---- filter: compact this bin's triangles into the LDS list ----------------
            while (!done && !listMode) {
                if (found + sh.count >= binTris) { done = true; break; }       // every triangle of the bin is listed
                const int nextScan = scan + kStep;
                uint4 nxt = make_uint4(kEmptyBox, kEmptyBox, kEmptyBox, kEmptyBox);
                if (nextScan < total) nxt = load_boxes(nextScan);      // prefetch the next group while this one is filtered
                bool full = false;
                for (; sub < 4; sub++) {
                    uint32_t box = sub == 0 ? cur.x : sub == 1 ? cur.y : sub == 2 ? cur.z : cur.w;
                    int txlo = box & 255, tylo = (box >> 8) & 255, txhi = (box >> 16) & 255, tyhi = box >> 24;
                    bool hit = (txlo <= btx0 + kBinTiles - 1) & (txhi >= btx0) & (tylo <= bty0 + kBinTiles - 1) & (tyhi >= bty0);
                    uint64_t m = __ballot(hit);
                    int nh = __popcll(m) - skip;
                    if (nh > 0) {
                        int base = 0;
                        if (lane == 0) base = atomicAdd(&sh.count, nh);
                        base = __builtin_amdgcn_readfirstlane(base);
                        int can = min(max(kListCap - base, 0), nh);
                        int rank = mask_rank(m) - skip;
                        if (hit && rank >= 0 && rank < can) {
                            int dst = base + rank;
                            int idx = scan + lane * 4 + sub;
                            int slot = (idx < dlen) ? scanLo + idx : p.poolBase + (idx - dlen);
                            sh.slot[dst] = (uint32_t)slot;
                            sh.box[dst] = box;
                        }
                        if (can < nh) { skip += can; full = true; break; }      // list full: resume here after the flush
                    }
                    skip = 0;
                }
                if (full) break;
                sub = 0;
                scan = nextScan;
                done = (scan >= total);
                cur = nxt;
            }
            unsigned long long tf0 = (DBG && p.dbgbuf) ? wall_clock64() : 0;
            // The bin is finished when every wave has scanned to the end or all its triangles are listed: a wave that is not at
            // its end says so (one LDS word, reset between the passes) -- __syncthreads_and() is a second barrier and computes the
            // flat thread id from all three id components, which kept the packed id register alive across the kernel.
            if (!done && lane == 0) sh.pending = 1;
            __syncthreads();
            const int cnt = min(sh.count, kListCap);
            found += cnt;
            const int allDone = __builtin_amdgcn_readfirstlane((found >= binTris || sh.pending == 0) ? 1 : 0);   // uniform: keep it in an SGPR
            unsigned long long tf1 = (DBG && p.dbgbuf) ? wall_clock64() : 0;
            if (DBG) { tstamp[1] += 1; tstamp[4] += cnt; }

            // ---- raster: the list's (triangle, tile) pairs are numbered through a prefix sum of the
            //      entries' pair counts; waves take 64 consecutive pair numbers at a time (lane = pair),
            //      so work is balanced over all waves whatever the triangles' shapes, and rasterise
            //      them into the shared key arrays (LDS atomics) ---------------------------------------
            auto pair_box = [&](uint32_t box, int& x0, int& y0, int& nx, int& ny) {
                const int txlo = box & 255, tylo = (box >> 8) & 255, txhi = (box >> 16) & 255, tyhi = box >> 24;
                x0 = max(txlo, btx0); y0 = max(tylo, bty0);
                const int x1 = min(txhi, btx0 + kBinTiles - 1), y1 = min(tyhi, bty0 + kBinTiles - 1);
                nx = max(x1 - x0 + 1, 0); ny = max(y1 - y0 + 1, 0);
                if (nx == 0) ny = 0;
            };
            if (wave == 0) {
                // exclusive scan of the pair counts: lane l owns entries [l*7, l*7+7)
                constexpr int kPer = (kListCap + 63) / 64;
                int lane;                                   // taken afresh: keeps lane * kPer out of the long-lived registers
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
                int local[kPer], sum = 0;
#pragma unroll
                for (int i = 0; i < kPer; i++) {
                    const int j = lane * kPer + i;
                    int x0, y0, nx = 0, ny = 0;
                    if (j < cnt) pair_box(sh.box[j], x0, y0, nx, ny);
                    local[i] = sum;
                    sum += nx * ny;
                }
                const int incl = wave_scan_incl(sum);
                const int base = incl - sum;
#pragma unroll
                for (int i = 0; i < kPer; i++) { const int j = lane * kPer + i; if (j <= cnt) sh.pfx[j] = (uint16_t)(base + local[i]); }
                // the entry that holds every 64th pair: entry j covers pairs [start, end); the first multiple of 64 at or after its
                // start is inside it at most once (an entry has at most 64 pairs)
#pragma unroll
                for (int i = 0; i < kPer; i++) {
                    const int j = lane * kPer + i;
                    const int start = base + local[i], end = (i + 1 < kPer) ? base + local[(i + 1) % kPer] : incl;
                    const int k64 = (start + 63) >> 6;
                    if (j < cnt && (k64 << 6) < end) sh.bstart[k64] = (uint16_t)j;
                }
                if (lane == 63) sh.totalPairs = incl;
            }
            __syncthreads();
            if (!(DBG && (p.dbg & 4))) {
                const int total = sh.totalPairs;
                // (NVDR_DEBUG 32, debug instantiation: the kernel without its depth cull, for comparison.  Not in the list
                // instantiation: a mesh of 32 k+ triangles is a mesh of small triangles -- little fragment work to save -- and the
                // cull's registers cost that instantiation 8 % at a million triangles, 0.245 -> 0.263 ms)
                const bool ezOn = !LIST && !(DBG && (p.dbg & 32));
                int turn = wave;                            // the tile whose depth bound this wave refreshes next
                int cold = 0;                               // batches since this wave last refreshed a bound with nothing to cull
                for (int q0 = wave * 64; q0 < total; q0 += kFineWaves * 64) {
                    const int q = q0 + lane;
                    const bool act = q < total;
                    int j;
                    {
                        // entry s holds pair q0 (wave 0's table); the entries s + 1 ... that START inside this batch each set one
                        // bit of a wave-wide mask (entries have at least one pair each: starts are distinct), and a pair's entry is s
                        // + the number of starts at or below it
                        const int s = __builtin_amdgcn_readfirstlane((int)sh.bstart[q0 >> 6]);
                        const int el = s + lane;
                        const int r = (lane >= 1 && el <= cnt) ? (int)sh.pfx[el] - q0 : 64;        // (pfx[cnt] = total: no start there)
                        const bool st = r < 64 && el < cnt;
                        uint32_t blo = (st && r < 32) ? 1u << r : 0u, bhi = (st && r >= 32) ? 1u << (r - 32) : 0u;
                        blo = wave_or_to_last(blo); bhi = wave_or_to_last(bhi);
                        const uint32_t mlo = (uint32_t)__builtin_amdgcn_readlane((int)blo, 63), mhi = (uint32_t)__builtin_amdgcn_readlane((int)bhi, 63);
                        // starts at positions <= lane: those strictly below (mbcnt) + this lane's own
                        const uint64_t M = ((uint64_t)mhi << 32) | mlo;
                        j = s + mask_rank(M) + (int)((M >> lane) & 1ull);
                        if (!act) j = 0;
                    }
                    int x0, y0, nx, ny;
                    pair_box(sh.box[j], x0, y0, nx, ny);
                    const int k = act ? q - (int)sh.pfx[j] : 0;
                    const int ky = (nx > 1) ? (int)(((float)k + 0.5f) / (float)nx) : k;   // exact for k < 64, nx <= 8
                    const int kx = k - ky * nx;
                    const int tx = x0 + kx - btx0, tyl = y0 + ky - bty0;
                    const int batch = raster_pairs<PEEL, DBG>(sh, p, grec, lane, n, act, (uint32_t)j | ((uint32_t)tx << 9) | ((uint32_t)tyl << 12),
                                                              btx0, bty0, DBG ? &dbgSurv : nullptr, ezOn);
                    // (kEarlyZTiles tiles per batch once covered tiles see more pairs; one after a batch with a large mask -- large
                    // triangles are what covers tiles; otherwise one every kColdRefresh batches: a mesh of small triangles without
                    // overdraw pays next to nothing -- refreshing after every batch was 9 us of the headline's 145, r05a)
                    if (ezOn) {
                        int nref = (batch & 1) ? kEarlyZTiles : (batch & 2) ? 1 : 0;
                        if (nref == 0 && ++cold >= kColdRefresh) { cold = 0; nref = 1; }
#pragma unroll 1
                        for (int rt = 0; rt < nref; rt++) {
                            refresh_tile_bound(sh, turn);
                            turn = (turn + kFineWaves) & (kBinTiles * kBinTiles - 1);
                        }
                    }
                }
            }
            if (DBG && p.dbgbuf) { unsigned long long tr = wall_clock64(); if (p.dbg & 64) { tstamp[2] += (unsigned long long)sh.totalPairs; tstamp[3] += dbgSurv; dbgSurv = 0; } else { tstamp[2] += tf1 - tf0; tstamp[3] += tr - tf1; } }
            __syncthreads();
            if (allDone) break;
            if (first) {
                // (the zero is made in place, opaque to the compiler: as a constant it was hoisted out of the pass loop as a 64-bit
                // pair for the two neighbouring words -- and, there being no registers for it, parked in scratch)
                int z; asm volatile("v_mov_b32 %0, 0" : "=v"(z));
                sh.count = z; sh.slot[0] = (uint32_t)z; sh.pending = z;
            }
            __syncthreads();
        }
    }

    if (DBG && p.dbgbuf) tstamp[5] = wall_clock64();
    // The lane number is taken afresh here (opaque to the compiler): derived from the copy made at kernel entry, the
    // shader's per-lane constants were computed up front and parked in scratch across the raster stage.
    int laneS;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(laneS));
    const int tileRow = wave / kWavesPerRow, tile0 = (wave % kWavesPerRow) * kTilesPerWave;
    // ---- a bin shared by several workgroups: every part publishes its key array in memory and counts itself in; the
    //      part that arrives last takes the minimum over all parts (the same order-free rule as in LDS) and shades the
    //      bin, the others are done.  Only RETURNING device-scope atomics touch the shared memory: they execute at the
    //      memory side, coherently for the whole device -- no dependence on where the parts run and no cache
    //      write-back (an agent-scope fence in a kernel with hundreds of MB of stores in flight costs more than the
    //      split saves: 126 -> 220 us) -- and a part knows its keys are in place when the old values have come back,
    //      which is before it counts itself in.  Nothing needs initialising: every part writes all of its keys.
    unsigned long long* gkeys = p.splitKeys + (size_t)((SPLIT && parts > 1) ? (LIST ? split : p.splitKeyBase[split]) : 0) * 4096 + (size_t)(tileRow * kBinTiles + tile0) * 64 + laneS;
    if (LIST && SPLIT && parts > 1) {
        if (!SHADE) {
            // a part of a shared bin: its keys into the bin's array (untouched pixels have nothing to say), and done
#pragma unroll
            for (int tt = 0; tt < kTilesPerWave; tt++) {
                const unsigned long long key = sh.key[tileRow][tile0 + tt][laneS];
                if (key != kInit) (void)__hip_atomic_fetch_min(&gkeys[tt * 64], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        // the shading launch: the merged keys, and the array back to all ones for the next call
#pragma unroll
        for (int tt = 0; tt < kTilesPerWave; tt++) {
            // (the array rests at all ones; an untouched pixel takes the kernel's own sentinel, kInit, like the pixels of every
            // other bin: the depth surface of a peeling pass then holds kDepthMax there, not 0xFFFFFFFF -- ADVICE r4)
            sh.key[tileRow][tile0 + tt][laneS] = min(gkeys[tt * 64], kInit);           // (read below by pixel column -- ANOTHER lane of this
            //  same wave: LDS operations of one wave complete in order, so no barrier is needed while a wave shades only tiles it
            //  wrote itself; a change to kWavesPerRow / kTilesPerWave that breaks that needs a __syncthreads() here)
            gkeys[tt * 64] = ~0ull;
        }
    }
    if (!LIST && SPLIT && parts > 1) {
        unsigned long long seen = 0ull;
#pragma unroll
        for (int tt = 0; tt < kTilesPerWave; tt++)
            seen |= atomicExch(&gkeys[part * 4096 + tt * 64], sh.key[tileRow][tile0 + tt][laneS]);
        // EVERY wave waits here, before the barrier, until the old values of its exchanges have come back: a returned
        // value is the proof that the exchange has been performed at the memory side.  (Left to the compiler, the wait
        // sank below the barrier -- the values are only consumed after it -- so that seven of the eight waves could pass
        // the barrier, and thread 0 count the part in, with their keys still in flight: ADVICE r2.  The ISA of this
        // instantiation is checked by tests/test_kernel_resources.py.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int landed = __builtin_amdgcn_readfirstlane((int)__popcll(__ballot(seen == 1ull)));      // consumes every returned value (whatever it is)
        __syncthreads();
        // The arrival counter carries the hand-off in the memory model's terms as well: release (this part's keys are
        // visible before its count) and acquire (the last part sees the others' keys after reading the full count), at
        // agent scope, by ONE thread of the shared bins' workgroups only -- not the per-wave fences in every workgroup
        // that cost 126 -> 220 us in round 2 (measured in round 3 at batch 16: k_fine 58 us relaxed, 63 us release-only, 64 us as here).
        if (first) {
            const int add = 1 + (landed >> 8);
            const int before = __hip_atomic_fetch_add(&p.splitDone[split], add, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            sh.count = (before == parts - 1) ? 1 : 0;
        }
        __syncthreads();
        if (!sh.count) return;
    }
    // ---- pixel shader (rasterize.cu:15-114) + stores: wave w shades tile row w, ONE PIXEL ROW PER ITERATION ----------------
    // lane l = pixel column l of the bin: a store instruction then writes 64 x 16 B = 1 KiB of one image row in one piece.  Shaded tile
    // by tile (lane = pixel of an 8x8 tile) every store was eight 128-byte pieces, 8 KB apart -- same bytes, but the memory side
    // takes whole kilobytes faster: the empty bins' zeros written row by row took the launch from 138 to 129 us (r05i), and four
    // fifths of this kernel's stores are zeros of empty tiles.  The keys of a row sit in eight tiles' arrays (an 8-way bank
    // conflict on eight LDS reads per wave: noise next to the stores).
    static_assert(kWavesPerRow == 1 && kTilesPerWave == 8, "the shader walks a whole tile row per wave");
    const int ty = bty0 + tileRow;
    const int X = btx0 * 8 + laneS;             // viewport-local pixel column
    const float4* vb = (const float4*)p.pos + (p.instance ? (size_t)n * p.V : 0);
    const int kcol = (laneS >> 3) * 64 + (laneS & 7);            // this column in its tile's key array (+ 8 * pixel row)
    uint64_t rowAny = 0ull;                     // bit l: column l shows a triangle in some row of this tile row
#pragma unroll 1
    for (int r = 0; r < 8; r++) {
        const int Y = ty * 8 + r;               // viewport-local pixel row
        unsigned long long key = (&sh.key[tileRow][0][0])[kcol + r * 8];
        if (!LIST && SPLIT && parts > 1) {                                       // (the last part only)
            // the other parts' keys: agent-scope atomic LOADS -- they observe the parts' exchanges wherever those ran, like the
            // returning atomics this loop used before, but several of them are in flight at a time (a bin shared by 64 parts reads
            // 512 values per lane here: one dependent round trip each was the launch's critical path)
            const unsigned long long* gk = gkeys - laneS + kcol + r * 8;           // (gkeys: the tile row's first array, + lane)
#pragma unroll 4
            for (int q = 0; q < parts; q++) {
                const unsigned long long other = __hip_atomic_load(&gk[(q == part ? part : q) * 4096], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                key = min(key, q == part ? key : other);
            }
        }
        if (WRITE_DEPTH) {
            if (X < vpwPad && Y < vphPad)
                p.depth[((size_t)n * p.Hp + (Y + p.vp.offy)) * p.Wp + (X + p.vp.offx)] = (uint32_t)(key >> 32);
        }
        const bool inImage = (X < p.vp.vpw) & (Y < p.vp.vph);
        rowAny |= __ballot(inImage & ((uint32_t)key != 0xFFFFFFFFu));
        if (!inImage) continue;
        const int px = X + p.vp.offx, py = Y + p.vp.offy;
        const size_t pidx = ((size_t)n * p.H + py) * p.W + px;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f), odb = make_float4(0.f, 0.f, 0.f, 0.f);
        const int triIdx = (DBG && (p.dbg & 8)) ? -1 : (int)(~(uint32_t)key) - 1;
        bool write = !(DBG && (p.dbg & 16));
        if (triIdx >= 0 && triIdx < p.T) {
            int vi0 = p.tri[triIdx * 3 + 0], vi1 = p.tri[triIdx * 3 + 1], vi2 = p.tri[triIdx * 3 + 2];
            if (!indices_ok(vi0, vi1, vi2, p.V)) {
                write = false;                       // reference leaves the pixel untouched (:43-47)
            } else {
                // Perspective-correct barycentrics from the edge functions of the pixel-relative
                // vertices (rasterize.cu:63-113), in the same cyclic form as the backward tape
                // (raster_pixel_grad): a_k = X_i Y_j - Y_i X_j, i = k+1, j = k+2.
                // Every multiply-add is spelled out (the sites nvcc's default -fmad contracts in the reference's expressions,
                // as in the oracle): left to the compiler, two instantiations of this kernel may fuse different products, and
                // on sub-pixel triangles -- a_k is a difference of nearly equal products -- that moves u, v by 2e-5.
#pragma clang fp contract(off)
                const float4 P[3] = {vb[vi0], vb[vi1], vb[vi2]};
                const float fx = __fmaf_rn(p.xs, (float)px, p.xo);
                const float fy = __fmaf_rn(p.ys, (float)py, p.yo);
                float Xr[3], Yr[3], a[3], DX[3], DY[3];
#pragma unroll
                for (int k = 0; k < 3; k++) { Xr[k] = __fmaf_rn(-fx, P[k].w, P[k].x); Yr[k] = __fmaf_rn(-fy, P[k].w, P[k].y); }
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int i = (k + 1) % 3, j = (k + 2) % 3;
                    a[k] = __fmaf_rn(Xr[i], Yr[j], -(Yr[i] * Xr[j]));
                    DX[k] = __fmaf_rn(P[j].y, P[i].w, -(P[i].y * P[j].w));          // d a_k / d fx
                    DY[k] = __fmaf_rn(P[i].x, P[j].w, -(P[j].x * P[i].w));          // d a_k / d fy
                }
                // v_rcp_f32 (1 ulp) instead of IEEE division: three divisions per pixel are a sixth of
                // this kernel's instruction count, and these outputs carry a 1e-5 tolerance.
                const float iw = __builtin_amdgcn_rcpf(a[0] + a[1] + a[2]);
                float b0 = __saturatef(a[0] * iw), b1 = __saturatef(a[1] * iw);
                const float z = __fmaf_rn(P[2].z, a[2], __fmaf_rn(P[1].z, a[1], P[0].z * a[0]));
                const float w = __fmaf_rn(P[2].w, a[2], __fmaf_rn(P[1].w, a[1], P[0].w * a[0]));
                const float zw = fmaxf(fminf(z * __builtin_amdgcn_rcpf(w), 1.f), -1.f);
                const float bs = __builtin_amdgcn_rcpf(fmaxf(b0 + b1, 1.f));     // renormalise after the clamp
                b0 *= bs; b1 *= bs;
                o = make_float4(b0, b1, zw, triidx_to_float(triIdx + 1));

                const float sx = p.xs * iw, sy = p.ys * iw;
                const float DtX = DX[0] + DX[1] + DX[2], DtY = DY[0] + DY[1] + DY[2];
                odb = make_float4(sx * __fmaf_rn(b0, DtX, -DX[0]), sy * __fmaf_rn(b0, DtY, -DY[0]),
                                  sx * __fmaf_rn(b1, DtX, -DX[1]), sy * __fmaf_rn(b1, DtY, -DY[1]));
            }
        }
        if (write) {
            ((float4*)p.out)[pidx] = o;
            // rast_db is write-once here and, in most op graphs, read late or never: a non-temporal store
            // keeps it from displacing `rast` (re-read by the next three kernels) in the Infinity Cache.
            store_streaming((float4*)p.out_db + pidx, odb);
        }
    }
    if (p.tileFlags && laneS < kBinTiles) {
        // tile occupancy for the consumers of rast (TileFlags): lane t stores the flag of the row's tile t -- does any of its pixels,
        // inside the viewport, show a triangle?
        const int Xt = (btx0 + laneS) * 8, Yt = ty * 8;
        if (Xt < p.vp.vpw && Yt < p.vp.vph)
            p.tileFlags[((size_t)n * p.tfH + ((Yt + p.vp.offy) >> 3)) * p.tfW + ((Xt + p.vp.offx) >> 3)] = ((rowAny >> (laneS * 8)) & 0xFFull) ? 1 : 0;
    }
    if (p.rowCov && laneS == 0) p.rowCov[(size_t)work * kBinTiles + tileRow] = rowAny ? 1 : 0;
    if (kPlain) {
        // (everything recomputed from the block index here: carrying the partner across the kernel cost spilled registers)
        const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        const FineParams& q = *(const FineParams*)ka;
        const int perXcd2 = (q.totalBins + 7) >> 3;
        const int xcd2 = (int)(blockIdx.x & 7), jj2 = (int)(blockIdx.x >> 3);
        const int nz = q.chunkNz[xcd2], ne = min(perXcd2, q.totalBins - xcd2 * perXcd2) - nz;
        if (jj2 < ne) {
            // which tile row a wave clears is immaterial: rows are handed out as the waves arrive (the wave number itself is
            // not kept in a register until here, and recomputing it would keep the thread id alive across the kernel)
            int row = 0;
            if (laneS == 0) row = atomicAdd(&sh.ticket, 1);
            clear_bin(__builtin_amdgcn_readfirstlane(q.order[xcd2 * perXcd2 + nz + jj2].z), __builtin_amdgcn_readfirstlane(row));
        }
    }
    if (DBG && p.dbgbuf && lane == 0 && item >= 0) {
        unsigned long long* d = p.dbgbuf + ((size_t)item * kFineWaves + wave) * 8;
        d[0] = tstamp[0]; d[1] = tstamp[1]; d[2] = tstamp[2]; d[3] = tstamp[3]; d[4] = tstamp[4]; d[5] = tstamp[5]; d[6] = wall_clock64(); d[7] = (unsigned long long)work;
    }
}

// ---------------------------------------------------------------------------------
// Backward (rasterize.cu:119-277)
// ---------------------------------------------------------------------------------

struct GradParams {
    const float* pos; const int* tri; const float* out; const float* dy; const float* ddb;
    float* grad;
    TileFlags flags;
    int instance, N, V, T, W, H;
    float xs, xo, ys, yo;
    int dbg;
};

// Workgroup = 64 x 16 pixel block of one image, 4 waves, each wave owning four 64-pixel rows
// (1 KiB coalesced loads of rast / dy).  The per-pixel
// gradients stay in registers between the two phases:
//   A  compute them, publish the block's largest magnitude (fixes the fixed-point scale);
//   B  sum them over runs of equal triangle id inside the wave (RunScan), add the run totals to
//      the workgroup's LDS vertex table (nvdr_device.hpp) with ds_add_u64, then flush one global
//      atomic per (vertex, x|y|w) the block touched.
// Work order: block b runs on XCD b % 8 (observed placement, speed only); an image's blocks are
// given to one XCD so that its vertices / indices are fetched into one L2.
constexpr int kGradBlockW = 64;
constexpr int kGradBlockH = 16;
constexpr int kGradThreads = 256;
constexpr int kGradRowsPerWave = 4;
constexpr int kGradSlots = 512;

struct PixelGrad { int tri, vi0, vi1, vi2; float g[9]; };

template <bool ENABLE_DB, bool DB_ONLY>
__device__ __forceinline__ bool raster_pixel_grad(const GradParams& p, const float4* __restrict__ vb, int px, int py, int pz, PixelGrad& r)
{
    if (px >= p.W) return false;
    if (p.flags.empty(pz, py, px)) return false;                      // nothing visible in this 8x8 tile: rast is not read
    const size_t pidx = ((size_t)pz * p.H + py) * p.W + px;
    float4 ddb = make_float4(0.f, 0.f, 0.f, 0.f);
    int grad_all_ddb = 0;
    if (DB_ONLY) {
        // dy == NULL: only rast_db's gradient contributes (the caller holds dy's share already).  It is looked at FIRST: where it
        // is all zeros -- the tensor autograd materialises for an output nobody used -- rast is not read either.
        ddb = ((const float4*)p.ddb)[pidx];
        grad_all_ddb = __float_as_int(ddb.x) | __float_as_int(ddb.y) | __float_as_int(ddb.z) | __float_as_int(ddb.w);
        if ((((uint32_t)grad_all_ddb) << 1) == 0u) return false;
    }
    const int triIdx = float_to_triidx(p.out[pidx * 4 + 3]) - 1;
    if (triIdx < 0 || triIdx >= p.T) return false;
    // Upstream gradients are fetched for covered pixels only (background rows of dy never leave HBM);
    // these loads travel together with the index loads.
    const float2 dy = DB_ONLY ? make_float2(0.f, 0.f) : ((const float2*)p.dy)[pidx * 2];
    if (ENABLE_DB && !DB_ONLY) ddb = ((const float4*)p.ddb)[pidx];
    const int vi0 = p.tri[triIdx * 3 + 0], vi1 = p.tri[triIdx * 3 + 1], vi2 = p.tri[triIdx * 3 + 2];
    const int grad_all_dy = __float_as_int(dy.x) | __float_as_int(dy.y);
    if (ENABLE_DB && !DB_ONLY) grad_all_ddb = __float_as_int(ddb.x) | __float_as_int(ddb.y) | __float_as_int(ddb.z) | __float_as_int(ddb.w);
    if ((((uint32_t)(grad_all_dy | grad_all_ddb)) << 1) == 0u) return false;          // all +-0 (:143-148)
    if (vi0 < 0 || vi0 >= p.V || vi1 < 0 || vi1 >= p.V || vi2 < 0 || vi2 >= p.V) return false;   // (short-circuit form: measured faster here than indices_ok)
    r.tri = triIdx; r.vi0 = vi0; r.vi1 = vi1; r.vi2 = vi2;

    // Reverse-mode differentiation of the pixel shader: nvdr_raster_tape.hpp (shared with the fused backward kernel).
    const float4 P[3] = {vb[vi0], vb[vi1], vb[vi2]};
    const float fx = p.xs * (float)px + p.xo;
    const float fy = p.ys * (float)py + p.yo;
    raster_tape<ENABLE_DB>(P, fx, fy, p.xs, p.ys, dy.x, dy.y, ddb, ENABLE_DB && (((uint32_t)grad_all_ddb) << 1) != 0u, r.g);
    return true;
}

template <bool ENABLE_DB, bool DB_ONLY = false>
__global__ __launch_bounds__(kGradThreads, ENABLE_DB ? 4 : 6) void k_raster_grad(const GradParams p, int gx, int gy)
{
    __shared__ uint32_t s_keys[kGradSlots];
    __shared__ unsigned long long s_vals[kGradSlots * 3];
    __shared__ uint32_t s_max, s_used;
    __shared__ uint16_t s_list[kGradSlots];
    int bx, by, pz;
    if (p.flags.order ? !decode_block_ordered(p.flags, gx, gy, kGradBlockW, kGradBlockH, bx, by, pz)
                      : !decode_block(gx, gy, p.N, bx, by, pz)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = bx * kGradBlockW + lane;
    const int row0 = by * kGradBlockH + wave * kGradRowsPerWave;
    if (DB_ONLY) {
        // The usual caller of this variant holds the zeros autograd materialised for a rast_db nobody used (_plugin.py, "fused
        // backward"): look at the block's ddb first -- four loads in flight per lane, nothing else set up -- and leave if it is
        // all zeros.  (A block that stays reads its ddb again below, from L2.)
        uint32_t any = 0u;
#pragma unroll
        for (int r = 0; r < kGradRowsPerWave; r++) {
            const int py = row0 + r;
            if (px < p.W && py < p.H && !p.flags.empty(pz, py, px)) {
                const float4 d = ((const float4*)p.ddb)[((size_t)pz * p.H + py) * p.W + px];
                any |= __float_as_uint(d.x) | __float_as_uint(d.y) | __float_as_uint(d.z) | __float_as_uint(d.w);
            }
        }
        if (!__syncthreads_or((any << 1) != 0u)) return;
    }
    VertexTable tab{s_keys, s_vals, kGradSlots, 3};
    tab.clear(threadIdx.x, kGradThreads);
    if (threadIdx.x == 0) { s_max = 0u; s_used = 0u; }
    __syncthreads();

    const size_t voff = p.instance ? (size_t)pz * p.V : 0;
    const float4* vb = (const float4*)p.pos + voff;
    float* gout = p.grad + voff * 4;

    // Phase A: per-pixel gradients (kept in registers) and the block's largest magnitude.
    PixelGrad pg[kGradRowsPerWave];
    bool ok[kGradRowsPerWave];
    uint32_t um = 0u;                                       // (as magnitude bits: nvdr_device.hpp mag_bits)
#pragma unroll
    for (int r = 0; r < kGradRowsPerWave; r++) {
        const int py = row0 + r;
        ok[r] = (py < p.H) && raster_pixel_grad<ENABLE_DB, DB_ONLY>(p, vb, px, py, pz, pg[r]);
        if (ok[r]) {
#pragma unroll
            for (int k = 0; k < 9; k++) um = max(um, mag_bits(pg[r].g[k]));
        } else {
            pg[r].tri = -1;
#pragma unroll
            for (int k = 0; k < 9; k++) pg[r].g[k] = 0.f;       // the run scan multiplies masked lanes by 0: keep them finite
        }
    }
    block_max_update(&s_max, __int_as_float((int)um));
    __syncthreads();
    const uint32_t maxBits = s_max;
    if (maxBits == 0u || (p.dbg & 1)) return;                    // nothing to accumulate

    if (maxBits >= 0x7F800000u) {                                // inf/NaN present: plain f32 atomics keep the semantics
#pragma unroll
        for (int r = 0; r < kGradRowsPerWave; r++) {
            if (!ok[r]) continue;
            const int vi[3] = {pg[r].vi0, pg[r].vi1, pg[r].vi2};
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float* q = gout + (size_t)vi[k] * 4;
                atomic_add_f32(q + 0, pg[r].g[k * 3 + 0]); atomic_add_f32(q + 1, pg[r].g[k * 3 + 1]); atomic_add_f32(q + 3, pg[r].g[k * 3 + 2]);
            }
        }
        return;
    }
    const FixedScale fs(maxBits);

    // Phase B: run totals -> LDS table.
#pragma unroll
    for (int r = 0; r < kGradRowsPerWave; r++) {
        if (__ballot(ok[r]) == 0) continue;
        const RunScan rs(pg[r].tri, ok[r]);
#pragma unroll
        for (int k = 0; k < 9; k += 3) rs.scan3(pg[r].g[k], pg[r].g[k + 1], pg[r].g[k + 2]);
        if (rs.tail) {
            const int vi[3] = {pg[r].vi0, pg[r].vi1, pg[r].vi2};
            int sl[3];
            tab.find3(vi[0], vi[1], vi[2], sl[0], sl[1], sl[2]);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int s = sl[k];
                if (s >= 0) {
                    tab.add(s, 0, fs.to_fixed(pg[r].g[k * 3 + 0]));
                    tab.add(s, 1, fs.to_fixed(pg[r].g[k * 3 + 1]));
                    tab.add(s, 2, fs.to_fixed(pg[r].g[k * 3 + 2]));
                } else {
                    float* q = gout + (size_t)vi[k] * 4;
                    atomic_add_f32(q + 0, pg[r].g[k * 3 + 0]); atomic_add_f32(q + 1, pg[r].g[k * 3 + 1]); atomic_add_f32(q + 3, pg[r].g[k * 3 + 2]);
                }
            }
        }
    }

    // Flush: one atomic per (vertex, component) this block touched.
    __syncthreads();
    const int nflush = tab.compact(s_list, &s_used, threadIdx.x, kGradThreads) * 3;
    for (int i = threadIdx.x; i < nflush; i += kGradThreads) {
        const int u = i / 3, c = i - u * 3;
        const int slot = s_list[u];
        const unsigned long long t = s_vals[slot * 3 + c];
        if (t) atomic_add_f32(gout + (size_t)(s_keys[slot] - 1u) * 4 + (c == 2 ? 3 : c), fs.to_float(t));
    }
}

// ---------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------

struct ScratchLayout { size_t rec, bbox, pool, binCount, binHi, binLoInv, ctlEnd, poolFinal, poolPeak, order, splitInfo, helpers, splitDone, splitKeys, binCursor, anyList, binList, total; int slots, poolBase, maxBins, poolSlots; long long listCap; };

// Per-bin triangle lists are made for meshes of kListMinTris triangles or more (NVDR_DEBUG bit 268435456: for every mesh,
// bit 536870912: for none -- development switches for A/B measurements).
static bool lists_enabled(int max_tri)
{
    if (debug_flags() & 536870912) return false;
    return max_tri >= kListMinTris || (debug_flags() & 268435456) != 0;
}

// pool_per_image: slots per image for the clipper's extra sub-triangles; < 0 or >= 6 * max_tri = the worst case.
static ScratchLayout scratch_layout(int N, int max_tri, int H, int W, long long pool_per_image = -1)
{
    ScratchLayout L;
    {   // bins of the largest viewport tile (torch_rasterize.cpp:99-102 tiling)
        const int Hp = (H + 7) & ~7, Wp = (W + 7) & ~7;
        const int tcx = (Wp + kMaxViewport - 1) / kMaxViewport, tcy = (Hp + kMaxViewport - 1) / kMaxViewport;
        const int tsx = ((Wp + tcx - 1) / tcx + 7) & ~7, tsy = ((Hp + tcy - 1) / tcy + 7) & ~7;
        L.maxBins = ((tsx + 63) / 64) * ((tsy + 63) / 64);
    }
    L.poolBase = (max_tri + 3) & ~3;                       // pool slots start 16-byte aligned in the AABB array
    const long long worst = (long long)(kSubPerTri - 1) * max_tri;
    L.poolSlots = (int)((pool_per_image < 0 || pool_per_image > worst) ? worst : pool_per_image);
    L.slots = L.poolBase + ((L.poolSlots + 3) & ~3);
    L.rec   = 0;
    L.bbox  = align_up(L.rec + (size_t)N * L.slots * 64, 256);
    L.pool  = align_up(L.bbox + (size_t)N * L.slots * 4, 256);
    // Control block [pool, ctlEnd): pool cursors and per-bin counts / slot ranges.  It must be zero when
    // k_setup starts; k_order, its last reader, leaves it zero again.
    L.binCount = L.pool + (size_t)N * 4;
    L.binHi = L.binCount + (size_t)N * L.maxBins * 4;
    L.binLoInv = L.binHi + (size_t)N * L.maxBins * 4;
    L.ctlEnd = L.binLoInv + (size_t)N * L.maxBins * 4;
    L.poolFinal = align_up(L.ctlEnd, 256);
    L.poolPeak = align_up(L.poolFinal + (size_t)N * 4, 256);
    L.order = align_up(L.poolPeak + 4, 256);
    // bins shared by several workgroups (k_order / k_fine): per work item a split descriptor, per XCD chunk the helper
    // items, per split an arrival counter and one key array per part (64 tiles x 64 pixels x 8 B)
    L.splitInfo = align_up(L.order + ((size_t)N * L.maxBins + 8) * 16, 256);          // (+ 8: a chunk's slots are the bins / 8 rounded up)
    L.helpers = align_up(L.splitInfo + ((size_t)N * L.maxBins + 8) * 4, 256);
    const int helpersPerChunk = lists_enabled(max_tri) ? kListHelpersPerChunk : kHelpersPerChunk;
    L.splitDone = align_up(L.helpers + (size_t)8 * helpersPerChunk * 16, 256);
    L.splitKeys = align_up(L.splitDone + (size_t)8 * kSplitsPerChunk * 4 * 3, 256);                 // arrival counters, then the splits' key-array numbers, then their bins
    // key arrays of 64 x 64 keys: one per PART in small launches (the exchange of k_fine<SPLIT>), one per shared bin in launches
    // over large meshes (k_fine<LIST>: kept at all ones between calls)
    L.total = align_up(L.splitKeys + (size_t)8 * (lists_enabled(max_tri) ? kSplitsPerChunk : helpersPerChunk + kSplitsPerChunk) * 4096 * 8, 256);
    // per-bin triangle lists (k_binscan / k_binfill), large meshes only: a cursor per bin, a flag per image, and room for two
    // list entries per triangle (one per clipper slot, up to a triangle's worth of those) -- triangles are small next to a 64x64-pixel bin when there are this many of them; lists that do
    // not fit are not made (their bins scan their slot range as before)
    L.binCursor = L.anyList = L.binList = L.total; L.listCap = 0;
    if (lists_enabled(max_tri)) {
        L.binCursor = L.total;
        L.anyList = align_up(L.binCursor + (size_t)N * L.maxBins * 4, 256);
        L.binList = align_up(L.anyList + (size_t)N * 4, 256);
        L.listCap = (long long)N * (2ll * L.poolBase + (L.poolSlots < L.poolBase ? L.poolSlots : L.poolBase));
        L.total = align_up(L.binList + (size_t)L.listCap * 4, 256);
    }
    return L;
}

}  // namespace nvdr

using namespace nvdr;

// Workgroups of k_fine the device keeps resident: 4 per CU (40 KB of LDS, 8 waves each).
static int resident_fine_workgroups()
{
    static int cached[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    if (!cached[dev]) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cached[dev] = cus * 4;
    }
    return cached[dev];
}

extern "C" size_t nvdr_rasterize_scratch_bytes(int N, int max_tri, int H, int W)
{
    if (N <= 0 || max_tri <= 0 || H <= 0 || W <= 0) return 0;
    return scratch_layout(N, max_tri, H, W).total;
}

extern "C" size_t nvdr_rasterize_scratch_bytes_pool(int N, int max_tri, int H, int W, long long pool_per_image)
{
    if (N <= 0 || max_tri <= 0 || H <= 0 || W <= 0) return 0;
    return scratch_layout(N, max_tri, H, W, pool_per_image).total;
}

extern "C" size_t nvdr_rasterize_pool_peak_offset(int N, int max_tri, int H, int W, long long pool_per_image)
{
    if (N <= 0 || max_tri <= 0 || H <= 0 || W <= 0) return 0;
    return scratch_layout(N, max_tri, H, W, pool_per_image).poolPeak;
}

extern "C" int nvdr_rasterize_fwd(const float* pos, const int32_t* tri, const int32_t* ranges,
                                  int instance_mode, int N, int V, int T, int max_tri, int H, int W,
                                  const uint32_t* peel_depth, uint32_t* depth_out,
                                  void* scratch, size_t scratch_bytes, int scratch_clean, long long pool_per_image,
                                  float* out, float* out_db, uint8_t* tile_flags, nvdrStream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (debug_flags() & 2097152) return NVDR_OK;             // development: host-overhead measurement, nothing is launched (tools/host_profile.py)
    NVDR_REQUIRE(pos && tri && out && out_db && scratch, "rasterize_fwd: null pointer");
    NVDR_REQUIRE(instance_mode || ranges, "rasterize_fwd: range mode needs ranges");
    NVDR_REQUIRE(N > 0 && V > 0 && T > 0 && max_tri > 0, "rasterize_fwd: empty input");
    NVDR_REQUIRE(H > 0 && W > 0, "resolution must be [>0, >0]");
    NVDR_REQUIRE(!((uintptr_t)pos & 15), "pos input tensor not aligned to float4");
    NVDR_REQUIRE(!((uintptr_t)out & 15), "out output tensor not aligned to float4");
    NVDR_REQUIRE(!((uintptr_t)out_db & 15), "out_db output tensor not aligned to float4");
    NVDR_REQUIRE(!((uintptr_t)scratch & 255), "scratch must be 256-byte aligned");
    NVDR_REQUIRE((long long)max_tri * kSubPerTri < (1ll << 31) / 4, "subtriangle count overflow");
    ScratchLayout L = scratch_layout(N, max_tri, H, W, pool_per_image);
    const bool shortPool = L.poolSlots < (kSubPerTri - 1) * max_tri;   // overflow possible: the caller reads poolPeak back
    if (scratch_bytes < L.total) { set_error("rasterize_fwd: scratch too small (%zu < %zu)", scratch_bytes, L.total); return NVDR_ERR_SCRATCH; }

    char* sb = (char*)scratch;
    uint4* rec = (uint4*)(sb + L.rec);
    uint32_t* bbox = (uint32_t*)(sb + L.bbox);
    int* pool = (int*)(sb + L.pool);
    int* binCount = (int*)(sb + L.binCount);
    int* binHi = (int*)(sb + L.binHi);
    int* binLoInv = (int*)(sb + L.binLoInv);
    int4* order = (int4*)(sb + L.order);

    const int Hp = (H + 7) & ~7, Wp = (W + 7) & ~7;
    // Viewport tiling for images beyond 2048 px (torch_rasterize.cpp:99-124).
    const int tcx = (Wp + kMaxViewport - 1) / kMaxViewport, tcy = (Hp + kMaxViewport - 1) / kMaxViewport;
    const int tsx = ((Wp + tcx - 1) / tcx + 7) & ~7, tsy = ((Hp + tcy - 1) / tcy + 7) & ~7;

    for (int ty = 0; ty < tcy; ty++)
    for (int tx = 0; tx < tcx; tx++) {
        Viewport vp;
        vp.offx = tx * tsx; vp.offy = ty * tsy;
        vp.vpw = (W - vp.offx) < tsx ? (W - vp.offx) : tsx;
        vp.vph = (H - vp.offy) < tsy ? (H - vp.offy) : tsy;
        if (vp.vpw <= 0 || vp.vph <= 0) continue;
        vp.xs = (float)W / (float)vp.vpw;
        vp.ys = (float)H / (float)vp.vph;
        vp.xo = (float)(W - vp.vpw - 2 * vp.offx) / (float)vp.vpw;
        vp.yo = (float)(H - vp.vph - 2 * vp.offy) / (float)vp.vph;


        SetupParams sp;
        sp.pos = pos; sp.tri = tri; sp.ranges = ranges;
        sp.instance = instance_mode ? 1 : 0; sp.N = N; sp.V = V; sp.T = T; sp.maxTri = max_tri; sp.poolBase = L.poolBase; sp.slots = L.slots;
        sp.vp = vp; sp.rec = rec; sp.bbox = bbox; sp.poolCount = pool;
        const int vpwPad = (vp.vpw + 7) & ~7, vphPad = (vp.vph + 7) & ~7;
        const int binsX = (vpwPad / 8 + kBinTiles - 1) / kBinTiles;
        const int binsY = (vphPad / 8 + kBinTiles - 1) / kBinTiles;
        const int totalBins = N * binsX * binsY;
        sp.binCount = binCount; sp.binsX = binsX; sp.binsY = binsY;
        sp.binHi = binHi; sp.binLoInv = binLoInv;
        int* poolFinal = (int*)(sb + L.poolFinal);
        int* poolPeak = shortPool ? (int*)(sb + L.poolPeak) : nullptr;
        if (shortPool && tx == 0 && ty == 0) NVDR_HIP_CHECK(hipMemsetAsync(poolPeak, 0, 4, stream));
        const int bpi = (max_tri + 255) / 256;
        const size_t histBytes = (size_t)3 * binsX * binsY * sizeof(int);
        // Every call leaves the control block zeroed (k_order); it is cleared here only when the caller
        // cannot vouch for that (first use of the buffer, another layout, a failed call).
        if (!scratch_clean && tx == 0 && ty == 0) {                // later viewport tiles inherit the clean block from the tile before
            NVDR_HIP_CHECK(hipMemsetAsync(pool, 0, L.ctlEnd - L.pool, stream));
            if (L.listCap > 0)                                     // the shared bins' key arrays of LIST launches rest at all ones
                NVDR_HIP_CHECK(hipMemsetAsync(sb + L.splitKeys, 0xFF, (size_t)8 * kSplitsPerChunk * 4096 * 8, stream));
        }
        {
            ProfileScope ps("raster_setup", stream);
            hipLaunchKernelGGL(k_setup, dim3((unsigned)((((long long)bpi * N + 7) / 8) * 8)), dim3(256), histBytes, stream, sp, bpi);
        }
        NVDR_LAUNCH_CHECK();
        // Bins with very many triangles are shared by several workgroups only when the launch cannot hide them: with
        // fewer bins than two rounds of resident workgroups the launch is as long as its heaviest bin (batch 16 at
        // 512^2: k_fine 91 -> 57 us); with more, the helpers' extra list building costs more than the shorter tail
        // gains (batch 64: 126 -> 130 us in round 2; measured again in round 5 with thresholds of 768 ... 2000 triangles:
        // 131 -> 133 ... 151 us -- the launch is bound by its throughput there, not by its longest workgroup).
        const bool dbgMode = debug_buffer() != nullptr || (debug_flags() & (4 | 8 | 16 | 128 | 256 | 512 | 2048 | 4096 | 8192)) != 0;
        const bool lists = L.listCap > 0 && !dbgMode;
        // (launches over large meshes always share: one bin may hold more triangles than all the others together)
        const bool split = !dbgMode && (lists || totalBins <= 2 * resident_fine_workgroups()) && !(debug_flags() & 1048576);
        const int splitTris = !split ? 0x7FFFFFFF : lists ? tune_int("NVDR_TUNE_LIST_SPLIT_TRIS", kListSplitTris) : kSplitTris;
        const int splitPart = lists ? tune_int("NVDR_TUNE_LIST_SPLIT_PART", kListSplitPart) : kSplitPart;
        const int helpersPerChunk = L.listCap > 0 ? kListHelpersPerChunk : kHelpersPerChunk;     // (as the layout was sized)
        int* binCursor = lists ? (int*)(sb + L.binCursor) : nullptr;
        if (lists) {
            ProfileScope ps("raster_binscan", stream);
            const bool all = (debug_flags() & 268435456) != 0;                    // development: a list for every bin with triangles
            hipLaunchKernelGGL(k_binscan, dim3(1), dim3(1024), 0, stream, binCount, binHi, binLoInv, pool, binCursor, (int*)(sb + L.anyList),
                               totalBins, binsX * binsY, N, L.slots - L.poolBase, L.listCap, all ? 0 : kListScanFactor, all ? -1 : kListScanBias);
            NVDR_LAUNCH_CHECK();
        }
        {
            ProfileScope ps("raster_order", stream);
            hipLaunchKernelGGL(k_order, dim3(8), dim3(1024), 0, stream, binCount, binHi, binLoInv, pool, poolFinal, poolPeak, N, order, totalBins,
                               (int*)(sb + L.splitInfo), (int4*)(sb + L.helpers), (int*)(sb + L.splitDone), splitTris, splitPart,
                               (int*)(sb + L.poolPeak + 16), binsX, binsY, (const int*)binCursor,
                               lists ? min(kListSplitMaxParts, tune_int("NVDR_TUNE_LIST_MAX_PARTS", kListSplitMaxParts)) : kSplitMaxParts, helpersPerChunk, (int*)(sb + L.splitDone) + 8 * kSplitsPerChunk,
                               (split && N < 8 && !(debug_flags() & 1073741824)) ? 1 : 0, (int*)(sb + L.splitDone) + 16 * kSplitsPerChunk);
        }
        NVDR_LAUNCH_CHECK();
        if (lists) {
            ListParams lp;
            lp.bbox = bbox; lp.ranges = ranges; lp.poolFinal = poolFinal; lp.binCursor = binCursor; lp.anyList = (int*)(sb + L.anyList);
            lp.binList = (uint32_t*)(sb + L.binList); lp.instance = sp.instance; lp.N = N; lp.T = T; lp.poolBase = L.poolBase; lp.slots = L.slots;
            lp.binsX = binsX; lp.binsY = binsY; lp.cap = L.listCap;
            const int fpi = (L.slots + kFillSlots - 1) / kFillSlots;
            ProfileScope ps("raster_binfill", stream);
            hipLaunchKernelGGL(k_binfill, dim3((unsigned)fpi * N), dim3(kFillThreads), 0, stream, lp, fpi);
            NVDR_LAUNCH_CHECK();
        }

        FineParams fp;
        fp.rec = rec; fp.bbox = bbox; fp.poolFinal = poolFinal; fp.ranges = ranges; fp.pos = pos; fp.tri = tri;
        fp.instance = sp.instance; fp.N = N; fp.V = V; fp.T = T; fp.maxTri = max_tri; fp.poolBase = L.poolBase; fp.slots = L.slots;
        fp.W = W; fp.H = H; fp.Wp = Wp; fp.Hp = Hp; fp.vp = vp;
        fp.binsX = binsX; fp.binsY = binsY; fp.totalBins = totalBins;
        fp.order = order;
        fp.splitInfo = (const int*)(sb + L.splitInfo); fp.helpers = (const int4*)(sb + L.helpers);
        fp.splitKeys = (unsigned long long*)(sb + L.splitKeys); fp.splitDone = (int*)(sb + L.splitDone);
        fp.splitKeyBase = (const int*)(sb + L.splitDone) + 8 * kSplitsPerChunk;
        fp.splitBin = (const int*)(sb + L.splitDone) + 16 * kSplitsPerChunk;
        fp.chunkNz = (const int*)(sb + L.poolPeak + 16);                                        // 8 ints behind the pool-demand counter
        fp.binList = lists ? (const uint32_t*)(sb + L.binList) : nullptr;
        fp.tileFlags = (debug_flags() & 67108864) ? nullptr : tile_flags; fp.tfW = (W + 7) >> 3; fp.tfH = (H + 7) >> 3;   // (timing switch; use with 33554432)
        {
            const TileFlags tfv = tile_flags_view(fp.tileFlags, N, H, W);
            fp.rowCov = (tfv.order && tcx == 1 && tcy == 1) ? tile_flags + tile_flags_rowcov_offset(tile_flags_order_offset(N, H, W), tfv.nBins) : nullptr;
        }
        fp.peel = peel_depth; fp.depth = depth_out; fp.out = out; fp.out_db = out_db;
        fp.xs = 2.f / (float)W; fp.xo = 1.f / (float)W - 1.f;
        fp.ys = 2.f / (float)H; fp.yo = 1.f / (float)H - 1.f;
        fp.dbg = debug_flags();
        fp.dbgbuf = debug_buffer();
        const int grid = ((totalBins + 7) / 8) * 8 + (split ? 8 * (lists ? kListHelpersPerChunk : kHelpersPerChunk) : 0);   // + the helper slots of every XCD
        {
            ProfileScope ps("raster_fine", stream);
#define NVDR_FINE(PEEL, WD)                                                                                                    \
    do {                                                                                                                       \
        if (dbgMode)    hipLaunchKernelGGL((k_fine<PEEL, WD, true, false>),  dim3(grid), dim3(kFineThreads), 0, stream, fp);  \
        else if (lists && split) {                                                                                             \
            hipLaunchKernelGGL((k_fine<PEEL, WD, false, true, true>),  dim3(grid), dim3(kFineThreads), 0, stream, fp);        \
            hipLaunchKernelGGL((k_fine<PEEL, WD, false, true, true, true>), dim3(8 * kSplitsPerChunk), dim3(kFineThreads), 0, stream, fp);   \
        }                                                                                                                      \
        else if (lists)          hipLaunchKernelGGL((k_fine<PEEL, WD, false, false, true>), dim3(grid), dim3(kFineThreads), 0, stream, fp);  \
        else if (split) hipLaunchKernelGGL((k_fine<PEEL, WD, false, true>),  dim3(grid), dim3(kFineThreads), 0, stream, fp);  \
        else            hipLaunchKernelGGL((k_fine<PEEL, WD, false, false>), dim3(grid), dim3(kFineThreads), 0, stream, fp);  \
    } while (0)
            if (peel_depth && depth_out)       NVDR_FINE(true, true);
            else if (peel_depth)               NVDR_FINE(true, false);
            else if (depth_out)                NVDR_FINE(false, true);
            else                               NVDR_FINE(false, false);
        }
        NVDR_LAUNCH_CHECK();
    }
    if (tile_flags && !(debug_flags() & 67108864)) {
        const TileFlags tf = tile_flags_view(tile_flags, N, H, W);
        if (tf.order) {
            ProfileScope ps("raster_flag_order", stream);
            const uint8_t* rowCov = tile_flags + tile_flags_rowcov_offset(tile_flags_order_offset(N, H, W), tf.nBins);
            int* partial = (int*)(sb + L.poolPeak + 64);                                 // 32 ints behind the pool counter and chunkNz
            const int single = tune_int("NVDR_TUNE_FLAG_ORDER_SINGLE", kFlagOrderSingle);
            if (tf.nBins <= single) {
                hipLaunchKernelGGL(k_flag_order<0>, dim3(1), dim3(kFlagOrderThreads), 0, stream, tf, rowCov, (int*)tf.order, partial, 1);
            } else {
                const int groups = min(kFlagOrderMaxGroups, (tf.nBins + kFlagOrderChunk - 1) / kFlagOrderChunk);
                hipLaunchKernelGGL(k_flag_order<1>, dim3(groups), dim3(kFlagOrderThreads), 0, stream, tf, rowCov, (int*)tf.order, partial, groups);
                hipLaunchKernelGGL(k_flag_order<2>, dim3(groups), dim3(kFlagOrderThreads), 0, stream, tf, rowCov, (int*)tf.order, partial, groups);
            }
            NVDR_LAUNCH_CHECK();
        }
    }
    return NVDR_OK;
}

extern "C" size_t nvdr_tile_flags_bytes(int N, int H, int W)
{
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    const long long nb = tile_flags_bins(N, H, W);
    const size_t off = tile_flags_order_offset(N, H, W);
    return nb > 0 ? tile_flags_rowcov_offset(off, nb) + (size_t)nb * 8 : off;                          // flags; order + count; k_fine's row bytes
}

extern "C" int nvdr_rasterize_grad(const float* pos, const int32_t* tri, const float* out,
                                   const float* dy, const float* ddb,
                                   int instance_mode, int N, int V, int T, int H, int W,
                                   float* grad_pos, const uint8_t* tile_flags, nvdrStream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (debug_flags() & 2097152) return NVDR_OK;             // development: host-overhead measurement, nothing is launched (tools/host_profile.py)
    NVDR_REQUIRE(pos && tri && out && (dy || ddb) && grad_pos, "rasterize_grad: null pointer");
    NVDR_REQUIRE(N > 0 && H > 0 && W > 0, "resolution must be [>0, >0, >0]");
    NVDR_REQUIRE(V > 0 && T > 0, "rasterize_grad: empty input");
    NVDR_REQUIRE(!((uintptr_t)pos & 15), "pos input tensor not aligned to float4");
    NVDR_REQUIRE(!((uintptr_t)dy & 15), "dy input tensor not aligned to float4");
    NVDR_REQUIRE(!((uintptr_t)ddb & 15), "ddb input tensor not aligned to float4");
    NVDR_REQUIRE(!((uintptr_t)out & 15), "out tensor not aligned to float4");
    GradParams p;
    p.pos = pos; p.tri = tri; p.out = out; p.dy = dy; p.ddb = ddb; p.grad = grad_pos;
    p.flags = tile_flags_view((debug_flags() & 33554432) ? nullptr : tile_flags, N, H, W, !(debug_flags() & 134217728));
    p.instance = instance_mode ? 1 : 0; p.N = N; p.V = V; p.T = T; p.W = W; p.H = H;
    p.xs = 2.f / (float)W; p.xo = 1.f / (float)W - 1.f;
    p.ys = 2.f / (float)H; p.yo = 1.f / (float)H - 1.f;
    p.dbg = debug_flags();
    const int gx = (W + kGradBlockW - 1) / kGradBlockW, gy = (H + kGradBlockH - 1) / kGradBlockH;
    const long long total = p.flags.order ? tile_flags_ordered_grid(p.flags, (64 / kGradBlockW) * (64 / kGradBlockH)) : (long long)gx * gy * N;   // (nvdr_device.hpp TileFlags)
    NVDR_REQUIRE(total < (1ll << 30), "rasterize_grad: too many pixel blocks");
    dim3 grid((unsigned)(((total + 7) / 8) * 8));
    {
        ProfileScope ps(!dy ? "raster_grad_db_only" : ddb ? "raster_grad_db" : "raster_grad", stream);
        if (!dy) hipLaunchKernelGGL((k_raster_grad<true, true>), grid, dim3(kGradThreads), 0, stream, p, gx, gy);   // ddb's share only, ADDED to grad_pos
        else if (ddb) hipLaunchKernelGGL(k_raster_grad<true>,  grid, dim3(kGradThreads), 0, stream, p, gx, gy);
        else     hipLaunchKernelGGL(k_raster_grad<false>, grid, dim3(kGradThreads), 0, stream, p, gx, gy);
    }
    NVDR_LAUNCH_CHECK();
    return NVDR_OK;
}
