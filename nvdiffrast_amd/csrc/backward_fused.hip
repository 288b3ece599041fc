// backward_fused.hip -- interpolate backward and rasterize backward in ONE pass over the image (gfx950).
//
// The metric's op graph is rasterize -> interpolate; its backward is two kernels that walk the same pixels:
// k_interp_grad (interpolate.cu:131-274) reads rast and dy and writes g_rast = dL/d(u, v), k_raster_grad
// (rasterize.cu:119-277) reads rast again plus that g_rast and turns it into position gradients.  Fused, a pixel's
// (u, v) gradient goes straight from the interpolation adjoint into the pixel-shader tape in registers:
//   * rast is read once instead of twice and g_rast is not re-read (537 MB less traffic at the headline batch),
//   * one vertex table per 64x16-pixel block accumulates BOTH gradients of a vertex -- A attribute components and the
//     (x, y, w) position components -- so a block's vertices are looked up, claimed and flushed once,
//   * one launch instead of two.
// g_rast is still WRITTEN when the caller asks for it (the operator layer does: autograd may have other consumers of
// rast's gradient; see ops.py `_InterpolateOp.backward`) and skipped when g_rast == NULL.
//
// Structure = the two kernels' (DESIGN.md 4.1): phase A computes per-pixel contributions in registers and publishes
// the block's largest magnitudes (two fixed-point scales: attribute and position contributions differ by orders of
// magnitude), phase B sums them over runs of equal triangle id (RunScan), adds run totals to the LDS table (64-bit
// fixed point) and flushes each touched (vertex, component) with one hardware f32 atomic.
// Workgroup = 64 x 16 pixels, 8 waves of two rows each: the eleven per-pixel values that must survive the barrier
// (dy[4], b0, b1 and nine position-gradient components... per row) fit 8 waves/SIMD only with two rows per wave.
#include "nvdr_device.hpp"
#include "nvdr_host.hpp"
#include "nvdr_raster_tape.hpp"

namespace nvdr {

constexpr int kFuMaxDiffAttrs = 32;                      // interpolate.h:18 IP_MAX_DIFF_ATTRS

struct FusedParams {
    const int* tri; const float* attr; const float* rast; const float* pos; const float* dy;
    float* gradAttr; float* gradPos; float* gradRaster;
    // pixel differentials (interpolate_grad_da + rasterize_grad_db): rast_db, upstream gradient of the attribute
    // differentials, gradient of rast_db (written like g_rast), the differentiated attributes
    const float* rastDB; const float* dda; float* gradRasterDB;
    int numDiffAttr, diffAll, dbToPos;                      // dbToPos: the rasterize call propagates rast_db's gradient (grad_db)
    int diffAttrs[kFuMaxDiffAttrs];
    int numTriangles, numVertices, numAttr;
    int width, height, depth;
    int attrBC, attrInstance, posInstance, dbg;
    float xs, xo, ys, yo;
    TileFlags flags;                                        // which 8x8 tiles of rast show a triangle at all, or f == nullptr
};

constexpr int kFuBlockW = 64;
constexpr int kFuBlockH = 16;
constexpr int kFuWaves = 8;
constexpr int kFuThreads = kFuWaves * 64;
constexpr int kFuRows = kFuBlockH / kFuWaves;          // rows per wave

__device__ __forceinline__ int fused_diff_index(const FusedParams& p, int i)
{
    int j = p.diffAll ? i : p.diffAttrs[i];
    if (j < 0) j += p.numAttr;                              // python-style (interpolate.cu:102-103)
    return (j >= 0 && j < p.numAttr) ? j : -1;
}

// ENABLE_DA: the variant with pixel differentials -- interpolate_grad_da (interpolate.cu:233-269) feeding
// rasterize_grad_db (rasterize.cu:214-267): config 3's backward pair.
template <int A_CT, bool WRITE_GRAST, bool ENABLE_DA>
__global__ __launch_bounds__(kFuThreads, ENABLE_DA ? 6 : 8) void k_interp_raster_grad(const FusedParams p, int slots, int gx, int gy)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_mem[];
    constexpr bool kRegs = (A_CT == 4 || A_CT == 2);        // upstream gradient stays in registers between the phases
    const int A = A_CT > 0 ? A_CT : p.numAttr;
    const int S = A + 3;                                    // table components per vertex: A attribute sums, then x, y, w
    unsigned long long* s_vals = (unsigned long long*)s_mem;
    uint32_t* s_keys = (uint32_t*)(s_mem + (size_t)slots * S * 8);
    uint32_t* s_max = s_keys + slots;                       // [0] attribute max, [1] used slots, [2] position max
    uint16_t* s_list = (uint16_t*)(s_max + 4);              // [slots] used slots (flush)
    int bx, by, pz;
    if (p.flags.order ? !decode_block_ordered(p.flags, gx, gy, kFuBlockW, kFuBlockH, bx, by, pz)
                      : !decode_block(gx, gy, p.depth, bx, by, pz)) return;
    // A block without any triangle (its 8 x 2 occupancy flags, two thirds of the bench's blocks) has nothing to accumulate:
    // its waves store their zeros and leave before the table is cleared and the workgroup meets at its barriers.
    if (p.flags.f && !(p.dbg & 536870912)) {
        const int l = threadIdx.x & 63;
        const int tx = bx * (kFuBlockW / 8) + (l & 7), ty = by * (kFuBlockH / 8) + ((l >> 3) & 1);
        uint8_t f = 0;
        if (l < 16 && tx < p.flags.w && ty < p.flags.h) f = p.flags.f[((size_t)pz * p.flags.h + ty) * p.flags.w + tx];
        if (__ballot(f != 0) == 0ull) {
            if (WRITE_GRAST) {
                const int x = bx * kFuBlockW + l;
#pragma unroll
                for (int r = 0; r < kFuRows; r++) {
                    const int y = by * kFuBlockH + (int)(threadIdx.x >> 6) * kFuRows + r;
                    if (y >= p.height || x >= p.width) continue;
                    const size_t pidx = ((size_t)pz * p.height + y) * p.width + x;
                    ((float4*)p.gradRaster)[pidx] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ENABLE_DA) ((float4*)p.gradRasterDB)[pidx] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            return;
        }
    }
    // Both rows of a wave lie in one 8-pixel tile row (the wave's first row is even): one flag, and both rows' rast, are
    // fetched before the table is cleared -- a wave's life is a chain of dependent loads (flag -> rast -> triangle ->
    // vertices, per row), and these are the two links whose addresses need nothing but the pixel.
    // (With pixel differentials only the flag: that variant has no registers for rast -- 52 instead of 20 bytes of scratch.)
    constexpr bool kEarlyRast = !ENABLE_DA;
    float4 rr2[kFuRows];
    const int px0 = bx * kFuBlockW + (int)(threadIdx.x & 63), y0 = by * kFuBlockH + (int)(threadIdx.x >> 6) * kFuRows;
    const bool tileEmpty = px0 < p.width && y0 < p.height && p.flags.empty(pz, y0, px0);
    if (kEarlyRast) {
#pragma unroll
        for (int r = 0; r < kFuRows; r++) {
            rr2[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (px0 < p.width && y0 + r < p.height && !tileEmpty)             // (an empty tile's rast is not read)
                rr2[r] = ((const float4*)p.rast)[((size_t)pz * p.height + (y0 + r)) * p.width + px0];
        }
        // (the upstream gradient, whose address is known just as early, is NOT fetched here: in a tile with triangles four
        // pixels in ten show none, and their dy would be fetched for nothing -- 140 instead of 133 us)
    }
    VertexTable tab{s_keys, s_vals, slots, S};
    tab.clear(threadIdx.x, kFuThreads);
    if (threadIdx.x < 4) s_max[threadIdx.x] = 0u;
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = bx * kFuBlockW + lane;
    const int row0 = by * kFuBlockH + wave * kFuRows;
    const size_t voffA = (p.attrInstance && !p.attrBC) ? (size_t)pz * p.numVertices : 0;
    const size_t voffP = p.posInstance ? (size_t)pz * p.numVertices : 0;
    const float* attr = p.attr + voffA * A;
    float* gattr = p.gradAttr + voffA * A;
    const float4* vb = (const float4*)p.pos + voffP;
    float* gpos = p.gradPos + voffP * 4;

    int   tri[kFuRows];
    bool  ok[kFuRows];
    float b0[kFuRows], b1[kFuRows];
    float4 yreg[kFuRows];
    float g[kFuRows][9];
    uint32_t uA = 0u, uP = 0u;                            // largest attribute / position contribution of this lane, as magnitude bits

    // ---- phase A -----------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < kFuRows; r++) {
        const int py = row0 + r;
        ok[r] = false; tri[r] = -1; b0[r] = 0.f; b1[r] = 0.f;
        yreg[r] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 9; k++) g[r][k] = 0.f;          // the run scan multiplies masked lanes by 0: keep them finite
        if (py >= p.height || px >= p.width) continue;
        const size_t pidx = ((size_t)pz * p.height + py) * p.width + px;
        float4 rr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kEarlyRast) rr = rr2[r];
        else if (!tileEmpty) rr = ((const float4*)p.rast)[pidx];                      // (an empty tile's rast is not read)
        const int triIdx = float_to_triidx(rr.w) - 1;
        if (triIdx < 0 || triIdx >= p.numTriangles) {
            if (WRITE_GRAST) {
                ((float4*)p.gradRaster)[pidx] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ENABLE_DA) ((float4*)p.gradRasterDB)[pidx] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            continue;
        }
        const int vi0 = p.tri[triIdx * 3 + 0], vi1 = p.tri[triIdx * 3 + 1], vi2 = p.tri[triIdx * 3 + 2];
        if (!indices_ok(vi0, vi1, vi2, p.numVertices))
            continue;                                       // corrupt indices: leave untouched (interpolate.cu:163-167)
        ok[r] = true; tri[r] = triIdx; b0[r] = rr.x; b1[r] = rr.y;
        const float* a0 = attr + (size_t)vi0 * A;
        const float* a1 = attr + (size_t)vi1 * A;
        const float* a2 = attr + (size_t)vi2 * A;
        const float4 P[3] = {vb[vi0], vb[vi1], vb[vi2]};    // in flight together with the attribute rows and dy
        const float* pdy = p.dy + pidx * A;
        const float bmax = fmaxf(fmaxf(fabsf(rr.x), fabsf(rr.y)), fabsf(1.f - rr.x - rr.y));

        float gb0 = 0.f, gb1 = 0.f, ymax = 0.f;
        if (A_CT == 4) {
            const float4 y = load_streaming((const float4*)pdy);
            const float4 x0 = *(const float4*)a0, x1 = *(const float4*)a1, x2 = *(const float4*)a2;
            gb0 = dot_diff(y, x0, x2);
            gb1 = dot_diff(y, x1, x2);
            ymax = __int_as_float((int)max(max(mag_bits(y.x), mag_bits(y.y)), max(mag_bits(y.z), mag_bits(y.w))));
            yreg[r] = y;
        } else if (A_CT == 2) {
            const float2 y = *(const float2*)pdy;
            const float2 x0 = *(const float2*)a0, x1 = *(const float2*)a1, x2 = *(const float2*)a2;
            gb0 = dot_diff(y, x0, x2);
            gb1 = dot_diff(y, x1, x2);
            ymax = __int_as_float((int)max(mag_bits(y.x), mag_bits(y.y)));
            yreg[r] = make_float4(y.x, y.y, 0.f, 0.f);
        } else {
            for (int i = 0; i < A; i++) {
                const float y = pdy[i];
                const float s2v = a2[i];
                gb0 = dot_diff(y, a0[i], s2v, gb0);
                gb1 = dot_diff(y, a1[i], s2v, gb1);
                ymax = __int_as_float((int)max(mag_bits(ymax), mag_bits(y)));
            }
        }
        if (WRITE_GRAST) ((float4*)p.gradRaster)[pidx] = make_float4(gb0, gb1, 0.f, 0.f);
        uA = max(uA, mag_bits(ymax * bmax));                // >= every |b_k * dy_i| (rounding is monotone)

        float4 gdb = make_float4(0.f, 0.f, 0.f, 0.f);       // gradient of rast_db = (du/dX, du/dY, dv/dX, dv/dY)
        if (ENABLE_DA) {
            const float4 db = ((const float4*)p.rastDB)[pidx];
            const float2* dda = ((const float2*)p.dda) + pidx * p.numDiffAttr;
            for (int i = 0; i < p.numDiffAttr; i++) {
                const int j = fused_diff_index(p, i);
                if (j < 0) continue;
                const float2 d = dda[i];
                const float dsdu = a0[j] - a2[j], dsdv = a1[j] - a2[j];
                gdb.x += dsdu * d.x; gdb.y += dsdu * d.y;
                gdb.z += dsdv * d.x; gdb.w += dsdv * d.y;
                const float du = d.x * db.x + d.y * db.y;
                const float dv = d.x * db.z + d.y * db.w;
                uA = max(max(uA, mag_bits(du)), max(mag_bits(dv), mag_bits(-du - dv)));
            }
            if (WRITE_GRAST) ((float4*)p.gradRasterDB)[pidx] = gdb;
            if (!p.dbToPos) gdb = make_float4(0.f, 0.f, 0.f, 0.f);       // rasterize(..., grad_db=False): not propagated to pos
        }

        // rasterize backward of this pixel with (gb0, gb1) [and gdb] as the upstream gradient of (u, v) [and their pixel
        // differentials]; pixels whose upstream gradient is all +-0 contribute nothing (rasterize.cu:143-148)
        const int nz_db = ENABLE_DA ? (__float_as_int(gdb.x) | __float_as_int(gdb.y) | __float_as_int(gdb.z) | __float_as_int(gdb.w)) : 0;
        if ((((uint32_t)(__float_as_int(gb0) | __float_as_int(gb1) | nz_db)) << 1) != 0u) {
            const float fx = p.xs * (float)px + p.xo;
            const float fy = p.ys * (float)py + p.yo;
            raster_tape<ENABLE_DA>(P, fx, fy, p.xs, p.ys, gb0, gb1, gdb, ENABLE_DA && (((uint32_t)nz_db) << 1) != 0u, g[r]);
#pragma unroll
            for (int k = 0; k < 9; k++) uP = max(uP, mag_bits(g[r][k]));
        }
    }
    block_max_update(&s_max[0], __int_as_float((int)uA));
    block_max_update(&s_max[2], __int_as_float((int)uP));
    __syncthreads();
    const uint32_t maxA = s_max[0], maxP = s_max[2];
    if ((maxA | maxP) == 0u) return;                        // no contribution anywhere in the block
    // no table (very wide vertices) / inf or NaN present in either gradient: the whole block goes through plain f32 atomics
    const bool direct = slots == 0 || maxA >= 0x7F800000u || maxP >= 0x7F800000u;
    const FixedScale fsA(direct || maxA == 0u ? 0x3F800000u : maxA);
    const FixedScale fsP(direct || maxP == 0u ? 0x3F800000u : maxP);

    // ---- phase B -----------------------------------------------------------------------
    int vi[kFuRows][3];
#pragma unroll
    for (int r = 0; r < kFuRows; r++) {
        const int t = ok[r] ? tri[r] : 0;
#pragma unroll
        for (int k = 0; k < 3; k++) vi[r][k] = p.tri[t * 3 + k];        // L1/L2 hits: cheaper than six registers across phase A
    }
#pragma unroll
    for (int r = 0; r < kFuRows; r++) {
        if (__ballot(ok[r]) == 0) continue;
        const RunScan rs(tri[r], ok[r]);
        // (no lane continues a run -- a mesh of sub-pixel triangles -- : every lane is its own run's tail and the seven scans
        // of the row, 84 DPP multiply-adds, would add zeros)
        const bool scanning = !direct && rs.any_merge();
        const bool emit = direct ? ok[r] : rs.tail;
        int s0 = -1, s1 = -1, s2 = -1;
        if (emit && !direct) tab.find3(vi[r][0], vi[r][1], vi[r][2], s0, s1, s2);
        const bool tabled = emit && (s0 | s1 | s2) >= 0;
        const bool anyLoose = __ballot(emit && !tabled) != 0ull;
        // attribute component i of the three vertices
        auto put3 = [&](int i, float v0, float v1, float v2) {
            if (tabled) { tab.add(s0, i, fsA.to_fixed(v0)); tab.add(s1, i, fsA.to_fixed(v1)); tab.add(s2, i, fsA.to_fixed(v2)); }
            if (anyLoose && emit && !tabled) {
                const int sl[3] = {s0, s1, s2}; const float vv[3] = {v0, v1, v2};
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    if (sl[k] >= 0) tab.add(sl[k], i, fsA.to_fixed(vv[k]));
                    else atomic_add_f32(gattr + (size_t)vi[r][k] * A + i, vv[k]);
                }
            }
        };
        const float c0 = ok[r] ? b0[r] : 0.f, c1 = ok[r] ? b1[r] : 0.f, c2 = ok[r] ? 1.f - b0[r] - b1[r] : 0.f;
        const size_t pidx = ((size_t)pz * p.height + (row0 + r)) * p.width + px;
        const float* pdy = p.dy + pidx * A;
        for (int i = 0; i < A; i++) {
            float y;
            if (kRegs) y = i == 0 ? yreg[r].x : i == 1 ? yreg[r].y : i == 2 ? yreg[r].z : yreg[r].w;
            else       y = ok[r] ? pdy[i] : 0.f;
            float v0 = c0 * y, v1 = c1 * y, v2 = c2 * y;
            if (scanning) rs.scan3(v0, v1, v2);
            put3(i, v0, v1, v2);
        }
        if (ENABLE_DA) {
            // attribute gradients through the pixel differentials: du, dv, -(du + dv) to the three vertices' attribute j
            float4 db = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok[r]) db = ((const float4*)p.rastDB)[pidx];
            const float2* dda = ((const float2*)p.dda) + pidx * p.numDiffAttr;
            for (int i = 0; i < p.numDiffAttr; i++) {
                const int j = fused_diff_index(p, i);
                if (j < 0) continue;
                float2 d = make_float2(0.f, 0.f);
                if (ok[r]) d = dda[i];
                float du = d.x * db.x + d.y * db.y;
                float dv = d.x * db.z + d.y * db.w;
                float dw = -du - dv;
                if (scanning) rs.scan3(du, dv, dw);
                put3(j, du, dv, dw);
            }
        }
        // position components (x, y, w) of vertex k
        if (maxP != 0u) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float v0 = g[r][k * 3 + 0], v1 = g[r][k * 3 + 1], v2 = g[r][k * 3 + 2];
                if (scanning) rs.scan3(v0, v1, v2);
                if (!emit) continue;
                const int s = k == 0 ? s0 : k == 1 ? s1 : s2;
                if (s >= 0) {
                    tab.add(s, A + 0, fsP.to_fixed(v0)); tab.add(s, A + 1, fsP.to_fixed(v1)); tab.add(s, A + 2, fsP.to_fixed(v2));
                } else {
                    float* q = gpos + (size_t)vi[r][k] * 4;
                    atomic_add_f32(q + 0, v0); atomic_add_f32(q + 1, v1); atomic_add_f32(q + 3, v2);
                }
            }
        }
    }
    if (direct) return;

    // Flush: one atomic per (vertex, component) this block touched; consecutive lanes take the components of one vertex.
    __syncthreads();
    const int n = tab.compact(s_list, &s_max[1], threadIdx.x, kFuThreads) * S;
    for (int i = threadIdx.x; i < n; i += kFuThreads) {
        const int u = i / S, c = i - u * S;
        const int slot = s_list[u];
        const unsigned long long t = s_vals[slot * S + c];
        if (!t) continue;
        const size_t v = (size_t)(s_keys[slot] - 1u);
        if (c < A) atomic_add_f32(gattr + v * A + c, fsA.to_float(t));
        else       atomic_add_f32(gpos + v * 4 + (c - A == 2 ? 3 : c - A), fsP.to_float(t));
    }
}

}  // namespace nvdr

using namespace nvdr;

// LDS of one workgroup for `slots` table entries of A + 3 components: sums, key, used-list entry, header.
static size_t fused_lds_bytes(int slots, int A) { return (size_t)slots * (8 * (size_t)(A + 3) + 6) + 16; }

extern "C" int nvdr_interpolate_rasterize_grad(const float* attr, const float* rast, const int32_t* tri, const float* pos,
                                               const float* dy, int attr_instance, int attr_n, int pos_instance,
                                               int N, int V, int A, int T, int H, int W,
                                               const float* rast_db, const float* dda,
                                               int diff_all, const int32_t* diff_attrs_host, int num_diff, int db_to_pos,
                                               float* g_attr, float* g_pos, float* g_rast, float* g_rast_db,
                                               const uint8_t* tile_flags, nvdrStream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (debug_flags() & 2097152) return NVDR_OK;             // development: host-overhead measurement, nothing is launched
    NVDR_REQUIRE(attr && rast && tri && pos && dy && g_attr && g_pos, "interpolate_rasterize_grad: null pointer");
    NVDR_REQUIRE(N > 0 && H > 0 && W > 0, "rast must have shape[>0, >0, >0, 4]");
    NVDR_REQUIRE(T > 0, "tri must have shape [>0, 3]");
    NVDR_REQUIRE(V > 0 && A > 0, "attr must have shape [>0, >0, >0] or [>0, >0]");
    if (attr_instance) NVDR_REQUIRE(attr_n == N || attr_n == 1, "minibatch size mismatch between inputs rast, attr");
    NVDR_REQUIRE(!((uintptr_t)rast & 15), "rast input tensor not aligned to float4");
    NVDR_REQUIRE(!((uintptr_t)pos & 15), "pos input tensor not aligned to float4");
    NVDR_REQUIRE(!((uintptr_t)g_rast & 15), "grad_rast output tensor not aligned to float4");
    const bool enable_da = rast_db && dda && (diff_all || num_diff > 0);
    NVDR_REQUIRE(!enable_da || !g_rast || g_rast_db, "interpolate_rasterize_grad: g_rast_db missing");
    NVDR_REQUIRE(!((uintptr_t)rast_db & 15), "rast_db input tensor not aligned to float4");
    NVDR_REQUIRE(!((uintptr_t)dda & 7), "dda input tensor not aligned to float2");
    NVDR_REQUIRE(!((uintptr_t)g_rast_db & 15), "grad_rast_db output tensor not aligned to float4");
    FusedParams p{};
    p.tri = tri; p.attr = attr; p.rast = rast; p.pos = pos; p.dy = dy;
    p.gradAttr = g_attr; p.gradPos = g_pos; p.gradRaster = g_rast;
    if (enable_da) {
        p.rastDB = rast_db; p.dda = dda; p.gradRasterDB = g_rast_db; p.dbToPos = db_to_pos ? 1 : 0;
        if (diff_all) { p.numDiffAttr = A; p.diffAll = 1; }
        else {
            NVDR_REQUIRE(num_diff <= kFuMaxDiffAttrs, "too many entries in diff_attrs list (increase IP_MAX_DIFF_ATTRS)");
            NVDR_REQUIRE(diff_attrs_host, "interpolate_rasterize_grad: diff_attrs list missing");
            p.numDiffAttr = num_diff;
            for (int i = 0; i < num_diff; i++) p.diffAttrs[i] = diff_attrs_host[i];
        }
    }
    p.numTriangles = T; p.numVertices = V; p.numAttr = A;
    p.width = W; p.height = H; p.depth = N;
    p.attrInstance = attr_instance ? 1 : 0;
    p.attrBC = (attr_instance && attr_n == 1) ? 1 : 0;
    p.posInstance = pos_instance ? 1 : 0;
    p.dbg = debug_flags();
    p.flags = tile_flags_view((p.dbg & 33554432) ? nullptr : tile_flags, N, H, W, !(p.dbg & 134217728));
    p.xs = 2.f / (float)W; p.xo = 1.f / (float)W - 1.f;
    p.ys = 2.f / (float)H; p.yo = 1.f / (float)H - 1.f;
    const int gx = (W + kFuBlockW - 1) / kFuBlockW, gy = (H + kFuBlockH - 1) / kFuBlockH;
    // with a work order behind the flags the launch walks the order's bins, four blocks each (nvdr_device.hpp TileFlags)
    const long long total = p.flags.order ? tile_flags_ordered_grid(p.flags, (64 / kFuBlockW) * (64 / kFuBlockH)) : (long long)gx * gy * N;
    NVDR_REQUIRE(total < (1ll << 30), "interpolate_rasterize_grad: too many pixel blocks");
    dim3 grid((unsigned)(((total + 7) / 8) * 8)), block(kFuThreads);
    // LDS vertex table: as many power-of-two slots as fit in 32 KiB (four 8-wave workgroups per CU), at most 512;
    // none (plain atomics) for vertices too wide for even 32 slots in 64 KiB
    int slots = 512;
    while (slots > 32 && fused_lds_bytes(slots, A) > 32 * 1024) slots >>= 1;
    if (fused_lds_bytes(slots, A) > 64 * 1024) slots = 0;
    const size_t lds = fused_lds_bytes(slots, A);
    const bool vec4 = (A == 4) && !((uintptr_t)attr & 15) && !((uintptr_t)dy & 15);
    const bool vec2 = (A == 2) && !((uintptr_t)attr & 7) && !((uintptr_t)dy & 7);
    {
        ProfileScope ps(enable_da ? "interp_raster_grad_da" : "interp_raster_grad", stream);
#define NVDR_FUSED(ACT, DA)                                                                                                      \
    do {                                                                                                                          \
        if (g_rast) hipLaunchKernelGGL((k_interp_raster_grad<ACT, true, DA>),  grid, block, lds, stream, p, slots, gx, gy);      \
        else        hipLaunchKernelGGL((k_interp_raster_grad<ACT, false, DA>), grid, block, lds, stream, p, slots, gx, gy);      \
    } while (0)
        if (enable_da) { if (vec4) NVDR_FUSED(4, true); else if (vec2) NVDR_FUSED(2, true); else NVDR_FUSED(0, true); }
        else           { if (vec4) NVDR_FUSED(4, false); else if (vec2) NVDR_FUSED(2, false); else NVDR_FUSED(0, false); }
    }
    NVDR_LAUNCH_CHECK();
    return NVDR_OK;
}
