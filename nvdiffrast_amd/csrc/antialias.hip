// antialias.hip -- silhouette antialiasing forward / backward for gfx950.
//
// Replaces csrc/common/antialias.cu + csrc/torch/torch_antialias.cpp behind the C ABI.
//
//  k_aa_mesh           one lane per triangle: edge -> opposite-vertex table (open addressing,
//                      64-bit CAS on the key, then CAS into the first free of two value slots;
//                      antialias.cu:82-96,139-160).  The table layout is private to this library
//                      (the reference's TopologyHashWrapper is opaque, torch_types.h:37-45).
//  k_aa_discontinuity  workgroup = 64x32 pixel block in scan-line order; candidate (pixel, right|down)
//                      pairs are counted per wave, and the block reserves their slots in the work
//                      buffer with ONE returning atomic per 2048 pixels (the shared counter is the
//                      only contended word; the reference: one atomic per 32x8 CTA, :197-209).
//  k_aa_analysis       one lane per work item, grid-stride (item count lives on the device):
//                      silhouette test and edge crossing (:236-379); the blends of a workgroup's 256
//                      items are staged in LDS and added with the channels of one pixel on
//                      consecutive lanes (one memory transaction per pixel instead of one per channel).
//  k_aa_grad           one lane per work item with alpha != 0 (:406-554); colour and position
//                      updates are emitted transposed in the same way.
#include "nvdr_device.hpp"
#include "nvdr_host.hpp"

namespace nvdr {

constexpr float kF32Max = 3.402823466e+38f;

struct AAParams {
    const float* color; const float* rast; const int* tri; const float* pos;
    float* output; const float* dy; float* gradColor; float* gradPos;
    int4* work; uint4* hash;
    unsigned hashMask;
    int numTriangles, numVertices, width, height, n, channels, instance;
    float xh, yh;
    TileFlags flags;                                        // which 8x8 tiles of rast show a triangle at all, or f == nullptr
};

__device__ __forceinline__ bool same_sign(float a, float b) { return (__float_as_int(a) ^ __float_as_int(b)) >= 0; }
__device__ __forceinline__ bool rational_gt(float n0, float n1, float d0, float d1)
{
#pragma clang fp contract(off)
    return (n0 * d1 > n1 * d0) == same_sign(d0, d1);
}
__device__ __forceinline__ int max_idx3(float n0, float n1, float n2, float d0, float d1, float d2)
{
    const bool g10 = rational_gt(n1, n0, d1, d0);
    const bool g20 = rational_gt(n2, n0, d2, d0);
    const bool g21 = rational_gt(n2, n1, d2, d1);
    if (g20 && g21) return 2;
    if (g10) return 1;
    return 0;
}

// ---- topology table ------------------------------------------------------------------------

__device__ __forceinline__ void hash_start(unsigned long long key, unsigned mask, unsigned& idx, unsigned& skip)
{
    unsigned long long k = key;
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    idx = (unsigned)k & mask;
    skip = ((unsigned)(k >> 32) & mask) | 1u;          // odd skip visits every slot of a power-of-two table
}

__device__ __forceinline__ unsigned long long edge_key(int va, int vb)
{
    const unsigned long long v0 = (unsigned)min(va, vb) + 1u, v1 = (unsigned)max(va, vb) + 1u;
    return v0 | (v1 << 32);
}

__device__ void hash_insert_vertex(const AAParams& p, int va, int vb, int vn)
{
    if (va == vb) return;
    const unsigned long long key = edge_key(va, vb);
    unsigned idx, skip;
    hash_start(key, p.hashMask, idx, skip);
    for (;;) {
        const unsigned long long prev = atomicCAS((unsigned long long*)&p.hash[idx], 0ull, key);
        if (prev == 0ull || prev == key) break;
        idx = (idx + skip) & p.hashMask;
    }
    int* q = (int*)&p.hash[idx];
    const int v = vn + 1;
    const int a = atomicCAS(q + 2, 0, v);
    if (a != 0 && a != v) atomicCAS(q + 3, 0, v);
}

__device__ __forceinline__ int hash_find_vertex(const AAParams& p, int va, int vb, int vr)
{
    if (va == vb) return -1;
    const unsigned long long key = edge_key(va, vb);
    unsigned idx, skip;
    hash_start(key, p.hashMask, idx, skip);
    for (;;) {
        const uint4 e = p.hash[idx];
        const unsigned long long k = (unsigned long long)e.x | ((unsigned long long)e.y << 32);
        if (k == key || k == 0ull) {
            const int x = (int)e.z - 1, y = (int)e.w - 1;
            if (k == 0ull) return -1;
            if (x == vr) return y;
            if (y == vr) return x;
            return -1;
        }
        idx = (idx + skip) & p.hashMask;
    }
}

__global__ __launch_bounds__(256) void k_aa_mesh(const AAParams p)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.numTriangles) return;
    const int v0 = p.tri[idx * 3 + 0], v1 = p.tri[idx * 3 + 1], v2 = p.tri[idx * 3 + 2];
    if (v0 < 0 || v1 < 0 || v2 < 0) return;                // numVertices is unknown here (torch_antialias.cpp:40)
    if (v0 == v1 || v1 == v2 || v2 == v0) return;
    hash_insert_vertex(p, v1, v2, v0);
    hash_insert_vertex(p, v2, v0, v1);
    hash_insert_vertex(p, v0, v1, v2);
}

// ---- discontinuity finder (antialias.cu:165-214) ------------------------------------------------

// Workgroup = 64 x 32 pixel block (4 waves x 8 rows of 64 pixels, 1 KiB-stride coalesced reads of
// the id channel).  Candidates are counted per wave with ballots, waves get their offsets from an
// LDS counter, and the block reserves its slots in the work buffer with ONE global atomic, so the
// shared counter sees one returning atomic per 2048 pixels.
constexpr int kAaBlockW = 64, kAaBlockH = 32, kAaRows = 8;

__global__ __launch_bounds__(256) void k_aa_discontinuity(const AAParams p, int gx, int gy)
{
    __shared__ int s_total, s_base;
    int bx, by, pz;
    if (p.flags.order ? !decode_block_ordered(p.flags, gx, gy, kAaBlockW, kAaBlockH, bx, by, pz)
                      : !decode_block(gx, gy, p.n, bx, by, pz)) return;
    if (threadIdx.x == 0) s_total = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = bx * kAaBlockW + lane;
    const int row0 = by * kAaBlockH + wave * kAaRows;
    // out = color for this block's pixels (torch_antialias.cpp:122 clones the whole image first): the launch visits every pixel
    // exactly once, so the copy rides along -- its 8 C bytes per pixel stream while the id reads below wait for theirs -- instead
    // of being a launch of its own in front (403 MB each way at config 3: 0.14 ms).  k_aa_analysis, the next launch, blends into it.
    if (p.output) {
        const int cols = min(kAaBlockW, p.width - bx * kAaBlockW);
        const int rowFloats = cols * p.channels;
        const size_t base0 = ((size_t)bx * kAaBlockW + (size_t)p.width * (row0 + (size_t)p.height * pz)) * p.channels;
        const size_t rowStride = (size_t)p.width * p.channels;
        // 16-byte granules when every row of the wave allows it (row starts 16-byte aligned in both tensors, whole granules per row)
        const bool vec = ((((uintptr_t)(p.color + base0)) | ((uintptr_t)(p.output + base0)) | (uintptr_t)(rowStride * 4)) & 15) == 0 && (rowFloats & 3) == 0;
        const int rows = min(kAaRows, p.height - row0);
        if (vec && rowFloats <= 256 && rows == kAaRows) {
            // all rows' loads first, then the stores: row by row every load waited for the store before it (loads and stores share
            // one in-order counter on this architecture), and the copy ran at eight memory round trips per wave (0.25 ms at
            // config 3 for what a plain copy does in 0.15)
            nvdr_v4f v[kAaRows];                                              // (the native vector type: an array of HIP's float4 stays in memory)
            const int n4 = rowFloats >> 2;
            if (lane < n4) {
#pragma unroll
                for (int r = 0; r < kAaRows; r++) v[r] = ((const nvdr_v4f*)(p.color + base0 + (size_t)r * rowStride))[lane];
#pragma unroll
                for (int r = 0; r < kAaRows; r++) ((nvdr_v4f*)(p.output + base0 + (size_t)r * rowStride))[lane] = v[r];
            }
        } else {
#pragma unroll 1
            for (int r = 0; r < rows; r++) {
                const size_t base = base0 + (size_t)r * rowStride;
                for (int i = lane; i < rowFloats; i += 64) p.output[base + i] = p.color[base + i];
            }
        }
    }
    uint32_t c1 = 0, c2 = 0;                                 // bit r: candidate (right / down) in row r of this lane
    int cnt = 0;
    if (px < p.width) {
#pragma unroll
        for (int r = 0; r < kAaRows; r++) {
            const int py = row0 + r;
            if (py >= p.height) break;
            const size_t pidx = (size_t)px + (size_t)p.width * (py + (size_t)p.height * pz);
            // no triangle in this pixel's tile nor in its right / lower neighbour's: three equal ids, nothing to read
            if (p.flags.f && p.flags.empty(pz, py, px) && (px + 1 >= p.width || p.flags.empty(pz, py, px + 1))
                && (py + 1 >= p.height || p.flags.empty(pz, py + 1, px))) continue;
            const float tri0 = p.rast[pidx * 4 + 3];         // compared as floats, like the reference
            if (px < p.width - 1 && p.rast[(pidx + 1) * 4 + 3] != tri0) { c1 |= 1u << r; cnt++; }
            if (py < p.height - 1 && p.rast[(pidx + p.width) * 4 + 3] != tri0) { c2 |= 1u << r; cnt++; }
        }
    }
    // Exclusive prefix of `cnt` over the wave (the order of work items is irrelevant).
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
    const int waveTotal = __shfl(incl, 63, 64);
    int waveBase = 0;
    if (lane == 0 && waveTotal) waveBase = atomicAdd(&s_total, waveTotal);
    waveBase = __builtin_amdgcn_readfirstlane(waveBase);
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_total ? atomicAdd(&p.work[0].x, s_total) : 0;
    __syncthreads();
    if (cnt == 0) return;
    int idx = s_base + 1 + waveBase + (incl - cnt);          // slot 0 holds the counters
#pragma unroll
    for (int r = 0; r < kAaRows; r++) {
        const int py = row0 + r;
        if (c1 & (1u << r)) p.work[idx++] = make_int4(px, py, pz << 16, 0);
        if (c2 & (1u << r)) p.work[idx++] = make_int4(px, py, (pz << 16) + (1 << 2), 0);
    }
}

// ---- analysis + blend (antialias.cu:219-382) ------------------------------------------------------

__device__ __forceinline__ void swapf(float& a, float& b) { const float t = a; a = b; b = t; }

// c * w * half - f and a * b - c * d as nvcc's default -fmad=true fuses them in the reference (antialias.cu:307-325,
// 346-348,510-517): whether an edge lying exactly on a pixel boundary yields |alpha| = 0.5 (position gradient
// killed, :541-546) or 0.49999997 depends on it.  Pinned by the fma build of oracle/_ref.
__device__ __forceinline__ float aa_proj(float c, float w, float half, float f) { return __fmaf_rn(c * w, half, -f); }
__device__ __forceinline__ float aa_cross(float a, float b, float c, float d)  { return __fmaf_rn(a, b, -(c * d)); }

// One work item: silhouette test and edge crossing; writes the blend back into the item and returns it
// (0 = no blend) with the two pixel indices.
__device__ __forceinline__ float aa_analyse_item(const AAParams& p, int item_idx, int& o_pix0, int& o_pix1)
{
#pragma clang fp contract(off)
    {
        int4* pItem = p.work + item_idx + 1;
        const int4 item = *pItem;
        int px = item.x, py = item.y;
        const int pz = (int)(((unsigned)item.z) >> 16);
        const int d = (item.z >> 2) & 1;

        const size_t pixel0 = (size_t)px + (size_t)p.width * (py + (size_t)p.height * pz);
        const size_t pixel1 = pixel0 + (d ? (size_t)p.width : 1);
        const float2 zt0 = ((const float2*)p.rast)[pixel0 * 2 + 1];
        const float2 zt1 = ((const float2*)p.rast)[pixel1 * 2 + 1];
        const int tri0 = float_to_triidx(zt0.y) - 1;
        const int tri1 = float_to_triidx(zt1.y) - 1;

        int tri = (tri0 >= 0) ? tri0 : tri1;
        if (tri0 >= 0 && tri1 >= 0) tri = (zt0.x < zt1.x) ? tri0 : tri1;
        if (tri == tri1) { px += 1 - d; py += d; }
        if (tri < 0 || tri >= p.numTriangles) return 0.f;

        int vi[3];
        bool bad = false;
#pragma unroll
        for (int k = 0; k < 3; k++) { vi[k] = p.tri[tri * 3 + k]; bad |= (vi[k] < 0 || vi[k] >= p.numVertices); }
        if (bad) return 0.f;

        // Triangle corners and, per edge, the vertex across it in the neighbouring triangle (the corner
        // itself when the edge has no neighbour: always a silhouette), projected to pixel units
        // relative to the centre of the pixel the edge distance is measured from (:275-319).
        const float4* vb = (const float4*)p.pos + (p.instance ? (size_t)pz * p.numVertices : 0);
        const float fx = (float)px + .5f - p.xh;
        const float fy = (float)py + .5f - p.yh;
        float x[3], y[3], ox[3], oy[3];
        // The three corners first, then the three edges' table lookups SIDE BY SIDE -- one probe of each per round, so the
        // rounds' loads are in flight together -- then the three opposite vertices together: looked up one edge after the other
        // (probe, corner, opposite vertex, next edge ...) an item was a chain of up to nine dependent loads, and the kernel
        // waits for memory 83 % of its time.  The probe sequence of each edge is hash_find_vertex's.
        float4 c[3];
#pragma unroll
        for (int k = 0; k < 3; k++) c[k] = vb[vi[k]];
        int op[3];
        unsigned long long key[3];
        unsigned hidx[3], hskip[3];
        bool open[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int i = (k + 1) % 3, j = (k + 2) % 3;             // vertex opposite to corner k across edge (i, j)
            op[k] = -1;
            open[k] = vi[j] != vi[i];
            key[k] = edge_key(vi[j], vi[i]);
            hash_start(key[k], p.hashMask, hidx[k], hskip[k]);
        }
        while (open[0] | open[1] | open[2]) {
            uint4 e[3];
#pragma unroll
            for (int k = 0; k < 3; k++) e[k] = p.hash[open[k] ? hidx[k] : 0u];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (!open[k]) continue;
                const unsigned long long kk = (unsigned long long)e[k].x | ((unsigned long long)e[k].y << 32);
                if (kk == key[k] || kk == 0ull) {
                    const int a = (int)e[k].z - 1, b = (int)e[k].w - 1;
                    if (kk != 0ull) op[k] = (a == vi[k]) ? b : (b == vi[k]) ? a : -1;
                    open[k] = false;
                } else hidx[k] = (hidx[k] + hskip[k]) & p.hashMask;
            }
        }
        float4 o[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (op[k] >= p.numVertices) op[k] = -1;                 // a table built for another mesh may name vertices this one lacks
            o[k] = vb[op[k] < 0 ? vi[k] : op[k]];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float w = 1.f / c[k].w, ow = 1.f / o[k].w;
            x[k] = aa_proj(c[k].x, w, p.xh, fx);   y[k] = aa_proj(c[k].y, w, p.yh, fy);
            ox[k] = aa_proj(o[k].x, ow, p.xh, fx); oy[k] = aa_proj(o[k].y, ow, p.yh, fy);
        }

        // Orientation of the triangle and of each "wing" (edge + opposite vertex): an edge whose wing
        // folds to the same side as the triangle is a silhouette (:321-328).
        const float bb = aa_cross(x[1] - x[0], y[2] - y[0], x[2] - x[0], y[1] - y[0]);
        bool sil[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int i = (k + 1) % 3, j = (k + 2) % 3;
            const float wing = aa_cross(x[i] - ox[k], y[j] - oy[k], x[j] - ox[k], y[i] - oy[k]);
            sil[k] = same_sign(wing, bb);
        }
        if (!(sil[0] || sil[1] || sil[2])) return 0.f;

        // Work in a frame where the pixel pair is horizontal (:330-336), then find the edge that
        // crosses the segment between the two pixel centres nearest to this pixel (:338-359).
        if (d) { swapf(x[0], y[0]); swapf(x[1], y[1]); swapf(x[2], y[2]); }
        const float ds = (tri == tri0) ? 1.f : -1.f;
        float ex[3], ey[3], dist[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int i = (k + 1) % 3, j = (k + 2) % 3;
            ex[k] = x[j] - x[i]; ey[k] = y[j] - y[i];
            dist[k] = ds * aa_cross(x[i], ey[k], y[i], ex[k]);
            if (same_sign(y[i], y[j])) { dist[k] = -kF32Max; ey[k] = 1.f; }     // the edge does not cross the row
        }
        const int di = max_idx3(dist[0], dist[1], dist[2], ey[0], ey[1], ey[2]);
        float dc = -kF32Max;
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (di == k && sil[k] && fabsf(ey[k]) >= fabsf(ex[k])) dc = dist[k] / ey[k];
        const float eps = .0625f;

        if (dc > -eps && dc < 1.f + eps) {
            dc = fminf(fmaxf(dc, 0.f), 1.f);
            const float alpha = ds * (.5f - dc);
            unsigned flags = (unsigned)pz << 16;
            flags |= (unsigned)di;
            flags |= (unsigned)d << 2;
            flags |= ((unsigned)__float_as_int(ds) >> 31) << 3;
            ((int2*)pItem)[1] = make_int2((int)flags, __float_as_int(alpha));
            o_pix0 = (int)pixel0; o_pix1 = (int)pixel1;
            return alpha;
        }
    }
    return 0.f;
}

// The blend itself is emitted transposed (see k_aa_grad): consecutive lanes add the consecutive channels of
// one pixel, so the C atomics of a pair reach memory as one transaction instead of C.
__global__ __launch_bounds__(256) void k_aa_analysis(const AAParams p)
{
#pragma clang fp contract(off)
    __shared__ float s_alpha[256];
    __shared__ int s_pix[256][2];
    const int workCount = p.work[0].x;
    const int C = p.channels;
    for (int base = blockIdx.x * 256; base < workCount; base += gridDim.x * 256) {      // uniform trip count: barriers inside
        const int item_idx = base + threadIdx.x;
        int p0 = 0, p1 = 0;
        const float alpha = (item_idx < workCount) ? aa_analyse_item(p, item_idx, p0, p1) : 0.f;
        __syncthreads();                                                   // previous round's readers are done
        s_alpha[threadIdx.x] = alpha; s_pix[threadIdx.x][0] = p0; s_pix[threadIdx.x][1] = p1;
        __syncthreads();
        for (int f = threadIdx.x; f < 256 * C; f += 256) {
            const int j = f / C, c = f - j * C;
            const float al = s_alpha[j];
            if (al == 0.f) continue;
            const size_t q0 = (size_t)s_pix[j][0] * C + c, q1 = (size_t)s_pix[j][1] * C + c;
            atomic_add_f32(p.output + (al > 0.f ? q0 : q1), al * (p.color[q1] - p.color[q0]));
        }
    }
}

// ---- gradients (antialias.cu:387-556) ---------------------------------------------------------------

// Atomics are what this kernel costs (a blended pixel pair sends 2C colour and 6 position updates, and a
// scattered f32 atomic is one memory transaction per LANE: 12 per pair for RGB).  The per-pair values are
// therefore staged in LDS and emitted transposed: consecutive lanes take the consecutive channels of one
// pixel, or the x, y, (z,) w of one vertex, so each 12- or 16-byte group reaches memory as one transaction.
struct AAStage { float alpha; int pix0, pix1; int vert[2]; float g[2][3]; };

__global__ __launch_bounds__(256) void k_aa_grad(const AAParams p)
{
#pragma clang fp contract(off)
    __shared__ AAStage s_st[256];
    const int workCount = p.work[0].x;
    const int C = p.channels;
    for (int base = blockIdx.x * 256; base < workCount; base += gridDim.x * 256) {      // uniform trip count: barriers inside
        const int item_idx = base + threadIdx.x;
        AAStage st;
        st.alpha = 0.f; st.pix0 = st.pix1 = 0; st.vert[0] = st.vert[1] = -1;
        st.g[0][0] = st.g[0][1] = st.g[0][2] = st.g[1][0] = st.g[1][1] = st.g[1][2] = 0.f;
        const int4 item = (item_idx < workCount) ? p.work[item_idx + 1] : make_int4(0, 0, 0, 0);
        if (item.w != 0) {                                                // bits of alpha: 0 = no effect
            int px = item.x, py = item.y;
            const int pz = (int)(((unsigned)item.z) >> 16);
            const int d = (item.z >> 2) & 1;
            const float alpha = __int_as_float(item.w);
            const int tri1 = (item.z >> 3) & 1;
            const int di = item.z & 3;
            const size_t pixel0 = (size_t)px + (size_t)p.width * (py + (size_t)p.height * pz);
            const size_t pixel1 = pixel0 + (d ? (size_t)p.width : 1);
            const int tri = float_to_triidx(p.rast[((tri1 ? pixel1 : pixel0) << 2) + 3]) - 1;
            if (tri1) { px += 1 - d; py += d; }
            if (tri >= 0 && tri < p.numTriangles) {
                st.alpha = alpha; st.pix0 = (int)pixel0; st.pix1 = (int)pixel1;      // colour part: emitted transposed below
                // (the silhouette edge's two vertex indices are fetched BEFORE the colour loop, whose loads they do not depend on:
                // behind it they were one more link in the item's chain of dependent loads)
                const int e1 = (di + 1) % 3, e2 = (di + 2) % 3;
                const int ve[2] = {p.tri[3 * tri + e1], p.tri[3 * tri + e2]};
                const float* pDy = p.dy + (alpha > 0.f ? pixel0 : pixel1) * C;
                const float* pColor0 = p.color + pixel0 * C;
                const float* pColor1 = p.color + pixel1 * C;
                float dd = 0.f;
                for (int i = 0; i < C; i++) {
                    const float dy = pDy[i];
                    if (dy != 0.f) dd += dy * (pColor1[i] - pColor0[i]);
                }
                // The blend weight is alpha = +-(1/2 - c) with c = x1 - y1 (x2 - x1) / (y2 - y1): the crossing of
                // the silhouette edge (v1, v2) with the row through the pixel centre, in the frame where the
                // pixel pair is horizontal (:338-365).  dL/dc = -dd; the adjoint runs from c back through the
                // screen-space edge ends to the clip-space vertices (:508-546).  1 / (y2 - y1) is regularised by
                // a signed 1e-3 pixel; saturated blends (|alpha| >= 1/2) carry no position gradient.
                if (dd != 0.f && !(ve[0] < 0 || ve[0] >= p.numVertices || ve[1] < 0 || ve[1] >= p.numVertices)) {
                    const size_t vbase = p.instance ? (size_t)pz * p.numVertices : 0;
                    // axis a = direction of the pixel pair (0: x, 1: y), b = the other one
                    const float half_a = d ? p.yh : p.xh, half_b = d ? p.xh : p.yh;
                    const float fa = (float)(d ? py : px) + .5f - half_a;
                    const float fb = (float)(d ? px : py) + .5f - half_b;
                    float ca[2], cb[2], rw[2], sa[2], sb[2];           // clip-space coords, 1/w, screen-space coords of the two ends
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const float4 v = ((const float4*)p.pos)[vbase + ve[k]];
                        ca[k] = d ? v.y : v.x; cb[k] = d ? v.x : v.y;
                        rw[k] = 1.f / v.w;
                        sa[k] = aa_proj(ca[k], rw[k], half_a, fa);
                        sb[k] = aa_proj(cb[k], rw[k], half_b, fb);
                    }
                    const float da = sa[1] - sa[0], db = sb[1] - sb[0];
                    const float cross = aa_cross(sa[0], db, sb[0], da);
                    const float ib = 1.f / (db + copysignf(1e-3f, db));
                    const float c = cross * ib;
                    // adjoints of the screen-space ends: d c / d sa0 = sb1 / db, d c / d sa1 = -sb0 / db,
                    //                                    d c / d sb0 = (c - sa1) / db, d c / d sb1 = -(c - sa0) / db
                    const float gc = (fabsf(alpha) >= 0.5f) ? 0.f : -dd;
                    const float gsa[2] = {gc * ib * sb[1], -gc * ib * sb[0]};
                    const float gsb[2] = {gc * ib * (c - sa[1]), -gc * ib * (c - sa[0])};
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        // s = clip * (1/w) * half - f
                        const float ga = gsa[k] * rw[k] * half_a, gb = gsb[k] * rw[k] * half_b;
                        const float gw = -(ca[k] * ga + cb[k] * gb) * rw[k];
                        st.vert[k] = (int)(vbase + ve[k]);
                        st.g[k][0] = d ? gb : ga; st.g[k][1] = d ? ga : gb; st.g[k][2] = gw;
                    }
                }
            }
        }
        __syncthreads();                                                   // previous round's readers are done
        s_st[threadIdx.x] = st;
        __syncthreads();
        // colour: lane -> (pair, side, channel); the reference skips exact-zero upstream values (:449-462)
        const int twoC = 2 * C;
        for (int f = threadIdx.x; f < 256 * twoC; f += 256) {
            const int j = f / twoC, r = f - j * twoC;
            const int side = r >= C ? 1 : 0, c = r - side * C;
            const float alpha = s_st[j].alpha;
            if (alpha == 0.f) continue;
            const int p0 = s_st[j].pix0, p1 = s_st[j].pix1;
            const float dy = p.dy[(size_t)(alpha > 0.f ? p0 : p1) * C + c];
            if (dy != 0.f) atomic_add_f32(p.gradColor + (size_t)(side ? p1 : p0) * C + c, side ? alpha * dy : -(alpha * dy));
        }
        // positions: lane -> (pair, edge end, x|y|z|w); z carries nothing
        for (int f = threadIdx.x; f < 256 * 8; f += 256) {
            const int j = f >> 3, k = (f >> 2) & 1, comp = f & 3;
            const int v = s_st[j].vert[k];
            if (v < 0 || comp == 2) continue;
            atomic_add_f32(p.gradPos + 4 * (size_t)v + comp, s_st[j].g[k][comp == 3 ? 2 : comp]);
        }
    }
}

static int alloc_triangles(int T) { int a = 64; while (a < T) a <<= 1; return a; }           // torch_antialias.cpp:43-45
static int hash_elems_per_tri(int alloc) { return alloc >= (2 << 25) ? 4 : 8; }              // antialias.h:19

static int fill_aa(AAParams& p, const char* who, const float* color, const float* rast, const float* pos, const int32_t* tri,
                   int instance_mode, int N, int V, int T, int H, int W, int C)
{
    NVDR_REQUIRE(color && rast && pos && tri, "%s: null pointer", who);
    NVDR_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, "color must have shape[>0, >0, >0, >0]");
    NVDR_REQUIRE(T > 0, "tri must have shape [>0, 3]");
    NVDR_REQUIRE(V > 0, "pos must have shape [>0, >0, 4] or [>0, 4]");
    NVDR_REQUIRE(N < 65536, "%s: minibatch too large for the work-item encoding (16 bits)", who);
    NVDR_REQUIRE((long long)N * H * W * 2 + 1 < (1ll << 31), "%s: too many pixels for 32-bit work-item indices", who);
    NVDR_REQUIRE(!((uintptr_t)pos & 15), "pos input tensor not aligned to float4");
    NVDR_REQUIRE(!((uintptr_t)rast & 7), "raster_out input tensor not aligned to float2");
    p = AAParams{};
    p.color = color; p.rast = rast; p.pos = pos; p.tri = tri;
    p.numTriangles = T; p.numVertices = V; p.width = W; p.height = H; p.n = N; p.channels = C;
    p.instance = instance_mode ? 1 : 0;
    p.xh = .5f * (float)W; p.yh = .5f * (float)H;
    return NVDR_OK;
}

static int item_grid(long long max_items)
{
    long long blocks = (max_items + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;                   // grid-stride beyond 8 workgroups per CU
    return (int)(blocks < 1 ? 1 : blocks);
}

}  // namespace nvdr

using namespace nvdr;

extern "C" size_t nvdr_antialias_hash_bytes(int T)
{
    if (T <= 0) return 0;
    const int alloc = alloc_triangles(T);
    return (size_t)alloc * hash_elems_per_tri(alloc) * 16;
}

extern "C" size_t nvdr_antialias_work_bytes(int N, int H, int W)
{
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    return ((size_t)N * H * W * 8 + 4) * 4;                   // torch_antialias.cpp:123: two 16-byte items per pixel + counters
}

extern "C" int nvdr_antialias_construct_topology_hash(const int32_t* tri, int T, void* hash, size_t hash_bytes, nvdrStream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    NVDR_REQUIRE(tri && hash, "antialias_construct_topology_hash: null pointer");
    NVDR_REQUIRE(T > 0, "tri must have shape [>0, 3]");
    NVDR_REQUIRE(!((uintptr_t)hash & 15), "ev_hash internal tensor not aligned to int4");
    const size_t need = nvdr_antialias_hash_bytes(T);
    if (hash_bytes < need) { set_error("antialias_construct_topology_hash: hash buffer too small (%zu < %zu)", hash_bytes, need); return NVDR_ERR_SCRATCH; }
    AAParams p{};
    p.tri = tri; p.numTriangles = T; p.numVertices = 0x7fffffff;
    p.hash = (uint4*)hash;
    p.hashMask = (unsigned)(need / 16 - 1);
    NVDR_HIP_CHECK(hipMemsetAsync(hash, 0, need, stream));
    {
        ProfileScope ps("aa_mesh", stream);
        hipLaunchKernelGGL(k_aa_mesh, dim3((T + 255) / 256), dim3(256), 0, stream, p);
    }
    NVDR_LAUNCH_CHECK();
    return NVDR_OK;
}

extern "C" int nvdr_antialias_fwd(const float* color, const float* rast, const float* pos, const int32_t* tri,
                                  const void* hash, size_t hash_bytes,
                                  int instance_mode, int N, int V, int T, int H, int W, int C,
                                  float* out, void* work, size_t work_bytes, const uint8_t* tile_flags, nvdrStream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    AAParams p;
    int rc = fill_aa(p, "antialias_fwd", color, rast, pos, tri, instance_mode, N, V, T, H, W, C);
    if (rc) return rc;
    p.flags = tile_flags_view((debug_flags() & 33554432) ? nullptr : tile_flags, N, H, W, !(debug_flags() & 134217728));
    NVDR_REQUIRE(hash && out && work, "antialias_fwd: null pointer");
    NVDR_REQUIRE(!((uintptr_t)work & 15), "work_buffer internal tensor not aligned to int4");
    NVDR_REQUIRE(!((uintptr_t)hash & 15), "topology_hash internal tensor not aligned to int4");
    const size_t need_hash = nvdr_antialias_hash_bytes(T);
    if (hash_bytes < need_hash) { set_error("antialias_fwd: topology hash was built for fewer triangles (%zu < %zu bytes)", hash_bytes, need_hash); return NVDR_ERR_ARG; }
    if (work_bytes < nvdr_antialias_work_bytes(N, H, W)) { set_error("antialias_fwd: work buffer too small"); return NVDR_ERR_SCRATCH; }
    p.hash = (uint4*)hash;
    p.hashMask = (unsigned)(hash_bytes / 16 - 1);
    NVDR_REQUIRE(((hash_bytes / 16) & (hash_bytes / 16 - 1)) == 0, "antialias_fwd: topology hash size is not a power of two");
    p.output = out; p.work = (int4*)work;
    const size_t P = (size_t)N * H * W;
    // counters = 0; out = color (torch_antialias.cpp:122 clones) is done by k_aa_discontinuity on the way.
    NVDR_HIP_CHECK(hipMemsetAsync(work, 0, 16, stream));
    {
        ProfileScope ps("aa_discontinuity", stream);
        const int gx = (W + kAaBlockW - 1) / kAaBlockW, gy = (H + kAaBlockH - 1) / kAaBlockH;
        const long long blocks = p.flags.order ? tile_flags_ordered_grid(p.flags, (64 / kAaBlockW) * (64 / kAaBlockH)) : (long long)gx * gy * N;   // (nvdr_device.hpp TileFlags)
        hipLaunchKernelGGL(k_aa_discontinuity, dim3((unsigned)(((blocks + 7) / 8) * 8)), dim3(256), 0, stream, p, gx, gy);
    }
    NVDR_LAUNCH_CHECK();
    {
        ProfileScope ps("aa_analysis", stream);
        hipLaunchKernelGGL(k_aa_analysis, dim3(item_grid((long long)P * 2)), dim3(256), 0, stream, p);
    }
    NVDR_LAUNCH_CHECK();
    return NVDR_OK;
}

extern "C" int nvdr_antialias_grad(const float* color, const float* rast, const float* pos, const int32_t* tri,
                                   const float* dy, const void* work, size_t work_bytes,
                                   int instance_mode, int N, int V, int T, int H, int W, int C,
                                   float* g_color, float* g_pos, nvdrStream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    AAParams p;
    int rc = fill_aa(p, "antialias_grad", color, rast, pos, tri, instance_mode, N, V, T, H, W, C);
    if (rc) return rc;
    NVDR_REQUIRE(dy && work && g_color && g_pos, "antialias_grad: null pointer");
    NVDR_REQUIRE(!((uintptr_t)work & 15), "work_buffer internal tensor not aligned to int4");
    if (work_bytes < nvdr_antialias_work_bytes(N, H, W)) { set_error("antialias_grad: work buffer too small"); return NVDR_ERR_SCRATCH; }
    p.dy = dy; p.work = (int4*)work; p.gradColor = g_color; p.gradPos = g_pos;
    const size_t P = (size_t)N * H * W;
    // g_color = dy (torch_antialias.cpp:218 clones); g_pos is zero-filled by the caller.  (A copy in front of the sparse atomics of
    // k_aa_grad: they may land on any pixel, so no single launch can order "copy this pixel" before "add to it" across workgroups.)
    {
        ProfileScope ps("aa_copy_bwd", stream);
        NVDR_HIP_CHECK(hipMemcpyAsync(g_color, dy, P * C * sizeof(float), hipMemcpyDeviceToDevice, stream));
    }
    {
        ProfileScope ps("aa_grad", stream);
        hipLaunchKernelGGL(k_aa_grad, dim3(item_grid((long long)P * 2)), dim3(256), 0, stream, p);
    }
    NVDR_LAUNCH_CHECK();
    return NVDR_OK;
}
