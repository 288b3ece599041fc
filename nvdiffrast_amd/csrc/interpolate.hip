// interpolate.hip -- attribute interpolation forward / backward for gfx950.
//
// Replaces csrc/common/interpolate.cu + csrc/torch/torch_interpolate.cpp behind the C ABI.
//  k_interp_fwd   pure streaming: pixels are taken in linear order (one lane = one pixel,
//                 a wave = 64 consecutive pixels = 1 KiB of rast per load instruction);
//                 the three vertex attribute rows are gathered from L2.
//  k_interp_grad  one workgroup per 64x16 pixel block streaming whole rows; attribute gradients
//                 are summed over triangle runs in the wave, accumulated per vertex in an LDS
//                 fixed-point hash table (ds_add_u64) and flushed as one hardware f32 atomic per
//                 (vertex, attribute) per block.
#include "nvdr_device.hpp"
#include "nvdr_host.hpp"
#include <type_traits>

namespace nvdr {

constexpr int kMaxDiffAttrs = 32;       // interpolate.h:18 IP_MAX_DIFF_ATTRS

struct InterpParams {
    const int* tri; const float* attr; const float* rast; const float* rastDB;
    const float* dy; const float* dda;
    float* out; float* outDA; float* gradAttr; float* gradRaster; float* gradRasterDB;
    int numTriangles, numVertices, numAttr, numDiffAttr;
    int width, height, depth;
    int attrBC, instance_mode, diff_attrs_all, dbg;
    int streamOut;          // forward: the output is too large to stay in the Infinity Cache anyway -> non-temporal stores
    int widthShift;         // log2(width) when the width is a power of two, else -1 (forward: pixel row without a division)
    int daVec4;             // forward: A = 2 with diff_attrs = 'all' and a 16-byte aligned out_da: one float4 store per pixel
    int streamDA;           // ... written around the cache when it is larger than most of it (config 3: 0.224 -> 0.218 ms, the step 2.876 -> 2.866)
    TileFlags flags;        // which 8x8 tiles of rast show a triangle at all (nvdr_device.hpp), or f == nullptr
    int ordered;            // k_interp_fwd walks the work order behind the flags
    int diffAttrs[kMaxDiffAttrs];
};

__device__ __forceinline__ int diff_index(const InterpParams& p, int i)
{
    int j = p.diff_attrs_all ? i : p.diffAttrs[i];
    if (j < 0) j += p.numAttr;                              // python-style (interpolate.cu:102-103)
    return (j >= 0 && j < p.numAttr) ? j : -1;
}

// ---- forward (interpolate.cu:15-126) ---------------------------------------------------

// pixels per thread of k_interp_fwd: four where the pixel differentials are written as well (config 3: 0.285 -> 0.240 ms; eight:
// 0.274), one otherwise (headline batch: 82 us; two: 92; four: 85-93; eight: 99; workgroups of 128 threads: 100)
constexpr int ip_fwd_pixels(bool enable_da) { return enable_da ? 4 : 1; }

template <int A_CT, bool ENABLE_DA>
__global__ __launch_bounds__(256) void k_interp_fwd(const InterpParams p)
{
    // grid = (blocks per image, images in chunks of 32768): the image index comes from the block index; a 64-bit
    // pidx / HW per lane was more than half of this kernel's instructions.
    const unsigned HW = (unsigned)p.width * (unsigned)p.height;
    constexpr int kPixels = ip_fwd_pixels(ENABLE_DA);
    // With a work order behind the flags (nvdr_device.hpp TileFlags; p.ordered) a workgroup is a 64 x 4*kPixels pixel block of
    // the order's bins -- covered bins first, then the ones that are only zeros to store -- instead of a run of the image.
    int pz, obx = 0, oby = 0;
    if (p.ordered) {
        // (the partner clearing of k_interp_fwd_cols was tried here as well: 0.225 vs 0.227 ms at config 3 -- this variant is bound by
        // its 805 MB of stores either way)
        if (!decode_block_ordered(p.flags, (p.width + 63) >> 6, (p.height + 4 * kPixels - 1) / (4 * kPixels), 64, 4 * kPixels, obx, oby, pz)) return;
    } else {
        pz = (int)(blockIdx.y + blockIdx.z * 32768u);
        if (pz >= p.depth) return;
    }
    const int A = A_CT > 0 ? A_CT : p.numAttr;
    // several pixels per thread, 256 apart: fewer workgroups to start, and one thread's chains of dependent loads
    // (flag -> rast -> triangle -> vertices) overlap
#pragma unroll
    for (int kk = 0; kk < kPixels; kk++) {
    unsigned inImage = (blockIdx.x * (unsigned)kPixels + (unsigned)kk) * 256u + threadIdx.x;
    if (p.ordered) {
        const int x = obx * 64 + (int)(threadIdx.x & 63), y = (oby * kPixels + kk) * 4 + (int)(threadIdx.x >> 6);
        if (x >= p.width || y >= p.height) continue;
        inImage = (unsigned)y * (unsigned)p.width + (unsigned)x;
    }
    if (inImage >= HW) continue;
    const size_t pidx = (size_t)pz * HW + inImage;

    // A tile that rasterize() found empty: zeros without reading rast (/ rast_db): two thirds of the benchmark's tiles.
    bool known_empty = false;
    if (p.flags.f) {
        const unsigned py = p.widthShift >= 0 ? (inImage >> p.widthShift) : inImage / (unsigned)p.width;
        known_empty = p.flags.empty(pz, (int)py, (int)(inImage - py * (unsigned)p.width));
    }
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!known_empty) r = ((const float4*)p.rast)[pidx];
    int triIdx = float_to_triidx(r.w) - 1;
    bool valid = (triIdx >= 0 && triIdx < p.numTriangles);
    int vi0 = 0, vi1 = 0, vi2 = 0;
    if (valid) {
        vi0 = p.tri[triIdx * 3 + 0]; vi1 = p.tri[triIdx * 3 + 1]; vi2 = p.tri[triIdx * 3 + 2];
        if (!indices_ok(vi0, vi1, vi2, p.numVertices))
            continue;                                       // corrupt indices: leave untouched (:54-58)
    }
    float* out = p.out + pidx * A;
    float2* outDA = ENABLE_DA ? ((float2*)p.outDA) + pidx * p.numDiffAttr : nullptr;

    // The usual texture-coordinate case -- two attributes, both differentiated -- writes its four differentials as one
    // 16-byte store (two 8-byte stores at a 16-byte stride reach memory as partial lines) and takes the attribute
    // differences from the rows that are already in registers.
    const bool da4 = ENABLE_DA && A_CT == 2 && p.daVec4;
    if (!valid) {
        // No triangle: zeros (the reference reaches the same values via zero barycentrics, :73-80).
        if (A_CT == 4)      { if (p.streamOut) store_streaming((float4*)out, make_float4(0.f, 0.f, 0.f, 0.f)); else *(float4*)out = make_float4(0.f, 0.f, 0.f, 0.f); }
        else if (A_CT == 2) *(float2*)out = make_float2(0.f, 0.f);
        else for (int i = 0; i < A; i++) out[i] = 0.f;
        if (da4) { if (p.streamDA) store_streaming((float4*)outDA, make_float4(0.f, 0.f, 0.f, 0.f)); else *(float4*)outDA = make_float4(0.f, 0.f, 0.f, 0.f); }
        else if (ENABLE_DA) for (int i = 0; i < p.numDiffAttr; i++) outDA[i] = make_float2(0.f, 0.f);
        continue;
    }
    if (p.instance_mode && !p.attrBC) { vi0 += pz * p.numVertices; vi1 += pz * p.numVertices; vi2 += pz * p.numVertices; }
    const float* a0 = p.attr + (size_t)vi0 * A;
    const float* a1 = p.attr + (size_t)vi1 * A;
    const float* a2 = p.attr + (size_t)vi2 * A;
    float b0 = r.x, b1 = r.y, b2 = 1.f - r.x - r.y;

    if (A_CT == 4) {
        float4 x0 = *(const float4*)a0, x1 = *(const float4*)a1, x2 = *(const float4*)a2;
        const float4 o = make_float4(b0 * x0.x + b1 * x1.x + b2 * x2.x, b0 * x0.y + b1 * x1.y + b2 * x2.y,
                                     b0 * x0.z + b1 * x1.z + b2 * x2.z, b0 * x0.w + b1 * x1.w + b2 * x2.w);
        if (p.streamOut) store_streaming((float4*)out, o); else *(float4*)out = o;
    } else if (A_CT == 2) {
        float2 x0 = *(const float2*)a0, x1 = *(const float2*)a1, x2 = *(const float2*)a2;
        *(float2*)out = make_float2(b0 * x0.x + b1 * x1.x + b2 * x2.x, b0 * x0.y + b1 * x1.y + b2 * x2.y);
        if (da4) {
            const float4 db = ((const float4*)p.rastDB)[pidx];
            const float du0 = x0.x - x2.x, dv0 = x1.x - x2.x, du1 = x0.y - x2.y, dv1 = x1.y - x2.y;
            const float4 dav = make_float4(db.x * du0 + db.z * dv0, db.y * du0 + db.w * dv0, db.x * du1 + db.z * dv1, db.y * du1 + db.w * dv1);
            if (p.streamDA) store_streaming((float4*)outDA, dav); else *(float4*)outDA = dav;
            continue;
        }
    } else {
        for (int i = 0; i < A; i++) out[i] = b0 * a0[i] + b1 * a1[i] + b2 * a2[i];
    }
    if (!ENABLE_DA) continue;

    float4 db = ((const float4*)p.rastDB)[pidx];
    for (int i = 0; i < p.numDiffAttr; i++) {
        int j = diff_index(p, i);
        float dsdx = 0.f, dsdy = 0.f;
        if (j >= 0) {
            float dsdu = a0[j] - a2[j], dsdv = a1[j] - a2[j];
            dsdx = db.x * dsdu + db.z * dsdv;
            dsdy = db.y * dsdu + db.w * dsdv;
        }
        outDA[i] = make_float2(dsdx, dsdy);
    }
    }
}

// The forward pass without pixel differentials, for vector-sized attribute rows (A = 4 or 2).  A pixel is a chain of dependent
// loads -- flag -> rast -> triangle -> three vertices -- and the launch is bound by how many of those chains are in flight, not
// by bytes (one pixel per thread: 83 us for 340 MB at the headline batch).  Here a thread owns K pixels of one column and walks
// the chain in stages, K loads wide: K rast loads, then K index triples, then 3K attribute rows, then K stores.  No branches in
// between (a pixel without a triangle, or beyond the image, reads row 0 of the respective table instead and its result is
// discarded), so that every stage's loads are issued back to back.
// Workgroup = 64 x 4K pixels, wave w = rows [wK, wK + K): one 8-pixel tile row (K divides 8), so ONE occupancy flag per lane.
template <int A_CT, int K>
__global__ __launch_bounds__(256) void k_interp_fwd_cols(const InterpParams p, int gx, int gy)
{
    static_assert(A_CT == 4 || A_CT == 2, "vector rows only");
    static_assert(K == 1 || K == 2 || K == 4 || K == 8, "K rows of one tile row");
    typedef typename std::conditional<A_CT == 4, float4, float2>::type Row;
    int bx, by, pz;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (p.ordered) {
        // Along the work order a covered bin's workgroup also stores the zeros of one EMPTY bin of its XCD's share (the same block
        // of it), and that bin's own workgroup leaves at once: the launch used to be a latency-bound phase over the covered bins
        // followed by a store-bound one over the empty bins (35 + 30 us at the headline batch); interleaved, the stores go out
        // while the other waves of a CU wait for their loads.
        constexpr int per = 16 / K;                                  // workgroups per bin (64 x 4K pixels each)
        const TileFlags& t = p.flags;
        const int xcd = (int)(blockIdx.x & 7), j = (int)(blockIdx.x >> 3);
        const int slot = j / per, sub = j - slot * per;
        int own, partner; bool skip;
        ordered_list_pair(t.nBins, t.order[t.nBins], xcd, slot, own, partner, skip);
        if (own < 0 || skip) return;
        auto place = [&](int idx, int& bx_, int& by_, int& pz_) {
            const int bin = __builtin_amdgcn_readfirstlane(t.order[idx]);
            pz_ = bin / (t.binsX * t.binsY);
            const int rem = bin - pz_ * (t.binsX * t.binsY);
            const int binY = rem / t.binsX;
            bx_ = rem - binY * t.binsX; by_ = binY * per + sub;
        };
        if (partner >= 0) {
            int qx, qy, qz;
            place(partner, qx, qy, qz);
            const int xq = qx * 64 + lane, yq = qy * (4 * K) + wave * K;
            if (xq < p.width) {
                Row* o = (Row*)p.out + ((size_t)qz * p.height + yq) * p.width + xq;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    if (yq + k >= p.height) break;
                    if constexpr (A_CT == 4) { if (p.streamOut) store_streaming(o + (size_t)k * p.width, make_float4(0.f, 0.f, 0.f, 0.f)); else o[(size_t)k * p.width] = make_float4(0.f, 0.f, 0.f, 0.f); }
                    else o[(size_t)k * p.width] = make_float2(0.f, 0.f);
                }
            }
        }
        place(own, bx, by, pz);
        if (bx >= gx || by >= gy) return;
    } else if (!decode_block(gx, gy, p.depth, bx, by, pz)) return;
    const int x = bx * 64 + lane, y0 = by * (4 * K) + wave * K;
    if (x >= p.width || y0 >= p.height) return;
    const bool empty = p.flags.empty(pz, y0, x);
    const size_t pix0 = ((size_t)pz * p.height + y0) * p.width + x;
    const size_t voff = (p.instance_mode && !p.attrBC) ? (size_t)pz * p.numVertices : 0;
    const Row* rows = (const Row*)p.attr + voff;

    float4 r[K];
#pragma unroll
    for (int k = 0; k < K; k++)
        r[k] = ((const float4*)p.rast)[(!empty && y0 + k < p.height) ? pix0 + (size_t)k * p.width : 0];
    int vi[K][3];
    bool valid[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int t = float_to_triidx(r[k].w) - 1;
        valid[k] = !empty && y0 + k < p.height && t >= 0 && t < p.numTriangles;
        const int* tp = p.tri + (valid[k] ? t : 0) * 3;
        vi[k][0] = tp[0]; vi[k][1] = tp[1]; vi[k][2] = tp[2];
    }
    Row a[K][3];
    bool keep[K];                                           // false: corrupt indices, the pixel is left untouched (interpolate.cu:54-58)
#pragma unroll
    for (int k = 0; k < K; k++) {
        const bool ok = indices_ok(vi[k][0], vi[k][1], vi[k][2], p.numVertices);
        keep[k] = !valid[k] || ok;
        valid[k] = valid[k] && ok;
#pragma unroll
        for (int j = 0; j < 3; j++) a[k][j] = rows[valid[k] ? vi[k][j] : 0];
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        if (y0 + k >= p.height || !keep[k]) continue;
        const float b0 = valid[k] ? r[k].x : 0.f, b1 = valid[k] ? r[k].y : 0.f, b2 = valid[k] ? 1.f - r[k].x - r[k].y : 0.f;
        Row* out = (Row*)p.out + pix0 + (size_t)k * p.width;
        if constexpr (A_CT == 4) {
            const float4 o = make_float4(b0 * a[k][0].x + b1 * a[k][1].x + b2 * a[k][2].x, b0 * a[k][0].y + b1 * a[k][1].y + b2 * a[k][2].y,
                                         b0 * a[k][0].z + b1 * a[k][1].z + b2 * a[k][2].z, b0 * a[k][0].w + b1 * a[k][1].w + b2 * a[k][2].w);
            if (p.streamOut) store_streaming(out, o); else *out = o;
        } else {
            *out = make_float2(b0 * a[k][0].x + b1 * a[k][1].x + b2 * a[k][2].x, b0 * a[k][0].y + b1 * a[k][1].y + b2 * a[k][2].y);
        }
    }
}

// ---- backward (interpolate.cu:131-274) -------------------------------------------------

// Workgroup = 64 x 16 pixel block of one image, 4 waves, each wave owning four 64-pixel rows
// (1 KiB coalesced accesses).  Small workgroups on purpose: the two phases meet at a barrier, and
// eight 4-wave groups per CU overlap one group's memory phase with another's LDS phase better than
// four 8-wave groups did (0.149 -> 0.144 ms; 2-wave groups lose again to the extra flushes).
// Two phases around one barrier:
//   A  per pixel: gradients w.r.t. the barycentrics (and their pixel differentials) are written
//      out; the largest attribute-gradient contribution of the block is published (it fixes the
//      fixed-point scale of the LDS accumulator, nvdr_device.hpp);
//   B  the contributions b_k * dy_i are summed over runs of equal triangle id inside the wave
//      (RunScan), the run totals go to the workgroup's LDS vertex table with ds_add_u64, and
//      every touched (vertex, attribute) reaches memory as ONE global atomic per block.
// The reference issues 3*A atomics per pixel behind a __match_any_sync coalescer
// (interpolate.cu:198-210, common.h:205-216).  Table size comes from the host (dynamic LDS).
constexpr int kIpBlockW = 64;
constexpr int kIpBlockH = 16;
constexpr int kIpThreads = 256;
constexpr int kIpRowsPerWave = 4;

struct IpPixel { int tri; float b0, b1; };

template <int A_CT, bool ENABLE_DA>
__global__ __launch_bounds__(kIpThreads, (ENABLE_DA && A_CT > 0) ? 6 : 8) void k_interp_grad(const InterpParams p, int slots, int gx, int gy)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_mem[];
    constexpr bool kRegs = (A_CT == 4 || A_CT == 2);        // upstream gradient stays in registers between the phases
    const int A = A_CT > 0 ? A_CT : p.numAttr;
    unsigned long long* s_vals = (unsigned long long*)s_mem;
    uint32_t* s_keys = (uint32_t*)(s_mem + (size_t)slots * A * 8);
    uint32_t* s_max = s_keys + slots;                       // [0] block max, [1] number of used slots
    uint16_t* s_list = (uint16_t*)(s_max + 4);              // [slots] used slots (flush)
    int bx, by, pz;
    if (p.flags.order ? !decode_block_ordered(p.flags, gx, gy, kIpBlockW, kIpBlockH, bx, by, pz)
                      : !decode_block(gx, gy, p.depth, bx, by, pz)) return;
    VertexTable tab{s_keys, s_vals, slots, A};
    tab.clear(threadIdx.x, kIpThreads);
    if (threadIdx.x == 0) { s_max[0] = 0u; s_max[1] = 0u; }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = bx * kIpBlockW + lane;
    const int row0 = by * kIpBlockH + wave * kIpRowsPerWave;
    const size_t voff = (p.instance_mode && !p.attrBC) ? (size_t)pz * p.numVertices : 0;
    const float* attr = p.attr + voff * A;
    float* gattr = p.gradAttr + voff * A;

    IpPixel q[kIpRowsPerWave];
    bool ok[kIpRowsPerWave];
    float4 yreg[kIpRowsPerWave];
    uint32_t um = 0u;                                       // the lane's largest contribution, as magnitude bits (nvdr_device.hpp mag_bits)

    // ---- phase A -----------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < kIpRowsPerWave; r++) {
        const int py = row0 + r;
        ok[r] = false;
        yreg[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (py >= p.height || px >= p.width) continue;
        const size_t pidx = ((size_t)pz * p.height + py) * p.width + px;
        float4 rr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!p.flags.empty(pz, py, px)) rr = ((const float4*)p.rast)[pidx];     // (an empty tile's rast is not read)
        const int triIdx = float_to_triidx(rr.w) - 1;
        if (triIdx < 0 || triIdx >= p.numTriangles) {
            ((float4*)p.gradRaster)[pidx] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ENABLE_DA) ((float4*)p.gradRasterDB)[pidx] = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        const int vi0 = p.tri[triIdx * 3 + 0], vi1 = p.tri[triIdx * 3 + 1], vi2 = p.tri[triIdx * 3 + 2];
        if (!indices_ok(vi0, vi1, vi2, p.numVertices))
            continue;                                       // corrupt indices: leave untouched (:163-167)
        ok[r] = true;
        q[r].tri = triIdx; q[r].b0 = rr.x; q[r].b1 = rr.y;
        const float* a0 = attr + (size_t)vi0 * A;
        const float* a1 = attr + (size_t)vi1 * A;
        const float* a2 = attr + (size_t)vi2 * A;
        const float* pdy = p.dy + pidx * A;
        const float bmax = fmaxf(fmaxf(fabsf(rr.x), fabsf(rr.y)), fabsf(1.f - rr.x - rr.y));

        float gb0 = 0.f, gb1 = 0.f, ymax = 0.f;
        if (A_CT == 4) {
            const float4 y = load_streaming((const float4*)pdy);      // read once per step: keep it out of the Infinity Cache
            const float4 x0 = *(const float4*)a0, x1 = *(const float4*)a1, x2 = *(const float4*)a2;
            gb0 = dot_diff(y, x0, x2);
            gb1 = dot_diff(y, x1, x2);
            ymax = __int_as_float((int)max(max(mag_bits(y.x), mag_bits(y.y)), max(mag_bits(y.z), mag_bits(y.w))));
            yreg[r] = y;
        } else if (A_CT == 2) {                             // the usual texture-coordinate case
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
        ((float4*)p.gradRaster)[pidx] = make_float4(gb0, gb1, 0.f, 0.f);
        um = max(um, mag_bits(ymax * bmax));                // >= every |b_k * dy_i| (rounding is monotone); as magnitude bits (mag_bits)

        if (ENABLE_DA) {
            const float4 db = ((const float4*)p.rastDB)[pidx];
            const float2* dda = ((const float2*)p.dda) + pidx * p.numDiffAttr;
            float gdudx = 0.f, gdudy = 0.f, gdvdx = 0.f, gdvdy = 0.f;
            for (int i = 0; i < p.numDiffAttr; i++) {
                const int j = diff_index(p, i);
                if (j < 0) continue;
                const float2 d = dda[i];
                const float dsdu = a0[j] - a2[j], dsdv = a1[j] - a2[j];
                gdudx += dsdu * d.x; gdudy += dsdu * d.y;
                gdvdx += dsdv * d.x; gdvdy += dsdv * d.y;
                const float du = d.x * db.x + d.y * db.y;
                const float dv = d.x * db.z + d.y * db.w;
                um = max(max(um, mag_bits(du)), max(mag_bits(dv), mag_bits(-du - dv)));
            }
            ((float4*)p.gradRasterDB)[pidx] = make_float4(gdudx, gdudy, gdvdx, gdvdy);
        }
    }
    block_max_update(s_max, __int_as_float((int)um));
    __syncthreads();
    const uint32_t maxBits = *s_max;
    if (maxBits == 0u || (p.dbg & 1)) return;               // no contribution anywhere in the block
    const bool direct = slots == 0 || maxBits >= 0x7F800000u;   // no table (very wide vertices) / inf or NaN present: plain f32 atomics
    const FixedScale fs(direct ? 0x3F800000u : maxBits);

    // ---- phase B -----------------------------------------------------------------------
    // The vertex indices are fetched again here (L1/L2 hits) instead of being carried in twelve
    // registers across phase A's peak: that is what lets the kernel run at 8 waves per SIMD.
    int vi[kIpRowsPerWave][3];
#pragma unroll
    for (int r = 0; r < kIpRowsPerWave; r++) {
        const int t = ok[r] ? q[r].tri : 0;
#pragma unroll
        for (int k = 0; k < 3; k++) vi[r][k] = p.tri[t * 3 + k];
    }
#pragma unroll
    for (int r = 0; r < kIpRowsPerWave; r++) {
        if (__ballot(ok[r]) == 0) continue;
        const RunScan rs(q[r].tri, ok[r]);
        const bool emit = direct ? ok[r] : rs.tail;
        int s0 = -1, s1 = -1, s2 = -1;
        if (emit && !direct) tab.find3(vi[r][0], vi[r][1], vi[r][2], s0, s1, s2);
        // Lanes whose three vertices all have slots (the rule) add to the table under ONE test per value triple;
        // the rest (table full, inf/NaN mode) take the per-vertex path under a wave-uniform test.
        const bool tabled = emit && (s0 | s1 | s2) >= 0;
        const bool anyLoose = __ballot(emit && !tabled) != 0ull;
        auto put3 = [&](int i, float v0, float v1, float v2) {
            if (tabled) { tab.add(s0, i, fs.to_fixed(v0)); tab.add(s1, i, fs.to_fixed(v1)); tab.add(s2, i, fs.to_fixed(v2)); }
            if (anyLoose && emit && !tabled) {
                const int sl[3] = {s0, s1, s2}; const float vv[3] = {v0, v1, v2};
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    if (sl[k] >= 0) tab.add(sl[k], i, fs.to_fixed(vv[k]));
                    else atomic_add_f32(gattr + (size_t)vi[r][k] * A + i, vv[k]);
                }
            }
        };
        const float b0 = ok[r] ? q[r].b0 : 0.f, b1 = ok[r] ? q[r].b1 : 0.f, b2 = ok[r] ? 1.f - q[r].b0 - q[r].b1 : 0.f;
        const size_t pidx = ((size_t)pz * p.height + (row0 + r)) * p.width + px;
        const float* pdy = p.dy + pidx * A;
        for (int i = 0; i < A; i++) {
            float y;
            if (kRegs) y = i == 0 ? yreg[r].x : i == 1 ? yreg[r].y : i == 2 ? yreg[r].z : yreg[r].w;
            else       y = ok[r] ? pdy[i] : 0.f;
            float v0 = b0 * y, v1 = b1 * y, v2 = b2 * y;
            if (!direct) rs.scan3(v0, v1, v2);
            put3(i, v0, v1, v2);
        }
        if (ENABLE_DA) {
            float4 db = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok[r]) db = ((const float4*)p.rastDB)[pidx];
            const float2* dda = ((const float2*)p.dda) + pidx * p.numDiffAttr;
            for (int i = 0; i < p.numDiffAttr; i++) {
                const int j = diff_index(p, i);
                if (j < 0) continue;
                float2 d = make_float2(0.f, 0.f);
                if (ok[r]) d = dda[i];
                float du = d.x * db.x + d.y * db.y;
                float dv = d.x * db.z + d.y * db.w;
                float dw = -du - dv;
                if (!direct) rs.scan3(du, dv, dw);
                put3(j, du, dv, dw);
            }
        }
    }
    if (direct) return;

    // Flush: one atomic per (vertex, attribute) this block touched.
    __syncthreads();
    const int n = tab.compact(s_list, &s_max[1], threadIdx.x, kIpThreads) * A;
    for (int i = threadIdx.x; i < n; i += kIpThreads) {
        const int u = i / A, c = i - u * A;
        const int slot = s_list[u];
        const unsigned long long t = s_vals[slot * A + c];
        if (t) atomic_add_f32(gattr + (size_t)(s_keys[slot] - 1u) * A + c, fs.to_float(t));
    }
}

static int fill_params(InterpParams& p, const float* attr, const float* rast, const int32_t* tri, const float* rast_db,
                       int attr_instance, int attr_n, int N, int V, int A, int T, int H, int W,
                       int diff_all, const int32_t* diff_attrs_host, int num_diff, bool enable_da, const char* who,
                       const uint8_t* tile_flags)
{
    NVDR_REQUIRE(attr && rast && tri, "%s: null pointer", who);
    NVDR_REQUIRE(N > 0 && H > 0 && W > 0, "rast must have shape[>0, >0, >0, 4]");
    NVDR_REQUIRE(T > 0, "tri must have shape [>0, 3]");
    NVDR_REQUIRE(V > 0 && A > 0, "attr must have shape [>0, >0, >0] or [>0, >0]");
    if (attr_instance) NVDR_REQUIRE(attr_n == N || attr_n == 1, "minibatch size mismatch between inputs rast, attr");
    NVDR_REQUIRE(!((uintptr_t)rast & 15), "rast input tensor not aligned to float4");
    NVDR_REQUIRE(!((uintptr_t)rast_db & 15), "rast_db input tensor not aligned to float4");
    p = InterpParams{};
    p.attr = attr; p.rast = rast; p.tri = tri; p.rastDB = enable_da ? rast_db : nullptr;
    p.numTriangles = T; p.numVertices = V; p.numAttr = A;
    p.width = W; p.height = H; p.depth = N;
    p.instance_mode = attr_instance ? 1 : 0;
    p.attrBC = (attr_instance && attr_n == 1 && N > 1) ? 1 : 0;
    if (attr_instance && attr_n == 1) p.attrBC = 1;
    p.numDiffAttr = 0;
    p.dbg = debug_flags();
    p.flags = tile_flags_view((p.dbg & 33554432) ? nullptr : tile_flags, N, H, W, !(p.dbg & 134217728));
    p.widthShift = (W & (W - 1)) == 0 ? __builtin_ctz((unsigned)W) : -1;
    if (enable_da) {
        if (diff_all) { p.numDiffAttr = A; p.diff_attrs_all = 1; }
        else {
            NVDR_REQUIRE(num_diff <= kMaxDiffAttrs, "too many entries in diff_attrs list (increase IP_MAX_DIFF_ATTRS)");
            NVDR_REQUIRE(diff_attrs_host, "%s: diff_attrs list missing", who);
            p.numDiffAttr = num_diff;
            for (int i = 0; i < num_diff; i++) p.diffAttrs[i] = diff_attrs_host[i];
        }
    }
    return NVDR_OK;
}

}  // namespace nvdr

using namespace nvdr;

#define NVDR_DISPATCH_A(KERNEL, DA, ...)                                                        \
    do {                                                                                         \
        const bool vec4 = (A == 4) && !((uintptr_t)attr & 15) && !((uintptr_t)VECPTR & 15);      \
        const bool vec2 = (A == 2) && !((uintptr_t)attr & 7) && !((uintptr_t)VECPTR & 7);        \
        if (vec4)      hipLaunchKernelGGL((KERNEL<4, DA>), __VA_ARGS__);                         \
        else if (vec2) hipLaunchKernelGGL((KERNEL<2, DA>), __VA_ARGS__);                         \
        else           hipLaunchKernelGGL((KERNEL<0, DA>), __VA_ARGS__);                         \
    } while (0)

extern "C" int nvdr_interpolate_fwd(const float* attr, const float* rast, const int32_t* tri,
                                    const float* rast_db, int attr_instance, int attr_n,
                                    int N, int V, int A, int T, int H, int W,
                                    int diff_all, const int32_t* diff_attrs_host, int num_diff,
                                    float* out, float* out_da, const uint8_t* tile_flags, nvdrStream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (debug_flags() & 2097152) return NVDR_OK;             // development: host-overhead measurement, nothing is launched (tools/host_profile.py)
    const bool enable_da = rast_db && (diff_all || num_diff > 0);
    InterpParams p;
    int rc = fill_params(p, attr, rast, tri, rast_db, attr_instance, attr_n, N, V, A, T, H, W,
                         diff_all, diff_attrs_host, num_diff, enable_da, "interpolate_fwd", tile_flags);
    if (rc) return rc;
    NVDR_REQUIRE(out, "interpolate_fwd: null output");
    NVDR_REQUIRE(!enable_da || out_da, "interpolate_fwd: out_da missing");
    NVDR_REQUIRE(!((uintptr_t)out_da & 7), "out_da output tensor not aligned to float2");
    p.out = out; p.outDA = enable_da ? out_da : nullptr;
    p.daVec4 = (enable_da && A == 2 && diff_all && !((uintptr_t)out_da & 15)) ? 1 : 0;
    // An output larger than most of the 256 MB Infinity Cache cannot be found there by its consumer anyway; written
    // around the cache it leaves `rast` (read again by the backward kernels) in place.
    p.streamOut = ((size_t)N * H * W * A * sizeof(float) > ((size_t)192 << 20)) ? 1 : 0;
    p.streamDA = (p.daVec4 && (size_t)N * H * W * 16 > ((size_t)192 << 20) && tune_int("NVDR_TUNE_IPFWD_STREAM_DA", 1)) ? 1 : 0;
    NVDR_REQUIRE((long long)H * W < (1ll << 31), "interpolate_fwd: image too large");
    const int perWg = 256 * ip_fwd_pixels(enable_da);
    dim3 grid((unsigned)(((long long)H * W + perWg - 1) / perWg), (unsigned)(N < 32768 ? N : 32768), (unsigned)((N + 32767) / 32768)), block(256);
    // the variant with differentials walks the work order behind the flags (0.253 -> 0.223 ms at config 3: a covered pixel costs
    // it 56 bytes more than an empty one); the plain one keeps the image order (85 vs 93 us at the headline batch)
    p.ordered = (p.flags.order && enable_da) ? 1 : 0;
    if (p.ordered) grid = dim3((unsigned)tile_flags_ordered_grid(p.flags, 16 / ip_fwd_pixels(enable_da)));
    const float* VECPTR = out;
    const bool vecA = (A == 4 && !((uintptr_t)attr & 15) && !((uintptr_t)out & 15)) || (A == 2 && !((uintptr_t)attr & 7) && !((uintptr_t)out & 7));
    {
        ProfileScope ps(enable_da ? "interp_fwd_da" : "interp_fwd", stream);
        if (enable_da) NVDR_DISPATCH_A(k_interp_fwd, true, grid, block, 0, stream, p);
        else if (vecA && tune_int("NVDR_TUNE_IPFWD_K", 4) > 0) {
            // vector rows: K pixels per thread, staged (k_interp_fwd_cols), along the work order where there is one.  Headline batch /
            // dense scene, us (r04n): one pixel per thread in image order 92 / 130; K = 1: 113 / 145 (ordered 102 / 141); K = 2: 97 / 121
            // (86 / 122); K = 4: 92 / 115 (83 / 117); K = 8: 91 / 119 (81 / 117).  For comparison, torch's copy of the dense
            // scene's 537 MB takes 100 us on these boxes (tools/write_bw.py: 5.35 TB/s read + written).  With the empty bins' zeros
            // stored by the covered bins' workgroups (ordered launches, see the kernel): 82 -> 75.5 us at the headline batch (r04t).
            const int K = tune_int("NVDR_TUNE_IPFWD_K", 4);
            p.ordered = (p.flags.order && tune_int("NVDR_TUNE_IPFWD_ORDERED", 1)) ? 1 : 0;
            const int gx = (W + 63) / 64, gy = (H + 4 * K - 1) / (4 * K);
            const long long total = p.ordered ? tile_flags_ordered_grid(p.flags, 16 / K) : (long long)gx * gy * N;
            NVDR_REQUIRE(total < (1ll << 30), "interpolate_fwd: too many pixel blocks");
            const dim3 cgrid((unsigned)(((total + 7) / 8) * 8));
#define NVDR_IPFWD(KK) do { if (A == 4) hipLaunchKernelGGL((k_interp_fwd_cols<4, KK>), cgrid, block, 0, stream, p, gx, gy);   \
                            else        hipLaunchKernelGGL((k_interp_fwd_cols<2, KK>), cgrid, block, 0, stream, p, gx, gy); } while (0)
            if (K == 1) NVDR_IPFWD(1); else if (K == 2) NVDR_IPFWD(2); else if (K == 8) NVDR_IPFWD(8); else NVDR_IPFWD(4);
        }
        else           NVDR_DISPATCH_A(k_interp_fwd, false, grid, block, 0, stream, p);
    }
    NVDR_LAUNCH_CHECK();
    return NVDR_OK;
}

extern "C" int nvdr_interpolate_grad(const float* attr, const float* rast, const int32_t* tri,
                                     const float* dy, const float* rast_db, const float* dda,
                                     int attr_instance, int attr_n,
                                     int N, int V, int A, int T, int H, int W,
                                     int diff_all, const int32_t* diff_attrs_host, int num_diff,
                                     float* g_attr, float* g_rast, float* g_rast_db, const uint8_t* tile_flags, nvdrStream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (debug_flags() & 2097152) return NVDR_OK;             // development: host-overhead measurement, nothing is launched (tools/host_profile.py)
    const bool enable_da = rast_db && dda && (diff_all || num_diff > 0);
    InterpParams p;
    int rc = fill_params(p, attr, rast, tri, rast_db, attr_instance, attr_n, N, V, A, T, H, W,
                         diff_all, diff_attrs_host, num_diff, enable_da, "interpolate_grad", tile_flags);
    if (rc) return rc;
    NVDR_REQUIRE(dy && g_attr && g_rast, "interpolate_grad: null pointer");
    NVDR_REQUIRE(!enable_da || g_rast_db, "interpolate_grad: g_rast_db missing");
    NVDR_REQUIRE(!((uintptr_t)dda & 7), "dda input tensor not aligned to float2");
    NVDR_REQUIRE(!((uintptr_t)g_rast & 15), "grad_rast output tensor not aligned to float4");
    NVDR_REQUIRE(!((uintptr_t)g_rast_db & 15), "grad_rast_db output tensor not aligned to float4");
    p.dy = dy; p.dda = enable_da ? dda : nullptr;
    p.gradAttr = g_attr; p.gradRaster = g_rast; p.gradRasterDB = enable_da ? g_rast_db : nullptr;
    const int gx = (W + kIpBlockW - 1) / kIpBlockW, gy = (H + kIpBlockH - 1) / kIpBlockH;
    const long long total = p.flags.order ? tile_flags_ordered_grid(p.flags, (64 / kIpBlockW) * (64 / kIpBlockH)) : (long long)gx * gy * N;   // (nvdr_device.hpp TileFlags)
    NVDR_REQUIRE(total < (1ll << 30), "interpolate_grad: too many pixel blocks");
    dim3 grid((unsigned)(((total + 7) / 8) * 8)), block(kIpThreads);
    // LDS vertex table: as many power-of-two slots as fit in 20 KiB (8 workgroups per CU), at most 512.
    int slots = 512;
    while (slots > 32 && (size_t)slots * (8 * A + 6) + 16 > 20 * 1024) slots >>= 1;
    // Vertices too wide for even the smallest table (A >= 256) go without one: every contribution becomes a
    // hardware f32 atomic, as in the reference (interpolate.cu:198-210), instead of an error.
    if ((size_t)slots * (8 * A + 6) + 16 > 64 * 1024) slots = 0;
    const size_t lds = (size_t)slots * (8 * A + 6) + 16;    // sums + key + used-list entry per slot
    const bool vec4 = (A == 4) && !((uintptr_t)attr & 15) && !((uintptr_t)dy & 15);
    const bool vec2 = (A == 2) && !((uintptr_t)attr & 7) && !((uintptr_t)dy & 7);
    {
        ProfileScope ps(enable_da ? "interp_grad_da" : "interp_grad", stream);
        if (enable_da) {
            if (vec4)      hipLaunchKernelGGL((k_interp_grad<4, true>), grid, block, lds, stream, p, slots, gx, gy);
            else if (vec2) hipLaunchKernelGGL((k_interp_grad<2, true>), grid, block, lds, stream, p, slots, gx, gy);
            else           hipLaunchKernelGGL((k_interp_grad<0, true>), grid, block, lds, stream, p, slots, gx, gy);
        } else {
            if (vec4)      hipLaunchKernelGGL((k_interp_grad<4, false>), grid, block, lds, stream, p, slots, gx, gy);
            else if (vec2) hipLaunchKernelGGL((k_interp_grad<2, false>), grid, block, lds, stream, p, slots, gx, gy);
            else           hipLaunchKernelGGL((k_interp_grad<0, false>), grid, block, lds, stream, p, slots, gx, gy);
        }
    }
    NVDR_LAUNCH_CHECK();
    return NVDR_OK;
}
