// interpolate.hip -- attribute interpolation forward / backward for gfx950.
//
// Replaces csrc/common/interpolate.cu + csrc/torch/torch_interpolate.cpp behind the C ABI.
//  k_interp_fwd   pure streaming: pixels are taken in linear order (one lane = one pixel,
//                 a wave = 64 consecutive pixels = 1 KiB of rast per load instruction);
//                 the three vertex attribute rows are gathered from L2.
//  k_interp_grad  one wave per 8x8 pixel tile so that the pixels of one triangle meet in
//                 one wave; attribute gradients are reduced per triangle with DPP and
//                 issued as one hardware f32 atomic per (triangle, vertex, attribute).
#include "nvdr_device.hpp"
#include "nvdr_host.hpp"

namespace nvdr {

constexpr int kMaxDiffAttrs = 32;       // interpolate.h:18 IP_MAX_DIFF_ATTRS

struct InterpParams {
    const int* tri; const float* attr; const float* rast; const float* rastDB;
    const float* dy; const float* dda;
    float* out; float* outDA; float* gradAttr; float* gradRaster; float* gradRasterDB;
    int numTriangles, numVertices, numAttr, numDiffAttr;
    int width, height, depth;
    int attrBC, instance_mode, diff_attrs_all, dbg;
    int diffAttrs[kMaxDiffAttrs];
};

__device__ __forceinline__ int diff_index(const InterpParams& p, int i)
{
    int j = p.diff_attrs_all ? i : p.diffAttrs[i];
    if (j < 0) j += p.numAttr;                              // python-style (interpolate.cu:102-103)
    return (j >= 0 && j < p.numAttr) ? j : -1;
}

// ---- forward (interpolate.cu:15-126) ---------------------------------------------------

template <int A_CT, bool ENABLE_DA>
__global__ __launch_bounds__(256) void k_interp_fwd(const InterpParams p)
{
    const size_t HW = (size_t)p.width * p.height;
    const size_t total = HW * p.depth;
    const size_t pidx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pidx >= total) return;
    const int A = A_CT > 0 ? A_CT : p.numAttr;
    const int pz = (int)(pidx / HW);

    float4 r = ((const float4*)p.rast)[pidx];
    int triIdx = float_to_triidx(r.w) - 1;
    bool valid = (triIdx >= 0 && triIdx < p.numTriangles);
    int vi0 = 0, vi1 = 0, vi2 = 0;
    if (valid) {
        vi0 = p.tri[triIdx * 3 + 0]; vi1 = p.tri[triIdx * 3 + 1]; vi2 = p.tri[triIdx * 3 + 2];
        if (vi0 < 0 || vi0 >= p.numVertices || vi1 < 0 || vi1 >= p.numVertices || vi2 < 0 || vi2 >= p.numVertices)
            return;                                         // corrupt indices: leave untouched (:54-58)
    }
    float* out = p.out + pidx * A;
    float2* outDA = ENABLE_DA ? ((float2*)p.outDA) + pidx * p.numDiffAttr : nullptr;

    if (!valid) {
        // No triangle: zeros (the reference reaches the same values via zero barycentrics, :73-80).
        if (A_CT == 4)      *(float4*)out = make_float4(0.f, 0.f, 0.f, 0.f);
        else if (A_CT == 2) *(float2*)out = make_float2(0.f, 0.f);
        else for (int i = 0; i < A; i++) out[i] = 0.f;
        if (ENABLE_DA) for (int i = 0; i < p.numDiffAttr; i++) outDA[i] = make_float2(0.f, 0.f);
        return;
    }
    if (p.instance_mode && !p.attrBC) { vi0 += pz * p.numVertices; vi1 += pz * p.numVertices; vi2 += pz * p.numVertices; }
    const float* a0 = p.attr + (size_t)vi0 * A;
    const float* a1 = p.attr + (size_t)vi1 * A;
    const float* a2 = p.attr + (size_t)vi2 * A;
    float b0 = r.x, b1 = r.y, b2 = 1.f - r.x - r.y;

    if (A_CT == 4) {
        float4 x0 = *(const float4*)a0, x1 = *(const float4*)a1, x2 = *(const float4*)a2;
        *(float4*)out = make_float4(b0 * x0.x + b1 * x1.x + b2 * x2.x, b0 * x0.y + b1 * x1.y + b2 * x2.y,
                                    b0 * x0.z + b1 * x1.z + b2 * x2.z, b0 * x0.w + b1 * x1.w + b2 * x2.w);
    } else if (A_CT == 2) {
        float2 x0 = *(const float2*)a0, x1 = *(const float2*)a1, x2 = *(const float2*)a2;
        *(float2*)out = make_float2(b0 * x0.x + b1 * x1.x + b2 * x2.x, b0 * x0.y + b1 * x1.y + b2 * x2.y);
    } else {
        for (int i = 0; i < A; i++) out[i] = b0 * a0[i] + b1 * a1[i] + b2 * a2[i];
    }
    if (!ENABLE_DA) return;

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

// ---- backward (interpolate.cu:131-274) -------------------------------------------------

template <int A_CT, bool ENABLE_DA>
__global__ __launch_bounds__(256) void k_interp_grad(const InterpParams p)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = (blockIdx.x * 4 + wave) * 8 + (lane & 7);
    const int py = blockIdx.y * 8 + (lane >> 3);
    const int pz = blockIdx.z;
    const int A = A_CT > 0 ? A_CT : p.numAttr;
    const bool inImage = (px < p.width) && (py < p.height);
    const size_t pidx = inImage ? ((size_t)pz * p.height + py) * p.width + px : 0;

    bool active = false;
    int triIdx = -1, vi0 = 0, vi1 = 0, vi2 = 0;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (inImage) {
        r = ((const float4*)p.rast)[pidx];
        triIdx = float_to_triidx(r.w) - 1;
        if (triIdx < 0 || triIdx >= p.numTriangles) {
            ((float4*)p.gradRaster)[pidx] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ENABLE_DA) ((float4*)p.gradRasterDB)[pidx] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            vi0 = p.tri[triIdx * 3 + 0]; vi1 = p.tri[triIdx * 3 + 1]; vi2 = p.tri[triIdx * 3 + 2];
            active = !(vi0 < 0 || vi0 >= p.numVertices || vi1 < 0 || vi1 >= p.numVertices || vi2 < 0 || vi2 >= p.numVertices);
        }
    }
    if (__ballot(active) == 0) return;

    if (p.instance_mode && !p.attrBC) { vi0 += pz * p.numVertices; vi1 += pz * p.numVertices; vi2 += pz * p.numVertices; }
    const float* a0 = p.attr + (size_t)vi0 * A;
    const float* a1 = p.attr + (size_t)vi1 * A;
    const float* a2 = p.attr + (size_t)vi2 * A;
    const float* pdy = p.dy + pidx * A;
    const float b0 = r.x, b1 = r.y, b2 = 1.f - r.x - r.y;

    // Per-pixel part: gradients w.r.t. the barycentrics (and their pixel differentials).
    float4 db = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        float gb0 = 0.f, gb1 = 0.f;
        for (int i = 0; i < A; i++) {
            float y = pdy[i];
            float s2 = a2[i];
            gb0 += y * (a0[i] - s2);
            gb1 += y * (a1[i] - s2);
        }
        ((float4*)p.gradRaster)[pidx] = make_float4(gb0, gb1, 0.f, 0.f);
        if (ENABLE_DA) {
            db = ((const float4*)p.rastDB)[pidx];
            const float2* dda = ((const float2*)p.dda) + pidx * p.numDiffAttr;
            float gdudx = 0.f, gdudy = 0.f, gdvdx = 0.f, gdvdy = 0.f;
            for (int i = 0; i < p.numDiffAttr; i++) {
                int j = diff_index(p, i);
                if (j < 0) continue;
                float2 d = dda[i];
                float dsdu = a0[j] - a2[j], dsdv = a1[j] - a2[j];
                gdudx += dsdu * d.x; gdudy += dsdu * d.y;
                gdvdx += dsdv * d.x; gdvdy += dsdv * d.y;
            }
            ((float4*)p.gradRasterDB)[pidx] = make_float4(gdudx, gdudy, gdvdx, gdvdy);
        }
    }

    // Attribute gradients: one atomic per (triangle, vertex, attribute) per wave.
    const bool foldDA = ENABLE_DA && p.diff_attrs_all;           // diff attr i == attr i: fold into one pass
    GroupIter it(active, triIdx);
    while (it.next()) {
        const int w0 = it.bcast(vi0), w1 = it.bcast(vi1), w2 = it.bcast(vi2);
        float* g0 = p.gradAttr + (size_t)w0 * A;
        float* g1 = p.gradAttr + (size_t)w1 * A;
        float* g2 = p.gradAttr + (size_t)w2 * A;
        for (int i = 0; i < A; i++) {
            float y = it.member ? pdy[i] : 0.f;
            float c0 = b0 * y, c1 = b1 * y, c2 = b2 * y;
            if (foldDA && it.member) {
                float2 d = (((const float2*)p.dda) + pidx * p.numDiffAttr)[i];
                float du = d.x * db.x + d.y * db.y;
                float dv = d.x * db.z + d.y * db.w;
                c0 += du; c1 += dv; c2 += -du - dv;
            }
            float s0 = (p.dbg & 2) ? c0 : it.sum(c0), s1 = (p.dbg & 2) ? c1 : it.sum(c1), s2 = (p.dbg & 2) ? c2 : it.sum(c2);
            if (it.writer() && !(p.dbg & 1)) { atomic_add_f32(g0 + i, s0); atomic_add_f32(g1 + i, s1); atomic_add_f32(g2 + i, s2); }
        }
        if (ENABLE_DA && !foldDA) {
            for (int i = 0; i < p.numDiffAttr; i++) {
                int j = diff_index(p, i);                         // uniform
                if (j < 0) continue;
                float du = 0.f, dv = 0.f;
                if (it.member) {
                    float2 d = (((const float2*)p.dda) + pidx * p.numDiffAttr)[i];
                    du = d.x * db.x + d.y * db.y;
                    dv = d.x * db.z + d.y * db.w;
                }
                float s0 = it.sum(du), s1 = it.sum(dv), s2 = it.sum(-du - dv);
                if (it.writer()) { atomic_add_f32(g0 + j, s0); atomic_add_f32(g1 + j, s1); atomic_add_f32(g2 + j, s2); }
            }
        }
    }
}

static int fill_params(InterpParams& p, const float* attr, const float* rast, const int32_t* tri, const float* rast_db,
                       int attr_instance, int attr_n, int N, int V, int A, int T, int H, int W,
                       int diff_all, const int32_t* diff_attrs_host, int num_diff, bool enable_da, const char* who)
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
                                    float* out, float* out_da, nvdrStream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    const bool enable_da = rast_db && (diff_all || num_diff > 0);
    InterpParams p;
    int rc = fill_params(p, attr, rast, tri, rast_db, attr_instance, attr_n, N, V, A, T, H, W,
                         diff_all, diff_attrs_host, num_diff, enable_da, "interpolate_fwd");
    if (rc) return rc;
    NVDR_REQUIRE(out, "interpolate_fwd: null output");
    NVDR_REQUIRE(!enable_da || out_da, "interpolate_fwd: out_da missing");
    NVDR_REQUIRE(!((uintptr_t)out_da & 7), "out_da output tensor not aligned to float2");
    p.out = out; p.outDA = enable_da ? out_da : nullptr;
    const size_t total = (size_t)N * H * W;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    const float* VECPTR = out;
    {
        ProfileScope ps(enable_da ? "interp_fwd_da" : "interp_fwd", stream);
        if (enable_da) NVDR_DISPATCH_A(k_interp_fwd, true, grid, block, 0, stream, p);
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
                                     float* g_attr, float* g_rast, float* g_rast_db, nvdrStream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    const bool enable_da = rast_db && dda && (diff_all || num_diff > 0);
    InterpParams p;
    int rc = fill_params(p, attr, rast, tri, rast_db, attr_instance, attr_n, N, V, A, T, H, W,
                         diff_all, diff_attrs_host, num_diff, enable_da, "interpolate_grad");
    if (rc) return rc;
    NVDR_REQUIRE(dy && g_attr && g_rast, "interpolate_grad: null pointer");
    NVDR_REQUIRE(!enable_da || g_rast_db, "interpolate_grad: g_rast_db missing");
    NVDR_REQUIRE(!((uintptr_t)dda & 7), "dda input tensor not aligned to float2");
    NVDR_REQUIRE(!((uintptr_t)g_rast & 15), "grad_rast output tensor not aligned to float4");
    NVDR_REQUIRE(!((uintptr_t)g_rast_db & 15), "grad_rast_db output tensor not aligned to float4");
    p.dy = dy; p.dda = enable_da ? dda : nullptr;
    p.gradAttr = g_attr; p.gradRaster = g_rast; p.gradRasterDB = enable_da ? g_rast_db : nullptr;
    dim3 grid((W + 31) / 32, (H + 7) / 8, N), block(256);
    {
        ProfileScope ps(enable_da ? "interp_grad_da" : "interp_grad", stream);
        if (enable_da) hipLaunchKernelGGL((k_interp_grad<0, true>),  grid, block, 0, stream, p);
        else if (A == 4) hipLaunchKernelGGL((k_interp_grad<4, false>), grid, block, 0, stream, p);
        else           hipLaunchKernelGGL((k_interp_grad<0, false>), grid, block, 0, stream, p);
    }
    NVDR_LAUNCH_CHECK();
    return NVDR_OK;
}
