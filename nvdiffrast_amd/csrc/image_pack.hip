// image_pack.hip -- output images in the form the xGMI links can carry.
//
// No counterpart in the reference (it has no multi-GPU path: docs/index.html:758-759).  north_star's 8-GPU layout all-gathers the
// per-item output images every step; as f32 that is 4 MiB per 512^2 RGBA item and link -- 1.75 ms for the headline's 64 items
// against 0.33 ms of rendering.  k_image_pack turns the f32 image into what a consumer of rendered images needs -- unorm8
// (round(clamp(x, 0, 1) * 255), a quarter of the bytes) or f16 (half) -- in one streaming pass on the producing rank;
// k_image_unpack is the inverse for a receiver that wants f32 again.  Both are pure HBM streams: 8 elements per lane, 16-byte
// loads, 8- / 16-byte stores.
#include "nvdr_host.hpp"
#include "nvdr_device.hpp"

#include <hip/hip_fp16.h>

namespace {

using nvdr::load_streaming;
using nvdr::store_streaming;

constexpr int kThreads = 256;

__device__ __forceinline__ unsigned unorm8(float x) {
    // round-half-even of clamp(x, 0, 1) * 255, the arithmetic of torch's (x.clamp(0, 1) * 255).round().to(uint8); NaN -> 0
    x = x > 0.f ? x : 0.f;
    x = x < 1.f ? x : 1.f;
    return (unsigned)__float2int_rn(x * 255.f);
}

template <int FMT>   // 1 = f16, 2 = unorm8
__global__ __launch_bounds__(kThreads) void k_image_pack(const float* __restrict__ src, void* __restrict__ dst, size_t count) {
    const size_t groups = count >> 3;
    for (size_t g = (size_t)blockIdx.x * kThreads + threadIdx.x; g < groups; g += (size_t)gridDim.x * kThreads) {
        const float4 a = load_streaming((const float4*)src + 2 * g);
        const float4 b = load_streaming((const float4*)src + 2 * g + 1);
        if (FMT == 2) {
            uint2 o;
            o.x = unorm8(a.x) | unorm8(a.y) << 8 | unorm8(a.z) << 16 | unorm8(a.w) << 24;
            o.y = unorm8(b.x) | unorm8(b.y) << 8 | unorm8(b.z) << 16 | unorm8(b.w) << 24;
            ((uint2*)dst)[g] = o;
        } else {
            union { __half2 h[4]; uint4 u; } o;
            o.h[0] = __floats2half2_rn(a.x, a.y); o.h[1] = __floats2half2_rn(a.z, a.w);
            o.h[2] = __floats2half2_rn(b.x, b.y); o.h[3] = __floats2half2_rn(b.z, b.w);
            ((uint4*)dst)[g] = o.u;
        }
    }
    if (blockIdx.x == 0) {                                        // the last count % 8 elements
        const size_t i = (groups << 3) + threadIdx.x;
        if (threadIdx.x < 8 && i < count) {
            if (FMT == 2) ((uint8_t*)dst)[i] = (uint8_t)unorm8(src[i]);
            else          ((__half*)dst)[i] = __float2half_rn(src[i]);
        }
    }
}

template <int FMT>
__global__ __launch_bounds__(kThreads) void k_image_unpack(const void* __restrict__ src, float* __restrict__ dst, size_t count) {
    const size_t groups = count >> 3;
    for (size_t g = (size_t)blockIdx.x * kThreads + threadIdx.x; g < groups; g += (size_t)gridDim.x * kThreads) {
        float v[8];
        if (FMT == 2) {
            const uint2 q = ((const uint2*)src)[g];
#pragma unroll
            for (int k = 0; k < 4; k++) { v[k] = (float)((q.x >> (8 * k)) & 255u) / 255.f; v[4 + k] = (float)((q.y >> (8 * k)) & 255u) / 255.f; }
        } else {
            union { uint4 u; __half2 h[4]; } q;
            q.u = ((const uint4*)src)[g];
#pragma unroll
            for (int k = 0; k < 4; k++) { const float2 f = __half22float2(q.h[k]); v[2 * k] = f.x; v[2 * k + 1] = f.y; }
        }
        store_streaming((float4*)dst + 2 * g, make_float4(v[0], v[1], v[2], v[3]));
        store_streaming((float4*)dst + 2 * g + 1, make_float4(v[4], v[5], v[6], v[7]));
    }
    if (blockIdx.x == 0) {
        const size_t i = (groups << 3) + threadIdx.x;
        if (threadIdx.x < 8 && i < count) {
            if (FMT == 2) dst[i] = (float)((const uint8_t*)src)[i] / 255.f;
            else          dst[i] = __half2float(((const __half*)src)[i]);
        }
    }
}

// The first `CO` of `CI` channels per pixel (an RGB image out of RGBA / four attributes).  4 -> 3 as unorm8 has its own kernel:
// four pixels per lane, four 16-byte loads, three 4-byte stores; every other combination goes element by element.
__global__ __launch_bounds__(kThreads) void k_image_pack_rgb8_of4(const float* __restrict__ src, uint32_t* __restrict__ dst, size_t pixels) {
    const size_t groups = pixels >> 2;
    for (size_t g = (size_t)blockIdx.x * kThreads + threadIdx.x; g < groups; g += (size_t)gridDim.x * kThreads) {
        const float4 a = load_streaming((const float4*)src + 4 * g), b = load_streaming((const float4*)src + 4 * g + 1);
        const float4 c = load_streaming((const float4*)src + 4 * g + 2), d = load_streaming((const float4*)src + 4 * g + 3);
        dst[3 * g + 0] = unorm8(a.x) | unorm8(a.y) << 8 | unorm8(a.z) << 16 | unorm8(b.x) << 24;
        dst[3 * g + 1] = unorm8(b.y) | unorm8(b.z) << 8 | unorm8(c.x) << 16 | unorm8(c.y) << 24;
        dst[3 * g + 2] = unorm8(c.z) | unorm8(d.x) << 8 | unorm8(d.y) << 16 | unorm8(d.z) << 24;
    }
    if (blockIdx.x == 0 && threadIdx.x < 12) {                   // the last pixels % 4
        const size_t px = (groups << 2) + threadIdx.x / 3;
        if (px < pixels) ((uint8_t*)dst)[px * 3 + threadIdx.x % 3] = (uint8_t)unorm8(src[px * 4 + threadIdx.x % 3]);
    }
}

template <int FMT>
__global__ __launch_bounds__(kThreads) void k_image_pack_channels(const float* __restrict__ src, void* __restrict__ dst, size_t pixels, int CI, int CO) {
    const size_t n = pixels * (size_t)CO;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
        const float x = src[(i / CO) * CI + i % CO];
        if (FMT == 2) ((uint8_t*)dst)[i] = (uint8_t)unorm8(x);
        else if (FMT == 1) ((__half*)dst)[i] = __float2half_rn(x);
        else ((float*)dst)[i] = x;
    }
}

inline int grid_for(size_t count) {
    const size_t blocks = ((count >> 3) + kThreads - 1) / kThreads;
    return (int)(blocks < 1 ? 1 : (blocks > 256 * 32 ? 256 * 32 : blocks));           // at most 32 workgroups per CU, grid-stride beyond
}

}  // namespace

extern "C" size_t nvdr_image_packed_bytes(size_t count, int format) {
    return format == NVDR_IMAGE_UNORM8 ? count : format == NVDR_IMAGE_F16 ? 2 * count : 4 * count;
}

extern "C" int nvdr_image_pack(const float* src, void* dst, size_t pixels, int channels_in, int channels_out, int format, nvdrStream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NVDR_REQUIRE(src && dst, "image_pack: null pointer");
    NVDR_REQUIRE(format == NVDR_IMAGE_F32 || format == NVDR_IMAGE_F16 || format == NVDR_IMAGE_UNORM8, "image_pack: unknown format %d", format);
    NVDR_REQUIRE(channels_in > 0 && channels_out > 0 && channels_out <= channels_in, "image_pack: need 0 < channels_out <= channels_in");
    NVDR_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "image_pack: buffers must be 16-byte aligned");
    NVDR_REQUIRE(!(format == NVDR_IMAGE_F32 && channels_in == channels_out), "image_pack: nothing to do (f32, all channels)");
    if (pixels == 0) return NVDR_OK;
    nvdr::ProfileScope prof("image_pack", stream);
    const size_t count = pixels * (size_t)channels_out;
    if (channels_in == channels_out) {
        if (format == NVDR_IMAGE_UNORM8) k_image_pack<2><<<grid_for(count), kThreads, 0, stream>>>(src, dst, count);
        else                             k_image_pack<1><<<grid_for(count), kThreads, 0, stream>>>(src, dst, count);
    } else if (format == NVDR_IMAGE_UNORM8 && channels_in == 4 && channels_out == 3) {
        k_image_pack_rgb8_of4<<<grid_for(pixels * 2), kThreads, 0, stream>>>(src, (uint32_t*)dst, pixels);
    } else if (format == NVDR_IMAGE_UNORM8) {
        k_image_pack_channels<2><<<grid_for(count * 8), kThreads, 0, stream>>>(src, dst, pixels, channels_in, channels_out);
    } else if (format == NVDR_IMAGE_F16) {
        k_image_pack_channels<1><<<grid_for(count * 8), kThreads, 0, stream>>>(src, dst, pixels, channels_in, channels_out);
    } else {
        k_image_pack_channels<0><<<grid_for(count * 8), kThreads, 0, stream>>>(src, dst, pixels, channels_in, channels_out);
    }
    NVDR_LAUNCH_CHECK();
    return NVDR_OK;
}

extern "C" int nvdr_image_unpack(const void* src, float* dst, size_t count, int format, nvdrStream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    NVDR_REQUIRE(src && dst, "image_unpack: null pointer");
    NVDR_REQUIRE(format == NVDR_IMAGE_F16 || format == NVDR_IMAGE_UNORM8, "image_unpack: format must be NVDR_IMAGE_F16 or NVDR_IMAGE_UNORM8");
    NVDR_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "image_unpack: buffers must be 16-byte aligned");
    if (count == 0) return NVDR_OK;
    nvdr::ProfileScope prof("image_unpack", stream);
    if (format == NVDR_IMAGE_UNORM8) k_image_unpack<2><<<grid_for(count), kThreads, 0, stream>>>(src, dst, count);
    else                             k_image_unpack<1><<<grid_for(count), kThreads, 0, stream>>>(src, dst, count);
    NVDR_LAUNCH_CHECK();
    return NVDR_OK;
}
