/*
 * nvdr_hip.h -- C ABI of libnvdr_hip.so, the MI355X (gfx950) implementation of the
 * nvdiffrast.torch hot path.  This is the drop-in boundary: one entry point per op of
 * the reference's pybind11 module `_nvdiffrast_c` (csrc/torch/torch_bindings.cpp:43-71),
 * with torch types replaced by raw device pointers, sizes and a hipStream_t.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *  - the library never allocates device memory: outputs and scratch are passed in
 *    (`*_scratch_bytes` tells how much); every launch goes to `stream` and is async;
 *  - tensors are contiguous f32 / i32 with the reference's layouts ([N,H,W,C], row 0 =
 *    bottom scan line; pos [N,V,4] or [V,4]; tri [T,3]);
 *  - return value: 0 = ok, otherwise an NVDR_ERR_* code; nvdr_last_error() returns a
 *    thread-local message (mirrors the reference's NVDR_CHECK text where one exists).
 */
#ifndef NVDR_HIP_H
#define NVDR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* nvdrStream_t;           /* hipStream_t */

enum {
    NVDR_OK            = 0,
    NVDR_ERR_ARG       = 1,           /* bad shape / null pointer / misalignment       */
    NVDR_ERR_SCRATCH   = 2,           /* scratch buffer too small                       */
    NVDR_ERR_LAUNCH    = 3,           /* HIP runtime reported an error                  */
    NVDR_ERR_OVERFLOW  = 4            /* "subtriangle count overflow" (torch_rasterize.cpp:123) */
};

const char* nvdr_last_error(void);
int         nvdr_abi_version(void);

/* Process-wide options.  Defaults reproduce the reference; nvdr_set_option returns NVDR_ERR_ARG for an unknown id.
 *  NVDR_OPT_LOG_LEVEL        c10 log severity threshold, the value behind get_log_level / set_log_level
 *                            (torch_bindings.cpp:50-51): messages of severity >= level go to stderr
 *                            (0 INFO, 1 WARNING (default), 2 ERROR, 3 FATAL).  The one message on this path is the
 *                            rasterizer's INFO "Internal buffers grown to N MB" (RasterImpl.cpp:195), which the
 *                            host glue emits when it regrows the scratch it owns.
 *  NVDR_OPT_CUBE_CORNER_FIX  0 (default): cube-map corner texels exactly as the reference samples them, including
 *                            its loss of the corner flag for texture slices >= 1 (texture_kernel.cu:85-88,431-432);
 *                            1: the missing corner texel is the average of the other three for every slice.
 *  NVDR_OPT_SCRATCH_LIMIT_MB rasterizer scratch policy of the HOST GLUE (the library itself never allocates): up to this
 *                            many MiB (default 4096: 1.4 % of the MI355X's 288 GB; batch 256 of a 10 k-triangle mesh needs 1.2 GiB) the glue reserves the worst case -- 6 extra record slots per
 *                            triangle for the clipper -- so that nothing can overflow and no call ever synchronises
 *                            the host; above it, it starts from a small clip pool, reads the pool demand back after
 *                            each call (one host synchronisation, as the reference does every call,
 *                            RasterImpl.cpp:174-231,367) and repeats the call with a larger pool when it was short. */
enum { NVDR_OPT_LOG_LEVEL = 0, NVDR_OPT_CUBE_CORNER_FIX = 1, NVDR_OPT_SCRATCH_LIMIT_MB = 2, NVDR_OPT_COUNT = 3 };
int nvdr_set_option(int option, int value);
int nvdr_get_option(int option);
/* Writes `msg` to stderr as "[nvdr] msg" when `severity` >= NVDR_OPT_LOG_LEVEL; returns 1 if it was written. */
int nvdr_log(int severity, const char* msg);

/* Per-kernel timing hooks used by bench.py (hipEvent pairs recorded on the launch
 * stream around every kernel while enabled).  nvdr_profile_read fills up to `cap`
 * entries (name pointers stay valid for the library's lifetime) and returns the count;
 * it synchronises the recorded events.  Not used on the product path. */
void nvdr_profile_enable(int on);
void nvdr_profile_reset(void);
int  nvdr_profile_read(const char** names, double* total_ms, int* launches, int cap);

/* ---- rasterize ------------------------------------------------------------------
 * Replaces rasterize_fwd_cuda / rasterize_grad / rasterize_grad_db
 * (torch_bindings.cpp:54-56; csrc/torch/torch_rasterize.cpp:43-166, 171-263) and the
 * CudaRaster runtime behind them (csrc/common/cudaraster/). */

/* Scratch for one forward call.  max_tri = T (instanced) or max(ranges[:,1]) (range mode).
 * The buffer starts with triangle records and ends with a small control block (counters).  Every
 * successful nvdr_rasterize_fwd leaves the control block zeroed, which is the state the next call
 * needs: a caller that passes the SAME buffer with the SAME (N, max_tri, H, W) as its previous
 * successful call, and has not written to it in between, may say so with scratch_clean = 1 and saves
 * a memset launch per call.  scratch_clean = 0 is always safe (the library clears the block itself). */
size_t nvdr_rasterize_scratch_bytes(int N, int max_tri, int H, int W);

/* The same with a caller-chosen clip pool: `pool_per_image` record slots per image receive the sub-triangles that the
 * frustum clipper produces beyond the first (a clipped triangle becomes 1..7 sub-triangles); < 0 or >= 6 * max_tri
 * means the worst case, which is what nvdr_rasterize_scratch_bytes sizes for (record slots are 68 bytes: at
 * N = 64, T = 1 M the worst case is 30 GB, a pool of T / 4 makes it 5.4 GB).  With a short pool a call can run
 * out of slots: the sub-triangles that do not fit are left out, and the int at byte offset
 * nvdr_rasterize_pool_peak_offset(...) of the scratch buffer holds, after the call, the largest per-image demand;
 * if it exceeds pool_per_image the output is incomplete and the call must be repeated with a larger pool
 * (the reference sizes its buffers by the same retry, RasterImpl.cpp:174-231). */
size_t nvdr_rasterize_scratch_bytes_pool(int N, int max_tri, int H, int W, long long pool_per_image);
size_t nvdr_rasterize_pool_peak_offset(int N, int max_tri, int H, int W, long long pool_per_image);

/* instance_mode != 0: pos [N,V,4]; else pos [V,4] and ranges [N,2] (device copy of the
 * reference's CPU `ranges` tensor).  peel_depth: previous layer's depth surface
 * [N,Hpad,Wpad] u32 or NULL (NULL = no peel test, i.e. peeling_idx <= 0).  depth_out:
 * [N,Hpad,Wpad] u32 or NULL (only a DepthPeeler needs it).  Hpad/Wpad = H/W rounded up
 * to 8.  out, out_db: [N,H,W,4] f32. */
/* tile_flags (optional output, NULL = none): a buffer of nvdr_tile_flags_bytes(N, H, W) bytes.  First [N][ceil(H/8)]
 * [ceil(W/8)] bytes, 1 = some pixel of that 8x8-pixel tile of `out` shows a triangle, 0 = the whole tile is background;
 * then, 16-byte aligned and only for batches of 2048 .. 65536 bins of 64x64 pixels of images up to 2048 pixels a side, a
 * work order for the consuming kernels: the bins with a covered tile first, then the others, as int32 bin numbers, and
 * their count; then, 8-byte aligned, what that order is built from: one byte per bin and tile row (csrc/nvdr_device.hpp
 * TileFlags).  The entry points below that READ a rast tensor take the same buffer as an optional input (`tile_flags`,
 * NULL = none); they then skip the rast / rast_db bytes of empty tiles and may walk the image in that order -- legal only
 * while that rast tensor is exactly what this call wrote (the operator layer checks identity, shape and version). */
size_t nvdr_tile_flags_bytes(int N, int H, int W);
int nvdr_rasterize_fwd(const float* pos, const int32_t* tri, const int32_t* ranges,
                       int instance_mode, int N, int V, int T, int max_tri, int H, int W,
                       const uint32_t* peel_depth, uint32_t* depth_out,
                       void* scratch, size_t scratch_bytes, int scratch_clean, long long pool_per_image,
                       float* out, float* out_db, uint8_t* tile_flags, nvdrStream_t stream);

/* grad_pos (shape of pos) must be zero-filled by the caller (reference: zeros_like,
 * torch_rasterize.cpp:237).  ddb == NULL selects the rasterize_grad variant.  dy == NULL (with ddb): only ddb's share is
 * ADDED to grad_pos -- for a caller that holds dy's share already (nvdr_interpolate_rasterize_grad); by linearity the sum
 * is rasterize_grad_db(dy, ddb). */
int nvdr_rasterize_grad(const float* pos, const int32_t* tri, const float* out,
                        const float* dy, const float* ddb,
                        int instance_mode, int N, int V, int T, int H, int W,
                        float* grad_pos, const uint8_t* tile_flags, nvdrStream_t stream);

/* ---- interpolate ----------------------------------------------------------------
 * Replaces interpolate_fwd / interpolate_fwd_da / interpolate_grad / interpolate_grad_da
 * (torch_bindings.cpp:57-60; csrc/torch/torch_interpolate.cpp:42-132, 137-248). */

/* attr_instance != 0: attr [attr_n,V,A] with attr_n == N or 1 (broadcast); else attr [V,A].
 * diff_attrs_host: HOST array of num_diff indices (<= 32, negative wrap) or NULL with
 * diff_all != 0.  rast_db / out_da are NULL when no pixel differentials are requested.
 * out [N,H,W,A], out_da [N,H,W,2*D]. */
int nvdr_interpolate_fwd(const float* attr, const float* rast, const int32_t* tri,
                         const float* rast_db, int attr_instance, int attr_n,
                         int N, int V, int A, int T, int H, int W,
                         int diff_all, const int32_t* diff_attrs_host, int num_diff,
                         float* out, float* out_da, const uint8_t* tile_flags, nvdrStream_t stream);

/* g_attr (shape of attr) must be zero-filled by the caller (torch_interpolate.cpp:211);
 * g_rast [N,H,W,4] and g_rast_db [N,H,W,4] (NULL without differentials) are fully written. */
int nvdr_interpolate_grad(const float* attr, const float* rast, const int32_t* tri,
                          const float* dy, const float* rast_db, const float* dda,
                          int attr_instance, int attr_n,
                          int N, int V, int A, int T, int H, int W,
                          int diff_all, const int32_t* diff_attrs_host, int num_diff,
                          float* g_attr, float* g_rast, float* g_rast_db, const uint8_t* tile_flags, nvdrStream_t stream);

/* ---- interpolate + rasterize backward in one pass --------------------------------------
 * Not an entry point of the reference: the work of interpolate_grad (torch_interpolate.cpp:242-248; interpolate.cu:131-274)
 * followed by rasterize_grad (torch_rasterize.cpp:259-263; rasterize.cu:119-277) for the common graph
 * rasterize -> interpolate (no pixel differentials), in one kernel that reads `rast` once and hands a pixel's (u, v)
 * gradient to the rasterizer's backward in registers.
 *   g_attr (shape of attr) and g_pos (shape of pos) must be zero-filled by the caller, as for the two separate calls;
 *   g_rast [N,H,W,4] = what interpolate_grad would have written, or NULL to skip it (only legal when nothing else
 *   consumes the gradient of `rast`; the operator layer passes NULL and hands autograd a stand-in that computes the
 *   tensor only if somebody looks at it, nvdiffrast_amd/torch/ops.py `_LazyGrad`).
 * attr_instance / attr_n as for nvdr_interpolate_grad; pos_instance != 0: pos [N,V,4], else pos [V,4] (range mode).
 * Results equal those of the two separate calls up to the summation order of the f32 atomics.
 * With pixel differentials (rast_db and dda given, diff_all / diff_attrs_host / num_diff as for nvdr_interpolate_grad) it is
 * interpolate_grad_da followed by rasterize_grad_db; g_rast_db [N,H,W,4] is written next to g_rast; db_to_pos == 0 keeps
 * the gradient of rast_db away from pos (a rasterize call made with grad_db = False). */
int nvdr_interpolate_rasterize_grad(const float* attr, const float* rast, const int32_t* tri, const float* pos,
                                    const float* dy, int attr_instance, int attr_n, int pos_instance,
                                    int N, int V, int A, int T, int H, int W,
                                    const float* rast_db, const float* dda,
                                    int diff_all, const int32_t* diff_attrs_host, int num_diff, int db_to_pos,
                                    float* g_attr, float* g_pos, float* g_rast, float* g_rast_db,
                                    const uint8_t* tile_flags, nvdrStream_t stream);

/* ---- texture --------------------------------------------------------------------
 * Replaces texture_construct_mip / texture_fwd / texture_fwd_mip / texture_grad_nearest /
 * texture_grad_linear / texture_grad_linear_mipmap_nearest / texture_grad_linear_mipmap_linear
 * (torch_bindings.cpp:61-67; csrc/torch/torch_texture.cpp:98-716).
 * filter_mode: 0 nearest, 1 linear, 2 linear-mipmap-nearest, 3 linear-mipmap-linear;
 * boundary_mode: 0 cube, 1 wrap, 2 clamp, 3 zero (ops.py:415-420).
 * 2D: tex [tex_n,tex_h,tex_w,C] with tex_n == N or 1; uv [N,H,W,2]; uv_da [N,H,W,4] or NULL.
 * Cube (boundary_mode 0): tex [tex_n,6,S,S,C] passed as tex_h = tex_w = S; uv [N,H,W,3] direction
 * vectors; uv_da [N,H,W,6] = (dx/dX, dx/dY, dy/dX, dy/dY, dz/dX, dz/dY) or NULL.
 * mip_level_bias [N,H,W] or NULL; out / dy [N,H,W,C]. */

/* Mip geometry (pure host code; csrc/common/texture.cpp:62-102): fills widths / heights /
 * offsets-in-floats (relative to the mip buffer; entry 0 is the base level, offset -1) for levels
 * 0..L and returns L, or -1 when an extent cannot be halved (odd size above 1).  Arrays need 17
 * entries; any of them may be NULL. */
int nvdr_texture_mip_info(int tex_n, int tex_h, int tex_w, int C, int cube, int max_mip_level,
                          int* lvl_w, int* lvl_h, int64_t* lvl_off, int64_t* total_floats);

/* Builds levels 1..L into `mip` (total_floats from nvdr_texture_mip_info). */
int nvdr_texture_construct_mip(const float* tex, int tex_n, int tex_h, int tex_w, int C, int cube,
                               int max_mip_level, float* mip, nvdrStream_t stream);

/* tile_flags of the two texture entry points (optional, NULL = none; 2-D textures only): the rasterizer's occupancy flags
 * for the image that uv / uv_da were interpolated over -- in a tile flagged 0 both are KNOWN to be zero (interpolate writes
 * zeros where no triangle is visible) and are not read.  Only legal while uv and uv_da are exactly what the interpolate
 * call for that rast wrote (the operator layer checks identity and version). */
/* mip_ptrs_host: HOST array of L device pointers (levels 1..L; the wrapper's flat buffer plus
 * offsets, or the tensors of a custom stack); NULL / L = 0 for the non-mipmapped filters. */
int nvdr_texture_fwd(const float* tex, const float* const* mip_ptrs_host, int L,
                     const float* uv, const float* uv_da, const float* mip_level_bias,
                     int tex_n, int tex_h, int tex_w, int C, int N, int H, int W,
                     int filter_mode, int boundary_mode, float* out, const uint8_t* tile_flags, nvdrStream_t stream);

/* g_tex (shape of tex) and every g_mip level must be zero-filled by the caller
 * (torch_texture.cpp:523,583-604); g_uv [N,H,W,2] (NULL for nearest), g_uv_da [N,H,W,4] and
 * g_mip_level_bias [N,H,W] (linear-mipmap-linear only, NULL when the input is absent) are fully
 * written.  pull_mip_grads != 0 folds the level gradients into g_tex afterwards (the internal mip
 * chain, torch_texture.cpp:679-687); custom stacks keep their own gradients. */
/* scratch (optional, NULL = none; nvdr_texture_grad_scratch_bytes(N, H, W, C) bytes, 4-byte aligned, contents irrelevant):
 * with it, pixels that share one texel quad wave by wave -- a rendered image's background, where interpolate() leaves
 * uv = 0 -- are reduced in two levels (per-wave records, then a fold kernel) instead of one flush per 16x16-pixel block
 * into the same four texels; results are the same up to the order of the f32 sums. */
size_t nvdr_texture_grad_scratch_bytes(int N, int H, int W, int C);
int nvdr_texture_grad(const float* tex, const float* const* mip_ptrs_host, int L,
                      const float* uv, const float* uv_da, const float* mip_level_bias, const float* dy,
                      int tex_n, int tex_h, int tex_w, int C, int N, int H, int W,
                      int filter_mode, int boundary_mode, int pull_mip_grads,
                      float* g_tex, float* const* g_mip_ptrs_host,
                      float* g_uv, float* g_uv_da, float* g_mip_level_bias,
                      void* scratch, size_t scratch_bytes, const uint8_t* tile_flags, nvdrStream_t stream);

/* ---- antialias ------------------------------------------------------------------
 * Replaces antialias_construct_topology_hash / antialias_fwd / antialias_grad
 * (torch_bindings.cpp:68-70; csrc/torch/torch_antialias.cpp:25-241). */

size_t nvdr_antialias_hash_bytes(int T);                 /* torch_antialias.cpp:43-49 sizing */
size_t nvdr_antialias_work_bytes(int N, int H, int W);   /* torch_antialias.cpp:123: (P*8+4) floats */

/* Fills `hash` (cleared by the call) with the edge -> opposite-vertex table of `tri` [T,3]. */
int nvdr_antialias_construct_topology_hash(const int32_t* tri, int T, void* hash, size_t hash_bytes,
                                           nvdrStream_t stream);

/* color / out [N,H,W,C]; rast [N,H,W,4]; pos [N,V,4] (instance_mode) or [V,4].  `work` receives
 * the work items that the gradient pass replays; out is fully written (copy of color + blend). */
int nvdr_antialias_fwd(const float* color, const float* rast, const float* pos, const int32_t* tri,
                       const void* hash, size_t hash_bytes,
                       int instance_mode, int N, int V, int T, int H, int W, int C,
                       float* out, void* work, size_t work_bytes, const uint8_t* tile_flags, nvdrStream_t stream);

/* g_color [N,H,W,C] is fully written (copy of dy + corrections); g_pos (shape of pos) must be
 * zero-filled by the caller (torch_antialias.cpp:219). */
int nvdr_antialias_grad(const float* color, const float* rast, const float* pos, const int32_t* tri,
                        const float* dy, const void* work, size_t work_bytes,
                        int instance_mode, int N, int V, int T, int H, int W, int C,
                        float* g_color, float* g_pos, nvdrStream_t stream);

/* ---- output images for the xGMI links --------------------------------------------------------------------------
 * Not entry points of the reference (it has no multi-GPU path, docs/index.html:758-759): north_star's 8-GPU layout all-gathers
 * the per-item output images every step, and an f32 image is 4x what a consumer of rendered images needs.  nvdr_image_pack
 * converts an image of `pixels` pixels with `channels_in` f32 channels each on the producing rank: the first `channels_out`
 * channels of every pixel (an RGB image out of RGBA / four attributes), as NVDR_IMAGE_UNORM8: round(clamp(x, 0, 1) * 255)
 * (round-half-even, NaN -> 0), one byte each; NVDR_IMAGE_F16: round-to-nearest-even halves; NVDR_IMAGE_F32: the selected
 * channels as they are.  nvdr_image_unpack is the inverse for `count` packed values (q / 255; exact widening) for a receiver
 * that wants f32 again.  Buffers 16-byte aligned; nvdr_image_packed_bytes(pixels * channels_out, format) sizes `dst`. */
enum { NVDR_IMAGE_F32 = 0, NVDR_IMAGE_F16 = 1, NVDR_IMAGE_UNORM8 = 2 };
size_t nvdr_image_packed_bytes(size_t count, int format);
int nvdr_image_pack(const float* src, void* dst, size_t pixels, int channels_in, int channels_out, int format, nvdrStream_t stream);
int nvdr_image_unpack(const void* src, float* dst, size_t count, int format, nvdrStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
