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

/* Scratch for one forward call.  max_tri = T (instanced) or max(ranges[:,1]) (range mode). */
size_t nvdr_rasterize_scratch_bytes(int N, int max_tri, int H, int W);

/* instance_mode != 0: pos [N,V,4]; else pos [V,4] and ranges [N,2] (device copy of the
 * reference's CPU `ranges` tensor).  peel_depth: previous layer's depth surface
 * [N,Hpad,Wpad] u32 or NULL (NULL = no peel test, i.e. peeling_idx <= 0).  depth_out:
 * [N,Hpad,Wpad] u32 or NULL (only a DepthPeeler needs it).  Hpad/Wpad = H/W rounded up
 * to 8.  out, out_db: [N,H,W,4] f32. */
int nvdr_rasterize_fwd(const float* pos, const int32_t* tri, const int32_t* ranges,
                       int instance_mode, int N, int V, int T, int max_tri, int H, int W,
                       const uint32_t* peel_depth, uint32_t* depth_out,
                       void* scratch, size_t scratch_bytes,
                       float* out, float* out_db, nvdrStream_t stream);

/* grad_pos (shape of pos) must be zero-filled by the caller (reference: zeros_like,
 * torch_rasterize.cpp:237).  ddb == NULL selects the rasterize_grad variant. */
int nvdr_rasterize_grad(const float* pos, const int32_t* tri, const float* out,
                        const float* dy, const float* ddb,
                        int instance_mode, int N, int V, int T, int H, int W,
                        float* grad_pos, nvdrStream_t stream);

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
                         float* out, float* out_da, nvdrStream_t stream);

/* g_attr (shape of attr) must be zero-filled by the caller (torch_interpolate.cpp:211);
 * g_rast [N,H,W,4] and g_rast_db [N,H,W,4] (NULL without differentials) are fully written. */
int nvdr_interpolate_grad(const float* attr, const float* rast, const int32_t* tri,
                          const float* dy, const float* rast_db, const float* dda,
                          int attr_instance, int attr_n,
                          int N, int V, int A, int T, int H, int W,
                          int diff_all, const int32_t* diff_attrs_host, int num_diff,
                          float* g_attr, float* g_rast, float* g_rast_db, nvdrStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
