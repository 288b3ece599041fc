/*
 * nvdr_oracle.h -- CPU oracle for the nvdiffrast hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is a CPU restatement of the algorithms of the reference's four
 * ops (rasterize / interpolate / texture / antialias, forward + backward).  It is
 * the CHECKER for the HIP product path under nvdiffrast_amd/; nothing in the
 * product path may import, link or call it.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it.
 *
 * Parity pinning: the reference has no tests of its own, but its sources compile for
 * the host on the CUDA-on-CPU shim of oracle/refshim/ (-> oracle/_ref).  Every test
 * call into this library is also run through that build and must agree with it
 * (oracle/pinned.py; tests/test_ref_pins_oracle.py); docs/img/tri.png and the vectors
 * of tests/golden/reference_pipeline.npz (produced by the reference) are reproduced.
 *
 * All pointers are HOST pointers.  Layouts are the reference's: contiguous
 * [N,H,W,C] f32, row 0 = bottom scanline; pos [N,V,4] (instanced) or [V,4] +
 * ranges [N,2]; tri [T,3] i32.  Every function returns 0 on success.
 */
#ifndef NVDR_ORACLE_H
#define NVDR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- rasterize ------------------------------------------------------------- */

/* Forward.  Follows csrc/torch/torch_rasterize.cpp:43-166 (glue, viewport tiling),
 * cudaraster/impl/TriangleSetup.inl:11-435 (snap, cull, setup, clip),
 * Util.inl:101-160,184-210,304-309 (clipper, depth plane, fill rule),
 * FineRaster.inl:75-101,152-172,345-361 (samples, depth test, ROP) and
 * csrc/common/rasterize.cu:15-114 (pixel shader).
 *
 * instance_mode: pos is [N,V,4]; else pos is [V,4] and ranges [N,2] selects triangles.
 * depth_buf / peel_buf: [N,Hpad,Wpad] u32 (Hpad,Wpad = H,W rounded up to 8).
 *   depth_buf is always written (final depth surface).  peel_buf is read only when
 *   peel != 0 (previous layer's depth surface), cf. torch_rasterize.cpp:93-96.
 */
int nvdro_rasterize_fwd(const float* pos, const int32_t* tri, const int32_t* ranges,
                        int instance_mode, int N, int V, int T, int H, int W,
                        int peel, const uint32_t* peel_buf, uint32_t* depth_buf,
                        float* out, float* out_db);

/* Raw triangle-ID/depth surface only (no shader) -- used by the tests that probe the
 * integer rules directly.  id_buf [N,Hpad,Wpad] u32 (0 = background, else tri+1). */
int nvdro_rasterize_ids(const float* pos, const int32_t* tri, const int32_t* ranges,
                        int instance_mode, int N, int V, int T, int H, int W,
                        int peel, const uint32_t* peel_buf, uint32_t* depth_buf,
                        uint32_t* id_buf);

/* Backward.  rasterize.cu:119-277.  ddb may be NULL (rasterize_grad).  grad_pos has
 * the shape of pos and is fully overwritten (accumulated in f64, fixed order). */
int nvdro_rasterize_grad(const float* pos, const int32_t* tri, const float* out,
                         const float* dy, const float* ddb,
                         int instance_mode, int N, int V, int T, int H, int W,
                         float* grad_pos);

/* ---- interpolate ----------------------------------------------------------- */

/* interpolate.cu:15-126.  attr_instance: attr is [Nattr,V,A] (Nattr == N or 1 = broadcast),
 * else [V,A].  diff_attrs: list of num_diff indices (negative wrap), or NULL with
 * diff_all != 0.  rast_db/out_da may be NULL when num_diff == 0 and !diff_all. */
int nvdro_interpolate_fwd(const float* attr, const float* rast, const int32_t* tri,
                          const float* rast_db, int attr_instance, int Nattr,
                          int N, int V, int A, int T, int H, int W,
                          int diff_all, const int32_t* diff_attrs, int num_diff,
                          float* out, float* out_da);

/* interpolate.cu:131-274.  g_attr has the shape of attr; g_rast [N,H,W,4];
 * g_rast_db [N,H,W,4] or NULL. dda may be NULL when no diff attrs. */
int nvdro_interpolate_grad(const float* attr, const float* rast, const int32_t* tri,
                           const float* dy, const float* rast_db, const float* dda,
                           int attr_instance, int Nattr,
                           int N, int V, int A, int T, int H, int W,
                           int diff_all, const int32_t* diff_attrs, int num_diff,
                           float* g_attr, float* g_rast, float* g_rast_db);

/* ---- texture ---------------------------------------------------------------- */

/* Mip level geometry: csrc/common/texture.cpp:62-102.  Fills w/h/offset (in floats,
 * relative to the start of the mip buffer; level 0 gets offset -1) for levels
 * 0..max_level and returns the number of mip levels L (>=0) or -1 on a bad extent.
 * cube != 0: 6 faces per slice. */
int nvdro_texture_mip_info(int tex_n, int tex_h, int tex_w, int C, int cube,
                           int max_mip_level, int* lvl_w, int* lvl_h,
                           int64_t* lvl_off, int64_t* total_floats);

/* texture_kernel.cu:644-704: 2x2 box (1x2 on degenerate extents).  mip is the flat
 * buffer for levels 1..L. */
int nvdro_texture_build_mip(const float* tex, int tex_n, int tex_h, int tex_w, int C,
                            int cube, int L, float* mip);

/* texture_kernel.cu:709-800.  filter: 0 nearest, 1 linear, 2 l-m-nearest, 3 l-m-linear.
 * boundary: 0 cube (tex [tex_n,6,S,S,C], uv 3 and uv_da 6 components per pixel), 1 wrap, 2 clamp,
 * 3 zero.  uv_da / mip_level_bias may be NULL.
 * mip_ptrs: L pointers (levels 1..L), each [tex_n,(6,)h,w,C]; may be NULL if L == 0. */
/* 0 (default): cube corner texels as the reference treats them, including the lost corner flag for texture
 * slices >= 1 (texture_kernel.cu:431-432); 1: the flag is kept for every slice (opt-in fix). */
void nvdro_set_cube_corner_fix(int on);

int nvdro_texture_fwd(const float* tex, const float* const* mip_ptrs, int L,
                      const float* uv, const float* uv_da, const float* mip_level_bias,
                      int tex_n, int tex_h, int tex_w, int C,
                      int N, int H, int W, int filter, int boundary, float* out);

/* texture_kernel.cu:905-1140 (+ 843-900 when pull_mip_grads != 0: mip-level gradients
 * are folded into g_tex as MipGradKernel does and g_mip_ptrs are scratch).
 * Outputs may be NULL when the mode does not produce them. */
int nvdro_texture_grad(const float* tex, const float* const* mip_ptrs, int L,
                       const float* uv, const float* uv_da, const float* mip_level_bias,
                       const float* dy,
                       int tex_n, int tex_h, int tex_w, int C,
                       int N, int H, int W, int filter, int boundary,
                       int pull_mip_grads,
                       float* g_tex, float* const* g_mip_ptrs,
                       float* g_uv, float* g_uv_da, float* g_mip_level_bias);

/* Test hooks: cube-map face lookup and the edge fold (texel index x + w*(y + w*face), -1 = the texel
 * that does not exist at a cube corner). */
long long nvdro_cube_texel(int face, int ix, int iy, int w);
int nvdro_cube_index(const float* v, float* s, float* t);

/* ---- antialias -------------------------------------------------------------- */

/* antialias.cu:139-382 and 387-556.  The oracle does not model the hash table; it
 * uses an exact edge->opposite-vertex map with the reference's "first two triangles
 * that reach an edge are recorded, later ones ignored" rule in triangle order
 * (antialias.cu:82-96 under sequential insertion).
 * work (optional, may be NULL): per candidate record [px,py,pz|flags, alpha] is not
 * exposed; the grad entry point recomputes the analysis. */
int nvdro_antialias_fwd(const float* color, const float* rast, const float* pos,
                        const int32_t* tri, int instance_mode,
                        int N, int V, int T, int H, int W, int C, float* out);

int nvdro_antialias_grad(const float* color, const float* rast, const float* pos,
                         const int32_t* tri, const float* dy, int instance_mode,
                         int N, int V, int T, int H, int W, int C,
                         float* g_color, float* g_pos);

/* Number of OpenMP threads the library will use. */
int nvdro_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
