/* nvdr_stub.c -- TEST INFRASTRUCTURE.  The eleven entry points of include/nvdr_hip.h that the compiled host layer
 * (nvdiffrast_amd/csrc_host/nvdr_torch_host.cpp) calls, implemented on HOST memory by the CPU oracle (oracle/nvdr_oracle.h), with
 * the C ABI's contracts: gradients are ADDED into buffers the caller zero-filled, dy / ddb may be NULL, tile flags are written.
 * tests/test_host_layer_logic.py runs the host layer's test build (CPU tensors) against it where there is no GPU: what is
 * being tested is the bookkeeping and the autograd plumbing of the host layer, not kernels. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/nvdr_hip.h"
#include "../../oracle/nvdr_oracle.h"

static int g_log_level = 1, g_limit_mb = 1024, g_calls[8];
enum { C_RAST_FWD, C_RAST_GRAD, C_INTERP_FWD, C_INTERP_GRAD, C_FUSED, C_CLEAN };
int nvdr_stub_calls(int i) { return g_calls[i]; }

const char* nvdr_last_error(void) { return "stub error"; }
int nvdr_get_option(int o) { return o == NVDR_OPT_LOG_LEVEL ? g_log_level : o == NVDR_OPT_SCRATCH_LIMIT_MB ? g_limit_mb : 0; }
int nvdr_set_option(int o, int v) { if (o == NVDR_OPT_SCRATCH_LIMIT_MB) g_limit_mb = v; else if (o == NVDR_OPT_LOG_LEVEL) g_log_level = v; return 0; }
int nvdr_log(int sev, const char* msg) { (void)msg; return sev >= g_log_level; }

size_t nvdr_rasterize_scratch_bytes_pool(int N, int max_tri, int H, int W, long long pool) {
    (void)H; (void)W;
    if (pool < 0 || pool >= 6ll * max_tri) pool = 6ll * max_tri;
    return 256 + (size_t)N * ((size_t)max_tri + (size_t)pool) * 68;
}
size_t nvdr_rasterize_pool_peak_offset(int N, int max_tri, int H, int W, long long pool) { (void)N; (void)max_tri; (void)H; (void)W; (void)pool; return 0; }
size_t nvdr_tile_flags_bytes(int N, int H, int W) { return (size_t)N * ((H + 7) / 8) * ((W + 7) / 8); }

int nvdr_rasterize_fwd(const float* pos, const int32_t* tri, const int32_t* ranges, int instance_mode, int N, int V, int T, int max_tri,
                       int H, int W, const uint32_t* peel_depth, uint32_t* depth_out, void* scratch, size_t scratch_bytes,
                       int scratch_clean, long long pool, float* out, float* out_db, uint8_t* tile_flags, nvdrStream_t stream) {
    (void)max_tri; (void)stream; (void)pool;
    const int hp = (H + 7) & ~7, wp = (W + 7) & ~7;
    uint32_t* depth = depth_out ? depth_out : (uint32_t*)malloc((size_t)N * hp * wp * 4);
    g_calls[C_RAST_FWD]++;
    g_calls[C_CLEAN] += scratch_clean;
    if (scratch_bytes >= 4) *(int32_t*)scratch = 0;          /* pool peak demand: nothing was clipped */
    int rc = nvdro_rasterize_fwd(pos, tri, ranges, instance_mode, N, V, T, H, W, peel_depth != NULL, peel_depth, depth, out, out_db);
    if (!depth_out) free(depth);
    if (tile_flags) {
        const int th = (H + 7) / 8, tw = (W + 7) / 8;
        memset(tile_flags, 0, (size_t)N * th * tw);
        for (int n = 0; n < N; n++) for (int y = 0; y < H; y++) for (int x = 0; x < W; x++)
            if (out[(((size_t)n * H + y) * W + x) * 4 + 3] > 0.f) tile_flags[((size_t)n * th + y / 8) * tw + x / 8] = 1;
    }
    return rc;
}

static float* zeros(size_t n) { return (float*)calloc(n ? n : 1, sizeof(float)); }

int nvdr_rasterize_grad(const float* pos, const int32_t* tri, const float* out, const float* dy, const float* ddb, int instance_mode,
                        int N, int V, int T, int H, int W, float* grad_pos, const uint8_t* tile_flags, nvdrStream_t stream) {
    (void)tile_flags; (void)stream;
    const size_t P = (size_t)N * H * W * 4, np = (size_t)(instance_mode ? N : 1) * V * 4;
    float* z = dy ? NULL : zeros(P);
    float* g = zeros(np);
    g_calls[C_RAST_GRAD]++;
    int rc = nvdro_rasterize_grad(pos, tri, out, dy ? dy : z, ddb, instance_mode, N, V, T, H, W, g);
    for (size_t i = 0; i < np; i++) grad_pos[i] += g[i];
    free(g); free(z);
    return rc;
}

int nvdr_interpolate_fwd(const float* attr, const float* rast, const int32_t* tri, const float* rast_db, int attr_instance, int attr_n,
                         int N, int V, int A, int T, int H, int W, int diff_all, const int32_t* diff, int num_diff,
                         float* out, float* out_da, const uint8_t* tile_flags, nvdrStream_t stream) {
    (void)tile_flags; (void)stream;
    g_calls[C_INTERP_FWD]++;
    return nvdro_interpolate_fwd(attr, rast, tri, rast_db, attr_instance, attr_n, N, V, A, T, H, W, diff_all, diff, num_diff, out, out_da);
}

int nvdr_interpolate_grad(const float* attr, const float* rast, const int32_t* tri, const float* dy, const float* rast_db, const float* dda,
                          int attr_instance, int attr_n, int N, int V, int A, int T, int H, int W, int diff_all, const int32_t* diff,
                          int num_diff, float* g_attr, float* g_rast, float* g_rast_db, const uint8_t* tile_flags, nvdrStream_t stream) {
    (void)tile_flags; (void)stream;
    const size_t na = (size_t)attr_n * V * A;
    float* g = zeros(na);
    g_calls[C_INTERP_GRAD]++;
    int rc = nvdro_interpolate_grad(attr, rast, tri, dy, rast_db, dda, attr_instance, attr_n, N, V, A, T, H, W, diff_all, diff, num_diff,
                                    g, g_rast, g_rast_db);
    for (size_t i = 0; i < na; i++) g_attr[i] += g[i];
    free(g);
    return rc;
}

int nvdr_interpolate_rasterize_grad(const float* attr, const float* rast, const int32_t* tri, const float* pos, const float* dy,
                                    int attr_instance, int attr_n, int pos_instance, int N, int V, int A, int T, int H, int W,
                                    const float* rast_db, const float* dda, int diff_all, const int32_t* diff, int num_diff, int db_to_pos,
                                    float* g_attr, float* g_pos, float* g_rast, float* g_rast_db, const uint8_t* tile_flags, nvdrStream_t stream) {
    const size_t P = (size_t)N * H * W * 4;
    float* gr = g_rast ? g_rast : zeros(P);
    float* gd = rast_db ? (g_rast_db ? g_rast_db : zeros(P)) : NULL;
    g_calls[C_FUSED]++;
    int rc = nvdr_interpolate_grad(attr, rast, tri, dy, rast_db, dda, attr_instance, attr_n, N, V, A, T, H, W, diff_all, diff, num_diff,
                                   g_attr, gr, gd, tile_flags, stream);
    g_calls[C_INTERP_GRAD]--;
    if (rc == 0) { rc = nvdr_rasterize_grad(pos, tri, rast, gr, db_to_pos ? gd : NULL, pos_instance, N, V, T, H, W, g_pos, tile_flags, stream); g_calls[C_RAST_GRAD]--; }
    if (!g_rast) free(gr);
    if (gd && !g_rast_db) free(gd);
    return rc;
}

/* ---- texture / antialias (the same contracts: gradients are added into zero-filled buffers) ---- */
enum { C_TEX_FWD = 6, C_TEX_GRAD = 7 };

int nvdr_texture_mip_info(int tex_n, int tex_h, int tex_w, int C, int cube, int max_mip_level, int* lw, int* lh, int64_t* off, int64_t* total) {
    return nvdro_texture_mip_info(tex_n, tex_h, tex_w, C, cube, max_mip_level, lw, lh, off, total);
}

int nvdr_texture_construct_mip(const float* tex, int tex_n, int tex_h, int tex_w, int C, int cube, int max_mip_level, float* mip, nvdrStream_t stream) {
    int lw[17], lh[17]; int64_t off[17], total;
    (void)stream;
    int L = nvdro_texture_mip_info(tex_n, tex_h, tex_w, C, cube, max_mip_level, lw, lh, off, &total);
    if (L < 0) return NVDR_ERR_ARG;
    return nvdro_texture_build_mip(tex, tex_n, tex_h, tex_w, C, cube, L, mip);
}

int nvdr_texture_fwd(const float* tex, const float* const* mip_ptrs, int L, const float* uv, const float* uv_da, const float* bias,
                     int tex_n, int tex_h, int tex_w, int C, int N, int H, int W, int filter, int boundary, float* out,
                     const uint8_t* tile_flags, nvdrStream_t stream) {
    (void)tile_flags; (void)stream;
    g_calls[C_TEX_FWD]++;
    return nvdro_texture_fwd(tex, mip_ptrs, L, uv, uv_da, bias, tex_n, tex_h, tex_w, C, N, H, W, filter, boundary, out);
}

size_t nvdr_texture_grad_scratch_bytes(int N, int H, int W, int C) { (void)N; (void)H; (void)W; (void)C; return 64; }

int nvdr_texture_grad(const float* tex, const float* const* mip_ptrs, int L, const float* uv, const float* uv_da, const float* bias,
                      const float* dy, int tex_n, int tex_h, int tex_w, int C, int N, int H, int W, int filter, int boundary,
                      int pull_mip_grads, float* g_tex, float* const* g_mip_ptrs, float* g_uv, float* g_uv_da, float* g_bias,
                      void* scratch, size_t scratch_bytes, const uint8_t* tile_flags, nvdrStream_t stream) {
    (void)scratch; (void)scratch_bytes; (void)tile_flags; (void)stream;
    const size_t nt = (size_t)tex_n * (boundary == 0 ? 6 : 1) * tex_h * tex_w * C;
    float* g = zeros(nt);
    g_calls[C_TEX_GRAD]++;
    int rc = nvdro_texture_grad(tex, mip_ptrs, L, uv, uv_da, bias, dy, tex_n, tex_h, tex_w, C, N, H, W, filter, boundary, pull_mip_grads,
                                g, g_mip_ptrs, g_uv, g_uv_da, g_bias);
    for (size_t i = 0; i < nt; i++) g_tex[i] += g[i];
    free(g);
    return rc;
}

int nvdr_antialias_fwd(const float* color, const float* rast, const float* pos, const int32_t* tri, const void* hash, size_t hash_bytes,
                       int instance_mode, int N, int V, int T, int H, int W, int C, float* out, void* work, size_t work_bytes,
                       const uint8_t* tile_flags, nvdrStream_t stream) {
    (void)hash; (void)hash_bytes; (void)work; (void)work_bytes; (void)tile_flags; (void)stream;
    return nvdro_antialias_fwd(color, rast, pos, tri, instance_mode, N, V, T, H, W, C, out);
}

int nvdr_antialias_grad(const float* color, const float* rast, const float* pos, const int32_t* tri, const float* dy, const void* work,
                        size_t work_bytes, int instance_mode, int N, int V, int T, int H, int W, int C, float* g_color, float* g_pos,
                        nvdrStream_t stream) {
    (void)work; (void)work_bytes; (void)stream;
    const size_t np = (size_t)(instance_mode ? N : 1) * V * 4;
    float* g = zeros(np);
    int rc = nvdro_antialias_grad(color, rast, pos, tri, dy, instance_mode, N, V, T, H, W, C, g_color, g);
    for (size_t i = 0; i < np; i++) g_pos[i] += g[i];
    free(g);
    return rc;
}
