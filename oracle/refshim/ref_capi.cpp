// C entry points over the reference's own operator functions (the ones csrc/torch/torch_bindings.cpp:23-39
// declares and binds), compiled for the host by oracle/refshim/build.py.  Tensors cross this boundary as
// (pointer, dtype, shape) descriptors and are copied into the miniature tensor type of
// include/torch/extension.h; results come back as a list the caller copies out of.
// TEST INFRASTRUCTURE ONLY — the product never links this.
#include <torch/extension.h>
#include "torch_types.h"
#include "../CudaRaster.hpp"

// Prototypes exactly as torch_bindings.cpp:23-39 declares them.
std::tuple<torch::Tensor, torch::Tensor> rasterize_fwd_cuda(RasterizeCRStateWrapper& stateWrapper, torch::Tensor pos, torch::Tensor tri, std::tuple<int, int> resolution, torch::Tensor ranges, int peeling_idx);
torch::Tensor rasterize_grad(torch::Tensor pos, torch::Tensor tri, torch::Tensor out, torch::Tensor dy);
torch::Tensor rasterize_grad_db(torch::Tensor pos, torch::Tensor tri, torch::Tensor out, torch::Tensor dy, torch::Tensor ddb);
std::tuple<torch::Tensor, torch::Tensor> interpolate_fwd(torch::Tensor attr, torch::Tensor rast, torch::Tensor tri);
std::tuple<torch::Tensor, torch::Tensor> interpolate_fwd_da(torch::Tensor attr, torch::Tensor rast, torch::Tensor tri, torch::Tensor rast_db, bool diff_attrs_all, std::vector<int>& diff_attrs_vec);
std::tuple<torch::Tensor, torch::Tensor> interpolate_grad(torch::Tensor attr, torch::Tensor rast, torch::Tensor tri, torch::Tensor dy);
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> interpolate_grad_da(torch::Tensor attr, torch::Tensor rast, torch::Tensor tri, torch::Tensor dy, torch::Tensor rast_db, torch::Tensor dda, bool diff_attrs_all, std::vector<int>& diff_attrs_vec);
TextureMipWrapper texture_construct_mip(torch::Tensor tex, int max_mip_level, bool cube_mode);
torch::Tensor texture_fwd(torch::Tensor tex, torch::Tensor uv, int filter_mode, int boundary_mode);
torch::Tensor texture_fwd_mip(torch::Tensor tex, torch::Tensor uv, torch::Tensor uv_da, torch::Tensor mip_level_bias, TextureMipWrapper mip_wrapper, std::vector<torch::Tensor> mip_stack, int filter_mode, int boundary_mode);
torch::Tensor texture_grad_nearest(torch::Tensor tex, torch::Tensor uv, torch::Tensor dy, int filter_mode, int boundary_mode);
std::tuple<torch::Tensor, torch::Tensor> texture_grad_linear(torch::Tensor tex, torch::Tensor uv, torch::Tensor dy, int filter_mode, int boundary_mode);
std::tuple<torch::Tensor, torch::Tensor, std::vector<torch::Tensor> > texture_grad_linear_mipmap_nearest(torch::Tensor tex, torch::Tensor uv, torch::Tensor dy, torch::Tensor uv_da, torch::Tensor mip_level_bias, TextureMipWrapper mip_wrapper, std::vector<torch::Tensor> mip_stack, int filter_mode, int boundary_mode);
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, std::vector<torch::Tensor> > texture_grad_linear_mipmap_linear(torch::Tensor tex, torch::Tensor uv, torch::Tensor dy, torch::Tensor uv_da, torch::Tensor mip_level_bias, TextureMipWrapper mip_wrapper, std::vector<torch::Tensor> mip_stack, int filter_mode, int boundary_mode);
TopologyHashWrapper antialias_construct_topology_hash(torch::Tensor tri);
std::tuple<torch::Tensor, torch::Tensor> antialias_fwd(torch::Tensor color, torch::Tensor rast, torch::Tensor pos, torch::Tensor tri, TopologyHashWrapper topology_hash);
std::tuple<torch::Tensor, torch::Tensor> antialias_grad(torch::Tensor color, torch::Tensor rast, torch::Tensor pos, torch::Tensor tri, torch::Tensor dy, torch::Tensor work_buffer);

int FLAGS_caffe2_log_level = 1;
namespace { std::string g_log; }
c10::LogLine::~LogLine(void)
{
    g_log += s.str() + "\n";
    if (sev >= FLAGS_caffe2_log_level) fprintf(stderr, "[nvdr_ref] %s\n", s.str().c_str());
}

namespace
{
thread_local std::string g_error;

struct Results { std::vector<torch::Tensor> t; };
}

extern "C"
{
// dtype: 0 = float32, 1 = int32.  device: 1 = "cuda" (what the glue requires of every input but `ranges`), 0 = cpu.
// ndim < 0 means "no tensor" (an undefined torch::Tensor).
struct nvdr_ref_tensor { const void* data; int dtype; int device; int ndim; int64_t shape[8]; };
}

namespace
{
torch::Tensor import_tensor(const nvdr_ref_tensor* d)
{
    if (!d || d->ndim < 0)
        return torch::Tensor();
    std::vector<int64_t> shape(d->shape, d->shape + d->ndim);
    torch::TensorOptions o = torch::TensorOptions().dtype(d->dtype == 1 ? torch::kInt32 : torch::kFloat32)
                                                   .device(d->device ? torch::kCUDA : torch::kCPU);
    torch::Tensor t = torch::empty(c10::IntArrayRef(shape), o);
    if (t.nbytes())
        memcpy(t.data_ptr(), d->data, t.nbytes());
    return t;
}

void put(Results* r, const torch::Tensor& t)                                   { r->t.push_back(t); }
void put(Results* r, const std::vector<torch::Tensor>& v)                      { for (const torch::Tensor& t : v) r->t.push_back(t); }
void put(Results* r, const std::tuple<torch::Tensor, torch::Tensor>& v)        { put(r, std::get<0>(v)); put(r, std::get<1>(v)); }
void put(Results* r, const std::tuple<torch::Tensor, torch::Tensor, torch::Tensor>& v) { put(r, std::get<0>(v)); put(r, std::get<1>(v)); put(r, std::get<2>(v)); }
void put(Results* r, const std::tuple<torch::Tensor, torch::Tensor, std::vector<torch::Tensor> >& v) { put(r, std::get<0>(v)); put(r, std::get<1>(v)); put(r, std::get<2>(v)); }
void put(Results* r, const std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, std::vector<torch::Tensor> >& v)
{ put(r, std::get<0>(v)); put(r, std::get<1>(v)); put(r, std::get<2>(v)); put(r, std::get<3>(v)); put(r, std::get<4>(v)); }

std::vector<torch::Tensor> import_list(const nvdr_ref_tensor* list, int n)
{
    std::vector<torch::Tensor> v;
    for (int i = 0; i < n; i++) v.push_back(import_tensor(&list[i]));
    return v;
}
}

#define NVDR_REF_TRY(BODY) \
    try { Results* res_ = new Results(); *out = 0; try { BODY } catch (...) { delete res_; throw; } *out = res_; return 0; } \
    catch (const std::exception& e) { g_error = e.what(); return 1; } \
    catch (...) { g_error = "unknown exception"; return 1; }
#define T(x) import_tensor(x)

extern "C"
{
const char* nvdr_ref_last_error(void) { return g_error.c_str(); }
int  nvdr_ref_get_log_level(void) { return FLAGS_caffe2_log_level; }
void nvdr_ref_set_log_level(int l) { FLAGS_caffe2_log_level = l; }
const char* nvdr_ref_log_text(void) { return g_log.c_str(); }       // every LOG line so far, whatever the level
const char* nvdr_ref_reference_root(void) { return NVDR_REFERENCE_ROOT; }
int  nvdr_ref_uses_fma(void)
{
#ifdef __FP_FAST_FMAF
    return 1;
#else
    return 0;
#endif
}

int     nvdr_ref_result_count(Results* r)                    { return (int)r->t.size(); }
int     nvdr_ref_result_defined(Results* r, int i)           { return r->t[i].defined() ? 1 : 0; }
int     nvdr_ref_result_ndim(Results* r, int i)              { return (int)r->t[i].dim(); }
int64_t nvdr_ref_result_size(Results* r, int i, int d)       { return r->t[i].size(d); }
int     nvdr_ref_result_dtype(Results* r, int i)             { return r->t[i].dtype() == torch::kInt32 ? 1 : 0; }
void    nvdr_ref_result_copy(Results* r, int i, void* dst)   { if (r->t[i].nbytes()) memcpy(dst, r->t[i].data_ptr(), r->t[i].nbytes()); }
void    nvdr_ref_result_free(Results* r)                     { delete r; }

//------------------------------------------------------------------------ rasterize
void* nvdr_ref_ctx_create(void)
{
    try { return new RasterizeCRStateWrapper(0); }
    catch (const std::exception& e) { g_error = e.what(); return 0; }
}
void  nvdr_ref_ctx_destroy(void* ctx) { delete (RasterizeCRStateWrapper*)ctx; }

int nvdr_ref_rasterize_fwd_cuda(void* ctx, const nvdr_ref_tensor* pos, const nvdr_ref_tensor* tri, int h, int w, const nvdr_ref_tensor* ranges, int peeling_idx, Results** out)
{ NVDR_REF_TRY( put(res_, rasterize_fwd_cuda(*(RasterizeCRStateWrapper*)ctx, T(pos), T(tri), std::tuple<int, int>(h, w), T(ranges), peeling_idx)); ) }
int nvdr_ref_rasterize_grad(const nvdr_ref_tensor* pos, const nvdr_ref_tensor* tri, const nvdr_ref_tensor* rast, const nvdr_ref_tensor* dy, Results** out)
{ NVDR_REF_TRY( put(res_, rasterize_grad(T(pos), T(tri), T(rast), T(dy))); ) }
int nvdr_ref_rasterize_grad_db(const nvdr_ref_tensor* pos, const nvdr_ref_tensor* tri, const nvdr_ref_tensor* rast, const nvdr_ref_tensor* dy, const nvdr_ref_tensor* ddb, Results** out)
{ NVDR_REF_TRY( put(res_, rasterize_grad_db(T(pos), T(tri), T(rast), T(dy), T(ddb))); ) }

// The rasterizer's internal surfaces after the last forward call of this context: triangle id + 1 ("colour") and
// the U32 depth, [N, Hpad, Wpad] with Hpad/Wpad = H/W rounded up to 8 (RasterImpl.cpp:87-99).  Lets tests compare the
// integer depth surface itself, not only which triangle won.
void nvdr_ref_ctx_surfaces(void* ctx, uint32_t* color, uint32_t* depth, size_t count)
{
    CR::CudaRaster* cr = ((RasterizeCRStateWrapper*)ctx)->cr;
    if (color) memcpy(color, cr->getColorBuffer(), count * 4);
    if (depth) memcpy(depth, cr->getDepthBuffer(), count * 4);
}

//------------------------------------------------------------------------ interpolate
int nvdr_ref_interpolate_fwd(const nvdr_ref_tensor* attr, const nvdr_ref_tensor* rast, const nvdr_ref_tensor* tri, Results** out)
{ NVDR_REF_TRY( put(res_, interpolate_fwd(T(attr), T(rast), T(tri))); ) }
int nvdr_ref_interpolate_fwd_da(const nvdr_ref_tensor* attr, const nvdr_ref_tensor* rast, const nvdr_ref_tensor* tri, const nvdr_ref_tensor* rast_db, int diff_all, const int* diff_list, int n_diff, Results** out)
{ NVDR_REF_TRY( std::vector<int> dl(diff_list, diff_list + n_diff); put(res_, interpolate_fwd_da(T(attr), T(rast), T(tri), T(rast_db), diff_all != 0, dl)); ) }
int nvdr_ref_interpolate_grad(const nvdr_ref_tensor* attr, const nvdr_ref_tensor* rast, const nvdr_ref_tensor* tri, const nvdr_ref_tensor* dy, Results** out)
{ NVDR_REF_TRY( put(res_, interpolate_grad(T(attr), T(rast), T(tri), T(dy))); ) }
int nvdr_ref_interpolate_grad_da(const nvdr_ref_tensor* attr, const nvdr_ref_tensor* rast, const nvdr_ref_tensor* tri, const nvdr_ref_tensor* dy, const nvdr_ref_tensor* rast_db, const nvdr_ref_tensor* dda, int diff_all, const int* diff_list, int n_diff, Results** out)
{ NVDR_REF_TRY( std::vector<int> dl(diff_list, diff_list + n_diff); put(res_, interpolate_grad_da(T(attr), T(rast), T(tri), T(dy), T(rast_db), T(dda), diff_all != 0, dl)); ) }

//------------------------------------------------------------------------ texture
void* nvdr_ref_mip_wrapper_empty(void) { return new TextureMipWrapper(); }
void  nvdr_ref_mip_wrapper_free(void* m) { delete (TextureMipWrapper*)m; }
int   nvdr_ref_texture_construct_mip(const nvdr_ref_tensor* tex, int max_mip_level, int cube_mode, void** wrapper, Results** out)
{ NVDR_REF_TRY( TextureMipWrapper* m = new TextureMipWrapper(texture_construct_mip(T(tex), max_mip_level, cube_mode != 0)); *wrapper = m; put(res_, m->mip); ) }
int nvdr_ref_texture_fwd(const nvdr_ref_tensor* tex, const nvdr_ref_tensor* uv, int filter_mode, int boundary_mode, Results** out)
{ NVDR_REF_TRY( put(res_, texture_fwd(T(tex), T(uv), filter_mode, boundary_mode)); ) }
int nvdr_ref_texture_fwd_mip(const nvdr_ref_tensor* tex, const nvdr_ref_tensor* uv, const nvdr_ref_tensor* uv_da, const nvdr_ref_tensor* bias, void* wrapper, const nvdr_ref_tensor* stack, int n_stack, int filter_mode, int boundary_mode, Results** out)
{ NVDR_REF_TRY( put(res_, texture_fwd_mip(T(tex), T(uv), T(uv_da), T(bias), *(TextureMipWrapper*)wrapper, import_list(stack, n_stack), filter_mode, boundary_mode)); ) }
int nvdr_ref_texture_grad_nearest(const nvdr_ref_tensor* tex, const nvdr_ref_tensor* uv, const nvdr_ref_tensor* dy, int filter_mode, int boundary_mode, Results** out)
{ NVDR_REF_TRY( put(res_, texture_grad_nearest(T(tex), T(uv), T(dy), filter_mode, boundary_mode)); ) }
int nvdr_ref_texture_grad_linear(const nvdr_ref_tensor* tex, const nvdr_ref_tensor* uv, const nvdr_ref_tensor* dy, int filter_mode, int boundary_mode, Results** out)
{ NVDR_REF_TRY( put(res_, texture_grad_linear(T(tex), T(uv), T(dy), filter_mode, boundary_mode)); ) }
int nvdr_ref_texture_grad_linear_mipmap_nearest(const nvdr_ref_tensor* tex, const nvdr_ref_tensor* uv, const nvdr_ref_tensor* dy, const nvdr_ref_tensor* uv_da, const nvdr_ref_tensor* bias, void* wrapper, const nvdr_ref_tensor* stack, int n_stack, int filter_mode, int boundary_mode, Results** out)
{ NVDR_REF_TRY( put(res_, texture_grad_linear_mipmap_nearest(T(tex), T(uv), T(dy), T(uv_da), T(bias), *(TextureMipWrapper*)wrapper, import_list(stack, n_stack), filter_mode, boundary_mode)); ) }
int nvdr_ref_texture_grad_linear_mipmap_linear(const nvdr_ref_tensor* tex, const nvdr_ref_tensor* uv, const nvdr_ref_tensor* dy, const nvdr_ref_tensor* uv_da, const nvdr_ref_tensor* bias, void* wrapper, const nvdr_ref_tensor* stack, int n_stack, int filter_mode, int boundary_mode, Results** out)
{ NVDR_REF_TRY( put(res_, texture_grad_linear_mipmap_linear(T(tex), T(uv), T(dy), T(uv_da), T(bias), *(TextureMipWrapper*)wrapper, import_list(stack, n_stack), filter_mode, boundary_mode)); ) }

//------------------------------------------------------------------------ antialias
void  nvdr_ref_hash_free(void* h) { delete (TopologyHashWrapper*)h; }
int   nvdr_ref_antialias_construct_topology_hash(const nvdr_ref_tensor* tri, void** wrapper, Results** out)
{ NVDR_REF_TRY( TopologyHashWrapper* h = new TopologyHashWrapper(antialias_construct_topology_hash(T(tri))); *wrapper = h; put(res_, h->ev_hash); ) }
int nvdr_ref_antialias_fwd(const nvdr_ref_tensor* color, const nvdr_ref_tensor* rast, const nvdr_ref_tensor* pos, const nvdr_ref_tensor* tri, void* hash, Results** out)
{ NVDR_REF_TRY( put(res_, antialias_fwd(T(color), T(rast), T(pos), T(tri), *(TopologyHashWrapper*)hash)); ) }
int nvdr_ref_antialias_grad(const nvdr_ref_tensor* color, const nvdr_ref_tensor* rast, const nvdr_ref_tensor* pos, const nvdr_ref_tensor* tri, const nvdr_ref_tensor* dy, const nvdr_ref_tensor* work_buffer, Results** out)
{ NVDR_REF_TRY( put(res_, antialias_grad(T(color), T(rast), T(pos), T(tri), T(dy), T(work_buffer))); ) }
}

//------------------------------------------------------------------------ debugging aid
// Prints the faulting address and instruction pointer (as a library offset for addr2line) on SIGSEGV; fibres
// run on their own stacks, so ordinary unwinders cannot walk them.
#include <signal.h>
#include <unistd.h>
#include <ucontext.h>
#include <dlfcn.h>
static void nvdr_ref_segv_handler(int, siginfo_t* si, void* uc_)
{
    ucontext_t* uc = (ucontext_t*)uc_;
    void* ip = (void*)uc->uc_mcontext.gregs[REG_RIP];
    Dl_info info;
    char buf[512];
    int n;
    if (dladdr(ip, &info) && info.dli_fname)
        n = snprintf(buf, sizeof(buf), "nvdr_ref: SIGSEGV at address %p, ip %p = %s+0x%lx (%s)\n", si->si_addr, ip, info.dli_fname,
                     (unsigned long)((char*)ip - (char*)info.dli_fbase), info.dli_sname ? info.dli_sname : "?");
    else
        n = snprintf(buf, sizeof(buf), "nvdr_ref: SIGSEGV at address %p, ip %p\n", si->si_addr, ip);
    if (write(2, buf, n) < 0) {}
    // Return addresses near the top of the faulting stack (a call through a null pointer leaves ip = 0).
    void** sp = (void**)uc->uc_mcontext.gregs[REG_RSP];
    for (int i = 0; i < 24; i++)
        if (dladdr(sp[i], &info) && info.dli_fname && strstr(info.dli_fname, "nvdr_ref"))
        {
            n = snprintf(buf, sizeof(buf), "  stack[%d] = %s+0x%lx (%s)\n", i, info.dli_fname, (unsigned long)((char*)sp[i] - (char*)info.dli_fbase), info.dli_sname ? info.dli_sname : "?");
            if (write(2, buf, n) < 0) {}
        }
    _exit(139);
}
extern "C" void nvdr_ref_install_segv_handler(void)
{
    static char altstack[1 << 16];
    stack_t ss; ss.ss_sp = altstack; ss.ss_size = sizeof(altstack); ss.ss_flags = 0;
    sigaltstack(&ss, 0);
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = nvdr_ref_segv_handler;
    sa.sa_flags = SA_ONSTACK | SA_SIGINFO;
    sigaction(SIGSEGV, &sa, 0);
}
