// nvdr_torch_host.cpp -- the compiled host layer of rasterize(), interpolate(), texture() and antialias().
//
// What csrc/torch/torch_rasterize.cpp:43-263 and csrc/torch/torch_interpolate.cpp:42-248 are in the reference -- validate the
// tensors, allocate outputs with torch, launch on torch's current stream -- for the two ops of the metric's graph and, further
// down, what torch_texture.cpp:98-716 and torch_antialias.cpp:68-241 are for the other two, PLUS the autograd nodes of all four
// (Python in the reference: nvdiffrast/torch/ops.py:75-90, 210-258, 261-375, 466-490).  With small batches the step time of this
// path is the host's; here a step of rasterize -> interpolate -> backward crosses into Python four times (two calls, their two
// returns) instead of running ~90 us of interpreter per step.
//
// HOST CODE ONLY: built with plain g++ against torch's headers (nothing is hipified, there is no device code); the kernels are
// reached through the C ABI of include/nvdr_hip.h, whose entry points arrive as addresses from _capi (the same library
// instance the rest of the package uses).  nvdiffrast_amd/torch/_plugin.py stays the path for everything unusual: every
// function here DECLINES (returns None) unless the call is the ordinary case, and the Python layer then produces the
// reference's error message or handles the rare mode.
//
// Backward of rasterize -> interpolate: nvdr_interpolate_rasterize_grad computes interpolate's and rasterize's gradients in one
// pass.  Both kernels ADD into the position gradient with atomics and rasterize_grad[_db] is linear in (dy, ddb), so
// InterpolateNode leaves its share of the position gradient with the RasterizeNode that made `rast` ("pending"), returns NO
// gradient for rast, and RasterizeNode adds whatever other consumers of rast contributed on top.  This is legal only when
// nobody can observe rast's gradient itself: a plain backward pass (no autograd.grad capture of rast), no hook and no
// retain_grad() on rast -- checked when the backward runs; otherwise the two separate kernels run (nvdr_interpolate_grad,
// nvdr_rasterize_grad), as in the reference.
#include <torch/extension.h>
#include <torch/csrc/autograd/function.h>
#include <torch/csrc/autograd/functions/utils.h>
#include <torch/csrc/autograd/graph_task.h>
#include <torch/csrc/autograd/saved_variable.h>
#ifndef NVDR_HOST_TEST_BUILD
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPFunctions.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>
#endif

#include <array>
#include <atomic>
#include <map>
#include <mutex>
#include <optional>
#include <unordered_map>

#include "../../include/nvdr_hip.h"

namespace {

using torch::autograd::Node;
using torch::autograd::SavedVariable;
using torch::autograd::variable_list;

// ---------------------------------------------------------------------------------------------------------------- C ABI
struct Api {
    decltype(&nvdr_last_error) last_error = nullptr;
    decltype(&nvdr_get_option) get_option = nullptr;
    decltype(&nvdr_log) log = nullptr;
    decltype(&nvdr_rasterize_scratch_bytes_pool) scratch_bytes_pool = nullptr;
    decltype(&nvdr_rasterize_pool_peak_offset) pool_peak_offset = nullptr;
    decltype(&nvdr_tile_flags_bytes) tile_flags_bytes = nullptr;
    decltype(&nvdr_rasterize_fwd) rasterize_fwd = nullptr;
    decltype(&nvdr_rasterize_grad) rasterize_grad = nullptr;
    decltype(&nvdr_interpolate_fwd) interpolate_fwd = nullptr;
    decltype(&nvdr_interpolate_grad) interpolate_grad = nullptr;
    decltype(&nvdr_interpolate_rasterize_grad) interpolate_rasterize_grad = nullptr;
    decltype(&nvdr_texture_mip_info) texture_mip_info = nullptr;
    decltype(&nvdr_texture_construct_mip) texture_construct_mip = nullptr;
    decltype(&nvdr_texture_fwd) texture_fwd = nullptr;
    decltype(&nvdr_texture_grad) texture_grad = nullptr;
    decltype(&nvdr_texture_grad_scratch_bytes) texture_grad_scratch_bytes = nullptr;
    decltype(&nvdr_antialias_fwd) antialias_fwd = nullptr;
    decltype(&nvdr_antialias_grad) antialias_grad = nullptr;
    bool ready = false;
} api;

template <class F> void take(F& slot, const py::dict& d, const char* name) {
    if (!d.contains(name)) throw std::runtime_error(std::string("nvdr host layer: no address for ") + name);
    slot = reinterpret_cast<F>(d[name].cast<uintptr_t>());
}

void init(const py::dict& addrs) {
    take(api.last_error, addrs, "nvdr_last_error");
    take(api.get_option, addrs, "nvdr_get_option");
    take(api.log, addrs, "nvdr_log");
    take(api.scratch_bytes_pool, addrs, "nvdr_rasterize_scratch_bytes_pool");
    take(api.pool_peak_offset, addrs, "nvdr_rasterize_pool_peak_offset");
    take(api.tile_flags_bytes, addrs, "nvdr_tile_flags_bytes");
    take(api.rasterize_fwd, addrs, "nvdr_rasterize_fwd");
    take(api.rasterize_grad, addrs, "nvdr_rasterize_grad");
    take(api.interpolate_fwd, addrs, "nvdr_interpolate_fwd");
    take(api.interpolate_grad, addrs, "nvdr_interpolate_grad");
    take(api.interpolate_rasterize_grad, addrs, "nvdr_interpolate_rasterize_grad");
    take(api.texture_mip_info, addrs, "nvdr_texture_mip_info");
    take(api.texture_construct_mip, addrs, "nvdr_texture_construct_mip");
    take(api.texture_fwd, addrs, "nvdr_texture_fwd");
    take(api.texture_grad, addrs, "nvdr_texture_grad");
    take(api.texture_grad_scratch_bytes, addrs, "nvdr_texture_grad_scratch_bytes");
    take(api.antialias_fwd, addrs, "nvdr_antialias_fwd");
    take(api.antialias_grad, addrs, "nvdr_antialias_grad");
    api.ready = true;
}

void check(int rc, const char* fn) {
    if (rc != 0) throw std::runtime_error(std::string(fn) + "(): " + api.last_error());
}

// ------------------------------------------------------------------------------------------------------------ switches
std::atomic<bool> g_fused{true};        // _plugin.set_fused_backward("auto" / "off")
std::atomic<bool> g_skip{true};         // _plugin.set_tile_skipping
std::atomic<bool> g_verify{false};      // _plugin.set_tile_flag_verification: the checking mode lives in Python -> decline
std::atomic<long long> g_n_fused{0}, g_n_fused_alone{0}, g_n_fused_plus{0}, g_n_separate{0}, g_n_fast_fwd{0};

enum { KIND_RAST = 0, KIND_ZERO = 1 };
constexpr int kMaxDiffAttrs = 32;       // csrc/common/interpolate.h:18

#ifndef NVDR_HOST_TEST_BUILD
inline bool on_gpu(const at::Tensor& t) { return t.is_cuda(); }
inline int index_of(const at::Tensor& t) { return (int)t.get_device(); }
inline void* stream_of(int dev) { return (void*)c10::hip::getCurrentHIPStream((c10::DeviceIndex)dev).stream(); }

inline bool capturing(void* stream) {
    hipStreamCaptureStatus s = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &s) != hipSuccess) { (void)hipGetLastError(); return false; }
    return s != hipStreamCaptureStatusNone;
}

struct OnDevice {                       // `with torch.cuda.device(dev)` that does nothing when dev is current already
    std::optional<c10::DeviceGuard> g;
    explicit OnDevice(const c10::Device& d) { if (c10::hip::current_device() != d.index()) g.emplace(d); }
};
#else
// tests/test_host_layer_logic.py builds this file a second time, into a temporary directory, with these four stand-ins: the
// bookkeeping and the autograd nodes then run over CPU tensors against a stub of the C ABI, where there is no GPU.  The
// module the package ships is never built this way (nvdiffrast_amd/_build.py) and has no CPU path.
inline bool on_gpu(const at::Tensor& t) { return t.is_cpu(); }
inline int index_of(const at::Tensor&) { return 0; }
inline void* stream_of(int) { return nullptr; }
inline bool capturing(void*) { return false; }
struct OnDevice { explicit OnDevice(const c10::Device&) {} };
#endif

// --------------------------------------------------------------------------------------------------- tile-flag records
// rasterize leaves with `rast` the tile occupancy the rasterizer wrote for it (include/nvdr_hip.h `tile_flags`), interpolate
// leaves the same flags with its outputs as "tiles of zeros".  A record is honoured while the tensor is what it was: the same
// StorageImpl, still alive (a weak reference keeps the struct's address from being reused), the same version counter, sizes,
// strides and offset.  What this cannot see, like the Python records it replaces: writes that do not bump the version counter.
struct TileRec {
    c10::weak_intrusive_ptr<c10::StorageImpl> st;
    int64_t version, offset;
    c10::SmallVector<int64_t, 5> sizes, strides;
    at::Tensor flags;
    int kind;
};
std::mutex g_rec_mu;
// (never destroyed: the records hold device tensors, and static destructors run after torch has begun to take its allocator down)
std::unordered_map<const c10::StorageImpl*, TileRec>& g_recs = *new std::unordered_map<const c10::StorageImpl*, TileRec>();

bool recordable(const at::Tensor& t) { return t.defined() && t.has_storage() && !t.is_inference(); }

void attach(const at::Tensor& t, const at::Tensor& flags, int kind) {
    if (!recordable(t)) return;
    std::lock_guard<std::mutex> l(g_rec_mu);
    for (auto it = g_recs.begin(); it != g_recs.end();)        // (a handful of live records: the sweep is a few compares)
        it = it->second.st.expired() ? g_recs.erase(it) : std::next(it);
    const c10::StorageImpl* key = t.storage().unsafeGetStorageImpl();
    TileRec r{t.storage().getWeakStorageImpl(), (int64_t)t._version(), t.storage_offset(), {}, {}, flags, kind};
    r.sizes.assign(t.sizes().begin(), t.sizes().end());
    r.strides.assign(t.strides().begin(), t.strides().end());
    g_recs.insert_or_assign(key, std::move(r));
}

at::Tensor flags_of(const at::Tensor& t, int kind) {
    if (!g_skip.load(std::memory_order_relaxed) || !recordable(t)) return {};
    std::lock_guard<std::mutex> l(g_rec_mu);
    auto it = g_recs.find(t.storage().unsafeGetStorageImpl());
    if (it == g_recs.end()) return {};
    const TileRec& r = it->second;
    if (r.kind != kind || r.st.expired() || r.version != (int64_t)t._version() || r.offset != t.storage_offset() ||
        !t.sizes().equals(r.sizes) || !t.strides().equals(r.strides))
        return {};
    return r.flags;
}

// ------------------------------------------------------------------------------------------------------ rasterizer state
// Per-context state (reference: csrc/torch/torch_types.h:15-23): the rasterizer's scratch memory and, while a DepthPeeler is
// active, the two depth surfaces; torch memory, grown on demand.  The capture rules are those of _plugin.RasterizeCRStateWrapper.
struct RasterState {
    int device_idx;
    at::Tensor scratch;
    bool has_clean = false;
    std::array<int64_t, 5> clean_layout{};
    bool captured = false;
    std::vector<at::Tensor> retired;
    int64_t reported_bytes = 0;
    std::map<std::pair<int64_t, int64_t>, int64_t> pools;
    at::Tensor depth, peel;
    at::Tensor last_flags;               // tile occupancy of the most recent call's output (the plugin-level entry point attaches it to that rast)
    explicit RasterState(int idx) : device_idx(idx) {}

    int64_t scratch_numel() const { return scratch.defined() ? scratch.numel() : 0; }
};

struct RasterizeNode;

// ------------------------------------------------------------------------------------------------------------- rasterize
struct RasterizeNode : public Node {
    SavedVariable pos_, tri_, rast_;
    at::Tensor flags_;
    bool grad_db_ = true;
    // identity of the tensors this call produced (interpolate checks that it is handed these very tensors, untouched)
    const c10::StorageImpl* rast_st_ = nullptr;
    const c10::StorageImpl* db_st_ = nullptr;
    int64_t rast_version_ = 0, db_version_ = 0;
    const void* tri_ptr_ = nullptr;
    int64_t T_ = 0, V_ = 0;
    // the share of the position gradient that interpolate's backward has prepared in the current backward pass
    at::Tensor pending_;
    int pending_task_ = -1;

    std::string name() const override { return "NvdrRasterizeBackward"; }
    void release_variables() override { pos_.reset_data(); tri_.reset_data(); rast_.reset_data(); flags_.reset(); pending_.reset(); }

    bool is_rast(const at::Tensor& t) const {
        return t.has_storage() && t.storage().unsafeGetStorageImpl() == rast_st_ && (int64_t)t._version() == rast_version_;
    }
    bool is_db(const at::Tensor& t) const {
        return t.has_storage() && t.storage().unsafeGetStorageImpl() == db_st_ && (int64_t)t._version() == db_version_;
    }
    // Nobody can look at rast's gradient in the backward pass that is running: no hook / retain_grad on rast or rast_db, and
    // the pass either executes everything (backward()) or executes this node without capturing its inputs.
    bool gradient_of_rast_unobservable() {
        if (!tensor_pre_hooks().empty() || !retains_grad_hooks().empty() || !pre_hooks().empty()) return false;
        const auto* info = torch::autograd::get_current_graph_task_exec_info();
        if (info == nullptr || info->empty()) return true;
        auto it = info->find(this);
        return it != info->end() && it->second.needed_ && !it->second.captures_;
    }

    variable_list apply(variable_list&& grads) override {
        at::Tensor d_rast = grads.size() > 0 ? grads[0] : at::Tensor();
        at::Tensor d_db = (grad_db_ && grads.size() > 1) ? grads[1] : at::Tensor();
        at::Tensor g_pos;
        if (pending_.defined() && pending_task_ == torch::autograd::get_current_graph_task_id()) g_pos = std::move(pending_);
        pending_.reset();
        pending_task_ = -1;
        const bool prepared = g_pos.defined();
        if (!d_rast.defined() && !d_db.defined()) {            // nothing else contributed: the prepared share is the gradient
            if (prepared) g_n_fused_alone++;
            variable_list out(1);
            out[0] = std::move(g_pos);
            return out;
        }
        at::Tensor pos = pos_.unpack(), tri = tri_.unpack(), rast = rast_.unpack(shared_from_this());
        const bool instance = pos.dim() > 2;
        const int64_t N = rast.size(0), H = rast.size(1), W = rast.size(2);
        auto shape_ok = [&](const at::Tensor& g) { return g.dim() == 4 && g.size(0) == N && g.size(1) == H && g.size(2) == W && g.size(3) == 4; };
        if (d_rast.defined()) {
            TORCH_CHECK(on_gpu(d_rast) && d_rast.device() == pos.device(), "rasterize_grad_db(): Inputs pos, tri, out, dy must reside on the same GPU device");
            TORCH_CHECK(d_rast.scalar_type() == at::kFloat, "rasterize_grad_db(): Inputs pos, out, dy must be float32 tensors");
            TORCH_CHECK(shape_ok(d_rast), "rasterize_grad_db(): dy must have shape [depth, height, width, 4]");
            d_rast = d_rast.contiguous();
        }
        if (d_db.defined()) {
            TORCH_CHECK(on_gpu(d_db) && d_db.device() == pos.device(), "rasterize_grad_db(): Inputs pos, tri, out, dy, ddb must reside on the same GPU device");
            TORCH_CHECK(d_db.scalar_type() == at::kFloat, "rasterize_grad_db(): Inputs pos, out, dy, ddb must be float32 tensors");
            TORCH_CHECK(shape_ok(d_db), "rasterize_grad_db(): ddb must have shape [depth, height, width, 4]");
            d_db = d_db.contiguous();
        }
        OnDevice guard(pos.device());
        if (!prepared) g_pos = at::zeros_like(pos);
        else g_n_fused_plus++;
        // (the flags were checked against this rast when the node was made; unpack() has just verified its version counter)
        const uint8_t* flags = (flags_.defined() && g_skip.load(std::memory_order_relaxed)) ? flags_.data_ptr<uint8_t>() : nullptr;
        check(api.rasterize_grad(pos.data_ptr<float>(), tri.data_ptr<int32_t>(), rast.data_ptr<float>(),
                                 d_rast.defined() ? d_rast.data_ptr<float>() : nullptr, d_db.defined() ? d_db.data_ptr<float>() : nullptr,
                                 (int)instance, (int)N, (int)(instance ? pos.size(1) : pos.size(0)), (int)tri.size(0), (int)H, (int)W,
                                 g_pos.data_ptr<float>(), flags, stream_of(index_of(pos))),
              d_db.defined() ? "rasterize_grad_db" : "rasterize_grad");
        g_n_separate++;
        variable_list out(1);
        out[0] = std::move(g_pos);
        return out;
    }
};

using RastPair = std::optional<std::tuple<at::Tensor, at::Tensor>>;

// torch_rasterize.cpp:43-166.  None = not the ordinary case (the Python layer takes over, with the reference's messages).
RastPair rasterize(const std::shared_ptr<RasterState>& st, const at::Tensor& pos, const at::Tensor& tri, int64_t height, int64_t width,
                   const at::Tensor& ranges, bool grad_db, int64_t peeling_idx) {
    if (!api.ready || g_verify.load(std::memory_order_relaxed)) return std::nullopt;
    const bool instance = pos.dim() > 2;
    if (!(on_gpu(pos) && tri.device() == pos.device() && index_of(pos) == st->device_idx && ranges.device().is_cpu() &&
          pos.scalar_type() == at::kFloat && tri.scalar_type() == at::kInt && ranges.scalar_type() == at::kInt &&
          pos.is_contiguous() && tri.is_contiguous() && ranges.is_contiguous() &&
          tri.dim() == 2 && tri.size(0) > 0 && tri.size(1) == 3 && height > 0 && width > 0))
        return std::nullopt;
    if (instance ? !(pos.dim() == 3 && pos.size(0) > 0 && pos.size(1) > 0 && pos.size(2) == 4)
                 : !(pos.dim() == 2 && pos.size(0) > 0 && pos.size(1) == 4 && ranges.dim() == 2 && ranges.size(0) > 0 && ranges.size(1) == 2))
        return std::nullopt;
    const c10::Device dev = pos.device();
    const int64_t N = instance ? pos.size(0) : ranges.size(0);
    const int64_t V = instance ? pos.size(1) : pos.size(0);
    const int64_t T = tri.size(0);
    int64_t max_tri = T;
    at::Tensor ranges_dev;
    if (!instance) {
        const int32_t* r = ranges.data_ptr<int32_t>();
        max_tri = 1;
        for (int64_t i = 0; i < N; i++) max_tri = std::max<int64_t>(max_tri, r[2 * i + 1]);
    }
    if (N > INT32_MAX || V > INT32_MAX || T > INT32_MAX || height > INT32_MAX || width > INT32_MAX) return std::nullopt;

    OnDevice guard(dev);
    void* stream = stream_of(st->device_idx);
    const bool cap = capturing(stream);
    // Scratch policy (include/nvdr_hip.h NVDR_OPT_SCRATCH_LIMIT_MB): the worst case while it is affordable -- nothing can
    // overflow, no host synchronisation, capturable -- otherwise a clip pool that grows on demand (one read-back per call).
    const size_t worst = api.scratch_bytes_pool((int)N, (int)max_tri, (int)height, (int)width, -1);
    const bool adaptive = worst > ((size_t)api.get_option(NVDR_OPT_SCRATCH_LIMIT_MB) << 20);
    if (adaptive && cap) return std::nullopt;                   // (an error; the Python layer words it)
    const long long worst_pool = 6 * (long long)max_tri;
    long long pool = -1;
    if (adaptive) {
        auto it = st->pools.find({N, max_tri});
        pool = it != st->pools.end() ? it->second : std::min<long long>(worst_pool, std::max<long long>(6 * 4096, max_tri / 4));
    }
    if (!instance) ranges_dev = ranges.to(dev);

    const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
    const auto u8 = at::TensorOptions().dtype(at::kByte).device(dev);
    at::Tensor out = at::empty({N, height, width, 4}, f32);
    at::Tensor out_db = at::empty({N, height, width, 4}, f32);
    at::Tensor flags = at::empty({(int64_t)api.tile_flags_bytes((int)N, (int)height, (int)width)}, u8);
    at::Tensor peel_in, depth_out;
    if (peeling_idx >= 0) {                                     // swapDepthAndPeel, RasterImpl.cpp:123-130
        const bool swap = peeling_idx > 0;
        if (swap) std::swap(st->depth, st->peel);
        const int64_t hp = (height + 7) & ~7ll, wp = (width + 7) & ~7ll;
        if (!st->depth.defined() || st->depth.size(0) != N || st->depth.size(1) != hp || st->depth.size(2) != wp || st->depth.device() != dev)
            st->depth = at::empty({N, hp, wp}, at::TensorOptions().dtype(at::kInt).device(dev));
        if (swap) peel_in = st->peel;
        depth_out = st->depth;
    }

    std::array<int64_t, 5> layout{};
    for (;;) {
        const size_t nbytes = api.scratch_bytes_pool((int)N, (int)max_tri, (int)height, (int)width, pool);
        layout = {N, max_tri, height, width, (int64_t)pool};
        if (!st->scratch.defined() || (size_t)st->scratch.numel() < nbytes || st->scratch.device() != dev) {
            if (st->captured && st->scratch.defined()) st->retired.push_back(st->scratch);      // a recorded graph may point to it
            st->scratch = at::empty({(int64_t)nbytes}, u8);
            st->has_clean = false;
            if ((int64_t)nbytes > st->reported_bytes) {         // RasterImpl.cpp:189-197: growth at 10 MB granularity, INFO
                const int64_t mb = ((((int64_t)nbytes - 1) >> 20) + 1 + 9) / 10 * 10;
                api.log(0, ("Internal buffers grown to " + std::to_string(mb) + " MB").c_str());
                st->reported_bytes = mb << 20;
            }
        }
        if (cap) st->captured = true;
        const bool clean = st->has_clean && st->clean_layout == layout && !st->captured;
        st->has_clean = false;                                  // re-armed below once the call has succeeded
        const int rc = api.rasterize_fwd(pos.data_ptr<float>(), tri.data_ptr<int32_t>(), instance ? nullptr : ranges_dev.data_ptr<int32_t>(),
                                         (int)instance, (int)N, (int)V, (int)T, (int)max_tri, (int)height, (int)width,
                                         peel_in.defined() ? (const uint32_t*)peel_in.data_ptr<int32_t>() : nullptr,
                                         depth_out.defined() ? (uint32_t*)depth_out.data_ptr<int32_t>() : nullptr,
                                         st->scratch.data_ptr<uint8_t>(), (size_t)st->scratch.numel(), (int)clean, pool,
                                         out.data_ptr<float>(), out_db.data_ptr<float>(), flags.data_ptr<uint8_t>(), stream);
        check(rc, "rasterize_fwd_cuda");
        if (!adaptive || pool < 0 || pool >= worst_pool) break;
        const int64_t off = (int64_t)api.pool_peak_offset((int)N, (int)max_tri, (int)height, (int)width, pool);
        const int64_t need = st->scratch.narrow(0, off, 4).view(at::kInt).item<int32_t>();   // the one host synchronisation of this mode
        if (need <= pool) break;
        const long long grown = std::min<long long>(worst_pool, need + need / 4 + 1024);
        st->pools[{N, max_tri}] = grown;
        TORCH_CHECK(grown > pool, "rasterize_fwd_cuda(): clip pool demand ", need, " exceeds the worst case of ", worst_pool, " slots per image");
        pool = grown;
        api.log(0, ("Clip pool grown to " + std::to_string(pool) + " sub-triangle slots per image").c_str());
    }
    st->has_clean = true;
    st->clean_layout = layout;
    attach(out, flags, KIND_RAST);
    st->last_flags = flags;
    g_n_fast_fwd++;

    if (torch::autograd::compute_requires_grad(pos)) {
        auto node = std::shared_ptr<RasterizeNode>(new RasterizeNode(), torch::autograd::deleteNode);
        node->set_next_edges(torch::autograd::collect_next_edges(pos));
        node->pos_ = SavedVariable(pos, false);
        node->tri_ = SavedVariable(tri, false);
        node->flags_ = flags;
        node->grad_db_ = grad_db;
        node->rast_st_ = out.storage().unsafeGetStorageImpl();
        node->db_st_ = out_db.storage().unsafeGetStorageImpl();
        node->tri_ptr_ = tri.data_ptr();
        node->T_ = T;
        node->V_ = V;
        torch::autograd::set_history(out, node);
        torch::autograd::set_history(out_db, node);
        node->rast_ = SavedVariable(out, true);
        node->rast_version_ = (int64_t)out._version();
        node->db_version_ = (int64_t)out_db._version();
    }
    return std::make_tuple(std::move(out), std::move(out_db));
}

// ----------------------------------------------------------------------------------------------------------- interpolate
struct InterpolateNode : public Node {
    SavedVariable attr_, rast_, tri_, rast_db_;
    at::Tensor flags_;
    bool with_da_ = false, diff_all_ = false;
    std::vector<int32_t> diff_;
    std::shared_ptr<RasterizeNode> origin_;      // the node that made rast (and rast_db), when interpolate was handed its very outputs

    std::string name() const override { return "NvdrInterpolateBackward"; }
    void release_variables() override {
        attr_.reset_data(); rast_.reset_data(); tri_.reset_data(); rast_db_.reset_data(); flags_.reset(); origin_.reset();
    }

    variable_list apply(variable_list&& grads) override {
        at::Tensor d_out = grads.size() > 0 ? grads[0] : at::Tensor();
        at::Tensor d_da = grads.size() > 1 ? grads[1] : at::Tensor();
        variable_list result(3);
        if (!d_out.defined() && !d_da.defined()) return result;
        at::Tensor attr = attr_.unpack(), rast = rast_.unpack(), tri = tri_.unpack();
        // differentials computed but their gradient unused: the plain gradient is the same (ops.py _InterpolateOp.backward)
        const bool da = with_da_ && d_da.defined() && d_da.numel() > 0;
        at::Tensor rast_db = da ? rast_db_.unpack() : at::Tensor();
        const bool attr_inst = attr.dim() > 2;
        const int64_t N = rast.size(0), H = rast.size(1), W = rast.size(2);
        const int64_t V = attr.size(attr_inst ? 1 : 0), A = attr.size(attr_inst ? 2 : 1), attr_n = attr_inst ? attr.size(0) : 1;
        const c10::Device dev = attr.device();
        OnDevice guard(dev);
        if (!d_out.defined()) d_out = at::zeros({N, H, W, A}, attr.options());
        TORCH_CHECK(on_gpu(d_out) && d_out.device() == dev && (!da || d_da.device() == dev),
                    "interpolate_grad(): Inputs attr, rast, tri, dy must reside on the same GPU device");
        TORCH_CHECK(d_out.scalar_type() == at::kFloat && (!da || d_da.scalar_type() == at::kFloat), "interpolate_grad(): Inputs attr, rast, dy must be float32 tensors");
        TORCH_CHECK(d_out.dim() == 4 && d_out.size(0) == N && d_out.size(1) == H && d_out.size(2) == W && d_out.size(3) == A,
                    "interpolate_grad(): dy must have shape [>0, height, width, >0]");
        d_out = d_out.contiguous();
        const int64_t D = da ? (diff_all_ ? A : (int64_t)diff_.size()) : 0;
        if (da) {
            TORCH_CHECK(d_da.dim() == 4 && d_da.size(0) == N && d_da.size(1) == H && d_da.size(2) == W && d_da.size(3) == 2 * D,
                        "interpolate_grad_da(): dda must have shape [>0, height, width, ?]");
            d_da = d_da.contiguous();
        }
        const uint8_t* flags = (flags_.defined() && g_skip.load(std::memory_order_relaxed)) ? flags_.data_ptr<uint8_t>() : nullptr;
        void* stream = stream_of(index_of(attr));
        const int32_t* lst = (da && !diff_all_ && !diff_.empty()) ? diff_.data() : nullptr;
        const int nlst = (da && !diff_all_) ? (int)diff_.size() : 0;

        RasterizeNode* org = origin_.get();
        if (org != nullptr && g_fused.load(std::memory_order_relaxed) && task_should_compute_output(1) && post_hooks().empty() &&
            org->gradient_of_rast_unobservable()) {
            at::Tensor pos = org->pos_.unpack();
            const bool pos_inst = pos.dim() > 2;
            at::Tensor g_attr, g_pos;
            const int task = torch::autograd::get_current_graph_task_id();
            if (org->pending_.defined() && org->pending_task_ == task) {          // another interpolation of this rast went first
                g_pos = org->pending_;
                g_attr = at::zeros_like(attr);
            } else {
                // both zero-initialised gradients from ONE buffer: one fill launch instead of two (attr.grad and pos.grad then are
                // views into one allocation, as on the Python path: _plugin.interpolate_rasterize_grad)
                const int64_t na = (attr.numel() + 3) & ~3ll;                      // keeps g_pos 16-byte aligned
                at::Tensor zeros = at::zeros({na + pos.numel()}, attr.options());
                g_attr = zeros.narrow(0, 0, attr.numel()).view(attr.sizes());
                g_pos = zeros.narrow(0, na, pos.numel()).view(pos.sizes());
            }
            check(api.interpolate_rasterize_grad(attr.data_ptr<float>(), rast.data_ptr<float>(), tri.data_ptr<int32_t>(), pos.data_ptr<float>(),
                                                 d_out.data_ptr<float>(), (int)attr_inst, (int)attr_n, (int)pos_inst,
                                                 (int)N, (int)V, (int)A, (int)tri.size(0), (int)H, (int)W,
                                                 da ? rast_db.data_ptr<float>() : nullptr, da ? d_da.data_ptr<float>() : nullptr,
                                                 (int)diff_all_, lst, nlst, (int)org->grad_db_,
                                                 g_attr.data_ptr<float>(), g_pos.data_ptr<float>(), nullptr, nullptr, flags, stream),
                  "interpolate_rasterize_grad");
            org->pending_ = std::move(g_pos);
            org->pending_task_ = task;
            g_n_fused++;
            result[0] = std::move(g_attr);
            return result;                                                         // (no gradient for rast / rast_db: see the file header)
        }
        at::Tensor g_attr = at::zeros_like(attr);
        at::Tensor g_rast = at::empty_like(rast);
        at::Tensor g_db = da ? at::empty_like(rast_db) : at::Tensor();
        check(api.interpolate_grad(attr.data_ptr<float>(), rast.data_ptr<float>(), tri.data_ptr<int32_t>(), d_out.data_ptr<float>(),
                                   da ? rast_db.data_ptr<float>() : nullptr, da ? d_da.data_ptr<float>() : nullptr,
                                   (int)attr_inst, (int)attr_n, (int)N, (int)V, (int)A, (int)tri.size(0), (int)H, (int)W,
                                   (int)diff_all_, lst, nlst, g_attr.data_ptr<float>(), g_rast.data_ptr<float>(),
                                   da ? g_db.data_ptr<float>() : nullptr, flags, stream),
              da ? "interpolate_grad_da" : "interpolate_grad");
        result[0] = std::move(g_attr);
        result[1] = std::move(g_rast);
        result[2] = std::move(g_db);
        return result;
    }
};

// torch_interpolate.cpp:42-132.  rast_db is None unless pixel differentials are requested (diff_all or a non-empty list).
RastPair interpolate(const at::Tensor& attr, const at::Tensor& rast, const at::Tensor& tri, const std::optional<at::Tensor>& rast_db_opt,
                     bool diff_all, const std::vector<int32_t>& diff_list) {
    if (!api.ready || g_verify.load(std::memory_order_relaxed)) return std::nullopt;
    const bool da = rast_db_opt.has_value() && rast_db_opt->defined() && (diff_all || !diff_list.empty());
    const at::Tensor rast_db = da ? *rast_db_opt : at::Tensor();
    const c10::Device dev = attr.device();
    if (!(on_gpu(attr) && rast.device() == dev && tri.device() == dev &&
          attr.scalar_type() == at::kFloat && rast.scalar_type() == at::kFloat && tri.scalar_type() == at::kInt &&
          attr.is_contiguous() && rast.is_contiguous() && tri.is_contiguous() &&
          rast.dim() == 4 && rast.size(0) > 0 && rast.size(1) > 0 && rast.size(2) > 0 && rast.size(3) == 4 &&
          tri.dim() == 2 && tri.size(0) > 0 && tri.size(1) == 3 &&
          (attr.dim() == 2 || attr.dim() == 3) && attr.size(0) > 0 && attr.size(1) > 0 && (attr.dim() == 2 || attr.size(2) > 0)))
        return std::nullopt;
    const bool attr_inst = attr.dim() > 2;
    const int64_t N = rast.size(0), H = rast.size(1), W = rast.size(2);
    if (attr_inst && !(attr.size(0) == N || attr.size(0) == 1)) return std::nullopt;
    if (da && !(rast_db.device() == dev && rast_db.scalar_type() == at::kFloat && rast_db.is_contiguous() && rast_db.dim() == 4 &&
                rast_db.size(0) == N && rast_db.size(1) == H && rast_db.size(2) == W && rast_db.size(3) == 4 &&
                (diff_all || (int)diff_list.size() <= kMaxDiffAttrs)))
        return std::nullopt;
    const int64_t V = attr.size(attr_inst ? 1 : 0), A = attr.size(attr_inst ? 2 : 1), attr_n = attr_inst ? attr.size(0) : 1;
    const int64_t D = da ? (diff_all ? A : (int64_t)diff_list.size()) : 0;
    if (N > INT32_MAX || V > INT32_MAX || A > INT32_MAX || tri.size(0) > INT32_MAX || H > INT32_MAX || W > INT32_MAX) return std::nullopt;

    at::Tensor flags = flags_of(rast, KIND_RAST);
    OnDevice guard(dev);
    at::Tensor out = at::empty({N, H, W, A}, attr.options());
    at::Tensor out_da = at::empty({N, H, W, 2 * D}, attr.options());
    check(api.interpolate_fwd(attr.data_ptr<float>(), rast.data_ptr<float>(), tri.data_ptr<int32_t>(), da ? rast_db.data_ptr<float>() : nullptr,
                              (int)attr_inst, (int)attr_n, (int)N, (int)V, (int)A, (int)tri.size(0), (int)H, (int)W,
                              (int)diff_all, (da && !diff_all) ? diff_list.data() : nullptr, (da && !diff_all) ? (int)diff_list.size() : 0,
                              out.data_ptr<float>(), da ? out_da.data_ptr<float>() : nullptr,
                              flags.defined() ? flags.data_ptr<uint8_t>() : nullptr, stream_of(index_of(attr))),
          da ? "interpolate_fwd_da" : "interpolate_fwd");
    if (flags.defined()) {
        // zeros are written where no triangle is visible: the rasterizer's empty tiles are tiles of zeros in both outputs, and
        // texture() need not read them there (while the tensors stay what they are now)
        attach(out, flags, KIND_ZERO);
        if (out_da.numel() > 0) attach(out_da, flags, KIND_ZERO);
    }
    g_n_fast_fwd++;

    if (torch::autograd::compute_requires_grad(attr, rast, rast_db)) {
        auto node = std::shared_ptr<InterpolateNode>(new InterpolateNode(), torch::autograd::deleteNode);
        node->set_next_edges(torch::autograd::collect_next_edges(attr, rast, rast_db));
        node->attr_ = SavedVariable(attr, false);
        node->rast_ = SavedVariable(rast, false);
        node->tri_ = SavedVariable(tri, false);
        if (da) node->rast_db_ = SavedVariable(rast_db, false);
        node->flags_ = flags;
        node->with_da_ = da;
        node->diff_all_ = diff_all;
        if (da && !diff_all) node->diff_ = diff_list;
        // Is `rast` (and `rast_db`) the untouched output of a rasterize() of this layer, over the same triangles and vertex count?
        if (rast.requires_grad()) {
            auto fn = rast.grad_fn();
            auto* rn = dynamic_cast<RasterizeNode*>(fn.get());
            if (rn != nullptr && rast.output_nr() == 0 && rn->is_rast(rast) && (!da || rn->is_db(rast_db)) &&
                rn->tri_ptr_ == tri.data_ptr() && rn->T_ == tri.size(0) && rn->V_ == V)
                node->origin_ = std::static_pointer_cast<RasterizeNode>(fn);
        }
        torch::autograd::set_history(out, node);
        torch::autograd::set_history(out_da, node);
    }
    return std::make_tuple(std::move(out), std::move(out_da));
}

// --------------------------------------------------------------------------------------------------------------- texture
// torch_texture.cpp:98-716 for 2-D and cube textures with the internal mip chain (a TextureMipWrapper's flat buffer); custom mip
// stacks -- per-level tensors with gradients of their own -- stay with the Python layer.
constexpr int kTexMaxLevels = 17;
enum { FILTER_NEAREST = 0, FILTER_LINEAR = 1, FILTER_LMN = 2, FILTER_LML = 3, BOUNDARY_CUBE = 0 };

struct TexGeom {
    bool cube;
    int64_t tn, th, tw, C;
};

bool tex_geom(const at::Tensor& tex, bool cube, TexGeom& g) {
    if (!cube) {
        if (!(tex.dim() == 4 && tex.size(0) > 0 && tex.size(1) > 0 && tex.size(2) > 0 && tex.size(3) > 0)) return false;
        g = {false, tex.size(0), tex.size(1), tex.size(2), tex.size(3)};
    } else {
        if (!(tex.dim() == 5 && tex.size(0) > 0 && tex.size(1) == 6 && tex.size(2) > 0 && tex.size(3) > 0 && tex.size(4) > 0 && tex.size(2) == tex.size(3))) return false;
        g = {true, tex.size(0), tex.size(2), tex.size(3), tex.size(4)};
    }
    return g.tw <= (1 << 16) && g.th <= (1 << 16) && g.tn <= INT32_MAX && g.C <= INT32_MAX;
}

struct MipLevels {
    int L = 0;
    int64_t off[kTexMaxLevels] = {0};
    int64_t total = 0;
};

bool mip_levels(const TexGeom& g, int64_t max_mip_level, MipLevels& m) {
    int lw[kTexMaxLevels], lh[kTexMaxLevels];
    m.L = api.texture_mip_info((int)g.tn, (int)g.th, (int)g.tw, (int)g.C, (int)g.cube, (int)max_mip_level, lw, lh, m.off, &m.total);
    return m.L >= 0;
}

// texture_construct_mip (torch_texture.cpp:98-169): the flat buffer of levels 1..L, or None (bad extents, unusual tensor: the
// Python layer words the error).
std::optional<at::Tensor> construct_mip(const at::Tensor& tex, int64_t max_mip_level, bool cube) {
    if (!api.ready || max_mip_level < -1) return std::nullopt;
    TexGeom g;
    if (!(on_gpu(tex) && tex.scalar_type() == at::kFloat && tex.is_contiguous() && tex_geom(tex, cube, g))) return std::nullopt;
    MipLevels m;
    if (!mip_levels(g, max_mip_level, m)) return std::nullopt;
    OnDevice guard(tex.device());
    at::Tensor mip = at::empty({m.total}, tex.options());
    check(api.texture_construct_mip(tex.data_ptr<float>(), (int)g.tn, (int)g.th, (int)g.tw, (int)g.C, (int)cube, (int)max_mip_level,
                                    mip.data_ptr<float>(), stream_of(index_of(tex))),
          "texture_construct_mip");
    return mip;
}

at::Tensor zero_flags_of(const at::Tensor& uv, const at::Tensor& uv_da) {
    at::Tensor f = flags_of(uv, KIND_ZERO);
    if (f.defined() && uv_da.defined() && uv_da.numel() > 0) {
        at::Tensor fd = flags_of(uv_da, KIND_ZERO);
        if (!fd.defined() || fd.data_ptr() != f.data_ptr()) return {};
    }
    return f;
}

struct TextureNode : public Node {
    SavedVariable tex_, uv_, uv_da_, bias_;
    at::Tensor mip_, flags_;
    int filter_ = 0, boundary_ = 0;
    int64_t max_mip_level_ = 0;
    bool scratch_ = true;

    std::string name() const override { return "NvdrTextureBackward"; }
    void release_variables() override { tex_.reset_data(); uv_.reset_data(); uv_da_.reset_data(); bias_.reset_data(); mip_.reset(); flags_.reset(); }

    variable_list apply(variable_list&& grads) override {
        variable_list result(4);
        at::Tensor dy = grads.size() > 0 ? grads[0] : at::Tensor();
        if (!dy.defined()) return result;
        at::Tensor tex = tex_.unpack(), uv = uv_.unpack();
        const bool mipmapped = filter_ == FILTER_LMN || filter_ == FILTER_LML;
        at::Tensor uv_da = mipmapped ? uv_da_.unpack() : at::Tensor(), bias = mipmapped ? bias_.unpack() : at::Tensor();
        const bool has_da = uv_da.defined() && uv_da.numel() > 0, has_bias = bias.defined() && bias.numel() > 0;
        const bool cube = boundary_ == BOUNDARY_CUBE;
        TexGeom g;
        TORCH_CHECK(tex_geom(tex, cube, g), "texture_grad(): tex must have shape[>0, >0, >0, >0]");
        const int64_t n = uv.size(0), H = uv.size(1), W = uv.size(2);
        const c10::Device dev = tex.device();
        TORCH_CHECK(on_gpu(dy) && dy.device() == dev, "texture_grad_linear_mipmap_linear(): Inputs dy must reside on the same GPU device");
        TORCH_CHECK(dy.scalar_type() == at::kFloat, "texture_grad_linear_mipmap_linear(): Inputs dy must be float32 tensors");
        TORCH_CHECK(dy.dim() == 4 && dy.size(0) == n && dy.size(1) == H && dy.size(2) == W && dy.size(3) == g.C,
                    "texture_grad_linear_mipmap_linear(): dy must have shape [minibatch_size, height, width, channels]");
        dy = dy.contiguous();
        OnDevice guard(dev);
        MipLevels m;
        const float* ptrs[kTexMaxLevels] = {nullptr};
        float* gptrs[kTexMaxLevels] = {nullptr};
        at::Tensor g_flat;
        if (mipmapped) {
            TORCH_CHECK(mip_levels(g, max_mip_level_, m) && mip_.defined() && mip_.numel() == m.total, "texture_grad(): wrapped mip tensor size mismatch");
            g_flat = at::zeros_like(mip_);
            for (int i = 1; i <= m.L; i++) { ptrs[i - 1] = mip_.data_ptr<float>() + m.off[i]; gptrs[i - 1] = g_flat.data_ptr<float>() + m.off[i]; }
        }
        at::Tensor g_tex = at::zeros_like(tex), g_uv, g_uv_da, g_bias;
        if (filter_ != FILTER_NEAREST) {
            g_uv = at::empty_like(uv);
            if (filter_ == FILTER_LML) {
                if (has_da) g_uv_da = at::empty_like(uv_da);
                if (has_bias) g_bias = at::empty_like(bias);
            }
        }
        // the flags the forward pass used, if uv / uv_da still are what interpolate wrote (asked again: the records go by version)
        const uint8_t* flags = nullptr;
        if (flags_.defined() && !cube) {
            at::Tensor now = zero_flags_of(uv, (mipmapped && has_da) ? uv_da : at::Tensor());
            if (now.defined() && now.data_ptr() == flags_.data_ptr()) flags = flags_.data_ptr<uint8_t>();
        }
        at::Tensor scratch;                                    // two-level reduction of constant-uv regions (include/nvdr_hip.h)
        if (scratch_ && filter_ != FILTER_NEAREST && !cube && n * H * W >= 4096)
            scratch = at::empty({(int64_t)(api.texture_grad_scratch_bytes((int)n, (int)H, (int)W, (int)g.C) / 4)}, at::TensorOptions().dtype(at::kInt).device(dev));
        check(api.texture_grad(tex.data_ptr<float>(), mipmapped ? ptrs : nullptr, m.L, uv.data_ptr<float>(),
                               (mipmapped && has_da) ? uv_da.data_ptr<float>() : nullptr, (mipmapped && has_bias) ? bias.data_ptr<float>() : nullptr,
                               dy.data_ptr<float>(), (int)g.tn, (int)g.th, (int)g.tw, (int)g.C, (int)n, (int)H, (int)W,
                               filter_, boundary_, (int)mipmapped, g_tex.data_ptr<float>(), mipmapped ? gptrs : nullptr,
                               g_uv.defined() ? g_uv.data_ptr<float>() : nullptr, g_uv_da.defined() ? g_uv_da.data_ptr<float>() : nullptr,
                               g_bias.defined() ? g_bias.data_ptr<float>() : nullptr,
                               scratch.defined() ? scratch.data_ptr() : nullptr, scratch.defined() ? (size_t)scratch.numel() * 4 : 0,
                               flags, stream_of(index_of(tex))),
              "texture_grad");
        result[0] = std::move(g_tex);
        result[1] = std::move(g_uv);
        result[2] = std::move(g_uv_da);
        result[3] = std::move(g_bias);
        return result;
    }
};

// texture_fwd / texture_fwd_mip (torch_texture.cpp:174-416).  mip: the wrapper's flat buffer (mipmapped filters), with the
// wrapper's max_mip_level, texture_size and cube_mode for the consistency check of torch_texture.cpp:309-310.
std::optional<at::Tensor> texture_op(const at::Tensor& tex, const at::Tensor& uv, const std::optional<at::Tensor>& uv_da_opt,
                                  const std::optional<at::Tensor>& bias_opt, const std::optional<at::Tensor>& mip_opt, int64_t max_mip_level,
                                  const std::vector<int64_t>& mip_texture_size, bool mip_cube, int64_t filter, int64_t boundary, bool grad_scratch) {
    if (!api.ready || g_verify.load(std::memory_order_relaxed) || filter < 0 || filter > 3 || boundary < 0 || boundary > 3) return std::nullopt;
    const bool cube = boundary == BOUNDARY_CUBE, mipmapped = filter == FILTER_LMN || filter == FILTER_LML;
    TexGeom g;
    if (!(on_gpu(tex) && uv.device() == tex.device() && tex.scalar_type() == at::kFloat && uv.scalar_type() == at::kFloat &&
          tex.is_contiguous() && uv.is_contiguous() && tex_geom(tex, cube, g) &&
          uv.dim() == 4 && uv.size(0) > 0 && uv.size(1) > 0 && uv.size(2) > 0 && uv.size(3) == (cube ? 3 : 2) &&
          (g.tn == 1 || g.tn == uv.size(0))))
        return std::nullopt;
    const c10::Device dev = tex.device();
    const int64_t n = uv.size(0), H = uv.size(1), W = uv.size(2);
    if (n > INT32_MAX || H > INT32_MAX || W > INT32_MAX) return std::nullopt;
    at::Tensor uv_da, bias, mip;
    bool has_da = false, has_bias = false;
    MipLevels m;
    const float* ptrs[kTexMaxLevels] = {nullptr};
    if (mipmapped) {
        if (uv_da_opt.has_value() && uv_da_opt->defined()) uv_da = *uv_da_opt;
        if (bias_opt.has_value() && bias_opt->defined()) bias = *bias_opt;
        has_da = uv_da.defined() && uv_da.numel() > 0;
        has_bias = bias.defined() && bias.numel() > 0;
        if (!(has_da || has_bias) || !mip_opt.has_value() || !mip_opt->defined() || max_mip_level < -1) return std::nullopt;
        mip = *mip_opt;
        auto ok = [&](const at::Tensor& t) { return t.device() == dev && t.scalar_type() == at::kFloat && t.is_contiguous(); };
        if (!ok(mip) || (has_da && !(ok(uv_da) && uv_da.dim() == 4 && uv_da.size(0) == n && uv_da.size(1) == H && uv_da.size(2) == W && uv_da.size(3) == (cube ? 6 : 4))) ||
            (has_bias && !(ok(bias) && bias.dim() == 3 && bias.size(0) == n && bias.size(1) == H && bias.size(2) == W)))
            return std::nullopt;
        if (mip_cube != cube || !tex.sizes().equals(mip_texture_size) || !mip_levels(g, max_mip_level, m) || mip.dim() != 1 || mip.size(0) != m.total)
            return std::nullopt;
        for (int i = 1; i <= m.L; i++) ptrs[i - 1] = mip.data_ptr<float>() + m.off[i];
    }
    at::Tensor flags = cube ? at::Tensor() : zero_flags_of(uv, (mipmapped && has_da) ? uv_da : at::Tensor());
    OnDevice guard(dev);
    at::Tensor out = at::empty({n, H, W, g.C}, tex.options());
    check(api.texture_fwd(tex.data_ptr<float>(), mipmapped ? ptrs : nullptr, m.L, uv.data_ptr<float>(),
                          (mipmapped && has_da) ? uv_da.data_ptr<float>() : nullptr, (mipmapped && has_bias) ? bias.data_ptr<float>() : nullptr,
                          (int)g.tn, (int)g.th, (int)g.tw, (int)g.C, (int)n, (int)H, (int)W, (int)filter, (int)boundary,
                          out.data_ptr<float>(), flags.defined() ? flags.data_ptr<uint8_t>() : nullptr, stream_of(index_of(tex))),
          mipmapped ? "texture_fwd_mip" : "texture_fwd");
    g_n_fast_fwd++;
    if (torch::autograd::compute_requires_grad(tex, uv, uv_da, bias)) {
        auto node = std::shared_ptr<TextureNode>(new TextureNode(), torch::autograd::deleteNode);
        node->set_next_edges(torch::autograd::collect_next_edges(tex, uv, uv_da, bias));
        node->tex_ = SavedVariable(tex, false);
        node->uv_ = SavedVariable(uv, false);
        if (mipmapped) {
            // (absent optional tensors are saved as empty ones, as the reference's ops.py does: ops.py:301-307)
            node->uv_da_ = SavedVariable(uv_da.defined() ? uv_da : at::empty({0}, tex.options()), false);
            node->bias_ = SavedVariable(bias.defined() ? bias : at::empty({0}, tex.options()), false);
            node->mip_ = mip;
        }
        node->flags_ = flags;
        node->filter_ = (int)filter;
        node->boundary_ = (int)boundary;
        node->max_mip_level_ = max_mip_level;
        node->scratch_ = grad_scratch;
        torch::autograd::set_history(out, node);
    }
    return out;
}

// ------------------------------------------------------------------------------------------------------------- antialias
struct AntialiasNode : public Node {
    SavedVariable color_, rast_, pos_, tri_;
    at::Tensor work_;
    double boost_ = 1.0;

    std::string name() const override { return "NvdrAntialiasBackward"; }
    void release_variables() override { color_.reset_data(); rast_.reset_data(); pos_.reset_data(); tri_.reset_data(); work_.reset(); }

    variable_list apply(variable_list&& grads) override {
        variable_list result(2);
        at::Tensor dy = grads.size() > 0 ? grads[0] : at::Tensor();
        if (!dy.defined()) return result;
        at::Tensor color = color_.unpack(), rast = rast_.unpack(), pos = pos_.unpack(), tri = tri_.unpack();
        const c10::Device dev = color.device();
        TORCH_CHECK(on_gpu(dy) && dy.device() == dev, "antialias_grad(): Inputs color, rast, pos, tri, dy, work_buffer must reside on the same GPU device");
        TORCH_CHECK(dy.scalar_type() == at::kFloat, "antialias_grad(): Inputs color, rast, pos, dy, work_buffer must be float32 tensors");
        TORCH_CHECK(dy.dim() == 4 && dy.sizes().equals(color.sizes()), "antialias_grad(): color and dy inputs must have same dimensions");
        dy = dy.contiguous();
        const bool instance = pos.dim() > 2;
        OnDevice guard(dev);
        at::Tensor g_color = at::empty_like(dy);                // the library copies dy into it
        at::Tensor g_pos = at::zeros_like(pos);
        check(api.antialias_grad(color.data_ptr<float>(), rast.data_ptr<float>(), pos.data_ptr<float>(), tri.data_ptr<int32_t>(),
                                 dy.data_ptr<float>(), work_.data_ptr<float>(), (size_t)work_.numel() * 4, (int)instance,
                                 (int)color.size(0), (int)(instance ? pos.size(1) : pos.size(0)), (int)tri.size(0),
                                 (int)color.size(1), (int)color.size(2), (int)color.size(3),
                                 g_color.data_ptr<float>(), g_pos.data_ptr<float>(), stream_of(index_of(color))),
              "antialias_grad");
        if (boost_ != 1.0) g_pos.mul_(boost_);
        result[0] = std::move(g_color);
        result[1] = std::move(g_pos);
        return result;
    }
};

// antialias_fwd (torch_antialias.cpp:68-155).  ev_hash: the TopologyHashWrapper's table.
using AaPair = std::optional<std::tuple<at::Tensor, at::Tensor>>;

// (out, work buffer), and the autograd node on `out` when `with_node`; the plugin-level entry point antialias_fwd -- which hands
// the work buffer to ITS caller's autograd function -- takes the pair without a node.
AaPair antialias_impl(const at::Tensor& color, const at::Tensor& rast, const at::Tensor& pos, const at::Tensor& tri,
                      const at::Tensor& ev_hash, double boost, bool with_node) {
    if (!api.ready || g_verify.load(std::memory_order_relaxed)) return std::nullopt;
    const c10::Device dev = color.device();
    const bool instance = pos.dim() > 2;
    if (!(on_gpu(color) && rast.device() == dev && pos.device() == dev && tri.device() == dev && ev_hash.device() == dev &&
          color.scalar_type() == at::kFloat && rast.scalar_type() == at::kFloat && pos.scalar_type() == at::kFloat &&
          tri.scalar_type() == at::kInt && ev_hash.scalar_type() == at::kInt &&
          color.is_contiguous() && rast.is_contiguous() && pos.is_contiguous() && tri.is_contiguous() && ev_hash.is_contiguous() &&
          color.dim() == 4 && color.size(0) > 0 && color.size(1) > 0 && color.size(2) > 0 && color.size(3) > 0 &&
          rast.dim() == 4 && rast.size(0) == color.size(0) && rast.size(1) == color.size(1) && rast.size(2) == color.size(2) && rast.size(3) == 4 &&
          tri.dim() == 2 && tri.size(0) > 0 && tri.size(1) == 3 && ev_hash.dim() == 1))
        return std::nullopt;
    if (instance ? !(pos.dim() == 3 && pos.size(0) == color.size(0) && pos.size(1) > 0 && pos.size(2) == 4)
                 : !(pos.dim() == 2 && pos.size(0) > 0 && pos.size(1) == 4))
        return std::nullopt;
    const int64_t N = color.size(0), H = color.size(1), W = color.size(2), C = color.size(3);
    if (N > INT32_MAX || H > INT32_MAX || W > INT32_MAX || C > INT32_MAX || N * H * W * 8 + 4 > (int64_t)INT32_MAX * 16) return std::nullopt;
    at::Tensor flags = flags_of(rast, KIND_RAST);
    OnDevice guard(dev);
    at::Tensor out = at::empty_like(color);                     // the library copies color into it
    at::Tensor work = at::empty({N * W * H * 8 + 4}, color.options());
    check(api.antialias_fwd(color.data_ptr<float>(), rast.data_ptr<float>(), pos.data_ptr<float>(), tri.data_ptr<int32_t>(),
                            ev_hash.data_ptr<int32_t>(), (size_t)ev_hash.numel() * 4, (int)instance, (int)N,
                            (int)(instance ? pos.size(1) : pos.size(0)), (int)tri.size(0), (int)H, (int)W, (int)C,
                            out.data_ptr<float>(), work.data_ptr<float>(), (size_t)work.numel() * 4,
                            flags.defined() ? flags.data_ptr<uint8_t>() : nullptr, stream_of(index_of(color))),
          "antialias_fwd");
    g_n_fast_fwd++;
    if (with_node && torch::autograd::compute_requires_grad(color, pos)) {
        auto node = std::shared_ptr<AntialiasNode>(new AntialiasNode(), torch::autograd::deleteNode);
        node->set_next_edges(torch::autograd::collect_next_edges(color, pos));
        node->color_ = SavedVariable(color, false);
        node->rast_ = SavedVariable(rast, false);
        node->pos_ = SavedVariable(pos, false);
        node->tri_ = SavedVariable(tri, false);
        node->work_ = work;
        node->boost_ = boost;
        torch::autograd::set_history(out, node);
    }
    return std::make_tuple(std::move(out), std::move(work));
}

std::optional<at::Tensor> antialias_op(const at::Tensor& color, const at::Tensor& rast, const at::Tensor& pos, const at::Tensor& tri,
                                       const at::Tensor& ev_hash, double boost) {
    AaPair r = antialias_impl(color, rast, pos, tri, ev_hash, boost, true);
    if (!r.has_value()) return std::nullopt;
    return std::get<0>(*r);
}

AaPair antialias_fwd_raw(const at::Tensor& color, const at::Tensor& rast, const at::Tensor& pos, const at::Tensor& tri, const at::Tensor& ev_hash) {
    return antialias_impl(color, rast, pos, tri, ev_hash, 1.0, false);
}

py::dict counters() {
    py::dict d;
    d["fused"] = (long long)g_n_fused;                  // interpolate backward passes that prepared a share of the position gradient
    d["fused_alone"] = (long long)g_n_fused_alone;      // rasterize backward passes that returned the prepared share as is
    d["fused_plus"] = (long long)g_n_fused_plus;        // ... that added other consumers' contributions on top of it
    d["separate"] = (long long)g_n_separate;            // nvdr_rasterize_grad launches
    d["fast_forward"] = (long long)g_n_fast_fwd;        // forward calls served here
    return d;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "compiled host layer of nvdiffrast_amd (rasterize / interpolate: validation, allocation, launch, autograd nodes)";
    py::class_<RasterState, std::shared_ptr<RasterState>>(m, "RasterState")
        .def(py::init<int>())
        .def_readonly("device_idx", &RasterState::device_idx)
        .def_readonly("captured", &RasterState::captured)
        .def_property_readonly("scratch_bytes", &RasterState::scratch_numel)
        .def_property_readonly("retired", [](const RasterState& s) { return (int)s.retired.size(); })
        // the depth surfaces of the last DepthPeeler pass and the clip-pool bookkeeping, for tests and tools
        .def_property_readonly("depth", [](const RasterState& s) -> std::optional<at::Tensor> { if (!s.depth.defined()) return std::nullopt; return s.depth; })
        .def_property_readonly("peel", [](const RasterState& s) -> std::optional<at::Tensor> { if (!s.peel.defined()) return std::nullopt; return s.peel; })
        .def_property_readonly("last_flags", [](const RasterState& s) -> std::optional<at::Tensor> { if (!s.last_flags.defined()) return std::nullopt; return s.last_flags; })
        .def("poison_scratch", [](RasterState& s, int value) { if (s.scratch.defined()) s.scratch.fill_(value); s.has_clean = false; })
        .def("set_scratch", [](RasterState& s, const at::Tensor& t) { s.scratch = t; s.has_clean = false; })
        .def("set_pool", [](RasterState& s, int64_t n, int64_t max_tri, int64_t slots) { s.pools[{n, max_tri}] = slots; })
        .def("get_pool", [](const RasterState& s, int64_t n, int64_t max_tri) -> int64_t { auto it = s.pools.find({n, max_tri}); return it == s.pools.end() ? -1 : it->second; });
    m.def("init", &init);
    m.def("ready", []() { return api.ready; });
    m.def("rasterize", &rasterize, py::call_guard<py::gil_scoped_release>());
    m.def("interpolate", &interpolate, py::call_guard<py::gil_scoped_release>());
    m.def("construct_mip", &construct_mip, py::call_guard<py::gil_scoped_release>());
    m.def("texture", &texture_op, py::call_guard<py::gil_scoped_release>());
    m.def("antialias", &antialias_op, py::call_guard<py::gil_scoped_release>());
    m.def("antialias_fwd_raw", &antialias_fwd_raw, py::call_guard<py::gil_scoped_release>());
    m.def("attach", &attach);
    m.def("flags_of", [](const at::Tensor& t, int kind) -> std::optional<at::Tensor> {
        at::Tensor f = flags_of(t, kind);
        if (!f.defined()) return std::nullopt;
        return f;
    });
    m.def("set_fused", [](bool on) { g_fused = on; });
    m.def("set_skip", [](bool on) { g_skip = on; });
    m.def("set_verify", [](bool on) { g_verify = on; });
    m.def("counters", &counters);
    m.attr("KIND_RAST") = (int)KIND_RAST;
    m.attr("KIND_ZERO") = (int)KIND_ZERO;
}
