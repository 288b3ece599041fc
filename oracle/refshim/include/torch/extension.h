// Miniature stand-in for <torch/extension.h>: the subset of the ATen tensor API that the reference's
// glue files (csrc/torch/torch_{rasterize,interpolate,texture,antialias}.cpp) use, over host memory, so that
// those files compile UNMODIFIED with g++ and run on the CPU.  "CUDA" tensors are host tensors with a device
// tag; the kernels they feed run in the CUDA-on-CPU shim (nvdr_cuda_shim.h).
// TEST INFRASTRUCTURE ONLY — used to build oracle/_ref/libnvdr_ref.so, never by the product.
#pragma once
#include <cuda_runtime.h>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <memory>
#include <vector>
#include <tuple>
#include <string>
#include <sstream>
#include <stdexcept>
#include <initializer_list>
#include <algorithm>

namespace c10
{
    class Error : public std::runtime_error { public: explicit Error(const std::string& m) : std::runtime_error(m) {} };
    enum class DeviceType : int { CPU = 0, CUDA = 1 };
    struct Device
    {
        DeviceType t; int idx;
        Device(DeviceType t_ = DeviceType::CPU, int idx_ = -1) : t(t_), idx(idx_) {}
        DeviceType type(void) const { return t; }
        int index(void) const { return idx; }
        bool is_cuda(void) const { return t == DeviceType::CUDA; }
        bool operator==(const Device& o) const { return t == o.t && idx == o.idx; }
        bool operator!=(const Device& o) const { return !(*this == o); }
    };
    enum class ScalarType : int { Float = 0, Int = 1, Long = 2, Byte = 3, Undefined = 4 };
    inline size_t elementSize(ScalarType t) { return t == ScalarType::Long ? 8 : (t == ScalarType::Byte ? 1 : 4); }
    template<class T> struct optional_lite
    {
        bool has; T v;
        optional_lite(void) : has(false), v() {}
        optional_lite(const T& v_) : has(true), v(v_) {}
        bool has_value(void) const { return has; }
        const T& value(void) const { return v; }
    };
    template<class T> class ArrayRef
    {
    public:
        ArrayRef(void) : p(0), n(0) {}
        ArrayRef(const T* p_, size_t n_) : p(p_), n(n_) {}
        ArrayRef(const std::vector<T>& v) : p(v.data()), n(v.size()) {}
        ArrayRef(std::initializer_list<T> il) : p(il.begin()), n(il.size()) {}
        size_t size(void) const { return n; }
        bool empty(void) const { return n == 0; }
        const T& operator[](size_t i) const { return p[i]; }
        const T* begin(void) const { return p; }
        const T* end(void) const { return p + n; }
        std::vector<T> vec(void) const { return std::vector<T>(p, p + n); }
        bool operator==(const ArrayRef& o) const { return n == o.n && std::equal(p, p + n, o.p); }
        bool operator!=(const ArrayRef& o) const { return !(*this == o); }
    private:
        const T* p; size_t n;
    };
    typedef ArrayRef<int64_t> IntArrayRef;

    namespace detail
    {
        inline void cat(std::ostringstream&) {}
        template<class A, class... R> inline void cat(std::ostringstream& s, const A& a, const R&... r) { s << a; cat(s, r...); }
        template<class... A> inline std::string str(const A&... a) { std::ostringstream s; cat(s, a...); return s.str(); }
    }
}

// c10 logging: LOG(INFO) << ... prints to stderr when the severity is at or above FLAGS_caffe2_log_level
// (INFO = 0, WARNING = 1; c10's default level is WARNING), torch_bindings.cpp:50-51.
extern int FLAGS_caffe2_log_level;
namespace c10
{
    enum { LOG_INFO = 0, LOG_WARNING = 1, LOG_ERROR = 2 };
    struct LogLine
    {
        int sev; std::ostringstream s;
        explicit LogLine(int sev_) : sev(sev_) {}
        ~LogLine(void);
        template<class A> LogLine& operator<<(const A& a) { s << a; return *this; }
    };
}
#define LOG(sev) c10::LogLine(c10::LOG_##sev)

#define TORCH_CHECK(cond, ...) if (!(cond)) { throw c10::Error(c10::detail::str(__VA_ARGS__)); }
#define AT_ASSERTM(cond, ...)  TORCH_CHECK(cond, __VA_ARGS__)

namespace at
{
    using c10::Device; using c10::DeviceType; using c10::ScalarType; using c10::ArrayRef; using c10::IntArrayRef;

    struct TensorOptions
    {
        ScalarType dt; Device dev;
        TensorOptions(void) : dt(ScalarType::Float), dev(DeviceType::CPU, -1) {}
        TensorOptions dtype(ScalarType t) const { TensorOptions o = *this; o.dt = t; return o; }
        TensorOptions device(DeviceType t) const { TensorOptions o = *this; o.dev = Device(t, t == DeviceType::CUDA ? 0 : -1); return o; }
        TensorOptions device(Device d) const { TensorOptions o = *this; o.dev = d; return o; }
        TensorOptions device(DeviceType t, int idx) const { TensorOptions o = *this; o.dev = Device(t, idx); return o; }
    };

    class Tensor
    {
    public:
        Tensor(void) : dt(ScalarType::Undefined), dev(), nelem(0) {}

        static Tensor make(IntArrayRef shape_, const TensorOptions& o, bool zero)
        {
            Tensor t;
            t.shape = shape_.vec();
            t.dt = o.dt; t.dev = o.dev;
            t.nelem = 1;
            for (int64_t s : t.shape) { TORCH_CHECK(s >= 0, "negative dimension"); t.nelem *= s; }
            size_t bytes = (size_t)t.nelem * c10::elementSize(t.dt);
            void* p = 0;
            if (posix_memalign(&p, 256, bytes ? bytes : 256)) throw c10::Error("out of memory");
            if (zero) memset(p, 0, bytes); else memset(p, 0xcd, bytes);     // torch::empty is uninitialised
            t.store = std::shared_ptr<void>(p, free);
            return t;
        }

        bool defined(void) const { return (bool)store; }
        IntArrayRef sizes(void) const { return IntArrayRef(shape); }
        int64_t size(int64_t d) const
        {
            int64_t n = (int64_t)shape.size();
            if (d < 0) d += n;
            TORCH_CHECK(d >= 0 && d < n, "Dimension out of range (expected to be in range of [", -n, ", ", n - 1, "], but got ", d, ")");
            return shape[(size_t)d];
        }
        int64_t dim(void) const { return (int64_t)shape.size(); }
        int64_t numel(void) const { return nelem; }
        size_t nbytes(void) const { return (size_t)nelem * c10::elementSize(dt); }
        ScalarType dtype(void) const { return dt; }
        ScalarType scalar_type(void) const { return dt; }
        Device device(void) const { return dev; }
        int get_device(void) const { return dev.idx; }
        bool is_cuda(void) const { return dev.is_cuda(); }
        bool is_contiguous(void) const { return true; }
        Tensor contiguous(void) const { return *this; }
        Tensor detach(void) const { return *this; }
        Tensor clone(void) const
        {
            Tensor t = make(IntArrayRef(shape), options(), false);
            memcpy(t.store.get(), store.get(), nbytes());
            return t;
        }
        TensorOptions options(void) const { TensorOptions o; o.dt = dt; o.dev = dev; return o; }
        template<class T> T* data_ptr(void) const
        {
            TORCH_CHECK(defined(), "data_ptr() of an undefined tensor");
            TORCH_CHECK(sizeof(T) == c10::elementSize(dt), "data_ptr<T>(): element size mismatch");
            return (T*)store.get();
        }
        void* data_ptr(void) const { return store.get(); }

    private:
        std::shared_ptr<void> store;
        std::vector<int64_t>  shape;
        ScalarType            dt;
        Device                dev;
        int64_t               nelem;
    };

    inline c10::optional_lite<Device> device_of(const Tensor& t) { return t.defined() ? c10::optional_lite<Device>(t.device()) : c10::optional_lite<Device>(); }

    inline Tensor empty(IntArrayRef s, const TensorOptions& o = TensorOptions())   { return Tensor::make(s, o, false); }
    inline Tensor zeros(IntArrayRef s, const TensorOptions& o = TensorOptions())   { return Tensor::make(s, o, true); }
    inline Tensor empty_like(const Tensor& t)                                      { return Tensor::make(t.sizes(), t.options(), false); }
    inline Tensor zeros_like(const Tensor& t)                                      { return Tensor::make(t.sizes(), t.options(), true); }

    namespace cuda
    {
        struct OptionalCUDAGuard
        {
            OptionalCUDAGuard(void) {}
            explicit OptionalCUDAGuard(int) {}
            explicit OptionalCUDAGuard(c10::optional_lite<Device>) {}
        };
        struct CUDAStream { operator cudaStream_t(void) const { return 0; } cudaStream_t stream(void) const { return 0; } };
        inline CUDAStream getCurrentCUDAStream(int = -1) { return CUDAStream(); }
        // True iff every tensor is a CUDA tensor on one device (ATen/cuda/CUDAUtils.h).
        inline bool check_device(ArrayRef<Tensor> ts)
        {
            if (ts.empty()) return true;
            Device d = ts[0].device();
            for (const Tensor& t : ts)
                if (!t.defined() || !t.is_cuda() || t.device() != d) return false;
            return true;
        }
    }
}

namespace c10 { namespace cuda { using at::cuda::OptionalCUDAGuard; } }

namespace torch
{
    using at::Tensor; using at::TensorOptions; using at::empty; using at::zeros; using at::empty_like; using at::zeros_like;
    using at::device_of;
    const c10::DeviceType kCUDA = c10::DeviceType::CUDA;
    const c10::DeviceType kCPU  = c10::DeviceType::CPU;
    const c10::ScalarType kFloat32 = c10::ScalarType::Float;
    const c10::ScalarType kFloat   = c10::ScalarType::Float;
    const c10::ScalarType kInt32   = c10::ScalarType::Int;
    const c10::ScalarType kInt     = c10::ScalarType::Int;
    const c10::ScalarType kInt64   = c10::ScalarType::Long;
}
using at::device_of;
