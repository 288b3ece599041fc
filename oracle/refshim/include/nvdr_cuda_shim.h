// CUDA-on-CPU shim: just enough of the CUDA C++ language surface and runtime API for the UNMODIFIED
// reference sources under /root/reference/csrc to compile with g++ and run on the host.
//
// TEST INFRASTRUCTURE ONLY (part of oracle/).  Nothing in the product (nvdiffrast_amd/) includes or links
// this; it exists so that the repo's CPU oracle can be pinned to the reference's own code
// (oracle/_ref/libnvdr_ref.so, built by oracle/refshim/build.py from the sources where they lie).
//
// Execution model: every CUDA thread of a block is a fibre (own stack, cooperative switch written in
// assembly, see shim_runtime.cpp) on ONE OS thread.  __syncthreads / __syncwarp / __ballot_sync / ... are
// scheduling points with CUDA's semantics for 32-lane warps; exited threads count as arrived.  Blocks of a
// grid run one after another, so atomics are plain read-modify-writes and every floating-point atomic sum
// has a deterministic order (block-major, then the fibre schedule).
//
// __CUDA_ARCH__ is left undefined, so the reference's common.h:246-260 selects its plain-atomicAdd branch.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>

// __CUDACC__ is defined on the command line for the .cu translation units only (as nvcc would), so that the
// reference's host files see the host half of its headers.

//------------------------------------------------------------------------ qualifiers
#define __global__
#define __device__
#define __host__
#define __constant__
#define __shared__          thread_local
#define __forceinline__     inline __attribute__((always_inline))
#define __inline__          inline
#define __launch_bounds__(...)
#define __restrict__        __restrict

//------------------------------------------------------------------------ vector types
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int2   { int x, y; };
struct int3   { int x, y, z; };
struct int4   { int x, y, z, w; };
struct uint2  { unsigned int x, y; };
struct uint3  { unsigned int x, y, z; };
struct uint4  { unsigned int x, y, z, w; };
struct dim3
{
    unsigned int x, y, z;
    constexpr dim3(unsigned int x_ = 1, unsigned int y_ = 1, unsigned int z_ = 1) : x(x_), y(y_), z(z_) {}
    constexpr dim3(uint3 v) : x(v.x), y(v.y), z(v.z) {}
};

static inline float2 make_float2(float x, float y)                       { float2 r = {x, y}; return r; }
static inline float3 make_float3(float x, float y, float z)              { float3 r = {x, y, z}; return r; }
static inline float4 make_float4(float x, float y, float z, float w)     { float4 r = {x, y, z, w}; return r; }
static inline int2   make_int2(int x, int y)                             { int2 r = {x, y}; return r; }
static inline int3   make_int3(int x, int y, int z)                      { int3 r = {x, y, z}; return r; }
static inline int4   make_int4(int x, int y, int z, int w)               { int4 r = {x, y, z, w}; return r; }
static inline uint2  make_uint2(unsigned x, unsigned y)                  { uint2 r = {x, y}; return r; }
static inline uint3  make_uint3(unsigned x, unsigned y, unsigned z)      { uint3 r = {x, y, z}; return r; }
static inline uint4  make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 r = {x, y, z, w}; return r; }

//------------------------------------------------------------------------ built-in index variables
// (Statically initialised thread-locals; the build passes -fno-extern-tls-init so that uses in other
// translation units are plain TLS accesses instead of calls through the C++11 init wrapper.)
extern thread_local uint3 threadIdx;
extern thread_local uint3 blockIdx;
extern thread_local dim3  blockDim;
extern thread_local dim3  gridDim;
static const int warpSize = 32;

// Dynamic shared memory of the one kernel that uses it (texture_kernel.cu:859).
extern thread_local float s_texelAccum[];

//------------------------------------------------------------------------ scheduling points
namespace nvdr_shim
{
    enum { OP_SYNC = 0, OP_BALLOT, OP_ANY, OP_ALL, OP_MATCH_ANY, OP_SHFL };
    void     block_barrier(void);
    unsigned warp_collective(int op, unsigned mask, unsigned value, unsigned aux);
    unsigned lane_id(void);
}
static inline void     __syncthreads(void)                              { nvdr_shim::block_barrier(); }
static inline void     __syncwarp(unsigned mask = 0xffffffffu)          { nvdr_shim::warp_collective(nvdr_shim::OP_SYNC, mask, 0, 0); }
static inline unsigned __ballot_sync(unsigned mask, int pred)           { return nvdr_shim::warp_collective(nvdr_shim::OP_BALLOT, mask, pred != 0, 0); }
static inline int      __any_sync(unsigned mask, int pred)              { return (int)nvdr_shim::warp_collective(nvdr_shim::OP_ANY, mask, pred != 0, 0); }
static inline int      __all_sync(unsigned mask, int pred)              { return (int)nvdr_shim::warp_collective(nvdr_shim::OP_ALL, mask, pred != 0, 0); }
static inline unsigned __match_any_sync(unsigned mask, unsigned value)  { return nvdr_shim::warp_collective(nvdr_shim::OP_MATCH_ANY, mask, value, 0); }
static inline unsigned __match_any_sync(unsigned mask, int value)       { return nvdr_shim::warp_collective(nvdr_shim::OP_MATCH_ANY, mask, (unsigned)value, 0); }
static inline unsigned __shfl_sync(unsigned mask, unsigned v, int src, int width = 32)
{ (void)width; return nvdr_shim::warp_collective(nvdr_shim::OP_SHFL, mask, v, (unsigned)src & 31u); }
static inline int      __shfl_sync(unsigned mask, int v, int src, int width = 32)
{ (void)width; return (int)nvdr_shim::warp_collective(nvdr_shim::OP_SHFL, mask, (unsigned)v, (unsigned)src & 31u); }
static inline void     __threadfence(void)                              {}
static inline void     __threadfence_block(void)                        {}

//------------------------------------------------------------------------ bit casts and integer intrinsics
static inline int          __float_as_int(float f)          { int i; memcpy(&i, &f, 4); return i; }
static inline unsigned     __float_as_uint(float f)         { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float        __int_as_float(int i)            { float f; memcpy(&f, &i, 4); return f; }
static inline float        __uint_as_float(unsigned i)      { float f; memcpy(&f, &i, 4); return f; }
static inline double       __longlong_as_double(long long i){ double d; memcpy(&d, &i, 8); return d; }
static inline long long    __double_as_longlong(double d)   { long long i; memcpy(&i, &d, 8); return i; }
static inline int          __double2loint(double d)         { long long i = __double_as_longlong(d); return (int)(unsigned)(i & 0xffffffffll); }
static inline int          __double2hiint(double d)         { long long i = __double_as_longlong(d); return (int)(unsigned)((unsigned long long)i >> 32); }
static inline double       __hiloint2double(int hi, int lo) { long long i = (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo); return __longlong_as_double(i); }
static inline int          __popc(unsigned v)               { return __builtin_popcount(v); }
static inline int          __popcll(unsigned long long v)   { return __builtin_popcountll(v); }
static inline int          __clz(int v)                     { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int          __clzll(long long v)             { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline int          __ffs(int v)                     { return __builtin_ffs(v); }
static inline int          __ffsll(long long v)             { return __builtin_ffsll(v); }
static inline unsigned     __brev(unsigned v)
{
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
    return __builtin_bswap32(v);
}
static inline int          __mul24(int a, int b)            { return a * b; }   // callers stay inside 24 bits
static inline unsigned     __umul24(unsigned a, unsigned b) { return a * b; }
static inline unsigned     __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline int          __mulhi(int a, int b)            { return (int)(((long long)a * b) >> 32); }

//------------------------------------------------------------------------ float intrinsics
// __saturatef: clamp to [0,1], NaN -> 0 (CUDA math API).
static inline float __saturatef(float x)        { return (x >= 0.f) ? ((x <= 1.f) ? x : 1.f) : 0.f; }
// __float2int_rd: round towards -inf, saturating, NaN -> 0.
static inline int   __float2int_rd(float x)
{
    if (x != x) return 0;
    float f = floorf(x);
    if (f >= 2147483648.f) return 0x7fffffff;
    if (f <= -2147483648.f) return (int)0x80000000;
    return (int)f;
}
// __frcp_rz: reciprocal rounded towards zero.  1/x in double is exact to 53 bits and a float reciprocal
// can never lie within 2^-29 relative of a float boundary unless it is exactly representable, so
// truncating the double quotient gives the correctly rounded-towards-zero float.
static inline float __frcp_rz(float x)
{
    double d = 1.0 / (double)x;
    float f = (float)d;
    if (f != f || isinf(f)) { if (isinf(f) && !isinf(d) && d == d) f = copysignf(3.402823466e+38f, f); return f; }
    if (fabs((double)f) > fabs(d)) f = nextafterf(f, 0.f);
    return f;
}
// __log2f: hardware approximation (lg2.approx.f32, max abs error 2^-22 near 1, 1-2 ulp elsewhere) — the
// host's log2f stands in for it; results differ from any GPU in the last bits (documented ambiguity).
// (glibc's <math.h> already declares extern __log2f/__expf, hence macros.)
static inline float nvdr_shim_log2f(float x)    { return log2f(x); }
static inline float nvdr_shim_expf(float x)     { return expf(x); }
#define __log2f(x) nvdr_shim_log2f(x)
#define __expf(x)  nvdr_shim_expf(x)
static inline float __fdividef(float a, float b){ return a / b; }
static inline float rsqrtf(float x)             { return 1.f / sqrtf(x); }

// CUDA's global-namespace integer/float min/max overloads.
static inline int                min(int a, int b)                               { return a < b ? a : b; }
static inline int                max(int a, int b)                               { return a > b ? a : b; }
static inline unsigned           min(unsigned a, unsigned b)                     { return a < b ? a : b; }
static inline unsigned           max(unsigned a, unsigned b)                     { return a > b ? a : b; }
static inline long long          min(long long a, long long b)                   { return a < b ? a : b; }
static inline long long          max(long long a, long long b)                   { return a > b ? a : b; }
static inline unsigned long long min(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
static inline unsigned long long max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
static inline long               min(long a, long b)                             { return a < b ? a : b; }
static inline long               max(long a, long b)                             { return a > b ? a : b; }
static inline unsigned long      min(unsigned long a, unsigned long b)           { return a < b ? a : b; }
static inline unsigned long      max(unsigned long a, unsigned long b)           { return a > b ? a : b; }
static inline float              min(float a, float b)                           { return fminf(a, b); }
static inline float              max(float a, float b)                           { return fmaxf(a, b); }

//------------------------------------------------------------------------ atomics
// One OS thread runs a grid at a time, so these are plain read-modify-writes that return the old value.
template<class T> static inline T nvdr_atomic_ptr_load(volatile T* p) { return *p; }
#define NVDR_ATOMIC_RMW(NAME, T, EXPR) \
    static inline T NAME(volatile T* p, T v) { T old = *p; *p = (T)(EXPR); return old; }
NVDR_ATOMIC_RMW(atomicAdd, int, old + v)
NVDR_ATOMIC_RMW(atomicAdd, unsigned, old + v)
NVDR_ATOMIC_RMW(atomicAdd, unsigned long long, old + v)
NVDR_ATOMIC_RMW(atomicAdd, float, old + v)
NVDR_ATOMIC_RMW(atomicAdd, double, old + v)
NVDR_ATOMIC_RMW(atomicSub, int, old - v)
NVDR_ATOMIC_RMW(atomicSub, unsigned, old - v)
NVDR_ATOMIC_RMW(atomicMin, int, old < v ? old : v)
NVDR_ATOMIC_RMW(atomicMin, unsigned, old < v ? old : v)
NVDR_ATOMIC_RMW(atomicMin, unsigned long long, old < v ? old : v)
NVDR_ATOMIC_RMW(atomicMax, int, old > v ? old : v)
NVDR_ATOMIC_RMW(atomicMax, unsigned, old > v ? old : v)
NVDR_ATOMIC_RMW(atomicMax, unsigned long long, old > v ? old : v)
NVDR_ATOMIC_RMW(atomicOr, int, old | v)
NVDR_ATOMIC_RMW(atomicOr, unsigned, old | v)
NVDR_ATOMIC_RMW(atomicAnd, int, old & v)
NVDR_ATOMIC_RMW(atomicAnd, unsigned, old & v)
NVDR_ATOMIC_RMW(atomicXor, unsigned, old ^ v)
NVDR_ATOMIC_RMW(atomicExch, int, v)
NVDR_ATOMIC_RMW(atomicExch, unsigned, v)
NVDR_ATOMIC_RMW(atomicExch, float, v)
#undef NVDR_ATOMIC_RMW
#define NVDR_ATOMIC_CAS(T) \
    static inline T atomicCAS(volatile T* p, T cmp, T v) { T old = *p; if (old == cmp) *p = v; return old; }
NVDR_ATOMIC_CAS(int)
NVDR_ATOMIC_CAS(unsigned)
NVDR_ATOMIC_CAS(unsigned long long)
#undef NVDR_ATOMIC_CAS

//------------------------------------------------------------------------ runtime API (host side)
typedef int cudaError_t;
enum { cudaSuccess = 0, cudaErrorInvalidValue = 1, cudaErrorMemoryAllocation = 2, cudaErrorLaunchFailure = 719 };
struct CUstream_st;
typedef CUstream_st* cudaStream_t;
enum cudaMemcpyKind { cudaMemcpyHostToHost = 0, cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2, cudaMemcpyDeviceToDevice = 3, cudaMemcpyDefault = 4 };
enum cudaFuncCache  { cudaFuncCachePreferNone = 0, cudaFuncCachePreferShared = 1, cudaFuncCachePreferL1 = 2, cudaFuncCachePreferEqual = 3 };
enum cudaDeviceAttr { cudaDevAttrMultiProcessorCount = 16 };
struct cudaFuncAttributes
{
    size_t sharedSizeBytes, constSizeBytes, localSizeBytes;
    int maxThreadsPerBlock, numRegs, ptxVersion, binaryVersion;
};

cudaError_t cudaGetLastError(void);
cudaError_t cudaGetDevice(int* dev);
cudaError_t cudaDeviceGetAttribute(int* value, cudaDeviceAttr attr, int dev);
cudaError_t cudaFuncGetAttributes(cudaFuncAttributes* attr, const void* func);
cudaError_t cudaFuncSetCacheConfig(const void* func, cudaFuncCache cfg);
cudaError_t cudaOccupancyMaxActiveBlocksPerMultiprocessor(int* numBlocks, const void* func, int blockSize, size_t dynSmem);
cudaError_t cudaMalloc(void** p, size_t bytes);
cudaError_t cudaFree(void* p);
cudaError_t cudaMallocHost(void** p, size_t bytes);
cudaError_t cudaFreeHost(void* p);
cudaError_t cudaMemcpyAsync(void* dst, const void* src, size_t bytes, cudaMemcpyKind kind, cudaStream_t stream = 0);
cudaError_t cudaMemcpy(void* dst, const void* src, size_t bytes, cudaMemcpyKind kind);
cudaError_t cudaMemsetAsync(void* dst, int value, size_t bytes, cudaStream_t stream = 0);
cudaError_t cudaStreamSynchronize(cudaStream_t stream);
cudaError_t cudaDeviceSynchronize(void);
// Every reference kernel is `void K(const Params p)` with a trivially copyable Params of more than 16 and
// at most NVDR_SHIM_MAX_PARAM_BYTES bytes: on the x86-64 SysV ABI such an argument is passed in memory at
// the start of the argument area, so the runtime calls K with a fixed-size blob holding the bytes at args[0].
#define NVDR_SHIM_MAX_PARAM_BYTES 1024
cudaError_t cudaLaunchKernel(const void* func, dim3 grid, dim3 block, void** args, size_t sharedMem, cudaStream_t stream);
template<class T> static inline cudaError_t cudaMalloc(T** p, size_t bytes) { return cudaMalloc((void**)p, bytes); }
template<class T> static inline cudaError_t cudaMallocHost(T** p, size_t bytes) { return cudaMallocHost((void**)p, bytes); }
