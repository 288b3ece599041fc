// nvdr_host.hpp -- host-side plumbing shared by the C-ABI entry points.
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/nvdr_hip.h"

namespace nvdr {

void set_error(const char* fmt, ...);

// Process-wide options (nvdr_set_option); defaults reproduce the reference.
int get_option(int option);

// Optional per-kernel hipEvent timing (bench.py only).
bool profile_on();
void profile_begin(const char* name, hipStream_t s);
void profile_end(hipStream_t s);

struct ProfileScope {
    hipStream_t s; bool on;
    ProfileScope(const char* name, hipStream_t s_) : s(s_), on(profile_on()) { if (on) profile_begin(name, s); }
    ~ProfileScope() { if (on) profile_end(s); }
};

// Development knobs (NVDR_DEBUG env, read once; never set in production):
//   1     rasterize/interpolate grad: stop after phase A (no accumulation)
//   4/8/16 k_fine (debug instantiation): skip raster / force empty ids / skip output stores
//   64    k_fine timeline records pair counts instead of phase times (tools/fine_pairs.py)
//   256   texture grad: no LDS patch table (direct global atomics)
//   1024 / 2048  texture grad: skip slot lookups / table clear+flush (cost splits)
//   16384 texture grad: ignore the caller's scratch (one-level reduction);  65536 / 131072  texture grad: no LDS adds /
//         no slot lookups and no scatter at all (timing splits, tools/exp_tex_split.py)
//   33554432 consumers ignore the tile flags;  67108864 k_fine does not produce them (nor the work order behind them);
//         134217728 consumers ignore the work order (image order, decode_block);  268435456 texture grad: no
//         k_tex_grad_light pass;  536870912 fused backward: no early exit of blocks without triangles
//   4194304 / 8388608  k_fine shared bins: arrival counter relaxed / release-only instead of acquire-release
int debug_flags();
// Development: an integer read once from the environment (tuning experiments; the product path uses the defaults).
int tune_int(const char* name, int fallback);
// Optional device buffer for in-kernel timestamps (development only; nvdr_debug_buffer()).
unsigned long long* debug_buffer();

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace nvdr

#define NVDR_REQUIRE(cond, ...)                                         \
    do { if (!(cond)) { nvdr::set_error(__VA_ARGS__); return NVDR_ERR_ARG; } } while (0)

#define NVDR_HIP_CHECK(expr)                                                            \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) {                                \
        nvdr::set_error("HIP error: %s (%s)", hipGetErrorString(e_), #expr);            \
        return NVDR_ERR_LAUNCH; } } while (0)

#define NVDR_LAUNCH_CHECK()  NVDR_HIP_CHECK(hipGetLastError())
