// nvdr_device.hpp -- device-side helpers shared by the gfx950 kernels.
//
// Everything here assumes wave64 (CDNA4).  No CUDA compatibility paths.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nvdr {

constexpr int kWave = 64;

// ---- lane / wave primitives ----------------------------------------------------

__device__ __forceinline__ int lane_id() {
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// Number of set bits of `m` strictly below the calling lane.
__device__ __forceinline__ int mask_rank(uint64_t m) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__device__ __forceinline__ int first_lane_value(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_move0(float v) {
    // Lanes without a valid source (or masked out) receive 0.
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false));
}

// Sum over all 64 lanes; the total is valid in lane 63 (only).
__device__ __forceinline__ float wave_sum_to_last(float v) {
    v += dpp_move0<0xb1>(v);            // quad_perm [1,0,3,2]
    v += dpp_move0<0x4e>(v);            // quad_perm [2,3,0,1]
    v += dpp_move0<0x114>(v);           // row_shr:4
    v += dpp_move0<0x118>(v);           // row_shr:8
    v += dpp_move0<0x142, 0xa>(v);      // row_bcast:15 -> rows 1,3
    v += dpp_move0<0x143, 0xc>(v);      // row_bcast:31 -> rows 2,3
    return v;
}

// Bitwise OR over all 64 lanes; the result is valid in lane 63 (only).
__device__ __forceinline__ uint32_t wave_or_to_last(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false);           // quad_perm [1,0,3,2]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false);           // quad_perm [2,3,0,1]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);          // row_shr:4
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);          // row_shr:8
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);          // row_bcast:15 -> rows 1, 3
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);          // row_bcast:31 -> rows 2, 3
    return v;
}

// Inclusive prefix sum over the 64 lanes, by data-parallel-primitive moves only (no LDS permutes, and no per-lane source
// addresses for the compiler to hoist out of a loop and park somewhere): Hillis-Steele inside each row of 16 lanes (row_shr
// fills with the "old" operand, 0, at the row's start), then the rows' totals across (row_bcast 15 / 31).
__device__ __forceinline__ int wave_scan_incl(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);          // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);          // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);          // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);          // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);          // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);          // row_bcast:31 -> rows 2, 3
    return v;
}

// sum_i y_i (a_i - b_i), the gradient of a barycentric (interpolate.cu:172-182: `gb0 += y * (a0 - a2)` attribute by attribute,
// which nvcc contracts into one fma per attribute).  Spelled out ONCE, with every operation explicit, for the kernels that
// compute it (k_interp_grad and the fused k_interp_raster_grad): their g_rast must agree bit for bit -- the stand-in for rast's
// gradient materialises through the former what the latter used (ops.py _LazyGrad) -- whatever the compiler would otherwise
// contract, reassociate or pack differently in two translation units built with different flags.
__device__ __forceinline__ float dot_diff(float y, float a, float b, float acc) {
#pragma clang fp contract(off)
    return __fmaf_rn(y, a - b, acc);
}
__device__ __forceinline__ float dot_diff(float4 y, float4 a, float4 b) {
    return dot_diff(y.w, a.w, b.w, dot_diff(y.z, a.z, b.z, dot_diff(y.y, a.y, b.y, dot_diff(y.x, a.x, b.x, 0.f))));
}
__device__ __forceinline__ float dot_diff(float2 y, float2 a, float2 b) {
    return dot_diff(y.y, a.y, b.y, dot_diff(y.x, a.x, b.x, 0.f));
}

// Hardware f32 atomic add, no return value (global_atomic_add_f32).
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

// All three vertex indices inside [0, V): one unsigned compare each, combined without short-circuit
// evaluation (a chain of `<0 || >=V` tests compiles to nested divergent branches, each re-initialising
// the values that are live across it).
__device__ __forceinline__ bool indices_ok(int a, int b, int c, int V)
{
    return ((uint32_t)a < (uint32_t)V) & ((uint32_t)b < (uint32_t)V) & ((uint32_t)c < (uint32_t)V);
}

// XCD-aware decode of a 1-D grid into (block x, block y, image): blocks b, b+8, ... share an XCD
// and walk a contiguous chunk of the (image-major) work list.
__device__ __forceinline__ bool decode_block(int gx, int gy, int N, int& bx, int& by, int& pz)
{
    const int total = gx * gy * N;
    const int perXcd = (total + 7) >> 3;
    const int j = (int)(blockIdx.x >> 3);
    const int item = (int)(blockIdx.x & 7) * perXcd + j;
    if (j >= perXcd || item >= total) return false;
    pz = item / (gx * gy);
    const int rem = item - pz * (gx * gy);
    by = rem / gx;
    bx = rem - by * gx;
    return true;
}

// ---- wave-uniform loads through the scalar cache ----------------------------------------
// A load whose address is the same in every lane (built from kernel arguments and readfirstlane'd values) of memory that
// nothing writes during the launch: the pointer is cast into the constant address space, for which the compiler itself
// selects s_load_dword(xN) AND tracks the outstanding load (its own s_waitcnt lgkmcnt placement).  This replaces hand-written
// `s_load_dword` asm statements with a separate `s_waitcnt` statement, between which the register allocator was free to
// touch the destination SGPRs (SMEM returns are not interlocked: ADVICE r4).
template <typename T>
__device__ __forceinline__ T scalar_load(const T* p) {
    typedef const T __attribute__((address_space(4))) * ConstPtr;
    return *(ConstPtr)p;
}

// ---- streaming accesses ---------------------------------------------------------------
// Non-temporal 16-byte accesses for tensors that are touched once per kernel and are larger than
// what can stay cached until their next use: they do not displace the tensors the NEXT kernels
// re-read (rast, g_rast) from the 256 MB Infinity Cache.
typedef float nvdr_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_streaming(float4* p, float4 v) {
    __builtin_nontemporal_store(nvdr_v4f{v.x, v.y, v.z, v.w}, (nvdr_v4f*)p);
}
__device__ __forceinline__ float4 load_streaming(const float4* p) {
    const nvdr_v4f v = __builtin_nontemporal_load((const nvdr_v4f*)p);
    return make_float4(v.x, v.y, v.z, v.w);
}

// ---- tile occupancy of a rasterized image ---------------------------------------------
// k_fine notes for every 8x8-pixel tile whether ANY of its pixels shows a triangle (one byte per tile, [N][ceil(H/8)]
// [ceil(W/8)]).  The kernels that consume `rast` -- interpolate forward / backward, rasterize backward, the antialias
// discontinuity pass -- get the flags from the operator layer when the rast they are given is, untouched, the tensor that
// rasterize() returned (ops.py `_RasterOrigin`), and then do not read the 16 (32) bytes per pixel of rast (rast_db) of an
// empty tile at all: two thirds of the tiles of the benchmark scene.  f == nullptr: no information, read everything.
//
// Behind the flags the same buffer holds a WORK ORDER for those kernels (written by k_flag_order at the end of the rasterizer's
// forward pass): the image's 64x64-pixel bins, those with a covered tile first (image-major), then the others, and the number
// of the former.  A consumer's launch walks that list instead of the image (decode_block_ordered): every XCD takes an eighth
// of the bins with triangles, first, then an eighth of the empty ones.  Why: workgroups are dealt to the XCDs round robin
// and in order, so with each XCD walking through its own images (decode_block) the eight are busy only while ALL of them are
// in a covered region, and the XCD whose images show the most pixels finishes last (the benchmark scene: 929 covered blocks
// against a mean of 736).  Measured on the fused backward pass: 193 -> 133 us.
constexpr int kOrderBinShift = 6;                  // bins of 64 x 64 pixels (= the rasterizer's bins)
constexpr int kOrderMaxBins  = 1 << 16;            // beyond that (e.g. 64 images of 2048^2) no order is kept,
constexpr int kOrderMinBins  = 2048;               // nor below this: launches of a few hundred workgroups are over before it matters,
                                                   // and small workloads are bound by the host, where k_flag_order is one more launch
struct TileFlags {
    const uint8_t* f; int w, h;
    const int* order;                              // nullptr or [nBins + 1]: the bins as described above, [nBins] = bins with a covered tile
    int binsX, binsY, nBins;
    __device__ __forceinline__ bool empty(int pz, int py, int px) const {
        return f != nullptr && f[((size_t)pz * h + (py >> 3)) * w + (px >> 3)] == 0;
    }
};

// The buffer's layout (include/nvdr_hip.h `tile_flags`): N*h*w flag bytes, padding to 16, then (nBins + 1) ints.
__host__ __device__ inline size_t tile_flags_order_offset(int N, int H, int W)
{
    return ((size_t)N * (size_t)((H + 7) >> 3) * (size_t)((W + 7) >> 3) + 15) / 16 * 16;
}
// ... and behind the order, 8-byte aligned, one byte per bin and tile row from k_fine (what k_flag_order reads)
__host__ __device__ inline size_t tile_flags_rowcov_offset(size_t order_offset, long long nBins)
{
    return (order_offset + (size_t)(nBins + 1) * 4 + 7) / 8 * 8;
}
__host__ inline long long tile_flags_bins(int N, int H, int W)       // 0: no order for this size
{
    if (H > 2048 || W > 2048) return 0;                 // several viewports (raster.hip kMaxViewport): k_fine's bins are not the image's
    const long long nb = (long long)N * ((H + 63) >> kOrderBinShift) * ((W + 63) >> kOrderBinShift);
    return (nb >= kOrderMinBins && nb <= kOrderMaxBins) ? nb : 0;
}
__host__ inline TileFlags tile_flags_view(const uint8_t* p, int N, int H, int W, bool with_order = true)
{
    TileFlags t{};
    if (!p) return t;
    t.f = p; t.w = (W + 7) >> 3; t.h = (H + 7) >> 3;
    const long long nb = tile_flags_bins(N, H, W);
    if (with_order && nb > 0) {
        t.binsX = (W + 63) >> kOrderBinShift; t.binsY = (H + 63) >> kOrderBinShift; t.nBins = (int)nb;
        t.order = (const int*)(p + tile_flags_order_offset(N, H, W));
    }
    return t;
}
// Workgroups an ordered launch needs when each covers 1/per of a bin: every XCD may get its share rounded up, twice.
__host__ inline long long tile_flags_ordered_grid(const TileFlags& t, int per) { return 8ll * ((t.nBins + 7) / 8 + 2) * per; }

// Which entry of the order the `slot`-th bin of XCD `xcd` is: an eighth of the covered bins (the first nCov entries) first,
// then an eighth of the others; -1 beyond the XCD's share.  (Host-callable: tests/test_work_order_index.py walks it over
// every (nBins, nCov) and checks that tile_flags_ordered_grid's launch visits every entry exactly once.)
__host__ __device__ inline int ordered_list_index(int nBins, int nCov, int xcd, int slot)
{
    const int nEmp = nBins - nCov;
    const int cc = (nCov + 7) >> 3, ec = (nEmp + 7) >> 3;
    const int c0 = xcd * cc < nCov ? xcd * cc : nCov, cn = (c0 + cc < nCov ? c0 + cc : nCov) - c0;
    const int e0 = xcd * ec < nEmp ? xcd * ec : nEmp, en = (e0 + ec < nEmp ? e0 + ec : nEmp) - e0;
    if (slot < cn) return c0 + slot;
    if (slot - cn < en) return nCov + e0 + (slot - cn);
    return -1;
}

// A covered bin's workgroups may take on the zeros of an empty bin of the same XCD share (k_interp_fwd_cols: the stores of a
// bin that only needs zeros go out while other waves of the CU wait for their loads, instead of all together at the end of the
// launch).  For slot `slot` of XCD `xcd`: own = the list index of its own bin (-1: none); partner = the list index of the EMPTY
// bin this workgroup also clears (-1: none); skip = this is an empty bin that a covered bin's workgroup clears (leave at once).
__host__ __device__ inline void ordered_list_pair(int nBins, int nCov, int xcd, int slot, int& own, int& partner, bool& skip)
{
    const int nEmp = nBins - nCov;
    const int cc = (nCov + 7) >> 3, ec = (nEmp + 7) >> 3;
    const int c0 = xcd * cc < nCov ? xcd * cc : nCov, cn = (c0 + cc < nCov ? c0 + cc : nCov) - c0;
    const int e0 = xcd * ec < nEmp ? xcd * ec : nEmp, en = (e0 + ec < nEmp ? e0 + ec : nEmp) - e0;
    own = -1; partner = -1; skip = false;
    if (slot < cn) { own = c0 + slot; if (slot < en) partner = nCov + e0 + slot; }
    else if (slot - cn < en) { own = nCov + e0 + (slot - cn); skip = (slot - cn) < cn; }
}

// Ordered counterpart of decode_block() for workgroups of bw x bh pixels (both dividing 64): false = nothing to do.
__device__ __forceinline__ bool decode_block_ordered(const TileFlags& t, int gx, int gy, int bw, int bh, int& bx, int& by, int& pz)
{
    const int sx = 64 / bw, per = sx * (64 / bh);                  // workgroups per bin
    const int xcd = (int)(blockIdx.x & 7), j = (int)(blockIdx.x >> 3);
    const int slot = j / per, sub = j - slot * per;
    const int idx = ordered_list_index(t.nBins, t.order[t.nBins], xcd, slot);
    if (idx < 0) return false;
    const int bin = __builtin_amdgcn_readfirstlane(t.order[idx]);
    pz = bin / (t.binsX * t.binsY);
    const int rem = bin - pz * (t.binsX * t.binsY);
    const int binY = rem / t.binsX, binX = rem - binY * t.binsX;
    const int sy = sub / sx;
    bx = binX * sx + (sub - sy * sx);
    by = binY * (64 / bh) + sy;
    return bx < gx && by < gy;                                     // (bins at the right / bottom border reach beyond the image)
}

// ---- triangle-id <-> f32 codec (reference csrc/common/common.h:186-193) ----------
// Identity up to 2^24; above that the id is stored as a bit-offset float so that it
// survives the f32 channel.

__device__ __forceinline__ int float_to_triidx(float x) {
    if (x <= 16777216.f) return (int)x;
    return __float_as_int(x) - 0x4a800000;
}
__device__ __forceinline__ float triidx_to_float(int x) {
    if (x <= 0x01000000) return (float)x;
    return __int_as_float(0x4a800000 + x);
}

// ---- grouped atomic accumulation -------------------------------------------------
// The reference coalesces gradient atomics with __match_any_sync (common.h:198-241).
// gfx950 has no match instruction; groups are peeled off with a readfirstlane
// "waterfall" and each value is reduced over the group's lanes with DPP, so that one
// atomic is issued per (group, value) per wave.
//
// Usage:
//   GroupIter it(active, key);
//   while (it.next()) { float s = it.sum(v); if (it.writer()) atomic_add_f32(ptr, s); }
// `ptr` must be computed from the group's key by every lane of the group (lane 63
// performs the atomic and may not be a member, so callers pass data via it.bcast()).

struct GroupIter {
    uint64_t remaining;
    int      key;
    bool     active;
    bool     member;
    int      leader;

    __device__ __forceinline__ GroupIter(bool active_, int key_) : key(key_), active(active_), member(false), leader(0) {
        remaining = __ballot(active_);
    }
    __device__ __forceinline__ bool next() {
        if (remaining == 0) return false;
        leader = __builtin_ctzll(remaining);
        int cur = __builtin_amdgcn_readlane(key, leader);
        member = active && (key == cur);
        remaining &= ~__ballot(member);
        return true;
    }
    // Sum of v over the current group; valid in lane 63.
    __device__ __forceinline__ float sum(float v) const { return wave_sum_to_last(member ? v : 0.f); }
    __device__ __forceinline__ bool writer() const { return lane_id() == 63; }
    // Broadcast a per-lane int from the group's leader to all lanes.
    __device__ __forceinline__ int bcast(int v) const { return __builtin_amdgcn_readlane(v, leader); }
};

// ---- per-workgroup vertex accumulator ------------------------------------------------
// Gradient scatter for the backward kernels.  Measured on MI355X (tools/lds_atomic_bench.hip):
// global f32 atomics sustain ~21 G lane-ops/s however they are packed (they execute at the
// memory side: the per-XCD L2s are not coherent), LDS ds_add_f32 costs ~200 cycles per wave
// instruction, while LDS 64-bit INTEGER atomics cost ~7 cycles (+7 per same-address lane).
// So a workgroup owns a block of pixels and an LDS open-addressing table keyed by vertex
// index.  Contributions are first summed over runs of equal triangle id inside the wave
// (RunScan: pixels of one triangle are contiguous in a scan line), the run totals are added
// to the table in 64-bit fixed point with ds_add_u64, and each touched (vertex, component)
// is flushed once with a hardware global f32 atomic.
//
// Fixed point: x -> round(x * 2^s) with the power of two s chosen per block from the largest
// contribution magnitude M (M < 2^(e+1)  =>  s = 45 - e), so that a run total (<= 16 terms)
// stays below 2^51 and the table sums (<= 2^13 terms) far below 2^63.  Every contribution is
// represented to 2^-46 * M -- finer than an f32 sum of the same terms could resolve -- and
// the integer sum is order independent.  Vertices that cannot be placed (table full) and
// blocks holding inf/NaN fall back to direct global atomics, so any input is handled.

__device__ __forceinline__ float wave_max_to_last(float v) {        // v >= 0; result valid in lane 63
    v = fmaxf(v, dpp_move0<0xb1>(v));
    v = fmaxf(v, dpp_move0<0x4e>(v));
    v = fmaxf(v, dpp_move0<0x114>(v));
    v = fmaxf(v, dpp_move0<0x118>(v));
    v = fmaxf(v, dpp_move0<0x142, 0xa>(v));
    v = fmaxf(v, dpp_move0<0x143, 0xc>(v));
    return v;
}

// Publish a wave's largest magnitude to the workgroup.  `m` >= 0 or NaN per lane; NaN/inf end
// up as a bit pattern >= 0x7F800000 (uint compare == float compare for non-negative floats).
__device__ __forceinline__ void block_max_update(uint32_t* s_max, float m) {
    if (__ballot(m != 0.f) == 0) return;                            // NaN != 0 is true
    uint32_t bits = (uint32_t)__float_as_int(m) & 0x7FFFFFFFu;
    float mm = wave_max_to_last(__int_as_float((int)min(bits, 0x7F800000u)));
    if (lane_id() == 63) atomicMax(s_max, (uint32_t)__float_as_int(mm));
}

// A magnitude as BITS: for non-negative floats unsigned order is float order, and inf / NaN patterns sort above every finite one, so
// max(m, mag_bits(a)) keeps a NaN visible like max_abs_keep_nan below -- with one v_and and one v_max_u32 instead of a compare, a
// select, an fmax and an fabs.  The backward kernels run at 95 % of their vector issue rate: instructions are their time.
__device__ __forceinline__ uint32_t mag_bits(float v) { return (uint32_t)__float_as_int(v) & 0x7FFFFFFFu; }

// max(|a|, m) that keeps NaN visible (fmaxf would drop it).
__device__ __forceinline__ float max_abs_keep_nan(float m, float a) {
    float r = fmaxf(m, fabsf(a));
    return (a != a) ? a : r;
}

struct FixedScale {
    double scale, inv;
    __device__ __forceinline__ explicit FixedScale(uint32_t max_bits) {
        const int e = (int)(max_bits >> 23) - 127;                   // M < 2^(e+1)
        scale = __longlong_as_double((long long)(1023 + 45 - e) << 52);
        inv   = __longlong_as_double((long long)(1023 - 45 + e) << 52);
    }
    // round-to-nearest-even of x * 2^s as a two's complement 64-bit integer: adding 1.5 * 2^52
    // leaves the integer in the low mantissa bits of the double (|x * 2^s| < 2^51).
    __device__ __forceinline__ unsigned long long to_fixed(float x) const {
        const double d = __fma_rn((double)x, scale, 6755399441055744.0);
        return (unsigned long long)__double_as_longlong(d) - 0x4338000000000000ull;
    }
    __device__ __forceinline__ float to_float(unsigned long long t) const {
        return (float)((double)(long long)t * inv);
    }
};

// 32-bit variant for accumulators with few terms per cell (texel patches): x -> round(x * 2^s) with
// s = 20 - e (M < 2^(e+1) the largest summand), so one summand stays below 2^21, a pre-summed run of 16
// below 2^25 and a cell holding <= 512 summands below 2^30.  Resolution 2^-21 of M (rounding error
// <= 2^-22 M per summand: the ulp scale of an f32 sum of that size).
struct FixedScale32 {
    float scale, inv;
    __device__ __forceinline__ explicit FixedScale32(uint32_t max_bits) {
        int e = (int)(max_bits >> 23) - 127;                       // M < 2^(e+1)
        e = max(e, -100);                                            // keep 2^(20-e) finite for denormal maxima
        scale = __int_as_float((127 + 20 - e) << 23);
        inv   = __int_as_float((127 - 20 + e) << 23);
    }
    __device__ __forceinline__ int to_fixed(float x) const { return __float2int_rn(x * scale); }
    __device__ __forceinline__ float to_float(int t) const { return (float)t * inv; }
};

// Segmented inclusive scan over runs of equal `key` among consecutive active lanes, limited to
// the 16-lane DPP rows (a run that crosses a row boundary simply yields two totals).  After
// scan(v) the last lane of every run (`tail`) holds the run's sum.
struct RunScan {
    float c1, c2, c4, c8;
    bool  tail;
    // Value of the previous lane of the 16-lane row (`fallback` in the row's first lane).
    static __device__ __forceinline__ int prev_lane(int v, int fallback) {
        return __builtin_amdgcn_update_dpp(fallback, v, 0x111, 0xf, 0xf, false);         // row_shr:1
    }
    __device__ __forceinline__ RunScan(int key, bool active) {
        if (!active) key = -1;
        init((prev_lane(key, -2) != key) | !active, active);
    }
    // From a caller-made "starts a new run" predicate (callers compare several fields with prev_lane()).
    struct FromHead {};
    __device__ __forceinline__ RunScan(FromHead, bool head, bool active) { init(head | !active, active); }
    __device__ __forceinline__ bool any_merge() const { return merges; }
    bool merges;
    __device__ __forceinline__ void init(bool head, bool active) {
        const int lane = lane_id();
        // Lanes that are not executing (divergent callers) count as run boundaries.
        const uint64_t H = __ballot(head) | ~__ballot(true) | 0x0001000100010001ull;
        merges = (~H) != 0ull;                                                            // some lane continues a run
        const uint64_t below = H & ((2ull << lane) - 1ull);                                // heads at or below me
        const int dist = lane - (63 - __builtin_clzll(below));
        c1 = dist >= 1 ? 1.f : 0.f; c2 = dist >= 2 ? 1.f : 0.f;
        c4 = dist >= 4 ? 1.f : 0.f; c8 = dist >= 8 ? 1.f : 0.f;
        tail = active && (lane == 63 || ((H >> (lane + 1)) & 1ull));
    }
    template <int CTRL> static __device__ __forceinline__ float shr(float v) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
    }
    // Three scans at once with the DPP source folded into the multiply-add (the compiler keeps
    // v_mov_dpp + v_fma apart); interleaving the three chains covers the 2 wait states a DPP read
    // needs after a VALU write of its source, the leading s_nop covers a preceding EXEC/VGPR write.
    __device__ __forceinline__ void scan3(float& a, float& b, float& c) const {
        asm volatile(
            "s_nop 4\n"
            "v_fmac_f32_dpp %0, %0, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %1, %1, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %2, %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %0, %0, %4 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %1, %1, %4 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %2, %2, %4 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %0, %0, %5 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %1, %1, %5 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %2, %2, %5 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %0, %0, %6 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %1, %1, %6 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %2, %2, %6 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            : "+v"(a), "+v"(b), "+v"(c) : "v"(c1), "v"(c2), "v"(c4), "v"(c8));
    }
    __device__ __forceinline__ float scan(float v) const {
        v = __fmaf_rn(shr<0x111>(v), c1, v);
        v = __fmaf_rn(shr<0x112>(v), c2, v);
        v = __fmaf_rn(shr<0x114>(v), c4, v);
        v = __fmaf_rn(shr<0x118>(v), c8, v);
        return v;
    }
};

struct VertexTable {
    uint32_t*           keys;    // [slots]   0 = empty, else vertex + 1
    unsigned long long* vals;    // [slots * stride] fixed-point sums
    int                 slots;   // power of two
    int                 stride;  // components per vertex

    __device__ __forceinline__ void clear(int tid, int nthreads) {
        for (int i = tid; i < slots; i += nthreads) keys[i] = 0u;
        for (int i = tid; i < slots * stride; i += nthreads) vals[i] = 0ull;
    }
    // Slot of `vertex` (inserting it if needed) or -1 when the probe budget is exhausted.  Six probes: a block of an ordinary mesh
    // claims a tenth of the table and finds its slot with the first or second; a block of sub-pixel triangles (a million-triangle
    // mesh: up to 3072 vertices for 512 slots) cannot be held anyway, and with sixteen probes every lane walked sixteen dependent
    // compare-and-swaps to learn that before taking the direct path.
    static constexpr int kProbes = 6;
    __device__ __forceinline__ int find(int vertex) const {
        const uint32_t key = (uint32_t)vertex + 1u;
        uint32_t h = (key * 0x9E3779B1u) >> 8;
#pragma unroll 1
        for (int probe = 0; probe < kProbes; probe++) {
            h &= (uint32_t)(slots - 1);
            uint32_t old = atomicCAS(&keys[h], 0u, key);
            if (old == 0u || old == key) return (int)h;
            h++;
        }
        return -1;
    }
    // The same for three vertices at once: the three first probes are in flight together (one LDS round trip instead of
    // three); only a probe that lands on another vertex's slot continues on its own.
    __device__ __forceinline__ void find3(int va, int vb, int vc, int& sa, int& sb, int& sc) const {
        const uint32_t ka = (uint32_t)va + 1u, kb = (uint32_t)vb + 1u, kc = (uint32_t)vc + 1u;
        const uint32_t ha = ((ka * 0x9E3779B1u) >> 8) & (uint32_t)(slots - 1);
        const uint32_t hb = ((kb * 0x9E3779B1u) >> 8) & (uint32_t)(slots - 1);
        const uint32_t hc = ((kc * 0x9E3779B1u) >> 8) & (uint32_t)(slots - 1);
        const uint32_t oa = atomicCAS(&keys[ha], 0u, ka);
        const uint32_t ob = atomicCAS(&keys[hb], 0u, kb);
        const uint32_t oc = atomicCAS(&keys[hc], 0u, kc);
        sa = (oa == 0u || oa == ka) ? (int)ha : probe_on(ka, ha + 1u);
        sb = (ob == 0u || ob == kb) ? (int)hb : probe_on(kb, hb + 1u);
        sc = (oc == 0u || oc == kc) ? (int)hc : probe_on(kc, hc + 1u);
    }
    __device__ __forceinline__ int probe_on(uint32_t key, uint32_t h) const {
#pragma unroll 1
        for (int probe = 1; probe < kProbes; probe++) {
            h &= (uint32_t)(slots - 1);
            const uint32_t old = atomicCAS(&keys[h], 0u, key);
            if (old == 0u || old == key) return (int)h;
            h++;
        }
        return -1;
    }
    __device__ __forceinline__ void add(int slot, int comp, unsigned long long v) const {
        atomicAdd(&vals[slot * stride + comp], v);
    }
    // Lists the claimed slots in `list` and returns their number (`counter` must be zero on entry; every
    // thread of the workgroup calls this, it contains a barrier).  The flush then walks only what was used:
    // a 64x16-pixel block touches a quarter of the table, a scan of all of it was a tenth of the
    // backward kernels' instructions.
    __device__ __forceinline__ int compact(uint16_t* list, uint32_t* counter, int tid, int nthreads) const {
        for (int i = tid; i < slots; i += nthreads)
            if (keys[i] != 0u) list[atomicAdd(counter, 1u)] = (uint16_t)i;
        __syncthreads();
        return (int)*counter;
    }
};

}  // namespace nvdr
