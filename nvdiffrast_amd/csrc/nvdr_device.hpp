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

// Hardware f32 atomic add, no return value (global_atomic_add_f32).
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

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

}  // namespace nvdr
