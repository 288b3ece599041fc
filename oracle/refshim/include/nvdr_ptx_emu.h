// Host emulation of the PTX instructions that the reference's cudaraster/impl/Util.inl:23-79 issues through
// inline asm.  oracle/refshim/build.py rewrites each `asm("<ptx>" : "=c"(out) : "c"(in)...)` of a temporary
// copy of Util.inl into `out = ptx_<ptx with every non-alphanumeric character replaced by '_'>(in...)`;
// the functions below are those targets, one per distinct instruction string, written from the PTX ISA's
// description of each instruction (video instructions: operands are extended to 33/34-bit integers,
// combined, optionally saturated, then merged/secondary-op'ed).
// TEST INFRASTRUCTURE ONLY.
#pragma once
#include <stdint.h>
#include <math.h>
#include <fenv.h>
#include "nvdr_cuda_shim.h"

typedef uint32_t ptx_r;

static inline int64_t ptx_sel_s(ptx_r v, int half)  { return (int64_t)(int16_t)(half ? (v >> 16) : (v & 0xffffu)); }
static inline int64_t ptx_sel_u(ptx_r v, int half)  { return (int64_t)(half ? (v >> 16) : (v & 0xffffu)); }
static inline int64_t ptx_byte(ptx_r v, int i)      { return (int64_t)((v >> (8 * i)) & 0xffu); }
static inline int64_t ptx_s(ptx_r v)                { return (int64_t)(int32_t)v; }
static inline int64_t ptx_u(ptx_r v)                { return (int64_t)v; }

//------------------------------------------------------------------------ special registers, bit search
static inline ptx_r ptx_mov_u32__0___lanemask_lt_(void) { return (1u << nvdr_shim::lane_id()) - 1u; }
static inline ptx_r ptx_mov_u32__0___lanemask_le_(void) { return (2u << nvdr_shim::lane_id()) - 1u; }
static inline ptx_r ptx_mov_u32__0___lanemask_gt_(void) { return ~((2u << nvdr_shim::lane_id()) - 1u); }
static inline ptx_r ptx_mov_u32__0___lanemask_ge_(void) { return ~((1u << nvdr_shim::lane_id()) - 1u); }
static inline ptx_r ptx_bfind_u32__0___1_(ptx_r v)      { return v ? (ptx_r)(31 - __builtin_clz(v)) : 0xffffffffu; }

//------------------------------------------------------------------------ float -> integer conversions
// .rni = round to nearest even, .rmi = round towards -inf, .sat = clamp to the destination range; NaN -> 0.
static inline ptx_r ptx_cvt_rni_sat_s32_f32__0___1_(float a)
{
    if (a != a) return 0;
    float r = nearbyintf(a);
    if (r >= 2147483648.f) return 0x7fffffffu;
    if (r <= -2147483648.f) return 0x80000000u;
    return (ptx_r)(int32_t)r;
}
static inline ptx_r ptx_cvt_rni_sat_u32_f32__0___1_(float a)
{
    if (a != a) return 0;
    float r = nearbyintf(a);
    if (r >= 4294967296.f) return 0xffffffffu;
    if (r <= 0.f) return 0;
    return (ptx_r)r;
}
static inline ptx_r ptx_cvt_rmi_sat_u32_f32__0___1_(float a)
{
    if (a != a) return 0;
    float r = floorf(a);
    if (r >= 4294967296.f) return 0xffffffffu;
    if (r <= 0.f) return 0;
    return (ptx_r)r;
}
static inline ptx_r ptx_cvt_rni_sat_u8_f32__0___1_(float a)
{
    if (a != a) return 0;
    float r = nearbyintf(a);
    if (r >= 255.f) return 255u;
    if (r <= 0.f) return 0;
    return (ptx_r)r;
}
static inline int64_t ptx_cvt_rni_s64_f32__0___1_(float a)
{
    if (a != a) return 0;
    float r = nearbyintf(a);
    if (r >= 9223372036854775808.f) return INT64_MAX;
    if (r <= -9223372036854775808.f) return INT64_MIN;
    return (int64_t)r;
}
// What nvcc makes of the C cast (unsigned)f: cvt.rzi.u32.f32 -- truncation, result clamped to [0, 2^32-1], NaN -> 0
// (PTX ISA, "cvt": integer results of float conversions are clamped to the destination range).  Used for the three
// casts of setupPleq, see build.py.
static inline uint32_t nvdr_cuda_cvt_rzi_u32_f32(float a)
{
    if (!(a > 0.f)) return 0u;
    if (a >= 4294967296.f) return 0xffffffffu;
    return (uint32_t)a;
}
// cvt.s16.u32 into a 32-bit register: low 16 bits, sign-extended.
static inline ptx_r ptx_cvt_s16_u32__0___1_(ptx_r a) { return (ptx_r)(int32_t)(int16_t)(a & 0xffffu); }

//------------------------------------------------------------------------ video add/sub with half-word selectors
#define PTX_VOP_HALF(NAME, SEL, OP, HA, HB) \
    static inline ptx_r NAME(ptx_r a, ptx_r b) { return (ptx_r)(SEL(a, HA) OP SEL(b, HB)); }
PTX_VOP_HALF(ptx_vadd_s32_s32_s32__0___1_h0___2_h0_, ptx_sel_s, +, 0, 0)
PTX_VOP_HALF(ptx_vadd_s32_s32_s32__0___1_h1___2_h0_, ptx_sel_s, +, 1, 0)
PTX_VOP_HALF(ptx_vadd_s32_s32_s32__0___1_h0___2_h1_, ptx_sel_s, +, 0, 1)
PTX_VOP_HALF(ptx_vadd_s32_s32_s32__0___1_h1___2_h1_, ptx_sel_s, +, 1, 1)
PTX_VOP_HALF(ptx_vsub_s32_s32_s32__0___1_h0___2_h0_, ptx_sel_s, -, 0, 0)
PTX_VOP_HALF(ptx_vsub_s32_s32_s32__0___1_h1___2_h0_, ptx_sel_s, -, 1, 0)
PTX_VOP_HALF(ptx_vsub_s32_s32_s32__0___1_h0___2_h1_, ptx_sel_s, -, 0, 1)
PTX_VOP_HALF(ptx_vsub_s32_s32_s32__0___1_h1___2_h1_, ptx_sel_s, -, 1, 1)
PTX_VOP_HALF(ptx_vsub_s32_u32_u32__0___1_h0___2_h0_, ptx_sel_u, -, 0, 0)
PTX_VOP_HALF(ptx_vsub_s32_u32_u32__0___1_h1___2_h0_, ptx_sel_u, -, 1, 0)
PTX_VOP_HALF(ptx_vsub_s32_u32_u32__0___1_h0___2_h1_, ptx_sel_u, -, 0, 1)
PTX_VOP_HALF(ptx_vsub_s32_u32_u32__0___1_h1___2_h1_, ptx_sel_u, -, 1, 1)
#undef PTX_VOP_HALF

//------------------------------------------------------------------------ byte-selected add / multiply-add (unsigned, wrap-around)
static inline ptx_r ptx_vadd_u32_u32_u32__0___1_b0___2_(ptx_r a, ptx_r b) { return (ptx_r)(ptx_byte(a, 0) + ptx_u(b)); }
static inline ptx_r ptx_vadd_u32_u32_u32__0___1_b1___2_(ptx_r a, ptx_r b) { return (ptx_r)(ptx_byte(a, 1) + ptx_u(b)); }
static inline ptx_r ptx_vadd_u32_u32_u32__0___1_b2___2_(ptx_r a, ptx_r b) { return (ptx_r)(ptx_byte(a, 2) + ptx_u(b)); }
static inline ptx_r ptx_vadd_u32_u32_u32__0___1_b3___2_(ptx_r a, ptx_r b) { return (ptx_r)(ptx_byte(a, 3) + ptx_u(b)); }
static inline ptx_r ptx_vmad_u32_u32_u32__0___1_b0___2___3_(ptx_r a, ptx_r b, ptx_r c) { return (ptx_r)((uint64_t)ptx_byte(a, 0) * b + c); }
static inline ptx_r ptx_vmad_u32_u32_u32__0___1_b1___2___3_(ptx_r a, ptx_r b, ptx_r c) { return (ptx_r)((uint64_t)ptx_byte(a, 1) * b + c); }
static inline ptx_r ptx_vmad_u32_u32_u32__0___1_b2___2___3_(ptx_r a, ptx_r b, ptx_r c) { return (ptx_r)((uint64_t)ptx_byte(a, 2) * b + c); }
static inline ptx_r ptx_vmad_u32_u32_u32__0___1_b3___2___3_(ptx_r a, ptx_r b, ptx_r c) { return (ptx_r)((uint64_t)ptx_byte(a, 3) * b + c); }
static inline ptx_r ptx_vmad_u32_u32_u32__0___1_b0___2_b3___3_(ptx_r a, ptx_r b, ptx_r c) { return (ptx_r)((uint64_t)ptx_byte(a, 0) * (uint64_t)ptx_byte(b, 3) + c); }
static inline ptx_r ptx_vmad_u32_u32_u32__0___1_b1___2_b3___3_(ptx_r a, ptx_r b, ptx_r c) { return (ptx_r)((uint64_t)ptx_byte(a, 1) * (uint64_t)ptx_byte(b, 3) + c); }
static inline ptx_r ptx_vmad_u32_u32_u32__0___1_b2___2_b3___3_(ptx_r a, ptx_r b, ptx_r c) { return (ptx_r)((uint64_t)ptx_byte(a, 2) * (uint64_t)ptx_byte(b, 3) + c); }
static inline ptx_r ptx_vmad_u32_u32_u32__0___1_b3___2_b3___3_(ptx_r a, ptx_r b, ptx_r c) { return (ptx_r)((uint64_t)ptx_byte(a, 3) * (uint64_t)ptx_byte(b, 3) + c); }

//------------------------------------------------------------------------ data merge into one byte of c
static inline ptx_r ptx_merge_byte(ptx_r c, int64_t tmp, int byte) { return (c & ~(0xffu << (8 * byte))) | (((ptx_r)tmp & 0xffu) << (8 * byte)); }
static inline int64_t ptx_sat_u8(int64_t v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
static inline ptx_r ptx_vadd_u32_u32_u32__0_b0___1___2___3_(ptx_r a, ptx_r b, ptx_r c) { return ptx_merge_byte(c, ptx_u(a) + ptx_u(b), 0); }
static inline ptx_r ptx_vsub_u32_u32_u32__0_b0___1___2___3_(ptx_r a, ptx_r b, ptx_r c) { return ptx_merge_byte(c, ptx_u(a) - ptx_u(b), 0); }
static inline ptx_r ptx_vadd_u32_s32_s32_sat__0_b0___1___2___3_(ptx_r a, ptx_r b, ptx_r c) { return ptx_merge_byte(c, ptx_sat_u8(ptx_s(a) + ptx_s(b)), 0); }
static inline ptx_r ptx_vadd_u32_s32_s32_sat__0_b2___1___2___3_(ptx_r a, ptx_r b, ptx_r c) { return ptx_merge_byte(c, ptx_sat_u8(ptx_s(a) + ptx_s(b)), 2); }

//------------------------------------------------------------------------ secondary arithmetic op with c
static inline ptx_r ptx_vmax_s32_s32_s32_max__0___1___2___3_(ptx_r a, ptx_r b, ptx_r c) { int64_t t = ptx_s(a) > ptx_s(b) ? ptx_s(a) : ptx_s(b); return (ptx_r)(t > ptx_s(c) ? t : ptx_s(c)); }
static inline ptx_r ptx_vmin_s32_s32_s32_min__0___1___2___3_(ptx_r a, ptx_r b, ptx_r c) { int64_t t = ptx_s(a) < ptx_s(b) ? ptx_s(a) : ptx_s(b); return (ptx_r)(t < ptx_s(c) ? t : ptx_s(c)); }
static inline ptx_r ptx_vmax_s32_s32_s32_add__0___1___2___3_(ptx_r a, ptx_r b, ptx_r c) { int64_t t = ptx_s(a) > ptx_s(b) ? ptx_s(a) : ptx_s(b); return (ptx_r)(t + ptx_s(c)); }
static inline ptx_r ptx_vmin_s32_s32_s32_add__0___1___2___3_(ptx_r a, ptx_r b, ptx_r c) { int64_t t = ptx_s(a) < ptx_s(b) ? ptx_s(a) : ptx_s(b); return (ptx_r)(t + ptx_s(c)); }
static inline ptx_r ptx_vadd_u32_u32_u32_add__0___1___2___3_(ptx_r a, ptx_r b, ptx_r c) { return (ptx_r)(ptx_u(a) + ptx_u(b) + ptx_u(c)); }
static inline ptx_r ptx_vsub_u32_u32_u32_add__0___1___2___3_(ptx_r a, ptx_r b, ptx_r c) { return (ptx_r)(ptx_u(a) - ptx_u(b) + ptx_u(c)); }
// Signed sum saturated to the unsigned 32-bit range, then unsigned min with c.
static inline ptx_r ptx_vadd_u32_s32_s32_sat_min__0___1___2___3_(ptx_r a, ptx_r b, ptx_r c)
{
    int64_t t = ptx_s(a) + ptx_s(b);
    if (t < 0) t = 0;
    if (t > 0xffffffffll) t = 0xffffffffll;
    return (ptx_r)(t < ptx_u(c) ? t : ptx_u(c));
}

//------------------------------------------------------------------------ permute / select / compare
// prmt.b32 (default mode): result byte i = byte (sel & 7) of {b:a}; selector bit 3 replicates its sign bit.
static inline ptx_r ptx_prmt_b32__0___1___2___3_(ptx_r a, ptx_r b, ptx_r c)
{
    uint64_t src = ((uint64_t)b << 32) | a;
    ptx_r d = 0;
    for (int i = 0; i < 4; i++)
    {
        ptx_r sel = (c >> (4 * i)) & 0xfu;
        ptx_r byte = (ptx_r)((src >> (8 * (sel & 7u))) & 0xffu);
        if (sel & 8u) byte = (byte & 0x80u) ? 0xffu : 0u;
        d |= byte << (8 * i);
    }
    return d;
}
static inline ptx_r ptx_slct_u32_s32__0___1___2___3_(ptx_r a, ptx_r b, ptx_r c) { return ((int32_t)c >= 0) ? a : b; }
static inline ptx_r ptx_slct_s32_s32__0___1___2___3_(ptx_r a, ptx_r b, ptx_r c) { return ((int32_t)c >= 0) ? a : b; }
static inline float ptx_slct_f32_s32__0___1___2___3_(float a, float b, ptx_r c) { return ((int32_t)c >= 0) ? a : b; }
static inline ptx_r ptx_set_ge_u32_s32__0___1___2_(ptx_r a, ptx_r b)            { return ((int32_t)a >= (int32_t)b) ? 0xffffffffu : 0u; }

//------------------------------------------------------------------------ floating point
// rcp.approx.ftz.f64: gross approximation — low 32 mantissa bits of the input ignored and of the result zero.
static inline double ptx_rcp_approx_ftz_f64__0___1_(double a)
{
    uint64_t u; memcpy(&u, &a, 8); u &= 0xffffffff00000000ull; memcpy(&a, &u, 8);
    double r = 1.0 / a;
    memcpy(&u, &r, 8); u &= 0xffffffff00000000ull; memcpy(&r, &u, 8);
    return r;
}
// fma.rm.f32: fused multiply-add rounded towards -inf.
static inline float ptx_fma_rm_f32__0___1___2___3_(float a, float b, float c)
{
    int old = fegetround();
    fesetround(FE_DOWNWARD);
    volatile float va = a, vb = b, vc = c;
    float r = fmaf(va, vb, vc);
    fesetround(old);
    return r;
}
