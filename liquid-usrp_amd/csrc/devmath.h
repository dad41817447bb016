// devmath.h -- device-side helpers shared by the gfx950 kernels (complex arithmetic,
// 32-bit-phase oscillator, wave64 reductions).  CDNA4 only: wave size is 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcrx {

typedef float2 cfd;     // interleaved complex float, same layout as std::complex<float>

__device__ __forceinline__ cfd cmake(float re, float im) { return make_float2(re, im); }
__device__ __forceinline__ cfd cadd(cfd a, cfd b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cfd csub(cfd a, cfd b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cfd cmul(cfd a, cfd b)
{ return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// the same with the roundings pinned (explicit fma shape): the compiler contracts cmul's a.x b.x - a.y b.y either way round, per inlined copy --
// two unrolled copies of one butterfly can differ in the last bit.  Where a value must not depend on WHICH copy computed it (the
// channelizer's transform: a block lands in different tile rows when a stream is cut differently) use this one.
__device__ __forceinline__ cfd cmul_fx(cfd a, cfd b)
{ return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x)); }
__device__ __forceinline__ cfd cmulc(cfd a, cfd b)      // a * conj(b)
{ return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
__device__ __forceinline__ cfd cscale(cfd a, float g) { return make_float2(a.x * g, a.y * g); }
__device__ __forceinline__ cfd cmulnj(cfd a) { return make_float2(a.y, -a.x); }      // a * (-j)
// a + (-j) b and a - (-j) b in ONE packed add each: the operand swizzle (op_sel) and a negation of one half only (neg_lo / neg_hi) do
// the rotation.  The compiler's form of cadd(a, cmulnj(b)) / csub(a, cmulnj(b)) is two packed adds (sum and difference of the
// swizzled pair) and three moves that recombine their halves.  Same values: (a.x + b.y, a.y - b.x) and (a.x - b.y, a.y + b.x).
typedef float devmath_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cfd cadd_nj(cfd a, cfd b)
{
    devmath_v2f r; const devmath_v2f av = { a.x, a.y }, bv = { b.x, b.y };
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(av), "v"(bv));
    return make_float2(r.x, r.y);
}
__device__ __forceinline__ cfd csub_nj(cfd a, cfd b)
{
    devmath_v2f r; const devmath_v2f av = { a.x, a.y }, bv = { b.x, b.y };
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(av), "v"(bv));
    return make_float2(r.x, r.y);
}

// sin/cos of a 32-bit phase (theta * 2 pi / 2^32): octant-centred reduction is exact in
// integers, then degree-7/8 minimax polynomials on [-pi/4, pi/4] (abs error ~1e-7).
__device__ __forceinline__ void sincos_u32(uint32_t th, float &s, float &c)
{
    uint32_t t = th + 0x20000000u;
    uint32_t q = t >> 30;
    int32_t r = (int32_t)(t & 0x3FFFFFFFu) - 0x20000000;
    float a = (float)r * 1.4629180792671596e-09f;          // (pi/2) / 2^30
    float a2 = a * a;
    float sp = a + a * a2 * (-1.6666654611e-1f + a2 * (8.3321608736e-3f + a2 * (-1.9515295891e-4f)));
    float cp = 1.0f - 0.5f * a2 + a2 * a2 * (4.166664568298827e-2f + a2 * (-1.388731625493765e-3f + a2 * 2.443315711809948e-5f));
    float ss = (q & 1) ? cp : sp;
    float cc = (q & 1) ? sp : cp;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}
// The same on the transcendental unit: v_sin_f32 / v_cos_f32 take revolutions; measured max abs
// error on gfx950 is 1.24e-7 over [-2, 2) (scratch/sintest.hip), the phase conversion adds < 2^-25 rev.
__device__ __forceinline__ void sincos_u32_hw(uint32_t th, float &s, float &c)
{
    const float rev = (float)(int32_t)th * 2.3283064365386963e-10f;    // 2^-32
    s = __builtin_amdgcn_sinf(rev);
    c = __builtin_amdgcn_cosf(rev);
}
__device__ __forceinline__ cfd mix_down_hw(cfd x, uint32_t th)
{
    float s, c; sincos_u32_hw(th, s, c);
    // explicit fma shape: every inlined copy rounds identically, so a stream split over several
    // calls reproduces the single-call result bit for bit
    return make_float2(fmaf(x.x, c, x.y * s), fmaf(x.y, c, -(x.x * s)));
}
// x * conj(e^{j theta})
__device__ __forceinline__ cfd mix_down(cfd x, uint32_t th)
{
    float s, c; sincos_u32(th, s, c);
    return make_float2(x.x * c + x.y * s, x.y * c - x.x * s);
}
__device__ __forceinline__ cfd mix_up(cfd x, uint32_t th)
{
    float s, c; sincos_u32(th, s, c);
    return make_float2(x.x * c - x.y * s, x.y * c + x.x * s);
}
// radians -> 32-bit phase; must match the host / oracle conversion bit for bit
__device__ __forceinline__ uint32_t rad2u32(float rad)
{
    double p = (double)rad * 0.15915494309189535;       // 1 / (2 pi)
    p -= floor(p);
    return (uint32_t)(unsigned long long)__double2ll_rn(p * 4294967296.0);
}
__device__ __forceinline__ float u32rad(uint32_t u)
{ return (float)((double)(int32_t)u * 1.4629180792671596e-09); }   // 2 pi / 2^32

// ---- cross-lane primitives.  Lane exchanges use DPP modifiers where the pattern exists
// (xor 1, 2, 8: folded into the consuming VALU instruction) and the LDS crossbar
// (ds_swizzle / ds_bpermute: no LDS memory is touched) for xor 4, 16, 32.
template <int CTRL, bool ZERO_OOB = true>
__device__ __forceinline__ float dpp_mov(float v, float old = 0.f)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                 CTRL, 0xf, 0xf, ZERO_OOB));
}
__device__ __forceinline__ int lane_bperm32() { return (int)((__lane_id() ^ 32u) << 2); }   // ds_bpermute address of lane ^ 32
template <int H>
__device__ __forceinline__ float xor_lane(float v, int bperm32)
{
    if constexpr (H == 1)       return dpp_mov<0xB1>(v);           // quad_perm [1,0,3,2]
    else if constexpr (H == 2)  return dpp_mov<0x4E>(v);           // quad_perm [2,3,0,1]
    else if constexpr (H == 8)  return dpp_mov<0x128>(v);          // row_ror:8
    else if constexpr (H == 4)  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x101F));
    else if constexpr (H == 16) return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
    else                        return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bperm32, __builtin_bit_cast(int, v)));
}
// x[l ^ 4] without the LDS crossbar: two bank-masked row shifts
__device__ __forceinline__ float xor4_dpp(float v)
{
    const int b = __builtin_bit_cast(int, v);
    const int t = __builtin_amdgcn_update_dpp(b, b, 0x104, 0xf, 0x5, false);      // row_shl:4 into banks 0, 2
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(t, b, 0x114, 0xf, 0xa, false));   // row_shr:4 into banks 1, 3
}
// One radix-2 DIF butterfly leg across lanes H apart, all on the VALU (the symbol loops are one
// long dependent chain, so an LDS-crossbar round trip per stage is what they wait on):
//   lanes with (l & H) == 0 (sg = +1): x + x[l ^ H];  lanes with (l & H) != 0 (sg = -1): x[l ^ H] - x.
// H = 32 / 16 use gfx950's v_permlane32_swap / v_permlane16_swap on two copies of x: afterwards
// the first copy holds the lower partner and the second the upper one in every lane.
template <int H>
__device__ __forceinline__ float bfly_leg(float x, float sg)
{
    if constexpr (H == 32) {
        // (inline asm: with the builtin, hipcc 7.2 forwards the pre-swap copy into the use of the
        //  second result.  s_nop covers the VALU-write -> permlane-read wait states.)
        float lo = x, hi = x;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
        return fmaf(sg, hi, lo);
    } else if constexpr (H == 16) {
        float lo = x, hi = x;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
        return fmaf(sg, hi, lo);
    } else if constexpr (H == 8) return fmaf(sg, x, dpp_mov<0x128>(x));
    else if constexpr (H == 4)   return fmaf(sg, x, xor4_dpp(x));
    else if constexpr (H == 2)   return fmaf(sg, x, dpp_mov<0x4E>(x));
    else                         return fmaf(sg, x, dpp_mov<0xB1>(x));
}
// wave64 all-reduce sums (butterfly: every lane ends with the total)
__device__ __forceinline__ float wave_sum_fast(float v, int bperm32)
{
    v += xor_lane<1>(v, bperm32);  v += xor_lane<2>(v, bperm32);  v += xor_lane<4>(v, bperm32);
    v += xor_lane<8>(v, bperm32);  v += xor_lane<16>(v, bperm32); v += xor_lane<32>(v, bperm32);
    return v;
}
// (wave_sum / wave_csum: all-VALU, see wave_total_dpp below)
__device__ __forceinline__ uint32_t wave_xor_u32(uint32_t v)
{
    const int bp = lane_bperm32();
#define MCRX_XSTEP(H) v ^= __builtin_bit_cast(uint32_t, xor_lane<H>(__builtin_bit_cast(float, v), bp));
    MCRX_XSTEP(1) MCRX_XSTEP(2) MCRX_XSTEP(4) MCRX_XSTEP(8) MCRX_XSTEP(16) MCRX_XSTEP(32)
#undef MCRX_XSTEP
    return v;
}
// inclusive prefix sum over the 64 lanes (DPP Kogge-Stone inside rows, row broadcasts across)
__device__ __forceinline__ float wave_scan_fast(float v)
{
    v += dpp_mov<0x111>(v);                      // row_shr:1, lanes without a source add 0
    v += dpp_mov<0x112>(v);
    v += dpp_mov<0x114>(v);
    v += dpp_mov<0x118>(v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));  // row_bcast:15
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));  // row_bcast:31
    return v;
}
// the same inside the first row only (lanes 0..15 carry the data, the rest hold zeros or are ignored)
__device__ __forceinline__ float row_scan_fast(float v)
{
    v += dpp_mov<0x111>(v); v += dpp_mov<0x112>(v); v += dpp_mov<0x114>(v); v += dpp_mov<0x118>(v);
    return v;
}
__device__ __forceinline__ float row_total_dpp(float v)      // total of lanes 0..15
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, row_scan_fast(v)), 15));
}
// wave total through the DPP scan and one v_readlane (no LDS crossbar): wave-uniform result
__device__ __forceinline__ float wave_total_dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_scan_fast(v)), 63));
}
__device__ __forceinline__ float wave_sum(float v) { return wave_total_dpp(v); }
__device__ __forceinline__ cfd wave_csum(cfd v) { return make_float2(wave_total_dpp(v.x), wave_total_dpp(v.y)); }
// atan2 for finite arguments: degree-7 minimax in t^2 on [0, 1] (max error 1.2e-7 evaluated in
// float), octant folding, no special-case handling (0, 0 -> 0)
__device__ __forceinline__ float atan2_fast(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float t = mn * __builtin_amdgcn_rcpf(fmaxf(mx, 1e-37f));
    const float s = t * t;
    float p = -4.0545611898e-03f;
    p = fmaf(p, s, 2.1862935605e-02f); p = fmaf(p, s, -5.5912294062e-02f); p = fmaf(p, s, 9.6421949108e-02f);
    p = fmaf(p, s, -1.3908628615e-01f); p = fmaf(p, s, 1.9946565475e-01f); p = fmaf(p, s, -3.3329860772e-01f);
    p = fmaf(p, s, 9.9999933558e-01f);
    float r = t * p;
    r = ay > ax ? 1.5707963267948966f - r : r;
    r = x < 0.f ? 3.14159265358979323846f - r : r;
    return copysignf(r, y);
}
// e^{-j 2 pi rev} on the transcendental unit (see sincos_u32_hw)
__device__ __forceinline__ cfd rot_down(cfd x, float rev)
{
    const float s = __builtin_amdgcn_sinf(rev), c = __builtin_amdgcn_cosf(rev);
    return make_float2(fmaf(x.x, c, x.y * s), fmaf(x.y, c, -(x.x * s)));
}
__device__ __forceinline__ float u32rev(uint32_t th) { return (float)(int32_t)th * 2.3283064365386963e-10f; }   // 2^-32
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ cfd shfl_c(cfd v, int src) { return make_float2(__shfl(v.x, src, 64), __shfl(v.y, src, 64)); }

}  // namespace mcrx
