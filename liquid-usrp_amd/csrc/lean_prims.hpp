// lean_prims.hpp -- the packed-f32 / LDS-crossbar building blocks of the lean M = 64 kernels (payload_lean.hpp: the payload workers;
// acq_lean.hpp: the segment waves of the acquisition).  Included inside namespace mcrx, after devmath.h.
#pragma once
namespace lean {

typedef float v2f __attribute__((ext_vector_type(2)));       // a complex sample in an aligned register pair: the packed-f32 VALU takes it whole
// a (c + j s), tw = (c, s): one packed multiply, one packed fma with the halves of `a` swapped and the low one negated
__device__ __forceinline__ v2f cmul_pk(v2f a, v2f tw)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(tw));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(tw), "v"(t));
    return r;
}
// a (c - j s)
__device__ __forceinline__ v2f cmulc_pk(v2f a, v2f tw)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(tw));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(tw), "v"(t));
    return r;
}
// a e^{-j 2 pi rev} on the transcendental unit
__device__ __forceinline__ v2f rot_down_pk(v2f a, float rev)
{
    v2f cs; cs.x = __builtin_amdgcn_cosf(rev); cs.y = __builtin_amdgcn_sinf(rev);
    return cmulc_pk(a, cs);
}
// sg x + p, sg = the low (HI = 0) or high (HI = 1) half of `sgp` for both components
template <int HI>
__device__ __forceinline__ v2f bfly_pk(v2f x, v2f sgp, v2f p)
{
    v2f r;
    if constexpr (HI == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x), "v"(sgp), "v"(p));
    else                   asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(x), "v"(sgp), "v"(p));
    return r;
}

template <int H>
__device__ __forceinline__ float xch(float v, int bp32)         // v of lane l ^ H, through the LDS crossbar
{
    if constexpr (H == 32) return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bp32, __builtin_bit_cast(int, v)));
    else return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (H << 10) | 0x1F));
}
// one radix-2 DIF stage across lanes H apart: lower lanes x + x', upper lanes (x' - x) W
template <int H, int XB, int HI>
__device__ __forceinline__ v2f stage(v2f x, v2f sgp, v2f tw, int bp32)
{
    v2f u;
    if constexpr ((XB & H) != 0) { v2f p; p.x = xch<H>(x.x, bp32); p.y = xch<H>(x.y, bp32); u = bfly_pk<HI>(x, sgp, p); }
    else { const float sg = HI ? sgp.y : sgp.x; u.x = bfly_leg<H>(x.x, sg); u.y = bfly_leg<H>(x.y, sg); }
    if constexpr (H == 1) return u;
    else return cmul_pk(u, tw);
}

}  // namespace lean
