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


// ---- the lane <-> subcarrier map and the transform, per symbol width.  MM = 64: six radix-2 DIF stages, lane l ends with subcarrier
// bitrev6(l).  MM = 48 (the reference applications' default, src/multichannel_rx.cc:93-95; round 5): 48 = 3 x 16 -- lane l = 16 a + b
// holds sample n = l (lanes 48..63 idle, zero); one radix-3 stage across the three rows (X' = sum_a x[16 a + b] W3^(a a'), times
// W48^(b a'): the three inputs come through the LDS crossbar), then the 16-point transform inside every DPP row = the LAST FOUR stages
// of the 64-point one, same twiddles; lane 16 a' + bitrev4(k2) ends with subcarrier a' + 3 k2.
template <int MM> __device__ __forceinline__ int lane_k(int l)          // subcarrier held by lane l after the transform (-1: idle lane)
{
    if constexpr (MM == 64) return (int)(__brev((unsigned)l) >> 26);
    else return l < 48 ? (l >> 4) + 3 * (int)(__brev((unsigned)(l & 15)) >> 28) : -1;
}
template <int MM> __device__ __forceinline__ int k_lane(int k)          // ... and the lane that holds subcarrier k
{
    if constexpr (MM == 64) return (int)(__brev((unsigned)k) >> 26);
    else return 16 * (k % 3) + (int)(__brev((unsigned)(k / 3)) >> 28);
}
struct Radix3 { v2f w1, w2, t; int a0, a1, a2; };                       // lane constants of the 48-point transform's first stage
__device__ __forceinline__ Radix3 radix3_consts(int l)
{
    Radix3 r;
    const int a = l >> 4, b = l & 15;
    const float r1 = (float)(a % 3) * (1.0f / 3.0f), r2 = (float)((2 * a) % 3) * (1.0f / 3.0f), rt = (float)(b * (a % 3)) * (1.0f / 48.0f);
    r.w1.x = __builtin_amdgcn_cosf(r1); r.w1.y = -__builtin_amdgcn_sinf(r1);
    r.w2.x = __builtin_amdgcn_cosf(r2); r.w2.y = -__builtin_amdgcn_sinf(r2);
    r.t.x = __builtin_amdgcn_cosf(rt);  r.t.y = -__builtin_amdgcn_sinf(rt);
    r.a0 = b << 2; r.a1 = (16 + b) << 2; r.a2 = (32 + b) << 2;
    return r;
}
template <int XB>
__device__ __forceinline__ v2f fft48(v2f x, const v2f (&tw)[6], const v2f (&sgp)[3], const Radix3 &r3, int bp32, int l)
{
    // (every lane asks its three inputs by address: the gathers stand outside any lane condition -- a masked source lane reads as zero)
    const float xr = x.x, xi = x.y;
    v2f x0, x1, x2;
    x0.x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(r3.a0, __builtin_bit_cast(int, xr)));
    x0.y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(r3.a0, __builtin_bit_cast(int, xi)));
    x1.x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(r3.a1, __builtin_bit_cast(int, xr)));
    x1.y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(r3.a1, __builtin_bit_cast(int, xi)));
    x2.x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(r3.a2, __builtin_bit_cast(int, xr)));
    x2.y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(r3.a2, __builtin_bit_cast(int, xi)));
    const v2f p1 = cmul_pk(x1, r3.w1), p2 = cmul_pk(x2, r3.w2);
    v2f y; y.x = (x0.x + p1.x) + p2.x; y.y = (x0.y + p1.y) + p2.y;
    y = cmul_pk(y, r3.t);
    if (l >= 48) { y.x = 0.f; y.y = 0.f; }
    y = stage<8, XB, 0>(y, sgp[1], tw[2], bp32);
    y = stage<4, XB, 1>(y, sgp[1], tw[3], bp32);
    y = stage<2, XB, 0>(y, sgp[2], tw[4], bp32);
    y = stage<1, XB, 1>(y, sgp[2], tw[5], bp32);
    return y;
}
template <int XB>
__device__ __forceinline__ v2f fft64(v2f x, const v2f (&tw)[6], const v2f (&sgp)[3], int bp32)
{
    x = stage<32, XB, 0>(x, sgp[0], tw[0], bp32);
    x = stage<16, XB, 1>(x, sgp[0], tw[1], bp32);
    x = stage<8, XB, 0>(x, sgp[1], tw[2], bp32);
    x = stage<4, XB, 1>(x, sgp[1], tw[3], bp32);
    x = stage<2, XB, 0>(x, sgp[2], tw[4], bp32);
    x = stage<1, XB, 1>(x, sgp[2], tw[5], bp32);
    return x;
}

}  // namespace lean
