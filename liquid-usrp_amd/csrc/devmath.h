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
__device__ __forceinline__ cfd cmulc(cfd a, cfd b)      // a * conj(b)
{ return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
__device__ __forceinline__ cfd cscale(cfd a, float g) { return make_float2(a.x * g, a.y * g); }
__device__ __forceinline__ cfd cmulnj(cfd a) { return make_float2(a.y, -a.x); }      // a * (-j)

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

// wave64 all-reduce sums
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ cfd wave_csum(cfd v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { v.x += __shfl_xor(v.x, o, 64); v.y += __shfl_xor(v.y, o, 64); }
    return v;
}
__device__ __forceinline__ cfd shfl_c(cfd v, int src) { return make_float2(__shfl(v.x, src, 64), __shfl(v.y, src, 64)); }

}  // namespace mcrx
