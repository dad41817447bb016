// fftplan.h -- the in-LDS transform shared by the analysis bank (channelizer.hip) and the synthesis bank (synth.hip):
// S radix-4 decimation-in-frequency stages through an LDS row while the sub-transform is larger than 16 points, then
// F-point (F = 2..16) transforms in registers, one group per thread; rows are padded by one element per F so that the
// register stage, whose lanes are F elements apart, is bank-conflict free.
#pragma once
#include "devmath.h"

namespace mcrx {

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain
// vmcnt, so the prefetched IQ loads and the granule stores stay in flight across it.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int K> struct Log2 { enum { v = 1 + Log2<K / 2>::v }; };
template <> struct Log2<1> { enum { v = 0 }; };

// transform plan: S radix-4 LDS stages, then F-point register transforms
template <int K> struct Plan {
    static constexpr int stages() { int L = K, s = 0; while (L > 16) { L /= 4; s++; } return s; }
    static constexpr int final_size() { int L = K; while (L > 16) L /= 4; return L; }
    enum { S = stages(), F = final_size(), RL = K + K / F, ROWP = RL + 1 };
};
// element index -> padded LDS index within a row
template <int K> __device__ __forceinline__ int pad(int e) { return e + e / Plan<K>::F; }
// position of bin k after the S DIF stages followed by natural-order F-point transforms
template <int K>
__device__ __forceinline__ int dif_pos(int k)
{
    int L = K, pos = 0;
#pragma unroll
    for (int s = 0; s < Plan<K>::S; s++) { pos += (k & 3) * (L >> 2); k >>= 2; L >>= 2; }
    return pos + k;
}

// exp(-j 2 pi k / 16), k = 0..7
__device__ __forceinline__ float2 w16(int k)
{
    const float c[8] = { 1.0f, 0.92387953251f, 0.70710678119f, 0.38268343236f, 0.0f, -0.38268343236f, -0.70710678119f, -0.92387953251f };
    const float s[8] = { 0.0f, -0.38268343236f, -0.70710678119f, -0.92387953251f, -1.0f, -0.92387953251f, -0.70710678119f, -0.38268343236f };
    return make_float2(c[k], s[k]);
}
constexpr int bitrev_c(int i, int bits) { int r = 0; for (int b = 0; b < bits; b++) if (i & (1 << b)) r |= 1 << (bits - 1 - b); return r; }

// F-point forward DFT in registers: natural order in, bit-reversed order out.  LOWER: only the outputs X[0 .. F/2-1] are
// wanted (the receiver keeps bins 0 .. N-1, which sit in the lower half of every final group): they are the sums of the last
// stage -- v[i] for even i -- so its differences are not formed
template <int F, bool LOWER = false>
__device__ __forceinline__ void fft_reg(float2 (&v)[F])
{
#pragma unroll
    for (int h = F / 2; h >= 1; h >>= 1) {
#pragma unroll
        for (int i = 0; i < F; i++) {
            if ((i & h) == 0) {
                const float2 u = v[i], w = v[i + h];
                v[i] = cadd(u, w);
                if (LOWER && h == 1) continue;
                const float2 d = csub(u, w);
                const int tk = (i & (h - 1)) * (8 / h);       // W_{2h}^{i mod h} as a power of W_16
                if (tk == 0) v[i + h] = d;
                else if (tk == 4) v[i + h] = cmulnj(d);
                else v[i + h] = cmul(d, w16(tk));
            }
        }
    }
}   // result: v[i] holds X[bitrev(i)]; callers store v[i] at index bitrev_c(i, log2 F)

}  // namespace mcrx
