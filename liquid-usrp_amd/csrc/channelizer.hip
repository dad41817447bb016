// channelizer.hip -- fused NCO mix-down + 2N-channel polyphase analysis bank for gfx950.
//
// Replaces, per block of K = 2N wideband samples, the reference's
//   nco_crcf_mix_down / nco_crcf_step            (lib/multichannelrx.cc:163-164)
//   firpfbch_crcf_analyzer_execute               (lib/multichannelrx.cc:188)
// and keeps only bins 0..N-1, the ones RunChannelizer hands to the synchronizers (:193-194).
//
// Math (liquid firpfbch analyzer, K channels, p = 14 taps per branch):
//   u[t]   = x[t] * exp(-j * t * dtheta)                       32-bit phase, exact closed form
//   V_b[n] = sum_{j<p} h[K-1-n + j*K] * u[(b-j)*K + n]          column FIR down the time axis
//   y_b[k] = sum_n V_b[n] * exp(-j 2 pi n k / K),  k < N        forward FFT, unnormalised
//
// Mapping: a workgroup owns NS time slabs; a thread owns C adjacent columns n of one slab
// and walks down the time axis with a register sliding window (13 history + 8 new blocks),
// so every IQ sample is loaded from HBM exactly once (plus a 13-block halo per slab).
// Rounds of 8 blocks land in an LDS tile, are transformed in place (radix-4 DIF, twiddles
// staged in LDS), and the kept bins leave as 64-byte (channel, tile) granules:
//   out[g][tile][c][8],  channel = g*Cg + c  -- the layout the synchronizer streams and
// the per-destination chunking an xGMI all-to-all needs.
// HBM-bound by design: 8 B read + 4 B written per wideband sample.
#include "devmath.h"
#include "kernels.h"

namespace mcrx {

#define CH_R 8          // blocks per round == MCRX_TILE
#define CH_P 14         // taps per branch (m = 7)
#define CH_H (CH_P - 1) // history blocks

template <int K> struct Log2 { enum { v = 1 + Log2<K / 2>::v }; };
template <> struct Log2<1> { enum { v = 0 }; };

// position of bin k after the in-place mixed radix-(4,...,4[,2]) DIF
template <int K>
__device__ __forceinline__ int dif_pos(int k)
{
    int L = K, pos = 0;
#pragma unroll
    for (int s = 0; s < Log2<K>::v / 2; s++) { pos += (k & 3) * (L >> 2); k >>= 2; L >>= 2; }
    if (Log2<K>::v & 1) pos += (k & 1);
    return pos;
}

template <int K, int C, int T>
__global__ __launch_bounds__(T) void channelizer_kernel(ChanArgs a)
{
    constexpr int TPS = K / C;              // threads per slab
    constexpr int NS = T / TPS;             // slabs per workgroup
    constexpr int ROW = K + 1;              // padded row (complex elements)
    constexpr int N = K / 2;
    static_assert(TPS * C == K && NS * TPS == T && NS >= 1, "bad channelizer geometry");

    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    float2 *tile = lds;                                 // [NS][CH_R][ROW]
    float2 *twid = lds + NS * CH_R * ROW;               // [K]  W_K^k

    const int tid = threadIdx.x;
    const int sl = tid / TPS, cg = tid % TPS;
    const int n0 = cg * C;
    const long long slab = (long long)blockIdx.x * NS + sl;
    const long long bs = slab * (long long)a.slab_blocks;        // first block of my slab
    const bool active = bs < (long long)a.nblocks;

    for (int k = tid; k < K; k += T) {
        float s, c; sincos_u32((uint32_t)k * (uint32_t)(4294967296.0 / K), s, c);
        twid[k] = make_float2(c, -s);
    }

    // taps: tap[j][c] = h[K-1-n + j*K]
    float tap[CH_P][C];
#pragma unroll
    for (int j = 0; j < CH_P; j++)
#pragma unroll
        for (int c = 0; c < C; c++) tap[j][c] = a.taps[(K - 1 - (n0 + c)) + j * K];

    const uint32_t dth = a.dtheta;
    const uint32_t t0 = a.first_sample_lo;

    auto load_block = [&](long long b, float2 (&dst)[C]) {
        // samples of block b (relative to a.x), columns n0..n0+C-1, NCO applied
        const float2 *src = nullptr;
        if (b >= 0) { if (b < (long long)a.nblocks) src = a.x + (size_t)b * K + n0; }
        else if (a.halo) src = a.halo + (size_t)(b + CH_H) * K + n0;
        if (src) {
            if constexpr (C == 2) {
                float4 v = *reinterpret_cast<const float4 *>(src);
                dst[0] = make_float2(v.x, v.y); dst[1] = make_float2(v.z, v.w);
            } else dst[0] = src[0];
            const uint32_t tt = t0 + (uint32_t)((long long)b * K + n0);
#pragma unroll
            for (int c = 0; c < C; c++) dst[c] = mix_down(dst[c], (tt + (uint32_t)c) * dth);
        } else {
#pragma unroll
            for (int c = 0; c < C; c++) dst[c] = make_float2(0.f, 0.f);
        }
    };

    // s[0..12] history (oldest first), s[13..20] the round's new blocks
    float2 s[CH_H + CH_R][C];
    if (active) {
#pragma unroll
        for (int i = 0; i < CH_H; i++) load_block(bs - CH_H + i, s[i]);
    }

    const int rounds = a.slab_blocks / CH_R;
    for (int rd = 0; rd < rounds; rd++) {
        const long long b0 = bs + (long long)rd * CH_R;
        if (active) {
#pragma unroll
            for (int r = 0; r < CH_R; r++) load_block(b0 + r, s[CH_H + r]);
#pragma unroll
            for (int r = 0; r < CH_R; r++) {
                float2 v[C];
#pragma unroll
                for (int c = 0; c < C; c++) v[c] = make_float2(0.f, 0.f);
#pragma unroll
                for (int j = CH_P - 1; j >= 0; j--) {           // oldest tap first, like a window dot product
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        v[c].x += tap[j][c] * s[CH_H + r - j][c].x;
                        v[c].y += tap[j][c] * s[CH_H + r - j][c].y;
                    }
                }
                float2 *row = tile + (sl * CH_R + r) * ROW + n0;
#pragma unroll
                for (int c = 0; c < C; c++) row[c] = v[c];
            }
#pragma unroll
            for (int i = 0; i < CH_H; i++)
#pragma unroll
                for (int c = 0; c < C; c++) s[i][c] = s[i + CH_R][c];
        }
        __syncthreads();

        // ---- NS*CH_R independent K-point FFTs in LDS, in place, radix-4 DIF
        constexpr int NBF4 = NS * CH_R * (K / 4);       // radix-4 butterflies per stage
        constexpr int LOG2K = Log2<K>::v;
        int L = K;
        if constexpr (LOG2K >= 2) {
#pragma unroll
        for (int st = 0; st < LOG2K / 2; st++) {
            const int q4 = L >> 2;
            for (int q = tid; q < NBF4; q += T) {
                const int f = q / (K / 4), j = q % (K / 4);
                const int grp = j / q4, pos = j % q4;
                float2 *base = tile + f * ROW + grp * L + pos;
                float2 x0 = base[0], x1 = base[q4], x2 = base[2 * q4], x3 = base[3 * q4];
                float2 a0 = cadd(x0, x2), a1 = csub(x0, x2), a2 = cadd(x1, x3), a3 = cmulnj(csub(x1, x3));
                float2 y0 = cadd(a0, a2), y1 = cadd(a1, a3), y2 = csub(a0, a2), y3 = csub(a1, a3);
                if (q4 > 1) {
                    const int ts = K / L;               // twiddle stride
                    y1 = cmul(y1, twid[pos * ts]);
                    y2 = cmul(y2, twid[2 * pos * ts]);
                    y3 = cmul(y3, twid[3 * pos * ts]);
                }
                base[0] = y0; base[q4] = y1; base[2 * q4] = y2; base[3 * q4] = y3;
            }
            __syncthreads();
            L >>= 2;
        }
        }
        if (LOG2K & 1) {                                 // final radix-2 stage (L == 2)
            constexpr int NBF2 = NS * CH_R * (K / 2);
            for (int q = tid; q < NBF2; q += T) {
                const int f = q / (K / 2), j = q % (K / 2);
                float2 *base = tile + f * ROW + 2 * j;
                float2 x0 = base[0], x1 = base[1];
                base[0] = cadd(x0, x1); base[1] = csub(x0, x1);
            }
            __syncthreads();
        }

        // ---- store bins 0..N-1 as (channel, tile) granules of 8 time samples (64 B)
        constexpr int NOPS = NS * N * (CH_R / 2);       // 16-byte stores per round
        for (int o = tid; o < NOPS; o += T) {
            const int osl = o / (N * (CH_R / 2)), rem = o % (N * (CH_R / 2));
            const int ch = rem / (CH_R / 2), rp = rem % (CH_R / 2);
            const long long oslab = (long long)blockIdx.x * NS + osl;
            const long long ob0 = oslab * (long long)a.slab_blocks + (long long)rd * CH_R;
            if (ob0 < (long long)a.nblocks) {
                const int pos = dif_pos<K>(ch);
                const float2 *src = tile + (osl * CH_R + 2 * rp) * ROW + pos;
                float2 v0 = src[0], v1 = src[ROW];
                const long long tl = ob0 / CH_R;
                const int g = ch / a.cg, c = ch % a.cg;
                float4 *dst = reinterpret_cast<float4 *>(
                    a.out + (((size_t)g * a.ntiles + (size_t)tl) * a.cg + c) * CH_R + 2 * rp);
                *dst = make_float4(v0.x, v0.y, v1.x, v1.y);
            }
        }
        __syncthreads();
    }
}

template <int K, int C, int T>
static hipError_t launch_one(const ChanArgs &a, hipStream_t st)
{
    constexpr int NS = T / (K / C);
    size_t lds = (size_t)(NS * CH_R * (K + 1) + K) * sizeof(float2);
    long long nslabs = ((long long)a.nblocks + a.slab_blocks - 1) / a.slab_blocks;
    unsigned grid = (unsigned)((nslabs + NS - 1) / NS);
    if (grid == 0) return hipSuccess;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)channelizer_kernel<K, C, T>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((channelizer_kernel<K, C, T>), dim3(grid), dim3(T), lds, st, a);
    return hipGetLastError();
}

int channelizer_supported(unsigned K)
{ return (K >= 2 && K <= 1024 && (K & (K - 1)) == 0) ? 1 : 0; }

hipError_t channelizer_launch(unsigned K, const ChanArgs &a, hipStream_t st)
{
    switch (K) {
    case 2:    return launch_one<2, 1, 256>(a, st);
    case 4:    return launch_one<4, 2, 256>(a, st);
    case 8:    return launch_one<8, 2, 256>(a, st);
    case 16:   return launch_one<16, 2, 256>(a, st);
    case 32:   return launch_one<32, 2, 256>(a, st);
    case 64:   return launch_one<64, 2, 256>(a, st);
    case 128:  return launch_one<128, 2, 256>(a, st);
    case 256:  return launch_one<256, 2, 256>(a, st);
    case 512:  return launch_one<512, 2, 256>(a, st);
    case 1024: return launch_one<1024, 2, 512>(a, st);
    default:   return hipErrorInvalidValue;
    }
}

}  // namespace mcrx
