// synth.hip -- fused 2N-channel polyphase synthesis bank + NCO mix-up for gfx950: the transmit mirror of channelizer.hip.
//
// Replaces, per block of K = 2N wideband samples, the reference's
//   firpfbch_crcf_synthesizer_execute            (lib/multichanneltx.cc:213)
//   nco_crcf_mix_up / nco_crcf_step              (lib/multichanneltx.cc:216-219)
// for channel-rate input that is already in HBM as (channel, tile) granules
//   in[g][tile][c][8],  channel = g*Cg + c        (mctx_hip_traffic_tiles; the layout an all-to-all delivers)
//
// Math (liquid firpfbch synthesizer, K channels of which the lower N carry signal, p = 26 taps per branch):
//   v_b[i] = sum_{k<N} X_b[k] * exp(+j 2 pi i k / K)              inverse FFT, unnormalised
//   y_b[i] = sum_{j<p} h[i + j*K] * v_{b-j}[i]                    column FIR down the time axis
//   out    = y_b[i] * exp(+j * (b*K + i) * dtheta) * gain         32-bit phase, exact closed form
//
// Mapping: a workgroup owns NS time slabs; rounds of 8 blocks (one granule per channel) are written into an LDS tile
// and transformed in place by the same radix-4 / register stages as the analysis bank (forward transform of the
// conjugate); a thread then owns C adjacent columns i of its slab and keeps the last 25 + 8 transform outputs of each
// in a register window, so every granule is loaded from HBM once and every output sample stored once: 4 B read + 8 B
// written per wideband sample.  A slab starts 32 blocks early to fill its window (those rounds store nothing).
#include "devmath.h"
#include "kernels.h"
#include "fftplan.h"

#include <cstdlib>

namespace mcrx {

#define SY_R 8          // blocks per round == MCRX_TILE
#define SY_P 26         // synthesis taps per branch (m = 13)
#define SY_H (SY_P - 1) // history blocks
#define SY_WARM 32      // blocks a slab starts early (>= SY_H, whole rounds)

template <int K, int C, int T>
__global__ __launch_bounds__(T) void synth_kernel(SynthArgs a)
{
    constexpr int TPS = K / C;              // threads per slab
    constexpr int NS = T / TPS;             // slabs per workgroup
    constexpr int N = K / 2;
    constexpr int S = Plan<K>::S, F = Plan<K>::F, ROWP = Plan<K>::ROWP;
    static_assert(TPS * C == K && NS * TPS == T && NS >= 1, "bad synthesis geometry");
    extern __shared__ __attribute__((aligned(16))) float2 tile[];     // [NS][SY_R][ROWP]

    const int tid = threadIdx.x;
    const int sl = tid / TPS, cgi = tid % TPS;
    const int n0 = cgi * C;
    const long long slab = (long long)blockIdx.x * NS + sl;
    const long long bs = slab * (long long)a.slab_blocks;        // first block of my slab (local index)

    // radix-4 stage twiddles (see channelizer.hip)
    float2 tw[S > 0 ? S : 1][3];
#pragma unroll
    for (int st = 0; st < S; st++) {
        const int L = K >> (2 * st), q4 = L >> 2;
        const int pos = tid % q4;
#pragma unroll
        for (int r = 1; r <= 3; r++) {
            float sn, cs; sincos_u32((uint32_t)(r * pos) * (uint32_t)(4294967296.0 / L), sn, cs);
            tw[st][r - 1] = make_float2(cs, -sn);
        }
    }
    // ---- round-invariant addresses
    constexpr int NBF4 = NS * SY_R * (K / 4), NG = NS * SY_R * (K / F);
    static_assert(S == 0 || (NBF4 % T == 0 && T % (K / 4) == 0), "a thread's butterflies must differ by whole rows");
    constexpr int BTRIPS = S > 0 ? NBF4 / T : 0, BSTEP = S > 0 ? (T / (K / 4)) * ROWP : 0;
    int fa[S > 0 ? S : 1];
#pragma unroll
    for (int st = 0; st < S; st++) {
        const int L = K >> (2 * st), q4 = L >> 2;
        const int f = tid / (K / 4), j = tid % (K / 4);
        fa[st] = f * ROWP + pad<K>((j / q4) * L + j % q4);
    }
    constexpr int GTRIPS = (NG + T - 1) / T;
    int fg[GTRIPS];
#pragma unroll
    for (int i = 0; i < GTRIPS; i++) { const int g = tid + i * T; fg[i] = (g / (K / F)) * ROWP + (g % (K / F)) * (F + 1); }
    // the input channels this thread stages (channel k = cgi + i*TPS < N), the transform outputs it reads back
    constexpr int CPT = (N + TPS - 1) / TPS;            // channels per thread: 1 for C = 2, at most 1 for C = 1
    const bool has_ch = cgi < N;                        // (C = 1: only the first N threads of a slab stage a channel)
    const uint32_t gch = has_ch ? (uint32_t)cgi / a.cg : 0u, cch = has_ch ? (uint32_t)cgi % a.cg : 0u;
    const float4 *gin = reinterpret_cast<const float4 *>(a.in) + ((size_t)gch * a.ntiles * a.cg + cch) * (SY_R / 2);
    const size_t gstep = (size_t)a.cg * (SY_R / 2);    // float4 units between consecutive tiles of a channel group
    static_assert(CPT == 1, "one input channel per thread");
    int vpos[C];
#pragma unroll
    for (int c = 0; c < C; c++) vpos[c] = sl * SY_R * ROWP + pad<K>(dif_pos<K>(n0 + c));
    float2 *myrow = tile + sl * SY_R * ROWP;

    // the thread's 26 x C taps are fetched again every round (L2-resident, coalesced) right before the filter: held in
    // registers across the transform they would not fit beside the window
    const float *tapp = a.taps + n0;

    // granule of tile t (8 time samples of my channel); zeros outside the stream
    auto load_tile = [&](long long t, float4 (&dst)[SY_R / 2]) {
        const bool ok = has_ch && t >= 0 && t < (long long)a.ntiles;
        const float4 *src = gin + (ok ? (size_t)t * gstep : 0);
#pragma unroll
        for (int i = 0; i < SY_R / 2; i++) dst[i] = src[i];
        if (!ok) {
#pragma unroll
            for (int i = 0; i < SY_R / 2; i++) dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    float2 w[SY_H + SY_R][C];               // w[0..24] history (oldest first), w[25..32] the round's transform outputs
#pragma unroll
    for (int i = 0; i < SY_H + SY_R; i++)
#pragma unroll
        for (int c = 0; c < C; c++) w[i][c] = make_float2(0.f, 0.f);

    const uint32_t dth = a.dtheta;
    const long long t_first = bs / SY_R - SY_WARM / SY_R;
    const int rounds = (int)(a.slab_blocks / SY_R) + SY_WARM / SY_R;
    float4 x[SY_R / 2];
    load_tile(t_first, x);
    for (int rd = 0; rd < rounds; rd++) {
        const long long t = t_first + rd;
        const long long b0 = t * SY_R;                  // first block of the round
        // ---- stage the round: conj(X) in the lower half of every row, zeros above
        if (has_ch) {
#pragma unroll
            for (int i = 0; i < SY_R / 2; i++) {
                myrow[(2 * i) * ROWP + pad<K>(cgi)] = make_float2(x[i].x, -x[i].y);
                myrow[(2 * i + 1) * ROWP + pad<K>(cgi)] = make_float2(x[i].z, -x[i].w);
                myrow[(2 * i) * ROWP + pad<K>(cgi + N)] = make_float2(0.f, 0.f);
                myrow[(2 * i + 1) * ROWP + pad<K>(cgi + N)] = make_float2(0.f, 0.f);
            }
        }
        load_tile(t + 1, x);                            // next round's granule flies under the transform
        lds_barrier();
        if constexpr (S > 0) {
#pragma unroll
            for (int st = 0; st < S; st++) {
                const int L = K >> (2 * st), q4 = L >> 2;
                const int D = q4 + q4 / F;
#pragma unroll
                for (int i = 0; i < BTRIPS; i++) {
                    float2 *p = tile + fa[st] + i * BSTEP;
                    const float2 x0 = p[0], x1 = p[D], x2 = p[2 * D], x3 = p[3 * D];
                    const float2 a0 = cadd(x0, x2), a1 = csub(x0, x2), a2 = cadd(x1, x3), a3 = cmulnj(csub(x1, x3));
                    p[0] = cadd(a0, a2);
                    p[D] = cmul(cadd(a1, a3), tw[st][0]);
                    p[2 * D] = cmul(csub(a0, a2), tw[st][1]);
                    p[3 * D] = cmul(csub(a1, a3), tw[st][2]);
                }
                lds_barrier();
            }
        }
        {
#pragma unroll
            for (int i = 0; i < GTRIPS; i++) {
                if (NG % T != 0 && tid + i * T >= NG) break;
                float2 *p = tile + fg[i];
                float2 v[F];
#pragma unroll
                for (int m = 0; m < F; m++) v[m] = p[m];
                fft_reg<F>(v);
#pragma unroll
                for (int m = 0; m < F; m++) p[bitrev_c(m, Log2<F>::v)] = v[m];
            }
            lds_barrier();
        }
        // ---- my columns of the eight inverse transforms
#pragma unroll
        for (int r = 0; r < SY_R; r++)
#pragma unroll
            for (int c = 0; c < C; c++) { const float2 z = tile[vpos[c] + r * ROWP]; w[SY_H + r][c] = make_float2(z.x, -z.y); }
        lds_barrier();                                  // (the next round overwrites the tile)
        // ---- synthesis FIR, oscillator, gain
        if (b0 + SY_R <= bs || b0 + SY_R <= (long long)a.out_first) goto shift;       // warm-up round / lead blocks: nothing to emit
        {
        float tap[SY_P][C];
#pragma unroll
        for (int j = 0; j < SY_P; j++) {
            if constexpr (C == 2) { const float2 tt = *reinterpret_cast<const float2 *>(tapp + j * K); tap[j][0] = tt.x; tap[j][1] = tt.y; }
            else tap[j][0] = tapp[j * K];
        }
#pragma unroll
        for (int r = 0; r < SY_R; r++) {
            const long long b = b0 + r;
            const bool emit = b >= (long long)a.out_first && b >= bs && b < bs + (long long)a.slab_blocks && b < (long long)a.nblocks;
            if (!emit) continue;                        // (uniform per slab: warm-up rounds, the lead blocks of a call)
            float2 y[C];
#pragma unroll
            for (int c = 0; c < C; c++) y[c] = make_float2(0.f, 0.f);
#pragma unroll
            for (int j = SY_P - 1; j >= 0; j--) {       // oldest first, like the window dot product
#pragma unroll
                for (int c = 0; c < C; c++) {
                    y[c].x += tap[j][c] * w[SY_H + r - j][c].x;
                    y[c].y += tap[j][c] * w[SY_H + r - j][c].y;
                }
            }
            float2 o[C];
#pragma unroll
            for (int c = 0; c < C; c++) {
                const uint32_t ts = a.first_sample_lo + (uint32_t)((unsigned long long)b * K + (unsigned)(n0 + c));
                const float2 m = mix_up(y[c], ts * dth);
                o[c] = make_float2(m.x * a.gain, m.y * a.gain);
            }
            float2 *dst = a.out + (size_t)(b - (long long)a.out_first) * K + n0;
            if constexpr (C == 2) *reinterpret_cast<float4 *>(dst) = make_float4(o[0].x, o[0].y, o[1].x, o[1].y);
            else dst[0] = o[0];
        }
        }
shift:
#pragma unroll
        for (int i = 0; i < SY_H; i++)
#pragma unroll
            for (int c = 0; c < C; c++) w[i][c] = w[i + SY_R][c];
    }
}

template <int K, int C, int T>
static hipError_t synth_one(const SynthArgs &a, hipStream_t st)
{
    constexpr int NS = T / (K / C);
    const size_t lds = (size_t)(NS * SY_R * Plan<K>::ROWP) * sizeof(float2);
    const long long nslabs = ((long long)a.nblocks + a.slab_blocks - 1) / a.slab_blocks;
    const unsigned grid = (unsigned)((nslabs + NS - 1) / NS);
    if (grid == 0) return hipSuccess;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)synth_kernel<K, C, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((synth_kernel<K, C, T>), dim3(grid), dim3(T), lds, st, a);
    return hipGetLastError();
}

int synth_supported(unsigned K) { return K >= 4 && K <= 1024 && (K & (K - 1)) == 0; }

// blocks per slab: a whole number of rounds, about two slabs per compute unit, never so short that the 32 warm-up
// blocks dominate
uint32_t synth_auto_slab(size_t nblocks, unsigned ncu)
{
    size_t per = (nblocks + 2 * (size_t)ncu - 1) / (2 * (size_t)ncu);
    per = (per + SY_R - 1) / SY_R * SY_R;
    if (per < 128) per = 128;
    return (uint32_t)per;
}

hipError_t synth_launch(unsigned K, const SynthArgs &a, hipStream_t st)
{
    if (a.nblocks % SY_R || a.slab_blocks % SY_R || a.slab_blocks == 0) return hipErrorInvalidValue;
    switch (K) {
    case 4:    return synth_one<4, 2, 256>(a, st);
    case 8:    return synth_one<8, 2, 256>(a, st);
    case 16:   return synth_one<16, 2, 256>(a, st);
    case 32:   return synth_one<32, 2, 256>(a, st);
    case 64:   return synth_one<64, 2, 256>(a, st);
    case 128:  return synth_one<128, 2, 256>(a, st);
    case 256:  return synth_one<256, 2, 256>(a, st);
    case 512:  return synth_one<512, 2, 256>(a, st);
    case 1024: return synth_one<1024, 2, 512>(a, st);
    default:   return hipErrorInvalidValue;
    }
}

}  // namespace mcrx
