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
// Rounds of 8 blocks land in an LDS tile and are transformed in place:
//   * radix-4 DIF stages through LDS while the sub-transform is larger than 16 points
//     (twiddles are loop invariant per thread and live in registers);
//   * the remaining F-point (F = 2..16) transforms entirely in registers, one group per
//     thread; the tile is padded by one element per F so that this stage, whose lanes are
//     F elements apart, is LDS bank-conflict free.
// The kept bins leave as 128-byte (channel, tile) granules of 16 time samples
//   out[g][tile][c][16],  channel = g*Cg + c
// -- one channel per cache line, so a synchronizer wave that streams one channel's time series moves
// only that channel's bytes (with 64-byte granules two channels shared a line and the payload workers
// fetched 1.85 x their algorithmic bytes: profiles/r3_v4_traffic.json) -- and the per-destination chunking an
// xGMI all-to-all needs.  A round of 8 blocks fills one half of every granule; the next round of the same
// workgroup fills the other (slabs are whole tiles), so the halves meet in that XCD's L2.
// HBM-bound by design: 8 B read + 4 B written per wideband sample.
//
// Round 6: the kernel is a template of the taps per column P, and the oversampled front end (cfg.front_end = 1: liquid's
// firpfbch2 analysis bank, 2N channels at twice the channel rate, + a half-band decimator per kept channel) runs through it
// too.  That chain is linear and, per kept channel, time invariant at the block rate, so it IS a critically sampled
// polyphase bank: out_k[c] = FFT_K( V_k[(n - s) mod K] )[c],  V_k[n] = sum_{d < 28} G[n][d] u[(k - d) K + n],  s = K/2 + 1,
// with the composite taps G = (both phases of the firpfbch2 prototype, the even steps convolved with the half-band branch
// filter, the odd step delayed by seven blocks; design.hpp: pfb2_composite_taps) -- ONE transform per block instead of
// two, no rate-2 intermediate in HBM, the same 12 algorithmic bytes per wideband sample as the reference's bank instead of
// 8 + 8 (oscillator pass) + 8 + 16 (bank at rate 2) + 8 + 4 (adapter) = 52 over three kernels.  The rotation by s is where a
// column's FIR output lands in the LDS tile; the taps of 1024 columns x 28 do not fit LDS beside the tile (112 + 68 KB), so
// the newest TL = 22 of every column live in LDS and the oldest six come from a table in L2 at the start of every round.
#include "devel.h"
#include "devmath.h"
#include "kernels.h"
#include "devscope.hpp"

namespace mcrx {

#ifndef CH_NT_STORE
#define CH_NT_STORE 1   /* the half-granule stores are non-temporal: the line's other half arrives a round later and nobody on this CU reads
                           either -- 0.591 -> 0.577 ms alone, value 170.4 -> 175.0 in alternating runs of one call (scratch/r4u.sh) */
#endif
#ifndef CH_NT_LOAD
#define CH_NT_LOAD 1    /* ... and so are the loads of the IQ blocks, read once (0.578 -> 0.556-0.571 ms in the same kind of run; small) */
#endif
#define CH_R 8          // blocks per round == half a (channel, tile) granule
static_assert(MCRX_TILE_S == 2 * CH_R, "two rounds fill one granule");
#define CH_P_REF 14     // taps per column of the reference's bank (m = 7)
#ifndef CH_RH
#define CH_RH 4         // outputs of a round the many-tap FIR forms per pass over the taps
#endif
#define CH_P_OVS 28     // ... of the composite bank of the oversampled front end (14 + 14 - 1 through the half-band branch, one more for the half-block offset)

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain
// vmcnt, so the prefetched IQ loads and the granule stores stay in flight across it.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int K> struct Log2 { enum { v = 1 + Log2<K / 2>::v }; };
template <> struct Log2<1> { enum { v = 0 }; };

// transform plan: S radix-R LDS stages, then F-point register transforms.  K = 1024 = 8 x 8 x 16 (and 512 = 8 x 8 x 8) goes through LDS in two
// radix-8 stages (three radix-4 ones until round 3: a third fewer LDS instructions per round and a barrier less);
// the other sizes keep radix 4, whose stage count is the same or smaller for them.
#ifndef CH_RADIX8
#define CH_RADIX8 1
#endif
template <int K> struct Plan {
    enum { R = (CH_RADIX8 && (K == 1024 || K == 512)) ? 8 : 4, LR = R == 8 ? 3 : 2 };
    static constexpr int stages() { int L = K, s = 0; while (L > 16) { L /= R; s++; } return s; }
    static constexpr int final_size() { int L = K; while (L > 16) L /= R; return L; }
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
    for (int s = 0; s < Plan<K>::S; s++) { pos += (k & (Plan<K>::R - 1)) * (L >> Plan<K>::LR); k >>= Plan<K>::LR; L >>= Plan<K>::LR; }
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
                else v[i + h] = cmul_fx(d, w16(tk));
            }
        }
    }
}   // result: v[i] holds X[bitrev(i)]; callers store v[i] at index bitrev_c(i, log2 F)

// Geometry of one instantiation: slabs per workgroup, the LDS tile, and how many of a column's P taps live in LDS (the newest
// TL; the older ones are fetched from the tap table -- L2 -- when a round starts).  160 KB of LDS per workgroup.
#ifndef CH_WHOLE_LINES
#define CH_WHOLE_LINES 1
#endif
template <int K, int C, int T, int P> struct Geo {
    enum { TPS = K / C, NS = T / TPS, TILE_F2 = NS * CH_R * Plan<K>::ROWP };
    // WL: the kept bins of every EVEN round wait in LDS (`stash[8 rows][K/2 + 1]`) and leave with the odd round's as whole 128-byte granules --
    // a 64-byte half of a line, written twice, moves at 3.6 TB/s where whole lines move at 5.4 (scratch/r6/ubench/halfline.hip).  Where tile,
    // all taps and the stash fit the workgroup's 160 KB together: the reference's bank at K = 1024.
    enum { WL = (CH_WHOLE_LINES && K == 1024 && C == 2 && T == 512 && P == CH_P_REF) ? 1 : 0, STASH_ROW = K / 2 + 1, STASH_F2 = WL ? NS * CH_R * STASH_ROW : 0 };
    static constexpr int taps_in_lds() { const long room = (160l * 1024 - (long)(TILE_F2 + STASH_F2) * 8) / (4l * K); return room >= P ? P : (int)room; }
    enum { TL = taps_in_lds() };
    static_assert(!WL || TL == P, "whole-line stores need every tap in LDS beside the tile and the stash");
    static constexpr size_t lds_bytes() { return (size_t)(TILE_F2 + STASH_F2) * sizeof(float2) + (size_t)TL * K * sizeof(float); }
};

// EDGE = false: every block the workgroup touches (history, the slabs, the prefetch past the last round) lies inside
// the stream, so loads need no clamping, the oscillator no zeroing and the stores no guard -- all but the first and the
// last workgroup of a launch.  Addresses that do not change from round to round (the butterflies' LDS positions, the
// granule stores' LDS sources and HBM destinations) are computed once per launch, not once per use.
// P = taps per column: a.taps is the column tap table tap[j][n], j = 0 the newest block's tap, P * K floats.
// SHIFT: column n's FIR output goes to tile column (n + a.col_shift) mod K (the oversampled front end's rotation).
template <int K, int C, int T, int P, bool SHIFT, bool EDGE>
__device__ __forceinline__ void channelizer_rounds(const ChanArgs &a, float2 *tile)
{
    constexpr int TPS = K / C;              // threads per slab
    constexpr int NS = T / TPS;             // slabs per workgroup
    constexpr int N = K / 2;
    constexpr int H = P - 1;                // history blocks
    constexpr int TL = Geo<K, C, T, P>::TL, TG = P - TL;     // taps per column in LDS / fetched per round
    constexpr int S = Plan<K>::S, F = Plan<K>::F, ROWP = Plan<K>::ROWP, R = Plan<K>::R, LR = Plan<K>::LR;
    static_assert(TPS * C == K && NS * TPS == T && NS >= 1, "bad channelizer geometry");
    static_assert(TL >= 1 && TG >= 0 && TG <= 8, "tap split");

    const int tid = threadIdx.x;
    const int sl = tid / TPS, cg = tid % TPS;
    const int n0 = cg * C;
    const long long slab = (long long)blockIdx.x * NS + sl;
    const long long bs = slab * (long long)a.slab_blocks;        // first block of my slab

    // taps: tap[j][c] = column n0+c's tap on the block j back.  They live in LDS (TL*K floats behind the tile) and are
    // fetched into registers for the FIR of each round only: across the FFT stages the registers
    // hold the sliding window plus the next round's blocks that are already in flight.
    constexpr bool WL = Geo<K, C, T, P>::WL && !SHIFT;
    constexpr int SROW = Geo<K, C, T, P>::STASH_ROW;
    float2 *stash = tile + NS * CH_R * ROWP;                   // (WL) [CH_R rows][SROW]: the even round's kept bins, row = time sample, column = channel
    float *ltap = reinterpret_cast<float *>(tile + NS * CH_R * ROWP + Geo<K, C, T, P>::STASH_F2);
    {
        constexpr int NT = TL * K;                  // all requests first, then the LDS writes: one round trip
        constexpr int PER = (NT + T - 1) / T;
        float tv[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) { const int idx = tid + i * T; tv[i] = a.taps[idx < NT ? idx : 0]; }
#pragma unroll
        for (int i = 0; i < PER; i++) { const int idx = tid + i * T; if (idx < NT) ltap[idx] = tv[i]; }
    }
    __syncthreads();
    // radix-R stage twiddles W_L^{r*pos}, r = 1..R-1: pos = q % (L/R) does not depend on the
    // loop trip because L/R divides the workgroup size
    float2 tw[S > 0 ? S : 1][R - 1];
#pragma unroll
    for (int st = 0; st < S; st++) {
        const int L = K >> (LR * st), q4 = L >> LR;
        const int pos = tid % q4;
#pragma unroll
        for (int r = 1; r < R; r++) {
            float sn, cs; sincos_u32((uint32_t)(r * pos) * (uint32_t)(4294967296.0 / L), sn, cs);
            tw[st][r - 1] = make_float2(cs, -sn);
        }
    }

    const uint32_t dth = a.dtheta;
    float sd1, cd1; sincos_u32(dth, sd1, cd1);                  // e^{j dtheta}: column n+1 from column n
    float sk8, ck8; sincos_u32((uint32_t)K * dth, sk8, ck8);    // e^{j K dtheta}: block b+1 from block b
    const uint32_t t0 = a.first_sample_lo;

    // raw samples of block b (relative to a.x), columns n0..n0+C-1; zeros outside the stream
    // Branch free on purpose: a load inside a divergent `if` gets an s_waitcnt vmcnt(0) at the
    // join, which would serialise the round's loads into one HBM round trip each.
    auto load_raw = [&](long long b, float2 (&dst)[C]) {
        const float2 *src = a.x + n0;                                   // always mapped
        if constexpr (EDGE) {
            const bool inx = b >= 0 && b < (long long)a.nblocks;
            const bool inh = b < 0 && a.halo != nullptr;
            if (inx) src = a.x + (size_t)b * K + n0;
            if (inh) src = a.halo + (size_t)(b + H) * K + n0;
        } else src = a.x + (size_t)b * K + n0;
        // the value is not touched here (that would wait for it): the mixer zeroes blocks outside the stream
        if constexpr (C == 2) {
#if CH_NT_LOAD
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(src));
#else
            const float4 v = *reinterpret_cast<const float4 *>(src);
#endif
            dst[0] = make_float2(v.x, v.y); dst[1] = make_float2(v.z, v.w);
        } else dst[0] = src[0];
    };
    // NCO in place; blocks outside the stream (before a cold start, past the end) become zeros
    // Oscillator: one transcendental pair per thread and group of CH_R blocks.  Block g (a multiple of CH_R) gets
    // sin/cos of the exact phase of its first column; the following blocks of the group are that value turned by
    // the launch-constant per-block step K*dtheta, the neighbouring column by the per-sample step.  Every value is
    // a fixed function of (group start, position in the group, column), so a stream cut into several calls on
    // tile boundaries reproduces the single-call result bit for bit (explicit fma shapes, no reassociation).
    auto osc_start = [&](long long g, float &sn, float &cs) {
        sincos_u32_hw((t0 + (uint32_t)(g * K + n0)) * dth, sn, cs);
    };
    auto osc_next_block = [&](float &sn, float &cs) {
        const float s2 = fmaf(sn, ck8, cs * sk8), c2 = fmaf(cs, ck8, -(sn * sk8)); sn = s2; cs = c2;
    };
    auto mix_with = [&](long long b, float sn, float cs, float2 (&dst)[C]) {
        // blocks outside the stream become zeros: a zeroed oscillator (by value; the caller's copy keeps turning)
        // zeroes both columns, two selects per block instead of four.  (Clamped loads return finite samples.)
        if constexpr (EDGE) {
            const bool valid = (b >= 0 && b < (long long)a.nblocks) || (b < 0 && a.halo != nullptr);
            sn = valid ? sn : 0.f; cs = valid ? cs : 0.f;
        }
#pragma unroll
        for (int c = 0; c < C; c++) {
            if (c > 0) { const float s2 = fmaf(sn, cd1, cs * sd1), c2 = fmaf(cs, cd1, -(sn * sd1)); sn = s2; cs = c2; }
            dst[c] = make_float2(fmaf(dst[c].x, cs, dst[c].y * sn), fmaf(dst[c].y, cs, -(dst[c].x * sn)));
        }
    };

    // s[0..H-1] history (oldest first), s[H..H+7] the round's new blocks.  The new blocks of
    // round r+1 are requested into s[H..H+7] right after round r's FIR has consumed them, so
    // the HBM latency hides under the FFT stages; they are mixed in place when the round starts.
    float2 s[H + CH_R][C];
    // (slabs past the end of the stream run on clamped addresses and zeros; their stores are masked.
    //  Keeping this straight-line matters: a load under a branch is waited for at the join.)
#pragma unroll
    for (int i = 0; i < H + CH_R; i++) load_raw(bs - H + i, s[i]);
    {   // the history blocks bs-H .. bs-1 sit in the groups of 8 starting at bs - 8 HG, ..., bs - 8 (the first one from position HSKIP on)
        constexpr int HG = (H + CH_R - 1) / CH_R, HSKIP = HG * CH_R - H;
        static_assert(CH_R == 8, "groups of 8");
        int i = 0;
#pragma unroll
        for (int g = 0; g < HG; g++) {
            float sn, cs;
            osc_start(bs - (long long)CH_R * (HG - g), sn, cs);
#pragma unroll
            for (int k = 0; k < CH_R; k++) {
                if (g > 0 || k >= HSKIP) { mix_with(bs - H + i, sn, cs, s[i]); i++; }
                osc_next_block(sn, cs);
            }
        }
    }

    // ---- round-invariant addresses
    constexpr int NBF4 = NS * CH_R * (K / R);           // radix-R butterflies per stage
    constexpr int NG = NS * CH_R * (K / F);             // F-point groups
    constexpr int NOPS = NS * N * (CH_R / 2);           // 16-byte stores per round
    static_assert(S == 0 || (NBF4 % T == 0 && T % (K / R) == 0), "a thread's butterflies must differ by whole rows");
    static_assert(NG % T == 0 || NG < T, "F-point groups per thread");
    static_assert(NOPS % T == 0, "granule stores per thread");
    constexpr int BTRIPS = S > 0 ? NBF4 / T : 0, BSTEP = S > 0 ? (T / (K / R)) * ROWP : 0;
    int fa[S > 0 ? S : 1];                              // padded LDS index of a butterfly's first element, trip 0
    if constexpr (S > 0) {
#pragma unroll
        for (int st = 0; st < S; st++) {
            const int L = K >> (LR * st), q4 = L >> LR;
            const int f = tid / (K / R), j = tid % (K / R);
            fa[st] = f * ROWP + pad<K>((j / q4) * L + j % q4);
        }
    }
    constexpr int GTRIPS = (NG + T - 1) / T;
    int fg[GTRIPS];
#pragma unroll
    for (int i = 0; i < GTRIPS; i++) { const int g = tid + i * T; fg[i] = (g / (K / F)) * ROWP + (g % (K / F)) * (F + 1); }
    constexpr int OTRIPS = NOPS / T;
    int ssrc[OTRIPS];                                   // LDS source of granule store k
    uint32_t sdst[OTRIPS];                              // its destination in round 0 (16-byte units from a.out); odd rounds fill the granule's second half, every second round is one tile further
    int srem[OTRIPS];                                   // blocks of the stream from its slab's first one on (EDGE: stores past the end are masked)
#pragma unroll
    for (int k = 0; k < OTRIPS; k++) {
        const int o = tid + k * T;
        const int osl = o / (N * (CH_R / 2)), rem = o % (N * (CH_R / 2));
        const int ch = rem / (CH_R / 2), rp = rem % (CH_R / 2);
        const long long ob = ((long long)blockIdx.x * NS + osl) * (long long)a.slab_blocks;
        const int g = ch / a.cg, c = ch % a.cg;
        ssrc[k] = (osl * CH_R + 2 * rp) * ROWP + pad<K>(dif_pos<K>(ch));
        sdst[k] = (uint32_t)((((size_t)g * a.ntiles + (size_t)(ob / MCRX_TILE_S)) * a.cg + c) * (MCRX_TILE_S / 2) + rp);   // (launch_one checks the range; slabs start on tiles)
        const long long left = (long long)a.nblocks - ob;
        srem[k] = left < 0 ? 0 : (left > 0x40000000ll ? 0x40000000 : (int)left);
    }
    const uint32_t tile_step = a.cg * (MCRX_TILE_S / 2);     // 16-byte units between consecutive tiles of a channel group
    float4 *out4 = reinterpret_cast<float4 *>(a.out);
    // WL: store k of an odd round is 16-byte unit u = tid % 8 of channel tid / 8 + 64 k's granule: units 0-3 (samples 0-7) from the stash, 4-7 from
    // the tile.  K = 1024 = 8 x 8 x 16: 64 channels on, a bin sits one tile position further (dif_pos), so both sources are linear in k.
    constexpr int WTRIPS = WL ? N * CH_R / T : 1;
    int wsrc0 = 0, wsk = 0, wrow = 0, sg[GTRIPS];
    uint32_t wdst[WTRIPS];
    if constexpr (WL) {
        static_assert(!WL || (NS == 1 && T % 8 == 0 && S == 2 && R == 8 && F == 16), "the whole-line store path is written for K = 1024");
        const int u = tid & 7, ch0 = tid >> 3;
        if (u < 4) { wsrc0 = (int)(stash - tile) + 2 * u * SROW + ch0; wsk = T / 8; wrow = SROW; }
        else { wsrc0 = 2 * (u - 4) * ROWP + pad<K>(dif_pos<K>(ch0)); wsk = 1; wrow = ROWP; }
        const long long ob = (long long)blockIdx.x * (long long)a.slab_blocks;
#pragma unroll
        for (int k = 0; k < WTRIPS; k++) {
            const int ch = ch0 + k * (T / 8), g = ch / a.cg, c = ch % a.cg;
            wdst[k] = (uint32_t)((((size_t)g * a.ntiles + (size_t)(ob / MCRX_TILE_S)) * a.cg + c) * (MCRX_TILE_S / 2) + u);
        }
        // the register stage of an even round leaves bin (group gi, position q < 8) = channel chg + 64 q of time sample f at stash[f][chg + 64 q]
#pragma unroll
        for (int i = 0; i < GTRIPS; i++) { const int g = tid + i * T, f = g / (K / F), gi = g % (K / F); sg[i] = (int)(stash - tile) + f * SROW + (gi / R) + R * (gi % R); }
    }
    int wcol[C];                                        // where my columns' FIR outputs go in a tile row (padded index)
#pragma unroll
    for (int c = 0; c < C; c++) wcol[c] = pad<K>(SHIFT ? (int)((n0 + c + a.col_shift) & (K - 1)) : n0 + c);

    const int rounds = a.slab_blocks / CH_R;
    for (int rd = 0; rd < rounds; rd++) {
        const long long b0 = bs + (long long)rd * CH_R;
        float tg[TG > 0 ? TG : 1][C];                   // the column's oldest taps, from the table (L2): requested before the mixer, used first
#pragma unroll
        for (int j = 0; j < TG; j++)
#pragma unroll
            for (int c = 0; c < C; c++) tg[j][c] = a.taps[(size_t)(TL + j) * K + n0 + c];
        if constexpr (P <= CH_P_REF) {
            float tap[P][C];
#pragma unroll
            for (int j = 0; j < P; j++)
#pragma unroll
                for (int c = 0; c < C; c++) tap[j][c] = ltap[j * K + n0 + c];
            {
                float sn, cs;
                osc_start(b0, sn, cs);
#pragma unroll
                for (int r = 0; r < CH_R; r++) { mix_with(b0 + r, sn, cs, s[H + r]); osc_next_block(sn, cs); }
            }
#pragma unroll
            for (int r = 0; r < CH_R; r++) {
                float2 v[C];
#pragma unroll
                for (int c = 0; c < C; c++) v[c] = make_float2(0.f, 0.f);
#pragma unroll
                for (int j = P - 1; j >= 0; j--) {              // oldest tap first, like a window dot product
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        v[c].x += tap[j][c] * s[H + r - j][c].x;
                        v[c].y += tap[j][c] * s[H + r - j][c].y;
                    }
                }
                float2 *row = tile + (sl * CH_R + r) * ROWP;
                if constexpr (SHIFT) {
#pragma unroll
                    for (int c = 0; c < C; c++) row[wcol[c]] = v[c];
                } else {
                    float2 *rw = row + pad<K>(n0);              // n0, n0+1 share a pad group
#pragma unroll
                    for (int c = 0; c < C; c++) rw[c] = v[c];
                }
            }
        } else {
            // many taps: the eight outputs of the round accumulate side by side, one tap (from LDS, or of the fetched ones) at a time,
            // oldest first -- the taps never sit in registers all at once
            {
                float sn, cs;
                osc_start(b0, sn, cs);
#pragma unroll
                for (int r = 0; r < CH_R; r++) { mix_with(b0 + r, sn, cs, s[H + r]); osc_next_block(sn, cs); }
            }
            // (CH_RH outputs per pass: the accumulators of all eight at once cost 16 more registers than the kernel has)
#pragma unroll
            for (int rh = 0; rh < CH_R; rh += CH_RH) {
                float2 v[CH_RH][C];
#pragma unroll
                for (int r = 0; r < CH_RH; r++)
#pragma unroll
                    for (int c = 0; c < C; c++) v[r][c] = make_float2(0.f, 0.f);
#pragma unroll
                for (int j = P - 1; j >= 0; j--) {
                    float tj[C];
#pragma unroll
                    for (int c = 0; c < C; c++) tj[c] = j >= TL ? tg[j - TL][c] : ltap[j * K + n0 + c];
#pragma unroll
                    for (int r = 0; r < CH_RH; r++)
#pragma unroll
                        for (int c = 0; c < C; c++) {
                            v[r][c].x += tj[c] * s[H + rh + r - j][c].x;
                            v[r][c].y += tj[c] * s[H + rh + r - j][c].y;
                        }
                }
#pragma unroll
                for (int r = 0; r < CH_RH; r++) {
                    float2 *row = tile + (sl * CH_R + rh + r) * ROWP;
                    if constexpr (SHIFT) {
#pragma unroll
                        for (int c = 0; c < C; c++) row[wcol[c]] = v[r][c];
                    } else {
                        float2 *rw = row + pad<K>(n0);
#pragma unroll
                        for (int c = 0; c < C; c++) rw[c] = v[r][c];
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < H; i++)
#pragma unroll
            for (int c = 0; c < C; c++) s[i][c] = s[i + CH_R][c];
        // next round's blocks (past the slab's last round: clamped, never used)
#pragma unroll
        for (int r = 0; r < CH_R; r++) load_raw(b0 + CH_R + r, s[H + r]);
        lds_barrier();

        // ---- NS*CH_R independent K-point FFTs, in place
        if constexpr (S > 0) {
#pragma unroll
            for (int st = 0; st < S; st++) {
                const int L = K >> (LR * st), q4 = L >> LR;
                const int D = q4 + q4 / F;              // padded distance of the butterfly's legs (q4 is a multiple of F)
#pragma unroll
                for (int i = 0; i < BTRIPS; i++) {
                    float2 *p = tile + fa[st] + i * BSTEP;
                    if constexpr (R == 8) {
                        float2 v[8];
#pragma unroll
                        for (int m = 0; m < 8; m++) v[m] = p[m * D];
                        fft_reg<8>(v);                  // v[m] = X[bitrev(m)]
                        p[0] = v[0];
#pragma unroll
                        for (int r = 1; r < 8; r++) p[r * D] = cmul_fx(v[bitrev_c(r, 3)], tw[st][r - 1]);
                        continue;
                    }
                    const float2 x0 = p[0], x1 = p[D], x2 = p[2 * D], x3 = p[3 * D];
                    const float2 a0 = cadd(x0, x2), a1 = csub(x0, x2), a2 = cadd(x1, x3), a3 = cmulnj(csub(x1, x3));
                    p[0] = cadd(a0, a2);
                    p[D] = cmul_fx(cadd(a1, a3), tw[st][0]);
                    p[2 * D] = cmul_fx(csub(a0, a2), tw[st][1]);
                    p[3 * D] = cmul_fx(csub(a1, a3), tw[st][2]);
                }
                lds_barrier();
            }
        }
        {
#pragma unroll
            for (int i = 0; i < GTRIPS; i++) {
                if (NG % T != 0 && tid + i * T >= NG) break;
                float2 *p = tile + fg[i];                    // == pad(gi * F) in row f
                float2 v[F];
#pragma unroll
                for (int m = 0; m < F; m++) v[m] = p[m];
                fft_reg<F, (F >= 2)>(v);                    // bins >= N (the upper half of every group) are never stored
                if (WL && (rd & 1) == 0) {
                    float2 *ps = tile + sg[i];
#pragma unroll
                    for (int m = 0; m < F; m++) if (bitrev_c(m, Log2<F>::v) < F / 2) ps[bitrev_c(m, Log2<F>::v) * (R * R)] = v[m];
                } else {
#pragma unroll
                for (int m = 0; m < F; m++) if (F < 2 || bitrev_c(m, Log2<F>::v) < F / 2) p[bitrev_c(m, Log2<F>::v)] = v[m];
                }
            }
            lds_barrier();
        }
        if constexpr (WL) {
            if ((rd & 1) == 0) continue;                    // (the tile is free again: every thread has read its groups; the stash is read a round later)
            bool ok = true;
            if constexpr (EDGE) ok = (rd - 1) * CH_R < srem[0];
            if (ok) {
#pragma unroll
                for (int k = 0; k < WTRIPS; k++) {
                    const float2 *src = tile + wsrc0 + k * wsk;
                    const float2 v0 = src[0], v1 = src[wrow];
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    v4f nv = { v0.x, v0.y, v1.x, v1.y };
#if CH_WL_PLAIN
                    *reinterpret_cast<v4f *>(&out4[(size_t)(wdst[k] + ((uint32_t)rd >> 1) * tile_step)]) = nv;
#else
                    __builtin_nontemporal_store(nv, reinterpret_cast<v4f *>(&out4[(size_t)(wdst[k] + ((uint32_t)rd >> 1) * tile_step)]));
#endif
                }
            }
            lds_barrier();
            continue;
        }

        // ---- store bins 0..N-1: this round's 8 time samples = one half (64 B) of every channel's granule
#pragma unroll
        for (int k = 0; k < OTRIPS; k++) {
            bool ok = true;
            if constexpr (EDGE) ok = rd * CH_R < srem[k];
            if (ok) {
                const float2 *src = tile + ssrc[k];
                const float2 v0 = src[0], v1 = src[ROWP];
#if CH_NT_STORE
                { typedef float v4f __attribute__((ext_vector_type(4)));
                  v4f nv = { v0.x, v0.y, v1.x, v1.y };
                  __builtin_nontemporal_store(nv, reinterpret_cast<v4f *>(&out4[(size_t)(sdst[k] + ((uint32_t)rd >> 1) * tile_step + ((uint32_t)rd & 1u) * (CH_R / 2))])); }
#else
                out4[(size_t)(sdst[k] + ((uint32_t)rd >> 1) * tile_step + ((uint32_t)rd & 1u) * (CH_R / 2))] = make_float4(v0.x, v0.y, v1.x, v1.y);
#endif
            }
        }
        lds_barrier();
    }
}

template <int K, int C, int T, int P, bool SHIFT>
__global__ __launch_bounds__(T) void channelizer_kernel(ChanArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float2 tile[];     // [NS][CH_R][ROWP], then the taps in LDS
    constexpr int NS = T / (K / C), H = P - 1;
    const long long s0 = (long long)blockIdx.x * NS;
    const long long first = s0 * (long long)a.slab_blocks - H, last = (s0 + NS) * (long long)a.slab_blocks + CH_R;
    if (first >= 0 && last <= (long long)a.nblocks) channelizer_rounds<K, C, T, P, SHIFT, false>(a, tile);
    else channelizer_rounds<K, C, T, P, SHIFT, true>(a, tile);
}

template <int K, int C, int T, int P, bool SHIFT>
static hipError_t launch_one(const ChanArgs &a, hipStream_t st)
{
    constexpr int NS = T / (K / C);
    const size_t lds = Geo<K, C, T, P>::lds_bytes();
    long long nslabs = ((long long)a.nblocks + a.slab_blocks - 1) / a.slab_blocks;
    unsigned grid = (unsigned)((nslabs + NS - 1) / NS);
    if (grid == 0) return hipSuccess;
    // granule stores are addressed by 32-bit offsets in 16-byte units: 64 GB of output per launch
    if ((unsigned long long)(K / 2) * ((unsigned long long)a.ntiles + (unsigned long long)NS * a.slab_blocks / MCRX_TILE_S + 1ull) * (MCRX_TILE_S / 2) >= (1ull << 32))
        return hipErrorInvalidValue;
    static PerDeviceOnce attr_done;          // (per instantiation; devscope.hpp)
    hipError_t e = raise_lds_limit((const void *)channelizer_kernel<K, C, T, P, SHIFT>, lds, attr_done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((channelizer_kernel<K, C, T, P, SHIFT>), dim3(grid), dim3(T), lds, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Any other channel count the reference constructor accepts (lib/multichannelrx.cc:54-66 only asks for N >= 1;
// liquid's firpfbch takes any K): the same arithmetic without the register window and the radix-4 plan -- a
// workgroup per round of 8 blocks (half a tile), FIR columns straight from global memory (the 14-fold reuse is the caches'), then a
// direct DFT of the kept bins with an exact integer twiddle index.  A fallback for odd sizes, O(K N) per block:
// the power-of-two kernel above is the product's fast path.
#define CG_T 256
#define CH_P CH_P_REF
#define CH_H (CH_P_REF - 1)
__global__ __launch_bounds__(CG_T) void channelizer_generic_kernel(ChanArgs a, uint32_t K)
{
    extern __shared__ __attribute__((aligned(16))) float2 gl[];        // V[CH_R][K], then W[K]
    float2 *V = gl, *W = gl + (size_t)CH_R * K;
    const uint32_t N = K / 2;
    const int tid = threadIdx.x;
    const long long b0 = (long long)blockIdx.x * CH_R;
    const uint32_t dth = a.dtheta, t0 = a.first_sample_lo;
    for (uint32_t i = tid; i < K; i += CG_T) {
        float sn, cs; sincos_u32((uint32_t)((((uint64_t)i) << 32) / K), sn, cs);
        W[i] = make_float2(cs, -sn);                                    // exp(-j 2 pi i / K)
    }
    for (uint32_t n = tid; n < K; n += CG_T) {
        float2 acc[CH_R];
#pragma unroll
        for (int r = 0; r < CH_R; r++) acc[r] = make_float2(0.f, 0.f);
        // mixed samples of column n, blocks b0-13 .. b0+7, oldest first; each feeds up to 8 rows
        for (int i = 0; i < CH_H + CH_R; i++) {
            const long long b = b0 - CH_H + i;
            float2 x = make_float2(0.f, 0.f);
            bool valid = false;
            if (b >= 0 && b < (long long)a.nblocks) { x = a.x[(size_t)b * K + n]; valid = true; }
            else if (b < 0 && a.halo != nullptr && b + CH_H >= 0) { x = a.halo[(size_t)(b + CH_H) * K + n]; valid = true; }
            float2 u = make_float2(0.f, 0.f);
            if (valid) u = mix_down_hw(x, (t0 + (uint32_t)(b * (long long)K + n)) * dth);
#pragma unroll
            for (int r = 0; r < CH_R; r++) {
                const int j = CH_H + r - i;                             // tap branch this block is for row r
                if (j >= 0 && j < CH_P) {
                    const float h = a.taps[(uint32_t)j * K + n];                // column tap table
                    acc[r].x += h * u.x; acc[r].y += h * u.y;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < CH_R; r++) V[(size_t)r * K + n] = acc[r];
    }
    __syncthreads();
    const long long tl = b0 / MCRX_TILE_S; const int half = (int)((b0 / CH_R) & 1);
    for (uint32_t k = tid; k < N; k += CG_T) {
        float2 y[CH_R];
#pragma unroll
        for (int r = 0; r < CH_R; r++) y[r] = make_float2(0.f, 0.f);
        uint32_t idx = 0;
        for (uint32_t n = 0; n < K; n++) {
            const float2 w = W[idx];
#pragma unroll
            for (int r = 0; r < CH_R; r++) {
                const float2 v = V[(size_t)r * K + n];
                y[r].x += v.x * w.x - v.y * w.y; y[r].y += v.x * w.y + v.y * w.x;
            }
            idx += k; if (idx >= K) idx -= K;
        }
        const uint32_t g = k / a.cg, c = k % a.cg;
        float4 *dst = reinterpret_cast<float4 *>(a.out + (((size_t)g * a.ntiles + (size_t)tl) * a.cg + c) * MCRX_TILE_S + half * CH_R);
#pragma unroll
        for (int r = 0; r < CH_R; r += 2) dst[r / 2] = make_float4(y[r].x, y[r].y, y[r + 1].x, y[r + 1].y);
    }
}

static bool pow2_fast(unsigned K) { return K >= 2 && K <= 1024 && (K & (K - 1)) == 0; }
int channelizer_supported(unsigned K)
{ return (pow2_fast(K) || (K >= 2 && K % 2 == 0 && K <= 2048)) ? 1 : 0; }

// Slab sizing.  A slab costs a 13-block halo re-read, and a grid that is not a whole number of
// waves over the CUs idles part of the chip in its last wave (K = 1024: one 512-thread workgroup
// per CU).  Pick the largest slab <= 512 blocks for which the grid is k * capacity workgroups.
// The result does not depend on the slab size (each output block sums the same terms in the same order).
uint32_t channelizer_auto_slab(unsigned K, size_t nblocks, unsigned ncu)
{
    const unsigned threads = K >= 1024 ? 512u : 256u, C = K >= 4 ? 2u : 1u;
    const unsigned ns = threads / (K / C > 0 ? K / C : 1u) ? threads / (K / C) : 1u;   // slabs per workgroup
    const size_t capacity = (size_t)ncu * (K >= 1024 ? 1u : 2u) * ns;                   // slabs in one wave of workgroups
    const size_t k = (nblocks + capacity * 512 - 1) / (capacity * 512);
    size_t slab = (nblocks + capacity * k - 1) / (capacity * k);
    slab = (slab + MCRX_TILE_S - 1) & ~(size_t)(MCRX_TILE_S - 1);
    if (slab < 32) slab = 32;
    return (uint32_t)slab;
}

hipError_t channelizer_launch(unsigned K, unsigned P, const ChanArgs &a, hipStream_t st)
{
    if (P == CH_P_OVS) {                    // the composite bank of the oversampled front end: power-of-two channel counts only
        switch (K) {
        case 2:    return launch_one<2, 1, 256, CH_P_OVS, true>(a, st);
        case 4:    return launch_one<4, 2, 256, CH_P_OVS, true>(a, st);
        case 8:    return launch_one<8, 2, 256, CH_P_OVS, true>(a, st);
        case 16:   return launch_one<16, 2, 256, CH_P_OVS, true>(a, st);
        case 32:   return launch_one<32, 2, 256, CH_P_OVS, true>(a, st);
        case 64:   return launch_one<64, 2, 256, CH_P_OVS, true>(a, st);
        case 128:  return launch_one<128, 2, 256, CH_P_OVS, true>(a, st);
        case 256:  return launch_one<256, 2, 256, CH_P_OVS, true>(a, st);
        case 512:  return launch_one<512, 2, 512, CH_P_OVS, true>(a, st);     // (two slabs per workgroup: with 256 threads the taps + tile would leave one wave per SIMD)
        case 1024: return launch_one<1024, 2, 512, CH_P_OVS, true>(a, st);
        default:   return hipErrorInvalidValue;
        }
    }
    if (P != CH_P_REF) return hipErrorInvalidValue;
    if (!pow2_fast(K)) {
        if (a.nblocks == 0) return hipSuccess;
        const size_t lds = (size_t)(CH_R + 1) * K * sizeof(float2);
        static PerDeviceOnce gen_done;
        hipError_t e = raise_lds_limit((const void *)channelizer_generic_kernel, 160 * 1024, gen_done);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(channelizer_generic_kernel, dim3(a.nblocks / CH_R), dim3(CG_T), lds, st, a, (uint32_t)K);
        return hipGetLastError();
    }
    switch (K) {
    case 2:    return launch_one<2, 1, 256, CH_P_REF, false>(a, st);
    case 4:    return launch_one<4, 2, 256, CH_P_REF, false>(a, st);
    case 8:    return launch_one<8, 2, 256, CH_P_REF, false>(a, st);
    case 16:   return launch_one<16, 2, 256, CH_P_REF, false>(a, st);
    case 32:   return launch_one<32, 2, 256, CH_P_REF, false>(a, st);
    case 64:   return launch_one<64, 2, 256, CH_P_REF, false>(a, st);
    case 128:  return launch_one<128, 2, 256, CH_P_REF, false>(a, st);
    case 256:  return launch_one<256, 2, 256, CH_P_REF, false>(a, st);
    case 512:  return launch_one<512, 2, 256, CH_P_REF, false>(a, st);
    case 1024: { static const int c1 = devel_env("MCRX_CHAN_C1") ? atoi(devel_env("MCRX_CHAN_C1")) : 0;
                 return c1 ? launch_one<1024, 1, 1024, CH_P_REF, false>(a, st) : launch_one<1024, 2, 512, CH_P_REF, false>(a, st); }
    default:   return hipErrorInvalidValue;
    }
}

}  // namespace mcrx
