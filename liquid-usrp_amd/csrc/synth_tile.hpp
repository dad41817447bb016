// synth_tile.hpp -- fused synthesis bank of the GPU multichanneltx (included by txgen.hip behind TxSynthArgs / frame_sample).
//
// Replaces, per block of K = 2N wideband samples, multichanneltx::GenerateSamples (lib/multichanneltx.cc:192-227):
//   firpfbch_crcf_synthesizer_execute (K channels, m = 13: inverse FFT, polyphase FIR of p = 26 taps per branch)
//   nco_crcf_mix_up / nco_crcf_step
// in ONE kernel -- the mirror image of channelizer.hip:
//   X_b[k]  = channel k's sample of block b (k < N; bins N..K-1 are zero)        4 B read per wideband sample
//   v_b     = K-point inverse FFT of X_b, unnormalised                           LDS tile, radix-4 + 16-point register stage
//   y_b[n]  = sum_{j<26} h[n + jK] v_{b-j}[n]                                    register window down the time axis
//   out     = gain * y_b[n] * exp(+j t dtheta), t = absolute sample index        8 B written per wideband sample
// Round 2 ran this as two kernels with the inverse-FFT outputs in HBM between them (one workgroup per block, radix-2 Stockham
// with a sincos per butterfly; 28 B per sample, 7.5 + 5.0 GB of measured traffic per 207 M samples): 2.68 ms.  Here a workgroup owns
// a time slab; a thread owns two adjacent columns and keeps their last 25 inverse-FFT outputs in registers (the analysis bank's
// plan -- its window is 13 deep, this one 25: rounds of 4 blocks instead of 8 keep it at 116 registers), the taps live in LDS as
// the symmetric half of the prototype, and a slab starts 28 blocks early to fill its window (7 % recomputed at 400-block slabs).
// The inverse transform is the forward one on conjugated data (conj on the way into the tile and on the way out), so the
// butterflies are channelizer.hip's.
#pragma once

namespace mcrx {
#ifndef SYN_NT_STORE
#define SYN_NT_STORE 0      /* experiment: the wideband output as non-temporal stores */
#endif
namespace syn {

// development builds only (scratch/syn_ablate.sh: -DSYN_ABLATE=bits): 1 no radix-4 stages, 2 no register stage, 4 one tap instead
// of 26, 8 no stores, 16 no input loads, 32 no window shift, 64 no early wait for the inputs.  The shipped library is built without it.
#ifndef SYN_ABLATE
#define SYN_ABLATE 0
#endif
#define SYN_P 26            // taps per branch (m = 13)
#define SYN_H (SYN_P - 1)   // blocks of history

typedef float v2f __attribute__((ext_vector_type(2)));
// acc += x * t.lo / t.hi in both components (one v_pk_fma_f32, the tap broadcast by op_sel)
__device__ __forceinline__ void pk_fma_lo(float2 &acc, const float2 &x, v2f t)
{
    v2f a = { acc.x, acc.y }; const v2f xv = { x.x, x.y };
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a) : "v"(xv), "v"(t));
    acc.x = a.x; acc.y = a.y;
}
__device__ __forceinline__ void pk_fma_hi(float2 &acc, const float2 &x, v2f t)
{
    v2f a = { acc.x, acc.y }; const v2f xv = { x.x, x.y };
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(a) : "v"(xv), "v"(t));
    acc.x = a.x; acc.y = a.y;
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T> __device__ __forceinline__ T *uniform_ptr(T *p)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    asm volatile("" : "+s"(lo), "+s"(hi));
    typedef __attribute__((address_space(1))) T GT;
    return (T *)reinterpret_cast<GT *>(((unsigned long long)hi << 32) | lo);
}

// (K = 1024 and 512 cross LDS in radix-8 stages -- a stage fewer than in radix 4 --, like channelizer.hip since round 4)
#ifndef SYN_RADIX8
#define SYN_RADIX8 1
#endif
template <int K> struct Plan {
    enum { RR = (SYN_RADIX8 && (K == 1024 || K == 512)) ? 8 : 4, LR = RR == 8 ? 3 : 2 };
    static constexpr int stages() { int L = K, s = 0; while (L > 16) { L /= RR; s++; } return s; }
    static constexpr int final_size() { int L = K; while (L > 16) L /= RR; return L; }
    enum { S = stages(), F = final_size(), RL = K + K / F, ROWP = RL + 1 };
    static constexpr int tw_off(int st) { int o = 0; for (int i = 0; i < st; i++) o += (RR - 1) * ((K >> (LR * i)) >> LR); return o; }
    enum { TW = tw_off(stages()) };
};
template <int K> __device__ __forceinline__ int pad(int e) { return e + e / Plan<K>::F; }
template <int K> __device__ __forceinline__ int dif_pos(int k)
{
    int L = K, pos = 0;
#pragma unroll
    for (int s = 0; s < Plan<K>::S; s++) { pos += (k & (Plan<K>::RR - 1)) * (L >> Plan<K>::LR); k >>= Plan<K>::LR; L >>= Plan<K>::LR; }
    return pos + k;
}
__device__ __forceinline__ float2 w16(int k)
{
    const float c[8] = { 1.0f, 0.92387953251f, 0.70710678119f, 0.38268343236f, 0.0f, -0.38268343236f, -0.70710678119f, -0.92387953251f };
    const float s[8] = { 0.0f, -0.38268343236f, -0.70710678119f, -0.92387953251f, -1.0f, -0.92387953251f, -0.70710678119f, -0.38268343236f };
    return make_float2(c[k], s[k]);
}
constexpr int bitrev_c(int i, int bits) { int r = 0; for (int b = 0; b < bits; b++) if (i & (1 << b)) r |= 1 << (bits - 1 - b); return r; }
template <int K> struct Log2 { enum { v = 1 + Log2<K / 2>::v }; };
template <> struct Log2<1> { enum { v = 0 }; };
// F-point forward DFT in registers, natural order in, bit-reversed out
template <int F>
__device__ __forceinline__ void fft_reg(float2 (&v)[F])
{
#pragma unroll
    for (int h = F / 2; h >= 1; h >>= 1) {
#pragma unroll
        for (int i = 0; i < F; i++) {
            if ((i & h) == 0) {
                const float2 u = v[i], w = v[i + h];
                v[i] = cadd(u, w);
                const float2 d = csub(u, w);
                const int tk = (i & (h - 1)) * (8 / h);
                if (tk == 0) v[i + h] = d;
                else if (tk == 4) v[i + h] = cmulnj(d);
                else v[i + h] = cmul(d, w16(tk));
            }
        }
    }
}

template <int K, int R> struct Lds {
    static constexpr int TAPF = (SYN_P * K / 2 + 1 + 1) & ~1;       // floats: h[0 .. pK/2], the symmetric half
    static constexpr size_t bytes() { return (size_t)(R * Plan<K>::ROWP) * sizeof(float2) + (size_t)TAPF * sizeof(float) + (size_t)(Plan<K>::TW > 0 ? Plan<K>::TW : 1) * sizeof(float2); }
};

// K >= 128 (T = K/2 >= 64 threads: thread t is channel t when the inputs are gathered, columns 2t, 2t+1 afterwards)
// IN: where the bank's inputs come from -- SYN_TILES exchanged granules, SYN_SYMS the aligned symbol loader (syn_aligned below
// says when it applies), SYN_WALK frame_sample_at block by block (any geometry)
enum { SYN_TILES = 0, SYN_SYMS = 1, SYN_WALK = 2 };
__host__ __device__ inline bool syn_aligned(const TxSynthArgs &a)
{
    return !a.tiles && !a.ft0 && (a.L % 8) == 0 && (a.cp % 8) == 0 && (a.M % 8) == 0 && a.taper >= 0 && a.taper <= 4;
}
template <int K, int R, int IN>
__global__ __launch_bounds__(K / 2) void synth_kernel(TxSynthArgs a, uint32_t slab_blocks)
{
    constexpr int T = K / 2, N = K / 2, C = 2;
    constexpr int S = Plan<K>::S, F = Plan<K>::F, ROWP = Plan<K>::ROWP, TAPF = Lds<K, R>::TAPF, RR = Plan<K>::RR, LR = Plan<K>::LR;
    constexpr int LEAD = (SYN_H + R - 1) / R * R;                   // blocks a slab starts early to fill its window
    extern __shared__ __attribute__((aligned(16))) float2 tile[];   // [R][ROWP], taps, twiddles
    const int tid = threadIdx.x;
    const int n0 = tid * C;
    float *ltap = reinterpret_cast<float *>(tile + R * ROWP);
    {
        constexpr int NT = SYN_P * K / 2 + 1, PER = (NT + T - 1) / T;
        float tv[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) { const int idx = tid + i * T; tv[i] = a.taps[idx < NT ? idx : 0]; }
#pragma unroll
        for (int i = 0; i < PER; i++) { const int idx = tid + i * T; if (idx < NT) ltap[idx] = tv[i]; }
    }
    float2 *ltw = reinterpret_cast<float2 *>(ltap + TAPF);
#pragma unroll
    for (int st = 0; st < S; st++) {
        const int L = K >> (LR * st), q4 = L >> LR;
        for (int e = tid; e < (RR - 1) * q4; e += T) {
            const int r = e / q4 + 1, pos = e % q4;
            float sn, cs; sincos_u32((uint32_t)(r * pos) * (uint32_t)(4294967296.0 / L), sn, cs);
            ltw[Plan<K>::tw_off(st) + e] = make_float2(cs, -sn);
        }
    }
    __syncthreads();

    // my slab of output blocks (local indices, block 0 = a.first_sample_lo): [o0, o1), entered LEAD blocks early
    const long long o0 = (long long)a.out_first + (long long)blockIdx.x * slab_blocks;
    long long o1 = o0 + slab_blocks; if (o1 > (long long)a.nblocks) o1 = (long long)a.nblocks;
    if (o0 >= o1) return;
    const long long bstart = o0 - LEAD;

    const int tapb = opaque(R * ROWP * 2 + n0);                     // direct branches j < p/2: h[n + jK]            (floats from `tile`)
    const int tapm = opaque(R * ROWP * 2 + (K - 1 - n0));           // mirrored, j >= p/2: h[(p-j) K - n]: + (p-1-j) K, column c at -c
    const int xrow = opaque(pad<K>(tid));                           // where channel tid's input goes in a row
    const int vsrc0 = opaque(pad<K>(dif_pos<K>(n0))), vsrc1 = opaque(pad<K>(dif_pos<K>(n0 + 1)));
    const float *ltf = reinterpret_cast<const float *>(tile);
    constexpr int NBF4 = R * (K / RR), NG = R * (K / F);
    constexpr bool SPLIT = (2 * NG == T) && F >= 4;
    static_assert(S == 0 || (NBF4 % T == 0 && T % (K / RR) == 0), "a thread's butterflies differ by whole rows");
    static_assert(NG % T == 0 || SPLIT, "F-point groups per thread");
    constexpr int BTRIPS = S > 0 ? NBF4 / T : 0, BSTEP = S > 0 ? (T / (K / RR)) * ROWP : 0;
    auto stage_index = [&](int st, int t) {
        const int L = K >> (LR * st), q4 = L >> LR;
        const int f = t / (K / RR), j = t % (K / RR);
        return f * ROWP + pad<K>((j / q4) * L + j % q4);
    };
    auto group_index = [&](int g) { return (g / (K / F)) * ROWP + (g % (K / F)) * (F + 1); };

    float2 s[SYN_H + R][C];                                         // s[0..24] history (oldest first), s[25..] the round's blocks
#pragma unroll
    for (int i = 0; i < SYN_H + R; i++)
#pragma unroll
        for (int c = 0; c < C; c++) s[i][c] = make_float2(0.f, 0.f);
    char *outb = reinterpret_cast<char *>(a.out);
    const uint32_t ooff = (uint32_t)n0 * (uint32_t)sizeof(float2);
    // Oscillator e^{+j t dtheta}: exact 32-bit phase through v_sin / v_cos at the first column of every group of 8 blocks
    // (groups start on multiples of 8 of the launch's block axis, which callers keep aligned with the absolute one), then
    // turned by the per-block step K dtheta and, for the second column, by dtheta -- explicit fma shapes, so a block's
    // samples are a fixed function of its absolute index whatever slab, launch or rank makes them.
    float sd1, cd1; sincos_u32(a.dtheta, sd1, cd1);
    float sk8, ck8; sincos_u32((uint32_t)K * a.dtheta, sk8, ck8);
    sd1 = __builtin_bit_cast(float, uniform_i(__builtin_bit_cast(int, sd1))); cd1 = __builtin_bit_cast(float, uniform_i(__builtin_bit_cast(int, cd1)));
    sk8 = __builtin_bit_cast(float, uniform_i(__builtin_bit_cast(int, sk8))); ck8 = __builtin_bit_cast(float, uniform_i(__builtin_bit_cast(int, ck8)));
    float osn = 0.f, ocs = 1.f;

    const int rounds = (int)((o1 - bstart + R - 1) / R);
    // Inputs of a round: channel tid of blocks b0 .. b0+R-1.  Granules received from the channel shards ([g][tile][c][8],
    // channel = g cg + c; R consecutive blocks of a granule are contiguous: R divides 8 and rounds start on multiples of R) are
    // requested one round ahead with clamped addresses -- no load sits under a branch, so the request stays in flight across
    // the whole round -- and blocks outside the stream are zeroed when they are written to the tile.
    constexpr bool tiled = IN == SYN_TILES;
    const uint32_t tg = tiled ? (uint32_t)tid / a.cg : 0u, tc = tiled ? (uint32_t)tid % a.cg : 0u;
    float4 xq[R / 2];
    (void)tg; (void)tc;
    auto request_tiles = [&](long long b0) {
#pragma unroll
        for (int r = 0; r < R; r += 2) {
            long long b = b0 + r;
            if (!(b >= 0 && b + 1 < (long long)a.nblocks)) b = 0;
            xq[r / 2] = *reinterpret_cast<const float4 *>(a.tiles + (((size_t)tg * a.ntiles + (size_t)(b >> 3)) * a.cg + tc) * 8 + (size_t)(b & 7));
        }
    };
    // Batch / ragged layout when a round of 8 blocks never straddles an OFDM symbol (8 | L, cp, M; rounds start on multiples of 8:
    // every configuration of the reference's applications): the round's inputs are 64 contiguous, 64-byte aligned bytes of one
    // symbol body -- four 16-byte loads with clamped addresses, requested one round ahead like the granules -- plus the first four
    // samples of the previous body for the raised-cosine overlap; the symbol's role (ragged traffic: a byte per symbol and channel)
    // is requested two rounds ahead so that no address waits for it.  What the samples mean is decided when they are consumed.
    constexpr bool fastsym = IN == SYN_SYMS;
    static_assert(!fastsym || R == 8, "the aligned loader works in rounds of 8 blocks");
    const uint32_t nsym = (uint32_t)(a.frames * a.S);
    const float2 *xch = xs_channel(a, (uint32_t)tid);
    const size_t xstep = xs_sym_of(a);
    float4 pq[2];
    float twin[4] = { 0.f, 0.f, 0.f, 0.f };                             // the raised-cosine ramp (taper <= 4 here), read once
    if constexpr (fastsym) {
#pragma unroll
        for (int k = 0; k < 4; k++) if (k < a.taper) twin[k] = a.taperwin[k];
    }
    // What a round's 64 bytes mean is decided by very little: txsym writes zeros into the bodies of idle and tail symbols, so
    // outside a symbol's first round the samples are used as they arrive; S0a differs by its read offset (two prefixes back),
    // and in a symbol's first round (i = 0, the same on every channel) the first `taper` samples are the raised-cosine overlap
    // with the previous body -- for every role but S0b, whose reference form has none: a zero previous body (S0a) or a zero own
    // body (tail) make frame_sample_sym's three special cases instances of the one fma shape.
    uint32_t kq = 0;                                                    // ragged: kind of the symbol the NEXT request reads
    int s0b_q = 0;                                                      // the round in xq belongs to an S0b symbol
    // position of the next request on the symbol axis, walked (uniform): block bq, symbol gq, offset iq in it
    long long bq = bstart;
    uint32_t gq = (uint32_t)(bstart < 0 ? 0 : bstart) / (uint32_t)a.L, iq = (uint32_t)(bstart < 0 ? 0 : bstart) % (uint32_t)a.L;
    gq = (uint32_t)uniform_i((int)gq); iq = (uint32_t)uniform_i((int)iq);
    uint32_t gf = 0, if_ = 0; bool okf = false;                         // ... of the round whose samples are in xq
    // The overlap needs the PREVIOUS body's first `taper` samples.  They went through this thread's registers when that symbol's round
    // with read offset 0 was consumed (i = cp; S0a: 2 cp), so they are kept from there -- 8 registers -- instead of being fetched again
    // at every symbol start: 32 bytes used of a line that had been evicted by then, a ninth of the kernel's fetches (round 5).  Only a
    // slab's first symbol start, whose predecessor's round belonged to the slab before, still asks memory (req_keep_g != gq - 1).
    float4 keep[2] = { make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f) };
    uint32_t req_keep_g = ~0u;                                          // symbol whose offset-0 round this workgroup has requested
    bool keep_f = false, pq_f = false;                                  // the round in xq: is the offset-0 round of its symbol / its overlap comes from pq
    auto step = [&](long long &bb, uint32_t &g, uint32_t &i) {
        if (bb >= 0) { i += R; if (i >= (uint32_t)a.L) { i -= (uint32_t)a.L; g++; } }
        bb += R;
    };
    auto request_kind = [&]() {                                         // for the next request (the position has moved on to it)
        if (a.symkind) {
            const uint32_t gk = gq < nsym ? gq : nsym - 1;
            kq = a.symkind[a.ks_sym ? (size_t)tid * a.ks_ch + (size_t)gk * a.ks_sym : (size_t)tid * nsym + gk];
        }
    };
    auto request_syms = [&]() {
        if constexpr (R == 8) {
            const bool ok = bq >= 0 && bq < (long long)a.nblocks && gq < nsym;
            bool s0a;
            if (a.symkind) { s0a = kq == TXK_S0A; s0b_q = kq == TXK_S0B; }
            else { const int sidx = (int)(gq % (uint32_t)a.S); s0a = sidx == 0; s0b_q = sidx == 1; }
            gf = gq; if_ = iq; okf = ok;
            keep_f = ok && iq == (s0a ? 2u : 1u) * (uint32_t)a.cp;
            const bool have_prev = iq == 0 && gq > 0 && req_keep_g == gq - 1u;
            if (keep_f) req_keep_g = gq;
            const uint32_t gsc = ok ? gq : 0u;
            const float2 *x = xch + (size_t)gsc * xstep;
            const uint32_t base1 = (iq + (uint32_t)a.M - (uint32_t)a.cp) % (uint32_t)a.M, base2 = (iq + (uint32_t)a.M - 2u * (uint32_t)a.cp) % (uint32_t)a.M;
            const float4 *xp = reinterpret_cast<const float4 *>(x + xs_at(a.xs_grp, s0a ? base2 : base1));      // (8 | base: one group of 8 samples)
#pragma unroll
            for (int q = 0; q < 4; q++) xq[q] = xp[q];
            pq_f = iq == 0 && !have_prev;
            if (pq_f) {                                                 // (uniform) the previous body's first samples: the overlap
                const float4 *pp = reinterpret_cast<const float4 *>(gsc > 0 ? x - xstep : x);
                pq[0] = pp[0]; pq[1] = pp[1];
            }
            step(bq, gq, iq);
        }
    };
    auto finish_syms = [&](long long b0, float2 (&xin)[R]) {
        if constexpr (R == 8) {
#pragma unroll
            for (int r = 0; r < 8; r++) xin[r] = (r & 1) ? make_float2(xq[r / 2].z, xq[r / 2].w) : make_float2(xq[r / 2].x, xq[r / 2].y);
            if (!okf) {
#pragma unroll
                for (int r = 0; r < 8; r++) xin[r] = make_float2(0.f, 0.f);
            } else if (if_ == 0) {
                const float4 o0 = pq_f ? pq[0] : keep[0], o1 = pq_f ? pq[1] : keep[1];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (r < a.taper) {
                        const float2 p = r == 0 ? make_float2(o0.x, o0.y) : (r == 1 ? make_float2(o0.z, o0.w) : (r == 2 ? make_float2(o1.x, o1.y) : make_float2(o1.z, o1.w)));
                        const int kb = a.taper - 1 - r;
                        const float wa = twin[r], wb = gf > 0 ? (kb == 0 ? twin[0] : (kb == 1 ? twin[1] : (kb == 2 ? twin[2] : twin[3]))) : 0.f;
                        const float2 bl = taper_blend(xin[r], wa, p, wb);
                        if (!s0b_q) xin[r] = bl;
                    }
                }
            }
            if (keep_f) { keep[0] = xq[0]; keep[1] = xq[1]; }          // (this body's first samples, as loaded: the next symbol's overlap)
            if (b0 + R > (long long)a.nblocks) {                        // (the stream's last, partly filled round)
#pragma unroll
                for (int r = 0; r < 8; r++) if (b0 + r >= (long long)a.nblocks) xin[r] = make_float2(0.f, 0.f);
            }
        }
    };
    if constexpr (SYN_ABLATE & 16) { for (int q = 0; q < R / 2; q++) xq[q] = make_float4(1.f, 0.f, 0.f, 1.f); pq[0] = pq[1] = make_float4(0.f, 0.f, 0.f, 0.f); okf = true; if_ = 8; }
    else if constexpr (tiled) request_tiles(bstart);
    else if constexpr (fastsym) {
        pq[0] = pq[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        request_kind(); request_syms(); request_kind();
    }
    // (the first round's inputs are waited for here, so that the loop is entered with nothing pending on either path and the
    //  compiler puts no wait at its top -- see the wait in front of the stores)
    if constexpr (tiled || fastsym) {
#pragma unroll
        for (int q = 0; q < R / 2; q++) asm volatile("" : "+v"(xq[q].x), "+v"(xq[q].y), "+v"(xq[q].z), "+v"(xq[q].w) :: "memory");
    }
    if constexpr (fastsym) asm volatile("" : "+v"(pq[0].x), "+v"(pq[0].y), "+v"(pq[0].z), "+v"(pq[0].w), "+v"(pq[1].x), "+v"(pq[1].y), "+v"(pq[1].z), "+v"(pq[1].w), "+v"(kq) :: "memory");
    for (int rd = 0; rd < rounds; rd++) {
        const long long b0 = bstart + (long long)rd * R;
        const int tq = opaque(tid);
        // ---- inputs, conjugated (inverse transform = conj(forward(conj))); bins >= N are zero and never stored: the first
        //      radix-4 stage below knows it
        float2 xin[R];
        if constexpr (tiled) {
#pragma unroll
            for (int r = 0; r < R; r += 2) { xin[r] = make_float2(xq[r / 2].x, xq[r / 2].y); xin[r + 1] = make_float2(xq[r / 2].z, xq[r / 2].w); }
            if (b0 < 0 || b0 + R >= (long long)a.nblocks) {             // (uniform, the stream's edges only: pairs outside it are zero)
#pragma unroll
                for (int r = 0; r < R; r += 2) {
                    const long long b = b0 + r;
                    if (!(b >= 0 && b + 1 < (long long)a.nblocks)) xin[r] = xin[r + 1] = make_float2(0.f, 0.f);
                }
            }
        } else if constexpr (fastsym) {
            finish_syms(b0, xin);
        } else {
            // batch / ragged layout: the channel's frame axis, symbol gs = b / L, position i = b % L -- one division per
            // round, then the position walks on (frame_sample_at: the role of symbol gs is looked up where it changes)
            const long long bc = b0 < 0 ? 0 : b0;
            uint32_t gs = (uint32_t)bc / (uint32_t)a.L, i = (uint32_t)bc % (uint32_t)a.L;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const long long b = b0 + r;
                const bool in = b >= 0 && b < (long long)a.nblocks;
                xin[r] = in ? frame_sample_at(a, (uint32_t)tid, gs, i) : make_float2(0.f, 0.f);
                if (in) { i++; if (i == (uint32_t)a.L) { i = 0; gs++; } }
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) tile[r * ROWP + xrow] = make_float2(xin[r].x, -xin[r].y);
        if ((SYN_ABLATE & 16) == 0 && rd + 1 < rounds) {
            if constexpr (tiled) request_tiles(b0 + R);
            else if constexpr (fastsym) { request_syms(); request_kind(); }
        }
        lds_barrier();
        // ---- R forward K-point transforms in place (channelizer.hip's plan)
        if constexpr (S > 0 && (SYN_ABLATE & 1) == 0) {
#pragma unroll
            for (int st = 0; st < S; st++) {
                const int L = K >> (LR * st), q4 = L >> LR;
                const int D = q4 + q4 / F;
                const float2 *twp = tile + R * ROWP + TAPF / 2 + Plan<K>::tw_off(st) + tq % q4;
                const int fa_st = stage_index(st, tq);
                if constexpr (RR == 8) {
                    float2 tw[7];
#pragma unroll
                    for (int r = 0; r < 7; r++) tw[r] = twp[r * q4];
#pragma unroll
                    for (int i = 0; i < BTRIPS; i++) {
                        float2 *p = tile + fa_st + i * BSTEP;
                        float2 v[8];
                        if (st == 0) {
                            // legs 4 .. 7 are the zero bins N .. K-1: the first level's sums are the inputs themselves, its
                            // differences the inputs turned by W_8^m; two 4-point transforms give the even and the odd outputs
                            float2 e[4], o[4];
#pragma unroll
                            for (int m = 0; m < 4; m++) e[m] = p[m * D];
                            o[0] = e[0]; o[1] = cmul(e[1], w16(2)); o[2] = cmulnj(e[2]); o[3] = cmul(e[3], w16(6));
                            fft_reg<4>(e); fft_reg<4>(o);           // e[m] = E[bitrev(m)], o likewise
#pragma unroll
                            for (int k = 0; k < 4; k++) { v[2 * k] = e[bitrev_c(k, 2)]; v[2 * k + 1] = o[bitrev_c(k, 2)]; }     // v[r] = X[r]
                            p[0] = v[0];
#pragma unroll
                            for (int r = 1; r < 8; r++) p[r * D] = cmul(v[r], tw[r - 1]);
                        } else {
#pragma unroll
                            for (int m = 0; m < 8; m++) v[m] = p[m * D];
                            fft_reg<8>(v);                          // v[m] = X[bitrev(m)]
                            p[0] = v[0];
#pragma unroll
                            for (int r = 1; r < 8; r++) p[r * D] = cmul(v[bitrev_c(r, 3)], tw[r - 1]);
                        }
                    }
                    lds_barrier();
                    continue;
                }
                const float2 tw1 = twp[0], tw2 = twp[q4], tw3 = twp[2 * q4];
                if (st == 0) {
                    // legs 2 and 3 are the zero bins N .. K-1: a0 = a1 = x0, a2 = x1, a3 = -j x1 (the values the full butterfly
                    // forms from x2 = x3 = 0)
#pragma unroll
                    for (int i = 0; i < BTRIPS; i++) {
                        float2 *p = tile + fa_st + i * BSTEP;
                        const float2 x0 = p[0], x1 = p[D];
                        p[0] = cadd(x0, x1);
                        p[D] = cmul(cadd_nj(x0, x1), tw1);
                        p[2 * D] = cmul(csub(x0, x1), tw2);
                        p[3 * D] = cmul(csub_nj(x0, x1), tw3);
                    }
                    lds_barrier();
                    continue;
                }
#pragma unroll
                for (int i = 0; i < BTRIPS; i++) {
                    float2 *p = tile + fa_st + i * BSTEP;
                    const float2 x0 = p[0], x1 = p[D], x2 = p[2 * D], x3 = p[3 * D];
                    const float2 a0 = cadd(x0, x2), a1 = csub(x0, x2), a2 = cadd(x1, x3), d3 = csub(x1, x3);
                    p[0] = cadd(a0, a2);
                    p[D] = cmul(cadd_nj(a1, d3), tw1);              // a1 + (-j) d3: the rotation rides on the add (devmath.h)
                    p[2 * D] = cmul(csub(a0, a2), tw2);
                    p[3 * D] = cmul(csub_nj(a1, d3), tw3);
                }
                lds_barrier();
            }
        }
        if constexpr ((SYN_ABLATE & 2) != 0) {
        } else if constexpr (SPLIT) {
            // half as many F-point groups as threads: waves 0 .. NG/64-1 transform the sums x[i] + x[i+F/2] (even bins), the
            // others the twiddled differences (odd bins); both read the whole group and write into it, hence the barrier
            const int role = uniform_i(tq / NG);
            float2 *p = tile + group_index(tq - role * NG);
            float2 x[F], v[F / 2];
#pragma unroll
            for (int m = 0; m < F; m++) x[m] = p[m];
            lds_barrier();
            if (role == 0) {
#pragma unroll
                for (int i = 0; i < F / 2; i++) v[i] = cadd(x[i], x[i + F / 2]);
            } else {
#pragma unroll
                for (int i = 0; i < F / 2; i++) {
                    const float2 d = csub(x[i], x[i + F / 2]);
                    const int tk = i * (16 / F);
                    if (tk == 0) v[i] = d; else if (tk == 4) v[i] = cmulnj(d); else v[i] = cmul(d, w16(tk));
                }
            }
            fft_reg<F / 2>(v);
#pragma unroll
            for (int m = 0; m < F / 2; m++) p[2 * bitrev_c(m, Log2<F / 2>::v) + role] = v[m];
            lds_barrier();
        } else {
            constexpr int GTRIPS = NG / T;
#pragma unroll
            for (int i = 0; i < GTRIPS; i++) {
                float2 *p = tile + group_index(tq + i * T);
                float2 v[F];
#pragma unroll
                for (int m = 0; m < F; m++) v[m] = p[m];
                fft_reg<F>(v);
#pragma unroll
                for (int m = 0; m < F; m++) p[bitrev_c(m, Log2<F>::v)] = v[m];
            }
            lds_barrier();
        }
        // ---- my two columns of the R new blocks (conjugated back), into the window
#pragma unroll
        for (int r = 0; r < R; r++) {
            const float2 u0 = tile[r * ROWP + vsrc0], u1 = tile[r * ROWP + vsrc1];
            s[SYN_H + r][0] = make_float2(u0.x, -u0.y); s[SYN_H + r][1] = make_float2(u1.x, -u1.y);
        }
        // ---- synthesis FIR, oscillator, gain, store (rounds that only fill the window skip it)
        if (b0 + R > o0) {
            // taps outermost, oldest first (the order of the window dot product for every output): a tap pair is read from
            // LDS where it is used and serves the round's R blocks, so the prototype costs two registers instead of 2 p
            float2 acc[R][C];
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int c = 0; c < C; c++) acc[r][c] = make_float2(0.f, 0.f);
            // A tap pair (columns n0, n0 + 1) is one 64-bit LDS read and feeds v_pk_fma_f32 as it stands: op_sel broadcasts its low
            // or high half to both components of the product, so no (t, t) pairs are built (the compiler's form: two moves and two
            // registers per tap and column -- what spilled the window).  Direct branches hold column 0 in the low half, mirrored
            // ones in the high half.  Reads run two pairs ahead of the multiplies; the memory fences keep that order.
            auto tap_pair = [&](int j) -> v2f {
                v2f t;
                if (j >= SYN_P / 2) { t.x = ltf[tapm - 1 + (SYN_P - 1 - j) * K + 1]; t.y = ltf[tapm + (SYN_P - 1 - j) * K + 1]; }
                else { t.x = ltf[tapb + j * K]; t.y = ltf[tapb + 1 + j * K]; }
                return t;
            };
            v2f tq0 = tap_pair(SYN_P - 1), tq1 = tap_pair(SYN_P - 2);
#pragma unroll
            for (int j = SYN_P - 1; j >= ((SYN_ABLATE & 4) ? SYN_P - 1 : 0); j--) {
                const v2f tp = tq0;
                tq0 = tq1;
                if (j >= 2) tq1 = tap_pair(j - 2);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if ((j >= SYN_P / 2) == false) {
                        pk_fma_lo(acc[r][0], s[SYN_H + r - j][0], tp);
                        pk_fma_hi(acc[r][1], s[SYN_H + r - j][1], tp);
                    } else {
                        pk_fma_hi(acc[r][0], s[SYN_H + r - j][0], tp);
                        pk_fma_lo(acc[r][1], s[SYN_H + r - j][1], tp);
                    }
                }
            }
            // Loads and stores share one counter and complete out of order against each other, so a wait for the prefetched inputs
            // also waits for every store in flight.  Placed here -- behind a round of arithmetic, in front of this round's stores --
            // it finds the inputs (requested at the top of the round) and the previous round's stores long complete; at the top of
            // the next round, where the inputs are used, it would find this round's stores just issued.
            if constexpr ((SYN_ABLATE & 64) == 0) {
                if constexpr (tiled || fastsym) {
#pragma unroll
                    for (int q = 0; q < R / 2; q++) asm volatile("" : "+v"(xq[q].x), "+v"(xq[q].y), "+v"(xq[q].z), "+v"(xq[q].w) :: "memory");
                }
                if constexpr (fastsym) asm volatile("" : "+v"(pq[0].x), "+v"(pq[0].y), "+v"(pq[0].z), "+v"(pq[0].w), "+v"(pq[1].x), "+v"(pq[1].y), "+v"(pq[1].z), "+v"(pq[1].w), "+v"(kq) :: "memory");
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const long long b = b0 + r;
                if ((b & 7) == 0)       // first block of a group (slabs start on multiples of 8, so a slab's first output block is one)
                    sincos_u32_hw((a.first_sample_lo + (uint32_t)((unsigned long long)b * K + (unsigned)n0)) * a.dtheta, osn, ocs);
                if (b >= o0 && b < o1 && (!(SYN_ABLATE & 8) || acc[r][0].x == 1.2345e30f)) {
                    const float s1 = fmaf(osn, cd1, ocs * sd1), c1 = fmaf(ocs, cd1, -(osn * sd1));            // the second column
                    const float2 y0 = make_float2(fmaf(acc[r][0].x, ocs, -(acc[r][0].y * osn)), fmaf(acc[r][0].y, ocs, acc[r][0].x * osn));
                    const float2 y1 = make_float2(fmaf(acc[r][1].x, c1, -(acc[r][1].y * s1)), fmaf(acc[r][1].y, c1, acc[r][1].x * s1));
#if SYN_NT_STORE
                    { typedef float v4f __attribute__((ext_vector_type(4)));
                      const v4f nv = { y0.x * a.gain, y0.y * a.gain, y1.x * a.gain, y1.y * a.gain };
                      __builtin_nontemporal_store(nv, reinterpret_cast<v4f *>(uniform_ptr(outb + (size_t)(b - (long long)a.out_first) * K * sizeof(float2)) + ooff)); }
#else
                    *reinterpret_cast<float4 *>(uniform_ptr(outb + (size_t)(b - (long long)a.out_first) * K * sizeof(float2)) + ooff) =
                        make_float4(y0.x * a.gain, y0.y * a.gain, y1.x * a.gain, y1.y * a.gain);
#endif
                }
                { const float s2 = fmaf(osn, ck8, ocs * sk8), c2 = fmaf(ocs, ck8, -(osn * sk8)); osn = s2; ocs = c2; }      // next block
            }
        }
        else if constexpr ((SYN_ABLATE & 64) == 0) {                  // (rounds that only fill the window: same wait, so that no path reaches the loop's top with loads pending)
            if constexpr (tiled || fastsym) {
#pragma unroll
                for (int q = 0; q < R / 2; q++) asm volatile("" : "+v"(xq[q].x), "+v"(xq[q].y), "+v"(xq[q].z), "+v"(xq[q].w) :: "memory");
            }
            if constexpr (fastsym) asm volatile("" : "+v"(pq[0].x), "+v"(pq[0].y), "+v"(pq[0].z), "+v"(pq[0].w), "+v"(pq[1].x), "+v"(pq[1].y), "+v"(pq[1].z), "+v"(pq[1].w), "+v"(kq) :: "memory");
        }
        if constexpr ((SYN_ABLATE & 32) == 0) {
#pragma unroll
        for (int i = 0; i < SYN_H; i++)
#pragma unroll
            for (int c = 0; c < C; c++) {                            // one packed move per sample (the compiler's form: two v_mov_b32)
                v2f d; const v2f o = { s[i + R][c].x, s[i + R][c].y };
                asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]" : "=v"(d) : "v"(o));
                s[i][c] = make_float2(d.x, d.y);
            }
        }
        lds_barrier();
    }
}

}  // namespace syn
}  // namespace mcrx
