// txgen.hip -- multichanneltx on the GPU: the synthetic IQ source of the receive path.
//
// Replaces liquid-usrp's multichanneltx (lib/multichanneltx.cc: ctor :41-100, UpdateData
// :165-189, GenerateSamples :192-227, GenerateFrameSamples :230-242) driven by the traffic
// recipe of src/multichannel_tx.cc:163-213 (header = [pid_hi, pid_lo, channel, 5 random bytes],
// random payloads, frames back to back on every channel, soft gain), writing the wideband cf32
// stream straight into HBM:
//   1. txsym_kernel   one wave per (channel, OFDM symbol): subcarrier mapping (data symbols
//                     from the host-assembled frame bits, pilots from the order-8 m-sequence,
//                     gain 1/sqrt(M_pilot+M_data)) and the M-point inverse FFT across the lanes;
//   2. txifft_kernel  per block b: X[k<N] = frame sample b of channel k (cyclic prefix and
//                     raised-cosine overlap applied on the fly), 2N-point inverse FFT in LDS;
//   3. txfir_kernel   polyphase synthesis FIR (2N channels, m = 13 -> 26 taps per branch)
//                     y_b[i] = sum_j h[i + jK] v_{b-j}[i], NCO mix-up by the exact 32-bit phase
//                     (bK+i)*dtheta, soft gain.
// Bit-level frame assembly (CRC/FEC/interleaver/scrambler) is host code: txcode.hpp.
#include "../../include/mcrx_hip.h"
#include "devel.h"
#include "devmath.h"
#include "txcode.hpp"
#include "devscope.hpp"
#include <random>
#include <string>
#include <vector>

namespace mcrx {

#define TXW 64
#define TX_P 26                 // synthesis taps per branch (m = 13)

struct TxSymArgs {
    int M, log2M, cp, taper, L, M_pilot, M_data, S, S_hdr, S_pay, frames, bps, mod;
    float g_data;
    const uint8_t *sctype; const int16_t *data_rank, *pilot_rank; const uint8_t *pilot_seq;
    const float2 *s0t, *s1t;
    const uint8_t *hdr;         // [ch][frame][S_hdr*M_data]
    const uint8_t *pay;         // [ch][frame][S_pay*M_data]
    float2 *xsym;               // [ch][frames*S][M], or symbol gs of channel ch at ch xs_ch + gs xs_sym when xs_sym != 0
    size_t xs_ch = 0, xs_sym = 0;
    size_t xs_grp = 0;          // != 0: sample n of a body sits (n >> 3) xs_grp + (n & 7) behind its start (TxSynthArgs::xs_grp)
    uint32_t nch;
    // ragged traffic (mctx_hip_generate_ragged): frames of different lengths anywhere on the channel's symbol axis.
    // frames = 1, S = symbols of the whole axis; symdesc[ch][S] says what symbol gs is: kind | s << 8 | row << 32
    // (kind: TXK_* below; s: index inside its frame; row: its M_data bytes in hdr / pay)
    const unsigned long long *symdesc = nullptr;
};
__device__ __forceinline__ size_t xs_at(size_t xs_grp, uint32_t n);
enum { TXK_IDLE = 0, TXK_S0A = 1, TXK_S0B = 2, TXK_S1 = 3, TXK_HDR = 4, TXK_PAY = 5, TXK_TAIL = 6 };

__device__ __forceinline__ unsigned gray_dec_t(unsigned x) { unsigned y = x; while (x >>= 1) y ^= x; return y; }
__device__ __forceinline__ float2 modulate(int mod, unsigned sym)
{
    if (mod == 39) return make_float2(sym ? -1.0f : 1.0f, 0.f);
    if (mod == 40) { const float a = 0.70710678118654752f; return make_float2((sym & 1) ? -a : a, (sym & 2) ? -a : a); }
    const unsigned bps = (mod == 27) ? 4u : 6u, mq = bps / 2;
    const float alpha = (mod == 27) ? 0.31622776601683794f : 0.1543033499620919f;
    const int Lq = 1 << mq;
    const int gi = 2 * (int)gray_dec_t(sym >> mq) - Lq + 1, gq = 2 * (int)gray_dec_t(sym & (unsigned)(Lq - 1)) - Lq + 1;
    return make_float2((float)gi * alpha, (float)gq * alpha);
}

// one (channel, global symbol index): time-domain symbol body x[M] (no prefix yet); twl = the six lane-stage twiddles of lane l.
// Two halves, so that a wave can have the loads of the symbols behind the current one in flight while it transforms it (one symbol
// after the other, each waiting for its subcarrier map, then its rank, then its byte: 0.59 ms per 1.44 M symbols, two thirds of the
// wave cycles waiting): txsym_fetch requests what the symbol needs from memory, txsym_emit consumes it.
struct TxSymWhat { int s; bool table, zero, is_hdr; const uint8_t *bits; };
__device__ __forceinline__ TxSymWhat txsym_what(const TxSymArgs &a, const uint32_t gs, const uint32_t ch, const unsigned long long d, int f = -1, int sidx = 0)
{
    TxSymWhat w;
    if (f < 0) { f = gs / a.S; sidx = gs % a.S; }       // (callers that walk the symbol axis pass frame and index)
    w.s = sidx;
    w.table = w.s < 3 || w.s == a.S - 1; w.zero = w.s == a.S - 1; w.is_hdr = w.s < 3 + a.S_hdr;
    w.bits = a.hdr;                                  // (always mapped: table symbols load nothing from it)
    if (a.symdesc) {
        const int kind = (int)(d & 0xff);
        w.s = (int)((d >> 8) & 0xffff);
        w.table = kind != TXK_HDR && kind != TXK_PAY; w.zero = kind == TXK_IDLE || kind == TXK_TAIL; w.is_hdr = kind == TXK_HDR;
        if (!w.table) w.bits = (w.is_hdr ? a.hdr : a.pay) + (size_t)(d >> 32) * a.M_data;
    } else if (!w.table)
        w.bits = w.is_hdr ? a.hdr + ((size_t)ch * a.frames + f) * a.S_hdr * a.M_data + (size_t)(w.s - 3) * a.M_data
                          : a.pay + ((size_t)ch * a.frames + f) * a.S_pay * a.M_data + (size_t)(w.s - 3 - a.S_hdr) * a.M_data;
    return w;
}
template <int E> struct TxSymPre { unsigned long long d; uint8_t bit[E], pil[E]; };
// the lane's subcarriers: type, rank among the data / pilot subcarriers -- the same for every symbol, read once per wave
template <int E> struct TxSymLane { int t[E], dr[E], pr[E]; };
template <int E>
__device__ __forceinline__ TxSymPre<E> txsym_fetch(const TxSymArgs &a, const uint32_t gs, const uint32_t ch, const TxSymLane<E> &ln)
{
    TxSymPre<E> p;
    p.d = a.symdesc ? a.symdesc[(size_t)ch * a.S + gs] : 0ull;
    const TxSymWhat w = txsym_what(a, gs, ch, p.d);
    const uint32_t pcount = (uint32_t)(w.s >= 3 ? w.s - 3 : 0) * (uint32_t)a.M_pilot;      // pilot generator resets per frame
#pragma unroll
    for (int e = 0; e < E; e++) {
        p.bit[e] = w.bits[(!w.table && ln.t[e] == 2) ? ln.dr[e] : 0];          // (no load under a branch: it would be waited for at the join)
        p.pil[e] = a.pilot_seq[(pcount + (uint32_t)ln.pr[e]) % 255u];
    }
    return p;
}
template <int E>
__device__ __forceinline__ void txsym_emit(const TxSymArgs &a, const uint32_t gs, const uint32_t ch, const int l, const float2 (&twl)[6],
                                           const TxSymLane<E> &ln, const TxSymPre<E> &pre)
{
    const TxSymWhat w = txsym_what(a, gs, ch, pre.d);
    float2 *dst = a.xsym + (a.xs_sym ? (size_t)ch * a.xs_ch + (size_t)gs * a.xs_sym : ((size_t)ch * a.frames * a.S + gs) * a.M);
    if (w.table) {                                  // S0a, S0b, S1 bodies come from the tables; tail (and idle symbols) have none
        const float2 *src = (w.s == 2) ? a.s1t : a.s0t;
        for (int i = l; i < a.M; i += TXW) dst[xs_at(a.xs_grp, (uint32_t)i)] = w.zero ? make_float2(0.f, 0.f) : src[i];
        return;
    }
    float2 x[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        float2 v = make_float2(0.f, 0.f);
        if (ln.t[e] == 1) v = make_float2(pre.pil[e] ? a.g_data : -a.g_data, 0.f);
        else if (ln.t[e] == 2) { v = modulate(w.is_hdr ? 39 : a.mod, pre.bit[e]); v.x *= a.g_data; v.y *= a.g_data; }
        x[e] = make_float2(v.x, -v.y);              // inverse FFT = conj(FFT(conj(X)))
    }
    // forward DIF across position i = l + 64 e (natural subcarrier order in), bit-reversed out
#pragma unroll
    for (int j = E / 2; j >= 1; j >>= 1) {
#pragma unroll
        for (int e = 0; e < E; e++) if ((e & j) == 0 && e + j < E) {
            const int h = TXW * j;
            const float2 u = x[e], w2 = x[e + j];
            float sn, cs; sincos_u32((uint32_t)((l + TXW * e) & (h - 1)) * (uint32_t)(0x80000000u / (unsigned)h), sn, cs);
            x[e] = cadd(u, w2);
            x[e + j] = cmul(csub(u, w2), make_float2(cs, -sn));
        }
    }
#pragma unroll
    for (int st = 0; st < 6; st++) {
        const int h = 32 >> st;
        if (h < a.M) {
            const float2 tw = twl[st];
            const bool up = (l & h) != 0;
#pragma unroll
            for (int e = 0; e < E; e++) {
                const float2 p = make_float2(__shfl_xor(x[e].x, h, TXW), __shfl_xor(x[e].y, h, TXW));
                x[e] = up ? cmul(csub(p, x[e]), tw) : cadd(x[e], p);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int i = l + TXW * e;
        if (i < a.M) {
            const int n = (int)(__brev((unsigned)i) >> (32 - a.log2M));
            dst[xs_at(a.xs_grp, (uint32_t)n)] = make_float2(x[e].x, -x[e].y);
        }
    }
}
// a wave makes TXSYM_PER consecutive symbols of one channel: the lane twiddles (a sin / cos pair per stage), the lane's subcarrier
// map and the launch's per-wave set-up are paid once for them (one symbol per wave: 403 M VALU + 201 M SALU instructions per
// 1.44 M symbols); narrow symbols (E <= 2) have all eight symbols' bytes requested before the first transform, wider ones the next one's
#define TXSYM_PER 8
template <int E>
__global__ __launch_bounds__(TXW) void txsym_kernel(TxSymArgs a, uint32_t nsym)
{
    const int l = threadIdx.x & 63;
    const uint32_t ch = blockIdx.y;
    float2 twl[6];
#pragma unroll
    for (int st = 0; st < 6; st++) {
        const int h = 32 >> st;
        float sn, cs; sincos_u32((uint32_t)(l & (h - 1)) * (uint32_t)(0x80000000u / (unsigned)h), sn, cs);
        twl[st] = make_float2(cs, -sn);
    }
    TxSymLane<E> ln;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int k = l + TXW * e, kk = k < a.M ? k : 0;
        ln.t[e] = k < a.M ? (int)a.sctype[kk] : 0; ln.dr[e] = a.data_rank[kk]; ln.pr[e] = a.pilot_rank[kk];
        if (ln.t[e] != 2) ln.dr[e] = 0;
        if (ln.t[e] != 1) ln.pr[e] = 0;
    }
    constexpr int D = E <= 2 ? TXSYM_PER : 1;       // symbols requested ahead
    const uint32_t gs0 = blockIdx.x * TXSYM_PER;
    TxSymPre<E> pre[D];
#pragma unroll
    for (int d = 0; d < D; d++) { const uint32_t gs = gs0 + d; pre[d] = txsym_fetch<E>(a, gs < nsym ? gs : nsym - 1, ch, ln); }
#pragma unroll
    for (int k = 0; k < TXSYM_PER; k++) {
        const uint32_t gs = gs0 + k;
        const TxSymPre<E> cur = pre[k % D];
        if (k + D < TXSYM_PER) { const uint32_t gn = gs0 + k + D; pre[k % D] = txsym_fetch<E>(a, gn < nsym ? gn : nsym - 1, ch, ln); }
        if (gs < nsym) txsym_emit<E>(a, gs, ch, l, twl, ln, cur);
    }
}

// M = 64 (the benchmark's and the applications' symbol): the wave's eight symbols side by side, eight lanes each, a lane holding
// the eight subcarriers i = j + 8 e of its symbol (j = lane & 7).  The transform's first three stages (partners 32, 16, 8 apart) are
// register butterflies with the lane's own twiddles, the last three cross lanes inside the group of eight; every lane ends with the
// eight CONTIGUOUS outputs n = 8 bitrev3(j) + bitrev3(e), written as four 16-byte stores.  One pass of ~350 wave instructions per eight
// symbols where the one-point-per-lane kernel above runs eight passes of ~125 (0.43 ms per 1.44 M symbols, the VALU its bound).
__global__ __launch_bounds__(TXW) void txsym64_kernel(TxSymArgs a, uint32_t nsym)
{
    const int l = threadIdx.x & 63, j = l & 7;
    // (bodies in channel-interleaved groups, TxSynthArgs::xs_grp: a lane's 64 bytes are half a cache line whose other half is the
    //  neighbouring CHANNEL's -- the wave then takes four symbols of two channels, so that its stores are whole lines)
    const bool pair = a.xs_grp != 0 && (a.nch & 1u) == 0;
    const uint32_t per = pair ? TXSYM_PER / 2 : TXSYM_PER;
    const uint32_t ch = pair ? 2u * blockIdx.y + (uint32_t)((l >> 3) & 1) : blockIdx.y;
    const uint32_t gs_raw = blockIdx.x * per + (uint32_t)(pair ? (l >> 4) : (l >> 3));
    const bool live = gs_raw < nsym;
    const uint32_t gs = live ? gs_raw : nsym - 1;
    // what this group's symbol is, and its bytes (requested first: everything below up to the modulator is independent of them)
    const unsigned long long d = a.symdesc ? a.symdesc[(size_t)ch * a.S + gs] : 0ull;
    // (frame and index of the wave's first symbol by one scalar division, the lane's own by walking on from there)
    const uint32_t gs0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * per));
    int f = (int)(gs0 / (uint32_t)a.S), sidx = (int)(gs0 % (uint32_t)a.S) + (int)(gs - gs0);
    while (sidx >= a.S) { sidx -= a.S; f++; }
    const TxSymWhat w = txsym_what(a, gs, ch, d, f, sidx);
    int t[8]; int dr[8], pr[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int k = j + 8 * e;
        t[e] = (int)a.sctype[k]; dr[e] = a.data_rank[k]; pr[e] = a.pilot_rank[k];
    }
    const uint32_t pcount = (uint32_t)(w.s >= 3 ? w.s - 3 : 0) * (uint32_t)a.M_pilot;
    uint8_t bit[8], pil[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        bit[e] = w.bits[(!w.table && t[e] == 2) ? dr[e] : 0];
        pil[e] = a.pilot_seq[(pcount + (uint32_t)(t[e] == 1 ? pr[e] : 0)) % 255u];
    }
    // lane twiddles: W_64^(j + 8 e) (e < 4), W_32^(j + 8 e) (e < 2), W_16^j; across the lanes W_8^(j & 3), W_4^(j & 1)
    // (on the transcendental unit: nine twiddles per lane and wave are a third of the instructions otherwise; 1.2e-7 absolute)
    auto tw = [](int k, unsigned h) { float sn, cs; sincos_u32_hw((uint32_t)k * (uint32_t)(0x80000000u / h), sn, cs); return make_float2(cs, -sn); };
    float2 t32[4], t16[2];
#pragma unroll
    for (int e = 0; e < 4; e++) t32[e] = tw(j + 8 * e, 32u);
#pragma unroll
    for (int e = 0; e < 2; e++) t16[e] = tw(j + 8 * e, 16u);
    const float2 t8 = tw(j, 8u), t4 = tw(j & 3, 4u), t2 = tw(j & 1, 2u);
    const float2 *src = (w.s == 2) ? a.s1t : a.s0t;
    float2 x[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        float2 v = make_float2(0.f, 0.f);
        if (t[e] == 1) v = make_float2(pil[e] ? a.g_data : -a.g_data, 0.f);
        else if (t[e] == 2) { v = modulate(w.is_hdr ? 39 : a.mod, bit[e]); v.x *= a.g_data; v.y *= a.g_data; }
        x[e] = make_float2(v.x, -v.y);              // inverse FFT = conj(FFT(conj(X)))
    }
#pragma unroll
    for (int e = 0; e < 4; e++) { const float2 u = x[e], v = x[e + 4]; x[e] = cadd(u, v); x[e + 4] = cmul(csub(u, v), t32[e]); }
#pragma unroll
    for (int b = 0; b < 8; b += 4)
#pragma unroll
        for (int e = 0; e < 2; e++) { const float2 u = x[b + e], v = x[b + e + 2]; x[b + e] = cadd(u, v); x[b + e + 2] = cmul(csub(u, v), t16[e]); }
#pragma unroll
    for (int b = 0; b < 8; b += 2) { const float2 u = x[b], v = x[b + 1]; x[b] = cadd(u, v); x[b + 1] = cmul(csub(u, v), t8); }
    // (across the lanes: y = (p + sg x) tw with sg = -1, tw = W in the upper lane of a pair and sg = +1, tw = 1 in the lower one --
    //  one fma per component and one complex multiply instead of both results and a select; the last stage has no twiddle)
#pragma unroll
    for (int st = 0; st < 3; st++) {
        const int h = 4 >> st;
        const bool up = (j & h) != 0;
        const float sg = up ? -1.f : 1.f;
        const float2 twl = st == 0 ? t4 : t2;
        const float2 twu = up ? twl : make_float2(1.f, 0.f);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float2 p = make_float2(__shfl_xor(x[e].x, h, TXW), __shfl_xor(x[e].y, h, TXW));
            const float2 y = make_float2(fmaf(sg, x[e].x, p.x), fmaf(sg, x[e].y, p.y));
            x[e] = st == 2 ? y : cmul(y, twu);
        }
    }
    if (!live) return;
    // position j + 8 e holds X[8 bitrev3(j) + bitrev3(e)]
    const int n0 = 8 * (int)(__brev((unsigned)j) >> 29);
    float2 *dst = a.xsym + (a.xs_sym ? (size_t)ch * a.xs_ch + (size_t)gs * a.xs_sym : ((size_t)ch * a.frames * a.S + gs) * a.M) + xs_at(a.xs_grp, (uint32_t)n0);      // (n0 is a multiple of 8: one group)
    float2 o[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
        constexpr int br3[8] = { 0, 4, 2, 6, 1, 5, 3, 7 };
        o[m] = make_float2(x[br3[m]].x, -x[br3[m]].y);
    }
    if (w.table) {                                  // S0a, S0b, S1 bodies come from the tables; tail (and idle symbols) have none
#pragma unroll
        for (int m = 0; m < 8; m++) o[m] = w.zero ? make_float2(0.f, 0.f) : src[n0 + m];
    }
    float4 *d4 = reinterpret_cast<float4 *>(dst);
#pragma unroll
    for (int m = 0; m < 4; m++) d4[m] = make_float4(o[2 * m].x, o[2 * m].y, o[2 * m + 1].x, o[2 * m + 1].y);
}

// the same for a subcarrier count that is not a power of two (the reference applications default to M = 48):
// direct inverse DFT, X staged in LDS, exact integer phase (k n mod M) / M
__global__ __launch_bounds__(TXW) void txsym_dft_kernel(TxSymArgs a)
{
    __shared__ float2 X[1024];
    const int l = threadIdx.x & 63;
    const uint32_t gs = blockIdx.x, ch = blockIdx.y;
    const int f = gs / a.S, s = gs % a.S;
    float2 *dst = a.xsym + (a.xs_sym ? (size_t)ch * a.xs_ch + (size_t)gs * a.xs_sym : ((size_t)ch * a.frames * a.S + gs) * a.M);
    if (s < 3 || s == a.S - 1) {
        const float2 *src = (s == 2) ? a.s1t : a.s0t;
        for (int i = l; i < a.M; i += TXW) dst[xs_at(a.xs_grp, (uint32_t)i)] = (s == a.S - 1) ? make_float2(0.f, 0.f) : src[i];
        return;
    }
    const bool is_hdr = s < 3 + a.S_hdr;
    const uint8_t *bits = is_hdr ? a.hdr + ((size_t)ch * a.frames + f) * a.S_hdr * a.M_data + (size_t)(s - 3) * a.M_data
                                 : a.pay + ((size_t)ch * a.frames + f) * a.S_pay * a.M_data + (size_t)(s - 3 - a.S_hdr) * a.M_data;
    const uint32_t pcount = (uint32_t)(s - 3) * (uint32_t)a.M_pilot;
    for (int k = l; k < a.M; k += TXW) {
        float2 v = make_float2(0.f, 0.f);
        const int t = a.sctype[k];
        if (t == 1) v = make_float2(a.pilot_seq[(pcount + (uint32_t)a.pilot_rank[k]) % 255u] ? a.g_data : -a.g_data, 0.f);
        else if (t == 2) { v = modulate(is_hdr ? 39 : a.mod, bits[a.data_rank[k]]); v.x *= a.g_data; v.y *= a.g_data; }
        X[k] = v;
    }
    __syncthreads();
    const double stepd = 4294967296.0 / (double)a.M;                    // phase of (k n mod M) / M revolutions, to 2^-32
    for (int n = l; n < a.M; n += TXW) {
        float2 acc = make_float2(0.f, 0.f);
        uint32_t kn = 0;                                                // k n mod M
        for (int k = 0; k < a.M; k++) {
            float sn, cs; sincos_u32((uint32_t)((double)kn * stepd), sn, cs);
            acc.x += X[k].x * cs - X[k].y * sn;
            acc.y += X[k].x * sn + X[k].y * cs;
            kn += (uint32_t)n; if (kn >= (uint32_t)a.M) kn -= (uint32_t)a.M;
        }
        dst[xs_at(a.xs_grp, (uint32_t)n)] = acc;
    }
}

struct TxSynthArgs {
    int M, cp, taper, L, S, frames;
    const float *taperwin;      // [taper]
    const float2 *xsym;         // [ch][frames*S][M]; batch / ragged generators: [frames*S][ch][M] through the strides below
    // (the fused synthesis kernel reads 64 B of every channel per round: with the channel as the slow axis those are 512
    // pages a round, with the symbol as the slow axis one 32 KB span)
    size_t xs_ch = 0, xs_sym = 0;   // elements between channels / between a channel's consecutive symbols (0: the legacy layout)
    // Round 5: bodies in groups of 8 samples with the CHANNEL between the groups -- [symbol][M / 8][channel][8]: xs_ch = 8, xs_grp = 8 N,
    // xs_sym = N M; sample n of a body sits (n >> 3) xs_grp + (n & 7) behind its start.  A round of the fused synthesis kernel reads 64 B
    // = one group of every channel: with the channel's M samples contiguous (rounds 3-4) that was HALF of a 128-byte line per thread,
    // the other half being the same thread's next round -- and a line fetched for one half is fetched again for the other (FETCH_SIZE
    // 1.86 x the bytes used, profiles/r4_t5_traffic.json; profiles/r4_fetchcal.txt: a half-line read moves the line).  With the channel
    // between the groups the two halves of a line are two neighbouring threads' requests of the SAME round: whole lines, once.
    // 0 = the samples of a body are contiguous.
    size_t xs_grp = 0;
    size_t ks_ch = 0, ks_sym = 0;   // the same for symkind
    const float *taps;          // 26*K synthesis prototype
    float2 *v;                  // [nblocks][K] inverse-FFT outputs
    float2 *out;                // [nblocks][K] wideband samples
    uint32_t nblocks, N;
    uint32_t dtheta, first_sample_lo;
    float gain;
    // streaming (multichanneltx class semantics): one current frame per channel, anywhere on the block axis
    const long long *ft0;       // [N] absolute block index of the channel's frame start (NULL: batch layout above)
    const int *fS;              // [N] symbols of the channel's current frame (0: none yet)
    uint32_t xstride;           // symbols per channel slot in xsym
    long long b_first;          // absolute index of this launch's block 0
    int hist;                   // blocks of inverse-FFT history stored in front of v (0: cold start)
    // sharded synthesis (mctx_hip_synthesize_tiles): the bank's inputs come from exchanged channel-rate tiles
    const float2 *tiles = nullptr;  // [groups][ntiles][cg][8], channel = g*cg + c (NULL: frame_sample of xsym)
    uint32_t ntiles = 0, cg = 1;
    uint32_t out_first = 0;       // blocks in front of this one only feed the filter: out holds blocks >= out_first
    // ragged traffic: symkind[ch][S] (TXK_*), frames = 1, S = symbols of the whole axis
    const uint8_t *symkind = nullptr;
};

__device__ __forceinline__ float2 frame_sample_sym(const TxSynthArgs &a, uint32_t ch, const float2 *xb, int S, uint32_t nsym, uint32_t gs, uint32_t i);
// raised-cosine overlap of a symbol's first samples with the previous symbol's postfix: one fma shape wherever it is formed
// (frame_sample_sym and the fused synthesis kernel's aligned loader must agree bit for bit)
__device__ __forceinline__ float2 taper_blend(float2 v, float wa, float2 p, float wb)
{
    return make_float2(fmaf(v.x, wa, p.x * wb), fmaf(v.y, wa, p.y * wb));
}
// batch / ragged layout only (no streaming slots): the caller walks (gs, i) itself
__device__ __forceinline__ size_t xs_sym_of(const TxSynthArgs &a) { return a.xs_sym ? a.xs_sym : (size_t)a.M; }
// where sample n of a symbol body sits behind the body's start (TxSynthArgs::xs_grp)
__device__ __forceinline__ size_t xs_at(size_t xs_grp, uint32_t n) { return xs_grp ? (size_t)(n >> 3) * xs_grp + (size_t)(n & 7u) : (size_t)n; }
__device__ __forceinline__ const float2 *xs_channel(const TxSynthArgs &a, uint32_t ch)
{
    return a.xsym + (a.xs_sym ? (size_t)ch * a.xs_ch : (size_t)ch * a.frames * a.S * a.M);
}
__device__ __forceinline__ float2 frame_sample_at(const TxSynthArgs &a, uint32_t ch, uint32_t gs, uint32_t i)
{
    return frame_sample_sym(a, ch, xs_channel(a, ch), a.S, (uint32_t)(a.frames * a.S), gs, i);
}
// frame sample t of channel ch: cyclic prefix + raised-cosine overlap of consecutive symbols
// (liquid ofdmframegen_gensymbol / write_S0a / write_S0b / writetail)
__device__ __forceinline__ float2 frame_sample(const TxSynthArgs &a, uint32_t ch, uint32_t b)
{
    uint32_t t = b; int S = a.S; uint32_t nsym = (uint32_t)(a.frames * a.S);
    const float2 *xb = xs_channel(a, ch);
    if (a.ft0) {                                    // streaming: position inside the channel's current frame
        const long long rel = a.b_first + (long long)b - a.ft0[ch];
        S = a.fS[ch]; nsym = (uint32_t)S;
        if (rel < 0 || S == 0 || rel >= (long long)S * a.L) return make_float2(0.f, 0.f);
        t = (uint32_t)rel;
        xb = a.xsym + (size_t)ch * a.xstride * a.M;
    }
    const uint32_t gs = t / (uint32_t)a.L, i = t % (uint32_t)a.L;
    return frame_sample_sym(a, ch, xb, S, nsym, gs, i);
}
// ... of symbol gs (counted along the channel's frame axis), position i in it
__device__ __forceinline__ float2 frame_sample_sym(const TxSynthArgs &a, uint32_t ch, const float2 *xb, int S, uint32_t nsym, uint32_t gs, uint32_t i)
{
    if (gs >= nsym) return make_float2(0.f, 0.f);
    int s = (int)(gs % (uint32_t)S);
    if (a.symkind) {                                // ragged traffic: the symbol's role comes from the map
        const int kind = a.symkind[a.ks_sym ? (size_t)ch * a.ks_ch + (size_t)gs * a.ks_sym : (size_t)ch * nsym + gs];
        if (kind == TXK_IDLE) return make_float2(0.f, 0.f);
        s = kind == TXK_S0A ? 0 : (kind == TXK_S0B ? 1 : (kind == TXK_TAIL ? S - 1 : 2));
    }
    const size_t xs = xs_sym_of(a);
    const float2 *x = xb + (size_t)gs * xs;
    const int M = a.M, cp = a.cp;
    if (s == 0) {                                   // S0a: shifted copy, ramp up only
        float2 v = x[xs_at(a.xs_grp, (i + M - 2 * cp) % M)];
        if ((int)i < a.taper) { v.x *= a.taperwin[i]; v.y *= a.taperwin[i]; }
        return v;
    }
    if (s == 1) return x[xs_at(a.xs_grp, (i + M - cp) % M)];         // S0b: plain cyclic extension
    if (s == S - 1) {                               // tail: previous symbol's postfix ramping down
        if ((int)i >= a.taper) return make_float2(0.f, 0.f);
        const float2 p = (x - xs)[xs_at(a.xs_grp, i)]; const float b = a.taperwin[a.taper - 1 - i];
        return make_float2(p.x * b, p.y * b);
    }
    float2 v = x[xs_at(a.xs_grp, (i + M - cp) % M)];
    if ((int)i < a.taper) {
        const float2 p = (x - xs)[xs_at(a.xs_grp, i)];                // first samples of the previous symbol body (S0b: s0)
        const float wa = a.taperwin[i], wb = a.taperwin[a.taper - 1 - i];
        v = taper_blend(v, wa, p, wb);
    }
    return v;
}

// K-point inverse FFT of one block in LDS (radix-2 Stockham), bins >= N are zero
template <int K>
__global__ void txifft_kernel(TxSynthArgs a)
{
    constexpr int T = (K / 2 < 64) ? 64 : K / 2;
    __shared__ float2 buf[2][K];
    const uint32_t b = blockIdx.x;
    const int tid = threadIdx.x;
    if (a.tiles) {
        for (int k = tid; k < K; k += T)
            buf[0][k] = (k < (int)a.N) ? a.tiles[(((size_t)((uint32_t)k / a.cg) * a.ntiles + (b >> 3)) * a.cg + (uint32_t)k % a.cg) * 8 + (b & 7)]
                                       : make_float2(0.f, 0.f);
    } else {
        for (int k = tid; k < K; k += T) buf[0][k] = (k < (int)a.N) ? frame_sample(a, (uint32_t)k, b) : make_float2(0.f, 0.f);
    }
    __syncthreads();
    int cur = 0;
    // Stockham autosort, decimation in frequency: stage with n = current sub-length, s = stride
    for (int n = K, s = 1; n > 1; n >>= 1, s <<= 1) {
        const int m = n >> 1;
        for (int q = tid; q < K / 2; q += T) {
            const int p = q / s, r = q % s;         // butterfly p of sub-length n, interleave slot r
            float sn, cs; sincos_u32((uint32_t)p * (uint32_t)(4294967296.0 / n), sn, cs);
            const float2 w = make_float2(cs, sn);   // e^{+j 2 pi p / n}: inverse transform
            const float2 u = buf[cur][r + s * p], v = buf[cur][r + s * (p + m)];
            buf[cur ^ 1][r + s * 2 * p] = cadd(u, v);
            buf[cur ^ 1][r + s * (2 * p + 1)] = cmul(csub(u, v), w);
        }
        __syncthreads();
        cur ^= 1;
    }
    float2 *dst = a.v + (size_t)b * K;
    for (int k = tid; k < K; k += T) dst[k] = buf[cur][k];
}

// one channel's frame as channel-rate samples (ofdmflexframegen_writesymbol output, symbol after symbol)
__global__ void txframe_kernel(TxSynthArgs a, uint32_t nsamples)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nsamples) return;
    const float2 v = frame_sample(a, 0, t);
    a.out[t] = make_float2(v.x * a.gain, v.y * a.gain);
}

// synthesis FIR down the time axis + NCO mix-up + gain; a thread owns one column for 8 blocks
__global__ void txfir_kernel(TxSynthArgs a, uint32_t K)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    const long long b0 = (long long)blockIdx.y * 8;
    float h[TX_P];
#pragma unroll
    for (int j = 0; j < TX_P; j++) h[j] = a.taps[i + (uint32_t)j * K];
    float2 w[TX_P + 7];                             // w[q] = v[b0 - 25 + q][i]
#pragma unroll
    for (int q = 0; q < TX_P + 7; q++) {
        const long long b = b0 - (TX_P - 1) + q;
        w[q] = (b >= -(long long)a.hist && b < (long long)a.nblocks) ? a.v[b * (long long)K + (long long)i] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const long long b = b0 + r;
        if (b >= (long long)a.nblocks) break;
        if (b < (long long)a.out_first) continue;
        float2 acc = make_float2(0.f, 0.f);
#pragma unroll
        for (int j = TX_P - 1; j >= 0; j--) {       // oldest first, like the window dot product
            acc.x += h[j] * w[TX_P - 1 + r - j].x;
            acc.y += h[j] * w[TX_P - 1 + r - j].y;
        }
        const uint32_t t = a.first_sample_lo + (uint32_t)((unsigned long long)b * K + i);
        float2 y = mix_up(acc, t * a.dtheta);
        a.out[(size_t)(b - (long long)a.out_first) * K + i] = make_float2(y.x * a.gain, y.y * a.gain);
    }
}

// channel-rate samples of a channel shard for blocks [first_block, first_block + 8 ntiles) as granules
// tiles[tile][c][8] (zeros outside the traffic): what one rank contributes to another rank's time slab
__global__ void txtiles_kernel(TxSynthArgs a, long long first_block, uint32_t ntiles, uint32_t nch, float2 *tiles)
{
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= ntiles * nch) return;
    const uint32_t tile = id / nch, c = id % nch;
    float2 v[8];
    float4 *dst = reinterpret_cast<float4 *>(tiles + (size_t)id * 8);
    const long long b0 = first_block + (long long)tile * 8;
    // A granule that never straddles an OFDM symbol (8 | L, cp, M and first_block; no role map): 64 aligned bytes of one symbol
    // body as they stand -- txsym zero-fills tail bodies, S0a reads two prefixes back -- and, in a symbol's first granule, the
    // raised-cosine overlap with the previous body in frame_sample_sym's fma shape (S0b has none).  Same values, an eighth of
    // the address arithmetic, whole cache lines.
    if (!a.symkind && !a.ft0 && (a.L % 8) == 0 && (a.cp % 8) == 0 && (a.M % 8) == 0 && a.taper >= 0 && a.taper <= 4 && (first_block % 8) == 0 &&
        b0 >= 0 && b0 + 7 < 0xffffffffll) {
        const uint32_t nsym = (uint32_t)(a.frames * a.S);
        const uint32_t gs = (uint32_t)b0 / (uint32_t)a.L, i = (uint32_t)b0 % (uint32_t)a.L;
        if (gs >= nsym) {
#pragma unroll
            for (int t = 0; t < 4; t++) dst[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            return;
        }
        const int sidx = (int)(gs % (uint32_t)a.S);
        const size_t xs = xs_sym_of(a);
        const float2 *x = xs_channel(a, c) + (size_t)gs * xs;
        const uint32_t base = (i + (uint32_t)a.M - (sidx == 0 ? 2u : 1u) * (uint32_t)a.cp) % (uint32_t)a.M;
        const float4 *xp = reinterpret_cast<const float4 *>(x + xs_at(a.xs_grp, base));      // (base is a multiple of 8: one group, 64 contiguous bytes in either layout)
        float4 q4[4];
#pragma unroll
        for (int t = 0; t < 4; t++) q4[t] = xp[t];
        if (i == 0 && sidx != 1 && a.taper > 0) {
            const float4 *pp = reinterpret_cast<const float4 *>(gs > 0 ? x - xs : x);
            const float4 p0 = pp[0], p1 = pp[1];
            const float2 pv[4] = { make_float2(p0.x, p0.y), make_float2(p0.z, p0.w), make_float2(p1.x, p1.y), make_float2(p1.z, p1.w) };
            float2 xv[4] = { make_float2(q4[0].x, q4[0].y), make_float2(q4[0].z, q4[0].w), make_float2(q4[1].x, q4[1].y), make_float2(q4[1].z, q4[1].w) };
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (r < a.taper) xv[r] = taper_blend(xv[r], a.taperwin[r], pv[r], gs > 0 ? a.taperwin[a.taper - 1 - r] : 0.f);
            q4[0] = make_float4(xv[0].x, xv[0].y, xv[1].x, xv[1].y); q4[1] = make_float4(xv[2].x, xv[2].y, xv[3].x, xv[3].y);
        }
#pragma unroll
        for (int t = 0; t < 4; t++) dst[t] = q4[t];
        return;
    }
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const long long b = b0 + t;
        v[t] = (b >= 0 && b < 0xffffffffll) ? frame_sample(a, c, (uint32_t)b) : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int t = 0; t < 4; t++) dst[t] = make_float4(v[2 * t].x, v[2 * t].y, v[2 * t + 1].x, v[2 * t + 1].y);
}

}  // namespace mcrx

#include "synth_tile.hpp"       // the fused synthesis bank (needs TxSynthArgs / frame_sample above)

using namespace mcrx;

static thread_local std::string g_tx_err;
#define TXCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_tx_err = std::string(#x) + ": " + hipGetErrorString(e_); return MCRX_EHIP; } } while (0)

struct mctx_hip_s {
    int device = -1;            // the HIP device the handle was created on: every entry point runs with it current (devscope.hpp)
    unsigned N, K, M, cp, taper, ncu = 256;
    bool taps_symmetric = false;                        // h[i] == h[pK - i] bit for bit (the fused synthesis kernel keeps half of it)
    OfdmDesign od;
    std::vector<float> taps;
    uint32_t dtheta;
    std::vector<void *> owned;
    const uint8_t *d_sctype, *d_pseq; const int16_t *d_drank, *d_prank; const float2 *d_s0t, *d_s1t;
    const float *d_taper, *d_taps;
    // ---- streaming state (mctx_hip_stream_*): the multichanneltx class semantics, one symbol period per launch
    bool st_on = false;
    unsigned st_maxpay = 0, st_Sh = 0, st_Spmax = 0, st_Smax = 0;
    uint8_t *d_shdr = nullptr, *d_spay = nullptr;       // [N][Sh*Md], [N][Spmax*Md] modem symbols of each channel's current frame
    float2 *d_sxsym = nullptr;                          // [N][Smax][M] its time-domain symbol bodies
    long long *d_ft0 = nullptr; int *d_fS = nullptr;    // [N] frame start block, symbols per frame
    float2 *d_sv[2] = { nullptr, nullptr };             // [25 + L][K] inverse-FFT outputs: history, then the period
    float2 *d_sout = nullptr, *h_sout = nullptr;        // [L][K] the period's wideband samples (device, pinned host)
    int sv_cur = 0;
    std::vector<long long> ft0; std::vector<int> fS, fS_new; std::vector<uint8_t> assembled, pending;
    long long period = 0, blocks_out = 0; unsigned out_pos = 0;
    hipStream_t sst = nullptr;
    float2 *d_synv = nullptr; size_t syn_cap = 0;       // sharded synthesis: inverse-FFT outputs of one slab (+ lead)
    template <class T> int up(const T **dst, const T *src, size_t n)
    {
        T *p = nullptr;
        TXCHK(hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T)));
        if (n) TXCHK(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
        owned.push_back(p); *dst = p;
        return MCRX_OK;
    }
};

extern "C" const char *mctx_hip_last_error(void) { return g_tx_err.c_str(); }

extern "C" int mctx_hip_create(mctx_hip_t *out, unsigned N, unsigned M, unsigned cp, unsigned taper, const unsigned char *p)
{
    if (!out) return MCRX_EINVAL;
    *out = nullptr;
    // argument checks of multichanneltx::multichanneltx (lib/multichanneltx.cc:48-60)
    if (N < 1) { g_tx_err = "error: multichanneltx, must have at least one channel"; return MCRX_EINVAL; }
    if (M < 8) { g_tx_err = "error: multichanneltx, number of subcarriers must be at least 8"; return MCRX_EINVAL; }
    if (cp < 1) { g_tx_err = "error: multichanneltx, cyclic prefix length must be at least 1"; return MCRX_EINVAL; }
    if (taper > cp) { g_tx_err = "error: multichanneltx, taper length cannot exceed cyclic prefix length"; return MCRX_EINVAL; }
    const unsigned K = 2 * N;
    if ((K & (K - 1)) || K > 1024) { g_tx_err = "2N must be a power of two <= 1024"; return MCRX_EUNSUPP; }
    if (M > 1024) { g_tx_err = "at most 1024 subcarriers"; return MCRX_EUNSUPP; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_tx_err = "no HIP device (no CPU fallback)"; return MCRX_EHIP; }
    mctx_hip_t q = new mctx_hip_s();
    q->device = current_device();
    q->N = N; q->K = K; q->M = M; q->cp = cp; q->taper = taper;
    if (q->od.init(M, cp, taper, p) != 0) { delete q; g_tx_err = "invalid subcarrier allocation"; return MCRX_EINVAL; }
    q->taps = pfb_prototype(K, 13, 60.0f);
    q->dtheta = channel_center_step(N);
    {   // the windowed-sinc prototype is evaluated at +-t: h[i] == h[pK - i]; the fused kernel relies on it bit for bit
        const size_t n = q->taps.size();
        bool sym = n == (size_t)26 * K;
        for (size_t i = 1; sym && i < n; i++) sym = memcmp(&q->taps[i], &q->taps[n - i], sizeof(float)) == 0;
        q->taps_symmetric = sym;
        int dev = 0, ncu = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0) q->ncu = (unsigned)ncu;
    }
    std::vector<int16_t> dr(M), pr(M);
    for (unsigned i = 0; i < M; i++) { dr[i] = (int16_t)q->od.data_rank[i]; pr[i] = (int16_t)q->od.pilot_rank[i]; }
    int rc;
    if ((rc = q->up(&q->d_sctype, q->od.p.data(), M)) || (rc = q->up(&q->d_pseq, q->od.pilot_seq, 255)) ||
        (rc = q->up(&q->d_drank, dr.data(), M)) || (rc = q->up(&q->d_prank, pr.data(), M)) ||
        (rc = q->up(&q->d_s0t, reinterpret_cast<const float2 *>(q->od.s0.data()), M)) ||
        (rc = q->up(&q->d_s1t, reinterpret_cast<const float2 *>(q->od.s1.data()), M)) ||
        (rc = q->up(&q->d_taper, q->od.taperwin.data(), q->od.taperwin.size())) ||
        (rc = q->up(&q->d_taps, q->taps.data(), q->taps.size()))) { mctx_hip_destroy(q); return rc; }
    *out = q;
    return MCRX_OK;
}

extern "C" int mctx_hip_destroy(mctx_hip_t q)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q) return MCRX_OK;
    (void)hipDeviceSynchronize();
    for (void *p : q->owned) (void)hipFree(p);
    for (void *p : { (void *)q->d_shdr, (void *)q->d_spay, (void *)q->d_sxsym, (void *)q->d_ft0, (void *)q->d_fS,
                     (void *)q->d_sv[0], (void *)q->d_sv[1], (void *)q->d_sout, (void *)q->d_synv }) if (p) (void)hipFree(p);
    if (q->h_sout) (void)hipHostFree(q->h_sout);
    if (q->sst) (void)hipStreamDestroy(q->sst);
    delete q;
    return MCRX_OK;
}

static int tx_launch_sym(mctx_hip_t q, const TxSymArgs &sa, unsigned nsym, unsigned nch, hipStream_t st);
static int tx_launch_ifft(mctx_hip_t q, const TxSynthArgs &ya, unsigned nblocks, hipStream_t st);
static int tx_synthesize(mctx_hip_t q, const TxSynthArgs &ya, hipStream_t st);
static bool tx_fused_ok(mctx_hip_t q, const TxSynthArgs &ya);

static void frame_geometry(mctx_hip_t q, unsigned payload_len, int mod, int fec0, int fec1,
                           unsigned &S_hdr, unsigned &S_pay, unsigned &S)
{
    const unsigned Md = q->od.M_data, bps = mod_bps(mod);
    const unsigned nb = 8 * packet_enc_len(payload_len, CRC_32, fec0, fec1);
    const unsigned mod_len = nb / bps + ((nb % bps) ? 1 : 0);
    S_hdr = (288 + Md - 1) / Md; S_pay = (mod_len + Md - 1) / Md; S = 3 + S_hdr + S_pay + 1;
}

extern "C" size_t mctx_hip_blocks_for(mctx_hip_t q, unsigned frames_per_channel, unsigned payload_len, int mod, int fec0, int fec1)
{
    if (!q || !mod_bps(mod) || !fec_supported(fec0) || !fec_supported(fec1)) return 0;
    unsigned Sh, Sp, S; frame_geometry(q, payload_len, mod, fec0, fec1, Sh, Sp, S);
    size_t nb = (size_t)frames_per_channel * S * (q->M + q->cp) + 64;     // + idle tail for the filter to ring out
    return (nb + 15) / 16 * 16;     // whole receiver tiles (MCRX_TILE blocks), so that the stream can be pushed as it is
}

extern "C" int mctx_hip_generate(mctx_hip_t q, void *d_iq, size_t nblocks, unsigned frames, unsigned payload_len,
                                 int mod, int fec0, int fec1, float gain, uint32_t seed,
                                 uint8_t *hdr_out, uint8_t *pay_out, void *stream)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q || !d_iq || !mod_bps(mod)) { g_tx_err = "bad argument"; return MCRX_EINVAL; }
    if (!fec_supported(fec0) || !fec_supported(fec1)) { g_tx_err = "unsupported fec scheme"; return MCRX_EUNSUPP; }
    hipStream_t st = (hipStream_t)stream;
    unsigned Sh, Sp, S; frame_geometry(q, payload_len, mod, fec0, fec1, Sh, Sp, S);
    const unsigned Md = q->od.M_data, N = q->N, M = q->M, K = q->K;
    // ---- host: traffic + bit-level assembly (src/multichannel_tx.cc:166-190 with a seeded generator)
    std::vector<uint8_t> hdr((size_t)N * frames * Sh * Md), pay((size_t)N * frames * Sp * Md);
    FrameSymbols fsym;
    for (unsigned ch = 0; ch < N; ch++) {
        std::mt19937 rng(seed + ch);
        for (unsigned f = 0; f < frames; f++) {
            uint8_t h8[8] = { (uint8_t)(f >> 8), (uint8_t)f, (uint8_t)ch, 0, 0, 0, 0, 0 };
            for (int i = 3; i < 8; i++) h8[i] = (uint8_t)(rng() & 0xff);
            std::vector<uint8_t> pl(payload_len);
            for (auto &b : pl) b = (uint8_t)(rng() & 0xff);
            assemble_frame(h8, pl, mod, fec0, fec1, Md, Sh, Sp, fsym);
            memcpy(&hdr[((size_t)ch * frames + f) * Sh * Md], fsym.hdr.data(), fsym.hdr.size());
            memcpy(&pay[((size_t)ch * frames + f) * Sp * Md], fsym.pay.data(), fsym.pay.size());
            if (hdr_out) memcpy(hdr_out + ((size_t)ch * frames + f) * 8, h8, 8);
            if (pay_out && payload_len) memcpy(pay_out + ((size_t)ch * frames + f) * payload_len, pl.data(), payload_len);
        }
    }
    uint8_t *d_hdr = nullptr, *d_pay = nullptr; float2 *d_xsym = nullptr, *d_v = nullptr;
    const size_t nsym = (size_t)frames * S;
    TXCHK(hipMalloc((void **)&d_hdr, hdr.size())); TXCHK(hipMalloc((void **)&d_pay, std::max<size_t>(pay.size(), 1)));
    TXCHK(hipMalloc((void **)&d_xsym, (size_t)N * nsym * M * sizeof(float2)));
    { TxSynthArgs probe; probe.ft0 = nullptr; probe.hist = 0; probe.out_first = 0;
      if (!tx_fused_ok(q, probe)) TXCHK(hipMalloc((void **)&d_v, nblocks * K * sizeof(float2))); }      // (inverse-FFT outputs of the two-kernel path)
    TXCHK(hipMemcpyAsync(d_hdr, hdr.data(), hdr.size(), hipMemcpyHostToDevice, st));
    TXCHK(hipMemcpyAsync(d_pay, pay.data(), pay.size(), hipMemcpyHostToDevice, st));
    TxSymArgs sa;
    sa.M = (int)M; sa.log2M = 0; while ((1u << sa.log2M) < M) sa.log2M++;
    sa.cp = (int)q->cp; sa.taper = (int)q->taper; sa.L = (int)(M + q->cp); sa.M_pilot = (int)q->od.M_pilot; sa.M_data = (int)Md;
    sa.S = (int)S; sa.S_hdr = (int)Sh; sa.S_pay = (int)Sp; sa.frames = (int)frames; sa.bps = (int)mod_bps(mod); sa.mod = mod;
    sa.g_data = 1.0f / sqrtf((float)(q->od.M_pilot + q->od.M_data));
    sa.sctype = q->d_sctype; sa.data_rank = q->d_drank; sa.pilot_rank = q->d_prank; sa.pilot_seq = q->d_pseq;
    sa.s0t = q->d_s0t; sa.s1t = q->d_s1t; sa.hdr = d_hdr; sa.pay = d_pay; sa.xsym = d_xsym; sa.nch = N; sa.xs_ch = M; sa.xs_sym = (size_t)N * M;   // symbol-major: see TxSynthArgs
    if (M % 8 == 0) { sa.xs_ch = 8; sa.xs_grp = (size_t)8 * N; }                                   // ... in groups of 8 samples with the channel between them
    { int rc = tx_launch_sym(q, sa, (unsigned)nsym, N, st); if (rc) return rc; }
    TxSynthArgs ya;
    ya.M = (int)M; ya.cp = (int)q->cp; ya.taper = (int)q->taper; ya.L = (int)(M + q->cp); ya.S = (int)S; ya.frames = (int)frames;
    ya.taperwin = q->d_taper; ya.xsym = d_xsym; ya.xs_ch = M; ya.xs_sym = (size_t)N * M; ya.taps = q->d_taps; ya.v = d_v; ya.out = (float2 *)d_iq;
    ya.xs_ch = sa.xs_ch; ya.xs_grp = sa.xs_grp;
    ya.nblocks = (uint32_t)nblocks; ya.N = N; ya.dtheta = q->dtheta; ya.first_sample_lo = 0; ya.gain = gain;
    ya.ft0 = nullptr; ya.fS = nullptr; ya.xstride = 0; ya.b_first = 0; ya.hist = 0;
    { int rc = tx_synthesize(q, ya, st); if (rc) return rc; }
    TXCHK(hipStreamSynchronize(st));
    (void)hipFree(d_hdr); (void)hipFree(d_pay); (void)hipFree(d_xsym); (void)hipFree(d_v);
    return MCRX_OK;
}


// Ragged traffic: what src/multichannel_txrx.cc:227-267 puts on the air -- every packet its own length
// (`rand() % payload_len` there; uniform in [len_lo, len_hi] here), handed to whichever channel is free, so a channel's
// frames follow each other after short, irregular pauses, with a long silence now and then (the bursts).  Frames start on
// OFDM symbol boundaries (the class steps all frame generators one symbol at a time, lib/multichanneltx.cc:230-242).
// Per channel (seeded): gap of 0 .. gap_max symbols before every frame, with probability 1 / long_every a silence of
// 16 .. 16 + long_max symbols instead; frames are placed until the next one would not end 64 blocks before the end of
// the stream.  count[ch] frames were placed; frame f of channel ch: hdr[ch][f][8], len[ch][f], pay[ch][f][len_hi],
// start[ch][f] = block index of its first sample (host arrays sized for max_frames per channel; may be NULL).
extern "C" int mctx_hip_generate_ragged(mctx_hip_t q, void *d_iq, size_t nblocks, unsigned max_frames, unsigned len_lo, unsigned len_hi,
                                        unsigned gap_max, unsigned long_every, unsigned long_max, int mod, int fec0, int fec1,
                                        float gain, uint32_t seed, uint32_t *count_out, uint8_t *hdr_out, uint32_t *len_out,
                                        uint8_t *pay_out, uint64_t *start_out, void *stream)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q || !d_iq || !mod_bps(mod) || len_hi < len_lo || !max_frames) { g_tx_err = "bad argument"; return MCRX_EINVAL; }
    if (!fec_supported(fec0) || !fec_supported(fec1)) { g_tx_err = "unsupported fec scheme"; return MCRX_EUNSUPP; }
    if (q->M & (q->M - 1)) { g_tx_err = "ragged traffic needs a power-of-two subcarrier count"; return MCRX_EUNSUPP; }
    hipStream_t st = (hipStream_t)stream;
    const unsigned Md = q->od.M_data, N = q->N, M = q->M, K = q->K, L = M + q->cp;
    const size_t T = nblocks / L;                                   // symbols on a channel's axis
    if (T < 8) { g_tx_err = "stream too short"; return MCRX_EINVAL; }
    std::vector<unsigned long long> desc((size_t)N * T, 0ull);
    std::vector<uint8_t> kind((size_t)N * T, (uint8_t)TXK_IDLE);
    std::vector<uint8_t> hdr, pay;
    FrameSymbols fsym;
    for (unsigned ch = 0; ch < N; ch++) {
        std::mt19937 rng(seed + 7919u * ch);
        size_t t = 0; unsigned f = 0;
        while (f < max_frames) {
            const unsigned plen = len_lo + (unsigned)(rng() % (len_hi - len_lo + 1));
            unsigned gap = gap_max ? (unsigned)(rng() % (gap_max + 1)) : 0;
            if (long_every && (rng() % long_every) == 0) gap = 16 + (long_max ? (unsigned)(rng() % (long_max + 1)) : 0);
            unsigned Sh, Sp, S; frame_geometry(q, plen, mod, fec0, fec1, Sh, Sp, S);
            if ((t + gap + S) * L + 64 > nblocks) break;
            t += gap;
            uint8_t h8[8] = { (uint8_t)(f >> 8), (uint8_t)f, (uint8_t)ch, 0, 0, 0, 0, 0 };
            for (int i = 3; i < 8; i++) h8[i] = (uint8_t)(rng() & 0xff);
            std::vector<uint8_t> pl(plen);
            for (auto &b : pl) b = (uint8_t)(rng() & 0xff);
            assemble_frame(h8, pl, mod, fec0, fec1, Md, Sh, Sp, fsym);
            const size_t hrow = hdr.size() / Md, prow = pay.size() / Md;
            hdr.insert(hdr.end(), fsym.hdr.begin(), fsym.hdr.end());
            pay.insert(pay.end(), fsym.pay.begin(), fsym.pay.end());
            for (unsigned s = 0; s < S; s++) {
                int k; unsigned long long row = 0;
                if (s == 0) k = TXK_S0A; else if (s == 1) k = TXK_S0B; else if (s == 2) k = TXK_S1;
                else if (s == S - 1) k = TXK_TAIL;
                else if (s < 3 + Sh) { k = TXK_HDR; row = hrow + (s - 3); }
                else { k = TXK_PAY; row = prow + (s - 3 - Sh); }
                kind[(size_t)(t + s) * N + ch] = (uint8_t)k;                    // [symbol][channel], like the symbol bodies
                desc[(size_t)ch * T + t + s] = (unsigned long long)k | ((unsigned long long)s << 8) | (row << 32);
            }
            if (hdr_out) memcpy(hdr_out + ((size_t)ch * max_frames + f) * 8, h8, 8);
            if (len_out) len_out[(size_t)ch * max_frames + f] = plen;
            if (pay_out && plen) memcpy(pay_out + ((size_t)ch * max_frames + f) * len_hi, pl.data(), plen);
            if (start_out) start_out[(size_t)ch * max_frames + f] = (uint64_t)t * L;
            t += S; f++;
        }
        if (count_out) count_out[ch] = f;
    }
    uint8_t *d_hdr = nullptr, *d_pay = nullptr, *d_kind = nullptr; unsigned long long *d_desc = nullptr; float2 *d_xsym = nullptr, *d_v = nullptr;
    TXCHK(hipMalloc((void **)&d_hdr, std::max<size_t>(hdr.size(), 1))); TXCHK(hipMalloc((void **)&d_pay, std::max<size_t>(pay.size(), 1)));
    TXCHK(hipMalloc((void **)&d_kind, kind.size())); TXCHK(hipMalloc((void **)&d_desc, desc.size() * sizeof(unsigned long long)));
    TXCHK(hipMalloc((void **)&d_xsym, (size_t)N * T * M * sizeof(float2)));
    { TxSynthArgs probe; probe.ft0 = nullptr; probe.hist = 0; probe.out_first = 0;
      if (!tx_fused_ok(q, probe)) TXCHK(hipMalloc((void **)&d_v, nblocks * K * sizeof(float2))); }
    TXCHK(hipMemcpyAsync(d_hdr, hdr.data(), hdr.size(), hipMemcpyHostToDevice, st));
    TXCHK(hipMemcpyAsync(d_pay, pay.data(), pay.size(), hipMemcpyHostToDevice, st));
    TXCHK(hipMemcpyAsync(d_kind, kind.data(), kind.size(), hipMemcpyHostToDevice, st));
    TXCHK(hipMemcpyAsync(d_desc, desc.data(), desc.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
    TxSymArgs sa;
    sa.M = (int)M; sa.log2M = 0; while ((1u << sa.log2M) < M) sa.log2M++;
    sa.cp = (int)q->cp; sa.taper = (int)q->taper; sa.L = (int)L; sa.M_pilot = (int)q->od.M_pilot; sa.M_data = (int)Md;
    sa.S = (int)T; sa.S_hdr = 0; sa.S_pay = 0; sa.frames = 1; sa.bps = (int)mod_bps(mod); sa.mod = mod;
    sa.g_data = 1.0f / sqrtf((float)(q->od.M_pilot + q->od.M_data));
    sa.sctype = q->d_sctype; sa.data_rank = q->d_drank; sa.pilot_rank = q->d_prank; sa.pilot_seq = q->d_pseq;
    sa.s0t = q->d_s0t; sa.s1t = q->d_s1t; sa.hdr = d_hdr; sa.pay = d_pay; sa.xsym = d_xsym; sa.nch = N; sa.symdesc = d_desc; sa.xs_ch = M; sa.xs_sym = (size_t)N * M;
    if (M % 8 == 0) { sa.xs_ch = 8; sa.xs_grp = (size_t)8 * N; }
    { int rc = tx_launch_sym(q, sa, (unsigned)T, N, st); if (rc) return rc; }
    TxSynthArgs ya;
    ya.M = (int)M; ya.cp = (int)q->cp; ya.taper = (int)q->taper; ya.L = (int)L; ya.S = (int)T; ya.frames = 1;
    ya.taperwin = q->d_taper; ya.xsym = d_xsym; ya.xs_ch = M; ya.xs_sym = (size_t)N * M; ya.ks_ch = 1; ya.ks_sym = N; ya.taps = q->d_taps; ya.v = d_v; ya.out = (float2 *)d_iq;
    ya.xs_ch = sa.xs_ch; ya.xs_grp = sa.xs_grp;
    ya.nblocks = (uint32_t)nblocks; ya.N = N; ya.dtheta = q->dtheta; ya.first_sample_lo = 0; ya.gain = gain;
    ya.ft0 = nullptr; ya.fS = nullptr; ya.xstride = 0; ya.b_first = 0; ya.hist = 0; ya.symkind = d_kind;
    { int rc = tx_synthesize(q, ya, st); if (rc) return rc; }
    TXCHK(hipStreamSynchronize(st));
    (void)hipFree(d_hdr); (void)hipFree(d_pay); (void)hipFree(d_kind); (void)hipFree(d_desc); (void)hipFree(d_xsym); (void)hipFree(d_v);
    return MCRX_OK;
}

// ------------------------------------------------------------------------------------------------
// Sharded form of the generator (multi-GPU transmit side of src/multichannel_txrx.cc): the frame generators are
// independent per channel, the synthesis bank couples all channels inside one block but is independent across
// blocks up to its 25 blocks of filter memory -- the mirror image of the receiver.  So a rank
//   1. traffic_create   assembles and modulates the frames of ITS channel shard (same seeds -> same frames as
//                       mctx_hip_generate makes for those channels),
//   2. traffic_tiles    writes their channel-rate samples for another rank's time slab as granules [tile][c][8],
//   (all-to-all: channel shards -> time shards)
//   3. synthesize_tiles runs the 2N-point inverse FFT, synthesis FIR and NCO over ITS time slab from the
//                       received granules [source rank][tile][c][8]; `lead` blocks in front only fill the filter.
struct mctx_hip_traffic_s {
    int device = -1;
    mctx_hip_t q;
    unsigned ch_first, ch_count, frames, S;
    float2 *d_xsym;
};

extern "C" int mctx_hip_traffic_create(mctx_hip_t q, mctx_hip_traffic_t *out, unsigned ch_first, unsigned ch_count,
                                       unsigned frames, unsigned payload_len, int mod, int fec0, int fec1, uint32_t seed,
                                       uint8_t *hdr_out, uint8_t *pay_out, void *stream)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q || !out || !mod_bps(mod)) { g_tx_err = "bad argument"; return MCRX_EINVAL; }
    if (!fec_supported(fec0) || !fec_supported(fec1)) { g_tx_err = "unsupported fec scheme"; return MCRX_EUNSUPP; }
    *out = nullptr;
    if (!ch_count || ch_first + ch_count > q->N || !frames) { g_tx_err = "channel shard outside the transmitter"; return MCRX_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    unsigned Sh, Sp, S; frame_geometry(q, payload_len, mod, fec0, fec1, Sh, Sp, S);
    const unsigned Md = q->od.M_data, M = q->M;
    std::vector<uint8_t> hdr((size_t)ch_count * frames * Sh * Md), pay((size_t)ch_count * frames * Sp * Md);
    FrameSymbols fsym;
    for (unsigned c = 0; c < ch_count; c++) {
        const unsigned ch = ch_first + c;
        std::mt19937 rng(seed + ch);                                    // the recipe of mctx_hip_generate, channel by channel
        for (unsigned f = 0; f < frames; f++) {
            uint8_t h8[8] = { (uint8_t)(f >> 8), (uint8_t)f, (uint8_t)ch, 0, 0, 0, 0, 0 };
            for (int i = 3; i < 8; i++) h8[i] = (uint8_t)(rng() & 0xff);
            std::vector<uint8_t> pl(payload_len);
            for (auto &b : pl) b = (uint8_t)(rng() & 0xff);
            assemble_frame(h8, pl, mod, fec0, fec1, Md, Sh, Sp, fsym);
            memcpy(&hdr[((size_t)c * frames + f) * Sh * Md], fsym.hdr.data(), fsym.hdr.size());
            memcpy(&pay[((size_t)c * frames + f) * Sp * Md], fsym.pay.data(), fsym.pay.size());
            if (hdr_out) memcpy(hdr_out + ((size_t)c * frames + f) * 8, h8, 8);
            if (pay_out && payload_len) memcpy(pay_out + ((size_t)c * frames + f) * payload_len, pl.data(), payload_len);
        }
    }
    uint8_t *d_hdr = nullptr, *d_pay = nullptr; float2 *d_xsym = nullptr;
    const size_t nsym = (size_t)frames * S;
    TXCHK(hipMalloc((void **)&d_hdr, hdr.size())); TXCHK(hipMalloc((void **)&d_pay, std::max<size_t>(pay.size(), 1)));
    TXCHK(hipMalloc((void **)&d_xsym, (size_t)ch_count * nsym * M * sizeof(float2)));
    TXCHK(hipMemcpyAsync(d_hdr, hdr.data(), hdr.size(), hipMemcpyHostToDevice, st));
    TXCHK(hipMemcpyAsync(d_pay, pay.data(), pay.size(), hipMemcpyHostToDevice, st));
    TxSymArgs sa;
    sa.M = (int)M; sa.log2M = 0; while ((1u << sa.log2M) < M) sa.log2M++;
    sa.cp = (int)q->cp; sa.taper = (int)q->taper; sa.L = (int)(M + q->cp); sa.M_pilot = (int)q->od.M_pilot; sa.M_data = (int)Md;
    sa.S = (int)S; sa.S_hdr = (int)Sh; sa.S_pay = (int)Sp; sa.frames = (int)frames; sa.bps = (int)mod_bps(mod); sa.mod = mod;
    sa.g_data = 1.0f / sqrtf((float)(q->od.M_pilot + q->od.M_data));
    sa.sctype = q->d_sctype; sa.data_rank = q->d_drank; sa.pilot_rank = q->d_prank; sa.pilot_seq = q->d_pseq;
    sa.s0t = q->d_s0t; sa.s1t = q->d_s1t; sa.hdr = d_hdr; sa.pay = d_pay; sa.xsym = d_xsym; sa.nch = ch_count; sa.xs_ch = M; sa.xs_sym = (size_t)ch_count * M;      // symbol-major, like the generators
    if (M % 8 == 0) { sa.xs_ch = 8; sa.xs_grp = (size_t)8 * ch_count; }
    { int rc = tx_launch_sym(q, sa, (unsigned)nsym, ch_count, st); if (rc) return rc; }
    TXCHK(hipStreamSynchronize(st));
    (void)hipFree(d_hdr); (void)hipFree(d_pay);
    mctx_hip_traffic_t t = new mctx_hip_traffic_s();
    t->device = q->device;
    t->q = q; t->ch_first = ch_first; t->ch_count = ch_count; t->frames = frames; t->S = S; t->d_xsym = d_xsym;
    *out = t;
    return MCRX_OK;
}

extern "C" int mctx_hip_traffic_destroy(mctx_hip_traffic_t t)
{
    DevScope dev_scope_(t ? t->device : -1);
    if (!t) return MCRX_OK;
    (void)hipDeviceSynchronize();
    if (t->d_xsym) (void)hipFree(t->d_xsym);
    delete t;
    return MCRX_OK;
}

extern "C" int mctx_hip_traffic_tiles(mctx_hip_traffic_t t, long long first_block, size_t nblocks, void *d_tiles, void *stream)
{
    DevScope dev_scope_(t ? t->device : -1);
    if (!t || !d_tiles || (nblocks % 8)) { g_tx_err = "bad argument (blocks come in granules of 8)"; return MCRX_EINVAL; }
    if (!nblocks) return MCRX_OK;
    mctx_hip_t q = t->q;
    TxSynthArgs ya;
    ya.M = (int)q->M; ya.cp = (int)q->cp; ya.taper = (int)q->taper; ya.L = (int)(q->M + q->cp); ya.S = (int)t->S; ya.frames = (int)t->frames;
    ya.taperwin = q->d_taper; ya.xsym = t->d_xsym; ya.taps = q->d_taps; ya.v = nullptr; ya.out = nullptr;
    ya.xs_ch = q->M; ya.xs_sym = (size_t)t->ch_count * q->M;
    if (q->M % 8 == 0) { ya.xs_ch = 8; ya.xs_grp = (size_t)8 * t->ch_count; }
    ya.nblocks = (uint32_t)nblocks; ya.N = t->ch_count; ya.dtheta = 0; ya.first_sample_lo = 0; ya.gain = 1.0f;
    ya.ft0 = nullptr; ya.fS = nullptr; ya.xstride = 0; ya.b_first = 0; ya.hist = 0;
    const uint32_t ntiles = (uint32_t)(nblocks / 8), n = ntiles * t->ch_count;
    hipLaunchKernelGGL(txtiles_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ya, first_block, ntiles, t->ch_count,
                       (float2 *)d_tiles);
    TXCHK(hipGetLastError());
    return MCRX_OK;
}

static int tx_launch_ifft(mctx_hip_t q, const TxSynthArgs &ya, unsigned nblocks, hipStream_t st);

extern "C" int mctx_hip_synthesize_tiles(mctx_hip_t q, const void *d_tiles, unsigned groups, long long first_block, size_t nblocks,
                                         size_t lead_blocks, size_t keep_blocks, float gain, void *d_iq, void *stream)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q || !d_tiles || !d_iq || !groups || (q->N % groups) || (nblocks % 8) || (lead_blocks % 8) || keep_blocks > lead_blocks) {
        g_tx_err = "bad argument (granules of 8 blocks, groups dividing the channel count, keep <= lead)"; return MCRX_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    const unsigned K = q->K;
    const size_t tot = lead_blocks + nblocks;
    TxSynthArgs probe; probe.ft0 = nullptr; probe.hist = 0; probe.out_first = (uint32_t)(lead_blocks - keep_blocks);
    if (!tx_fused_ok(q, probe) && tot > q->syn_cap) {
        if (q->d_synv) { TXCHK(hipDeviceSynchronize()); TXCHK(hipFree(q->d_synv)); q->d_synv = nullptr; q->syn_cap = 0; }
        TXCHK(hipMalloc((void **)&q->d_synv, tot * K * sizeof(float2)));
        q->syn_cap = tot;
    }
    TxSynthArgs ya;
    ya.M = (int)q->M; ya.cp = (int)q->cp; ya.taper = (int)q->taper; ya.L = (int)(q->M + q->cp); ya.S = 0; ya.frames = 0;
    ya.taperwin = q->d_taper; ya.xsym = nullptr; ya.taps = q->d_taps; ya.v = q->d_synv; ya.out = (float2 *)d_iq;
    ya.nblocks = (uint32_t)tot; ya.N = q->N; ya.dtheta = q->dtheta; ya.gain = gain;
    // oscillator phase of local block 0 = absolute block first_block - lead (mod 2^32 samples, like the 32-bit accumulator)
    ya.first_sample_lo = (uint32_t)((unsigned long long)(first_block - (long long)lead_blocks) * (unsigned long long)K);
    ya.ft0 = nullptr; ya.fS = nullptr; ya.xstride = 0; ya.b_first = 0; ya.hist = 0;
    ya.tiles = (const float2 *)d_tiles; ya.ntiles = (uint32_t)(tot / 8); ya.cg = q->N / groups;
    ya.out_first = (uint32_t)(lead_blocks - keep_blocks);
    return tx_synthesize(q, ya, st);
}


// ------------------------------------------------------------------------------------------------
// Streaming interface: what the reference's multichanneltx class does per call
//   IsChannelReadyForData (lib/multichanneltx.cc:147-162), UpdateData (:165-189), GenerateSamples (:192-227)
// Frames start on OFDM symbol boundaries (the class steps all N frame generators together, one symbol of
// M+cp samples per GenerateFrameSamples, :230-242), so one launch produces one symbol period = M+cp blocks of
// 2N samples; a channel is ready again once the period with its frame's last (tail) symbol has been produced.
static int tx_launch_sym(mctx_hip_t q, const TxSymArgs &sa, unsigned nsym, unsigned nch, hipStream_t st)
{
    const dim3 gsym(nsym, nch), gsym8((nsym + TXSYM_PER - 1) / TXSYM_PER, nch);
    if (q->M & (q->M - 1)) {
        hipLaunchKernelGGL(txsym_dft_kernel, gsym, dim3(TXW), 0, st, sa);
        TXCHK(hipGetLastError());
        return MCRX_OK;
    }
    static const bool wide = devel_env("MCTX_TXSYM64") == nullptr || atoi(devel_env("MCTX_TXSYM64")) != 0;      // (0: the one-point-per-lane kernel, comparisons)
    if (q->M == 64 && wide && (sa.xs_sym % 2) == 0 && (sa.xs_ch % 2) == 0) {
        const bool pair = sa.xs_grp != 0 && (sa.nch & 1u) == 0 && sa.nch == nch;          // (txsym64_kernel: four symbols of two channels per wave)
        const dim3 g64 = pair ? dim3((nsym + TXSYM_PER / 2 - 1) / (TXSYM_PER / 2), nch / 2) : gsym8;
        hipLaunchKernelGGL(txsym64_kernel, g64, dim3(TXW), 0, st, sa, nsym);
        TXCHK(hipGetLastError());
        return MCRX_OK;
    }
    switch (std::max(1u, q->M / 64)) {
    case 1:  hipLaunchKernelGGL((txsym_kernel<1>), gsym8, dim3(TXW), 0, st, sa, nsym); break;
    case 2:  hipLaunchKernelGGL((txsym_kernel<2>), gsym8, dim3(TXW), 0, st, sa, nsym); break;
    case 4:  hipLaunchKernelGGL((txsym_kernel<4>), gsym8, dim3(TXW), 0, st, sa, nsym); break;
    case 8:  hipLaunchKernelGGL((txsym_kernel<8>), gsym8, dim3(TXW), 0, st, sa, nsym); break;
    case 16: hipLaunchKernelGGL((txsym_kernel<16>), gsym8, dim3(TXW), 0, st, sa, nsym); break;
    default: g_tx_err = "unsupported subcarrier count"; return MCRX_EUNSUPP;
    }
    TXCHK(hipGetLastError());
    return MCRX_OK;
}
static int tx_launch_ifft(mctx_hip_t q, const TxSynthArgs &ya, unsigned nblocks, hipStream_t st)
{
#define TX_IFFT(KK) hipLaunchKernelGGL((txifft_kernel<KK>), dim3(nblocks), dim3((KK) / 2 < 64 ? 64 : (KK) / 2), 0, st, ya)
    switch (q->K) {
    case 2: TX_IFFT(2); break;       case 4: TX_IFFT(4); break;     case 8: TX_IFFT(8); break;     case 16: TX_IFFT(16); break;
    case 32: TX_IFFT(32); break;     case 64: TX_IFFT(64); break;   case 128: TX_IFFT(128); break; case 256: TX_IFFT(256); break;
    case 512: TX_IFFT(512); break;   case 1024: TX_IFFT(1024); break;
    default: g_tx_err = "unsupported channel count"; return MCRX_EUNSUPP;
    }
#undef TX_IFFT
    TXCHK(hipGetLastError());
    return MCRX_OK;
}

// The synthesis bank + oscillator over blocks [ya.out_first, ya.nblocks) of the launch's local block axis: the fused kernel
// (synth_tile.hpp) where it exists -- a power-of-two K >= 128, a prototype that is bit-for-bit symmetric, the batch / ragged /
// sharded forms -- else the two-kernel path through `v` (inverse-FFT outputs in HBM).  MCTX_SYNTH=0 forces the latter.
static bool tx_fused_ok(mctx_hip_t q, const TxSynthArgs &ya)
{
    static const int env = devel_env("MCTX_SYNTH") ? atoi(devel_env("MCTX_SYNTH")) : 1;
    return env != 0 && q->taps_symmetric && !ya.ft0 && ya.hist == 0 && (q->K == 128 || q->K == 256 || q->K == 512 || q->K == 1024) &&
           (ya.out_first % 8) == 0;
}
template <int KK, int R, int IN>
static int tx_launch_fused_in(mctx_hip_t q, const TxSynthArgs &ya, hipStream_t st)
{
    const size_t lds = syn::Lds<KK, R>::bytes();
    static PerDeviceOnce attr;               // (per instantiation and device: devscope.hpp)
    TXCHK(raise_lds_limit((const void *)syn::synth_kernel<KK, R, IN>, lds, attr));
    const size_t nout = ya.nblocks - ya.out_first;
    // slab per workgroup: a whole number of workgroup waves over the CUs, at most 512 blocks (28 blocks of lead-in each)
    const size_t cap = (size_t)q->ncu * (KK >= 1024 ? 1 : 2);
    const size_t k = (nout + cap * 512 - 1) / (cap * 512);
    size_t slab = ((nout + cap * k - 1) / (cap * k) + 7) & ~(size_t)7;
    if (slab < 256) slab = 256;                     // (28 blocks of lead-in per slab)
    const unsigned grid = (unsigned)((nout + slab - 1) / slab);
    hipLaunchKernelGGL((syn::synth_kernel<KK, R, IN>), dim3(grid), dim3(KK / 2), lds, st, ya, (uint32_t)slab);
    TXCHK(hipGetLastError());
    return MCRX_OK;
}
// the input side of the kernel is a compile-time choice (three loaders in one body spilled the window): exchanged granules,
// the aligned symbol loader (rounds of 8 blocks only), or the block-by-block walk
template <int KK, int R>
static int tx_launch_fused(mctx_hip_t q, const TxSynthArgs &ya, hipStream_t st)
{
    if (ya.tiles) return tx_launch_fused_in<KK, R, syn::SYN_TILES>(q, ya, st);
    if constexpr (R == 8) {
        static const int al = devel_env("MCTX_ALIGNED") ? atoi(devel_env("MCTX_ALIGNED")) : 1;
        if (al && syn::syn_aligned(ya)) return tx_launch_fused_in<KK, 8, syn::SYN_SYMS>(q, ya, st);
    }
    return tx_launch_fused_in<KK, R, syn::SYN_WALK>(q, ya, st);
}
static int tx_synthesize(mctx_hip_t q, const TxSynthArgs &ya, hipStream_t st)
{
    if (tx_fused_ok(q, ya)) {
        switch (q->K) {
        case 128: return tx_launch_fused<128, 4>(q, ya, st);
        case 256: return tx_launch_fused<256, 4>(q, ya, st);
        case 512: { static const int r8 = devel_env("MCTX_R8") ? atoi(devel_env("MCTX_R8")) : 1; return r8 ? tx_launch_fused<512, 8>(q, ya, st) : tx_launch_fused<512, 4>(q, ya, st); }
        default:  { static const int r8 = devel_env("MCTX_R8") ? atoi(devel_env("MCTX_R8")) : 1; return r8 ? tx_launch_fused<1024, 8>(q, ya, st) : tx_launch_fused<1024, 4>(q, ya, st); }
        }
    }
    if (!ya.v) { g_tx_err = "two-kernel synthesis needs its inverse-FFT buffer"; return MCRX_EINVAL; }
    { int rc = tx_launch_ifft(q, ya, ya.nblocks, st); if (rc) return rc; }
    const unsigned K = q->K, tb = K < 256 ? 64 : 256;
    hipLaunchKernelGGL(txfir_kernel, dim3((K + tb - 1) / tb, (unsigned)((ya.nblocks + 7) / 8)), dim3(tb), 0, st, ya, K);
    TXCHK(hipGetLastError());
    return MCRX_OK;
}

// (re)size the per-channel frame slots to hold Sp payload symbols; frames in flight are carried over
static int tx_stream_slots(mctx_hip_t q, unsigned Sp)
{
    const unsigned N = q->N, M = q->M, Md = q->od.M_data, Sh = q->st_Sh, S = 3 + Sh + Sp + 1;
    uint8_t *npay = nullptr; float2 *nx = nullptr;
    TXCHK(hipMalloc((void **)&npay, (size_t)N * Sp * Md));
    TXCHK(hipMalloc((void **)&nx, (size_t)N * S * M * sizeof(float2)));
    if (q->d_spay) {
        TXCHK(hipStreamSynchronize(q->sst));
        TXCHK(hipMemcpy2D(npay, (size_t)Sp * Md, q->d_spay, (size_t)q->st_Spmax * Md, (size_t)q->st_Spmax * Md, N, hipMemcpyDeviceToDevice));
        TXCHK(hipMemcpy2D(nx, (size_t)S * M * sizeof(float2), q->d_sxsym, (size_t)q->st_Smax * M * sizeof(float2),
                          (size_t)q->st_Smax * M * sizeof(float2), N, hipMemcpyDeviceToDevice));
        TXCHK(hipFree(q->d_spay)); TXCHK(hipFree(q->d_sxsym));
    }
    q->d_spay = npay; q->d_sxsym = nx; q->st_Spmax = Sp; q->st_Smax = S;
    return MCRX_OK;
}

extern "C" int mctx_hip_stream_begin(mctx_hip_t q, unsigned max_payload_len)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q) { g_tx_err = "null handle"; return MCRX_EINVAL; }
    const unsigned N = q->N, M = q->M, K = q->K, L = M + q->cp, Md = q->od.M_data;
    // slots for the longest frame a payload of this size can make: BPSK behind two rate-1/2 codes
    unsigned Sh, Sp, S; frame_geometry(q, max_payload_len, 39, 7, 7, Sh, Sp, S);
    if (q->st_on) {                                 // called again: only ever grows (update also grows on demand)
        if (Sp > q->st_Spmax) { int rc = tx_stream_slots(q, Sp); if (rc) return rc; }
        q->st_maxpay = std::max(q->st_maxpay, max_payload_len);
        return MCRX_OK;
    }
    TXCHK(hipStreamCreate(&q->sst));
    q->st_maxpay = max_payload_len; q->st_Sh = Sh;
    TXCHK(hipMalloc((void **)&q->d_shdr, (size_t)N * Sh * Md));
    { int rc = tx_stream_slots(q, Sp); if (rc) return rc; }
    TXCHK(hipMalloc((void **)&q->d_ft0, N * sizeof(long long)));
    TXCHK(hipMalloc((void **)&q->d_fS, N * sizeof(int)));
    for (int i = 0; i < 2; i++) {
        TXCHK(hipMalloc((void **)&q->d_sv[i], (size_t)(TX_P - 1 + L) * K * sizeof(float2)));
        TXCHK(hipMemset(q->d_sv[i], 0, (size_t)(TX_P - 1 + L) * K * sizeof(float2)));
    }
    TXCHK(hipMalloc((void **)&q->d_sout, (size_t)L * K * sizeof(float2)));
    TXCHK(hipHostMalloc((void **)&q->h_sout, (size_t)L * K * sizeof(float2), hipHostMallocDefault));
    q->ft0.assign(N, 0); q->fS.assign(N, 0); q->fS_new.assign(N, 0); q->assembled.assign(N, 0); q->pending.assign(N, 0);
    q->period = 0; q->out_pos = L; q->sv_cur = 0;
    q->st_on = true;
    return MCRX_OK;
}

extern "C" int mctx_hip_stream_reset(mctx_hip_t q)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q || !q->st_on) { g_tx_err = "stream not started"; return MCRX_EINVAL; }
    const unsigned L = q->M + q->cp;
    TXCHK(hipStreamSynchronize(q->sst));
    for (int i = 0; i < 2; i++) TXCHK(hipMemset(q->d_sv[i], 0, (size_t)(TX_P - 1 + L) * q->K * sizeof(float2)));
    std::fill(q->fS.begin(), q->fS.end(), 0); std::fill(q->assembled.begin(), q->assembled.end(), 0);
    std::fill(q->pending.begin(), q->pending.end(), 0);
    q->out_pos = L;                         // (like the reference's Reset, the oscillator keeps its phase: blocks_out stays)
    return MCRX_OK;
}

extern "C" int mctx_hip_stream_ready(mctx_hip_t q, unsigned ch)
{
    if (!q || !q->st_on || ch >= q->N) return MCRX_EINVAL;
    return q->assembled[ch] ? 0 : 1;
}

extern "C" int mctx_hip_stream_update(mctx_hip_t q, unsigned ch, const uint8_t *header8, const uint8_t *payload, unsigned payload_len,
                                      int mod, int fec0, int fec1)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q || !q->st_on || !header8 || (!payload && payload_len)) { g_tx_err = "bad argument"; return MCRX_EINVAL; }
    if (ch >= q->N) { g_tx_err = "error: multichanneltx::UpdateData(), invalid channel id"; return MCRX_EINVAL; }
    if (!mod_bps(mod)) { g_tx_err = "unsupported modulation scheme"; return MCRX_EUNSUPP; }
    if (!fec_supported(fec0) || !fec_supported(fec1)) { g_tx_err = "unsupported fec scheme"; return MCRX_EUNSUPP; }
    if (q->assembled[ch]) { g_tx_err = "warning: multichanneltx::UpdateData(), channel busy"; return MCRX_EBUSY; }
    const unsigned M = q->M, Md = q->od.M_data;
    unsigned Sh, Sp, S; frame_geometry(q, payload_len, mod, fec0, fec1, Sh, Sp, S);
    if (Sp > q->st_Spmax) { int rc = tx_stream_slots(q, Sp + Sp / 4); if (rc) return rc; }
    FrameSymbols fsym;
    std::vector<uint8_t> pl(payload, payload + payload_len);
    assemble_frame(header8, pl, mod, fec0, fec1, Md, Sh, Sp, fsym);
    uint8_t *dh = q->d_shdr + (size_t)ch * q->st_Sh * Md, *dp = q->d_spay + (size_t)ch * q->st_Spmax * Md;
    TXCHK(hipMemcpyAsync(dh, fsym.hdr.data(), fsym.hdr.size(), hipMemcpyHostToDevice, q->sst));
    if (!fsym.pay.empty()) TXCHK(hipMemcpyAsync(dp, fsym.pay.data(), fsym.pay.size(), hipMemcpyHostToDevice, q->sst));
    TXCHK(hipStreamSynchronize(q->sst));                // (the staging vectors go out of scope)
    TxSymArgs sa;
    sa.M = (int)M; sa.log2M = 0; while ((1u << sa.log2M) < M) sa.log2M++;
    sa.cp = (int)q->cp; sa.taper = (int)q->taper; sa.L = (int)(M + q->cp); sa.M_pilot = (int)q->od.M_pilot; sa.M_data = (int)Md;
    sa.S = (int)S; sa.S_hdr = (int)Sh; sa.S_pay = (int)Sp; sa.frames = 1; sa.bps = (int)mod_bps(mod); sa.mod = mod;
    sa.g_data = 1.0f / sqrtf((float)(q->od.M_pilot + q->od.M_data));
    sa.sctype = q->d_sctype; sa.data_rank = q->d_drank; sa.pilot_rank = q->d_prank; sa.pilot_seq = q->d_pseq;
    sa.s0t = q->d_s0t; sa.s1t = q->d_s1t; sa.hdr = dh; sa.pay = dp; sa.nch = 1;
    sa.xsym = q->d_sxsym + (size_t)ch * q->st_Smax * M;
    int rc = tx_launch_sym(q, sa, S, 1, q->sst);
    if (rc) return rc;
    q->assembled[ch] = 1; q->pending[ch] = 1; q->fS_new[ch] = (int)S;
    return MCRX_OK;
}

static int tx_stream_period(mctx_hip_t q)
{
    const unsigned N = q->N, M = q->M, K = q->K, L = M + q->cp;
    const long long B0 = q->period * (long long)L;
    for (unsigned ch = 0; ch < N; ch++) if (q->pending[ch]) { q->ft0[ch] = B0; q->fS[ch] = q->fS_new[ch]; q->pending[ch] = 0; }
    TXCHK(hipMemcpyAsync(q->d_ft0, q->ft0.data(), N * sizeof(long long), hipMemcpyHostToDevice, q->sst));
    TXCHK(hipMemcpyAsync(q->d_fS, q->fS.data(), N * sizeof(int), hipMemcpyHostToDevice, q->sst));
    float2 *vbuf = q->d_sv[q->sv_cur];
    TxSynthArgs ya;
    ya.M = (int)M; ya.cp = (int)q->cp; ya.taper = (int)q->taper; ya.L = (int)L; ya.S = 0; ya.frames = 0;
    ya.taperwin = q->d_taper; ya.xsym = q->d_sxsym; ya.taps = q->d_taps; ya.v = vbuf + (size_t)(TX_P - 1) * K; ya.out = q->d_sout;
    ya.nblocks = L; ya.N = N; ya.dtheta = q->dtheta; ya.first_sample_lo = (uint32_t)((unsigned long long)q->blocks_out * K); ya.gain = 1.0f;
    ya.ft0 = q->d_ft0; ya.fS = q->d_fS; ya.xstride = q->st_Smax; ya.b_first = B0; ya.hist = TX_P - 1;
    int rc = tx_launch_ifft(q, ya, L, q->sst);
    if (rc) return rc;
    const unsigned tb = K < 256 ? 64 : 256;
    hipLaunchKernelGGL(txfir_kernel, dim3((K + tb - 1) / tb, (L + 7) / 8), dim3(tb), 0, q->sst, ya, K);
    TXCHK(hipGetLastError());
    TXCHK(hipMemcpyAsync(q->h_sout, q->d_sout, (size_t)L * K * sizeof(float2), hipMemcpyDeviceToHost, q->sst));
    // next period's history = the last 25 blocks of (history ++ this period), written into the other buffer
    float2 *nbuf = q->d_sv[q->sv_cur ^ 1];
    TXCHK(hipMemcpyAsync(nbuf, vbuf + (size_t)L * K, (size_t)(TX_P - 1) * K * sizeof(float2), hipMemcpyDeviceToDevice, q->sst));
    q->sv_cur ^= 1;
    TXCHK(hipStreamSynchronize(q->sst));
    for (unsigned ch = 0; ch < N; ch++)
        if (q->assembled[ch] && !q->pending[ch] && q->fS[ch] && B0 + (long long)L >= q->ft0[ch] + (long long)q->fS[ch] * L) q->assembled[ch] = 0;
    q->period++;
    q->out_pos = 0;
    return MCRX_OK;
}

extern "C" int mctx_hip_stream_generate(mctx_hip_t q, float *out)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q || !q->st_on || !out) { g_tx_err = "bad argument"; return MCRX_EINVAL; }
    const unsigned L = q->M + q->cp;
    if (q->out_pos >= L) { int rc = tx_stream_period(q); if (rc) return rc; }
    memcpy(out, q->h_sout + (size_t)q->out_pos * q->K, (size_t)q->K * sizeof(float2));
    q->out_pos++; q->blocks_out++;
    return MCRX_OK;
}


// ------------------------------------------------------------------------------------------------
// One frame of one frame generator, at the channel rate: what ofdmflexframegen_assemble followed by
// ofdmflexframegen_writesymbol until it reports the last symbol produces (lib/ofdmtxrx.cc:297-342,
// 385-388): S0a, S0b, S1, header symbols, payload symbols and the tail symbol, M + cp samples each.
extern "C" size_t mctx_hip_frame_len(mctx_hip_t q, unsigned payload_len, int mod, int fec0, int fec1)
{
    if (!q || !mod_bps(mod) || !fec_supported(fec0) || !fec_supported(fec1)) return 0;
    unsigned Sh, Sp, S; frame_geometry(q, payload_len, mod, fec0, fec1, Sh, Sp, S);
    return (size_t)S * (q->M + q->cp);
}

extern "C" int mctx_hip_frame(mctx_hip_t q, const uint8_t *header8, const uint8_t *payload, unsigned payload_len,
                              int mod, int fec0, int fec1, float gain, float *out, size_t out_cap)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q || !header8 || (!payload && payload_len) || !out) { g_tx_err = "bad argument"; return MCRX_EINVAL; }
    if (!mod_bps(mod)) { g_tx_err = "unsupported modulation scheme"; return MCRX_EUNSUPP; }
    if (!fec_supported(fec0) || !fec_supported(fec1)) { g_tx_err = "unsupported fec scheme"; return MCRX_EUNSUPP; }
    const unsigned M = q->M, Md = q->od.M_data, L = M + q->cp;
    unsigned Sh, Sp, S; frame_geometry(q, payload_len, mod, fec0, fec1, Sh, Sp, S);
    const size_t ns = (size_t)S * L;
    if (out_cap < ns) { g_tx_err = "output buffer shorter than mctx_hip_frame_len()"; return MCRX_EINVAL; }
    FrameSymbols fsym;
    std::vector<uint8_t> pl(payload, payload + payload_len);
    assemble_frame(header8, pl, mod, fec0, fec1, Md, Sh, Sp, fsym);
    uint8_t *d_hdr = nullptr, *d_pay = nullptr; float2 *d_x = nullptr, *d_out = nullptr;
    auto done = [&](int rc) { (void)hipFree(d_hdr); (void)hipFree(d_pay); (void)hipFree(d_x); (void)hipFree(d_out); return rc; };
#define TXF(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_tx_err = std::string(#x) + ": " + hipGetErrorString(e_); return done(MCRX_EHIP); } } while (0)
    TXF(hipMalloc((void **)&d_hdr, fsym.hdr.size())); TXF(hipMalloc((void **)&d_pay, std::max<size_t>(fsym.pay.size(), 1)));
    TXF(hipMalloc((void **)&d_x, (size_t)S * M * sizeof(float2))); TXF(hipMalloc((void **)&d_out, ns * sizeof(float2)));
    TXF(hipMemcpy(d_hdr, fsym.hdr.data(), fsym.hdr.size(), hipMemcpyHostToDevice));
    if (!fsym.pay.empty()) TXF(hipMemcpy(d_pay, fsym.pay.data(), fsym.pay.size(), hipMemcpyHostToDevice));
    TxSymArgs sa;
    sa.M = (int)M; sa.log2M = 0; while ((1u << sa.log2M) < M) sa.log2M++;
    sa.cp = (int)q->cp; sa.taper = (int)q->taper; sa.L = (int)L; sa.M_pilot = (int)q->od.M_pilot; sa.M_data = (int)Md;
    sa.S = (int)S; sa.S_hdr = (int)Sh; sa.S_pay = (int)Sp; sa.frames = 1; sa.bps = (int)mod_bps(mod); sa.mod = mod;
    sa.g_data = 1.0f / sqrtf((float)(q->od.M_pilot + q->od.M_data));
    sa.sctype = q->d_sctype; sa.data_rank = q->d_drank; sa.pilot_rank = q->d_prank; sa.pilot_seq = q->d_pseq;
    sa.s0t = q->d_s0t; sa.s1t = q->d_s1t; sa.hdr = d_hdr; sa.pay = d_pay; sa.xsym = d_x; sa.nch = 1;
    int rc = tx_launch_sym(q, sa, S, 1, nullptr);
    if (rc) return done(rc);
    TxSynthArgs ya;
    ya.M = (int)M; ya.cp = (int)q->cp; ya.taper = (int)q->taper; ya.L = (int)L; ya.S = (int)S; ya.frames = 1;
    ya.taperwin = q->d_taper; ya.xsym = d_x; ya.taps = q->d_taps; ya.v = nullptr; ya.out = d_out;
    ya.nblocks = (uint32_t)ns; ya.N = 1; ya.dtheta = 0; ya.first_sample_lo = 0; ya.gain = gain;
    ya.ft0 = nullptr; ya.fS = nullptr; ya.xstride = 0; ya.b_first = 0; ya.hist = 0;
    hipLaunchKernelGGL(txframe_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, nullptr, ya, (uint32_t)ns);
    TXF(hipGetLastError());
    TXF(hipMemcpy(out, d_out, ns * sizeof(float2), hipMemcpyDeviceToHost));
#undef TXF
    return done(MCRX_OK);
}
