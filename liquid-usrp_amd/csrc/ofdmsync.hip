// ofdmsync.hip -- bank of OFDM flexible-frame synchronizers for gfx950, one wavefront
// per channel.
//
// Replaces the reference's per-channel call
//   ofdmflexframesync_execute(framesync[i], &X[i], 1)      (lib/multichannelrx.cc:193-194)
// i.e. liquid's ofdmframesync state machine (seek S0 -> S0a -> S0b -> S1 -> symbols) plus
// ofdmflexframesync's header / payload recovery (BPSK header, Golay(24,12) + CRC-32;
// payload demodulated soft, de-interleaved, FEC decoded, CRC checked).
//
// The reference pushes one sample per call into a sample-serial state machine.  Here each
// wavefront walks its channel's stream event by event: the sample index of the next
// state-machine event is known in closed form from the timer, the (M+cp)-sample window is
// read straight from the channelizer's (channel, tile) granules in HBM, the NCO phase of
// every window sample is the exact 32-bit closed form theta_ref + (t - t_ref) * dtheta,
// the M-point FFT runs across the 64 lanes (shuffle butterflies, bit-reversed subcarrier
// per lane; direct DFT for M that are not powers of two), and the per-symbol pilot fit,
// equalisation and soft demodulation are lane-parallel with wave reductions.  State lives
// in HBM between launches so Execute() can be fed arbitrary pieces.
#include "devmath.h"
#include "kernels.h"
#include <utility>

namespace mcrx {

#define WV 64
#ifndef SY_PART
#define SY_PART -1          /* -1: the whole file in one translation unit; 0/1/2/3: see the launchers at the end */
#endif
#ifndef SY_PROFILE
#define SY_PROFILE 0        /* 1: MCRX_DEBUG=2 cycle counters per event / phase (they cost ~40 registers in the scout) */
#endif
#ifndef H128_EARLY_OUT
#define H128_EARLY_OUT 1    /* Hamming(12,8) soft decision: no neighbour search in a wave whose hard decisions are all codewords */
#endif
#ifndef SY_RANK
#define SY_RANK 1           /* scouts rank the chain of a window that holds many frames (Walker::rank_window) instead of hopping along it */
#endif
#ifndef SY_SEG_BURST
#define SY_SEG_BURST 1      /* segment waves take idle stretches four SEEK events at a time (Walker::seek_burst) */
#endif
#define SY_PROF(a) (SY_PROFILE && ((a).debug & 2) != 0)
// Phase-skipping switches (profiling ablations: MCRX_DEBUG=256 / 512 run the general decoder without the soft de-interleaver /
// the Viterbi decoder) exist in development builds only (-DMCRX_DEVEL); the release library cannot be told to skip work.
#ifdef MCRX_DEVEL
#define MCRX_DEVEL_ABLATE(a) ((unsigned)(a).debug >> 8 << 6)
#else
#define MCRX_DEVEL_ABLATE(a) 0u
#endif
#define TWO_PI_F 6.283185307179586f
#define PI_F 3.14159265358979323846f

// ------------------------------------------------------------------ small utilities
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
#include "lean_prims.hpp"       // packed-f32 / LDS-crossbar building blocks, the lane <-> subcarrier maps of the lean transforms
#include "viterbi_frames.hpp"   // the K = 7 soft decoder of decode_general_kernel: a frame per wave, a trellis block per lane

__device__ __forceinline__ unsigned fec_enc_len_d(unsigned fs, unsigned n)
{
    if (fs == 6) return (n / 2) * 3 + (n % 2) * 2;      // Hamming(12,8)
    if (fs == 7) return (n / 3) * 6 + (n % 3) * 3;      // Golay(24,12)
    if (fs == 11) return 2 * n + 2;                     // r = 1/2, K = 7 convolutional: 2 (8 n + 6) bits
    if (fs == 2) return 3 * n;                          // rep3
    if (fs == 3) return 5 * n;                          // rep5
    if (fs == 4) return (14 * n + 7) / 8;               // Hamming(7,4): two 7-bit symbols per byte, bit-packed
    if (fs == 5) return 2 * n;                          // Hamming(8,4)
    return n;
}
// liquid's fec_scheme ids this receiver decodes: none, rep3, rep5, h74, h84, h128, g2412 (1..7) and v27 (11)
__device__ __forceinline__ bool fec_known_d(unsigned fs) { return (fs >= 1 && fs <= 7) || fs == 11; }
// packetizer.c: every coded stage is followed by the four-pass interleaver
__device__ __forceinline__ unsigned fec_depth_d(unsigned fs) { return fs == 1 ? 0u : 4u; }
__device__ __forceinline__ unsigned mod_bps_d(unsigned m)
{ return m == 39 ? 1u : m == 40 ? 2u : m == 27 ? 4u : m == 29 ? 6u : 0u; }

// ------------------------------------------------------------------ CRC-32, lane parallel
__device__ uint32_t crc32_wave(const CodingDev cod, const uint8_t *p, uint32_t n)
{
    const int l = lane_id();
    const uint32_t Lc = (n + WV - 1) / WV;
    uint32_t start = (uint32_t)l * Lc, stop = start + Lc;
    if (start > n) start = n;
    if (stop > n) stop = n;
    uint32_t s = (l == 0) ? 0xFFFFFFFFu : 0u;
    for (uint32_t i = start; i < stop; i++) s = (s >> 8) ^ cod.crc_byte[(s ^ p[i]) & 0xff];
    uint32_t z = n - stop;                      // bytes that follow my chunk
    for (int k = 0; z; k++, z >>= 1) {
        if (z & 1) {
            const uint32_t *A = cod.crc_zadv + (size_t)k * 1024;
            s = A[s & 0xff] ^ A[256 + ((s >> 8) & 0xff)] ^ A[512 + ((s >> 16) & 0xff)] ^ A[768 + (s >> 24)];
        }
    }
    return ~wave_xor_u32(s);
}

// ------------------------------------------------------------------ interleaver (inverse)
__device__ __forceinline__ void il_dims(unsigned n, unsigned &Mi, unsigned &Ni)
{
    Mi = 1 + (unsigned)floorf(sqrtf((float)n));
    Ni = n / Mi;
    while (n >= Mi * Ni) Ni++;
}
// one pass: for the i-th valid cell j of the column walk swap masked bits of x[2i], x[2j+1].
// SOFT: elements are groups of 8 soft bits (uint64), the mask selects whole bytes.
template <bool SOFT>
__device__ void il_pass(uint8_t *x, unsigned n, unsigned Mi, unsigned Ncol, unsigned mask)
{
    const int l = lane_id();
    // liquid's walk starts in column n/3 WITHOUT reducing it modulo Ncol (so the first column is
    // usually a row-shifted alias of column (n/3) % Ncol), then steps (c+1) % Ncol; it stops
    // after n/2 valid cells, which is always before an aliased cell would repeat.
    const unsigned n2 = n / 2, total = Mi * (Ncol + 1), c0 = n / 3;
    unsigned long long m64 = 0;
    if (SOFT) for (int k = 0; k < 8; k++) if ((mask >> (7 - k)) & 1) m64 |= 0xFFull << (8 * k);
    // Lane l visits cells q = l, l + 64, ...: (row, column) advance incrementally (no division in
    // the loop), and UN cells per lane are in flight at once -- the pairs of one pass are disjoint,
    // so all loads of a group may precede its stores.
    constexpr int UN = 4;
    unsigned m = (unsigned)l % Mi, tcol = (unsigned)l / Mi, cm = (c0 + tcol) % Ncol;
    const unsigned dm = WV % Mi, dt = WV / Mi;
    const bool incremental = dt + 1 < Ncol;
    unsigned base = 0;
    for (unsigned q0 = 0; q0 < total && base < n2; q0 += UN * WV) {
        unsigned ia[UN], jb[UN]; bool ok[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const unsigned q = q0 + (unsigned)(u * WV + l);
            const unsigned c = tcol ? cm : c0;
            const unsigned j = m * Ncol + c;
            const bool valid = (q < total) && (j < n2);
            const unsigned long long bal = __ballot(valid);
            ia[u] = base + (unsigned)__popcll(bal & ((1ull << l) - 1ull));
            jb[u] = j;
            ok[u] = valid && ia[u] < n2;
            base += (unsigned)__popcll(bal);
            if (incremental) {
                m += dm; const unsigned carry = m >= Mi ? 1u : 0u; m -= carry ? Mi : 0u;
                const unsigned inc = dt + carry;
                tcol += inc; cm += inc; cm -= cm >= Ncol ? Ncol : 0u;
            } else {
                const unsigned qn = q + WV;
                m = qn % Mi; tcol = qn / Mi; cm = (c0 + tcol) % Ncol;
            }
        }
        if (SOFT) {
            unsigned long long va[UN], vb[UN];
            unsigned long long *x64 = reinterpret_cast<unsigned long long *>(x);
#pragma unroll
            for (int u = 0; u < UN; u++) if (ok[u]) { va[u] = x64[2 * ia[u]]; vb[u] = x64[2 * jb[u] + 1]; }
#pragma unroll
            for (int u = 0; u < UN; u++) if (ok[u]) {
                x64[2 * ia[u]]     = (va[u] & ~m64) | (vb[u] & m64);
                x64[2 * jb[u] + 1] = (vb[u] & ~m64) | (va[u] & m64);
            }
        } else {
            unsigned va[UN], vb[UN];
#pragma unroll
            for (int u = 0; u < UN; u++) if (ok[u]) { va[u] = x[2 * ia[u]]; vb[u] = x[2 * jb[u] + 1]; }
#pragma unroll
            for (int u = 0; u < UN; u++) if (ok[u]) {
                x[2 * ia[u]]     = (uint8_t)((va[u] & ~mask) | (vb[u] & mask));
                x[2 * jb[u] + 1] = (uint8_t)((vb[u] & ~mask) | (va[u] & mask));
            }
        }
    }
    __syncthreads();
}
template <bool SOFT>
__device__ void deinterleave(uint8_t *x, unsigned n, unsigned depth)
{
    unsigned Mi, Ni; il_dims(n, Mi, Ni);
    if (depth > 3) il_pass<SOFT>(x, n, Mi, Ni + 8, 0x33);
    if (depth > 2) il_pass<SOFT>(x, n, Mi, Ni + 4, 0x55);
    if (depth > 1) il_pass<SOFT>(x, n, Mi, Ni + 2, 0x0f);
    if (depth > 0) il_pass<SOFT>(x, n, Mi, Ni, 0xff);
}

// ------------------------------------------------------------------ block codes
__device__ __forceinline__ unsigned par_d(unsigned v) { return (unsigned)__popc(v) & 1u; }
__device__ __forceinline__ unsigned h128_dec_sym(unsigned c)
{
    unsigned z = (par_d(c & 0x01f) << 3) | (par_d(c & 0x1e1) << 2) | (par_d(c & 0x666) << 1) | par_d(c & 0xaaa);
    if (z && z <= 12) c ^= 1u << (12 - z);
    return (c & 0x00f) | ((c & 0x0e0) >> 1) | ((c & 0x200) >> 2);
}
// Soft decision without table walks (liquid: re-encode the hard estimate, then compare the distance-3
// neighbour codewords in ascending order).  Hamming(12,8) is linear, so the distance-3 neighbours
// of codeword enc(s0) are enc(s0 ^ u) for the fixed set {u : weight(enc(u)) == 3}, and
//   dist(enc(s0 ^ u)) - dist(enc(s0)) = sum over the three bits of enc(u) of the cost of flipping them.
// The patterns are enumerated at compile time; the reference order (estimate first, then neighbours
// by ascending symbol, strict `<`) becomes a lexicographic minimum over (distance, symbol).
constexpr unsigned h128c_par(unsigned v) { v ^= v >> 4; v ^= v >> 2; v ^= v >> 1; return v & 1u; }
constexpr unsigned h128c_enc(unsigned s)
{
    return (s & 0x0f) | ((s & 0x70) << 1) | ((s & 0x80) << 2) | (h128c_par(s & 0xda) << 11) | (h128c_par(s & 0xb6) << 10) |
           (h128c_par(s & 0x71) << 8) | (h128c_par(s & 0x0f) << 4);
}
constexpr int h128c_weight(unsigned v) { int n = 0; while (v) { n += (int)(v & 1u); v >>= 1; } return n; }
template <unsigned U>
__device__ __forceinline__ void h128_try(const int (&flip)[12], unsigned s0, int &best)
{
    constexpr unsigned cw = h128c_enc(U);
    if constexpr (h128c_weight(cw) == 3) {
        int d = 4096;
#pragma unroll
        for (int k = 0; k < 12; k++) if ((cw >> (11 - k)) & 1u) d += flip[k];
        const int key = (d << 9) | (int)((s0 ^ U) + 1u);
        best = key < best ? key : best;
    }
}
template <unsigned... U>
__device__ __forceinline__ void h128_try_all(const int (&flip)[12], unsigned s0, int &best, std::integer_sequence<unsigned, U...>)
{ (h128_try<U + 1u>(flip, s0, best), ...); }
__device__ __forceinline__ unsigned h128_dec_soft_words(uint32_t w0, uint32_t w1, uint32_t w2)
{
    const uint32_t w[3] = { w0, w1, w2 };
    int cost[12]; unsigned c = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const int sb = (int)((w[k >> 2] >> (8 * (k & 3))) & 0xffu);
        cost[k] = 255 - 2 * sb;                     // cost of deciding 1 minus cost of deciding 0
        c = (c << 1) | (sb > 127 ? 1u : 0u);
    }
    const unsigned s0 = h128_dec_sym(c);
    // A neighbour at distance 3 beats the estimate only if flipping its three bits LOWERS the soft distance, which takes a bit where the
    // estimate's codeword goes against the hard decision -- and there is none when the hard decisions are a codeword already (zero
    // syndrome: every flip costs |255 - 2 soft| >= 1).  On clean traffic that is every symbol: the seventeen neighbour sums are only
    // formed when some lane of the wave has a non-zero syndrome (same decisions either way: ties go to the estimate).
    const unsigned z0 = (par_d(c & 0x01f) << 3) | (par_d(c & 0x1e1) << 2) | (par_d(c & 0x666) << 1) | par_d(c & 0xaaa);
    if (H128_EARLY_OUT && __ballot(z0 != 0u) == 0ull) return s0;
    const unsigned cw0 = (s0 & 0x0f) | ((s0 & 0x70) << 1) | ((s0 & 0x80) << 2) | (par_d(s0 & 0xda) << 11) | (par_d(s0 & 0xb6) << 10) |
                         (par_d(s0 & 0x71) << 8) | (par_d(s0 & 0x0f) << 4);
    int flip[12];
#pragma unroll
    for (int k = 0; k < 12; k++) flip[k] = ((cw0 >> (11 - k)) & 1u) ? -cost[k] : cost[k];
    int best = 4096 << 9;
    h128_try_all(flip, s0, best, std::make_integer_sequence<unsigned, 255>{});
    const unsigned r = (unsigned)best & 0x1ffu;
    return r ? r - 1u : s0;
}
__device__ __forceinline__ unsigned h128_dec_soft_fast(const uint8_t *soft)
{
    const uint32_t *w32 = reinterpret_cast<const uint32_t *>(soft);           // 12-byte groups are 4-byte aligned
    return h128_dec_soft_words(w32[0], w32[1], w32[2]);
}
__device__ __forceinline__ unsigned golay_mulP(unsigned v)
{
    const unsigned P[12] = { 0x08ed, 0x01db, 0x03b5, 0x0769, 0x0ed1, 0x0da3, 0x0b47, 0x068f, 0x0d1d, 0x0a3b, 0x0477, 0x0ffe };
    unsigned y = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) if (v & (1u << (11 - i))) y ^= P[i];
    return y;
}
__device__ unsigned golay_dec_sym(unsigned r)
{
    const unsigned P[12] = { 0x08ed, 0x01db, 0x03b5, 0x0769, 0x0ed1, 0x0da3, 0x0b47, 0x068f, 0x0d1d, 0x0a3b, 0x0477, 0x0ffe };
    unsigned rp = (r >> 12) & 0xfff, rm = r & 0xfff;
    unsigned s = rp ^ golay_mulP(rm), em = 0;
    bool found = __popc(s) <= 3;
    if (!found) {
#pragma unroll
        for (int i = 0; i < 12; i++) if (!found && __popc(s ^ P[i]) <= 2) { em = 1u << (11 - i); found = true; }
    }
    if (!found) {
        unsigned sP = golay_mulP(s);
        if (__popc(sP) <= 3) { em = sP; found = true; }
        else {
#pragma unroll
            for (int i = 0; i < 12; i++) if (!found && __popc(sP ^ P[i]) <= 2) { em = sP ^ P[i]; found = true; }
        }
    }
    return (rm ^ em) & 0xfff;
}

// ---- r = 1/2, K = 7 convolutional code (liquid LIQUID_FEC_CONV_V27 = libfec viterbi27): one wave, lane = trellis state.
// Maximum-likelihood Viterbi over 8-bit soft symbols exactly as oracle/ll_fec.c states it (branch metric = sum of
// |symbol - expected|, 32-bit path metrics, ties to the predecessor with the older bit 0, full traceback from state 0).
// A full traceback needs a 64-bit decision word per trellis step -- 77 KB for a 1200-byte payload, which no buffer of the
// frame has room for.  So the forward pass keeps only the path metrics at every VIT_B-th step (64 x 16 bits, normalised:
// differences between states never exceed 6 x 510), and the traceback walks the blocks from the last to the first,
// re-running each block's forward pass from its checkpoint into an LDS scratch of VIT_B decision words and then tracing
// back through it: the same survivors as one pass with all decisions kept, at twice the add-compare-select work.
#define VIT_B 960u             /* steps per block: a multiple of 64 (traceback chunks own whole bytes) and of 6 (the lane layout's period) */
struct VitSym {                 // the two soft symbols of a step: from soft bytes, or from packed hard bits (0 / 255)
    const uint8_t *p; bool hard;
    // symbols of step t0 + lane as sa | sb << 8 (steps >= T: anything); one load per 64 steps instead of two per step
    __device__ __forceinline__ unsigned chunk(unsigned t0, unsigned T) const
    {
        unsigned t = t0 + (unsigned)lane_id(); t = t < T ? t : T - 1;
        if (!hard) return *reinterpret_cast<const uint16_t *>(p + 2 * (size_t)t);
        const unsigned b = 2 * t, v = p[b >> 3];
        return (((v >> (7 - (b & 7))) & 1u) ? 255u : 0u) | ((((v >> (6 - (b & 7))) & 1u) ? 255u : 0u) << 8);
    }
};
// forward pass over steps [t0, t1), t0 a multiple of 64: pm = this lane's path metric; decision words to `dec` (LDS) if not null.
// Lane layout: at time t the lane l holds the state rotl6(l, t mod 6).  A new state n = (p << 1 | b) & 63 is then stored in
// the lane of one of its two predecessors p = (n >> 1) | (x << 5) and the other predecessor is the lane one bit away
// (bit q = (6 - (t+1) mod 6) mod 6) -- so an add-compare-select step is ONE lane exchange (DPP / permlane) instead of two
// LDS-crossbar permutes, which is what the step time was.  Metrics, tie rule and decisions are those of the natural layout.
__device__ __forceinline__ unsigned rotl6(unsigned v, unsigned r) { return ((v << r) | (v >> (6u - r))) & 63u; }
// One add-compare-select step whose (t + 1) mod 6 is R1.  sa, sb: the step's soft symbols (wave-uniform), san = sa ^ 255,
// sbn = sb ^ 255; ma, mb: 255 where this lane's state expects a 1 on the first / second output, else 0 -- so
// sa ^ ma = |sa - expected| and the two candidates are two v_xad_u32 each.
template <unsigned R1>
__device__ __forceinline__ int vit_step(int pm, unsigned sa, unsigned sb, unsigned san, unsigned sbn, unsigned ma, unsigned mb,
                                        unsigned s, int bp32, bool &take1)
{
    constexpr unsigned q = R1 ? 6u - R1 : 0u;
    int p0, p1;                                             // metrics of the predecessors with x = 0 / x = 1 (x = bit q of the lane index)
    if constexpr (q >= 4) {                                 // v_permlane32_swap / v_permlane16_swap on two copies: lower partner, upper partner
        p0 = pm; p1 = pm;
        if constexpr (q == 5) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p0), "+v"(p1));
        else                  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(p0), "+v"(p1));
    } else {
        int other;
        if constexpr (q == 2) other = __builtin_bit_cast(int, xor4_dpp(__builtin_bit_cast(float, pm)));
        else                  other = __builtin_bit_cast(int, xor_lane<(1 << q)>(__builtin_bit_cast(float, pm), bp32));
        const bool xq = (s >> q) & 1u;
        p0 = xq ? other : pm; p1 = xq ? pm : other;
    }
    const int m0 = (int)(((unsigned)p0 + (sa ^ ma)) + (sb ^ mb));
    const int m1 = (int)(((unsigned)p1 + (san ^ ma)) + (sbn ^ mb));     // = p1 + 510 - (branch metric of x = 0)
    take1 = m1 < m0;
    return take1 ? m1 : m0;
}
template <bool DEC>                // DEC: keep the decision words (lane k of a chunk: step tc + k) in `dec`
__device__ __forceinline__ int vit_forward(const VitSym &sy, unsigned t0, unsigned t1, unsigned T, int pm, unsigned long long *dec)
{
    const unsigned s = (unsigned)lane_id();
    unsigned ma[6], mb[6];                                  // [r]: for the state this lane holds when (t + 1) mod 6 = r
#pragma unroll
    for (unsigned r = 0; r < 6; r++) {
        const unsigned n = rotl6(s, r);
        ma[r] = (__builtin_popcount(n & 0x6d) & 1) ? 255u : 0u; mb[r] = (__builtin_popcount(n & 0x4f) & 1) ? 255u : 0u;
    }
    const int bp32 = lane_bperm32();
    // chunks of 60 steps (t0 is a multiple of 6, so is every chunk start: the phases inside a chunk are compile-time)
    unsigned sy_next = sy.chunk(t0, T);
    for (unsigned tc = t0; tc < t1; tc += 60) {
        const unsigned sy60 = sy_next;
        sy_next = sy.chunk(tc + 60, T);                     // (clamped to the last step inside) in flight while this chunk's 60 steps run
        const unsigned cn = t1 - tc < 60 ? t1 - tc : 60;
        int wlo = 0, whi = 0;                               // lane k: the decision word of step tc + k
        // (a taken branch costs more than the step it guards: whole groups of six run unguarded, the last partial group apart)
#define VIT_STEP(R1, J)                                                                                        \
            {                                                                                              \
                const unsigned v = (unsigned)__builtin_amdgcn_readlane((int)sy60, (int)(k + J));           \
                const unsigned sa = v & 0xffu, sb = v >> 8;                                                \
                bool tk;                                                                                   \
                pm = vit_step<R1>(pm, sa, sb, sa ^ 255u, sb ^ 255u, ma[R1], mb[R1], s, bp32, tk);          \
                if constexpr (DEC) {                                                                       \
                    const unsigned long long w = __ballot(tk);                                             \
                    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(wlo) : "s"((int)(unsigned)w), "s"((int)(k + J)) : "m0");         \
                    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(whi) : "s"((int)(unsigned)(w >> 32)), "s"((int)(k + J)) : "m0"); \
                }                                                                                          \
            }
        unsigned k = 0;
        for (; k + 6 <= cn; k += 6) { VIT_STEP(1, 0) VIT_STEP(2, 1) VIT_STEP(3, 2) VIT_STEP(4, 3) VIT_STEP(5, 4) VIT_STEP(0, 5) }
        if (k < cn) {
            VIT_STEP(1, 0)
            if (k + 1 < cn) VIT_STEP(2, 1)
            if (k + 2 < cn) VIT_STEP(3, 2)
            if (k + 3 < cn) VIT_STEP(4, 3)
            if (k + 4 < cn) VIT_STEP(5, 4)
        }
#undef VIT_STEP
        if (DEC && s < cn) dec[tc - t0 + s] = ((unsigned long long)(unsigned)whi << 32) | (unsigned)wlo;
    }
    return pm;
}
// n decoded bytes from 2 (8 n + 6) symbols; `ckpt`: >= 128 * ceil(T / VIT_B) bytes of scratch in HBM; `lds`: VIT_B x 8 bytes
__device__ void conv27_decode_wave(const VitSym sy, unsigned n_, uint8_t *dec, uint16_t *ckpt, unsigned long long *lds)
{
    const int s = lane_id();
    const unsigned n = (unsigned)__builtin_amdgcn_readfirstlane((int)n_);    // wave-uniform: the step loops branch on the scalar unit
    const unsigned T = 8 * n + 6, nblk = (T + VIT_B - 1) / VIT_B;
    int pm = s ? (1 << 20) : 0;                             // the encoder starts in state 0
    for (unsigned b = 0; b < nblk; b++) {                   // checkpoints: normalised metrics at the start of every block
        int mn = pm;
#pragma unroll
        for (int h = 32; h >= 1; h >>= 1) { const int o = __shfl_xor(mn, h, WV); mn = o < mn ? o : mn; }
        pm -= mn; if (pm > 0xffff) pm = 0xffff;             // (only the unreachable states of the first steps saturate)
        ckpt[64 * b + s] = (uint16_t)pm;
        const unsigned t1 = (b + 1) * VIT_B < T ? (b + 1) * VIT_B : T;
        pm = vit_forward<false>(sy, b * VIT_B, t1, T, pm, nullptr);
    }
    unsigned state = 0;                                     // the tail bits return the encoder to state 0
    for (unsigned b = nblk; b-- > 0;) {
        const unsigned t0 = b * VIT_B, t1 = t0 + VIT_B < T ? t0 + VIT_B : T;
        (void)vit_forward<true>(sy, t0, t1, T, (int)ckpt[64 * b + s], lds);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // trace back through the block in chunks of 64 steps (aligned: a chunk owns whole bytes of the message): lane k
        // holds the decision word of step c0 + k
        for (unsigned c1 = t1; c1 > t0;) {
            const unsigned c0 = (c1 - 1) & ~63u, cn = c1 - c0;
            const unsigned long long w = (unsigned)s < cn ? lds[c0 - t0 + s] : 0ull;
            unsigned r1 = (c0 + cn) % 6;                    // (t + 1) mod 6 of the chunk's last step
            const unsigned wlo = (unsigned)w, whi = (unsigned)(w >> 32);
            // `state` before VIT_BACK(K) = the encoder state after step c0 + K: input bits of steps K, K-1 .. K-5 from bit 0 up.
            // The lane that held it (see vit_forward) is its rotation right by r1 = the low six bits of (state * 65) >> r1.
#define VIT_BACK(K)                                                                                            \
            {                                                                                              \
                const unsigned long long wk = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)whi, (int)(K)) << 32) | \
                                              (unsigned)__builtin_amdgcn_readlane((int)wlo, (int)(K));      \
                const unsigned ln = ((state * 65u) >> r1) & 63u;                                            \
                state = (state >> 1) | ((unsigned)((wk >> ln) & 1ull) << 5);                               \
                r1 = r1 ? r1 - 1u : 5u;                                                                    \
            }
            unsigned k = cn;
            while (k & 7u) { k--; VIT_BACK(k) }             // (the tail steps 8 n .. 8 n + 5: no message bits, and only the last chunk has them)
            unsigned long long bytes = 0;                   // byte c0 / 8 + i of the message at bits 8 i ..
            while (k) {                                     // eight steps = one byte (step t -> byte t / 8, bit 7 - t % 8), read off the state twice
                k -= 8;
                const unsigned s1 = state;                  // inputs of steps k+7 (bit 0) .. k+2 (bit 5)
                VIT_BACK(k + 7) VIT_BACK(k + 6) VIT_BACK(k + 5) VIT_BACK(k + 4) VIT_BACK(k + 3) VIT_BACK(k + 2)
                const unsigned s2 = state;                  // inputs of steps k+1 (bit 0), k (bit 1)
                VIT_BACK(k + 1) VIT_BACK(k)
                bytes |= (unsigned long long)(((s2 & 3u) << 6) | (s1 & 63u)) << k;
            }
#undef VIT_BACK
            const unsigned by = (c0 >> 3) + (unsigned)s;
            if (s < 8 && by < n) dec[by] = (uint8_t)(bytes >> (8 * (unsigned)s));
            c1 = c0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// Hamming(7,4) codeword p1 p2 d1 p4 d2 d3 d4 (MSB first); Hamming(8,4) = the same with the overall parity as the LSB
__device__ __forceinline__ unsigned hsmall_enc_d(unsigned s, unsigned nb)
{
    const unsigned d1 = (s >> 3) & 1, d2 = (s >> 2) & 1, d3 = (s >> 1) & 1, d4 = s & 1;
    const unsigned c = ((d1 ^ d2 ^ d4) << 6) | ((d1 ^ d3 ^ d4) << 5) | (d1 << 4) | ((d2 ^ d3 ^ d4) << 3) | (d2 << 2) | (d3 << 1) | d4;
    return nb == 7 ? c : ((c << 1) | ((unsigned)__popc(c) & 1u));
}
// hard decode `enc` -> `dec` (dec_len bytes), lanes in parallel
__device__ void fec_decode_hard(unsigned fs, unsigned n, const uint8_t *enc, uint8_t *dec)
{
    const int l = lane_id();
    if (fs == 6) {
        for (unsigned i = (unsigned)l; i < n; i += WV) {
            unsigned o = 3 * (i / 2), m;
            if ((i & 1) == 0) m = ((unsigned)enc[o] << 4) | ((unsigned)enc[o + 1] >> 4);
            else              m = (((unsigned)enc[o + 1] & 0x0f) << 8) | (unsigned)enc[o + 2];
            dec[i] = (uint8_t)h128_dec_sym(m);
        }
    } else if (fs == 7) {
        const unsigned G = n / 3, r = n % 3;
        for (unsigned g = (unsigned)l; g < G; g += WV) {
            const uint8_t *e = enc + 6 * g;
            unsigned s0 = golay_dec_sym(((unsigned)e[0] << 16) | ((unsigned)e[1] << 8) | e[2]);
            unsigned s1 = golay_dec_sym(((unsigned)e[3] << 16) | ((unsigned)e[4] << 8) | e[5]);
            dec[3 * g]     = (uint8_t)((s0 >> 4) & 0xff);
            dec[3 * g + 1] = (uint8_t)(((s0 << 4) & 0xf0) | ((s1 >> 8) & 0x0f));
            dec[3 * g + 2] = (uint8_t)(s1 & 0xff);
        }
        if ((unsigned)l < r) {
            const uint8_t *e = enc + 6 * G + 3 * l;
            dec[3 * G + l] = (uint8_t)(golay_dec_sym(((unsigned)e[0] << 16) | ((unsigned)e[1] << 8) | e[2]) & 0xff);
        }
    } else if (fs == 2) {                                   // rep3: bitwise majority of the three copies
        for (unsigned i = (unsigned)l; i < n; i += WV) {
            const unsigned s0 = enc[i], s1 = enc[i + n], s2 = enc[i + 2 * n];
            dec[i] = (uint8_t)((s0 & s1) | (s0 & s2) | (s1 & s2));
        }
    } else if (fs == 3) {                                   // rep5: bitwise majority of five (bit-sliced 3-bit counter)
        for (unsigned i = (unsigned)l; i < n; i += WV) {
            unsigned c0 = 0, c1 = 0, c2 = 0;
#pragma unroll
            for (unsigned r = 0; r < 5; r++) {
                const unsigned v = enc[i + r * n], k0 = c0 & v;
                c0 ^= v; const unsigned k1 = c1 & k0; c1 ^= k0; c2 |= k1;
            }
            dec[i] = (uint8_t)(c2 | (c1 & c0));                 // count >= 3: 4 or 5 (c2), or 3 (c1 & c0)
        }
    } else if (fs == 4 || fs == 5) {                        // Hamming(7,4) / (8,4): nearest codeword, symbols ascending, strict <
        const unsigned nb = fs == 4 ? 7u : 8u;
        for (unsigned i = (unsigned)l; i < n; i += WV) {
            unsigned by = 0;
#pragma unroll
            for (unsigned h = 0; h < 2; h++) {
                const unsigned k = 2 * nb * i + nb * h;             // bit index of the symbol, MSB first
                const unsigned w16 = ((unsigned)enc[k >> 3] << 8) | ((k >> 3) + 1 < fec_enc_len_d(fs, n) ? (unsigned)enc[(k >> 3) + 1] : 0u);
                const unsigned w = (w16 >> (16 - nb - (k & 7))) & ((1u << nb) - 1u);
                unsigned best = 0, dmin = 99;
#pragma unroll
                for (unsigned sy = 0; sy < 16; sy++) {
                    const unsigned d = (unsigned)__popc(w ^ hsmall_enc_d(sy, nb));
                    if (d < dmin) { dmin = d; best = sy; }
                }
                by = (by << 4) | best;
            }
            dec[i] = (uint8_t)by;
        }
    } else {
        for (unsigned i = (unsigned)l; i < n; i += WV) dec[i] = enc[i];
    }
    __syncthreads();
}
// soft decision for the short block codes (liquid fec_rep3_decode_soft / fec_rep5_decode_soft / fec_hamming74_decode_soft /
// fec_hamming84_decode_soft): `soft` = 8 soft bits per coded byte, de-interleaved; lanes in parallel over the message bytes
__device__ void fec_decode_soft_small(unsigned fs, unsigned n, const uint8_t *soft, uint8_t *dec)
{
    const int l = lane_id();
    if (fs == 2 || fs == 3) {                               // mean of the copies (integer division) against the erasure level 127
        const unsigned R = fs == 2 ? 3u : 5u;
        for (unsigned i = (unsigned)l; i < n; i += WV) {
            unsigned b = 0;
#pragma unroll
            for (unsigned k = 0; k < 8; k++) {
                unsigned sum = 0;
                for (unsigned r = 0; r < R; r++) sum += soft[8 * ((size_t)i + (size_t)r * n) + k];
                b = (b << 1) | ((sum / R) > 127u ? 1u : 0u);
            }
            dec[i] = (uint8_t)b;
        }
    } else {                                                // least soft distance over all 16 codewords, ascending, strict <
        const unsigned nb = fs == 4 ? 7u : 8u;
        for (unsigned i = (unsigned)l; i < n; i += WV) {
            unsigned by = 0;
#pragma unroll
            for (unsigned h = 0; h < 2; h++) {
                const uint8_t *sb = soft + 2 * (size_t)nb * i + nb * h;
                unsigned v[8];
#pragma unroll
                for (unsigned k = 0; k < 8; k++) v[k] = k < nb ? (unsigned)sb[k] : 0u;
                unsigned best = 0, dmin = 0;
#pragma unroll
                for (unsigned sy = 0; sy < 16; sy++) {
                    const unsigned c = hsmall_enc_d(sy, nb);
                    unsigned d = 0;
#pragma unroll
                    for (unsigned k = 0; k < 8; k++) if (k < nb) d += ((c >> (nb - 1 - k)) & 1u) ? 255u - v[k] : v[k];
                    if (sy == 0 || d < dmin) { dmin = d; best = sy; }
                }
                by = (by << 4) | best;
            }
            dec[i] = (uint8_t)by;
        }
    }
    __syncthreads();
}
// slice 8 soft bits per byte at 127 and pack MSB first
__device__ void soft_pack(const uint8_t *soft, unsigned nbytes, uint8_t *out, bool unscramble)
{
    const int l = lane_id();
    for (unsigned i = (unsigned)l; i < nbytes; i += WV) {
        unsigned long long w = *reinterpret_cast<const unsigned long long *>(soft + 8 * (size_t)i);
        unsigned b = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) b = (b << 1) | ((((unsigned)(w >> (8 * k)) & 0xff) > 127) ? 1u : 0u);
        if (unscramble) { const unsigned msk[4] = { 0xb4, 0x6a, 0x8b, 0xc5 }; b ^= msk[i & 3]; }
        out[i] = (uint8_t)b;
    }
    __syncthreads();
}

// CRC -> fec0 -> il -> fec1 -> il, inverted.  `soft` holds 8 soft bits per packet byte
// (8-byte aligned).  Result message in tmpb[0..n_msg); returns validity.
__device__ bool packet_decode(const CodingDev cod, bool soft_mode, bool scrambled, unsigned n_msg,
                              unsigned crc, unsigned fec0, unsigned fec1,
                              uint8_t *soft, uint8_t *tmpa, uint8_t *tmpb, unsigned long long *vit_lds = nullptr, unsigned ablate = 0,
                              bool pre_done = false)      // pre_done: soft de-interleaved and Viterbi-decoded into tmpa already (decode_kernel; decode_general_kernel's frame-per-wave decoder)
{
    const int l = lane_id();
    const unsigned crc_len = (crc == 6) ? 4u : 0u;
    const unsigned n0 = n_msg + crc_len;
    const unsigned e0 = fec_enc_len_d(fec0, n0), e1 = fec_enc_len_d(fec1, e0);
    const unsigned d0 = fec_depth_d(fec0), d1 = fec_depth_d(fec1);
    if ((fec0 == 11 || fec1 == 11) && !vit_lds) return false;      // (every launch that can get here provides the scratch)
    if (soft_mode && fec1 == 6) {
        if (!(ablate & 8)) deinterleave<true>(soft, e1, d1);
        if (!(ablate & 16)) for (unsigned i = (unsigned)l; i < e0; i += WV) tmpa[i] = (uint8_t)h128_dec_soft_fast(soft + 12 * (size_t)i);
        __syncthreads();
    } else if (soft_mode && fec1 == 11 && pre_done) {
        // (nothing to do here)
    } else if (soft_mode && fec1 == 11) {
        if (!(ablate & 64)) deinterleave<true>(soft, e1, d1);
        if (!(ablate & 128)) conv27_decode_wave(VitSym{ soft, false }, e0, tmpa, reinterpret_cast<uint16_t *>(tmpb), vit_lds);   // checkpoints: 128 * ceil((8 e0 + 6) / 960) bytes <= max_enc_len + 16 (mcrx_hip_create keeps max_enc_len >= 256)
        __syncthreads();
    } else if (soft_mode && fec1 >= 2 && fec1 <= 5) {
        deinterleave<true>(soft, e1, d1);
        fec_decode_soft_small(fec1, e0, soft, tmpa);
    } else {
        if (soft_mode) deinterleave<true>(soft, e1, d1);
        soft_pack(soft, e1, tmpb, scrambled);
        if (!soft_mode) deinterleave<false>(tmpb, e1, d1);
        if (fec1 == 11) { conv27_decode_wave(VitSym{ tmpb, true }, e0, tmpa, reinterpret_cast<uint16_t *>(soft), vit_lds); __syncthreads(); }   // (the soft bits are spent)
        else fec_decode_hard(fec1, e0, tmpb, tmpa);
    }
    deinterleave<false>(tmpa, e0, d0);
    if (fec0 == 11) { conv27_decode_wave(VitSym{ tmpa, true }, n0, tmpb, reinterpret_cast<uint16_t *>(soft), vit_lds); __syncthreads(); }
    else fec_decode_hard(fec0, n0, tmpa, tmpb);
    if (crc_len == 0 || (ablate & 32)) return true;
    uint32_t key = ((uint32_t)tmpb[n_msg] << 24) | ((uint32_t)tmpb[n_msg + 1] << 16) |
                   ((uint32_t)tmpb[n_msg + 2] << 8) | (uint32_t)tmpb[n_msg + 3];
    return crc32_wave(cod, tmpb, n_msg) == key;
}

// The same behind a call boundary, for the per-channel scout: it decodes a packet itself only when a frame straddles
// pushes, and its kernel is already at the register file's limit (256 VGPRs + AGPR and scratch spills) -- inlined
// there, the decoder's live ranges land in the middle of that pressure (and a build of it came out
// mis-scheduled: right bytes, wrong CRC verdict, depending on unrelated edits).
__device__ __attribute__((noinline)) bool packet_decode_call(const CodingDev *cod, bool soft_mode, bool scrambled, unsigned n_msg,
                                                             unsigned crc, unsigned fec0, unsigned fec1,
                                                             uint8_t *soft, uint8_t *tmpa, uint8_t *tmpb, unsigned long long *vit_lds)
{
    return packet_decode(*cod, soft_mode, scrambled, n_msg, crc, fec0, fec1, soft, tmpa, tmpb, vit_lds);
}

// ------------------------------------------------------------------ modem
__device__ __forceinline__ unsigned gray_dec_d(unsigned x) { unsigned y = x; while (x >>= 1) y ^= x; return y; }
__device__ __forceinline__ cfd qam_point(unsigned sym, unsigned mq, float alpha)
{
    int L = 1 << mq;
    int gi = 2 * (int)gray_dec_d(sym >> mq) - L + 1, gq = 2 * (int)gray_dec_d(sym & (unsigned)(L - 1)) - L + 1;
    return make_float2((float)gi * alpha, (float)gq * alpha);
}
__device__ __forceinline__ unsigned qam_slice(float v, unsigned b, float alpha)
{
    unsigned s = 0;
    for (unsigned k = b; k > 0; k--) {
        float ref = (float)(1u << (k - 1)) * alpha;
        s <<= 1;
        if (v > 0) { s |= 1; v -= ref; } else v += ref;
    }
    return s;
}
__device__ __forceinline__ uint8_t soft_clamp(float v)
{ int sb = (int)v; sb = sb > 255 ? 255 : sb; sb = sb < 0 ? 0 : sb; return (uint8_t)sb; }

// soft demodulate r; writes bps soft bits (MSB first), returns hard symbol
// `nbt`: the modem's nearest-neighbour table (cod.qam16_nb / cod.qam64_nb, or a copy of it in LDS)
__device__ __forceinline__ unsigned demod_soft(const uint8_t *nbt, unsigned mod, cfd r, uint8_t *soft)
{
    if (mod == 39) {
        soft[0] = soft_clamp((-2.0f * r.x * 4.0f) * 16.0f + 127.0f);
        return r.x > 0 ? 0u : 1u;
    }
    if (mod == 40) {
        soft[0] = soft_clamp((-2.0f * r.y * 5.8f) * 16.0f + 127.0f);
        soft[1] = soft_clamp((-2.0f * r.x * 5.8f) * 16.0f + 127.0f);
        return (r.x > 0 ? 0u : 1u) + (r.y > 0 ? 0u : 2u);
    }
    const unsigned bps = (mod == 27) ? 4u : 6u, mq = bps / 2;
    const float alpha = (mod == 27) ? 0.31622776601683794f : 0.1543033499620919f;
    unsigned si = qam_slice(r.x, mq, alpha), sq = qam_slice(r.y, mq, alpha);
    unsigned s = ((si ^ (si >> 1)) << mq) + (sq ^ (sq >> 1));
    const float gamma = 1.2f * (float)(1u << bps);
    float dmin0[6], dmin1[6];
#pragma unroll
    for (int k = 0; k < 6; k++) { dmin0[k] = 8.0f; dmin1[k] = 8.0f; }
    // constellation point of a symbol: Gray decode of <= 3 bits per axis is two shifts and two xors
    auto point = [&](unsigned sym) {
        const unsigned hi = sym >> mq, lo = sym & ((1u << mq) - 1u);
        const int gi = 2 * (int)(hi ^ (hi >> 1) ^ (hi >> 2)) - (1 << mq) + 1, gq = 2 * (int)(lo ^ (lo >> 1) ^ (lo >> 2)) - (1 << mq) + 1;
        return make_float2((float)gi * alpha, (float)gq * alpha);
    };
    cfd xh = point(s);
    float dr = r.x - xh.x, di = r.y - xh.y, d = dr * dr + di * di;
#pragma unroll
    for (int k = 0; k < 6; k++) if ((unsigned)k < bps) { if ((s >> (bps - k - 1)) & 1) dmin1[k] = d; else dmin0[k] = d; }
    const uint32_t nb4 = reinterpret_cast<const uint32_t *>(nbt)[s];       // the four neighbours in one word
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const unsigned nb = (nb4 >> (8 * i)) & 0xffu;
        xh = point(nb);
        dr = r.x - xh.x; di = r.y - xh.y; d = dr * dr + di * di;
#pragma unroll
        for (int k = 0; k < 6; k++) if ((unsigned)k < bps) {
            const bool one = ((nb >> (bps - k - 1)) & 1u) != 0;
            dmin1[k] = (one && d < dmin1[k]) ? d : dmin1[k];
            dmin0[k] = (!one && d < dmin0[k]) ? d : dmin0[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 6; k++) if ((unsigned)k < bps) soft[k] = soft_clamp(((dmin0[k] - dmin1[k]) * gamma) * 16.0f + 127.0f);
    return s;
}

// ------------------------------------------------------------------ the walker
// Wave-private LDS scratch, referenced by name so every access is a ds_* instruction.
#define SY_MAXM 1024
extern __shared__ __attribute__((aligned(16))) float2 sy_lds[];
#define sy_w  (sy_lds + lo)                                                     /* this wave's scratch (Walker::lo: 0 in one-wave workgroups) */
#define ldsc  sy_w                                                            /* complex scratch [M]   */
#define ldsf  (reinterpret_cast<float *>(sy_w + c.M))                         /* float scratch  [2*M]  */
#define ldspf (reinterpret_cast<float *>(sy_w + c.M) + 2 * c.M)               /* pilot fit rows [M]    */
#define ldsps (reinterpret_cast<uint8_t *>(reinterpret_cast<float *>(sy_w + c.M) + 3 * c.M))  /* pilot bits [256] */
#define ldshm (reinterpret_cast<uint16_t *>(ldsps + 256))                      /* header bit map [288]  */
#define ldshb (ldsps + 256 + 2 * MCRX_HDR_SYMS)                                /* header bits, decoded order [288] */
#define ldshd (reinterpret_cast<uint16_t *>(ldsps + 256 + 3 * MCRX_HDR_SYMS))   /* Golay-decoded 12-bit words [12] */
#define ldsad (reinterpret_cast<uint32_t *>(ldsps + 256 + 3 * MCRX_HDR_SYMS + 32)) /* job list entries of the adopted frames [MCRX_SPEC_MAX] */
#define ldsqn (ldsps + 256 + 3 * MCRX_HDR_SYMS + 32 + 4 * MCRX_SPEC_MAX)        /* QAM neighbour table of the frame's modem [256] */
#define SY_LDS_BYTES(M) ((((size_t)(M) * 8 + (size_t)(M) * 12 + 256 + 3 * MCRX_HDR_SYMS + 32 + 4 * MCRX_SPEC_MAX + 256) + 15) & ~(size_t)15)

// One wavefront per workgroup: LDS traffic of a wave is processed in order, so a compiler-level
// fence is all the hand-off between lanes needs (no s_barrier, and no vmcnt(0) drain of the
// outstanding global stores that __syncthreads() would add).
__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// slot key: a SEEK state is fully described by (next sample, timer) -- every path into SEEK resets the
// rest -- so a slot is keyed by both, packed (timer in bits 48..60 -- it stays below 2 (M + cp) --, positions stay below
// 2^48).  Bits 61..63 carry the synchronizer state of the one key that is not a SEEK state: the state a launch starts in when
// the previous push ended in the middle of an acquisition (segment 0 clones it whole; nothing else can produce that key).
__device__ __forceinline__ int64_t spec_key(int64_t pos, uint32_t timer, int state = 0)
{ return (int64_t)(((uint64_t)(unsigned)state << 61) | ((uint64_t)(timer & 0x1FFFu) << 48) | ((uint64_t)pos & 0xFFFFFFFFFFFFull)); }

// what a Walker instance is compiled for: everything (general configurations, and the tail kernel that walks the
// payload of a frame straddling two pushes), a speculative acquisition wave, or the lean per-channel scout
// (acquisition + header only: a frame whose payload does not fit the buffer is left to the tail kernel)
enum { SYM_FULL = 0, SYM_SPEC = 1, SYM_LEAN = 2 };

template <int E>
struct Walker {
    const SyncArgs &a;
    const SyncConsts &c;
    const int l;                // lane
    const uint32_t ch;          // channel within shard
    const int lo;               // this wave's LDS scratch, in float2 from sy_lds (workgroups of several waves: sync_walk_kernel)
    ChanState s;                // working copy of the channel state (wave uniform)
    // per (lane, e) constants
    int k[E];                   // subcarrier held after the FFT (-1: none)
    uint8_t sct[E];
    float S0v[E], S1v[E], fx[E];
    int drank[E], prank[E], erank[E];
    float2 R[E];
    float2 twx[6];              // cross-lane stage twiddles, stage h = 32 >> s
    // working buffers: per channel (scout / serial walk) or per job (payload worker)
    uint8_t *bsoft, *btmpa, *btmpb, *bhbits;
    float2 *bsyms, *bR;
    long long pre_off;          // >= 0: record space already reserved in the payload arena (payload worker)
    unsigned long long pre_soff;    // ... its framesyms in the symbol arena
    uint32_t pre_idx;           // ... and its record slot
    int64_t handoff_last;       // scout: last event index of the frame just handed off
    uint32_t handoff_job;       // segment wave: its job list entry
    uint32_t jres;              // scout: job slot reserved at frame detection (0xFFFFFFFF: none)
    SpecSlot *slot;             // != nullptr: this wave acquires speculatively into this slot (no side effects elsewhere)
    int64_t pf_t; float2 pf_x[E];   // scout: lookahead window (first sample, raw samples)
    long long ph[6];                // MCRX_DEBUG=2: cycles per phase of the symbol events
    // lean path (power-of-two M >= 64, <= 64 pilots): butterfly twiddles / signs, ranks, fit rows
    bool fastp;
    float2 tw[6]; float sg[6]; int bp32;
    int dr[E], pr[E]; float fxr[E];
    float pf0, pf1;

    __device__ __forceinline__ Walker(const SyncArgs &a_, uint32_t ch_, int lo_ = 0)
        : a(a_), c(a_.c), l(lane_id()), ch(ch_), lo(lo_)
    {
        const size_t tstride = (size_t)c.max_enc_len + 16;
        bsoft = a.soft + (size_t)ch * 8 * c.max_enc_len;
        btmpa = a.tmpa + (size_t)ch * tstride; btmpb = a.tmpb + (size_t)ch * tstride;
        bhbits = a.hbits + (size_t)ch * MCRX_HDR_SYMS;
        bsyms = a.syms + (size_t)ch * c.max_syms;
        bR = a.R + (size_t)ch * c.M;
        pre_off = -1; handoff_last = 0; jres = 0xFFFFFFFFu; slot = nullptr; nadopted = 0; fastp = false; pf_t = INT64_MIN; for (int i = 0; i < 6; i++) ph[i] = 0;
    }
    // payload worker: take over the synchronizer state of the job and reserve the record space
    // (payload bytes, then framesyms) in the frame arena; false if the arena is exhausted
    __device__ __forceinline__ bool bind_job(uint32_t j, const PayloadJob &job)
    {
        const size_t tstride = (size_t)c.max_enc_len + 16;
        bsoft = a.jsoft + (size_t)j * 8 * c.max_enc_len;
        btmpa = a.jtmp + (size_t)j * 2 * tstride; btmpb = btmpa + tstride;
        bR = a.jR + (size_t)j * c.M;
        s = job.s;
        const unsigned long long off = job.arena_off;             // set by place_jobs_kernel
        if (off == ~0ull) return false;
        pre_off = (long long)off; pre_soff = job.syms_off; pre_idx = job.pad;
        bsyms = reinterpret_cast<float2 *>(a.sarena + job.syms_off);
        return true;
    }

    __device__ __forceinline__ float2 sample(int64_t t) const
    {
        if (t < 0 || t < a.buf_first) return make_float2(0.f, 0.f);
        const int64_t r = t - a.buf_first;
        return a.chan[((size_t)(r >> MCRX_TILE_SH) * a.chan_stride + a.chan_off + ch) * MCRX_TILE_S + (size_t)(r & (MCRX_TILE_S - 1))];
    }
    __device__ __forceinline__ float2 mixed(int64_t t) const
    {
        float2 v = sample(t);
        if (t >= s.nco_t_ref) v = mix_down_hw(v, s.nco_theta_ref + (uint32_t)(t - s.nco_t_ref) * s.nco_dtheta);
        return v;
    }
    __device__ __forceinline__ void init_consts()
    {
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int i = l + WV * e;
            int kk = -1;
            if (i < c.M) kk = (E == 1 && c.M == 48 && c.M_pilot <= WV) ? lean::lane_k<48>(i) : (c.log2M ? (int)(__brev((unsigned)i) >> (32 - c.log2M)) : i);
            k[e] = kk;
            const int kq = kk < 0 ? 0 : kk;
            sct[e] = kk < 0 ? (uint8_t)0 : c.sctype[kq];
            S0v[e] = kk < 0 ? 0.f : c.S0[kq];
            S1v[e] = kk < 0 ? 0.f : c.S1[kq];
            drank[e] = kk < 0 ? -1 : c.data_rank[kq];
            prank[e] = kk < 0 ? -1 : c.pilot_rank[kq];
            erank[e] = kk < 0 ? -1 : c.en_rank[kq];
            fx[e] = (kq > c.M2) ? (float)kq - (float)c.M : (float)kq;
            R[e] = kk < 0 ? make_float2(0.f, 0.f) : bR[kq];
        }
#pragma unroll
        for (int st = 0; st < 6; st++) {
            const int h = 32 >> st;
            float sn, cs; sincos_u32((uint32_t)(l & (h - 1)) * (uint32_t)(0x80000000u / (unsigned)h), sn, cs);
            twx[st] = make_float2(cs, -sn);
        }
        // per-symbol tables into LDS (they sit on the symbol loop's dependency chain)
        // (word-wise, every request issued before the first LDS store: one round trip instead of one per chunk;
        //  the 255-byte pilot table sits in an allocation of at least 256)
        {
            const uint32_t wps = reinterpret_cast<const uint32_t *>(c.pilot_seq)[l];
            uint32_t whm[3];
#pragma unroll
            for (int u = 0; u < 3; u++) { const int i = l + WV * u; whm[u] = reinterpret_cast<const uint32_t *>(c.hdr_map)[i < MCRX_HDR_SYMS / 2 ? i : 0]; }
            const int npf = 2 * c.M_pilot;
            float wpf[2];
#pragma unroll
            for (int u = 0; u < 2; u++) { const int i = l + WV * u; wpf[u] = c.Pfit[i < npf ? i : 0]; }
            reinterpret_cast<uint32_t *>(ldsps)[l] = wps;
#pragma unroll
            for (int u = 0; u < 3; u++) { const int i = l + WV * u; if (i < MCRX_HDR_SYMS / 2) reinterpret_cast<uint32_t *>(ldshm)[i] = whm[u]; }
#pragma unroll
            for (int u = 0; u < 2; u++) { const int i = l + WV * u; if (i < npf) ldspf[i] = wpf[u]; }
            for (int i = l + 2 * WV; i < npf; i += WV) ldspf[i] = c.Pfit[i];        // more than 64 pilots (M > 512)
        }
        if (fast_ok()) init_fast();
        wave_sync_lds();
    }
    // constants of the lean path (also the whole setup of a payload worker)
    __device__ __forceinline__ void init_fast()
    {
        fastp = true;
#pragma unroll
        for (int st = 0; st < 6; st++) {
            const int h = 32 >> st;
            const bool up = (l & h) != 0;
            const float rev = (float)(l & (h - 1)) * (0.5f / (float)h);
            tw[st] = up ? make_float2(__builtin_amdgcn_cosf(rev), -__builtin_amdgcn_sinf(rev)) : make_float2(1.f, 0.f);
            sg[st] = up ? -1.f : 1.f;
        }
        bp32 = (l ^ 32) << 2;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int kq = fast_k(l + WV * e), kk = kq < 0 ? 0 : kq;        // (kq < 0: a lane behind a 48-sample symbol -- holds zeros, owns nothing)
            k[e] = kq;
            sct[e] = kq < 0 ? (uint8_t)0 : c.sctype[kk];
            dr[e] = kq < 0 ? -1 : c.data_rank[kk]; pr[e] = kq < 0 ? -1 : c.pilot_rank[kk];
            fxr[e] = ((kk > c.M2) ? (float)kk - (float)c.M : (float)kk) * 0.15915494309189535f;   // revolutions per rad of slope
        }
        const int Mp = c.M_pilot;
        pf0 = (l < Mp) ? c.Pfit[l] : 0.f; pf1 = (l < Mp) ? c.Pfit[Mp + l] : 0.f;
    }
    template <int J>
    __device__ __forceinline__ void inlane_stage(float2 (&x)[E])
    {
        constexpr int h = WV * J;
#pragma unroll
        for (int e = 0; e < E; e++) {
            if constexpr (true) {
                if ((e & J) == 0 && e + J < E) {
                    const float2 u = x[e], v = x[e + J];
                    float sn, cs;
                    sincos_u32((uint32_t)((l + WV * e) & (h - 1)) * (uint32_t)(0x80000000u / (unsigned)h), sn, cs);
                    x[e] = cadd(u, v);
                    x[e + J] = cmul(csub(u, v), make_float2(cs, -sn));
                }
            }
        }
    }
    // forward DFT of x (time position i = l + 64 e) -> X[k[e]]
    __device__ __forceinline__ void fft(float2 (&x)[E])
    {
        if (fastp) { fast_fft(x); return; }
        if (c.log2M) {
            if constexpr (E >= 16) inlane_stage<8>(x);         // in-lane stages, h = 64 j
            if constexpr (E >= 8) inlane_stage<4>(x);
            if constexpr (E >= 4) inlane_stage<2>(x);
            if constexpr (E >= 2) inlane_stage<1>(x);
#pragma unroll
            for (int st = 0; st < 6; st++) {
                const int h = 32 >> st;
                if (h < c.M) {
                    const bool up = (l & h) != 0;
#pragma unroll
                    for (int e = 0; e < E; e++) {
                        float2 p = make_float2(__shfl_xor(x[e].x, h, WV), __shfl_xor(x[e].y, h, WV));
                        x[e] = up ? cmul(csub(p, x[e]), twx[st]) : cadd(x[e], p);
                    }
                }
            }
        } else {
            wave_sync_lds();
#pragma unroll
            for (int e = 0; e < E; e++) { const int i = l + WV * e; if (i < c.M) ldsc[i] = x[e]; }
            wave_sync_lds();
#pragma unroll
            for (int e = 0; e < E; e++) {
                float2 acc = make_float2(0.f, 0.f);
                if (k[e] >= 0) {
                    int idx = 0;
                    for (int n = 0; n < c.M; n++) {
                        acc = cadd(acc, cmul(ldsc[n], c.dft_tw[idx]));
                        idx += k[e]; if (idx >= c.M) idx -= c.M;
                    }
                }
                x[e] = acc;
            }
            wave_sync_lds();
        }
    }
    __device__ __forceinline__ void load_raw_direct(int64_t t_start, float2 (&x)[E])
    {
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int i = l + WV * e;
            x[e] = (i < c.M) ? sample(t_start + i) : make_float2(0.f, 0.f);
        }
    }
    // Scout window fetch with a one-window lookahead.  The state machine is a serial chain of
    // HBM round trips; where the next event is usually one stride on (seek: +M, S0a -> S0b: +M/2,
    // symbols: +M+cp) its window is requested now and picked up from registers by the next event.
    // The lookahead load is branch free (clamped addresses): a load under a branch is waited for at the join.
    __device__ __forceinline__ void load_raw(int64_t t_start, float2 (&x)[E])
    {
        if (t_start == pf_t) {
#pragma unroll
            for (int e = 0; e < E; e++) x[e] = (l + WV * e < c.M) ? pf_x[e] : make_float2(0.f, 0.f);
        } else load_raw_direct(t_start, x);
        const int stride = s.state == SY_RX ? c.L : (s.state == SY_SEEK ? c.M : c.M2);
        const int64_t tn = t_start + stride;
        const int64_t rn = tn - a.buf_first, len = a.end - a.buf_first;
        const bool ok = a.scout != 0 && rn >= 0 && rn + c.M <= len;
        const int64_t rc = ok ? rn : 0;
#pragma unroll
        for (int e = 0; e < E; e++) {
            int64_t r = rc + l + WV * e;
            r = r < len ? r : len - 1;
            pf_x[e] = a.chan[((size_t)(r >> MCRX_TILE_SH) * a.chan_stride + a.chan_off + ch) * MCRX_TILE_S + (size_t)(r & (MCRX_TILE_S - 1))];
        }
        pf_t = ok ? tn : INT64_MIN;
    }
    __device__ __forceinline__ void mix_window(int64_t t_start, float2 (&x)[E])
    {
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int64_t t = t_start + l + WV * e;
            if (t >= s.nco_t_ref) x[e] = mix_down_hw(x[e], s.nco_theta_ref + (uint32_t)(t - s.nco_t_ref) * s.nco_dtheta);
        }
    }
    __device__ __forceinline__ void load_window(int64_t t_start, bool mix, float2 (&x)[E])
    {
        load_raw(t_start, x);
        if (mix) mix_window(t_start, x);
    }
    // S0 gain estimate + metric on the newest M samples ending at t_ev; returns s_hat (not scaled by g)
    __device__ __forceinline__ float2 s0_metric(int64_t t_ev, bool mix, float &power)
    {
        float2 x[E];
        load_window(t_ev - c.M + 1, mix, x);
        float pw = 0.f;
#pragma unroll
        for (int e = 0; e < E; e++) pw += x[e].x * x[e].x + x[e].y * x[e].y;
        power = wave_sum(pw);
        fft(x);
        const float gain = sqrtf((float)c.M_S0) / (float)c.M;
        wave_sync_lds();
#pragma unroll
        for (int e = 0; e < E; e++) if (k[e] >= 0) { x[e] = cscale(x[e], S0v[e] * gain); ldsc[k[e]] = x[e]; }
        wave_sync_lds();
        float2 acc = make_float2(0.f, 0.f);
#pragma unroll
        for (int e = 0; e < E; e++) if (k[e] >= 0 && (k[e] & 1) == 0) {
            int kn = k[e] + 2; if (kn >= c.M) kn -= c.M;
            acc = cadd(acc, cmulc(ldsc[kn], x[e]));
        }
        acc = wave_csum(acc);
        wave_sync_lds();
        return cscale(acc, 1.0f / (float)c.M_S0);
    }
    // S0 metric on a window that is already in registers (SEEK: the oscillator is not running yet)
    __device__ __forceinline__ float2 s0_metric_of(float2 (&x)[E], float &power)
    {
        float pw = 0.f;
#pragma unroll
        for (int e = 0; e < E; e++) pw += x[e].x * x[e].x + x[e].y * x[e].y;
        power = wave_sum(pw);
        fft(x);
        const float gain = sqrtf((float)c.M_S0) / (float)c.M;
        wave_sync_lds();
#pragma unroll
        for (int e = 0; e < E; e++) if (k[e] >= 0) { x[e] = cscale(x[e], S0v[e] * gain); ldsc[k[e]] = x[e]; }
        wave_sync_lds();
        float2 acc = make_float2(0.f, 0.f);
#pragma unroll
        for (int e = 0; e < E; e++) if (k[e] >= 0 && (k[e] & 1) == 0) {
            int kn = k[e] + 2; if (kn >= c.M) kn -= c.M;
            acc = cadd(acc, cmulc(ldsc[kn], x[e]));
        }
        acc = wave_csum(acc);
        wave_sync_lds();
        return cscale(acc, 1.0f / (float)c.M_S0);
    }
    // An idle channel is a chain of SEEK events M samples apart, each one HBM round trip plus a transform: SEEK_B of them
    // at a time, the windows requested together so that their latencies overlap, then the same events in the same order
    // (a detection ends the burst; the windows behind it are dropped).  Called in SEEK with timer == 0 and SEEK_B whole
    // events inside the buffer; the arithmetic of every event is sync_event()'s.
    static constexpr int SEEK_B = 4;
    __device__ __forceinline__ void seek_burst()
    {
        const int M = c.M, M2 = c.M2;
        float2 xw[SEEK_B][E];
#pragma unroll
        for (int j = 0; j < SEEK_B; j++) load_raw_direct(s.cur + (int64_t)j * M, xw[j]);
        pf_t = INT64_MIN;                               // (the one-window lookahead is stale behind a burst)
#pragma unroll
        for (int j = 0; j < SEEK_B; j++) {
            float pw; float2 sh = s0_metric_of(xw[j], pw);
            const float g = (float)M / pw;
            sh = cscale(sh, g);
            const float tau = atan2f(sh.y, sh.x) * (float)M2 / TWO_PI_F;
            s.g0 = g; s.timer = 0; s.cur += M;
            if (sqrtf(sh.x * sh.x + sh.y * sh.y) > c.detect_thresh) {
                const int dt = (int)roundf(tau);
                s.timer = (uint32_t)(M + dt) % (uint32_t)M2 + (uint32_t)M;
                s.state = SY_S0A;
                return;
            }
        }
    }
    __device__ __forceinline__ unsigned hbyte(int i) const { return (s.hw[i >> 2] >> (8 * (i & 3))) & 0xffu; }
    __device__ __forceinline__ void reset_framesync()
    {
        s.state = SY_SEEK; s.timer = 0; s.num_symbols = 0; s.pilot_count = 0;
        s.nco_theta_ref = 0; s.nco_dtheta = 0; s.nco_t_ref = 0;
        s.s_hat_0 = make_float2(0.f, 0.f); s.phi_prime = 0.f; s.p1_prime = 0.f;
        s.fstate = FX_HEADER; s.header_symbol_index = 0; s.payload_symbol_index = 0; s.evm_hat = 0.f;
    }

    __device__ __forceinline__ void emit(int64_t t_ev, bool with_payload, bool payload_valid, bool oversize = false, bool copy_payload = true)
    {
        if (slot) return;
        const uint32_t nsym = (with_payload && !oversize) ? s.mod_len : 0u;
        const uint32_t plen = (with_payload && !oversize) ? s.payload_len : 0u;
        const unsigned long long pbytes = ((unsigned long long)plen + 15ull) & ~15ull;
        const unsigned long long sbytes = 8ull * nsym;
        uint32_t idx = 0xFFFFFFFFu; unsigned long long off = 0, soff = 0;
        if (l == 0) {
            if (pre_off >= 0) { off = (unsigned long long)pre_off; soff = pre_soff; idx = pre_idx; }      // all placed by place_jobs_kernel
            else {
                off = atomicAdd(a.arena_used, pbytes);
                soff = atomicAdd(a.arena_used + 1, sbytes);
                if (off + pbytes <= a.arena_cap && soff + sbytes <= a.sarena_cap) idx = atomicAdd(a.nrec, 1u);
                else {      // no room: give the space back, so that later, smaller frames of this interval still fit
                    atomicAdd(a.arena_used, 0ull - pbytes);
                    atomicAdd(a.arena_used + 1, 0ull - sbytes);
                }
            }
            if (idx >= a.max_rec) { atomicAdd(a.nrec + 1, 1u); idx = 0xFFFFFFFFu; }
        }
        idx = (uint32_t)__shfl((int)idx, 0, WV);
        off = (unsigned long long)__shfl((long long)off, 0, WV);
        soff = (unsigned long long)__shfl((long long)soff, 0, WV);
        if ((a.debug & 16) && l == 0 && ch == 0) printf("[emit] ch %u t_ev %lld payload %d valid %d placed %d idx %u off %llu len %u\n", ch, (long long)t_ev, (int)with_payload, (int)payload_valid, (int)(pre_off >= 0), idx, off, plen);
        if (idx == 0xFFFFFFFFu) return;
        if (l == 0) {
            FrameRec r;
            r.channel = a.ch_first + ch; r.header_valid = s.header_valid; r.payload_valid = payload_valid ? 1 : 0;
            r.payload_len = plen;
            for (int i = 0; i < 8; i++) r.header[i] = (uint8_t)hbyte(i);
            r.evm = s.evm; r.rssi = -10.0f * log10f(s.g0);
            r.cfo = u32rad(s.nco_dtheta) / TWO_PI_F;
            r.mod_scheme = with_payload ? s.mod_scheme : 0u; r.mod_bps = with_payload ? s.bps : 0u;
            r.check = with_payload ? s.check : 0u; r.fec0 = with_payload ? s.fec0 : 0u; r.fec1 = with_payload ? s.fec1 : 0u;
            r.num_framesyms = nsym; r.end_sample = t_ev;
            r.payload_off = off; r.syms_off = soff;
            a.rec[idx] = r;
        }
        if (with_payload && !oversize) {
            const uint8_t *src = btmpb;
            uint8_t *dst = a.arena + off;
            if (copy_payload) for (uint32_t i = (uint32_t)l; i < plen; i += WV) dst[i] = src[i];
            if (pre_off < 0) {          // payload workers write framesyms straight into the record
                const float2 *ss = bsyms;
                float2 *ds = reinterpret_cast<float2 *>(a.sarena + soff);
                for (uint32_t i = (uint32_t)l; i < nsym; i += WV) ds[i] = ss[i];
            }
        }
    }

    // scout: hand the payload of the frame whose header was just decoded to a worker wave
    // if every payload symbol is already in the buffer.  Returns false to keep walking serially.
    // speculative wave: park the hand-off in the slot (R already sits in the slot's bR); no atomics, no records
    // segment wave: the hand-off goes into the job list at once, owner void (kernels.h, SpecSlot); no records, no channel state
    __device__ __forceinline__ bool try_handoff_spec(int64_t t_ev)
    {
        const int64_t nsym = (int64_t)((s.mod_len + (uint32_t)c.M_data - 1) / (uint32_t)c.M_data);
        const int64_t t_last = t_ev + nsym * (int64_t)c.L;
        if (s.enc_len > c.max_enc_len || s.mod_len > c.max_syms || s.payload_len > c.max_payload_len || t_last >= a.end) return false;
        const uint32_t j = park_state();
        if (j == 0xFFFFFFFFu) return false;
#pragma unroll
        for (int e = 0; e < E; e++) if (k[e] >= 0) a.jR[(size_t)j * c.M + k[e]] = R[e];
        handoff_job = j; handoff_last = t_last;
        return true;
    }
    // ... the synchronizer state into a job list entry nobody owns yet (the one reserved at S1, or a new one); ~0: the list is full
    // Job list entries come in blocks: one atomic on the launch's counter per a.seg_jobs frames of a wave instead of one per frame
    // (8192 waves queueing on one address cost the acquisition 30 us), requested when the wave starts, so that its round trip is
    // long over when the first frame is handed off.  What a wave leaves of its last block is voided (run_seg).
    uint32_t jblk_next = 0, jblk_end = 0;
    __device__ __forceinline__ void reserve_block()
    {
        const uint32_t nb = a.seg_jobs ? a.seg_jobs : 1u;
        uint32_t b = 0;
        if (l == 0) b = atomicAdd(a.njobs, nb);
        jblk_next = (uint32_t)__shfl((int)b, 0, WV); jblk_end = jblk_next + nb;
    }
    __device__ __forceinline__ uint32_t park_state()
    {
        if (jblk_next == jblk_end) reserve_block();
        const uint32_t j = jblk_next++;
        if (j >= a.max_jobs) return 0xFFFFFFFFu;
        if (l == 0) { PayloadJob jb; jb.s = s; jb.ch = 0xFFFFFFFFu; jb.pad = 0; jb.arena_off = 0; jb.syms_off = 0; a.jobs[j] = jb; }
        return j;
    }
    __device__ __forceinline__ bool try_handoff(int64_t t_ev)
    {
        if (!a.scout) return false;
        const int64_t nsym = (int64_t)((s.mod_len + (uint32_t)c.M_data - 1) / (uint32_t)c.M_data);
        const int64_t t_last = t_ev + nsym * (int64_t)c.L;

        if (s.enc_len > c.max_enc_len || s.mod_len > c.max_syms || s.payload_len > c.max_payload_len || t_last >= a.end) {
            void_reservation();
            return false;
        }
        uint32_t j = jres;
        if (j == 0xFFFFFFFFu) {             // frame detected in an earlier launch: no slot reserved yet
            if (l == 0) j = atomicAdd(a.njobs, 1u);
            j = (uint32_t)__shfl((int)j, 0, WV);
        }
        jres = 0xFFFFFFFFu;
        if (j >= a.max_jobs) return false;
#pragma unroll
        for (int e = 0; e < E; e++) if (k[e] >= 0) a.jR[(size_t)j * c.M + k[e]] = R[e];
        if (l == 0) {
            PayloadJob jb;
            jb.s = s; jb.ch = 0xFFFFFFFFu; jb.pad = 0; jb.arena_off = 0; jb.syms_off = 0;      // (owner and record space: place_owned)
            a.jobs[j] = jb;
        }
        if (l == 0) ldsad[nown] = j;            // (room for one: run() empties the list before it is full)
        nown++;
        handoff_last = t_last;
        return true;
    }
    // The job slot is requested when the frame is detected (S1), so the atomic's round trip hides
    // under the header symbols; a frame that ends up not handed off gives the slot back as void.
    __device__ __forceinline__ void reserve_job()
    {
        if (!a.scout || slot || jres != 0xFFFFFFFFu) return;       // (segment waves take their entries from blocks: park_state)
        uint32_t j = 0;
        if (l == 0) j = atomicAdd(a.njobs, 1u);
        jres = (uint32_t)__shfl((int)j, 0, WV);
    }
    __device__ __forceinline__ void void_block()
    {
        for (uint32_t q = jblk_next + (uint32_t)l; q < jblk_end; q += WV) if (q < a.max_jobs) a.jobs[q].ch = 0xFFFFFFFFu;
        jblk_next = jblk_end;
    }
    __device__ __forceinline__ void void_reservation()
    {
        if (jres != 0xFFFFFFFFu && jres < a.max_jobs && l == 0) a.jobs[jres].ch = 0xFFFFFFFFu;
        jres = 0xFFFFFFFFu;
    }

    // header complete: decode and configure the payload receiver
    __device__ __forceinline__ void decode_header()
    {
        uint8_t *soft = bsoft;
        uint8_t *ta = btmpa, *tb = btmpb;
        const uint8_t *hb = bhbits;
        __syncthreads();
        for (int i = l; i < MCRX_HDR_SYMS; i += WV) soft[i] = hb[i] ? 255 : 0;
        __syncthreads();
        bool ok = packet_decode_call(&c.cod, false, true, MCRX_HDR_DEC, 6, 7, 1, soft, ta, tb, nullptr);
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) if (4 * w + b < MCRX_HDR_DEC) v |= (uint32_t)tb[4 * w + b] << (8 * b);
            s.hw[w] = v;
        }
        header_fields(ok);
    }
    // decoded header bytes (s.hw) -> payload configuration
    __device__ __forceinline__ void header_fields(bool ok)
    {
        const unsigned proto = hbyte(8);
        const unsigned plen = (hbyte(9) << 8) | hbyte(10);
        const unsigned mod = hbyte(11);
        const unsigned check = (hbyte(12) >> 5) & 7, fec0 = hbyte(12) & 0x1f, fec1 = hbyte(13) & 0x1f;
        const unsigned bps = mod_bps_d(mod);
        if (proto != 104 || bps == 0 || !(check == 1 || check == 6) ||
            !fec_known_d(fec0) || !fec_known_d(fec1)) ok = false;
        s.header_valid = ok ? 1 : 0;
        if (ok) {
            s.payload_len = plen; s.mod_scheme = mod; s.bps = bps; s.check = check; s.fec0 = fec0; s.fec1 = fec1;
            const unsigned n0 = plen + (check == 6 ? 4u : 0u);
            s.enc_len = fec_enc_len_d(fec1, fec_enc_len_d(fec0, n0));
            const unsigned nb = 8 * s.enc_len;
            s.mod_len = nb / bps + ((nb % bps) ? 1u : 0u);
        }
    }

    // one received OFDM symbol X (equalised, de-rotated), flexible-frame level
    // returns 0: frame continues, 1: frame finished (synchronizer resets), 2: payload handed off
    __device__ __forceinline__ int flex_symbol(const float2 (&X)[E], int64_t t_ev)
    {
        uint8_t *hb = bhbits;
        if (s.fstate == FX_HEADER) {
            float ev = 0.f;
#pragma unroll
            for (int e = 0; e < E; e++) if (drank[e] >= 0) {
                const uint32_t idx = s.header_symbol_index + (uint32_t)drank[e];
                if (idx < MCRX_HDR_SYMS) {
                    const unsigned sym = X[e].x > 0 ? 0u : 1u;
                    hb[idx] = (uint8_t)sym;
                    const float xh = sym ? -1.0f : 1.0f;
                    const float dr = xh - X[e].x, di = -X[e].y;
                    const float evm = sqrtf(dr * dr + di * di);
                    ev += evm * evm;
                }
            }
            s.evm_hat += wave_sum(ev);
            s.header_symbol_index += (uint32_t)c.M_data;
            if (s.header_symbol_index >= MCRX_HDR_SYMS) {
                decode_header();
                s.evm = 10.0f * log10f(s.evm_hat / (float)MCRX_HDR_SYMS);
                if (s.header_valid) {
                    s.fstate = FX_PAYLOAD; s.payload_symbol_index = 0;
                    if (try_handoff(t_ev)) return 2;
                }
                else { emit(t_ev, false, false); return 1; }
            }
            return 0;
        }
        // payload
        const bool oversize = s.enc_len > c.max_enc_len || s.mod_len > c.max_syms || s.payload_len > c.max_payload_len;
        uint8_t *soft = bsoft;
        float2 *syms = bsyms;
        const uint32_t nbits = 8 * s.enc_len;
        if (!oversize) {
#pragma unroll
            for (int e = 0; e < E; e++) if (drank[e] >= 0) {
                const uint32_t idx = s.payload_symbol_index + (uint32_t)drank[e];
                if (idx < s.mod_len) {
                    syms[idx] = X[e];
                    uint8_t sb[6];
                    const unsigned hs = demod_soft(s.mod_scheme == 27 ? c.cod.qam16_nb : c.cod.qam64_nb, s.mod_scheme, X[e], sb);
                    for (unsigned kb = 0; kb < s.bps; kb++) {
                        const uint32_t pos = idx * s.bps + kb;
                        if (pos < nbits) soft[pos] = c.payload_soft ? sb[kb] : (uint8_t)(((hs >> (s.bps - 1 - kb)) & 1) ? 255 : 0);
                    }
                }
            }
        }
        s.payload_symbol_index += (uint32_t)c.M_data;
        if (s.payload_symbol_index >= s.mod_len) {
            bool valid = false;
            if (!oversize) {
                __syncthreads();
                valid = packet_decode_call(&c.cod, c.payload_soft != 0, false, s.payload_len, s.check, s.fec0, s.fec1, soft,
                                           btmpa, btmpb, a.vit_off ? reinterpret_cast<unsigned long long *>(reinterpret_cast<uint8_t *>(sy_lds) + a.vit_off) : nullptr);
            }
            emit(t_ev, true, valid, oversize);
            return 1;
        }
        return 0;
    }

    // one RXSYMBOLS event: FFT, equalise, pilot phase fit, de-rotate, NCO trim, frame level
    __device__ __forceinline__ int rx_event(int64_t t_ev)
    {
        float2 X[E];
        load_raw(t_ev - c.L + 1 + c.cp - c.backoff, X);
        return rx_core(t_ev, X);
    }
    // X: raw (unmixed) window samples of the symbol
    __device__ __forceinline__ int rx_core(int64_t t_ev, float2 (&X)[E])
    {
        const int L = c.L;
        mix_window(t_ev - L + 1 + c.cp - c.backoff, X);
        fft(X);
        // equalise, pilot phases
        float *yph = ldsf;
        wave_sync_lds();
#pragma unroll
        for (int e = 0; e < E; e++) {
            X[e] = cmul(X[e], R[e]);
            if (prank[e] >= 0) {
                const float pil = ldsps[(s.pilot_count + (uint32_t)prank[e]) % 255u] ? 1.0f : -1.0f;
                yph[prank[e]] = atan2f(X[e].y * pil, X[e].x * pil);
            }
        }
        wave_sync_lds();
        float p0 = 0.f, p1 = 0.f, prev = 0.f;
        for (int n = 0; n < c.M_pilot; n++) {
            float v = yph[n];
            if (n > 0) {
                // liquid's `while (d > pi) v -= 2 pi; while (d < -pi) v += 2 pi`, branch free:
                // |v| <= pi and the running reference drifts slowly, so three steps cover it
#pragma unroll
                for (int it = 0; it < 3; it++) v -= ((v - prev) > PI_F) ? 2.0f * PI_F : 0.0f;
#pragma unroll
                for (int it = 0; it < 3; it++) v += ((v - prev) < -PI_F) ? 2.0f * PI_F : 0.0f;
            }
            prev = v;
            p0 += ldspf[n] * v;
            p1 += ldspf[c.M_pilot + n] * v;
        }
        wave_sync_lds();
        s.pilot_count = (s.pilot_count + (uint32_t)c.M_pilot) % 255u;
        p1 = 0.3f * p1 + (1.0f - 0.3f) * s.p1_prime;
        s.p1_prime = p1;
#pragma unroll
        for (int e = 0; e < E; e++) {
            if (k[e] < 0 || sct[e] == 0) { X[e] = make_float2(0.f, 0.f); continue; }
            const float theta = p0 + p1 * fx[e];
            X[e] = mix_down(X[e], rad2u32(theta));
        }
        uint32_t new_dtheta = s.nco_dtheta;
        if (s.num_symbols > 0) {
            float dphi = p0 - s.phi_prime;
#pragma unroll
            for (int it = 0; it < 3; it++) dphi -= (dphi > PI_F) ? 2.0f * PI_F : 0.0f;
#pragma unroll
            for (int it = 0; it < 3; it++) dphi += (dphi < -PI_F) ? 2.0f * PI_F : 0.0f;
            new_dtheta += rad2u32(1e-3f * dphi);
        }
        // the frequency change applies to samples after this event
        s.nco_theta_ref = s.nco_theta_ref + (uint32_t)(t_ev + 1 - s.nco_t_ref) * s.nco_dtheta;
        s.nco_t_ref = t_ev + 1;
        s.nco_dtheta = new_dtheta;
        s.phi_prime = p0;
        s.num_symbols++;
        s.timer = (uint32_t)L;
        return flex_symbol(X, t_ev);
    }

    // payload worker: run the symbols of one handed-off frame to its end
    __device__ __forceinline__ void run_job(uint32_t j)
    {
        const PayloadJob job = a.jobs[j];
        if (!bind_job(j, job)) return;
        init_consts();
        const int64_t woff = (int64_t)(c.cp - c.backoff) - (int64_t)c.L + 1;
        float2 cur[E];
        load_raw_direct(s.cur + (int64_t)s.timer - 1 + woff, cur);
        while (true) {
            const int64_t t_ev = s.cur + (int64_t)s.timer - 1;
            if (t_ev >= a.end) break;               // cannot happen: the scout checked the frame fits
            s.cur = t_ev + 1;
            // the next symbol's window address is known now: fetch it under this symbol's work
            float2 nxt[E];
            const int64_t t_nx = t_ev + (int64_t)c.L;
            if (t_nx < a.end) load_raw_direct(t_nx + woff, nxt);
            else {
#pragma unroll
                for (int e = 0; e < E; e++) nxt[e] = make_float2(0.f, 0.f);
            }
            if (rx_core(t_ev, cur) != 0) break;
#pragma unroll
            for (int e = 0; e < E; e++) cur[e] = nxt[e];
        }
    }

    // ------------------------------------------------------------------------------------
    // Lean payload symbol loop (M = 64 E a power of two, at most 64 pilots): the same arithmetic
    // as rx_core + flex_symbol's payload branch, restructured for VALU issue: wave-uniform state
    // in scalars, window addresses relative to the buffer, twiddles and butterfly signs in
    // registers, DPP / LDS-crossbar lane exchanges, the pilot phase unwrap as a prefix sum of
    // 2 pi jumps (liquid's sequential `while` unwrap only ever adds -rint(d / 2 pi) turns per
    // step), and v_sin / v_cos for the two rotations per sample.
    // (round 5: and 48 subcarriers -- the reference applications' default -- as 3 x 16: lean_prims.hpp, fast_fft48 below)
    __device__ __forceinline__ bool fast_ok() const { return (c.log2M >= 6 && c.M == WV * E && c.M_pilot <= WV) || (E == 1 && c.M == 48 && c.M_pilot <= WV); }
    // lane -> subcarrier of the lean transforms: bit reversal for the power-of-two widths, lean::lane_k<48> (-1: idle lane) for 48
    __device__ __forceinline__ int fast_k(int i) const
    {
        if (c.M == 48) return lean::lane_k<48>(i);
        return (int)(__brev((unsigned)i) >> (32 - c.log2M));
    }

    // 48 = 3 x 16 (lean_prims.hpp): the radix-3 stage's three inputs through the LDS crossbar, then the last four stages of the 64-point
    // transform inside every DPP row.  (The lane constants are formed here, per call: these are the scouts' and the tail kernel's
    // rare paths -- the segment waves and the payload workers of 48-subcarrier symbols hold them in registers.)
    __device__ __forceinline__ void fast_fft48(float2 &x)
    {
        const lean::Radix3 r3 = lean::radix3_consts(l);
        const float xr = x.x, xi = x.y;
        float2 x0, x1, x2;
        x0.x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(r3.a0, __builtin_bit_cast(int, xr)));
        x0.y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(r3.a0, __builtin_bit_cast(int, xi)));
        x1.x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(r3.a1, __builtin_bit_cast(int, xr)));
        x1.y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(r3.a1, __builtin_bit_cast(int, xi)));
        x2.x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(r3.a2, __builtin_bit_cast(int, xr)));
        x2.y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(r3.a2, __builtin_bit_cast(int, xi)));
        const float2 p1 = cmul(x1, make_float2(r3.w1.x, r3.w1.y)), p2 = cmul(x2, make_float2(r3.w2.x, r3.w2.y));
        float2 y = make_float2((x0.x + p1.x) + p2.x, (x0.y + p1.y) + p2.y);
        y = cmul(y, make_float2(r3.t.x, r3.t.y));
        if (l >= 48) y = make_float2(0.f, 0.f);
#define SY_XSTAGE1(ST, H) { const float sx = bfly_leg<H>(y.x, sg[ST]), sy = bfly_leg<H>(y.y, sg[ST]);   \
            if (H == 1) y = make_float2(sx, sy); else y = make_float2(sx * tw[ST].x - sy * tw[ST].y, sx * tw[ST].y + sy * tw[ST].x); }
        SY_XSTAGE1(2, 8) SY_XSTAGE1(3, 4) SY_XSTAGE1(4, 2) SY_XSTAGE1(5, 1)
#undef SY_XSTAGE1
        x = y;
    }
    __device__ __forceinline__ void fast_fft(float2 (&x)[E])
    {
        if constexpr (E == 1) { if (c.M == 48) { fast_fft48(x[0]); return; } }
        // in-lane stages (span 64 J), twiddle W_{128 J}^{i mod 64 J}
#pragma unroll
        for (int J = E / 2; J >= 1; J >>= 1) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                if ((e & J) == 0) {
                    const float2 u = x[e], v = x[e + J];
                    const float rev = (float)((l + WV * e) & (WV * J - 1)) * (0.5f / (float)(WV * J));
                    x[e] = cadd(u, v);
                    x[e + J] = rot_down(csub(u, v), rev);
                }
            }
        }
#define SY_XSTAGE(ST, H)                                                                       \
        _Pragma("unroll") for (int e = 0; e < E; e++) {                                        \
            const float sx = bfly_leg<H>(x[e].x, sg[ST]), sy = bfly_leg<H>(x[e].y, sg[ST]);    \
            if (H == 1) x[e] = make_float2(sx, sy);                                            \
            else x[e] = make_float2(sx * tw[ST].x - sy * tw[ST].y, sx * tw[ST].y + sy * tw[ST].x); \
        }
        SY_XSTAGE(0, 32) SY_XSTAGE(1, 16) SY_XSTAGE(2, 8) SY_XSTAGE(3, 4) SY_XSTAGE(4, 2) SY_XSTAGE(5, 1)
#undef SY_XSTAGE
    }

    // One received symbol up to the pilot fit: NCO on the raw window (phase th_ws at its first
    // sample, step dth), FFT, equaliser, then the pilots -- lane n < M_pilot takes pilot n
    // (subcarrier order): phase, unwrap, linear fit p0 + p1 k, slope smoothing.
    __device__ __forceinline__ void fast_core(float2 (&X)[E], uint32_t th_ws, uint32_t dth, uint32_t &pc, float &p1_prime,
                                              float &p0, float &p1)
    {
        const int Mp = c.M_pilot;
#pragma unroll
        for (int e = 0; e < E; e++) X[e] = rot_down(X[e], u32rev(th_ws + (uint32_t)(l + WV * e) * dth));
        fast_fft(X);
#pragma unroll
        for (int e = 0; e < E; e++) {
            X[e] = cmul(X[e], R[e]);
            if (pr[e] >= 0) ldsc[pr[e]] = X[e];
        }
        wave_sync_lds();
        float2 P = ldsc[l < Mp ? l : 0];
        uint32_t pi_ = pc + (uint32_t)l; pi_ = pi_ >= 255u ? pi_ - 255u : pi_;
        const bool pneg = ldsps[pi_ < 255u ? pi_ : 0u] == 0;
        wave_sync_lds();
        if (pneg) { P.x = -P.x; P.y = -P.y; }
        const float v = atan2_fast(P.y, P.x);
        const float prev = dpp_mov<0x138, false>(v, v);                  // wave_shr:1, lane 0 keeps its own
        const float turns = rintf((v - prev) * 0.15915494309189535f);
        float y;
        if (Mp <= 16) {                     // the usual case (6 pilots at M = 64): everything stays in the first DPP row
            y = fmaf(-TWO_PI_F, row_scan_fast(turns), v);
            p0 = row_total_dpp(pf0 * y);
            p1 = row_total_dpp(pf1 * y);
        } else {
            y = fmaf(-TWO_PI_F, wave_scan_fast(turns), v);
            p0 = wave_total_dpp(pf0 * y);
            p1 = wave_total_dpp(pf1 * y);
        }
        pc += (uint32_t)Mp; pc = pc >= 255u ? pc - 255u : pc;
        p1 = 0.3f * p1 + (1.0f - 0.3f) * p1_prime;
        p1_prime = p1;
    }

    // payload worker, lean symbol loop: the same arithmetic as rx_core + flex_symbol's payload
    // branch with the wave-uniform state in scalars and buffer-relative window addresses
    __device__ __forceinline__ void run_job_fast(uint32_t j)
    {
        const PayloadJob job = a.jobs[j];
        if (!bind_job(j, job)) return;
        init_fast();
#pragma unroll
        for (int e = 0; e < E; e++) R[e] = sct[e] ? bR[k[e]] : make_float2(0.f, 0.f);
        reinterpret_cast<uint32_t *>(ldsps)[l] = reinterpret_cast<const uint32_t *>(c.pilot_seq)[l];     // 256 bytes, one word per lane
        // the soft demodulator walks the symbol's nearest neighbours: their table (64 or 256 bytes) goes to LDS,
        // it sits on every symbol's dependency chain
        if (s.mod_scheme == 27) { if (l < 16) reinterpret_cast<uint32_t *>(ldsqn)[l] = reinterpret_cast<const uint32_t *>(c.cod.qam16_nb)[l]; }
        else if (s.mod_scheme == 29) reinterpret_cast<uint32_t *>(ldsqn)[l] = reinterpret_cast<const uint32_t *>(c.cod.qam64_nb)[l];
        wave_sync_lds();

        // ---- wave-uniform state in scalars
        const int L = c.L, cb = c.cp - c.backoff, Md = c.M_data;
        const uint32_t mod = rfl(s.mod_scheme), bps = rfl(s.bps), mod_len = rfl(s.mod_len), nbits = 8u * rfl(s.enc_len);
        const uint32_t nsym = (mod_len + (uint32_t)Md - 1u) / (uint32_t)Md;
        const int64_t t_ev0 = s.cur + (int64_t)s.timer - 1;                 // event of the first payload symbol
        const int64_t ws0 = t_ev0 - L + 1 + cb;
        uint32_t dth = rfl(s.nco_dtheta);
        uint32_t th_ws = rfl(s.nco_theta_ref + (uint32_t)(ws0 - s.nco_t_ref) * s.nco_dtheta);    // NCO phase at the window start
        uint32_t pc = rfl(s.pilot_count);
        float phi_prime = s.phi_prime, p1_prime = s.p1_prime;
        int32_t r_ws = (int32_t)rfl((uint32_t)(ws0 - a.buf_first));         // window start relative to the buffer
        const float2 *chb = a.chan + ((size_t)a.chan_off + ch) * MCRX_TILE_S;
        const size_t tstride = (size_t)a.chan_stride * MCRX_TILE_S;
        uint8_t *soft = bsoft;
        float2 *syms = bsyms;
        const bool soft_mode = c.payload_soft != 0;

        float2 cur[E], nxt[E];
#pragma unroll
        for (int e = 0; e < E; e++) nxt[e] = make_float2(0.f, 0.f);
#pragma unroll
        for (int e = 0; e < E; e++) { const int r = r_ws + l + WV * e; cur[e] = chb[(size_t)(r >> MCRX_TILE_SH) * tstride + (size_t)(r & (MCRX_TILE_S - 1))]; }
        const bool short_sym = c.M < WV * E;                                // (48 subcarriers: the lanes behind the window hold zeros)
        if (short_sym) { for (int e = 0; e < E; e++) if (l + WV * e >= c.M) cur[e] = make_float2(0.f, 0.f); }
        uint32_t psi = 0;
        for (uint32_t n = 0; n < ((a.no_fast & 4) ? 1u : nsym); n++) {
            if (n + 1 < nsym) {
#pragma unroll
                for (int e = 0; e < E; e++) { const int r = r_ws + L + l + WV * e; nxt[e] = chb[(size_t)(r >> MCRX_TILE_SH) * tstride + (size_t)(r & (MCRX_TILE_S - 1))]; }
                if (short_sym) { for (int e = 0; e < E; e++) if (l + WV * e >= c.M) nxt[e] = make_float2(0.f, 0.f); }
            }
            float p0, p1;
            fast_core(cur, th_ws, dth, pc, p1_prime, p0, p1);
            // de-rotate, soft bits
            const float p0r = p0 * 0.15915494309189535f;
            const bool full = (psi + (uint32_t)Md) * bps <= nbits;           // no tail guard needed
#pragma unroll
            for (int e = 0; e < E; e++) {
                if (dr[e] < 0) continue;
                const uint32_t idx = psi + (uint32_t)dr[e];
                if (idx >= mod_len) continue;
                const float2 Z = rot_down(cur[e], fmaf(p1, fxr[e], p0r));
                syms[idx] = Z;
                uint8_t sb[6];
                const unsigned hs = demod_soft(ldsqn, mod, Z, sb);
                if (!soft_mode) {
#pragma unroll
                    for (int kb = 0; kb < 6; kb++) sb[kb] = (uint8_t)(((hs >> ((bps - 1 - kb) & 7)) & 1) ? 255 : 0);
                }
                uint8_t *dst = soft + (size_t)idx * bps;
                if (full) {
                    if (bps == 2)      *reinterpret_cast<uint16_t *>(dst) = (uint16_t)(sb[0] | (sb[1] << 8));
                    else if (bps == 1) dst[0] = sb[0];
                    else {
                        *reinterpret_cast<uint16_t *>(dst) = (uint16_t)(sb[0] | (sb[1] << 8));
                        *reinterpret_cast<uint16_t *>(dst + 2) = (uint16_t)(sb[2] | (sb[3] << 8));
                        if (bps == 6) *reinterpret_cast<uint16_t *>(dst + 4) = (uint16_t)(sb[4] | (sb[5] << 8));
                    }
                } else {
                    for (unsigned kb = 0; kb < bps; kb++) if (idx * bps + kb < nbits) dst[kb] = sb[kb];
                }
            }
            psi += (uint32_t)Md;
            // NCO trim: phase at the next window start uses the old step up to this event, the new one after
            float dphi = p0 - phi_prime;
            dphi -= TWO_PI_F * rintf(dphi * 0.15915494309189535f);
            phi_prime = p0;
            const uint32_t dnew = dth + rfl((uint32_t)__float2int_rn(dphi * (1e-3f * 683565275.5764316f)));    // |.| < 2^22: exact to the rounding of one product
            th_ws += (uint32_t)(L - cb) * dth + (uint32_t)cb * dnew;
            dth = dnew;
            r_ws += L;
#pragma unroll
            for (int e = 0; e < E; e++) cur[e] = nxt[e];
        }
        // ---- frame complete: decode and emit (same tail as flex_symbol)
        // ---- symbols done: decode_kernel (a workgroup per frame) takes the packet from here
        if (l == 0) a.jobs[j].s.nco_dtheta = dth;
    }

    // scout, lean RXSYMBOLS event: the counterpart of rx_core on the Walker's state
    template <int MODE = SYM_FULL>
    __device__ __forceinline__ int rx_event_fast(int64_t t_ev)
    {
        const int L = c.L, cb = c.cp - c.backoff;
        const int64_t ws = t_ev - L + 1 + cb;
        const bool prof = MODE != SYM_SPEC && SY_PROF(a);
        long long k0 = prof ? (long long)__builtin_readcyclecounter() : 0ll, k1;
#define SY_TICK(i) if (prof) { k1 = (long long)__builtin_readcyclecounter(); ph[i] += k1 - k0; k0 = k1; }
        float2 X[E];
        load_raw(ws, X);
        if (prof) { float z = 0.f; for (int e = 0; e < E; e++) z += X[e].x; if (z == 1.2345e-30f) ph[5]++; }   // wait for the window
        SY_TICK(0)
        const uint32_t dth = s.nco_dtheta;
        const uint32_t th_ws = s.nco_theta_ref + (uint32_t)(ws - s.nco_t_ref) * dth;
        float p0, p1;
        fast_core(X, th_ws, dth, s.pilot_count, s.p1_prime, p0, p1);
        if (prof && p0 == 1.2345e-30f) ph[5]++;
        SY_TICK(1)
        const float p0r = p0 * 0.15915494309189535f;
#pragma unroll
        for (int e = 0; e < E; e++) X[e] = sct[e] ? rot_down(X[e], fmaf(p1, fxr[e], p0r)) : make_float2(0.f, 0.f);
        uint32_t new_dtheta = dth;
        if (s.num_symbols > 0) {
            float dphi = p0 - s.phi_prime;
            dphi -= TWO_PI_F * rintf(dphi * 0.15915494309189535f);
            new_dtheta += rad2u32(1e-3f * dphi);
        }
        s.nco_theta_ref = s.nco_theta_ref + (uint32_t)(t_ev + 1 - s.nco_t_ref) * dth;
        s.nco_t_ref = t_ev + 1;
        s.nco_dtheta = new_dtheta;
        s.phi_prime = p0;
        s.num_symbols++;
        s.timer = (uint32_t)L;
        if (prof && new_dtheta == 0x12345u && X[0].x == 1.2345e-30f) ph[5]++;
        SY_TICK(2)
        int r;
        if (s.fstate == FX_HEADER) r = flex_header_fast<MODE>(X, t_ev);
        else if constexpr (MODE == SYM_SPEC) r = 1;          // a speculative wave never walks a payload itself
        else if constexpr (MODE == SYM_LEAN) r = 3;          // (the lean scout never gets here: payloads belong to the tail kernel)
        else r = flex_symbol(X, t_ev);
        SY_TICK(3)
#undef SY_TICK
        return r;
    }

    // header symbols: hard BPSK bits go straight to their de-interleaved, de-scrambled place in LDS
    // (the header packet is always 36 bytes, so its interleaver is one fixed bit permutation)
    // returns 0: header continues, 1: frame over (invalid header), 2: payload handed off, 3 (lean scout only): valid
    // header, but the payload is not entirely in the buffer (or is oversize): the tail kernel walks it
    template <int MODE = SYM_FULL>
    __device__ __forceinline__ int flex_header_fast(const float2 (&X)[E], int64_t t_ev)
    {
        float ev = 0.f;
#pragma unroll
        for (int e = 0; e < E; e++) if (dr[e] >= 0) {
            const uint32_t idx = s.header_symbol_index + (uint32_t)dr[e];
            if (idx < MCRX_HDR_SYMS) {
                const unsigned sym = X[e].x > 0 ? 0u : 1u;
                const unsigned m = ldshm[idx];
                ldshb[m & 0x1ffu] = (uint8_t)(sym ^ (m >> 15));
                const float xh = sym ? -1.0f : 1.0f;
                const float drr = xh - X[e].x, dii = -X[e].y;
                const float evm = sqrtf(drr * drr + dii * dii);
                ev += evm * evm;
            }
        }
        s.evm_hat += wave_total_dpp(ev);
        s.header_symbol_index += (uint32_t)c.M_data;
        if (s.header_symbol_index >= MCRX_HDR_SYMS) {
            const bool prof = MODE != SYM_SPEC && SY_PROF(a);
            long long k0 = prof ? (long long)__builtin_readcyclecounter() : 0ll;
            decode_header_fast();
            if (prof) { if (s.hw[0] == 0x12345678u && s.hw[1] == 0x9abcdef0u) ph[5]++; const long long k1 = (long long)__builtin_readcyclecounter(); ph[4] += k1 - k0; k0 = k1; }
            s.evm = 10.0f * log10f(s.evm_hat / (float)MCRX_HDR_SYMS);
            if (s.header_valid) {
                s.fstate = FX_PAYLOAD; s.payload_symbol_index = 0;
                if constexpr (MODE == SYM_SPEC) {
                    if (try_handoff_spec(t_ev)) return 2;
                    // the lean scout's own verdict on a frame whose payload runs past the end of the buffer (below): deferred when
                    // the next push still holds its beginning -- parked as such, so that the scout need not walk up to here to find out
                    const int64_t nsym_ = (int64_t)((s.mod_len + (uint32_t)c.M_data - 1) / (uint32_t)c.M_data);
                    const bool oversize_ = s.enc_len > c.max_enc_len || s.mod_len > c.max_syms || s.payload_len > c.max_payload_len;
                    return (!oversize_ && t_ev + nsym_ * (int64_t)c.L >= a.end && a.defer_limit > 0 && a.end - sk_cur <= a.defer_limit) ? 4 : 1;
                }
                const bool ho = try_handoff(t_ev);
                if (prof) ph[5] += (long long)__builtin_readcyclecounter() - k0;
                if (ho) return 2;
                if constexpr (MODE == SYM_LEAN) {
                    // Not handed off.  The lean scout never walks a payload itself:
                    const int64_t nsym = (int64_t)((s.mod_len + (uint32_t)c.M_data - 1) / (uint32_t)c.M_data);
                    const int64_t t_last = t_ev + nsym * (int64_t)c.L;
                    const bool oversize = s.enc_len > c.max_enc_len || s.mod_len > c.max_syms || s.payload_len > c.max_payload_len;
                    if (oversize) {
                        // longer than this handle decodes: reported now (header only, payload_valid = 0, the end time it will
                        // have) and jumped over -- after its last symbol liquid is in the fresh state whatever the symbols were
                        emit(t_last, true, false, true);
                        handoff_last = t_last;
                        return 2;
                    }
                    if (t_last >= a.end)
                        // its payload runs past the end of this buffer: deferred (4: the next push still holds the frame's
                        // beginning and acquires it again, whole) or left to the tail kernels (3)
                        return (a.defer_limit > 0 && a.end - sk_cur <= a.defer_limit) ? 4 : 3;
                    // it fits, but the job list (= the record pool) is full: dropped and counted
                    if (l == 0) atomicAdd(a.nrec + 1, 1u);
                    handoff_last = t_last;
                    return 2;
                }
            }
            else { if constexpr (MODE == SYM_SPEC) return 5; emit(t_ev, false, false); return 1; }       // (5: a segment wave parks the record for the scout to emit)
        }
        return 0;
    }
    // 12 Golay(24,12) words -> 18 bytes, CRC-32 over the first 14: registers and LDS only
    __device__ __forceinline__ void decode_header_fast()
    {
        wave_sync_lds();
        if (l < 12) {
            unsigned r = 0;
#pragma unroll
            for (int q = 0; q < 24; q++) r = (r << 1) | (unsigned)ldshb[24 * l + q];
            ldshd[l] = (uint16_t)golay_dec_sym(r);
        }
        wave_sync_lds();
        unsigned by[18];
#pragma unroll
        for (int g = 0; g < 6; g++) {
            const unsigned s0 = ldshd[2 * g], s1 = ldshd[2 * g + 1];
            by[3 * g] = (s0 >> 4) & 0xffu; by[3 * g + 1] = ((s0 << 4) & 0xf0u) | ((s1 >> 8) & 0x0fu); by[3 * g + 2] = s1 & 0xffu;
        }
        wave_sync_lds();
        uint32_t crc = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < MCRX_HDR_DEC; i++) {
            crc ^= by[i];
#pragma unroll
            for (int b = 0; b < 8; b++) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u)));
        }
        crc = ~crc;
        const uint32_t key = (by[14] << 24) | (by[15] << 16) | (by[16] << 8) | by[17];
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) if (4 * w + b < MCRX_HDR_DEC) v |= by[4 * w + b] << (8 * b);
            s.hw[w] = v;
        }
        header_fields(crc == key);
    }

    // one event of the acquisition states (SEEK, S0a, S0b, S1 incl. the equaliser fit)
    __device__ __forceinline__ void sync_event(int64_t t_ev)
    {
        const int M = c.M, M2 = c.M2, L = c.L;
        if (s.state == SY_SEEK) {
            float pw; float2 sh = s0_metric(t_ev, false, pw);
            const float g = (float)M / pw;
            sh = cscale(sh, g);
            const float tau = atan2f(sh.y, sh.x) * (float)M2 / TWO_PI_F;
            s.g0 = g; s.timer = 0;
            if (sqrtf(sh.x * sh.x + sh.y * sh.y) > c.detect_thresh) {
                const int dt = (int)roundf(tau);
                s.timer = (uint32_t)(M + dt) % (uint32_t)M2 + (uint32_t)M;
                s.state = SY_S0A;
            }
        } else if (s.state == SY_S0A) {
            float pw; float2 sh = s0_metric(t_ev, true, pw);
            s.s_hat_0 = cscale(sh, s.g0);
            s.timer = 0; s.state = SY_S0B;
        } else if (s.state == SY_S0B) {
            float pw; float2 sh = cscale(s0_metric(t_ev, true, pw), s.g0);
            const float2 ssum = cadd(s.s_hat_0, sh);
            const float tau = atan2f(ssum.y, ssum.x) * (float)M2 / TWO_PI_F;
            s.timer = (uint32_t)(M + c.cp - c.backoff) - (uint32_t)(int)roundf(tau);
            // CFO: time-domain ML estimate over the two halves of the oldest M window samples
            float2 acc = make_float2(0.f, 0.f);
            const int64_t w0 = t_ev - L + 1;
            for (int i = l; i < M2; i += WV) {
                const float2 r0 = mixed(w0 + i), r1 = mixed(w0 + i + M2);
                const float2 sa = c.s0t[i], sb = c.s0t[i + M2];
                acc = cadd(acc, cmul(cmulc(sa, r0), cmulc(r1, sb)));
            }
            acc = wave_csum(acc);
            const float nu = atan2f(acc.y, acc.x) / (float)M2;
            s.nco_dtheta = rad2u32(nu); s.nco_theta_ref = 0; s.nco_t_ref = t_ev + 1;
            s.state = SY_S1;
        } else if (s.state == SY_S1) {
            s.num_symbols++;
            float2 x[E];
            load_window(t_ev - M + 1, true, x);
            fft(x);
            const float gain = sqrtf((float)c.M_S1) / (float)M;
            wave_sync_lds();
#pragma unroll
            for (int e = 0; e < E; e++) if (k[e] >= 0) { x[e] = cscale(x[e], S1v[e] * gain); ldsc[k[e]] = x[e]; }
            wave_sync_lds();
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll
            for (int e = 0; e < E; e++) if (k[e] >= 0) {
                int kn = k[e] + 1; if (kn >= M) kn -= M;
                acc = cadd(acc, cmulc(ldsc[kn], x[e]));
            }
            acc = wave_csum(acc);
#if MCRX_S1_METRIC_G0_NORMALISED
            float2 gh = cscale(acc, s.g0 / (float)c.M_S1);
#else
            float2 gh = cscale(acc, 1.0f / (float)c.M_S1);
#endif
            gh = cmul(gh, c.backoff_rot);
            const float mag = sqrtf(gh.x * gh.x + gh.y * gh.y);
            if (mag > c.sync_thresh && fabsf(atan2f(gh.y, gh.x)) < 0.1f * PI_F) {
                s.state = SY_RX; s.timer = (uint32_t)(M + c.cp + c.backoff); s.num_symbols = 0;
                reserve_job();
                // equaliser: order-4 LSQ smoothing of |G| and unwrapped arg G, R = 1/G
                const float g = (float)M / sqrtf((float)(c.M_pilot + c.M_data));
                float *yabs = ldsf, *yarg = ldsf + c.Nen;
                wave_sync_lds();
#pragma unroll
                for (int e = 0; e < E; e++) if (erank[e] >= 0) {
#if MCRX_S1_BACKOFF_CORRECTION
                    float sb_, cb_; sincos_u32((uint32_t)(((uint64_t)(unsigned)k[e] * (unsigned)c.backoff << 32) / (unsigned)c.M), sb_, cb_);
                    const float2 G = cmul(cscale(x[e], g), make_float2(cb_, sb_));
#else
                    const float2 G = cscale(x[e], g);
#endif
                    yabs[erank[e]] = sqrtf(G.x * G.x + G.y * G.y);
                    yarg[erank[e]] = atan2f(G.y, G.x);
                }
                wave_sync_lds();
                // Lane l takes the enabled bins of rank l*E .. l*E+E-1 (fft-shifted order).  Phase unwrap as a
                // prefix sum of whole turns (each step of liquid's loop adds -rint(d / 2 pi)): in the lane, then
                // across lanes; then the fit's coefficients in its orthonormal basis: 2 x 5 wave totals.
                float ca[5], ct[5];
                float smk_e[E][5];          // the smoother's factors are fetched here, once per frame (one round trip),
                {                           // rather than held in 10 E registers for the whole walk
                    float smn_l[E][5];
#pragma unroll
                    for (int e = 0; e < E; e++)
#pragma unroll
                        for (int d = 0; d < 5; d++) {
                            smn_l[e][d] = (l * E + e < c.Nen) ? c.smn[(l * E + e) * 5 + d] : 0.f;
                            smk_e[e][d] = (k[e] >= 0) ? c.smk[k[e] * 5 + d] : 0.f;
                        }
                    float va[E], vy[E], tn[E];
#pragma unroll
                    for (int e = 0; e < E; e++) {
                        const int n = l * E + e;
                        va[e] = n < c.Nen ? yabs[n] : 0.f;
                        vy[e] = yarg[n < c.Nen ? n : c.Nen - 1];
                    }
                    float prev = dpp_mov<0x138, false>(vy[E - 1], vy[0]);     // last of the lane before; lane 0: its own first
                    float run = 0.f;
#pragma unroll
                    for (int e = 0; e < E; e++) { run += rintf((vy[e] - prev) * 0.15915494309189535f); tn[e] = run; prev = vy[e]; }
                    const float before = wave_scan_fast(run) - run;             // whole turns of all lower lanes
#pragma unroll
                    for (int e = 0; e < E; e++) vy[e] = fmaf(-TWO_PI_F, before + tn[e], vy[e]);
#pragma unroll
                    for (int d = 0; d < 5; d++) {
                        float sa = 0.f, st = 0.f;
#pragma unroll
                        for (int e = 0; e < E; e++) { sa = fmaf(smn_l[e][d], va[e], sa); st = fmaf(smn_l[e][d], vy[e], st); }
                        ca[d] = wave_total_dpp(sa); ct[d] = wave_total_dpp(st);
                    }
                }
                wave_sync_lds();
#pragma unroll
                for (int e = 0; e < E; e++) {
                    float2 r = make_float2(0.f, 0.f);
                    if (k[e] >= 0 && sct[e] != 0) {
                        float A = 0.f, th = 0.f;
#pragma unroll
                        for (int d = 0; d < 5; d++) { A = fmaf(smk_e[e][d], ca[d], A); th = fmaf(smk_e[e][d], ct[d], th); }
                        // A e^{j th}: two-constant reduction of th to [-pi, pi], then the transcendental unit
                        const float kk = rintf(th * 0.15915494309189535f);
                        float rr = fmaf(-kk, 6.28125f, th); rr = fmaf(-kk, 1.9353071795864769e-3f, rr);
                        const float rev = rr * 0.15915494309189535f;
                        const float gr = A * __builtin_amdgcn_cosf(rev), gi = A * __builtin_amdgcn_sinf(rev);
                        const float d = gr * gr + gi * gi;
                        r = make_float2(gr / d, -gi / d);
                    }
                    R[e] = r;
                    if (k[e] >= 0) bR[k[e]] = r;
                }
                wave_sync_lds();
            } else {
                if (s.num_symbols == 16) reset_framesync();
                s.timer = (uint32_t)M2;
            }
        }
    }

    // Scout side of the speculation.  The slot headers of the channel (start, status, frame end) are read
    // once into registers, two per lane, so finding the slot that started exactly at s.cur is a compare
    // and a ballot; the adopted slots are only noted (LDS) during the walk, and their parked jobs and
    // equalisers are copied into the job list in one pipelined pass after it.  Nothing on the scout's
    // serial chain waits for memory because of an adoption.
    static constexpr int SPH = MCRX_SPEC_MAX / WV;          // slot headers per lane
    int64_t sp_start[SPH]; int32_t sp_tlast[SPH]; uint32_t sp_aux[SPH]; uint32_t nadopted; uint32_t nwalked = 0;
    int32_t sp_rel[SPH];            // start position relative to the buffer of the slots acquired from a fresh post-frame state (INT32_MIN: any other)
    uint32_t nown = 0;              // job list entries noted in ldsad: this channel's frames of the launch, not placed yet
    int64_t sk_cur = 0; uint32_t sk_timer = 0;      // lean scout: the SEEK state before the last seek event (where a frame is re-acquired from if deferred)
    uint32_t win0 = 0;              // first slot of the window of MCRX_SPEC_MAX headers held in registers
    int64_t seg_base = 0, seg_len_s = 0;            // the launch's segment grid (run_seg's), to tell which wave a position belongs to
    __device__ __forceinline__ void load_spec_headers()
    {
        const SpecSlot *sl = a.spec + (size_t)ch * a.spec_stride;
#pragma unroll
        for (int h = 0; h < SPH; h++) {
            const uint32_t k = win0 + (uint32_t)l + WV * h;
            const bool live = k < a.spec_cap;
            const SpecSlot *q = sl + (live ? k : 0);
            const int32_t stt = q->status;
            sp_start[h] = (live && stt >= 1 && stt <= 3) ? q->start : -1;
            sp_tlast[h] = (int32_t)(q->t_last - a.buf_first);       // (positions inside the buffer: 31 bits are plenty)
            sp_aux[h] = (uint32_t)stt | (q->pad << 8);
            const bool fresh = sp_start[h] >= 0 && (sp_start[h] >> 48) == (spec_key(0, (uint32_t)c.L) >> 48);
            sp_rel[h] = fresh ? (int32_t)((sp_start[h] & 0xFFFFFFFFFFFFll) - a.buf_first) : INT32_MIN;
        }
    }
    // The common hop, tight: from a fresh post-frame state at `pos` to the frame a segment wave handed off from exactly there, to
    // the fresh state behind it, and so on -- 32-bit compares on buffer-relative positions, the hit read with v_readlane, nothing
    // but the position carried from hop to hop (through the general lookup below a hop cost ~1500 cycles: 0.09 ms of an 8-channel
    // push of 100 frames per channel).  Leaves at the first state that is not such a hit; the general lookup takes it from there.
    __device__ __forceinline__ bool hop_fresh(int64_t &pos, int64_t &fresh_prev, int64_t &fresh_last, uint32_t &nfresh, uint32_t &nsame)
    {
        static_assert(SPH == 4, "four rows of slot headers");
        bool any = false;
        int32_t rel = (int32_t)(pos - a.buf_first);
        while (nown < MCRX_SPEC_MAX) {
            const unsigned long long b0 = __ballot(sp_rel[0] == rel), b1 = __ballot(sp_rel[1] == rel), b2 = __ballot(sp_rel[2] == rel), b3 = __ballot(sp_rel[3] == rel);
            if (!(b0 | b1 | b2 | b3)) break;
            // (every row's candidate read with v_readlane, the hit chosen among scalars: choosing the ROW first makes the compiler index
            //  the header arrays dynamically -- a copy in scratch memory and a load from it per hop)
            const int l0 = b0 ? (int)__builtin_ctzll(b0) : 0, l1 = b1 ? (int)__builtin_ctzll(b1) : 0, l2 = b2 ? (int)__builtin_ctzll(b2) : 0, l3 = b3 ? (int)__builtin_ctzll(b3) : 0;
            const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)sp_aux[0], l0), a1 = (uint32_t)__builtin_amdgcn_readlane((int)sp_aux[1], l1),
                           a2 = (uint32_t)__builtin_amdgcn_readlane((int)sp_aux[2], l2), a3 = (uint32_t)__builtin_amdgcn_readlane((int)sp_aux[3], l3);
            const int32_t t0 = __builtin_amdgcn_readlane(sp_tlast[0], l0), t1 = __builtin_amdgcn_readlane(sp_tlast[1], l1),
                          t2 = __builtin_amdgcn_readlane(sp_tlast[2], l2), t3 = __builtin_amdgcn_readlane(sp_tlast[3], l3);
            const uint32_t ax = b0 ? a0 : (b1 ? a1 : (b2 ? a2 : a3));
            if ((ax & 0xffu) != 1u) break;
            const int32_t tl = b0 ? t0 : (b1 ? t1 : (b2 ? t2 : t3));
            if (l == 0) ldsad[nown] = ax >> 8;
            nown++; nadopted++;
            any = true;
            rel = tl + 1;
            const int64_t p = a.buf_first + (int64_t)rel;
            if (fresh_prev >= 0 && p - fresh_last == fresh_last - fresh_prev) nsame++;
            fresh_prev = fresh_last; fresh_last = p; nfresh++;
        }
        if (any) pos = a.buf_first + (int64_t)rel;
        return any;
    }
    // the slot that started from exactly `key`, if there is one: noted, and the sample its frame ended with returned
    // aux: the slot's status (1: frame handed off, t_end = its last symbol's event; 2: deferred, t_end = the SEEK position to go
    // back to, aux >> 8 its timer; 3: a frame whose header failed its check, t_end = the event it was decoded at: the scout emits its record)
    __device__ __forceinline__ bool adopt_match(int64_t key, int64_t &t_end, uint32_t &aux, uint32_t &kslot)
    {
        static_assert(MCRX_SPEC_MAX % WV == 0 && MCRX_SPEC_MAX <= 256, "whole rows of slot headers; slot numbers are noted as bytes");
        int hh = -1, hl = 0; int32_t tl = 0; uint32_t ax = 0;
#pragma unroll
        for (int h = 0; h < SPH; h++) {
            const unsigned long long b = __ballot(sp_start[h] == key);
            if (b && hh < 0) { hh = h; hl = (int)__builtin_ctzll(b); tl = sp_tlast[h]; ax = sp_aux[h]; }
        }
        if (hh < 0) return false;
        t_end = a.buf_first + (int64_t)__shfl((int)tl, hl, WV);
        aux = (uint32_t)__shfl((int)ax, hl, WV);
        kslot = (uint32_t)(hl + WV * hh);
        if ((aux & 0xffu) == 1u) {
            if (l == 0) ldsad[nown] = aux >> 8;
            nown++; nadopted++;
        }
        return true;
    }
    // The same chain by list ranking, for windows that hold many frames (a serial hop is ~95 dependent scalar instructions of one wave
    // alone on its SIMD, 0.3-0.9 us: a push of 400 frames per channel spent two thirds of its acquisition hopping).
    //   next:  a slot's successor is the slot whose key is the state behind its frame -- the next slot of the same wave when that wave
    //          went on from there (chained by construction), else one of the first four slots of the next wave (where two waves link:
    //          its first slot when that wave started on the lattice, its second when it started from a coarse position);
    //   rank:  eight rounds of pointer jumping over the window's 256 slots (byte arrays in the wave's event scratch) give every slot
    //          its distance to the end of its chain; the head is the slot keyed `pos`, a slot's place in the head's chain is the
    //          difference of the two distances;
    //   check: the places are filled by whoever claims them and then verified link by link against `next` -- the verified prefix is
    //          the chain, whatever else the window holds (chains that merge, stale slots);
    //   adopt: the prefix' job entries go to the owner's list in order, the cadence counters are what the serial hop would have counted.
    // Anything it cannot take (no head, a link it does not see, a non-frame slot) is left to hop_fresh / the general lookup.
    __device__ __forceinline__ bool rank_window(int64_t &pos, int64_t &fresh_prev, int64_t &fresh_last, uint32_t &nfresh, uint32_t &nsame)
    {
        static_assert(SPH == 4 && MCRX_SPEC_MAX == 256, "byte indices over four rows of slot headers");
        if (c.M < 64 || a.nseg == 0) return false;
        const int32_t rel0 = (int32_t)(pos - a.buf_first);
        bool val[SPH];
        unsigned long long vb[SPH]; int nvalid = 0;
#pragma unroll
        for (int h = 0; h < SPH; h++) { val[h] = sp_rel[h] != INT32_MIN && (sp_aux[h] & 0xffu) == 1u; vb[h] = __ballot(val[h]); nvalid += __popcll(vb[h]); }
        if (nvalid < 24) return false;                       // (a short chain: the serial hop is cheaper than the ~800 instructions below)
        // head
        int head = -1;
#pragma unroll
        for (int h = SPH - 1; h >= 0; h--) { const unsigned long long b = __ballot(val[h] && sp_rel[h] == rel0); if (b) head = 64 * h + (int)__builtin_ctzll(b); }
        if (head < 0) return false;
        uint8_t *nx = reinterpret_cast<uint8_t *>(sy_w), *rk = nx + 256, *nx0 = nx + 512, *Lp = nx + 768;
        const uint32_t spw = a.spec_cap / a.nseg;
        wave_sync_lds();
        // ---- next
        int nxt[SPH];
#pragma unroll
        for (int h = 0; h < SPH; h++) {
            const int i = l + 64 * h;
            const uint32_t gi = win0 + (uint32_t)i;                                 // slot number in the channel
            // the slot behind me: the lane to my right, or the next row's first lane
            int32_t rr = __shfl_down(sp_rel[h], 1, WV); bool vr = (__shfl_down(val[h] ? 1 : 0, 1, WV)) != 0;
            if (h + 1 < SPH) { const int32_t r0 = __builtin_amdgcn_readlane(sp_rel[h + 1], 0); const bool v0 = (vb[h + 1] & 1ull) != 0; if (l == 63) { rr = r0; vr = v0; } }
            else if (l == 63) vr = false;
            const int32_t want = sp_tlast[h] + 1;
            const bool same_wave = (gi + 1u) % spw != 0u;
            int n = i;                                                              // (nobody: a chain ends here)
            if (val[h] && same_wave && vr && rr == want) n = i + 1;
            nxt[h] = n;
        }
        // links between waves: the first four slots of the next wave, read from their lanes
#pragma unroll
        for (int h = 0; h < SPH; h++) {
            const int i = l + 64 * h;
            const uint32_t gi = win0 + (uint32_t)i;
            const uint32_t c0g = (gi / spw + 1u) * spw;                             // next wave's first slot (channel numbering)
            const int32_t want = sp_tlast[h] + 1;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t cg = c0g + (uint32_t)k;
                const bool inw = cg >= win0 && cg < win0 + 256u && cg < a.spec_cap && (uint32_t)k < spw;
                const int ci = inw ? (int)(cg - win0) : 0;
                int32_t cr = 0; int cv = 0;
#pragma unroll
                for (int r = 0; r < SPH; r++) {                                     // (every lane asks its own candidate: one permute per row of headers)
                    const int32_t t = __shfl(sp_rel[r], ci & 63, WV); const int tv = __shfl(val[r] ? 1 : 0, ci & 63, WV);
                    if ((ci >> 6) == r) { cr = t; cv = tv; }
                }
                if (val[h] && nxt[h] == i && inw && cv && cr == want) nxt[h] = ci;
            }
        }
#pragma unroll
        for (int h = 0; h < SPH; h++) {
            const int i = l + 64 * h;
            nx[i] = (uint8_t)nxt[h]; nx0[i] = (uint8_t)nxt[h]; rk[i] = (uint8_t)(nxt[h] != i ? 1 : 0);
        }
        wave_sync_lds();
        // ---- rank: distance to the end of the chain
#pragma unroll 1
        for (int round = 0; round < 8; round++) {
            int na[SPH], ra[SPH];
#pragma unroll
            for (int h = 0; h < SPH; h++) { const int i = l + 64 * h; const int a1 = nx[i]; ra[h] = (int)rk[i] + (int)rk[a1]; na[h] = nx[a1]; if (a1 == i) ra[h] = rk[i]; }
            wave_sync_lds();
#pragma unroll
            for (int h = 0; h < SPH; h++) { const int i = l + 64 * h; nx[i] = (uint8_t)na[h]; rk[i] = (uint8_t)(ra[h] > 255 ? 255 : ra[h]); }
            wave_sync_lds();
        }
        const int R = rk[head];
        int n = R + 1;
        const int room = MCRX_SPEC_MAX - (int)nown;
        if (n > room) n = room;
        // ---- places, claimed ...
#pragma unroll
        for (int h = 0; h < SPH; h++) { const int i = l + 64 * h; const int rki = rk[i]; if (val[h] && rki <= R) Lp[R - rki] = (uint8_t)i; }
        wave_sync_lds();
        // ... and verified link by link
        int mine[SPH]; int nok = n;
#pragma unroll
        for (int h = 0; h < SPH; h++) {
            const int pp = l + 64 * h;
            mine[h] = Lp[pp];
            const int prev = Lp[pp > 0 ? pp - 1 : 0];
            const bool good = pp == 0 ? mine[h] == head : (int)nx0[prev] == mine[h] && prev != mine[h];
            const unsigned long long bad = __ballot(pp < n && !good);
            if (bad) { const int first = 64 * h + (int)__builtin_ctzll(bad); if (first < nok) nok = first; }
        }
        if (nok <= 0) return false;
        n = nok;
        wave_sync_lds();
        // ---- adopt: job entries in chain order; the frames' ends for the cadence counters (the scratch is free again: positions as words)
        int32_t *fend = reinterpret_cast<int32_t *>(sy_w);
        int32_t myend[SPH];
#pragma unroll
        for (int h = 0; h < SPH; h++) {
            const int pp = l + 64 * h, i = mine[h];
            uint32_t ax = 0; int32_t tl = 0;
#pragma unroll
            for (int r = 0; r < SPH; r++) {
                const uint32_t t = (uint32_t)__shfl((int)sp_aux[r], i & 63, WV); const int32_t u = __shfl(sp_tlast[r], i & 63, WV);
                if ((i >> 6) == r) { ax = t; tl = u; }
            }
            myend[h] = tl + 1;
            if (pp < n) { ldsad[nown + (uint32_t)pp] = ax >> 8; fend[pp] = tl + 1; }
        }
        wave_sync_lds();
        const int32_t relp1 = fresh_last >= 0 ? (int32_t)(fresh_last - a.buf_first) : INT32_MIN, relp2 = fresh_prev >= 0 ? (int32_t)(fresh_prev - a.buf_first) : INT32_MIN;
        uint32_t same = 0;
#pragma unroll
        for (int h = 0; h < SPH; h++) {
            const int pp = l + 64 * h;
            const int32_t f1 = pp >= 1 ? fend[pp - 1] : relp1, f2 = pp >= 2 ? fend[pp - 2] : (pp == 1 ? relp1 : relp2);
            const bool have = pp >= 2 || (pp == 1 ? relp1 != INT32_MIN : (relp1 != INT32_MIN && relp2 != INT32_MIN));
            same += (uint32_t)__popcll(__ballot(pp < n && have && myend[h] - f1 == f1 - f2));
        }
        const int32_t e1 = fend[n - 1], e2 = n >= 2 ? fend[n - 2] : relp1;
        wave_sync_lds();
        nsame += same; nfresh += (uint32_t)n; nown += (uint32_t)n; nadopted += (uint32_t)n;
        fresh_prev = (n >= 2 || relp1 != INT32_MIN) ? a.buf_first + (int64_t)e2 : -1; fresh_last = a.buf_first + (int64_t)e1;
        pos = fresh_last;
        return true;
    }
    // ... looked up where a push holds more slots per channel than the window: the slots a frame acquired from `pos` can sit in are
    // the ones of the wave whose segment holds `pos` and of the wave before it (its last frame, the one that links the two, is
    // acquired from a state behind the segment boundary when the frame before it straddles it); if the window does not hold
    // both, it moves there and the lookup is repeated.  A miss only ever costs speed: the scout then walks that frame itself.
    __device__ __forceinline__ bool adopt_lookup(int64_t key, int64_t pos, int64_t &t_end, uint32_t &aux, uint32_t &kslot)
    {
        if (adopt_match(key, t_end, aux, kslot)) return true;
        if (a.spec_cap <= (uint32_t)MCRX_SPEC_MAX || seg_len_s <= 0) return false;
        const uint32_t spw = a.spec_cap / a.nseg;
        int64_t gq = (pos - seg_base) / seg_len_s;
        if (gq < 0) gq = 0;
        const uint32_t g = gq >= (int64_t)a.nseg ? a.nseg - 1u : (uint32_t)gq;
        const uint32_t want0 = (g ? g - 1u : 0u) * spw;
        if (want0 >= win0 && (g + 1u) * spw <= win0 + (uint32_t)MCRX_SPEC_MAX) return false;        // both waves' slots were in the window
        win0 = want0;
        load_spec_headers();
        return adopt_match(key, t_end, aux, kslot);
    }
    // The channel's frames of this launch become real: their job list entries (written by segment waves, or by this scout's own
    // hand-offs) get their owner, and record space -- payload bytes, equalised symbols, a record slot -- out of ONE reservation per
    // channel and launch: a wave-level prefix sum of the frames' sizes, then one atomic per arena.  (Rounds 1-3 placed all frames of a
    // launch in one workgroup between the scouts and the workers: 45 us on every push's critical chain.)  Frames of a block that
    // runs past the end of a pool are counted as dropped.
    __device__ __forceinline__ void place_owned()
    {
        if (!nown) return;
        wave_sync_lds();
        constexpr int PO = MCRX_SPEC_MAX / WV;
        uint32_t jj[PO], p16[PO]; unsigned long long sb[PO];
        unsigned long long mine_s = 0; uint32_t mine_p = 0, mine_n = 0, menc = 0;
#pragma unroll
        for (int u = 0; u < PO; u++) {                          // lane l takes entries l*PO .. l*PO+PO-1: offsets grow with the list
            const uint32_t i = (uint32_t)l * PO + u;
            const bool v = i < nown;
            jj[u] = v ? ldsad[i] : 0xFFFFFFFFu;
            const PayloadJob *q = a.jobs + (v ? jj[u] : 0u);
            const uint32_t ml = q->s.mod_len, pl = q->s.payload_len, el = q->s.enc_len, md = q->s.mod_scheme;
            sb[u] = v ? 8ull * ml : 0ull; p16[u] = v ? (pl + 15u) >> 4 : 0u;
            mine_s += sb[u]; mine_p += p16[u]; mine_n += v ? 1u : 0u;
            if (v && el > menc) menc = el;
            if (v && a.qam_list && md != 39u && md != 40u) { const uint32_t at = atomicAdd(a.qam_list, 1u); if (at < a.max_jobs) a.qam_list[1u + at] = jj[u]; }
        }
        // exclusive prefix over the lanes
        unsigned long long ex_s = mine_s; uint32_t ex_p = mine_p, ex_n = mine_n;
#pragma unroll
        for (int d = 1; d < WV; d <<= 1) {
            const unsigned long long ts = (unsigned long long)__shfl_up((long long)ex_s, d, WV); const uint32_t tp = (uint32_t)__shfl_up((int)ex_p, d, WV), tn = (uint32_t)__shfl_up((int)ex_n, d, WV);
            if (l >= d) { ex_s += ts; ex_p += tp; ex_n += tn; }
        }
        const unsigned long long tot_s = (unsigned long long)__shfl((long long)ex_s, WV - 1, WV); const uint32_t tot_p = (uint32_t)__shfl((int)ex_p, WV - 1, WV), tot_n = (uint32_t)__shfl((int)ex_n, WV - 1, WV);
        ex_s -= mine_s; ex_p -= mine_p; ex_n -= mine_n;
#pragma unroll
        for (int h = 32; h >= 1; h >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)menc, h, WV); menc = o > menc ? o : menc; }
        // (nothing is ever given back: a counter that moves both ways under concurrent reservations hands the same bytes out twice.
        //  A block that runs past the end of a pool is used as far as it fits; the counters may end up beyond the capacities, which
        //  the harvest clamps, and the pool stays exhausted until then)
        unsigned long long b_s = 0, b_p = 0; uint32_t b_r = 0, b_l = 0;
        if (l == 0) {
            b_p = atomicAdd(a.arena_used, 16ull * tot_p); b_s = atomicAdd(a.arena_used + 1, tot_s); b_r = atomicAdd(a.nrec, tot_n);
            b_l = atomicAdd(a.live, tot_n);
            if (menc && a.stats) atomicMax(a.stats + 7, menc);
        }
        b_p = (unsigned long long)__shfl((long long)b_p, 0, WV); b_s = (unsigned long long)__shfl((long long)b_s, 0, WV); b_r = (uint32_t)__shfl((int)b_r, 0, WV); b_l = (uint32_t)__shfl((int)b_l, 0, WV);
        unsigned long long os = b_s + ex_s, op = b_p + 16ull * ex_p; uint32_t r = b_r + ex_n, lv = b_l + ex_n, lost = 0;
#pragma unroll
        for (int u = 0; u < PO; u++) if (jj[u] != 0xFFFFFFFFu) {
            PayloadJob *q = a.jobs + jj[u];
            if (op + 16ull * p16[u] <= a.arena_cap && os + sb[u] <= a.sarena_cap && r < a.max_rec) { q->arena_off = op; q->syms_off = os; q->pad = r; }
            else { q->arena_off = ~0ull; q->syms_off = ~0ull; lost++; if (r < a.max_rec) a.rec[r].channel = 0xFFFFFFFFu; }     // (a record slot nobody fills: the harvest skips it)
            q->ch = ch;
            if (lv < a.max_jobs) a.live[1u + lv] = jj[u];
            op += 16ull * p16[u]; os += sb[u]; r++; lv++;
        }
        if (lost) atomicAdd(a.nrec + 1, lost);
        wave_sync_lds();
        nown = 0;
    }

    // Coarse preamble finder for the segment waves (never part of the synchronizer's decisions: it only chooses where a wave
    // starts).  S0 carries even subcarriers only, so the two S0 symbols are M/2-periodic in time: lane l takes the M samples
    // at from + l M/2 and compares their lag-M/2 autocorrelation with their energy (1 inside the preamble, ~ sqrt(2/M) on data
    // symbols or noise; exact zeros give NaN = no hit).  Returns the first grid position above 0.6, or -1 before `to`.
    __device__ __forceinline__ int64_t coarse_scan(int64_t from, int64_t to)
    {
        const int M2 = c.M2;
        if (to > a.end - c.M) to = a.end - c.M;                 // windows stay inside the buffer
        const float2 *chb = a.chan + ((size_t)a.chan_off + ch) * MCRX_TILE_S;
        const size_t tstride = (size_t)a.chan_stride * MCRX_TILE_S;
        if (from < a.buf_first) from = a.buf_first;             // (samples in front of the buffer are not there to be looked at)
        if (from < 0) from = 0;
        for (int64_t p0 = from; p0 < to; p0 += (int64_t)WV * M2) {
            const int64_t d = p0 + (int64_t)l * M2;
            const bool in = d < to;
            const uint32_t r0 = (uint32_t)((in ? d : p0) - a.buf_first);       // window [r0, r0 + M) lies inside the buffer: p0 < to <= end - M
            float2 acc = make_float2(0.f, 0.f); float en = 0.f;
            // eight products at a time, their sixteen loads in flight together (branch-free addresses: a chain of 2 M2 round trips
            // per pass was a frame's worth of latency for the two validation scans every segment wave makes)
            for (int n0 = 0; n0 < M2; n0 += 8) {
                float2 u[8], v[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const uint32_t ru = r0 + (uint32_t)(n0 + q), rv = ru + (uint32_t)M2;
                    u[q] = chb[(size_t)(ru >> MCRX_TILE_SH) * tstride + (ru & (uint32_t)(MCRX_TILE_S - 1))];
                    v[q] = chb[(size_t)(rv >> MCRX_TILE_SH) * tstride + (rv & (uint32_t)(MCRX_TILE_S - 1))];
                }
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    acc = cadd(acc, cmulc(u[q], v[q]));
                    en += u[q].x * u[q].x + u[q].y * u[q].y + v[q].x * v[q].x + v[q].y * v[q].y;
                }
            }
            const bool hit = in && (acc.x * acc.x + acc.y * acc.y) > 0.09f * en * en;        // |P| / (E / 2) > 0.6
            const unsigned long long b = __ballot(hit);
            if (b) return p0 + (int64_t)__builtin_ctzll(b) * M2;
        }
        return -1;
    }

    // Segment wave (kernels.h, SpecSlot): the synchronizer's own events -- sync_event, the header symbols, the header decode --
    // from a start state through this wave's segment, frame after frame; every hand-off parked in the next slot of the wave's
    // range under the key of the state the frame was acquired from.  No side effects outside the wave's slots (and anchor[ch]).
    __device__ __forceinline__ void run_seg(uint32_t g)
    {
        const uint32_t spw = a.spec_cap / a.nseg;                // slots of this wave
        SpecSlot *sl0 = a.spec + (size_t)ch * a.spec_stride + (size_t)g * spw;
        bR = a.spec_R + ((size_t)ch * MCRX_SEG_MAX + (g < MCRX_SEG_MAX ? g : 0u)) * c.M;       // (the S1 fit's memory copy: unused, the hand-off stores R from registers)
        const int M = c.M, M2 = c.M2, L = c.L;
        const int phase = a.seg_phase;
        s = a.st[ch];
        // (segment 0 may start in the middle of an acquisition the previous push began: only then does the detection lie in another push)
        if (a.seekst && g == 0 && s.state != SY_SEEK) { sk_cur = a.seekst[2 * (size_t)ch]; sk_timer = (uint32_t)a.seekst[2 * (size_t)ch + 1]; }
        const bool mid_payload = s.state == SY_RX && s.fstate == FX_PAYLOAD;      // (a payload that runs through this whole push: the tail kernel's)
        const int64_t base = s.cur, span = a.end - base;
        const int64_t seg_len = span > 0 ? (span + (int64_t)a.nseg - 1) / (int64_t)a.nseg : 0;
        const int64_t seg_start = base + (int64_t)g * seg_len;
        const bool last = g + 1 == a.nseg;
        const int64_t seg_end = last ? INT64_MAX : seg_start + seg_len;          // a frame detected at or behind it is the next wave's first: my last
        // idle for a whole segment behind mine: give up (the scout walks idle stretches).  Phase 1 is the one serial step every other
        // wave of the launch waits for: a channel whose push begins in silence has no anchor (everybody takes the coarse start)
        const int64_t seek_limit = phase == 1 ? base + 8 * (int64_t)M : (last ? INT64_MAX : seg_end + seg_len);
        int64_t key = -1;
        bool go = !mid_payload && seg_len > 0 && seg_start < a.end;
        uint32_t j = 0, jmax = spw;
        // cadence: the lattice anchor + n P of fresh post-frame states the frames of a periodic stream are acquired from
        // No launch in front of the segment waves where the lattice can be carried over from the previous push: phase 3 -- from the last
        // fresh state behind a frame that the channel's scout saw (ChanState::last_fresh): a continuous stream; phase 4 -- from where that
        // state stood RELATIVE TO THE PUSH'S BEGINNING (the scout leaves the offset in anchor[ch]): pushes that are bursts of their own, a
        // capture window per burst, a replayed slab with a gap at its end.  A wrong guess costs the two acquisitions per segment again;
        // the host tells by the frames the waves hand off for nothing (stats[2] against stats[1]) and moves on to the next one.
        const int64_t P = (int64_t)s.period_hint;
        const int64_t A = (phase == 2 && a.anchor) ? a.anchor[ch]
                        : (phase == 3 && P > 0 && s.last_fresh > 0) ? s.last_fresh
                        : (phase == 4 && P > 0 && a.anchor && a.anchor[ch] >= 0) ? base + a.anchor[ch] : -1;
        auto lattice = [&](int64_t from) -> int64_t {           // its first point at or behind `from` (phase 2: strictly behind the anchor)
            if (A < 0 || P <= 0) return -1;
            const int64_t d = from - A, pt = A + (d >= 0 ? (d + P - 1) / P : -((-d) / P)) * P;
            return (phase == 2 && pt <= A) ? A + P : pt;
        };
        auto preamble_behind = [&](int64_t p) -> bool {        // a frame does begin within a few symbols of p
            if (p < 0 || p + 6 * (int64_t)L >= a.end) return false;
            return coarse_scan(p, p + 4 * (int64_t)L + M2) >= 0;
        };
        int64_t p_next = -1;                                    // the next segment's start on the lattice, if that wave takes it
        if (go) {
            if (g == 0 && phase == 2 && A >= 0) {
                // phase 2: on from where phase 1's frame ended (slot 0 is that frame's)
                j = 1;
                reset_framesync(); s.timer = (uint32_t)L; s.cur = A; key = spec_key(A, (uint32_t)L); init_consts();
            } else if (g == 0) {
                // the channel's real state, whatever it is (normally the fresh state behind the last frame of the previous push)
                key = spec_key(s.cur, s.timer, s.state);
                { float2 *mine = bR; bR = a.R + (size_t)ch * c.M; init_consts(); bR = mine; }      // (R from the channel's equaliser: an acquisition in progress has it)
                if (fastp && s.state == SY_RX && s.fstate == FX_HEADER && s.header_symbol_index > 0) {
                    for (int i = l; i < MCRX_HDR_SYMS; i += WV) ldshb[i] = bhbits[i];
                    wave_sync_lds();
                }
                if (phase == 1) jmax = 1;
            } else {
                reset_framesync(); s.timer = 0; s.cur = seg_start;
                init_consts();
                int64_t p_me = lattice(seg_start);
                if (p_me >= 0 && !last && p_me >= seg_end) p_me = -1;              // the lattice skips my segment (that point is a later wave's): coarse start
                if (preamble_behind(p_me)) { s.timer = (uint32_t)L; s.cur = p_me; key = spec_key(p_me, (uint32_t)L); }
                else {
                    const int64_t hit = coarse_scan(seg_start, seek_limit < a.end ? seek_limit : a.end);
                    if (hit < 0) go = false;
                    else { const int64_t p = hit - M; s.cur = p > seg_start ? p : seg_start; }
                }
            }
            if (go && !last && phase != 1) {                    // (the next wave's own rule, on its own segment)
                int64_t pn = lattice(seg_end);
                if (pn >= 0 && g + 2 != a.nseg && pn >= seg_end + seg_len) pn = -1;
                if (preamble_behind(pn)) p_next = pn;
            }
        }
        if (go) reserve_block();
        while (go && j < jmax) {
            slot = sl0 + j;
            int verdict = 0;
            while (true) {
                if (s.state == SY_SEEK) {
                    sk_cur = s.cur; sk_timer = s.timer;
                    if (s.cur >= seek_limit) break;
                    if (SY_SEG_BURST && a.seek_burst && s.timer == 0 && s.cur >= a.buf_first && s.cur + (int64_t)SEEK_B * M <= a.end) {
                        seek_burst();                           // idle stretch: SEEK_B events per HBM round trip
                        if (s.state != SY_SEEK) { sk_cur = s.cur - M; sk_timer = 0; }       // detected in the burst: the SEEK state that event was taken from
                        continue;
                    }
                }
                int64_t t_ev;
                if (s.state == SY_SEEK)       t_ev = s.cur + ((s.timer + 1 >= (uint32_t)M) ? 0 : (int64_t)(M - 1 - (int)s.timer));
                else if (s.state == SY_S0A || s.state == SY_S0B)
                                              t_ev = s.cur + ((s.timer + 1 >= (uint32_t)M2) ? 0 : (int64_t)(M2 - 1 - (int)s.timer));
                else                          t_ev = s.cur + (int64_t)s.timer - 1;
                if (t_ev >= a.end) break;
                s.cur = t_ev + 1;
                if (s.state != SY_RX) { sync_event(t_ev); continue; }
                const int fr = rx_event_fast<SYM_SPEC>(t_ev);
                if (fr == 0) continue;
                verdict = fr;
                break;
            }
            if (verdict == 2) {                                 // handed off: the job sits in the slot (try_handoff_spec)
                if (l == 0) { slot->start = key; slot->t_last = handoff_last; slot->status = 1; slot->pad = handoff_job; }
                j++;
                if (phase == 1) { if (l == 0) { a.anchor[ch] = handoff_last + 1; if (a.stats) atomicAdd(a.stats + 2, 1u); } void_block(); return; }
                if (handoff_last + 1 == p_next) break;          // exactly the state the next segment's wave started from: linked
                // ... else the first frame detected at or behind the place the next wave started looking from -- its lattice point if
                // it took one, else the segment boundary -- is that wave's first frame: acquired from my (real) state it links us
                if (sk_cur >= (p_next >= 0 ? p_next : seg_end)) break;
                reset_framesync(); s.timer = (uint32_t)L; s.cur = handoff_last + 1;      // where liquid stands after the frame's last symbol
                key = spec_key(s.cur, s.timer);
                continue;
            }
            if (verdict == 5 && phase != 1) {                   // header decoded, check failed: the record is the scout's to write; liquid resets and seeks on
                const uint32_t jp = park_state();
                if (jp == 0xFFFFFFFFu) break;
                if (l == 0) { slot->start = key; slot->t_last = s.cur - 1; slot->status = 3; slot->pad = jp; }
                j++;
                reset_framesync(); s.timer = (uint32_t)L;
                key = spec_key(s.cur, s.timer);
                continue;
            }
            if (verdict == 4 && phase != 1) {                   // runs past the end of the buffer, the next push re-acquires it whole
                if (l == 0) { slot->start = key; slot->t_last = sk_cur; slot->status = 2; slot->pad = sk_timer; }
                j++;
            }
            break;                                              // anything else (invalid header, end of the buffer, idle) is the scout's
        }
        if ((a.debug & 64) && l == 0 && ch == 0) printf("[segw] g %u phase %d: A %lld P %lld seg [%lld, %lld) p_next %lld frames %u last state %d cur %lld\n", g, phase, (long long)A, (long long)P, (long long)seg_start, (long long)(last ? -1 : seg_end), (long long)p_next, j, s.state, (long long)s.cur);
        if (l == 0 && a.stats && j > ((g == 0 && phase == 2 && A >= 0) ? 1u : 0u)) atomicAdd(a.stats + 2, j - ((g == 0 && phase == 2 && A >= 0) ? 1u : 0u));      // (slots this wave filled: the host's waste count)
        void_block();
        if (phase == 1) {                                       // no hand-off: no anchor; segment 0 starts over from the entry state in phase 2
            if (l == 0) { a.anchor[ch] = -1; sl0[0].start = -1; sl0[0].status = 0; }
            return;
        }
        for (uint32_t q = j + (uint32_t)l; q < spw; q += WV) { sl0[q].start = -1; sl0[q].status = 0; }
    }

    // Lean configurations' tail kernel: a channel with a payload in progress (a frame that straddles two pushes and
    // could not be deferred, is oversize, or found the job list full) has its symbols walked here, to the frame's end
    // or the end of the buffer.  Nothing else -- no acquisition, no speculation -- so that the kernel stays small
    // enough to be scheduled beside the payload workers of the previous push.
    __device__ __forceinline__ void run_tail()
    {
        s = a.st[ch];
        if (!(s.state == SY_RX && s.fstate == FX_PAYLOAD)) return;
        init_consts();
        const int L = c.L;
        while (true) {
            const int64_t t_ev = s.cur + (int64_t)s.timer - 1;
            if (t_ev >= a.end) { s.timer -= (uint32_t)(a.end - s.cur); s.cur = a.end; break; }
            s.cur = t_ev + 1;
            const int fr = rx_event_fast<SYM_FULL>(t_ev);
            if (fr == 1) { reset_framesync(); s.timer = (uint32_t)L; break; }      // frame over: the lean scouts go on from here
        }
        if (l == 0) a.st[ch] = s;
    }

    // MODE SYM_FULL: the whole state machine (general configurations; with a.tail_only the tail kernel of the lean
    // configurations: only channels with a payload in progress, and only to that frame's end).
    // MODE SYM_LEAN: the lean scout -- acquisition, header, hand-off; a payload in progress is not its business.
    template <int MODE = SYM_FULL, bool BURST = true>
    __device__ __forceinline__ void run()
    {
        s = a.st[ch];
        if (a.seekst && s.state != SY_SEEK) { sk_cur = a.seekst[2 * (size_t)ch]; sk_timer = (uint32_t)a.seekst[2 * (size_t)ch + 1]; }
        const bool mid_payload = s.state == SY_RX && s.fstate == FX_PAYLOAD;
        if constexpr (MODE == SYM_LEAN) { if (mid_payload) return; }
        else if (a.tail_only && !mid_payload) return;
        init_consts();
        if (fastp && s.state == SY_RX && s.fstate == FX_HEADER && s.header_symbol_index > 0) {
            for (int i = l; i < MCRX_HDR_SYMS; i += WV) ldshb[i] = bhbits[i];
            wave_sync_lds();
        }
        const int M = c.M, M2 = c.M2, L = c.L;
        long long prof_cyc[5] = {0, 0, 0, 0, 0}; int prof_n[5] = {0, 0, 0, 0, 0};
        bool entry = true;
        int64_t fresh_prev = -1, fresh_last = -1;       // the last two fresh post-frame states (behind handed-off frames): the cadence the next push's segment waves try
        uint32_t nfresh = 0, nsame = 0;                 // hand-offs seen, and how many of them followed their predecessor at the distance of the pair before
        nadopted = 0;
        if (a.spec_cap) {
            win0 = 0; seg_base = s.cur;
            const int64_t span = a.end - seg_base;
            seg_len_s = (span > 0 && a.nseg) ? (span + (int64_t)a.nseg - 1) / (int64_t)a.nseg : 0;
            load_spec_headers();
        }
        while (true) {
            if (nown >= MCRX_SPEC_MAX - 1) place_owned();       // (a channel with hundreds of frames in one push)
            if (a.spec_cap && (entry || (s.state == SY_SEEK && s.timer == (uint32_t)L))) {
                // a state a segment wave may have started a frame from -- the one this launch starts in (whatever it is: segment 0
                // cloned it) and the fresh one after every frame: the frame is taken from that wave if it started from exactly this
                // state.  A channel with a hundred frames in the buffer runs key -> slot -> frame end -> next key a hundred times in
                // a row: chased here on the position alone, the synchronizer state written once behind the last adoption (every
                // adoption leaves it in the same fresh post-frame state, only the position differs).
                int64_t key = spec_key(s.cur, s.timer, s.state);
                int64_t pos = s.cur, t_end = 0; uint32_t aux = 0, kslot = 0; bool any = false, deferred = false;
                if (s.state == SY_SEEK && s.timer == (uint32_t)L) {
                    bool got = SY_RANK && rank_window(pos, fresh_prev, fresh_last, nfresh, nsame);
                    got = hop_fresh(pos, fresh_prev, fresh_last, nfresh, nsame) || got;
                    if (got) { any = true; key = spec_key(pos, (uint32_t)L); }
                }
                while (nown < MCRX_SPEC_MAX && adopt_lookup(key, pos, t_end, aux, kslot)) {
                    if ((aux & 0xffu) == 2u) { deferred = true; break; }
                    if ((aux & 0xffu) == 3u) {
                        // a frame whose header did not pass its check (noise, a neighbour's leakage): the record the synchronizer
                        // reports for it is written here, from the state the segment wave parked
                        const ChanState keep = s;
                        s = a.jobs[aux >> 8].s;
                        emit(t_end, false, false);
                        s = keep; nwalked++;
                    }
                    any = true; pos = t_end + 1;
                    if ((aux & 0xffu) == 1u) {
                        if (fresh_prev >= 0 && pos - fresh_last == fresh_last - fresh_prev) nsame++;
                        fresh_prev = fresh_last; fresh_last = pos; nfresh++;
                    }
                    if (SY_RANK) rank_window(pos, fresh_prev, fresh_last, nfresh, nsame);
                    hop_fresh(pos, fresh_prev, fresh_last, nfresh, nsame);         // (on from there in the tight loop)
                    key = spec_key(pos, (uint32_t)L);
                }
                if (deferred) {
                    // the frame acquired from here runs past the end of this buffer and the next push re-acquires it whole: back to
                    // the SEEK state it was detected from (what this scout does itself at `fr == 4` below)
                    reset_framesync(); s.cur = t_end; s.timer = aux >> 8;
                    break;
                }
                if (any) { reset_framesync(); s.timer = (uint32_t)L; s.cur = pos; }
                // (the owner's list is full: place what it holds and look again from this state -- falling through would take the next
                //  event here, and a scout that has left the fresh state walks the whole frame: 61 k cycles once per 255 frames)
                if (any && nown >= MCRX_SPEC_MAX - 1) { entry = false; continue; }
            }
            entry = false;
            if (MODE == SYM_LEAN && s.cur >= a.end) break;      // (a frame jumped over may end beyond this buffer)
            if (s.state == SY_SEEK) { sk_cur = s.cur; sk_timer = s.timer; }
            if (BURST && a.seek_burst && s.state == SY_SEEK && s.timer == 0 && s.cur >= a.buf_first && s.cur + (int64_t)SEEK_B * M <= a.end) {
                seek_burst();                               // idle stretch: SEEK_B events per HBM round trip
                if (s.state == SY_SEEK) continue;
                sk_cur = s.cur - M; sk_timer = 0;           // detected in the burst: the SEEK state that event was taken from
                continue;
            }
            // sample index of the next state-machine event
            int64_t t_ev;
            if (s.state == SY_SEEK)       t_ev = s.cur + ((s.timer + 1 >= (uint32_t)M) ? 0 : (int64_t)(M - 1 - (int)s.timer));
            else if (s.state == SY_S0A || s.state == SY_S0B)
                                          t_ev = s.cur + ((s.timer + 1 >= (uint32_t)M2) ? 0 : (int64_t)(M2 - 1 - (int)s.timer));
            else                          t_ev = s.cur + (int64_t)s.timer - 1;
            if (t_ev >= a.end) {
                // consume what is left of the buffer; keep counting the timer
                const uint32_t adv = (uint32_t)(a.end - s.cur);
                if (s.state == SY_S1 || s.state == SY_RX) s.timer -= adv; else s.timer += adv;
                s.cur = a.end;
                break;
            }
            s.cur = t_ev + 1;
            const int st_in = s.state; const long long tk0 = SY_PROF(a) ? (long long)__builtin_readcyclecounter() : 0ll;
            if ((a.debug & 1) && l == 0 && ch == 0) printf("[sync] ch0 t=%lld state=%d fstate=%d timer=%u hsi=%u psi=%u\n", (long long)t_ev, s.state, s.fstate, s.timer, s.header_symbol_index, s.payload_symbol_index);

            if (s.state != SY_RX) sync_event(t_ev);
            else {    // SY_RX
                int fr;
                if constexpr (MODE == SYM_LEAN) fr = rx_event_fast<SYM_LEAN>(t_ev);
                else fr = fastp ? rx_event_fast<SYM_FULL>(t_ev) : rx_event(t_ev);
                if (fr == 3) break;                 // lean scout: valid header, payload for the tail kernel (state saved below)
                if (fr == 4) {                      // ... or deferred: back to the state the frame was detected from; the next push re-acquires it
                    void_reservation(); reset_framesync(); s.cur = sk_cur; s.timer = sk_timer;
                    break;
                }
                if (fr == 1) {
                    void_reservation(); reset_framesync(); s.timer = (uint32_t)L; nwalked++;
                    if (MODE == SYM_FULL && a.tail_only == 1) break;    // tail kernel before the rounds: the frame in progress is done,
                                                                        // the lean scouts go on from here (after the rounds it walks on to the end)
                }
                else if (fr == 2) {
                    nwalked++;
                    // payload handed to a worker: jump over it; liquid leaves the synchronizer in
                    // SEEK with timer = M+cp after the frame's last symbol
                    reset_framesync(); s.timer = (uint32_t)L; s.cur = handoff_last + 1;
                    if (fresh_prev >= 0 && s.cur - fresh_last == fresh_last - fresh_prev) nsame++;
                    fresh_prev = fresh_last; fresh_last = s.cur; nfresh++;
                }
            }
            if (SY_PROF(a)) { prof_cyc[st_in] += (long long)__builtin_readcyclecounter() - tk0; prof_n[st_in]++; }
        }
        if (SY_PROF(a) && l == 0 && ch == 0)
            printf("[prof] ch0 cycles/events  seek %lld/%d  s0a %lld/%d  s0b %lld/%d  s1 %lld/%d  rx %lld/%d\n",
                   prof_cyc[0], prof_n[0], prof_cyc[1], prof_n[1], prof_cyc[2], prof_n[2], prof_cyc[3], prof_n[3], prof_cyc[4], prof_n[4]);
        if (SY_PROF(a) && l == 0 && ch == 0)
            printf("[prof] ch0 rx phases: load %lld  core %lld  derot+nco %lld  flex %lld (header decode %lld, hand-off %lld)\n", ph[0], ph[1], ph[2], ph[3], ph[4], ph[5]);
        void_reservation();
        place_owned();
        if (a.stats && l == 0) {
            if (nwalked) atomicAdd(a.stats, nwalked);
            if (nadopted) atomicAdd(a.stats + 1, nadopted);
            if (nsame) atomicAdd(a.stats + 4, nsame);           // (what a cadence would have predicted: the host runs the anchor phase only while it pays)
            if (nfresh > 2) atomicAdd(a.stats + 5, nfresh - 2);
        }
        if (fresh_last >= 0) {
            if (fresh_prev >= 0 && fresh_last - fresh_prev < (int64_t)0x7fffffff) s.period_hint = (uint32_t)(fresh_last - fresh_prev);
            s.last_fresh = fresh_last;
        }
        if (l == 0 && a.anchor && a.spec_cap) a.anchor[ch] = fresh_last >= 0 ? fresh_last - seg_base : -1;       // (where the lattice stood in THIS push, from its beginning: run_seg, phase 3)
        if ((a.debug & 4) && l == 0) printf("[seg] ch %u adopted %u walked %u (slots %u in %u segments)\n", ch, nadopted, nwalked, a.spec_cap, a.nseg);
        // a header in progress continues in the next launch: its bits move from LDS to the channel's HBM slot
        if (fastp && s.state == SY_RX && s.fstate == FX_HEADER && s.header_symbol_index > 0)
            for (int i = l; i < MCRX_HDR_SYMS; i += WV) bhbits[i] = ldshb[i];
        if (l == 0 && a.seekst && s.state != SY_SEEK) { a.seekst[2 * (size_t)ch] = sk_cur; a.seekst[2 * (size_t)ch + 1] = (int64_t)sk_timer; }      // (a push that ends in SEEK carries nothing over)
        if (l == 0) a.st[ch] = s;
    }
};

// Pointers that arrive inside a by-value struct are generic ("flat") to the compiler: every
// access becomes flat_load/flat_store, which also ties up the LDS/scalar wait counter.  Casting
// through the global address space lets InferAddressSpaces turn them into global_* / s_load.
template <class T> __device__ __forceinline__ T *as_global(T *p)
{
    typedef __attribute__((address_space(1))) T GT;
    return (T *)(GT *)p;
}
#define LAUNDER(f) a.f = as_global(a.f)
__device__ __forceinline__ void launder(SyncArgs &a)
{
    LAUNDER(c.sctype); LAUNDER(c.S0); LAUNDER(c.S1); LAUNDER(c.s0t); LAUNDER(c.smk); LAUNDER(c.smn); LAUNDER(c.Pfit);
    LAUNDER(c.data_rank); LAUNDER(c.pilot_rank); LAUNDER(c.en_rank); LAUNDER(c.pilot_seq); LAUNDER(c.dft_tw);
    LAUNDER(c.cod.crc_byte); LAUNDER(c.crc_pos); LAUNDER(c.il_off); LAUNDER(c.il_map);
    LAUNDER(c.cod.crc_zadv); LAUNDER(c.cod.qam16_nb); LAUNDER(c.cod.qam64_nb);
    LAUNDER(chan); LAUNDER(st); LAUNDER(hbits); LAUNDER(R); LAUNDER(soft); LAUNDER(tmpa); LAUNDER(tmpb);
    LAUNDER(syms); LAUNDER(rec); LAUNDER(arena); LAUNDER(sarena); LAUNDER(nrec); LAUNDER(arena_used);
    LAUNDER(jobs); LAUNDER(njobs); LAUNDER(jR); LAUNDER(jsoft); LAUNDER(jtmp); LAUNDER(vit_scratch); LAUNDER(vit_passes); LAUNDER(qam_list); LAUNDER(hint); LAUNDER(live);
    LAUNDER(spec); LAUNDER(spec_R); LAUNDER(pred); LAUNDER(pred_n); LAUNDER(stats); LAUNDER(walk_hint); LAUNDER(anchor); LAUNDER(seekst);
}
#undef LAUNDER

#if defined(SY_ACQ_WAVES_OVERRIDE)
#define SY_ACQ_WAVES SY_ACQ_WAVES_OVERRIDE
#elif SY_PART == 3
#define SY_ACQ_WAVES 2
#else
#define SY_ACQ_WAVES 1
#endif
template <int E>
__global__ __launch_bounds__(WV) void sync_kernel(SyncArgs a)
{
    launder(a);
    const uint32_t ch = blockIdx.x;
    if (ch >= a.nch) return;
    Walker<E> w(a, ch);
    w.template run<SYM_FULL>();
}

// (the tail kernel is launched twice per push and normally finds nothing to do -- but its waves have to find room first, on SIMDs full of
//  channelizer and worker waves: held to four waves per SIMD (126 registers) where the acquisition kernels get two.  Round 5's first library let it
//  grow from 131 to 212 registers with the 48-point transform in the Walker's fast path: `bench.py --pipeline` 177 -> 168 Gsample/s.)
#if SY_PART == 3 && !defined(SY_TAIL_WAVES)
#define SY_TAIL_WAVES 4
#elif !defined(SY_TAIL_WAVES)
#define SY_TAIL_WAVES SY_ACQ_WAVES
#endif
template <int E>
__global__ __launch_bounds__(WV, SY_TAIL_WAVES) void sync_tail_kernel(SyncArgs a)
{
    __builtin_amdgcn_s_setprio(3);      // a chain of dependent events every payload launch waits for: win the issue arbitration against the workers sharing the SIMD
    launder(a);
    const uint32_t ch = blockIdx.x;
    if (ch >= a.nch) return;
    Walker<E> w(a, ch);
    w.run_tail();
}

// the lean scout (lean configurations only): one wave per channel, acquisition + header + hand-off
// (E <= 2: held to three waves per SIMD = 168 registers, so that a scout fits on a SIMD beside four payload workers of
//  the previous push (4 x 80 of the 512) instead of waiting for a SIMD to drain.  These two kernels are built in their own part with
//  the basic vector-register allocator: under this budget hipcc 7.2's default one reloads 64-bit spills into
//  odd-aligned pairs, which its own verifier rejects.)
template <int E>
__global__ __launch_bounds__(WV, SY_ACQ_WAVES) void sync_lean_kernel(SyncArgs a)
{
    __builtin_amdgcn_s_setprio(3);      // a chain of dependent events every payload launch waits for: win the issue arbitration against the workers sharing the SIMD
    launder(a);
    const uint32_t ch = blockIdx.x;
    if (ch >= a.nch) return;
    Walker<E> w(a, ch);
    w.template run<SYM_LEAN, false>();      // (no seek bursts under the register budget: behind the speculative waves / the chain it adopts, it does not seek)
}

// The lean scout's code without its register budget, for traffic whose frame positions cannot be predicted (every frame its own
// length: launch_sync switches the speculative rounds off there): the budgeted build above spills ~300 registers, harmless
// while it only adopts what the speculative waves found, ruinous when it acquires every frame itself (6.2 ms per slab
// against 1.7 for the full kernel on the same stream).
// SY_WALK_WPB waves per workgroup, a channel and an LDS scratch each (no workgroup barrier: the waves never meet).  Measured with
// four (the walkers on 128 CUs, the other 128 free for the channelizer, whose workgroups otherwise cannot start before the
// one-per-SIMD walkers spread over every CU have finished): the channelizer does overlap then (1.65 -> 0.99 ms), but the
// walkers, now sharing their CUs with all the workers the channelizer displaced, take 2.03 ms instead of 1.65 -- 91.8 against
// 110 Gsample/s on ragged traffic.  One per workgroup it stays.
#define SY_WALK_WPB 1
template <int E>
__global__ __launch_bounds__(WV * SY_WALK_WPB) void sync_walk_kernel(SyncArgs a)
{
    __builtin_amdgcn_s_setprio(3);
    launder(a);
    const uint32_t wv = rfl(threadIdx.x / WV);
    const uint32_t ch = blockIdx.x * SY_WALK_WPB + wv;
    if (ch >= a.nch) return;
    Walker<E> w(a, ch, (int)(wv * (uint32_t)(SY_LDS_BYTES(E * WV) / sizeof(float2))));
    w.template run<SYM_LEAN>();
}

// segment-parallel acquisition: one wave per (channel, segment of the push); lean path only.  Segments of one channel are
// neighbours in the grid, so the waves that re-read the frame at a segment boundary share an XCD's L2 more often than not.
template <int E>
__global__ __launch_bounds__(WV, SY_ACQ_WAVES) void sync_spec_kernel(SyncArgs a)
{
    __builtin_amdgcn_s_setprio(3);      // a chain of dependent events every payload launch waits for: win the issue arbitration against the workers sharing the SIMD
    launder(a);
    const uint32_t ch = a.seg_phase == 1 ? blockIdx.x : blockIdx.x / a.nseg, g = a.seg_phase == 1 ? 0u : blockIdx.x % a.nseg;
    if (ch >= a.nch) return;
    Walker<E> w(a, ch);
    w.run_seg(g);
}

// one wave per handed-off frame.  Two builds: the lean symbol loop (power-of-two M >= 64, <= 64
// pilots: every configuration the reference's applications use) and the general walker; the host
// picks per design, so neither carries the other's registers.
// (four waves per SIMD up to M = 256; wider symbols hold 2 E window registers twice over and get a larger budget)
template <int E, bool FAST>
__global__ __launch_bounds__(WV, ((E == 1 && FAST) ? 6 : E <= 4 ? 4 : (E <= 8 ? 2 : 1))) void payload_kernel(SyncArgs a)
{
    launder(a);
    if (blockIdx.x >= a.live[0]) return;
    const uint32_t j = a.live[1u + blockIdx.x];
    const uint32_t ch = a.jobs[j].ch;
    if (ch >= a.nch) return;
    Walker<E> w(a, ch);
    if constexpr (FAST) w.run_job_fast(j);
    else w.run_job(j);
}

#if SY_PART <= 0        // kernels that do not depend on the symbol width live in part 0 only
// ------------------------------------------------------------------ payload workers, FR frames per wave (M = 64)
// The one-frame-per-wave worker above spends a third of its instructions on things that do not grow with the symbol:
// the pilot phase fit (atan2, unwrap scan, two projections -- executed by 64 lanes for 6 pilots), the oscillator trim,
// the loop itself.  Here a wave takes FR hand-offs, one per group of G = 64 / FR lanes: lane (g, i) holds samples
// i, i + G, ... of its frame's symbol window, the 64-point transform is log2(FR) in-register stages and log2(G) stages
// across the lanes of the group, every group fits its own pilots with row scans in its first DPP row, and what was
// wave-uniform scalar state is group-uniform vector state.  Same arithmetic per sample as run_job_fast (same butterfly
// order and twiddles, same fit, same trim), so the symbols agree with it to the rounding of the fma shapes.
template <int FR>
__global__ __launch_bounds__(WV) void payload_multi_kernel(SyncArgs a)
{
    constexpr int G = WV / FR, E = FR;
    static_assert(FR == 1 || FR == 2 || FR == 4, "groups are whole DPP rows");
    launder(a);
    __shared__ float2 qpil[FR * 16];
    __shared__ uint32_t qps[64];
    __shared__ uint32_t qnb16[16], qnb64[64];                           // the soft demodulator's nearest-neighbour tables: read on every QAM symbol's chain
    const SyncConsts &c = a.c;
    const int l = lane_id(), g = l / G, i = l % G;
    const uint32_t nj = a.live[0];
    const uint32_t k0 = (uint32_t)FR * blockIdx.x;
    if (k0 >= nj) return;
    bool active = k0 + (uint32_t)g < nj;
    const uint32_t j = a.live[1u + (active ? k0 + (uint32_t)g : k0)];
    const uint32_t jj = j;                                              // idle groups shadow the first job, store nothing
    const PayloadJob *job = a.jobs + jj;
    const uint32_t ch_owner = job->ch;                                  // (~0: an entry nobody owns -- a frame a segment wave acquired for nothing, a void reservation)
    active = active && ch_owner < a.nch && job->arena_off != ~0ull;
    const uint32_t ch = ch_owner < a.nch ? ch_owner : 0u;               // idle groups read channel 0's samples and store nothing
    qps[l] = reinterpret_cast<const uint32_t *>(c.pilot_seq)[l];        // 256 bytes, one word per lane
    qnb64[l] = reinterpret_cast<const uint32_t *>(c.cod.qam64_nb)[l];
    if (l < 16) qnb16[l] = reinterpret_cast<const uint32_t *>(c.cod.qam16_nb)[l];
    wave_sync_lds();
    const uint8_t *pseq = reinterpret_cast<const uint8_t *>(qps);

    // ---- per-lane constants: after the transform, sample position p = i + G e holds subcarrier bitrev6(p)
    float2 R[E]; float fxr[E]; int dr[E], pr[E];
    const float2 *bR = a.jR + (size_t)jj * c.M;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int kk = (int)(__brev((unsigned)(i + G * e)) >> 26);
        dr[e] = c.data_rank[kk]; pr[e] = c.pilot_rank[kk];
        fxr[e] = ((kk > c.M2) ? (float)kk - (float)c.M : (float)kk) * 0.15915494309189535f;
        R[e] = c.sctype[kk] ? bR[kk] : make_float2(0.f, 0.f);
    }
    float2 twc[6]; float sgc[6];                                        // cross-lane stages h = 32 >> st (used for h < G)
#pragma unroll
    for (int st = 0; st < 6; st++) {
        const int h = 32 >> st;
        const bool up = (i & h) != 0;
        const float rev = (float)(i & (h - 1)) * (0.5f / (float)h);
        twc[st] = up ? make_float2(__builtin_amdgcn_cosf(rev), -__builtin_amdgcn_sinf(rev)) : make_float2(1.f, 0.f);
        sgc[st] = up ? -1.f : 1.f;
    }
    const int Mp = c.M_pilot;
    const float pf0 = (i < Mp) ? c.Pfit[i] : 0.f, pf1 = (i < Mp) ? c.Pfit[Mp + i] : 0.f;
    // a value of lane 15 of every group's first row, in all lanes of the group
    auto group_total = [&](float v) {
        float r = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 15));
#pragma unroll
        for (int k = 1; k < FR; k++) {
            const float t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k * G + 15));
            r = g == k ? t : r;
        }
        return r;
    };

    // ---- group-uniform state (one frame per wave: wave-uniform, told to the compiler so that it lives in scalars)
    auto uni = [](uint32_t v) { return FR == 1 ? rfl(v) : v; };
    const int L = c.L, cb = c.cp - c.backoff, Md = c.M_data;
    const uint32_t mod = uni(job->s.mod_scheme), bps = uni(job->s.bps), mod_len = uni(job->s.mod_len), nbits = uni(8u * job->s.enc_len);
    const uint32_t nsym = (mod_len + (uint32_t)Md - 1u) / (uint32_t)Md;
    const int64_t t_ev0 = job->s.cur + (int64_t)job->s.timer - 1;
    const int64_t ws0 = t_ev0 - L + 1 + cb;
    uint32_t dth = uni(job->s.nco_dtheta);
    uint32_t th_ws = uni(job->s.nco_theta_ref + (uint32_t)(ws0 - job->s.nco_t_ref) * job->s.nco_dtheta);
    uint32_t pc = uni(job->s.pilot_count);
    float phi_prime = job->s.phi_prime, p1_prime = job->s.p1_prime;
    int32_t r_ws = (int32_t)uni((uint32_t)(ws0 - a.buf_first));
    const float2 *chb = a.chan + ((size_t)a.chan_off + ch) * MCRX_TILE_S;
    const size_t tstride = (size_t)a.chan_stride * MCRX_TILE_S;
    uint8_t *soft = a.jsoft + (size_t)jj * 8 * c.max_enc_len;
    float2 *syms = reinterpret_cast<float2 *>(a.sarena + job->syms_off);
    const bool soft_mode = c.payload_soft != 0;
    uint32_t nmax = 0;
#pragma unroll
    for (int k = 0; k < FR; k++) {
        const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)(active ? nsym : 0u), k * G);
        nmax = v > nmax ? v : nmax;
    }
    const int32_t r_max = (int32_t)(a.end - a.buf_first) - 1;

    auto load_win = [&](int32_t rw, float2 (&x)[E]) {
#pragma unroll
        for (int e = 0; e < E; e++) {
            int32_t r = rw + i + G * e;
            r = r < 0 ? 0 : (r > r_max ? r_max : r);                    // (groups that have finished keep reading in range)
            x[e] = chb[(size_t)(r >> MCRX_TILE_SH) * tstride + (size_t)(r & (MCRX_TILE_S - 1))];
        }
    };
    float2 cur[E], nxt[E];
    load_win(r_ws, cur);
    uint32_t psi = 0;
    for (uint32_t n = 0; n < nmax; n++) {
        const bool live = active && n < nsym;
        load_win(r_ws + L, nxt);
        // ---- oscillator, 64-point DIF transform
#pragma unroll
        for (int e = 0; e < E; e++) cur[e] = rot_down(cur[e], u32rev(th_ws + (uint32_t)(i + G * e) * dth));
#pragma unroll
        for (int J = E / 2; J >= 1; J >>= 1) {                          // in-register stages, span h = G J
#pragma unroll
            for (int e = 0; e < E; e++) {
                if ((e & J) == 0) {
                    const float2 u = cur[e], v = cur[e + J];
                    const float rev = (float)((i + G * e) & (G * J - 1)) * (0.5f / (float)(G * J));
                    cur[e] = cadd(u, v); cur[e + J] = rot_down(csub(u, v), rev);
                }
            }
        }
#define SY_QSTAGE(ST, H)                                                                           \
        if constexpr (H < G) {                                                                     \
            _Pragma("unroll") for (int e = 0; e < E; e++) {                                        \
                const float sx = bfly_leg<H>(cur[e].x, sgc[ST]), sy = bfly_leg<H>(cur[e].y, sgc[ST]);  \
                if (H == 1) cur[e] = make_float2(sx, sy);                                          \
                else cur[e] = make_float2(sx * twc[ST].x - sy * twc[ST].y, sx * twc[ST].y + sy * twc[ST].x); \
            }                                                                                      \
        }
        SY_QSTAGE(0, 32) SY_QSTAGE(1, 16) SY_QSTAGE(2, 8) SY_QSTAGE(3, 4) SY_QSTAGE(4, 2) SY_QSTAGE(5, 1)
#undef SY_QSTAGE
        // ---- equaliser, pilots of this group to the first lanes of its first row
#pragma unroll
        for (int e = 0; e < E; e++) {
            cur[e] = cmul(cur[e], R[e]);
            if (pr[e] >= 0) qpil[16 * g + pr[e]] = cur[e];
        }
        wave_sync_lds();
        float2 P = qpil[16 * g + (i < Mp ? i : 0)];
        uint32_t pi_ = pc + (uint32_t)(i < 16 ? i : 0); pi_ = pi_ >= 255u ? pi_ - 255u : pi_;
        const bool pneg = pseq[pi_ < 255u ? pi_ : 0u] == 0;
        wave_sync_lds();
        if (pneg) { P.x = -P.x; P.y = -P.y; }
        const float v = atan2_fast(P.y, P.x);
        const float prev = dpp_mov<0x111, false>(v, v);                 // row_shr:1, lane 0 of the row keeps its own
        const float turns = rintf((v - prev) * 0.15915494309189535f);
        const float y = fmaf(-TWO_PI_F, row_scan_fast(turns), v);
        const float p0 = group_total(row_scan_fast(pf0 * y));
        float p1 = group_total(row_scan_fast(pf1 * y));
        pc += (uint32_t)Mp; pc = pc >= 255u ? pc - 255u : pc;
        p1 = 0.3f * p1 + (1.0f - 0.3f) * p1_prime;
        p1_prime = p1;
        // ---- de-rotate, soft bits.  The modem is group state; the groups are served one distinct modem at a time
        // (normally one pass) so that the demodulator's case analysis is scalar
        const float p0r = p0 * 0.15915494309189535f;
        const bool full = (psi + (uint32_t)Md) * bps <= nbits;
        float2 Z[E];
#pragma unroll
        for (int e = 0; e < E; e++) Z[e] = rot_down(cur[e], fmaf(p1, fxr[e], p0r));
        unsigned long long pending = __ballot(live);
        while (pending) {
            const int src = __builtin_ctzll(pending);
            const uint32_t umod = (uint32_t)__builtin_amdgcn_readlane((int)mod, src), ubps = (uint32_t)__builtin_amdgcn_readlane((int)bps, src);
            const bool sel = live && mod == umod;
            pending &= ~__ballot(sel);
            const uint8_t *unbt = reinterpret_cast<const uint8_t *>(umod == 27 ? qnb16 : qnb64);
#pragma unroll
            for (int e = 0; e < E; e++) {
                const uint32_t idx = psi + (uint32_t)dr[e];
                const bool ok = sel && dr[e] >= 0 && idx < mod_len;
                uint8_t sb[6];
                const unsigned hs = demod_soft(unbt, umod, Z[e], sb);
                if (!soft_mode) {
#pragma unroll
                    for (int kb = 0; kb < 6; kb++) sb[kb] = (uint8_t)(((hs >> ((ubps - 1 - kb) & 7)) & 1) ? 255 : 0);
                }
                if (!ok) continue;
                syms[idx] = Z[e];
                uint8_t *dst = soft + (size_t)idx * ubps;
                if (full) {
                    if (ubps == 2)      *reinterpret_cast<uint16_t *>(dst) = (uint16_t)(sb[0] | (sb[1] << 8));
                    else if (ubps == 1) dst[0] = sb[0];
                    else {
                        *reinterpret_cast<uint16_t *>(dst) = (uint16_t)(sb[0] | (sb[1] << 8));
                        *reinterpret_cast<uint16_t *>(dst + 2) = (uint16_t)(sb[2] | (sb[3] << 8));
                        if (ubps == 6) *reinterpret_cast<uint16_t *>(dst + 4) = (uint16_t)(sb[4] | (sb[5] << 8));
                    }
                } else {
                    for (unsigned kb = 0; kb < ubps; kb++) if (idx * ubps + kb < nbits) dst[kb] = sb[kb];
                }
            }
        }
        psi += (uint32_t)Md;
        // ---- oscillator trim (liquid ofdmframesync: the phase at the next window start uses the old step up to this event)
        float dphi = p0 - phi_prime;
        dphi -= TWO_PI_F * rintf(dphi * 0.15915494309189535f);
        phi_prime = p0;
        const uint32_t dnew = dth + uni((uint32_t)__float2int_rn(dphi * (1e-3f * 683565275.5764316f)));
        th_ws += (uint32_t)(L - cb) * dth + (uint32_t)cb * dnew;
        if (live) dth = dnew;
        r_ws += L;
#pragma unroll
        for (int e = 0; e < E; e++) cur[e] = nxt[e];
    }
    if (i == 0 && active) a.jobs[j].s.nco_dtheta = dth;
}

#include "payload_lean.hpp"

// ------------------------------------------------------------------ packet decode, a workgroup per frame
// The payload workers leave 8 soft bits per coded byte in HBM.  De-interleaving them there costs
// four passes of scattered 8-byte read-modify-writes per frame (measured: 4.7x the algorithmic HBM
// bytes of the whole payload stage).  Here the frame's soft bits are staged once into LDS
// (coalesced), the four passes run in LDS across 256 threads, and the Hamming(12,8) soft decoder
// reads LDS.  A cell's rank in liquid's column walk is a closed form (columns are valid from row 0
// down to a per-column count), so the passes need no ballots or sequential ranking.
#define DK_T 256
#ifndef DK_FUSED
#define DK_FUSED 1
#endif
#ifndef DK_NT_LOAD
#define DK_NT_LOAD 1        /* the frame's soft bits (read once) as non-temporal loads: 0.153 -> 0.150 ms */
#endif
extern __shared__ __attribute__((aligned(16))) unsigned long long dk_soft[];
// The passes walk the 8-byte groups with strides of ~2 sqrt(n) (one side) and ~sqrt(n)/2 (the other):
// in a linear layout either lands on 4 of the 32 bank pairs.  XOR-folding index bits 5..9 into bits
// 0..4 (a permutation inside every 1024-group block) spreads any such stride over all banks.
#define DKP(e) ((e) ^ (((e) >> 5) & 31u))
// valid cells in the first `len` columns of the walk after the (aliased) first one
__device__ __forceinline__ unsigned il_cols_below(unsigned s0, unsigned len, unsigned Ncol, unsigned rem)
{
    auto span = [&](unsigned a0, unsigned b0) { const unsigned lo = a0 < rem ? a0 : rem, hi = b0 < rem ? b0 : rem; return hi - lo; };
    return (s0 + len <= Ncol) ? span(s0, s0 + len) : span(s0, Ncol) + span(0u, s0 + len - Ncol);
}
// a / b and a % b for a < 2^22, b >= 1 through the float reciprocal (a dozen instructions instead of
// the ~40 of a 32-bit division; the walk geometry is all such small numbers)
__device__ __forceinline__ void divmod_small(unsigned a, unsigned b, unsigned &q, unsigned &r)
{
    q = (unsigned)((float)a * __builtin_amdgcn_rcpf((float)b));
    int rr = (int)a - (int)(q * b);
    if (rr < 0) { q--; rr += (int)b; }
    if (rr >= (int)b) { q++; rr -= (int)b; }
    r = (unsigned)rr;
}
// One de-interleaver pass on the LDS-resident soft bits.  Straight-line per cell (no exec-mask
// branches: cells outside the walk swap a dummy group with itself), three cells per thread in flight.
__device__ void il_pass_lds(unsigned n, unsigned Mi, unsigned Ncol, unsigned mask, unsigned dummy)
{
    const unsigned n2 = n / 2, c0 = n / 3;
    unsigned long long m64 = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) if ((mask >> (7 - k)) & 1) m64 |= 0xFFull << (8 * k);
    unsigned R, rem, cnt0 = 0, t0, s1, c0r;
    divmod_small(n2, Ncol, R, rem);                                      // column c < Ncol holds R + (c < rem) valid cells
    if (c0 < n2) divmod_small(n2 - c0 + Ncol - 1, Ncol, cnt0, t0);      // first column: c0 is not reduced modulo Ncol
    divmod_small(c0, Ncol, t0, c0r);
    s1 = c0r + 1; s1 -= s1 >= Ncol ? Ncol : 0u;
    // The threads take the PAIRS i = 0 .. n/2-1 (not the cells of the walk, 60 % of which are invalid): the
    // i-th valid cell is found by inverting the column counts.  After the first column the walk meets, in
    // order, a columns of R+1 cells (residues s1 .. rem-1), b of R (.. Ncol-1), w of R+1 (0 .. min(rem, s1)-1), then R.
    const unsigned sa = s1 < rem ? rem - s1 : 0u, sb = Ncol - (s1 > rem ? s1 : rem), sw = rem < s1 ? rem : s1;
    const unsigned B1 = sa * (R + 1), B2 = B1 + sb * R, B3 = B2 + sw * (R + 1);
    constexpr int UN = 4;
    for (unsigned i0 = threadIdx.x; i0 < n2; i0 += UN * DK_T) {
        unsigned ea[UN], eb[UN];
        unsigned long long va[UN], vb[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const unsigned i = i0 + u * DK_T;
            const unsigned ip = i - cnt0;                                   // (meaningless, and unused, while i < cnt0)
            const bool g1 = ip < B1, g2 = ip < B2, g3 = ip < B3;
            unsigned d = (g1 || (!g2 && g3)) ? R + 1 : R;
            d = d ? d : 1u;
            const unsigned off = g1 ? ip : g2 ? ip - B1 : g3 ? ip - B2 : ip - B3;
            const unsigned lbase = g1 ? 0u : g2 ? sa : g3 ? sa + sb : sa + sb + sw;
            unsigned q, m; divmod_small(off, d, q, m);
            const unsigned len = lbase + q;                                 // whole columns walked after the first
            unsigned cm = s1 + len; cm -= cm >= Ncol ? Ncol : 0u;
            const bool first = i < cnt0;
            const unsigned j = first ? i * Ncol + c0 : m * Ncol + cm;
            const bool ok = i < n2 && j < n2 && (first || (len < Ncol && m < Mi));
            ea[u] = ok ? DKP(2 * i) : dummy;
            eb[u] = ok ? DKP(2 * j + 1) : dummy;
            va[u] = dk_soft[ea[u]]; vb[u] = dk_soft[eb[u]];
        }
#pragma unroll
        for (int u = 0; u < UN; u++) {
            dk_soft[ea[u]] = (va[u] & ~m64) | (vb[u] & m64);
            dk_soft[eb[u]] = (vb[u] & ~m64) | (va[u] & m64);
        }
    }
    __syncthreads();
}
// CRC-32 of p[0..n) by one wave without long dependent chains through HBM: power-of-two chunks
// aligned to the END of the message (lane r holds the r-th chunk from the right, the leftmost
// one may be short and carries the 0xFFFFFFFF preset), byte table in LDS, then a binary tree in
// which the right block always spans 2^k whole chunks, so every combine is one of the
// precomputed "advance through 2^m zero bytes" operators.
__device__ uint32_t crc32_tree(const CodingDev cod, uint32_t tab_off, uint32_t msg_off, uint32_t n)
{
    const uint32_t *lds_tab = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(dk_soft) + tab_off);
    const uint8_t *p = reinterpret_cast<const uint8_t *>(dk_soft) + msg_off;
    const int r = lane_id();
    uint32_t C = 1, lc = 0;
    while (64u * C < n) { C <<= 1; lc++; }
    const int64_t hi = (int64_t)n - (int64_t)r * C, lo0 = hi - (int64_t)C;
    const int64_t lo = lo0 < 0 ? 0 : lo0;
    const uint32_t rmax = n ? (n + C - 1) / C - 1 : 0;
    uint32_t s = ((uint32_t)r == rmax) ? 0xFFFFFFFFu : 0u;
    for (int64_t i = lo; i < hi; i++) s = (s >> 8) ^ lds_tab[(s ^ p[i]) & 0xff];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const uint32_t v = (uint32_t)__shfl_down((int)s, 1 << k, WV);
        if ((r & ((2 << k) - 1)) == 0 && r + (1 << k) < WV) {
            const uint32_t *A = cod.crc_zadv + (size_t)(lc + k) * 1024;
            s ^= A[v & 0xff] ^ A[256 + ((v >> 8) & 0xff)] ^ A[512 + ((v >> 16) & 0xff)] ^ A[768 + (v >> 24)];
        }
    }
    return ~(uint32_t)__shfl((int)s, 0, WV);
}
__device__ __forceinline__ void decode_frame(SyncArgs &a, const uint32_t j, uint32_t lds_soft_bytes, uint32_t msg_bytes)
{
    const uint32_t ch = a.jobs[j].ch;
    if (ch >= a.nch || a.jobs[j].arena_off == ~0ull) return;
    const SyncConsts &c = a.c;
    const uint32_t n_msg = a.jobs[j].s.payload_len, crc = a.jobs[j].s.check, fec0 = a.jobs[j].s.fec0, fec1 = a.jobs[j].s.fec1;
    const uint32_t e1 = a.jobs[j].s.enc_len;
    uint8_t *soft = a.jsoft + (size_t)j * 8 * c.max_enc_len;
    const uint32_t crc_len = (crc == 6) ? 4u : 0u, n0 = n_msg + crc_len;
    const bool prof = SY_PROF(a) && j == 7;
    long long tk[8]; int ntk = 0;
#define DK_TICK() if (prof) tk[ntk++] = (long long)__builtin_readcyclecounter();
    DK_TICK()
    // LDS path: soft decisions, no inner code, outer code Hamming(12,8) (soft decoder), Golay(24,12) (sliced) or none
    const bool lds_path = c.payload_soft && fec0 == 1 && (fec1 == 6 || fec1 == 7 || fec1 == 1) && 8u * e1 <= lds_soft_bytes && !(a.no_fast & 8);
    if (!lds_path) {                        // everything else: onto the list of decode_general_kernel (one wave per frame, in place in HBM)
        // ... the K = 7 convolutional code as the outer code first gets its soft bits de-interleaved here (the same gather as below,
        // written back in place): the general decoder's wave then runs the trellis with the frame-per-wave decoder (viterbi_frames.hpp)
        // instead of the one-state-per-lane one
        const uint32_t moff = (c.il_off && e1 < c.il_n) ? c.il_off[e1] : ~0u;
        const bool conv_pre = c.payload_soft && fec1 == 11 && a.vit_scratch && moff != ~0u && 8u * e1 <= lds_soft_bytes && !(a.no_fast & 8) &&
                              vf::rows_for(8u * fec_enc_len_d(fec0, n0) + 6u) <= a.vit_rows;
        // (Every frame has an entry of the general list to itself, so nothing has to be reserved before the gather below rewrites the frame's
        //  soft bits in place: the tag on that entry says they are de-interleaved.  Rounds 3-5 kept a second list, of trellis blocks and
        //  then of frames, for a kernel of the decoder's own between this one and the general decoder -- ADVICE r3 / VERDICT r4 #8 were about
        //  a frame that found it full.)
        const bool conv_go = conv_pre;
        // (tell the host that frames with the code arrive: it allocates the frame-per-wave decoder's scratch then -- cfg.conv_scratch = 0)
        if (c.payload_soft && fec1 == 11 && !a.vit_scratch && a.hint && threadIdx.x == 0) a.hint[11] = 1u;
        if (conv_go) {
            const unsigned long long *g64 = reinterpret_cast<const unsigned long long *>(soft);
            for (uint32_t i = threadIdx.x; i < e1; i += DK_T) dk_soft[DKP(i)] = g64[i];
            __syncthreads();
            const uint4 *mp = reinterpret_cast<const uint4 *>(c.il_map + (size_t)moff * 8);
            const uint8_t *sb = reinterpret_cast<const uint8_t *>(dk_soft);
            unsigned long long *o64 = reinterpret_cast<unsigned long long *>(soft);
            for (uint32_t i = threadIdx.x; i < e1; i += DK_T) {
                const uint4 m = mp[i];
                const uint32_t lo = (uint32_t)sb[m.x & 0xffffu] | ((uint32_t)sb[m.x >> 16] << 8) | ((uint32_t)sb[m.y & 0xffffu] << 16) | ((uint32_t)sb[m.y >> 16] << 24);
                const uint32_t hi = (uint32_t)sb[m.z & 0xffffu] | ((uint32_t)sb[m.z >> 16] << 8) | ((uint32_t)sb[m.w & 0xffffu] << 16) | ((uint32_t)sb[m.w >> 16] << 24);
                o64[i] = (unsigned long long)lo | ((unsigned long long)hi << 32);
            }
        }
        if (threadIdx.x == 0 && a.gen_list) {
            uint32_t *gl = as_global(a.gen_list);
            uint32_t tag = j;
            if (conv_go) {
                tag |= 0x80000000u;
            }                                         // (list full: this frame's trellis stays with the general decoder's single wave)
            gl[1u + atomicAdd(gl, 1u)] = tag;
        }
        return;
    }
    if (lds_path) {
        // Hamming(12,8) with a gather table: the decoder takes its 12 soft bits straight from where the frame was staged -- table
        // entries are byte addresses there -- so the de-interleaved copy, its barrier and its re-read never happen, and the first
        // symbols' table entries are requested before the soft bits, one round trip to memory instead of two in a row
        const uint32_t moff_h = (fec1 == 6 && c.il_off && e1 < c.il_n) ? c.il_off[e1] : ~0u;
        const bool fused = DK_FUSED && moff_h != ~0u;
        constexpr int SU = 4;
        const uint2 *mp2 = reinterpret_cast<const uint2 *>(c.il_map + (size_t)(fused ? moff_h : 0u) * 8);
        uint2 pm[SU][3];
        if (fused) {
#pragma unroll
            for (int u = 0; u < SU; u++) {
                const uint32_t i = threadIdx.x + u * DK_T, ii = i < n0 ? i : 0u;
#pragma unroll
                for (int k = 0; k < 3; k++) pm[u][k] = mp2[3u * ii + k];
            }
        }
        const unsigned long long *g64 = reinterpret_cast<const unsigned long long *>(soft);
        for (uint32_t b0 = 0; b0 < e1; b0 += 8 * DK_T) {           // eight requests per thread in flight
            unsigned long long v[8];
#pragma unroll
#if DK_NT_LOAD
            for (int u = 0; u < 8; u++) { const uint32_t i = b0 + u * DK_T + threadIdx.x; v[u] = __builtin_nontemporal_load(g64 + (i < e1 ? i : 0)); }
#else
            for (int u = 0; u < 8; u++) { const uint32_t i = b0 + u * DK_T + threadIdx.x; v[u] = g64[i < e1 ? i : 0]; }
#endif
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint32_t i = b0 + u * DK_T + threadIdx.x; if (i < e1) dk_soft[DKP(i)] = v[u]; }
        }
        __syncthreads();
        DK_TICK()
        // coded packets are interleaved (depth 4): where this coded length has a gather table (every length the LDS path's
        // codes produce up to the handle's payload limit), one pass builds the de-interleaved groups in the second half of
        // the soft area -- 8 byte reads per coded byte -- instead of four in-place passes of swaps over the first
        unsigned long long *cur = dk_soft;
        uint8_t *msg = reinterpret_cast<uint8_t *>(dk_soft) + 2 * lds_soft_bytes;
        if (fused) {
            const uint8_t *sb = reinterpret_cast<const uint8_t *>(dk_soft);
            for (uint32_t b0 = 0; b0 < n0; b0 += SU * DK_T) {
                if (b0) {
#pragma unroll
                    for (int u = 0; u < SU; u++) {
                        const uint32_t i = b0 + threadIdx.x + u * DK_T, ii = i < n0 ? i : 0u;
#pragma unroll
                        for (int k = 0; k < 3; k++) pm[u][k] = mp2[3u * ii + k];
                    }
                }
#pragma unroll
                for (int u = 0; u < SU; u++) {
                    const uint32_t i = b0 + threadIdx.x + u * DK_T;
                    if (i >= n0) break;
                    uint32_t w[3];
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const uint2 m = pm[u][k];
                        w[k] = (uint32_t)sb[m.x & 0xffffu] | ((uint32_t)sb[m.x >> 16] << 8) | ((uint32_t)sb[m.y & 0xffffu] << 16) | ((uint32_t)sb[m.y >> 16] << 24);
                    }
                    msg[i] = (uint8_t)h128_dec_soft_words(w[0], w[1], w[2]);
                }
            }
        } else if (fec1 != 1) {
            const uint32_t moff = (c.il_off && e1 < c.il_n) ? c.il_off[e1] : ~0u;
            if (moff != ~0u) {
                const uint4 *mp = reinterpret_cast<const uint4 *>(c.il_map + (size_t)moff * 8);
                const uint8_t *sb = reinterpret_cast<const uint8_t *>(dk_soft);
                unsigned long long *dst = dk_soft + lds_soft_bytes / 8;
                // (eight table entries per thread requested together: one round trip to the L2-resident table instead of one per entry --
                //  the gather was 10-16 k of the workgroup's ~32 k cycles, most of it these loads' latency)
                for (uint32_t i0 = threadIdx.x; i0 < e1; i0 += 8 * DK_T) {
                    uint4 mm[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) { const uint32_t i = i0 + u * DK_T; mm[u] = mp[i < e1 ? i : 0]; }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const uint32_t i = i0 + u * DK_T;
                        if (i >= e1) break;
                        const uint4 m = mm[u];
                        const uint32_t lo = (uint32_t)sb[m.x & 0xffffu] | ((uint32_t)sb[m.x >> 16] << 8) | ((uint32_t)sb[m.y & 0xffffu] << 16) | ((uint32_t)sb[m.y >> 16] << 24);
                        const uint32_t hi = (uint32_t)sb[m.z & 0xffffu] | ((uint32_t)sb[m.z >> 16] << 8) | ((uint32_t)sb[m.w & 0xffffu] << 16) | ((uint32_t)sb[m.w >> 16] << 24);
                        dst[DKP(i)] = (unsigned long long)lo | ((unsigned long long)hi << 32);
                    }
                }
                cur = dst;
                __syncthreads();
            } else {
                unsigned Mi, Ni; il_dims(e1, Mi, Ni);
                const unsigned dummy = 2 * (lds_soft_bytes / 8) + msg_bytes / 8;      // one spare group behind the message area
                il_pass_lds(e1, Mi, Ni + 8, 0x33, dummy);
                il_pass_lds(e1, Mi, Ni + 4, 0x55, dummy);
                il_pass_lds(e1, Mi, Ni + 2, 0x0f, dummy);
                il_pass_lds(e1, Mi, Ni, 0xff, dummy);
            }
        }
        DK_TICK()
        // decoded bytes (message + CRC key) go to LDS behind the soft bits; the payload leaves for the
        // frame arena from there, coalesced, by the whole workgroup
        const uint32_t *w32 = reinterpret_cast<const uint32_t *>(cur);
        auto word = [&](uint32_t w) { return w32[2u * DKP(w >> 1) + (w & 1u)]; };            // 12 soft bits = 3 words, group-swizzled
        // one coded byte = 8 soft bits sliced at 127, MSB first
        auto slice = [&](uint32_t g) -> unsigned {
            const unsigned long long v = cur[DKP(g)];
            unsigned b = 0;
#pragma unroll
            for (int kb = 0; kb < 8; kb++) b = (b << 1) | ((((unsigned)(v >> (8 * kb)) & 0xffu) > 127u) ? 1u : 0u);
            return b;
        };
        if (fused) {
        } else if (fec1 == 6) {
            for (uint32_t i = threadIdx.x; i < n0; i += DK_T) msg[i] = (uint8_t)h128_dec_soft_words(word(3 * i), word(3 * i + 1), word(3 * i + 2));
        } else if (fec1 == 7) {
            const uint32_t G = n0 / 3, rr = n0 % 3;                         // 6 coded bytes -> 3 message bytes; tail: 3 -> 1
            for (uint32_t g = threadIdx.x; g < G; g += DK_T) {
                const unsigned s0 = golay_dec_sym((slice(6 * g) << 16) | (slice(6 * g + 1) << 8) | slice(6 * g + 2));
                const unsigned s1 = golay_dec_sym((slice(6 * g + 3) << 16) | (slice(6 * g + 4) << 8) | slice(6 * g + 5));
                msg[3 * g]     = (uint8_t)((s0 >> 4) & 0xff);
                msg[3 * g + 1] = (uint8_t)(((s0 << 4) & 0xf0) | ((s1 >> 8) & 0x0f));
                msg[3 * g + 2] = (uint8_t)(s1 & 0xff);
            }
            if (threadIdx.x < rr) {
                const uint32_t b = 6 * G + 3 * threadIdx.x;
                msg[3 * G + threadIdx.x] = (uint8_t)(golay_dec_sym((slice(b) << 16) | (slice(b + 1) << 8) | slice(b + 2)) & 0xff);
            }
        } else {
            for (uint32_t i = threadIdx.x; i < n0; i += DK_T) msg[i] = (uint8_t)slice(i);
        }
        __syncthreads();
        DK_TICK()
        {
            uint32_t *dst = reinterpret_cast<uint32_t *>(a.arena + a.jobs[j].arena_off);     // 8-byte aligned by construction
            const uint32_t *src = reinterpret_cast<const uint32_t *>(msg);
            for (uint32_t i = threadIdx.x; i < (n_msg + 3) / 4; i += DK_T) dst[i] = src[i];   // record space is padded to 16 bytes
        }
        // CRC-32 by position (SyncConsts::crc_pos): every thread XORs the table entries of its bytes, the workgroup
        // XOR-reduces -- one round of independent loads instead of the tree's six dependent operator look-ups
        const bool by_pos = crc_len && c.crc_pos && n_msg >= 4 && n_msg <= c.crc_pos_n;
        if (by_pos) {
            uint32_t acc = 0;
            for (uint32_t i0 = threadIdx.x; i0 < n_msg; i0 += 8 * DK_T) {              // (eight table reads in flight per thread)
                uint32_t t[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t i = i0 + u * DK_T, ii = i < n_msg ? i : n_msg - 1;
                    const uint32_t b = (uint32_t)msg[ii] ^ (ii < 4 ? 0xffu : 0u);
                    t[u] = c.crc_pos[(size_t)(n_msg - 1 - ii) * 256 + b];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) if (i0 + u * DK_T < n_msg) acc ^= t[u];
            }
            acc = wave_xor_u32(acc);
            if ((threadIdx.x & (WV - 1)) == 0) reinterpret_cast<uint32_t *>(dk_soft)[256 + threadIdx.x / WV] = acc;      // (behind the byte table below)
        }
        reinterpret_cast<uint32_t *>(dk_soft)[threadIdx.x] = c.cod.crc_byte[threadIdx.x];     // soft bits are spent: byte table in their place
        __syncthreads();
        if (threadIdx.x >= WV) return;
    }
    bool valid = true;
    if (crc_len) {
        const uint8_t *msg = reinterpret_cast<const uint8_t *>(dk_soft) + 2 * lds_soft_bytes;
        const uint32_t key = ((uint32_t)msg[n_msg] << 24) | ((uint32_t)msg[n_msg + 1] << 16) |
                             ((uint32_t)msg[n_msg + 2] << 8) | (uint32_t)msg[n_msg + 3];
        const bool by_pos = c.crc_pos && n_msg >= 4 && n_msg <= c.crc_pos_n;
        if (by_pos) {
            const uint32_t *part = reinterpret_cast<const uint32_t *>(dk_soft) + 256;
            uint32_t tot = 0;
#pragma unroll
            for (int wv = 0; wv < DK_T / WV; wv++) tot ^= part[wv];
            valid = ~tot == key;
        } else valid = crc32_tree(c.cod, 0u, 2 * lds_soft_bytes, n_msg) == key;
    }
    DK_TICK()
    Walker<1> w(a, ch);
    const PayloadJob job = a.jobs[j];
    if (!w.bind_job(j, job)) return;
    const int64_t nsym = (int64_t)((w.s.mod_len + (uint32_t)c.M_data - 1) / (uint32_t)c.M_data);
    w.emit(w.s.cur + (int64_t)w.s.timer - 1 + (nsym - 1) * (int64_t)c.L, true, valid, false, /*copy_payload=*/false);
    DK_TICK()
    if (prof && threadIdx.x == 0) printf("[prof] decode wg7 cycles: stage %lld  passes %lld  h128 %lld  crc %lld  emit %lld\n", tk[1]-tk[0], tk[2]-tk[1], tk[3]-tk[2], tk[4]-tk[3], tk[5]-tk[4]);
#undef DK_TICK
}
// a workgroup per frame of the launch (live list), grid stride: the host sizes the grid from the previous launch's frame count
__global__ __launch_bounds__(DK_T) void decode_kernel(SyncArgs a, uint32_t lds_soft_bytes, uint32_t msg_bytes)
{
    launder(a);
    uint32_t nl = a.live[0];
    if (nl > a.max_jobs) nl = a.max_jobs;
    for (uint32_t k = blockIdx.x; k < nl; k += gridDim.x) {
        const uint32_t j = a.live[1u + k];
        if (a.dec_phase && j < a.max_jobs) {                    // (kernels.h, split_rest: whose frame is this -- the main workers' or the launch's behind them?)
            const uint32_t mod = a.jobs[j].s.mod_scheme;
            const bool main_frame = k < a.live_off && (mod == 39 || mod == 40);
            if (main_frame != (a.dec_phase == 1)) continue;
        }
        decode_frame(a, j, lds_soft_bytes, msg_bytes);
        __syncthreads();                                        // (the next frame reuses the staging area)
    }
}

// The frames the LDS path does not take (hard decisions, an inner code, the convolutional code, frames longer than the
// LDS sized for this launch): one wave per frame decodes in place in HBM with the walker's general decoder -- a separate
// kernel so that its registers (the Viterbi decoder's among them) do not set the occupancy of the one above.
// One launch for everything decode_kernel did not finish (round 5; rounds 3-5 had a kernel of the K = 7 decoder's own in front of it):
//   tagged entries  frames with the K = 7 code as the outer code, soft bits de-interleaved already: the frame-per-wave decoder
//                   (viterbi_frames.hpp; decision rows in the workgroup's region of a.vit_scratch), then inner code, CRC, delivery;
//   the rest        hard decisions, an inner code, the short block codes, frames longer than the LDS sized for the launch.
// It is launched for every push and normally finds its list empty -- but an empty launch still has to find room for its waves on SIMDs
// full of channelizer and worker waves, and the work stream waits for it: held to four waves per SIMD (128 registers: the K = 7 decoder
// fits, the general decoder's rare paths spill).  With 212 registers (the Walker it builds for the delivery grew with the 48-point
// transform) and the K = 7 kernel's launch in front, `bench.py --pipeline` had lost 5 %: 177 -> 168 Gsample/s.
__global__ __launch_bounds__(WV) __attribute__((amdgpu_num_vgpr(128))) void decode_general_kernel(SyncArgs a)
{
    launder(a);
    __shared__ uint16_t vf_ck[2 * 64 * 64];
    const uint32_t *gl = as_global(a.gen_list);
    uint32_t ng = gl[0];
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.list_hint) a.list_hint[2] = ng;
    if (ng > a.max_jobs) ng = a.max_jobs;
    const SyncConsts &c = a.c;
    const bool have_rows = a.vit_scratch && blockIdx.x < a.vit_waves;       // (the launcher keeps the grid inside the scratch: every workgroup has a region)
    uint2 *rows = a.vit_scratch + (have_rows ? (size_t)blockIdx.x * a.vit_rows * 64u : 0u);
    for (uint32_t k = blockIdx.x; k < ng; k += gridDim.x) {        // (a handful of workgroups; the list is normally empty)
        const bool pre_done = (gl[1 + k] & 0x80000000u) != 0;      // de-interleaved already, the K = 7 code outermost
        const uint32_t j = gl[1 + k] & 0x7fffffffu;
        const uint32_t ch = a.jobs[j].ch;
        if (ch >= a.nch || a.jobs[j].arena_off == ~0ull) continue;
        const uint32_t n_msg = a.jobs[j].s.payload_len, crc = a.jobs[j].s.check, fec0 = a.jobs[j].s.fec0, fec1 = a.jobs[j].s.fec1;
        const size_t tstride = (size_t)c.max_enc_len + 16;
        uint8_t *soft = a.jsoft + (size_t)j * 8 * c.max_enc_len;
        uint8_t *tmpa = a.jtmp + (size_t)j * 2 * tstride, *tmpb = tmpa + tstride;
        if (pre_done) {
            const uint32_t e0 = fec_enc_len_d(fec0, n_msg + ((crc == 6) ? 4u : 0u));
            if (have_rows) vf::decode_frame(soft, e0, tmpa, rows, vf_ck, a.vit_passes);
            else conv27_decode_wave(VitSym{ soft, false }, e0, tmpa, reinterpret_cast<uint16_t *>(tmpb), dk_soft);
            __syncthreads();
        }
        const bool valid = packet_decode(c.cod, c.payload_soft != 0, false, n_msg, crc, fec0, fec1, soft, tmpa, tmpb, dk_soft, MCRX_DEVEL_ABLATE(a), pre_done);   // 8 KB of LDS: the one-wave Viterbi decoder's block scratch
        Walker<1> w(a, ch);
        const PayloadJob job = a.jobs[j];
        if (!w.bind_job(j, job)) continue;
        const int64_t nsym = (int64_t)((w.s.mod_len + (uint32_t)c.M_data - 1) / (uint32_t)c.M_data);
        w.emit(w.s.cur + (int64_t)w.s.timer - 1 + (nsym - 1) * (int64_t)c.L, true, valid, false, /*copy_payload=*/true);
        __syncthreads();
    }
}

// Between the scouts and the workers of a launch: one wave of housekeeping.  (Record space is reserved by the scouts themselves,
// a block per channel: Walker::place_owned.)  The lists this launch's workers and decoders fill are emptied, the NEXT launch's job
// counter and QAM list are zeroed, and what the host sizes the next launches by goes to its mapped words.
__global__ __launch_bounds__(WV) void place_jobs_kernel(SyncArgs a)
{
    launder(a);
    if (threadIdx.x != 0) return;
    if (a.gen_list) as_global(a.gen_list)[0] = 0;
    uint32_t nj = *a.njobs;
    if (nj > a.max_jobs) nj = a.max_jobs;
    if (a.njobs_next) *a.njobs_next = 0;
    if (a.qam_next) as_global(a.qam_next)[0] = 0;
    if (a.live_next) as_global(a.live_next)[0] = 0;
    { uint32_t nl = a.live[0]; if (nl > a.max_jobs) a.live[0] = a.max_jobs; }
    if (a.qam_list) {
        uint32_t nq = a.qam_list[0];
        if (nq > a.max_jobs) { nq = a.max_jobs; a.qam_list[0] = nq; }
        if (a.list_hint) a.list_hint[0] = nq;
    }
    if (a.stats) {
        const uint32_t maxenc = a.stats[7];
        a.stats[7] = 0;
        if (a.hint && maxenc) *a.hint = maxenc;                 // host-mapped: sizes the next launch's decode LDS
    }
    if (a.walk_hint && a.stats) {
        // frames the scouts acquired themselves / adopted from segment waves so far, frames on a cadence: the host reads them without
        // a sync (mcrx_hip.hip launch_sync)
        volatile uint32_t *h = a.walk_hint;
        h[0] = a.stats[0]; h[1] = a.stats[1]; h[2] = a.stats[4]; h[3] = a.stats[5]; h[4] = a.stats[2];
        const uint32_t tot = a.stats[0] + a.stats[1], prev = a.stats[6] <= tot ? a.stats[6] : 0u;      // (a statistics reset zeroes them all)
        a.stats[6] = tot;
        h[5] = tot - prev;                      // frames of THIS launch: frames per channel and push, which sizes the next launches' segments
        __threadfence_system();
    }
}

// The packet de-interleaver as a gather table: push the index of every coded byte (low and high half, as soft-bit groups)
// through the inverse interleaver itself and read off where each soft bit came from.  One wave per coded length.
__global__ __launch_bounds__(WV) void ilmap_build_kernel(const uint32_t *lens, const uint32_t *offs, uint8_t *lo, uint8_t *hi, uint16_t *map)
{
    const uint32_t e = lens[blockIdx.x];
    const size_t o = (size_t)offs[blockIdx.x] * 8;
    uint8_t *xl = lo + o, *xh = hi + o;
    for (uint32_t i = threadIdx.x; i < 8 * e; i += WV) { xl[i] = (uint8_t)((i >> 3) & 0xff); xh[i] = (uint8_t)((i >> 3) >> 8); }
    __syncthreads();
    deinterleave<true>(xl, e, 4);
    deinterleave<true>(xh, e, 4);
    __syncthreads();
    // stored as the soft bit's byte address in decode_kernel's staging area (groups swizzled by DKP): nothing left to compute per read
    for (uint32_t i = threadIdx.x; i < 8 * e; i += WV) { const unsigned src = xl[i] | ((unsigned)xh[i] << 8); map[o + i] = (uint16_t)(8u * DKP(src) + (i & 7u)); }
}
hipError_t ilmap_build_launch(const uint32_t *d_lens, const uint32_t *d_offs, uint32_t nlen, uint8_t *d_lo, uint8_t *d_hi, uint16_t *d_map, hipStream_t st)
{
    if (!nlen) return hipSuccess;
    hipLaunchKernelGGL(ilmap_build_kernel, dim3(nlen), dim3(WV), 0, st, d_lens, d_offs, d_lo, d_hi, d_map);
    return hipGetLastError();
}

// restart in one launch: synchronizers back to SEEK, channelizer history cleared, result counters zeroed
__global__ void sync_reset_kernel(ChanState *st, uint32_t nch, int64_t cur, float4 *z0, float4 *z1, size_t nz,
                                  uint32_t *nrec, unsigned long long *arena_used, uint32_t *pred_n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nz) { const float4 z = make_float4(0.f, 0.f, 0.f, 0.f); if (z0) z0[i] = z; if (z1) z1[i] = z; }
    if (i == 0) { if (nrec) { nrec[0] = 0; nrec[1] = 0; } if (arena_used) { arena_used[0] = 0; arena_used[1] = 0; } }
    if (i >= nch) return;
    if (pred_n) pred_n[i] = 0;          // a restarted stream has no history to predict frame positions from
    ChanState z;
    memset(&z, 0, sizeof(z));
    z.state = SY_SEEK; z.cur = cur; z.g0 = 1.0f; z.fstate = FX_HEADER;
    st[i] = z;
}

hipError_t sync_reset_launch(ChanState *st, uint32_t nch, int64_t cur, float2 *hist0, float2 *hist1, size_t hist_n,
                             uint32_t *nrec, unsigned long long *arena_used, uint32_t *pred_n, hipStream_t stream)
{
    const size_t nz = hist_n / 2;                               // float4 = two cf32
    const size_t n = nz > nch ? nz : nch;
    hipLaunchKernelGGL(sync_reset_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, st, nch, cur,
                       reinterpret_cast<float4 *>(hist0), reinterpret_cast<float4 *>(hist1), nz, nrec, arena_used, pred_n);
    return hipGetLastError();
}

#endif  // SY_PART <= 0

// ---- launchers.  The file is compiled in three parts (-DSY_PART=0/1/2: symbol widths E = 1,2 / 4,8 / 16) so that
// the template instantiations build in parallel; every part defines the per-width launchers of its widths.
enum { SYK_SCOUT = 0, SYK_SPEC = 1, SYK_PAYLOAD_FAST = 2, SYK_PAYLOAD_GENERAL = 3, SYK_LEAN = 4, SYK_TAIL = 5, SYK_WALK = 6 };
// acquisition kernels of the narrow symbols (E = 1, 2): part 3, built with -mllvm -vgpr-regalloc=basic (see SY_ACQ_BUDGET)
hipError_t sy_launch_acq_e1(int what, const SyncArgs &a, unsigned grid, size_t lds, hipStream_t st);
hipError_t sy_launch_acq_e2(int what, const SyncArgs &a, unsigned grid, size_t lds, hipStream_t st);
template <int EE>
static hipError_t sy_launch_acq_width(int what, const SyncArgs &a, unsigned grid, size_t lds, hipStream_t st)
{
    if (what == SYK_SPEC) hipLaunchKernelGGL((sync_spec_kernel<EE>), dim3(grid), dim3(WV), lds, st, a);
    else if (what == SYK_TAIL) hipLaunchKernelGGL((sync_tail_kernel<EE>), dim3(grid), dim3(WV), lds, st, a);
    else hipLaunchKernelGGL((sync_lean_kernel<EE>), dim3(grid), dim3(WV), lds, st, a);
    return hipGetLastError();
}
template <int EE>
static hipError_t sy_launch_width(int what, const SyncArgs &a, unsigned grid, size_t lds, hipStream_t st)
{
    switch (what) {
    case SYK_SCOUT:           hipLaunchKernelGGL((sync_kernel<EE>), dim3(grid), dim3(WV), lds, st, a); break;
    case SYK_SPEC:
        if constexpr (EE == 1) return sy_launch_acq_e1(what, a, grid, lds, st);
        else if constexpr (EE == 2) return sy_launch_acq_e2(what, a, grid, lds, st);
        else hipLaunchKernelGGL((sync_spec_kernel<EE>), dim3(grid), dim3(WV), lds, st, a);
        break;
    case SYK_TAIL:
        if constexpr (EE == 1) return sy_launch_acq_e1(what, a, grid, lds, st);
        else if constexpr (EE == 2) return sy_launch_acq_e2(what, a, grid, lds, st);
        else hipLaunchKernelGGL((sync_kernel<EE>), dim3(grid), dim3(WV), lds, st, a);       // (a.tail_only set by the caller)
        break;
    case SYK_LEAN:
        // (wider symbols keep the full kernel as their scout: hipcc 7.2 cannot allocate the lean one's 64-bit values
        //  at E >= 4 under any budget worth having)
        if constexpr (EE == 1) return sy_launch_acq_e1(what, a, grid, lds, st);
        else if constexpr (EE == 2) return sy_launch_acq_e2(what, a, grid, lds, st);
        else hipLaunchKernelGGL((sync_kernel<EE>), dim3(grid), dim3(WV), lds, st, a);
        break;
    case SYK_WALK:
        if constexpr (EE <= 2) hipLaunchKernelGGL((sync_walk_kernel<EE>), dim3((grid + SY_WALK_WPB - 1) / SY_WALK_WPB), dim3(WV * SY_WALK_WPB), SY_WALK_WPB * SY_LDS_BYTES(EE * WV), st, a);
        else hipLaunchKernelGGL((sync_kernel<EE>), dim3(grid), dim3(WV), lds, st, a);
        break;
    case SYK_PAYLOAD_FAST:    hipLaunchKernelGGL((payload_kernel<EE, true>), dim3(grid), dim3(WV), lds, st, a); break;
    case SYK_PAYLOAD_GENERAL: hipLaunchKernelGGL((payload_kernel<EE, false>), dim3(grid), dim3(WV), lds, st, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t sy_launch_e1(int what, const SyncArgs &a, unsigned grid, size_t lds, hipStream_t st);
hipError_t sy_launch_e2(int what, const SyncArgs &a, unsigned grid, size_t lds, hipStream_t st);
hipError_t sy_launch_e4(int what, const SyncArgs &a, unsigned grid, size_t lds, hipStream_t st);
hipError_t sy_launch_e8(int what, const SyncArgs &a, unsigned grid, size_t lds, hipStream_t st);
hipError_t sy_launch_e16(int what, const SyncArgs &a, unsigned grid, size_t lds, hipStream_t st);
#if SY_PART < 0 || SY_PART == 0
hipError_t sy_launch_e1(int what, const SyncArgs &a, unsigned grid, size_t lds, hipStream_t st) { return sy_launch_width<1>(what, a, grid, lds, st); }
hipError_t sy_launch_e2(int what, const SyncArgs &a, unsigned grid, size_t lds, hipStream_t st) { return sy_launch_width<2>(what, a, grid, lds, st); }
#endif
#if SY_PART < 0 || SY_PART == 1
hipError_t sy_launch_e4(int what, const SyncArgs &a, unsigned grid, size_t lds, hipStream_t st) { return sy_launch_width<4>(what, a, grid, lds, st); }
hipError_t sy_launch_e8(int what, const SyncArgs &a, unsigned grid, size_t lds, hipStream_t st) { return sy_launch_width<8>(what, a, grid, lds, st); }
#endif
#if SY_PART < 0 || SY_PART == 2
hipError_t sy_launch_e16(int what, const SyncArgs &a, unsigned grid, size_t lds, hipStream_t st) { return sy_launch_width<16>(what, a, grid, lds, st); }
#endif
#if SY_PART < 0 || SY_PART == 3
hipError_t sy_launch_acq_e1(int what, const SyncArgs &a, unsigned grid, size_t lds, hipStream_t st) { return sy_launch_acq_width<1>(what, a, grid, lds, st); }
hipError_t sy_launch_acq_e2(int what, const SyncArgs &a, unsigned grid, size_t lds, hipStream_t st) { return sy_launch_acq_width<2>(what, a, grid, lds, st); }
#endif

// part 4 = the lean segment waves of the 64-subcarrier configurations (acq_lean.hpp)
hipError_t acq_lean_launch(const SyncArgs &a, unsigned grid, hipStream_t st);
int acq_lean_waves(const SyncConsts &c);        // waves per SIMD of the lean segment-wave kernel if this design takes it, else 0
#if SY_PART < 0 || SY_PART == 4
#include "acq_lean.hpp"
int acq_lean_waves(const SyncConsts &c) { return ((c.M == 64 || c.M == 48) && c.E == 1 && c.M_pilot <= 16 && c.M_pilot >= 1 && c.Nen <= 64) ? ACQ_LEAN_WAVES : 0; }
hipError_t acq_lean_launch(const SyncArgs &a, unsigned grid, hipStream_t st)
{
    if (a.c.M == 48) hipLaunchKernelGGL((acq_lean_kernel<63, 48>), dim3(grid), dim3(WV), 0, st, a);
    else hipLaunchKernelGGL((acq_lean_kernel<63, 64>), dim3(grid), dim3(WV), 0, st, a);
    return hipGetLastError();
}
#endif

// part 1 also holds the lean payload workers of the 128- / 256-subcarrier configurations (payload_wide.hpp)
hipError_t payload_wide_launch(const SyncArgs &a, unsigned grid, hipStream_t st);
static inline bool payload_wide_takes(const SyncConsts &c) { return (c.E == 2 || c.E == 4) && c.log2M >= 7 && c.M == WV * c.E && c.M_pilot >= 1 && c.M_pilot <= WV; }
#if SY_PART < 0 || SY_PART == 1
#include "payload_wide.hpp"
hipError_t payload_wide_launch(const SyncArgs &a, unsigned grid, hipStream_t st)
{
    if (a.c.E == 2) hipLaunchKernelGGL((payload_wide_kernel<63, 2>), dim3(grid), dim3(WV), 0, st, a);
    else hipLaunchKernelGGL((payload_wide_kernel<63, 4>), dim3(grid), dim3(WV), 0, st, a);
    return hipGetLastError();
}
#endif

#if SY_PART <= 0
static hipError_t sy_launch(int what, const SyncArgs &a0, unsigned grid, size_t lds, hipStream_t st)
{
    // kernels that can end up decoding a packet themselves get the Viterbi decoder's block scratch behind their LDS
    SyncArgs a = a0;
    a.vit_off = 0;
    if (what == SYK_SCOUT || what == SYK_TAIL || what == SYK_PAYLOAD_GENERAL || ((what == SYK_LEAN || what == SYK_WALK) && a.c.E > 2)) {
        a.vit_off = (uint32_t)((lds + 15) & ~(size_t)15);
        lds = a.vit_off + (size_t)VIT_B * 8;
    }
    switch (a.c.E) {
    case 1:  return sy_launch_e1(what, a, grid, lds, st);
    case 2:  return sy_launch_e2(what, a, grid, lds, st);
    case 4:  return sy_launch_e4(what, a, grid, lds, st);
    case 8:  return sy_launch_e8(what, a, grid, lds, st);
    case 16: return sy_launch_e16(what, a, grid, lds, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t sync_launch(const SyncArgs &a, hipStream_t st)
{
    if (a.nch == 0) return hipSuccess;
    if (a.c.M > SY_MAXM) return hipErrorInvalidValue;
    return sy_launch(SYK_SCOUT, a, a.nch, SY_LDS_BYTES(a.c.M), st);
}

hipError_t sync_launch_lean(const SyncArgs &a, hipStream_t st)
{
    if (a.nch == 0) return hipSuccess;
    if (a.c.M > SY_MAXM) return hipErrorInvalidValue;
    return sy_launch(SYK_LEAN, a, a.nch, SY_LDS_BYTES(a.c.M), st);
}

hipError_t sync_launch_walk(const SyncArgs &a, hipStream_t st)
{
    if (a.nch == 0) return hipSuccess;
    if (a.c.M > SY_MAXM) return hipErrorInvalidValue;
    return sy_launch(SYK_WALK, a, a.nch, SY_LDS_BYTES(a.c.M), st);
}

hipError_t sync_launch_tail(const SyncArgs &a, hipStream_t st)
{
    if (a.nch == 0) return hipSuccess;
    return sy_launch(SYK_TAIL, a, a.nch, SY_LDS_BYTES(a.c.M), st);
}

hipError_t sync_launch_spec(const SyncArgs &a, hipStream_t st)
{
    if (a.nch == 0 || a.spec_cap == 0 || a.nseg == 0) return hipSuccess;
    const unsigned grid = a.seg_phase == 1 ? a.nch : a.nch * a.nseg;
    // scout_build = 2 and 48 / 64 subcarriers with the pilots inside one DPP row: the lean segment waves (acq_lean.hpp); else the Walker's
    if (acq_lean_waves(a.c) && !a.seg_walker) return acq_lean_launch(a, grid, st);
    return sy_launch(SYK_SPEC, a, grid, SY_LDS_BYTES(a.c.M), st);
}

// bytes of LDS a frame's soft bits get in this push's decode launch: 8 per coded byte, at most 56 KiB (longer frames
// decode in HBM); sized from the longest frame the previous launch saw -- a.enc_hint, read without a sync -- so that more
// workgroups fit a CU; 0 = no history yet: size for the configured maximum
static uint32_t decode_soft_lds(const SyncArgs &a)
{
    const uint32_t enc_cap = a.enc_hint ? ((a.enc_hint + 127u) & ~127u) : a.c.max_enc_len;
    size_t soft_lds = (size_t)8 * (enc_cap < a.c.max_enc_len ? enc_cap : a.c.max_enc_len);
    if (soft_lds > 56 * 1024) soft_lds = 56 * 1024;      // (twice that is the workgroup's soft area: 112 KB + the message)
    if (soft_lds < 4096) soft_lds = 4096;
    return (uint32_t)soft_lds;
}

#ifndef SY_REST_FLOOR
#define SY_REST_FLOOR 256u
#endif
#ifndef SY_GEN_FLOOR
#define SY_GEN_FLOOR 128u
#endif
// does this design's payload stage consist of the lean one-frame-per-wave workers and a launch behind them (kernels.h, split_rest)?
bool sync_payload_splits(const SyncArgs &a)
{
    const bool fast = ((a.c.log2M >= 6 && a.c.M == WV * a.c.E) || (a.c.M == 48 && a.c.E == 1)) && a.c.M_pilot <= WV && !(a.no_fast & 1);
    return a.scout && fast && (a.c.M == WV || a.c.M == 48) && a.c.M_pilot <= 16 && a.payload_fr == 1 && a.payload_lean && !(a.no_fast & 6);
}
hipError_t sync_launch_payload(const SyncArgs &a0, int stage, hipStream_t st)
{
    if (a0.nch == 0 || !a0.scout || a0.max_jobs == 0) return hipSuccess;
    SyncArgs a = a0;
    const unsigned nj = a.max_jobs;
    // frames of the most recent launch + a quarter: the workers and decoders walk the live list with a grid stride, so a launch that
    // holds more than expected only takes a second turn
    unsigned ngrid = a.frames_hint == ~0u ? nj : a.frames_hint + a.frames_hint / 4 + 64;
    if (ngrid > nj) ngrid = nj;
    a.live_off = 0;
    const bool fast = ((a.c.log2M >= 6 && a.c.M == WV * a.c.E) || (a.c.M == 48 && a.c.E == 1)) && a.c.M_pilot <= WV && !(a.no_fast & 1);
    const bool m64or48 = a.c.M == WV || a.c.M == 48;                        // the symbol widths of the lean one-frame-per-wave workers (payload_lean.hpp)
    // the M = 64 lean workers take one BPSK / QPSK frame per wave out of the first `ngrid` of the live list; the launch behind them
    // walks the rest of it (live_off tells it where the grid ended)
    if (fast && m64or48 && a.c.M_pilot <= 16 && a.payload_fr == 1 && a.payload_lean && !(a.no_fast & 6)) a.live_off = ngrid;
    if (stage == 4 && !a.live_off) return hipSuccess;                    // (only the lean workers have a launch behind them)
    if (!a.live_off) { a.split_rest = 0; a.dec_phase = 0; }
    a.dec_lds_soft = fast ? decode_soft_lds(a) : 0u;
    if (!fast) a.gen_list = nullptr;
    if (stage == 0) {
        hipLaunchKernelGGL(place_jobs_kernel, dim3(1), dim3(WV), 0, st, a);
        return hipGetLastError();
    }
    if (stage == 2) {
        if (!fast) return hipSuccess;
        const size_t soft_lds = a.dec_lds_soft;
        const size_t msg_lds = ((size_t)a.c.max_payload_len + 4 + 15) & ~(size_t)15;
        hipLaunchKernelGGL(decode_kernel, dim3(ngrid), dim3(DK_T), 2 * soft_lds + msg_lds + 16, st, a, (uint32_t)soft_lds, (uint32_t)msg_lds);      // soft bits as received | de-interleaved | message
        return hipGetLastError();
    }
    if (stage == 3) {                       // the frames on the general list (filled by decode_kernel)
        if (!fast) return hipSuccess;

        // (grids from the lists' most recent sizes: kernels.h, list_hint)
        // (Grid from the list's most recent size -- kernels.h, list_hint -- inside the K = 7 decoder's scratch, a region per workgroup.
        //  Capping the workgroups per CU with unused dynamic LDS -- 4 = one wave per SIMD, 8, 2 -- changed nothing or cost, when the
        //  decoder had a kernel of its own: it is bound by vector issue at any occupancy, ~0.17 us per 1200-byte frame with the chip
        //  full, ~0.15 ms for a wave from start to end: scratch/r5/v27_pad.sh.)
        unsigned ggrid = a.grid_hint[2] ? (nj < 4096 ? nj : 4096) : SY_GEN_FLOOR;
        if (a.vit_scratch && ggrid > a.vit_waves) ggrid = a.vit_waves;
        hipLaunchKernelGGL(decode_general_kernel, dim3(ggrid), dim3(WV), (size_t)VIT_B * 8, st, a);
        return hipGetLastError();
    }
    // M = 64 with the pilots inside one DPP row: payload_multi_kernel, MCRX_PAYLOAD_FR frames per wave (default 1;
    // 0 = the width-generic worker).  Measured on the bench stream: 1 -> 141.7 Gsample/s, 0 -> 138, 2 -> 138, 4 -> 118.
    const int fr = a.payload_fr;                                         // (MCRX_PAYLOAD_FR / _LEAN / _XB: read once, when the handle is created)
    if (fast && a.c.M == 48 && a.c.M_pilot <= 16 && fr == 1 && a.payload_lean && !(a.no_fast & 6)) {
        // 48 subcarriers: the lean workers' 3 x 16 build (the packed-frame and round-2 workers are 64-subcarrier kernels)
        unsigned nq = a.grid_hint[0] == ~0u ? nj : 2u * a.grid_hint[0];
        nq = nq < SY_REST_FLOOR ? SY_REST_FLOOR : (nq > nj ? nj : nq);
        const size_t pad = (size_t)a.payload_lds_pad;
        if (stage != 4) hipLaunchKernelGGL((payload_lean_kernel<63, 48>), dim3(ngrid), dim3(WV), pad, st, a);
        if (stage == 4 || !a.split_rest) hipLaunchKernelGGL((payload_lean_rest_kernel<63, 48>), dim3(a.qam_list ? nq : nj), dim3(WV), pad, st, a);
        return hipGetLastError();
    }
    if (fast && a.c.M == WV && a.c.M_pilot <= 16 && fr > 0 && !(a.no_fast & 6)) {
        if (fr == 4)      hipLaunchKernelGGL(payload_multi_kernel<4>, dim3((nj + 3) / 4), dim3(WV), 0, st, a);
        else if (fr == 2) hipLaunchKernelGGL(payload_multi_kernel<2>, dim3((nj + 1) / 2), dim3(WV), 0, st, a);
        else {
            // one frame per wave: the lean build (payload_lean.hpp) unless MCRX_PAYLOAD_LEAN=0; MCRX_PAYLOAD_XB picks which
            // butterfly stages exchange through the LDS crossbar (bit H set: the stage with partners H lanes apart)
            const int xb = a.payload_xb;
            const size_t pad = (size_t)a.payload_lds_pad;
            if (!a.payload_lean) hipLaunchKernelGGL(payload_multi_kernel<1>, dim3(nj), dim3(WV), pad, st, a);
            else {
                // (... and one list-driven launch for what the main one does not take: frames beyond its grid, QAM payloads; sized by the
                //  QAM list's most recent length)
                unsigned nq = a.grid_hint[0] == ~0u ? nj : 2u * a.grid_hint[0];
                nq = nq < SY_REST_FLOOR ? SY_REST_FLOOR : (nq > nj ? nj : nq);
#define SY_LEAN(XB) do { if (stage != 4) hipLaunchKernelGGL(payload_lean_kernel<XB>, dim3(ngrid), dim3(WV), pad, st, a); \
                         if (stage == 4 || !a.split_rest) hipLaunchKernelGGL(payload_lean_rest_kernel<XB>, dim3(a.qam_list ? nq : nj), dim3(WV), pad, st, a); } while (0)
                if (xb == 0) SY_LEAN(0);
                else         SY_LEAN(63);
#undef SY_LEAN
            }
        }
        return hipGetLastError();
    }
    // 128 / 256 subcarriers: the lean one-frame-per-wave workers of payload_wide.hpp (round 6) unless the handle asks for the width-generic
    // kernel (worker_build = 5, or payload_lean off: A/B and parity tests)
    if (fast && payload_wide_takes(a.c) && a.payload_lean && a.payload_fr == 1 && !(a.no_fast & 6))
        return payload_wide_launch(a, ngrid, st);
    return sy_launch(fast ? SYK_PAYLOAD_FAST : SYK_PAYLOAD_GENERAL, a, nj, SY_LDS_BYTES(a.c.M), st);
}
#endif  // SY_PART <= 0

}  // namespace mcrx
