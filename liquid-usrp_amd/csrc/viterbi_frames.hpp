// r = 1/2, K = 7 soft Viterbi decoder (liquid LIQUID_FEC_CONV_V27: src/multichannel_tx.cc:46-52 lets the user pick it with -c / -k),
// one FRAME per wave, lane = a block of the frame's trellis, all 64 path metrics of the block in the lane's registers.
//
// Why this shape.  Rounds 3-4 had one wave per trellis BLOCK with lane = state: an add-compare-select step was a lane exchange, two
// selects, four adds, a compare, a select and two lane writes for the decision word -- 14 vector and 5 scalar instructions per
// step for 64 states -- followed by a traceback that is a scalar chain of ~12 instructions per step; 1.5 ms per 800 frames of 1200
// bytes (profiles/r5_v27_before_kernel_stats.csv).  With the states in registers the butterfly (states j, j + 32 -> 2j, 2j + 1) has
// compile-time operands and needs no exchange at all: 3 instructions per state and step FOR 64 BLOCKS AT ONCE (16-bit metrics, two
// states to a register: see bfly below), and the traceback is 8 vector instructions per step for 64 blocks at once.
//
// Exactness.  A lane does not know the path metrics at its block's first step, nor the survivor's state at its last.  It starts W
// steps early from equal metrics and runs W steps past its end -- and then both assumptions are CHECKED against the neighbours, and
// repaired where they do not hold, so the result is the full decoder's (oracle/ll_fec.c: 32-bit metrics, ties to the predecessor with
// the older bit 0, traceback from state 0) for any input, not only where survivors merge quickly:
//   forward   lane b keeps its metrics (minimum subtracted) at its first step, C0[b], and at its last, C1[b].  Lane 0 starts from the
//             encoder's state 0, so its metrics are the true ones; if C1[b-1] == C0[b] then lane b's metrics are the true ones (up to
//             a constant) from its first step on, hence all its decisions.  A lane where they differ is run again from C1[b-1]; its C1
//             may change, so the comparison is repeated until every boundary agrees (each pass settles at least the lowest open lane).
//   traceback lane b starts W steps past its end from the best state there (or from state 0 at the trellis' end) and notes the state
//             S1[b] it passes at its block's last step and the state E[b] it reaches at its first.  The last lane's path is the true
//             survivor; if S1[b] == E[b+1] then lane b's path inside its block is the true one.  A lane where they differ traces its
//             block again from E[b+1]; repeated until every boundary agrees.
// On a decodable signal the repairs are rare (all survivors merge within 48 steps at 3 dB per coded bit; W = 48; at 0 dB one boundary in a
// hundred has not, and the frame fails its check anyway); on
// noise -- where 7 % of the block boundaries of the old kernel's 192-step overlap had not merged, i.e. the old kernel was not exact --
// a frame takes a few extra passes.  (Merge-depth statistics: profiles/r5_viterbi_merge_depth.txt.)
//
#pragma once
// (included inside namespace mcrx, after lane_id(): ofdmsync.hip; needs <utility>)
namespace vf {

// (W, block_steps(T), rows_for(T): kernels.h -- the host sizes the scratch with them)
static_assert(W % 24u == 0u && W >= 48u, "warm-up: whole register-layout periods, whole bytes, and the traceback's prefetch reaches back 48 rows");

__device__ __forceinline__ constexpr unsigned par(unsigned v) { unsigned p = 0; while (v) { p ^= v & 1u; v >>= 1; } return p; }
// expected outputs of the transition (predecessor j, input 0): index into the step's four branch metrics (first output << 1 | second)
__device__ __forceinline__ constexpr unsigned bm_idx(unsigned j) { return (par((2u * j) & 0x6du) << 1) | par((2u * j) & 0x4fu); }

// Path metrics: 16 bits, two states to a register -- register k holds states 2k (low half) and 2k + 1 (high half).  The butterfly
// (states j, j + 32 -> 2j, 2j + 1) fills ONE new register from one half each of two old ones, and the packed instructions' operand
// selects broadcast that half for free: with c = the outputs the transition j -> 2j expects and c' their complement (both generators
// tap the oldest and the newest bit, so j -> 2j + 1 and j + 32 -> 2j expect c', j + 32 -> 2j + 1 expects c),
//     m0 = (old[j],      old[j])      + (BM[c],  BM[c'])      m1 = (old[j + 32], old[j + 32]) + (BM[c'], BM[c])
//     new = min(m0, m1)    decisions = the sign bits of m1 - m0 (16-bit wrap-around: differences stay below 6 x 510 + 510)
// = 6 instructions for two states (adds, minimum, difference, and the two that shift the sign bits into the step's decision word),
// where 32-bit metrics with one state to a register took 10.  A register file of 32 old + 32 new, no layout that rotates.  Metrics
// grow by at most 510 a step and never differ by more than 6 x 510, so the minimum is subtracted every 96 steps.
// Decision words of a step: state n = 2j + h -> word j >> 4, bit (j & 15) + 16 h.
template <unsigned J>
__device__ __forceinline__ void bfly(const unsigned (&P)[32], unsigned (&Q)[32], const unsigned (&PB)[4], unsigned &acc, unsigned k80008000)
{
    constexpr unsigned lo = J >> 1, hi = 16u + (J >> 1), ix = bm_idx(J);
    unsigned m0, m1, t;
#define VF_BFLY(SEL)                                                                                                       \
    asm("v_pk_add_u16 %1, %5, %7 " SEL "\n\t"                                                                               \
        "v_pk_add_u16 %2, %6, %8 " SEL "\n\t"                                                                               \
        "v_pk_min_u16 %0, %1, %2\n\t"                                                                                       \
        "v_pk_sub_u16 %3, %2, %1\n\t"                                                                                       \
        "v_pk_lshrrev_b16 %4, 1, %4 op_sel_hi:[0,1]\n\t"                                                                    \
        "v_and_or_b32 %4, %3, %9, %4"                                                                                        \
        : "=v"(Q[J]), "=&v"(m0), "=&v"(m1), "=&v"(t), "+v"(acc)                                                             \
        : "v"(P[lo]), "v"(P[hi]), "v"(PB[ix]), "v"(PB[3u - ix]), "s"(k80008000))
    if constexpr ((J & 1u) == 0u) VF_BFLY("op_sel:[0,0] op_sel_hi:[0,1]");
    else                          VF_BFLY("op_sel:[1,0] op_sel_hi:[1,1]");
#undef VF_BFLY
}
template <unsigned... J>
__device__ __forceinline__ void half_step(const unsigned (&P)[32], unsigned (&Q)[32], const unsigned (&PB)[4], unsigned &acc, unsigned k, std::integer_sequence<unsigned, J...>)
{
    (bfly<J>(P, Q, PB, acc, k), ...);
}
template <unsigned... J>
__device__ __forceinline__ void half_step_hi(const unsigned (&P)[32], unsigned (&Q)[32], const unsigned (&PB)[4], unsigned &acc, unsigned k, std::integer_sequence<unsigned, J...>)
{
    (bfly<16u + J>(P, Q, PB, acc, k), ...);
}
// one step, P -> Q: sy = first soft symbol | second << 8 (low 16 bits); w0 / w1: the decision words
__device__ __forceinline__ void step(const unsigned (&P)[32], unsigned (&Q)[32], unsigned sy, unsigned &w0, unsigned &w1, unsigned k80008000)
{
    const unsigned sa = sy & 255u, sb = (sy >> 8) & 255u;
    unsigned BM[4], PB[4];
    BM[0] = sa + sb; BM[1] = sa + 255u - sb; BM[2] = 255u - sa + sb; BM[3] = 510u - sa - sb;
#pragma unroll
    for (int i = 0; i < 4; i++) PB[i] = BM[i] | (BM[3 - i] << 16);
    w0 = 0u; w1 = 0u;
    half_step(P, Q, PB, w0, k80008000, std::make_integer_sequence<unsigned, 16>{});
    half_step_hi(P, Q, PB, w1, k80008000, std::make_integer_sequence<unsigned, 16>{});
}
// the minimum of the 64 metrics in both halves
__device__ __forceinline__ unsigned min_all(const unsigned (&P)[32])
{
    unsigned m = P[0];
#pragma unroll
    for (int k = 1; k < 32; k++) asm("v_pk_min_u16 %0, %0, %1" : "+v"(m) : "v"(P[k]));
    asm("v_pk_min_u16 %0, %0, %0 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(m));
    return m;
}
__device__ __forceinline__ void normalise(unsigned (&P)[32])
{
    const unsigned m = min_all(P);
#pragma unroll
    for (int k = 0; k < 32; k++) asm("v_pk_sub_u16 %0, %0, %1" : "+v"(P[k]) : "v"(m));
}

struct Frame {
    const uint8_t *soft;        // 2 T soft symbols (8-byte aligned)
    uint8_t *dec;               // n decoded bytes
    unsigned n, T, B, nblk;     // T = 8 n + 6 steps, B steps per lane, nblk lanes in use
    uint2 *rows;                // this wave's decision rows [B + 2 W][64]
    uint16_t *ck;               // LDS [2][64][64]: C0 / C1 [state][lane]
};

// soft symbols of steps t .. t + 5 (t even, per lane; steps outside [0, T) read something inside the frame)
__device__ __forceinline__ void load_syms(const Frame &f, int t, unsigned (&d)[3])
{
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int tt = t + 2 * k;
        tt = tt < 0 ? 0 : tt;
        tt = tt > (int)f.T - 2 ? (int)f.T - 2 : tt;
        d[k] = *reinterpret_cast<const unsigned *>(f.soft + 2 * (size_t)tt);
    }
}

// metrics with the minimum subtracted -> ck[which][state][lane]
__device__ __forceinline__ void checkpoint(const Frame &f, const unsigned (&P)[32], unsigned which, unsigned lane)
{
    const unsigned m = min_all(P);
#pragma unroll
    for (int k = 0; k < 32; k++) {
        unsigned v;
        asm("v_pk_sub_u16 %0, %1, %2" : "=v"(v) : "v"(P[k]), "v"(m));
        f.ck[(which * 64u + 2u * (unsigned)k) * 64u + lane] = (uint16_t)v;
        f.ck[(which * 64u + 2u * (unsigned)k + 1u) * 64u + lane] = (uint16_t)(v >> 16);
    }
}

// One forward pass of the wave.  first: every lane from equal metrics, W steps before its block (lane 0: the encoder's state 0 at its
// first step); otherwise the lanes in `commit` start at their block's first step from the metrics their predecessor ended with.
// Lanes outside `commit` run along and leave nothing behind.  Returns the best state at the lane's last step.
__device__ __forceinline__ unsigned forward(const Frame &f, bool first, bool commit, unsigned lane, int tw)
{
    unsigned P[32], Q[32];
    const unsigned rowsN = f.B + 2u * W, i0 = first ? 0u : W;
    const unsigned k8 = 0x80008000u;
    if (first) {
#pragma unroll
        for (int k = 0; k < 32; k++) P[k] = 0u;
    } else {
        const unsigned pl = lane ? lane - 1u : 0u;
#pragma unroll
        for (int k = 0; k < 32; k++) P[k] = (unsigned)f.ck[(64u + 2u * (unsigned)k) * 64u + pl] | ((unsigned)f.ck[(64u + 2u * (unsigned)k + 1u) * 64u + pl] << 16);
        if (commit) {
#pragma unroll
            for (int k = 0; k < 32; k++) {                  // C0 := what this run starts from
                f.ck[(2u * (unsigned)k) * 64u + lane] = (uint16_t)P[k];
                f.ck[(2u * (unsigned)k + 1u) * 64u + lane] = (uint16_t)(P[k] >> 16);
            }
        }
    }
    unsigned d[3], dn[3];
    load_syms(f, tw + (int)i0, d);
    unsigned since = 0u;                                    // steps since the minimum was last subtracted
    for (unsigned i = i0; i < rowsN; i += 6u) {
        load_syms(f, tw + (int)i + 6, dn);
        if (first && i == W) {
            // the encoder starts in state 0: every other state far enough behind never to win against a path from state 0 (6 x 510 would
            // do; oracle/ll_fec.c has 1 << 20), near enough not to overflow before the next subtraction
            if (lane == 0u) {
                P[0] = 0x2000u << 16;
#pragma unroll
                for (int k = 1; k < 32; k++) P[k] = 0x20002000u;
            }
            if (commit) checkpoint(f, P, 0u, lane);
        }
        if (i == W + f.B && commit) checkpoint(f, P, 1u, lane);
        if (since >= 90u) { normalise(P); since = 0u; }
        since += 6u;
        unsigned w[12];
        step(P, Q, d[0], w[0], w[1], k8);
        step(Q, P, d[0] >> 16, w[2], w[3], k8);
        step(P, Q, d[1], w[4], w[5], k8);
        step(Q, P, d[1] >> 16, w[6], w[7], k8);
        step(P, Q, d[2], w[8], w[9], k8);
        step(Q, P, d[2] >> 16, w[10], w[11], k8);
        if (commit) {
            uint2 *row = f.rows + (size_t)i * 64u + lane;
#pragma unroll
            for (int k = 0; k < 6; k++) row[(size_t)k * 64u] = make_uint2(w[2 * k], w[2 * k + 1]);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) d[k] = dn[k];
    }
    unsigned best = 0u, bm = P[0] & 0xffffu;
#pragma unroll
    for (int s = 1; s < 64; s++) {
        const unsigned v = (s & 1) ? (P[s >> 1] >> 16) : (P[s >> 1] & 0xffffu);
        const bool lt = v < bm; bm = lt ? v : bm; best = lt ? (unsigned)s : best;
    }
    return best;
}

// One traceback pass: rows hi - 1 .. W (hi a multiple of 24), starting in state n at row iend - 1 (rows at or above iend are skipped).
// Lanes in `commit` store their block's bytes.  Returns the state at the block's first step | the state passed at its last << 8.
__device__ __forceinline__ unsigned traceback(const Frame &f, unsigned hi, unsigned n, unsigned iend, bool commit, unsigned lane, unsigned t0)
{
    unsigned s1 = 0u;
    const uint2 *rows = f.rows + lane;
    uint2 cur[24], nxt[24];
#pragma unroll
    for (int r = 0; r < 24; r++) cur[r] = rows[(size_t)(hi - 24u + (unsigned)r) * 64u];
    for (unsigned k1 = hi; k1 > W; k1 -= 24u) {
#pragma unroll
        for (int r = 0; r < 24; r++) nxt[r] = rows[(size_t)(k1 - 48u + (unsigned)r) * 64u];       // (k1 >= W + 24 and W >= 48: rows that exist; the last batch's are not used)
        if (k1 == W + f.B) s1 = n;
        unsigned bytes = 0u;                                // rows k1 - 24 .. k1 - 1: three bytes, the lowest row's in bits 0..7
#pragma unroll
        for (int r = 23; r >= 0; r--) {
            const unsigned k = k1 - 24u + (unsigned)r;
            const unsigned up = 0u - (n >> 5), wsel = (cur[r].x & ~up) | (cur[r].y & up);         // (a select of the two would make the rows an indexed array: scratch)
            const unsigned h = n >> 1;
            const unsigned dbit = (wsel >> ((h & 15u) + ((n & 1u) << 4))) & 1u;                  // (state 2j + h: word j >> 4, bit (j & 15) + 16 h)
            bytes |= (n & 1u) << ((unsigned)(r >> 3) * 8u + 7u - (unsigned)(r & 7));       // (step t -> byte t / 8, bit 7 - t % 8)
            const unsigned nn = h | (dbit << 5);
            n = k < iend ? nn : n;
        }
        if (commit && k1 <= W + f.B) {
            const unsigned by = (t0 + (k1 - 24u - W)) >> 3;
#pragma unroll
            for (int q = 0; q < 3; q++) if (by + (unsigned)q < f.n) f.dec[by + (unsigned)q] = (uint8_t)(bytes >> (8 * q));
        }
#pragma unroll
        for (int r = 0; r < 24; r++) cur[r] = nxt[r];
    }
    return n | (s1 << 8);
}

// decode one frame with the calling wave: n bytes from 2 (8 n + 6) soft symbols
__device__ __forceinline__ void decode_frame(const uint8_t *soft, unsigned n_, uint8_t *dec, uint2 *rows, uint16_t *ck, uint32_t *passes)
{
    const unsigned lane = (unsigned)lane_id();
    Frame f;
    f.soft = soft; f.dec = dec; f.rows = rows; f.ck = ck;
    f.n = (unsigned)__builtin_amdgcn_readfirstlane((int)n_);
    f.T = 8u * f.n + 6u; f.B = block_steps(f.T); f.nblk = (f.T + f.B - 1u) / f.B;
    const unsigned t0 = lane * f.B;
    const int tw = (int)t0 - (int)W;
    const bool mine = lane < f.nblk;
    // forward passes
#ifdef VF_PROF
    const unsigned long long vf_t0 = wall_clock64();
#endif
    unsigned guess = 0u, np = 0u;
    {
        bool first = true, redo = mine;
        for (;;) {
            const unsigned g = forward(f, first, redo, lane, tw);
            if (redo) guess = g;
            first = false; np++;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            bool same = true;
            if (mine && lane) {
#pragma unroll 8
                for (unsigned s = 0; s < 64u; s++) same = same && f.ck[(64u + s) * 64u + lane - 1u] == f.ck[s * 64u + lane];
            }
            redo = mine && !same;
            if (!__ballot(redo)) break;
        }
    }
#ifdef VF_PROF
    const unsigned long long vf_t1 = wall_clock64();
#endif
    // traceback passes
    const unsigned rowsN = f.B + 2u * W;
    const bool exact_start = t0 + f.B + W >= f.T;             // the look-ahead reaches the trellis' end: state 0 there
    unsigned nfwd = np;
    {
        unsigned hi = rowsN, n0 = exact_start ? 0u : guess, iend = exact_start ? (unsigned)((int)f.T - tw) : rowsN;
        bool redo = mine;
        unsigned s1 = 0u, e = 0u;
        for (;;) {
            const unsigned r = traceback(f, hi, n0, iend, redo, lane, t0);
            if (redo) { e = r & 63u; s1 = r >> 8; }
            np++;
            const unsigned above = (unsigned)__shfl_down((int)e, 1, WV);
            redo = mine && !exact_start && lane + 1u < f.nblk && s1 != above;
            if (!__ballot(redo)) break;
            hi = W + f.B; n0 = above; iend = W + f.B;
        }
    }
    if (passes && lane == 0u) {
#ifdef VF_PROF
        atomicMax(passes + 6, ~(unsigned)vf_t0); atomicMax(passes + 7, (unsigned)vf_t0);
        atomicAdd(passes + 4, (unsigned)(vf_t1 - vf_t0)); atomicAdd(passes + 5, (unsigned)(wall_clock64() - vf_t1));     // 100 MHz ticks
#endif
        atomicAdd(passes + 2, 1u);
        if (np > 2u) { atomicAdd(passes, nfwd - 1u); atomicAdd(passes + 1, np - nfwd - 1u); }
    }
}

}  // namespace vf
