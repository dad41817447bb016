// r = 1/2, K = 7 soft Viterbi decoder (liquid LIQUID_FEC_CONV_V27: src/multichannel_tx.cc:46-52 lets the user pick it with -c / -k),
// one FRAME per wave, lane = a block of the frame's trellis, all 64 path metrics of the block in the lane's registers.
//
// Why this shape.  Rounds 3-4 had one wave per trellis BLOCK with lane = state: an add-compare-select step was a lane exchange, two
// selects, four adds, a compare, a select and two lane writes for the decision word -- 14 vector and 5 scalar instructions per
// step for 64 states -- followed by a traceback that is a scalar chain of ~12 instructions per step; 1.5 ms per 800 frames of 1200
// bytes (profiles/r5_v27_before_kernel_stats.csv).  With the states in registers the butterfly (states j, j + 32 -> 2j, 2j + 1) has
// compile-time operands, needs no exchange at all, and its decision bits shift into two accumulators through the carry flag: 10
// instructions per butterfly = 5 per state and step FOR 64 BLOCKS AT ONCE, and the traceback is 8 vector instructions per step for
// 64 blocks at once.
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
// Register layout of the forward pass: after i steps register r holds state rotl6(r, i mod 6) -- a new state 2j (2j + 1) goes where
// its predecessor j (j + 32) was, so the butterfly works in place -- and everything that looks at the registers (start, checkpoints,
// best state) does so at multiples of 6 steps, where the layout is the identity.  Decision words of a step: state n -> word n & 1,
// bit 31 - (n >> 1).
#pragma once
// (included inside namespace mcrx, after lane_id(): ofdmsync.hip; needs <utility>)
namespace vf {

// (W, block_steps(T), rows_for(T): kernels.h -- the host sizes the scratch with them)
static_assert(W % 24u == 0u && W >= 48u, "warm-up: whole register-layout periods, whole bytes, and the traceback's prefetch reaches back 48 rows");

__device__ __forceinline__ constexpr unsigned rr6(unsigned v, unsigned r) { return r == 0 ? v : (((v >> r) | (v << (6u - r))) & 63u); }
__device__ __forceinline__ constexpr unsigned par(unsigned v) { unsigned p = 0; while (v) { p ^= v & 1u; v >>= 1; } return p; }
// expected outputs of the transition (predecessor j, input 0): index into the step's four branch metrics (first output << 1 | second)
__device__ __forceinline__ constexpr unsigned bm_idx(unsigned j) { return (par((2u * j) & 0x6du) << 1) | par((2u * j) & 0x4fu); }

// butterfly j of a step whose index is PH mod 6: the transitions j -> 2j and j + 32 -> 2j + 1 expect the outputs c, the other two
// the complement (both generators tap the oldest and the newest bit)
template <unsigned PH, unsigned J>
__device__ __forceinline__ void bfly(unsigned (&R)[64], const unsigned (&BM)[4], unsigned &w0, unsigned &w1)
{
    constexpr unsigned ra = rr6(J, PH), rb = rr6(J + 32u, PH), ix = bm_idx(J);
    unsigned t0, t1, t2, t3;
    asm("v_add_u32_e32 %4, %0, %8\n\t"
        "v_add_u32_e32 %5, %1, %9\n\t"
        "v_add_u32_e32 %6, %0, %9\n\t"
        "v_add_u32_e32 %7, %1, %8\n\t"
        "v_cmp_lt_u32_e32 vcc, %5, %4\n\t"
        "v_cndmask_b32_e32 %0, %4, %5, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, %2, %2, vcc\n\t"
        "v_cmp_lt_u32_e32 vcc, %7, %6\n\t"
        "v_cndmask_b32_e32 %1, %6, %7, vcc\n\t"
        "v_addc_co_u32_e32 %3, vcc, %3, %3, vcc"
        : "+v"(R[ra]), "+v"(R[rb]), "+v"(w0), "+v"(w1), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(BM[ix]), "v"(BM[3u - ix])
        : "vcc");
}
// (Two butterflies interleaved, their compares in scalar register pairs of their own so that four are in flight, change nothing: a
//  wave alone on its SIMD issues a vector instruction every ~4.8 cycles either way, and waves that share a SIMD share that rate.)
template <unsigned PH, unsigned... J>
__device__ __forceinline__ void step_impl(unsigned (&R)[64], const unsigned (&BM)[4], unsigned &w0, unsigned &w1, std::integer_sequence<unsigned, J...>)
{
    (bfly<PH, J>(R, BM, w0, w1), ...);
}
// one step: sy = first soft symbol | second << 8 (low 16 bits)
template <unsigned PH>
__device__ __forceinline__ void step(unsigned (&R)[64], unsigned sy, unsigned &w0, unsigned &w1)
{
    const unsigned sa = sy & 255u, sb = (sy >> 8) & 255u;
    unsigned BM[4];
    BM[0] = sa + sb; BM[1] = sa + 255u - sb; BM[2] = 255u - sa + sb; BM[3] = 510u - sa - sb;
    step_impl<PH>(R, BM, w0, w1, std::make_integer_sequence<unsigned, 32>{});
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
__device__ __forceinline__ void checkpoint(const Frame &f, const unsigned (&R)[64], unsigned which, unsigned lane)
{
    unsigned mn = R[0];
#pragma unroll
    for (int s = 1; s < 64; s++) mn = R[s] < mn ? R[s] : mn;
#pragma unroll
    for (int s = 0; s < 64; s++) {
        const unsigned v = R[s] - mn;
        f.ck[(which * 64u + (unsigned)s) * 64u + lane] = (uint16_t)(v > 65535u ? 65535u : v);
    }
}

// One forward pass of the wave.  first: every lane from equal metrics, W steps before its block (lane 0: the encoder's state 0 at its
// first step); otherwise the lanes in `commit` start at their block's first step from the metrics their predecessor ended with.
// Lanes outside `commit` run along and leave nothing behind.  Returns the best state at the lane's last step.
__device__ __forceinline__ unsigned forward(const Frame &f, bool first, bool commit, unsigned lane, int tw)
{
    unsigned R[64];
    const unsigned rowsN = f.B + 2u * W, i0 = first ? 0u : W;
    if (first) {
#pragma unroll
        for (int s = 0; s < 64; s++) R[s] = 0u;
    } else {
        const unsigned pl = lane ? lane - 1u : 0u;
#pragma unroll
        for (int s = 0; s < 64; s++) R[s] = f.ck[(64u + (unsigned)s) * 64u + pl];
        if (commit) {
#pragma unroll
            for (int s = 0; s < 64; s++) f.ck[(unsigned)s * 64u + lane] = (uint16_t)R[s];        // C0 := what this run starts from
        }
    }
    unsigned d[3], dn[3];
    load_syms(f, tw + (int)i0, d);
    for (unsigned i = i0; i < rowsN; i += 6u) {
        load_syms(f, tw + (int)i + 6, dn);
        if (first && i == W) {
            if (lane == 0u) {
#pragma unroll
                for (int s = 0; s < 64; s++) R[s] = s ? (1u << 20) : 0u;
            }
            if (commit) checkpoint(f, R, 0u, lane);
        }
        if (i == W + f.B && commit) checkpoint(f, R, 1u, lane);
        unsigned w[12];
#pragma unroll
        for (int k = 0; k < 12; k++) w[k] = 0u;
        step<0>(R, d[0], w[0], w[1]);
        step<1>(R, d[0] >> 16, w[2], w[3]);
        step<2>(R, d[1], w[4], w[5]);
        step<3>(R, d[1] >> 16, w[6], w[7]);
        step<4>(R, d[2], w[8], w[9]);
        step<5>(R, d[2] >> 16, w[10], w[11]);
        if (commit) {
            uint2 *row = f.rows + (size_t)i * 64u + lane;
#pragma unroll
            for (int k = 0; k < 6; k++) row[(size_t)k * 64u] = make_uint2(w[2 * k], w[2 * k + 1]);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) d[k] = dn[k];
    }
    unsigned best = 0u, bm = R[0];
#pragma unroll
    for (int s = 1; s < 64; s++) { const bool lt = R[s] < bm; bm = lt ? R[s] : bm; best = lt ? (unsigned)s : best; }
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
            const unsigned odd = 0u - (n & 1u), wsel = (cur[r].x & ~odd) | (cur[r].y & odd);      // (a select of the two would make the rows an indexed array: scratch)
            const unsigned h = n >> 1;
            const unsigned dbit = (wsel << h) >> 31;
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
