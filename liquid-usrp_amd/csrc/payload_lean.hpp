// payload_lean.hpp -- payload workers for 64-subcarrier frames, one frame per wave, the instruction diet.
// Included by ofdmsync.hip (part 0) inside namespace mcrx.
//
// Replaces ofdmflexframesync's per-symbol path for the payload (reference: liquid-dsp ofdmframesync_execute_rxsymbols ->
// ofdmframesync_rxsymbol -> ofdmflexframesync_rxpayload, called per channel from src/multichannelrx.cc:185-204 through
// the channelizer callback).  Same arithmetic per sample as payload_multi_kernel<1> (same butterflies, twiddles, pilot
// fit, oscillator trim); what changed is where the instructions go.  payload_multi_kernel<1> issues 203 VALU
// instructions per symbol and is VALU-bound (272.8 M per 207.7 M-sample slab = 0.44 of its 0.52 ms, profiles/r3_v2_pmc.csv):
//   * window address: 15 VALU (two 64-bit multiplies per lane)      -> a scalar base that advances per symbol + a lane
//     offset that only changes when the window's phase inside the tiles does (L = 72 with 16-sample tiles: every other symbol);
//   * the six butterfly stages: 50 VALU, of which 24 move data (v_mov_dpp, copies for v_permlane*_swap) -> the partner
//     comes through the LDS crossbar (ds_swizzle / ds_bpermute: no VALU slot, no LDS memory), a stage is one packed fma
//     for the butterfly and two for the twiddle;
//   * pilots: LDS write + fence + read, polarity look-up with a modulo, two 4-step projections -> one ds_bpermute gather,
//     polarity as a sign mask from a table that runs past the sequence's end, both projections in one 3-step reduction
//     when there are at most 8 pilots (bit-identical sums: the dropped step only added zeros);
//   * modem: the demodulator and the stores were compiled for every modem at once and selected per lane -> the symbol loop
//     is instantiated per modem; stores take a scalar base + constant lane offsets.
// MCRX_PAYLOAD_LEAN=0 launches payload_multi_kernel<1> instead (A/B and parity tests).

#ifndef PL_NT_STORE
#define PL_NT_STORE 0       /* measured, not kept: the equalised symbols as non-temporal stores -- workers 0.333 -> 0.354 ms (scratch/r4u.sh) */
#endif
#include "lean_prims.hpp"
#include "demod_pk.hpp"
namespace lean {

// the symbols of one frame.  Everything passed by value is wave-uniform unless it says "lane"
template <int MOD, int XB, int MM = 64>
__device__ __forceinline__ uint32_t symbols(const SyncArgs &a, const PayloadJob *job, const uint32_t ch, const uint32_t j,
                                            const uint32_t *qsg, const uint32_t *qnb, const int *qsrc)
{
    constexpr unsigned bps = bits_per_symbol<MOD>::v;
    const SyncConsts &c = a.c;
    const int l = lane_id();
    const int bp32 = (l ^ 32) << 2;
    // ---- lane constants: after the transform lane l holds subcarrier lane_k<MM>(l) (64: bitrev6(l); 48: lean_prims.hpp; idle lanes: none)
    const int kq = lane_k<MM>(l);
    const bool lane_on = kq >= 0;
    const int kk = lane_on ? kq : 0;
    const int dr = lane_on ? c.data_rank[kk] : -1;
    const bool isdata = dr >= 0;
    Radix3 r3; if constexpr (MM == 48) r3 = radix3_consts(l);
    const float fxr = ((kk > c.M2) ? (float)kk - (float)c.M : (float)kk) * 0.15915494309189535f;
    v2f R = {0.f, 0.f};
    if (lane_on && c.sctype[kk]) { const float2 g = (a.jR + (size_t)j * c.M)[kk]; R.x = g.x; R.y = g.y; }
    v2f tw[6], sgp[3];                                                  // stage twiddles (1 in the lower lanes); butterfly signs, two stages to a pair
#pragma unroll
    for (int st = 0; st < 6; st++) {
        const int h = 32 >> st;
        const bool up = (l & h) != 0;
        const float rev = (float)(l & (h - 1)) * (0.5f / (float)h);
        tw[st].x = up ? __builtin_amdgcn_cosf(rev) : 1.f; tw[st].y = up ? -__builtin_amdgcn_sinf(rev) : 0.f;
        if (st & 1) sgp[st >> 1].y = up ? -1.f : 1.f; else sgp[st >> 1].x = up ? -1.f : 1.f;
    }
    const int Mp = c.M_pilot;
    const bool p8 = Mp <= 8;                                            // both projections in one row: pf0 in lanes 0..7, pf1 in lanes 8..15
    const float pf0 = (l < Mp) ? c.Pfit[l] : 0.f, pf1 = (l < Mp) ? c.Pfit[Mp + l] : 0.f;
    const float pfc = p8 ? ((l < 8) ? pf0 : ((l < 16 && l - 8 < Mp) ? c.Pfit[Mp + l - 8] : 0.f)) : pf0;
    const int psrc = qsrc[l < Mp ? l : 0] << 2;                          // ds_bpermute address of the lane that holds pilot l
    const uint32_t so_sym = (uint32_t)(isdata ? dr : 0) * 8u, so_soft = (uint32_t)(isdata ? dr : 0) * bps;

    // ---- wave-uniform state
    const int L = c.L, cb = c.cp - c.backoff, Md = c.M_data;
    const uint32_t mod_len = rfl(job->s.mod_len), nbits = rfl(8u * job->s.enc_len);
    const uint32_t nsym = (mod_len + (uint32_t)Md - 1u) / (uint32_t)Md;
    const int64_t t_ev0 = job->s.cur + (int64_t)job->s.timer - 1;
    const int64_t ws0 = t_ev0 - L + 1 + cb;
    uint32_t dth = rfl(job->s.nco_dtheta);
    uint32_t th_ws = rfl(job->s.nco_theta_ref + (uint32_t)(ws0 - job->s.nco_t_ref) * job->s.nco_dtheta);
    uint32_t pc4 = rfl(job->s.pilot_count) * 4u;                         // byte offset into the polarity table
    float phi_prime = job->s.phi_prime, p1_prime = job->s.p1_prime;
    int32_t r_ws = (int32_t)rfl((uint32_t)(ws0 - a.buf_first));
    const float2 *chb = a.chan + ((size_t)a.chan_off + ch) * MCRX_TILE_S;
    const uint32_t tstride = a.chan_stride * (uint32_t)MCRX_TILE_S;      // elements between a channel's consecutive tiles
    const int32_t r_max = (int32_t)(a.end - a.buf_first) - 1;
    const bool soft_mode = c.payload_soft != 0;
    const bool keep_syms = a.no_syms == 0;                              // (skip_framesyms = 2: the caller never reads stats.framesyms -- not even stored)
    uint8_t *soft = a.jsoft + (size_t)j * 8 * c.max_enc_len;
    const uint64_t syms_off = job->syms_off;
    uint8_t *syms = a.sarena + (((uint64_t)rfl((uint32_t)(syms_off >> 32)) << 32) | rfl((uint32_t)syms_off));
    const uint32_t l4 = (uint32_t)l * 4u;

    // the window's phase inside the tiles and the lane offset that goes with it: the two most recent ones are kept (L = 72 over
    // 16-sample tiles alternates between two phases), so the symbol loop never recomputes them
    int q0 = -1, q1 = -1; uint32_t offl = 0, offl1 = 0;
    auto load_win = [&](int32_t rw) -> v2f {
        if (rw >= 0 && rw + (WV - 1) <= r_max) {
            const int ph = rw & (MCRX_TILE_S - 1);
            if (ph != q0) {
                const int tq = q0; const uint32_t to = offl; q0 = q1; offl = offl1; q1 = tq; offl1 = to;      // swap: the other cached phase
                if (ph != q0) { q0 = ph; const uint32_t q = (uint32_t)ph + (uint32_t)l; offl = ((q >> MCRX_TILE_SH) * tstride + (q & (uint32_t)(MCRX_TILE_S - 1))) * 8u; }
            }
            const char *base = reinterpret_cast<const char *>(chb + (size_t)(uint32_t)(rw >> MCRX_TILE_SH) * tstride);
            v2f v = *reinterpret_cast<const v2f *>(base + offl);
            if constexpr (MM < WV) { if (l >= MM) { v.x = 0.f; v.y = 0.f; } }      // (a window is MM samples: the lanes behind it hold zeros)
            return v;
        }
        int32_t r = rw + l;
        r = r < 0 ? 0 : (r > r_max ? r_max : r);
        v2f v = *reinterpret_cast<const v2f *>(chb + ((size_t)(r >> MCRX_TILE_SH) * tstride + (size_t)(r & (MCRX_TILE_S - 1))));
        if constexpr (MM < WV) { if (l >= MM) { v.x = 0.f; v.y = 0.f; } }
        return v;
    };

    // Stores run one symbol late.  Loads and stores share one counter (vmcnt) and return out of order against each other, so
    // waiting for a window that was requested before a symbol's stores waits for those stores' acknowledgements as well --
    // several hundred cycles on every symbol's chain.  With the previous symbol's stores issued right after the wait and the
    // next window requested right behind them, everything the next wait covers is a whole symbol old.
    // (only a frame's last symbol can be partly filled -- mod_len = ceil(nbits / bps), so every earlier one ends below nbits --
    //  and the last one is stored behind the loop: the stores inside it take the first branch without asking)
    auto store_symbol = [&](uint32_t ps, v2f Z, uint64_t sw, bool inner) {
        uint8_t *ssym = syms + (size_t)ps * 8, *ssoft = soft + (size_t)ps * bps;
        if (inner || (ps + (uint32_t)Md <= mod_len && (ps + (uint32_t)Md) * bps <= nbits)) {
            if (isdata) {
#if PL_NT_STORE
                if (keep_syms) __builtin_nontemporal_store(Z, reinterpret_cast<v2f *>(ssym + so_sym));
#else
                if (keep_syms) *reinterpret_cast<v2f *>(ssym + so_sym) = Z;
#endif
                uint8_t *dst = ssoft + so_soft;
                if constexpr (bps == 1) dst[0] = (uint8_t)sw;
                else if constexpr (bps == 2) *reinterpret_cast<uint16_t *>(dst) = (uint16_t)sw;
                else {                                                  // (4 or 6 bytes at a multiple of 2)
                    *reinterpret_cast<uint16_t *>(dst) = (uint16_t)sw;
                    *reinterpret_cast<uint16_t *>(dst + 2) = (uint16_t)(sw >> 16);
                    if constexpr (bps == 6) *reinterpret_cast<uint16_t *>(dst + 4) = (uint16_t)(sw >> 32);
                }
            }
        } else if (isdata && ps + (uint32_t)dr < mod_len) {              // the frame's last symbol: part of the subcarriers, part of their bits
            if (keep_syms) *reinterpret_cast<v2f *>(ssym + so_sym) = Z;
            const uint32_t b0 = (ps + (uint32_t)dr) * bps;
#pragma unroll
            for (unsigned kb = 0; kb < bps; kb++) if (b0 + kb < nbits) ssoft[so_soft + kb] = (uint8_t)(sw >> (8 * kb));
        }
    };
    v2f cur = load_win(r_ws);
    v2f Zp = {0.f, 0.f}; uint64_t swp = 0;                               // the previous symbol's results, not stored yet
    uint32_t psi = 0;
    for (uint32_t n = 0; n < nsym; n++) {
        const uint32_t sgn = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(qsg) + (pc4 + (l < 16 ? l4 : 0u)));
        // ---- oscillator (the window's registers are free after it), the previous symbol's stores, the next window
        v2f x = rot_down_pk(cur, u32rev(th_ws + (uint32_t)l * dth));
        if (n + 1 < nsym) cur = load_win(r_ws + L);
        if (n > 0) store_symbol(psi - (uint32_t)Md, Zp, swp, true);
        // ---- MM-point DIF transform, equaliser
        if constexpr (MM == 64) x = fft64<XB>(x, tw, sgp, bp32);
        else x = fft48<XB>(x, tw, sgp, r3, bp32, l);
        x = cmul_pk(x, R);
        // ---- pilots to the first lanes, polarity, phase, unwrap, the two projections of the line fit
        const float xr = x.x, xi = x.y;                                  // (copies: __builtin_bit_cast of a vector element reads element 0)
        float2 P;
        P.x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(psrc, __builtin_bit_cast(int, xr)) ^ (int)sgn);
        P.y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(psrc, __builtin_bit_cast(int, xi)) ^ (int)sgn);
        const float v = atan2_fast(P.y, P.x);
        const float prev = dpp_mov<0x111, false>(v, v);                 // row_shr:1, lane 0 of the row keeps its own
        const float turns = rintf((v - prev) * 0.15915494309189535f);
        float p0, p1;
        if (p8) {
            float t = turns;
            t += dpp_mov<0x111>(t); t += dpp_mov<0x112>(t); t += dpp_mov<0x114>(t);
            const float y = fmaf(-TWO_PI_F, t, v);
            const int yi = __builtin_bit_cast(int, y);
            float z = pfc * __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(yi, yi, 0x118, 0xf, 0xc, false));   // lanes 8..15: y of lanes 0..7
            z += dpp_mov<0x111>(z); z += dpp_mov<0x112>(z); z += dpp_mov<0x114>(z);
            p0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z), 7));
            p1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z), 15));
        } else {
            const float y = fmaf(-TWO_PI_F, row_scan_fast(turns), v);
            p0 = row_total_dpp(pf0 * y);
            p1 = row_total_dpp(pf1 * y);
        }
        pc4 += (uint32_t)Mp * 4u; pc4 = pc4 >= 255u * 4u ? pc4 - 255u * 4u : pc4;
        p1 = 0.3f * p1 + (1.0f - 0.3f) * p1_prime;
        p1_prime = p1;
        // ---- de-rotate, soft bits (stored at the top of the next turn)
        Zp = rot_down_pk(x, fmaf(p1, fxr, p0 * 0.15915494309189535f));
        swp = demod_pk<MOD>(reinterpret_cast<const uint8_t *>(qnb), Zp, soft_mode);
        psi += (uint32_t)Md;
        // ---- oscillator trim (liquid ofdmframesync: the phase at the next window start uses the old step up to this event)
        float dphi = p0 - phi_prime;
        dphi -= TWO_PI_F * rintf(dphi * 0.15915494309189535f);
        phi_prime = p0;
        const uint32_t dnew = dth + rfl((uint32_t)__float2int_rn(dphi * (1e-3f * 683565275.5764316f)));
        th_ws += (uint32_t)(L - cb) * dth + (uint32_t)cb * dnew;
        dth = dnew;
        r_ws += L;
    }
    if (nsym > 0) store_symbol(psi - (uint32_t)Md, Zp, swp, false);
    return dth;
}

}  // namespace lean

// CLS 0: the frames with one or two bits per subcarrier (BPSK, QPSK), a wave per hand-off.  CLS 1: 16- and 64-QAM, a second
// launch that walks place_jobs_kernel's list of them with a grid stride (normally empty; the host sizes the grid from the last
// list it saw).  Two kernels because the QAM demodulator's registers (12 running minima, the neighbour walk) would otherwise
// set the budget of the loop every frame of the periodic benchmark runs in: 56 VGPRs / 70 SGPRs = 8 waves per SIMD
// (amdgpu_num_sgpr: at 86 SGPRs the scalar file admits 7, measured as 8192 waves taking two rounds).
// REST: the frames the main launch's grid (sized from the previous launch's frame count: a.live_off waves, one frame each) did not
// reach -- a small second launch with a grid stride, normally nothing to do.  (The stride loop in the main kernel cost it its eighth
// wave per SIMD: 59 -> 70 registers.)
template <int XB, int CLS, bool REST = false, int MM = 64>
__device__ __forceinline__ void payload_lean_body(SyncArgs &a)
{
    launder(a);
    __shared__ uint32_t qsg[256 + 16];      // pilot polarity as a sign mask; the 255-long sequence continued past its end: no wrap inside a symbol
    __shared__ uint32_t qnb[64];            // the soft demodulator's nearest-neighbour table of this frame's modem
    __shared__ int qsrc[16];                // the lane that holds pilot r after the transform
    const SyncConsts &c = a.c;
    const int l = lane_id();
    const uint32_t nj = a.max_jobs;
    uint32_t nlist = (CLS == 1 && a.qam_list) ? a.qam_list[0] : a.live[0];       // the QAM frames / every frame of the launch
    if (nlist > nj) nlist = nj;
    const uint32_t kfirst = REST ? a.live_off + blockIdx.x : blockIdx.x;
    if (kfirst >= nlist) return;
    for (int k = l; k < 256 + 16; k += WV) qsg[k] = c.pilot_seq[k >= 255 ? k - 255 : k] == 0 ? 0x80000000u : 0u;
    if (l < 16) qsrc[l] = 0;
    wave_sync_lds();
    { const int kq = lean::lane_k<MM>(l); const int pr = kq >= 0 ? c.pilot_rank[kq] : -1; if (pr >= 0 && pr < 16) qsrc[pr] = l; }
    wave_sync_lds();
    for (uint32_t k = kfirst; k < nlist; k += gridDim.x) {
        const uint32_t j = (CLS == 1 && a.qam_list) ? rfl(a.qam_list[1 + k]) : rfl(a.live[1 + k]);
        if (j >= nj) continue;
        const PayloadJob *job = a.jobs + j;
        const uint32_t ch = rfl(job->ch);
        if (ch >= a.nch || job->arena_off == ~0ull) continue;
        const uint32_t mod = rfl(job->s.mod_scheme);
        if (((mod == 39 || mod == 40) ? 0 : 1) != CLS) continue;
        uint32_t dth;
        if constexpr (CLS == 0) {
            if (mod == 39) dth = lean::symbols<39, XB, MM>(a, job, ch, j, qsg, qnb, qsrc);
            else           dth = lean::symbols<40, XB, MM>(a, job, ch, j, qsg, qnb, qsrc);
        } else {
            wave_sync_lds();
            if (mod == 27) { if (l < 16) qnb[l] = reinterpret_cast<const uint32_t *>(c.cod.qam16_nb)[l]; }
            else qnb[l] = reinterpret_cast<const uint32_t *>(c.cod.qam64_nb)[l];
            wave_sync_lds();
            if (mod == 27) dth = lean::symbols<27, XB, MM>(a, job, ch, j, qsg, qnb, qsrc);
            else           dth = lean::symbols<29, XB, MM>(a, job, ch, j, qsg, qnb, qsrc);
        }
        if (l == 0) a.jobs[j].s.nco_dtheta = dth;
        if (CLS == 0 && !REST) break;       // (one frame per wave; what the grid does not cover is the REST launch's)
    }
}
template <int XB, int MM = 64> __global__ __launch_bounds__(WV) __attribute__((amdgpu_num_sgpr(72))) void payload_lean_kernel(SyncArgs a) { payload_lean_body<XB, 0, false, MM>(a); }
// ONE list-driven launch behind the main one for everything it did not take: the BPSK / QPSK frames beyond its grid, then the QAM frames
// (rounds 2-4 had a launch each, nearly always empty -- and an empty kernel still has to find a free wave slot on a full chip before the
// decoder behind it may start: 70-85 us of the work stream's time per push and launch, profiles/r4_s1_kernel_stats.csv)
template <int XB, int MM = 64> __global__ __launch_bounds__(WV) void payload_lean_rest_kernel(SyncArgs a) { payload_lean_body<XB, 0, true, MM>(a); payload_lean_body<XB, 1, false, MM>(a); }
