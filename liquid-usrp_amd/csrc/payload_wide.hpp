// payload_wide.hpp -- payload workers for 128- and 256-subcarrier frames (E = M / 64 = 2, 4 samples per lane), one frame per wave:
// payload_lean.hpp's instruction diet carried over to the wider symbols (round 6).  Included by ofdmsync.hip (part 1) inside namespace mcrx.
//
// Replaces, for these widths, the width-generic worker payload_kernel<E, true> (Walker<E>::run_job_fast), whose symbol loop is
// ~3 000 static instructions with the modem's loops inside (every modem compiled at once and selected at run time, byte-wise soft-bit
// stores, 64-bit window addresses per element): 8.6 us per 256-subcarrier symbol for a wave that has a SIMD almost to itself --
// BASELINE.json configs[2] (64 channels, M = 256, 16-QAM) spent 0.26 of its 0.93 ms per push there, at two waves per SIMD.
// Same arithmetic per sample (reference: liquid-dsp ofdmframesync_execute_rxsymbols -> ofdmframesync_rxsymbol ->
// ofdmflexframesync_rxpayload, per channel from src/multichannelrx.cc:185-204): oscillator on the raw window, log2(E) radix-2 DIF stages
// inside the lane (sample i = l + 64 e sits in element e of lane l), the 64-point transform of lean_prims.hpp on every element (lane
// exchanges through the LDS crossbar), equaliser, pilots gathered into the first lanes with ds_bpermute (no LDS memory, no fence),
// polarity as a sign mask, unwrap as a prefix sum of turns, the two projections of the line fit, slope smoothing, de-rotation, the
// modem instantiated per scheme, soft bits stored a word at a time and one symbol late, oscillator trim in scalars.
// Element e of lane l ends the transform with subcarrier bitrev_{log2 M}(l + 64 e).
#include "lean_prims.hpp"
#include "demod_pk.hpp"
namespace lean {

template <int MOD, int XB, int E>
__device__ __forceinline__ uint32_t symbols_wide(const SyncArgs &a, const PayloadJob *job, const uint32_t ch, const uint32_t j,
                                                 const uint32_t *qsg, const uint32_t *qnb, const int *qsrc)
{
    constexpr unsigned bps = MOD == 39 ? 1u : MOD == 40 ? 2u : MOD == 27 ? 4u : 6u;
    constexpr int LOG2E = E == 2 ? 1 : (E == 4 ? 2 : 3);
    static_assert(E == 2 || E == 4 || E == 8, "symbol widths 128, 256, 512");
    const SyncConsts &c = a.c;
    const int l = lane_id();
    const int bp32 = (l ^ 32) << 2;
    // ---- lane constants
    int dr[E]; float fxr[E]; v2f R[E]; uint32_t so_sym[E], so_soft[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int kk = (int)(__brev((unsigned)(l + WV * e)) >> (32 - 6 - LOG2E));
        dr[e] = c.data_rank[kk];
        fxr[e] = ((kk > c.M2) ? (float)kk - (float)c.M : (float)kk) * 0.15915494309189535f;
        R[e].x = 0.f; R[e].y = 0.f;
        if (c.sctype[kk]) { const float2 g = (a.jR + (size_t)j * c.M)[kk]; R[e].x = g.x; R[e].y = g.y; }
        so_sym[e] = (uint32_t)(dr[e] >= 0 ? dr[e] : 0) * 8u; so_soft[e] = (uint32_t)(dr[e] >= 0 ? dr[e] : 0) * bps;
    }
    v2f tw[6], sgp[3];                                                  // the 64-point transform's stage twiddles and butterfly signs (payload_lean.hpp)
#pragma unroll
    for (int st = 0; st < 6; st++) {
        const int h = 32 >> st;
        const bool up = (l & h) != 0;
        const float rev = (float)(l & (h - 1)) * (0.5f / (float)h);
        tw[st].x = up ? __builtin_amdgcn_cosf(rev) : 1.f; tw[st].y = up ? -__builtin_amdgcn_sinf(rev) : 0.f;
        if (st & 1) sgp[st >> 1].y = up ? -1.f : 1.f; else sgp[st >> 1].x = up ? -1.f : 1.f;
    }
    // in-lane stages: span 64 J pairs elements e and e + J ((e & J) == 0), twiddle e^{-j 2 pi ((l + 64 e) mod 64 J) / (128 J)}: (l + 64 e) mod 64 J
    // = l + 64 (e mod J), so a stage has J different twiddles -- itw[J - 1 + (e mod J)] = (cos, sin)
    v2f itw[E - 1];
#pragma unroll
    for (int J = E / 2; J >= 1; J >>= 1)
#pragma unroll
        for (int r = 0; r < J; r++) {
            const float rev = (float)(l + WV * r) * (0.5f / (float)(WV * J));
            itw[J - 1 + r].x = __builtin_amdgcn_cosf(rev); itw[J - 1 + r].y = __builtin_amdgcn_sinf(rev);
        }
    const int Mp = c.M_pilot;
    const float pf0 = (l < Mp) ? c.Pfit[l] : 0.f, pf1 = (l < Mp) ? c.Pfit[Mp + l] : 0.f;
    const int psrc = qsrc[l < Mp ? l : 0];                               // (element << 8) | lane that holds pilot l after the transform
    const int paddr = (psrc & 63) << 2, pel = psrc >> 8;

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
    const bool keep_syms = a.no_syms == 0;
    uint8_t *soft = a.jsoft + (size_t)j * 8 * c.max_enc_len;
    const uint64_t syms_off = job->syms_off;
    uint8_t *syms = a.sarena + (((uint64_t)rfl((uint32_t)(syms_off >> 32)) << 32) | rfl((uint32_t)syms_off));
    const uint32_t l4 = (uint32_t)l * 4u;

    // window: sample rw + l + 64 e.  64 e is a whole number of tiles, so element e's address is element 0's plus a wave-uniform step;
    // the lane offset depends on the window's phase inside the tiles only (L = M + cp is a multiple of 16 in every configuration
    // the reference's applications use: it is computed once)
    int q0 = -1; uint32_t offl = 0;
    const uint32_t estep = (uint32_t)(WV >> MCRX_TILE_SH) * tstride * 8u;
    auto load_win = [&](int32_t rw, v2f (&w)[E]) {
        if (rw >= 0 && rw + (WV * E - 1) <= r_max) {
            const int ph = rw & (MCRX_TILE_S - 1);
            if (ph != q0) { q0 = ph; const uint32_t q = (uint32_t)ph + (uint32_t)l; offl = ((q >> MCRX_TILE_SH) * tstride + (q & (uint32_t)(MCRX_TILE_S - 1))) * 8u; }
            const char *base = reinterpret_cast<const char *>(chb + (size_t)(uint32_t)(rw >> MCRX_TILE_SH) * tstride) + offl;
#pragma unroll
            for (int e = 0; e < E; e++) w[e] = *reinterpret_cast<const v2f *>(base + (size_t)e * estep);
            return;
        }
#pragma unroll
        for (int e = 0; e < E; e++) {
            int32_t r = rw + l + WV * e;
            r = r < 0 ? 0 : (r > r_max ? r_max : r);
            w[e] = *reinterpret_cast<const v2f *>(chb + ((size_t)(r >> MCRX_TILE_SH) * tstride + (size_t)(r & (MCRX_TILE_S - 1))));
        }
    };
    // stores run one symbol late (payload_lean.hpp: loads and stores share vmcnt)
    auto store_symbol = [&](uint32_t ps, const v2f (&Z)[E], const uint64_t (&sw)[E], bool inner) {
        uint8_t *ssym = syms + (size_t)ps * 8, *ssoft = soft + (size_t)ps * bps;
        const bool whole = inner || (ps + (uint32_t)Md <= mod_len && (ps + (uint32_t)Md) * bps <= nbits);
#pragma unroll
        for (int e = 0; e < E; e++) {
            if (dr[e] < 0) continue;
            if (whole) {
                if (keep_syms) *reinterpret_cast<v2f *>(ssym + so_sym[e]) = Z[e];
                uint8_t *dst = ssoft + so_soft[e];
                if constexpr (bps == 1) dst[0] = (uint8_t)sw[e];
                else if constexpr (bps == 2) *reinterpret_cast<uint16_t *>(dst) = (uint16_t)sw[e];
                else if constexpr (bps == 4) *reinterpret_cast<uint32_t *>(dst) = (uint32_t)sw[e];          // (symbol index x 4 bytes: aligned)
                else {
                    *reinterpret_cast<uint16_t *>(dst) = (uint16_t)sw[e];
                    *reinterpret_cast<uint16_t *>(dst + 2) = (uint16_t)(sw[e] >> 16);
                    *reinterpret_cast<uint16_t *>(dst + 4) = (uint16_t)(sw[e] >> 32);
                }
            } else if (ps + (uint32_t)dr[e] < mod_len) {                   // the frame's last symbol: part of the subcarriers, part of their bits
                if (keep_syms) *reinterpret_cast<v2f *>(ssym + so_sym[e]) = Z[e];
                const uint32_t b0 = (ps + (uint32_t)dr[e]) * bps;
#pragma unroll
                for (unsigned kb = 0; kb < bps; kb++) if (b0 + kb < nbits) ssoft[so_soft[e] + kb] = (uint8_t)(sw[e] >> (8 * kb));
            }
        }
    };
    v2f cur[E];
    load_win(r_ws, cur);
    v2f Zp[E]; uint64_t swp[E];
#pragma unroll
    for (int e = 0; e < E; e++) { Zp[e].x = 0.f; Zp[e].y = 0.f; swp[e] = 0; }
    uint32_t psi = 0;
    for (uint32_t n = 0; n < nsym; n++) {
        const uint32_t sgn = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(qsg) + (pc4 + (l < Mp ? l4 : 0u)));
        // ---- oscillator, the previous symbol's stores, the next window
        v2f x[E];
#pragma unroll
        for (int e = 0; e < E; e++) x[e] = rot_down_pk(cur[e], u32rev(th_ws + (uint32_t)(l + WV * e) * dth));
        if (n + 1 < nsym) load_win(r_ws + L, cur);
        if (n > 0) store_symbol(psi - (uint32_t)Md, Zp, swp, true);
        // ---- M-point DIF transform: the in-lane stages, then 64 points per element; equaliser
#pragma unroll
        for (int J = E / 2; J >= 1; J >>= 1)
#pragma unroll
            for (int e = 0; e < E; e++)
                if ((e & J) == 0) {
                    const v2f u = x[e], v = x[e + J];
                    x[e] = u + v;
                    x[e + J] = cmulc_pk(u - v, itw[J - 1 + (e & (J - 1))]);
                }
#pragma unroll
        for (int e = 0; e < E; e++) { x[e] = fft64<XB>(x[e], tw, sgp, bp32); x[e] = cmul_pk(x[e], R[e]); }
        // ---- pilots to the first lanes (every lane asks by address, outside any lane condition: a masked source lane reads as zero)
        float px = 0.f, py = 0.f;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const float xr = x[e].x, xi = x[e].y;
            const int gx = __builtin_amdgcn_ds_bpermute(paddr, __builtin_bit_cast(int, xr)), gy = __builtin_amdgcn_ds_bpermute(paddr, __builtin_bit_cast(int, xi));
            if (pel == e) { px = __builtin_bit_cast(float, gx ^ (int)sgn); py = __builtin_bit_cast(float, gy ^ (int)sgn); }
        }
        const float v = atan2_fast(py, px);
        float p0, p1;
        if (Mp <= 16) {
            const float prev = dpp_mov<0x111, false>(v, v);             // row_shr:1, lane 0 of the row keeps its own
            const float turns = rintf((v - prev) * 0.15915494309189535f);
            const float y = fmaf(-TWO_PI_F, row_scan_fast(turns), v);
            p0 = row_total_dpp(pf0 * y);
            p1 = row_total_dpp(pf1 * y);
        } else {
            const float prev = dpp_mov<0x138, false>(v, v);             // wave_shr:1, lane 0 keeps its own
            const float turns = rintf((v - prev) * 0.15915494309189535f);
            const float y = fmaf(-TWO_PI_F, wave_scan_fast(turns), v);
            p0 = wave_total_dpp(pf0 * y);
            p1 = wave_total_dpp(pf1 * y);
        }
        pc4 += (uint32_t)Mp * 4u; pc4 = pc4 >= 255u * 4u ? pc4 - 255u * 4u : pc4;
        p1 = 0.3f * p1 + (1.0f - 0.3f) * p1_prime;
        p1_prime = p1;
        // ---- de-rotate, soft bits (stored at the top of the next turn)
        const float p0r = p0 * 0.15915494309189535f;
#pragma unroll
        for (int e = 0; e < E; e++) {
            Zp[e] = rot_down_pk(x[e], fmaf(p1, fxr[e], p0r));
            swp[e] = demod_pk<MOD>(reinterpret_cast<const uint8_t *>(qnb), Zp[e], soft_mode);
        }
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

// One wave per hand-off, the live list walked with a grid stride (the host sizes the grid from the previous launch's frame count);
// every modem in this one launch (no second launch behind it: receivers of wide symbols are few-channel receivers, whose pushes
// are chains of launches already).
template <int XB, int E>
__global__ __launch_bounds__(WV, (E >= 4 ? 2 : 4)) void payload_wide_kernel(SyncArgs a)
{
    launder(a);
    __shared__ uint32_t qsg[256 + 64];      // pilot polarity as a sign mask; the 255-long sequence continued past its end: no wrap inside a symbol
    __shared__ uint32_t qnb[64];            // the soft demodulator's nearest-neighbour table of this frame's modem
    __shared__ int qsrc[64];                // (element << 8) | lane holding pilot r after the transform
    const SyncConsts &c = a.c;
    const int l = lane_id();
    const uint32_t nj = a.max_jobs;
    uint32_t nlist = a.live[0];
    if (nlist > nj) nlist = nj;
    if (blockIdx.x >= nlist) return;
    for (int k = l; k < 256 + 64; k += WV) qsg[k] = c.pilot_seq[k >= 255 ? k - 255 : k] == 0 ? 0x80000000u : 0u;
    qsrc[l] = 0;
    wave_sync_lds();
    constexpr int LOG2E = E == 2 ? 1 : (E == 4 ? 2 : 3);
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int kk = (int)(__brev((unsigned)(l + WV * e)) >> (32 - 6 - LOG2E));
        const int pr = c.pilot_rank[kk];
        if (pr >= 0 && pr < 64) qsrc[pr] = (e << 8) | l;
    }
    wave_sync_lds();
    uint32_t mod_in_lds = 0;
    for (uint32_t k = blockIdx.x; k < nlist; k += gridDim.x) {
        const uint32_t j = rfl(a.live[1 + k]);
        if (j >= nj) continue;
        const PayloadJob *job = a.jobs + j;
        const uint32_t ch = rfl(job->ch);
        if (ch >= a.nch || job->arena_off == ~0ull) continue;
        const uint32_t mod = rfl(job->s.mod_scheme);
        if ((mod == 27 || mod == 29) && mod != mod_in_lds) {
            wave_sync_lds();
            if (mod == 27) { if (l < 16) qnb[l] = reinterpret_cast<const uint32_t *>(c.cod.qam16_nb)[l]; }
            else qnb[l] = reinterpret_cast<const uint32_t *>(c.cod.qam64_nb)[l];
            wave_sync_lds();
            mod_in_lds = mod;
        }
        uint32_t dth;
        if (mod == 39)      dth = lean::symbols_wide<39, XB, E>(a, job, ch, j, qsg, qnb, qsrc);
        else if (mod == 40) dth = lean::symbols_wide<40, XB, E>(a, job, ch, j, qsg, qnb, qsrc);
        else if (mod == 27) dth = lean::symbols_wide<27, XB, E>(a, job, ch, j, qsg, qnb, qsrc);
        else                dth = lean::symbols_wide<29, XB, E>(a, job, ch, j, qsg, qnb, qsrc);
        if (l == 0) a.jobs[j].s.nco_dtheta = dth;
    }
}
