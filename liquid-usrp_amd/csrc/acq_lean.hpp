// acq_lean.hpp -- the segment waves of the acquisition for 64-subcarrier symbols, the instruction and register diet.
// Included by ofdmsync.hip (part 3) inside namespace mcrx, behind the Walker.
//
// What a segment wave does is Walker::run_seg (ofdmsync.hip; kernels.h "Segment-parallel acquisition"): the synchronizer's own events
// -- liquid-dsp ofdmframesync_execute_seekplcp / _S0a / _S0b / _S1 / _rxsymbols and ofdmflexframesync's header, called per channel
// sample from the reference's lib/multichannelrx.cc:193-194 -- from a start state through a segment of the channel's stretch of the
// push, frame after frame, every hand-off parked in a job list entry and a slot keyed by the state the frame was acquired from.  The
// Walker's build of it (sync_spec_kernel<1>) holds the whole state machine's registers: 248 VGPRs, two waves per SIMD, one dependent
// chain each -- 2 x 0.10 ms per 8192-frame slab on a chip it fills with 2048 waves, while a frame's acquisition is seventeen events of
// the kind the lean payload workers run 165 of per frame at eight waves per SIMD (payload_lean.hpp).  The same treatment, tried here:
//   * the synchronizer state lives in scalars (every wave-uniform result goes through v_readfirstlane), positions inside the kernel
//     are 32-bit offsets into the buffer;
//   * one transform: the payload workers' six packed-f32 stages with the partner through the LDS crossbar (lean_prims.hpp);
//   * the neighbour subcarrier of the S0 / S1 metrics (X[k + 2], X[k + 1]), the pilots and the equaliser fit's rank order are
//     ds_bpermute gathers with addresses fixed per lane -- no LDS memory, no fences between the transform and the sums;
//   * the S1 fit's factors (5 + 5 floats per lane) are fetched when a frame is detected, not held across the walk.
// The arithmetic per event is the Walker's (same sums in the same lane order, same thresholds, atan2f / roundf where it decides a
// timer); the transform's twiddle products are fused differently (one packed mul + one packed fma), as in the payload workers.
// MEASURED, AND NOT THE DEFAULT (round 5; scratch/r5/acq_ab.sh, acq_ab2.sh, m48_ab.py; profiles/r5_acq_*): the kernel issues half the
// Walker's instructions (42 M against 84 M VALU per 8192-frame slab) and takes the same time -- 2 x 0.10 ms alone, 181-183 Gsample/s
// either way at 512 channels, M = 64 and M = 48 -- because a segment wave is a CHAIN: ~250 dependent instructions per event, ~6 k
// cycles whoever issues them (LDS-crossbar round trips, DPP scans, v_readfirstlane -> scalar -> vector turnarounds), and with every wave
// of the launch resident at once the launch lasts as long as its longest chain.  On ragged traffic the chain is a sixth longer here
// (the LDS staging's round trip in front of every ~700 samples; 37 against 32 us for one frame alone) and the receiver 10 % slower
// (127 against 142 Gsample/s).  It needs ~165 registers -- at 128 (four waves per SIMD) forty of them spill onto every event's chain,
// 136 against 110 us per launch.  mcrx_hip_config::scout_build = 2 selects it (designs with 48 or 64 subcarriers and at most 16
// pilots); the tests hold it to the oracle like the default (tests/test_gpu_parity.py, test_gpu_alloc.py).

#include "lean_prims.hpp"
#ifndef ACQ_LEAN_WAVES
#define ACQ_LEAN_WAVES 3        /* waves per SIMD the kernel is built for (mcrx_hip.hip sizes the segments by it: acq_lean_waves()).  Three = 168
                                   registers: the kernel's real pressure is ~165, and at four (128) the 40 spilled registers sit on every event's
                                   chain -- 136 against 110 us per launch, scratch/r5/acq_ab.sh */
#endif
#ifndef ACQ_WIN
#define ACQ_WIN 768             /* samples of the channel staged in LDS at a time: 6 KB per wave (a 64-subcarrier frame's acquisition spans ~720) */
#endif
// -DACQ_PROF (development builds): cycles per kind of event of one wave (channel 0, segment 0), printed when it leaves
#ifdef ACQ_PROF
#define ACQ_T0() const long long acq_t0_ = (long long)__builtin_readcyclecounter()
#define ACQ_T1(i) do { prof_c[i] += (long long)__builtin_readcyclecounter() - acq_t0_; prof_n[i]++; } while (0)
#else
#define ACQ_T0() do { } while (0)
#define ACQ_T1(i) do { } while (0)
#endif
namespace lean {

__device__ __forceinline__ float rff(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ int64_t rfl64(int64_t v)
{ return (int64_t)(((uint64_t)rfl((uint32_t)((uint64_t)v >> 32)) << 32) | (uint64_t)rfl((uint32_t)v)); }
__device__ __forceinline__ float gather(int addr, float v) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v))); }

template <int XB, int MM>
struct Acq {
    static constexpr int M = MM, M2 = MM / 2;
    SyncArgs &a;
    const SyncConsts &c;
    const int l;
    const uint32_t ch;
    // ---- lane constants
    int bp32, kk, dr, er, src1, src2, psrc, esrc, half;
    bool sct, lane_on;
    Radix3 r3;                  // (MM = 48 only)
    float S0v, S1v, fxr, pf0, pf1;
    v2f tw[6], sgp[3], R;
    // ---- LDS (one wave per workgroup)
    uint32_t *qsg; uint16_t *hmap; uint8_t *hbits; uint16_t *hd;
    v2f *wbuf; int32_t w0;      // ACQ_WIN samples of the channel from buffer offset w0 on (INT32_MIN: nothing): see window()
    // ---- buffer geometry
    const float2 *chb; uint32_t tstride; int32_t rlen, rmin;
    int L, cp, cb, backoff, Mp, Md, Nen;
    float gain0, gain1;
    // ---- synchronizer state (wave uniform, scalars)
    int st; uint32_t timer; int64_t cur;
    uint32_t th_ref, dth; int64_t t_ref;
    float g0, sh0x, sh0y;
    uint32_t nsym, pc;
    float phi, p1p;
    int fstate; uint32_t hsi;
    float evm_hat, evm;
    uint32_t hw[4]; int hvalid;
    uint32_t plen, mods, bps, chk, fec0, fec1, enc_len, mod_len;
    uint32_t period_hint, burst_hint; int64_t last_fresh;      // carried into the hand-offs as the channel state has them
    int64_t sk_cur; uint32_t sk_timer;
    // ---- hand-off bookkeeping
    uint32_t jblk_next, jblk_end, handoff_job; int64_t handoff_last;
#ifdef ACQ_PROF
    long long prof_c[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; int prof_n[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};     // seek, s0a, s0b, s1, rx, decode, scan, init, fill, park
#endif

    __device__ __forceinline__ Acq(SyncArgs &a_, uint32_t ch_, uint32_t *qsg_, uint16_t *hmap_, uint8_t *hbits_, uint16_t *hd_, v2f *wbuf_)
        : a(a_), c(a_.c), l(lane_id()), ch(ch_), qsg(qsg_), hmap(hmap_), hbits(hbits_), hd(hd_), wbuf(wbuf_), w0(INT32_MIN) {}

    // ------------------------------------------------------------------ setup
    __device__ __forceinline__ void init(int *qsrc, int *qesrc)
    {
        L = c.L; cp = c.cp; backoff = c.backoff; cb = cp - backoff; Mp = c.M_pilot; Md = c.M_data; Nen = c.Nen;
        bp32 = (l ^ 32) << 2;
        const int kq = lane_k<MM>(l);
        lane_on = kq >= 0; kk = lane_on ? kq : 0;                  // (a lane behind a 48-sample symbol holds zeros and owns no subcarrier)
        dr = lane_on ? c.data_rank[kk] : -1; er = lane_on ? c.en_rank[kk] : -1;
        const int pr = lane_on ? c.pilot_rank[kk] : -1;
        sct = lane_on && c.sctype[kk] != 0;
        S0v = lane_on ? c.S0[kk] : 0.f; S1v = lane_on ? c.S1[kk] : 0.f;
        fxr = ((kk > M2) ? (float)kk - (float)M : (float)kk) * 0.15915494309189535f;
        src1 = k_lane<MM>((kk + 1) % M) << 2;
        src2 = k_lane<MM>((kk + 2) % M) << 2;
        half = ((l + M2) & 63) << 2;                                 // the lane M / 2 samples on (the CFO estimate's second half)
        if constexpr (MM == 48) r3 = radix3_consts(l);
#pragma unroll
        for (int s = 0; s < 6; s++) {
            const int h = 32 >> s;
            const bool up = (l & h) != 0;
            const float rev = (float)(l & (h - 1)) * (0.5f / (float)h);
            tw[s].x = up ? __builtin_amdgcn_cosf(rev) : 1.f; tw[s].y = up ? -__builtin_amdgcn_sinf(rev) : 0.f;
            if (s & 1) sgp[s >> 1].y = up ? -1.f : 1.f; else sgp[s >> 1].x = up ? -1.f : 1.f;
        }
        pf0 = (l < Mp) ? c.Pfit[l] : 0.f; pf1 = (l < Mp) ? c.Pfit[Mp + l] : 0.f;
        gain0 = sqrtf((float)c.M_S0) / (float)M; gain1 = sqrtf((float)c.M_S1) / (float)M;
        {   // the two tables: every request issued before the first LDS store (one round trip to memory, not one per chunk -- the
            // set-up was 22 k cycles of a wave's ~110 k per frame)
            uint8_t ps[5]; uint32_t hm[3];
#pragma unroll
            for (int u = 0; u < 5; u++) { const int k = l + WV * u, kc = k < 256 + 16 ? k : 0; ps[u] = c.pilot_seq[kc >= 255 ? kc - 255 : kc]; }
#pragma unroll
            for (int u = 0; u < 3; u++) { const int k = l + WV * u; hm[u] = reinterpret_cast<const uint32_t *>(c.hdr_map)[k < MCRX_HDR_SYMS / 2 ? k : 0]; }
#pragma unroll
            for (int u = 0; u < 5; u++) { const int k = l + WV * u; if (k < 256 + 16) qsg[k] = ps[u] == 0 ? 0x80000000u : 0u; }
#pragma unroll
            for (int u = 0; u < 3; u++) { const int k = l + WV * u; if (k < MCRX_HDR_SYMS / 2) reinterpret_cast<uint32_t *>(hmap)[k] = hm[u]; }
        }
        if (l < 16) qsrc[l] = 0;
        qesrc[l] = 0;
        wave_sync_lds();
        if (pr >= 0 && pr < 16) qsrc[pr] = l;
        if (er >= 0 && er < 64) qesrc[er] = l;
        wave_sync_lds();
        psrc = qsrc[l < Mp ? l : 0] << 2;
        esrc = qesrc[l < Nen ? l : (Nen > 0 ? Nen - 1 : 0)] << 2;
        wave_sync_lds();
        chb = a.chan + ((size_t)a.chan_off + ch) * MCRX_TILE_S;
        tstride = a.chan_stride * (uint32_t)MCRX_TILE_S;
        rlen = (int32_t)(a.end - a.buf_first);
        rmin = a.buf_first < 0 ? (int32_t)(-a.buf_first) : 0;         // (samples of negative absolute index are zeros: Walker::sample)
        R.x = 0.f; R.y = 0.f;
        handoff_job = 0; handoff_last = 0;
    }
    __device__ __forceinline__ void reset_framesync()
    {
        st = SY_SEEK; timer = 0; nsym = 0; pc = 0; th_ref = 0; dth = 0; t_ref = 0; sh0x = 0.f; sh0y = 0.f; phi = 0.f; p1p = 0.f;
        fstate = FX_HEADER; hsi = 0; evm_hat = 0.f;
    }
    __device__ __forceinline__ void load_state(const ChanState &s)
    {
        st = (int)rfl((uint32_t)s.state); timer = rfl(s.timer); cur = rfl64(s.cur);
        th_ref = rfl(s.nco_theta_ref); dth = rfl(s.nco_dtheta); t_ref = rfl64(s.nco_t_ref);
        g0 = rff(s.g0); sh0x = rff(s.s_hat_0.x); sh0y = rff(s.s_hat_0.y);
        nsym = rfl(s.num_symbols); pc = rfl(s.pilot_count); phi = rff(s.phi_prime); p1p = rff(s.p1_prime);
        fstate = (int)rfl((uint32_t)s.fstate); hsi = rfl(s.header_symbol_index); evm_hat = rff(s.evm_hat); evm = rff(s.evm);
        plen = rfl(s.payload_len); mods = rfl(s.mod_scheme); bps = rfl(s.bps); chk = rfl(s.check); fec0 = rfl(s.fec0); fec1 = rfl(s.fec1);
        enc_len = rfl(s.enc_len); mod_len = rfl(s.mod_len); hvalid = (int)rfl((uint32_t)s.header_valid);
#pragma unroll
        for (int w = 0; w < 4; w++) hw[w] = rfl(s.hw[w]);
        period_hint = rfl(s.period_hint); burst_hint = rfl(s.burst_hint); last_fresh = rfl64(s.last_fresh);
    }
    __device__ __forceinline__ ChanState make_state() const
    {
        ChanState s;
        s.state = st; s.timer = timer; s.cur = cur; s.nco_theta_ref = th_ref; s.nco_dtheta = dth; s.nco_t_ref = t_ref;
        s.g0 = g0; s.s_hat_0 = make_float2(sh0x, sh0y); s.num_symbols = nsym; s.pilot_count = pc; s.phi_prime = phi; s.p1_prime = p1p;
        s.fstate = fstate; s.header_symbol_index = hsi; s.payload_symbol_index = 0; s.evm_hat = evm_hat; s.evm = evm;
        s.payload_len = plen; s.mod_scheme = mods; s.bps = bps; s.check = chk; s.fec0 = fec0; s.fec1 = fec1; s.enc_len = enc_len; s.mod_len = mod_len;
        s.header_valid = hvalid;
#pragma unroll
        for (int w = 0; w < 4; w++) s.hw[w] = hw[w];
        s.period_hint = period_hint; s.last_fresh = last_fresh; s.burst_hint = burst_hint; s.burst_pad = 0;
        return s;
    }

    // ------------------------------------------------------------------ windows
    // lane l's sample of the window that starts at absolute sample t0 (zeros in front of the stream / of the buffer).
    // A frame's acquisition is seventeen windows within ~700 consecutive samples, each one's address known only when the event before
    // it has set the timer: seventeen dependent HBM round trips, half of the 32 us a frame took.  The channel's samples are therefore
    // staged in LDS ACQ_WIN at a time -- ten requests per lane in flight together, from the first window that misses on -- and the
    // events read their windows from there (a window's 64 consecutive samples: conflict free whatever its alignment).
    __device__ __forceinline__ v2f sample_at(int32_t r) const
    {
        const bool zero = r < rmin;
        r = r < 0 ? 0 : (r >= rlen ? rlen - 1 : r);
        const v2f v = *reinterpret_cast<const v2f *>(chb + ((size_t)(uint32_t)(r >> MCRX_TILE_SH) * tstride + (uint32_t)(r & (MCRX_TILE_S - 1))));
        v2f z; z.x = zero ? 0.f : v.x; z.y = zero ? 0.f : v.y;
        return z;
    }
    __device__ __forceinline__ v2f window(int64_t t0)
    {
        const int32_t r0 = (int32_t)(t0 - a.buf_first);
        stage_from(r0, WV);
        v2f v = wbuf[r0 - w0 + l];
        if constexpr (MM < WV) { if (l >= MM) { v.x = 0.f; v.y = 0.f; } }          // (a window is MM samples)
        return v;
    }
    // make [r0, r0 + n) available in wbuf (n <= ACQ_WIN); staged from r0 on when it is not
    __device__ __forceinline__ void stage_from(int32_t r0, int32_t n)
    {
        if (!(w0 != INT32_MIN && r0 >= w0 && r0 + n <= w0 + ACQ_WIN)) {
            ACQ_T0();
            wave_sync_lds();
            static_assert((ACQ_WIN / WV) % 6 == 0, "the staging loop takes six requests per lane at a time");
#pragma unroll 1
            for (int h = 0; h < ACQ_WIN / WV; h += 6) {                  // six requests per lane in flight at a time (registers)
                v2f t[6];
#pragma unroll
                for (int i = 0; i < 6; i++) t[i] = sample_at(r0 + WV * (h + i) + l);
#pragma unroll
                for (int i = 0; i < 6; i++) wbuf[WV * (h + i) + l] = t[i];
            }
            wave_sync_lds();
            w0 = r0;
            ACQ_T1(8);
        }
    }
    __device__ __forceinline__ v2f fft64(v2f x) const             // (the MM-point transform; the name is round 5's first build's)
    {
        if constexpr (MM == 64) return lean::fft64<XB>(x, tw, sgp, bp32);
        else return lean::fft48<XB>(x, tw, sgp, r3, bp32, l);
    }
    // S0 metric of a window (Walker::s0_metric_of): its power, and sum over the even subcarriers of X[k + 2] conj X[k], / M_S0
    __device__ __forceinline__ float2 s0_metric(v2f x, float &power) const
    {
        power = wave_total_dpp(x.x * x.x + x.y * x.y);
        x = fft64(x);
        const float sc = S0v * gain0;
        const float xr = x.x * sc, xi = x.y * sc;
        const float pr_ = gather(src2, xr), pi_ = gather(src2, xi);
        float2 t = cmulc(make_float2(pr_, pi_), make_float2(xr, xi));
        if ((kk & 1) || !lane_on) t = make_float2(0.f, 0.f);
        const float inv = 1.0f / (float)c.M_S0;
        return make_float2(wave_total_dpp(t.x) * inv, wave_total_dpp(t.y) * inv);
    }

    // ------------------------------------------------------------------ events
    __device__ __forceinline__ void seek_decide(float2 sh, float pw)
    {
        const float g = (float)M / pw;
        sh = cscale(sh, g);
        const float tau = atan2f(sh.y, sh.x) * (float)M2 / TWO_PI_F;
        g0 = rff(g); timer = 0;
        if (sqrtf(sh.x * sh.x + sh.y * sh.y) > c.detect_thresh) {
            const int dt = (int)roundf(tau);
            timer = rfl((uint32_t)(M + dt) % (uint32_t)M2 + (uint32_t)M);
            st = SY_S0A;
        }
    }
    static constexpr int SEEK_B = 4;
    __device__ __forceinline__ void seek_burst()
    {
        // (the first window stages ACQ_WIN samples: the burst's windows, and a detected frame's next ones, are in LDS)
#pragma unroll 1
        for (int j = 0; j < SEEK_B; j++) {
            float pw; const float2 sh = s0_metric(window(cur), pw);
            cur += M;
            seek_decide(sh, pw);
            if (st != SY_SEEK) return;
        }
    }
    // SEEK, S0a, S0b, S1 (incl. the equaliser fit): Walker::sync_event
    __device__ __forceinline__ void sync_event(int64_t t_ev)
    {
        ACQ_T0();
        const int st_in = st; (void)st_in;
        if (st == SY_SEEK) {
            float pw; const float2 sh = s0_metric(window(t_ev - M + 1), pw);
            seek_decide(sh, pw);
        } else if (st == SY_S0A) {
            float pw; const float2 sh = s0_metric(window(t_ev - M + 1), pw);       // (the oscillator stands still until S0b sets it: mixing is the identity)
            sh0x = rff(sh.x * g0); sh0y = rff(sh.y * g0);
            timer = 0; st = SY_S0B;
        } else if (st == SY_S0B) {
            float pw; float2 sh = s0_metric(window(t_ev - M + 1), pw);
            sh = cscale(sh, g0);
            const float2 ssum = make_float2(sh0x + sh.x, sh0y + sh.y);
            const float tau = atan2f(ssum.y, ssum.x) * (float)M2 / TWO_PI_F;
            timer = rfl((uint32_t)(M + cp - backoff) - (uint32_t)(int)roundf(tau));
            // CFO: time-domain ML estimate over the two halves of the oldest M window samples (lanes 0..31: sample i and i + M/2)
            const v2f y = window(t_ev - L + 1);
            int i_s = l < M ? l : 0;
            asm volatile("" : "+v"(i_s));
            const float2 s0 = c.s0t[i_s];
            const float r1x = gather(half, y.x), r1y = gather(half, y.y), sbx = gather(half, s0.x), sby = gather(half, s0.y);
            float2 t = cmul(cmulc(s0, make_float2(y.x, y.y)), cmulc(make_float2(r1x, r1y), make_float2(sbx, sby)));
            if (l >= M2) t = make_float2(0.f, 0.f);
            const float ax = wave_total_dpp(t.x), ay = wave_total_dpp(t.y);
            const float nu = atan2f(ay, ax) / (float)M2;
            dth = rfl(rad2u32(nu)); th_ref = 0; t_ref = t_ev + 1;
            st = SY_S1;
        } else if (st == SY_S1) {
            nsym++;
            const int64_t t0 = t_ev - M + 1;
            v2f x = window(t0);
            {   // the oscillator runs from t_ref on (Walker::mix_window)
                const int64_t t = t0 + l;
                if (t >= t_ref) { const float2 m = mix_down_hw(make_float2(x.x, x.y), th_ref + (uint32_t)(t - t_ref) * dth); x.x = m.x; x.y = m.y; }
            }
            x = fft64(x);
            const float sc = S1v * gain1;
            const float xr = x.x * sc, xi = x.y * sc;
            float2 t = cmulc(make_float2(gather(src1, xr), gather(src1, xi)), make_float2(xr, xi));
            if (!lane_on) t = make_float2(0.f, 0.f);
            const float2 acc = make_float2(wave_total_dpp(t.x), wave_total_dpp(t.y));
#if MCRX_S1_METRIC_G0_NORMALISED
            float2 gh = cscale(acc, g0 / (float)c.M_S1);
#else
            float2 gh = cscale(acc, 1.0f / (float)c.M_S1);
#endif
            gh = cmul(gh, c.backoff_rot);
            const float mag = sqrtf(gh.x * gh.x + gh.y * gh.y);
            if (mag > c.sync_thresh && fabsf(atan2f(gh.y, gh.x)) < 0.1f * PI_F) {
                st = SY_RX; timer = (uint32_t)(M + cp + backoff); nsym = 0;
                // equaliser: order-4 LSQ smoothing of |G| and unwrapped arg G through the fit's orthonormal basis, R = 1 / G
                const float g = (float)M / sqrtf((float)(Mp + Md));
#if MCRX_S1_BACKOFF_CORRECTION
                float sb_, cb_; sincos_u32((uint32_t)(((uint64_t)(unsigned)kk * (unsigned)backoff << 32) / (unsigned)M), sb_, cb_);
                const float2 G = cmul(cscale(make_float2(xr, xi), g), make_float2(cb_, sb_));
#else
                const float2 G = cscale(make_float2(xr, xi), g);
#endif
                const float gab = sqrtf(G.x * G.x + G.y * G.y), gar = atan2f(G.y, G.x);
                float smn[5], smk[5];
                int i_n = l * 5, i_k = kk * 5;
                asm volatile("" : "+v"(i_n), "+v"(i_k));           // (opaque: the ten loads stay here, in the one event per frame that needs them -- hoisted
                                                                    //  out of the event loop they were ten registers held for the whole walk)
#pragma unroll
                for (int d = 0; d < 5; d++) { smn[d] = (l < Nen) ? c.smn[i_n + d] : 0.f; smk[d] = c.smk[i_k + d]; }
                // rank order: lane n takes the enabled bin of rank n
                // (both gathers outside any lane condition: ds_bpermute returns zero for a source lane that is masked off)
                const float gva = gather(esrc, gab);
                float vy = gather(esrc, gar);
                const float va = (l < Nen) ? gva : 0.f;
                const float prev = dpp_mov<0x138, false>(vy, vy);
                const float run = rintf((vy - prev) * 0.15915494309189535f);
                const float before = wave_scan_fast(run) - run;
                vy = fmaf(-TWO_PI_F, before + run, vy);
                float ca[5], ct[5];
#pragma unroll
                for (int d = 0; d < 5; d++) { ca[d] = wave_total_dpp(smn[d] * va); ct[d] = wave_total_dpp(smn[d] * vy); }
                float2 r = make_float2(0.f, 0.f);
                if (sct) {
                    float A = 0.f, th = 0.f;
#pragma unroll
                    for (int d = 0; d < 5; d++) { A = fmaf(smk[d], ca[d], A); th = fmaf(smk[d], ct[d], th); }
                    const float kq = rintf(th * 0.15915494309189535f);
                    float rr = fmaf(-kq, 6.28125f, th); rr = fmaf(-kq, 1.9353071795864769e-3f, rr);
                    const float rev = rr * 0.15915494309189535f;
                    const float gr = A * __builtin_amdgcn_cosf(rev), gi = A * __builtin_amdgcn_sinf(rev);
                    const float dd = gr * gr + gi * gi;
                    r = make_float2(gr / dd, -gi / dd);
                }
                R.x = r.x; R.y = r.y;
            } else {
                if (nsym == 16) reset_framesync();
                timer = (uint32_t)M2;
            }
        }
        ACQ_T1(st_in);
    }

    // 12 Golay(24,12) words -> 18 bytes, CRC-32 over the first 14, the payload's configuration (Walker::decode_header_fast + header_fields)
    __device__ __forceinline__ void decode_header()
    {
        wave_sync_lds();
        if (l < 12) {
            unsigned r = 0;
#pragma unroll
            for (int q = 0; q < 24; q++) r = (r << 1) | (unsigned)hbits[24 * l + q];
            hd[l] = (uint16_t)golay_dec_sym(r);
        }
        wave_sync_lds();
        unsigned by[18];
#pragma unroll
        for (int g = 0; g < 6; g++) {
            const unsigned s0 = hd[2 * g], s1 = hd[2 * g + 1];
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
            hw[w] = rfl(v);
        }
        bool ok = rfl(crc == key ? 1u : 0u) != 0;
        auto hbyte = [&](int i) { return (hw[i >> 2] >> (8 * (i & 3))) & 0xffu; };
        const unsigned proto = hbyte(8), pl = (hbyte(9) << 8) | hbyte(10), mod = hbyte(11);
        const unsigned check = (hbyte(12) >> 5) & 7, f0 = hbyte(12) & 0x1f, f1 = hbyte(13) & 0x1f;
        const unsigned bp = mod_bps_d(mod);
        if (proto != 104 || bp == 0 || !(check == 1 || check == 6) || !fec_known_d(f0) || !fec_known_d(f1)) ok = false;
        hvalid = ok ? 1 : 0;
        if (ok) {
            plen = pl; mods = mod; bps = bp; chk = check; fec0 = f0; fec1 = f1;
            const unsigned n0 = pl + (check == 6 ? 4u : 0u);
            enc_len = fec_enc_len_d(f1, fec_enc_len_d(f0, n0));
            const unsigned nb = 8 * enc_len;
            mod_len = nb / bp + ((nb % bp) ? 1u : 0u);
        }
    }

    // job list entries come in blocks (Walker::reserve_block / park_state / void_block)
    // (the counter's answer is only looked at when the first entry is needed -- jblk_wait -- so the atomic's round trip runs under
    //  whatever the wave does in between: the first block is asked for before the set-up)
    uint32_t jblk_v; bool jblk_pending = false;
    __device__ __forceinline__ void reserve_block()
    {
        const uint32_t nb = a.seg_jobs ? a.seg_jobs : 1u;
        jblk_v = 0;
        if (l == 0) jblk_v = atomicAdd(a.njobs, nb);
        jblk_pending = true;
    }
    __device__ __forceinline__ void jblk_wait()
    {
        if (!jblk_pending) return;
        const uint32_t nb = a.seg_jobs ? a.seg_jobs : 1u;
        jblk_next = rfl(jblk_v); jblk_end = jblk_next + nb; jblk_pending = false;
    }
    __device__ __forceinline__ uint32_t park_state()
    {
        jblk_wait();
        if (jblk_next == jblk_end) { reserve_block(); jblk_wait(); }
        const uint32_t j = jblk_next++;
        if (j >= a.max_jobs) return 0xFFFFFFFFu;
        if (l == 0) { PayloadJob jb; jb.s = make_state(); jb.ch = 0xFFFFFFFFu; jb.pad = 0; jb.arena_off = 0; jb.syms_off = 0; a.jobs[j] = jb; }
        return j;
    }
    __device__ __forceinline__ void void_block()
    {
        jblk_wait();
        for (uint32_t q = jblk_next + (uint32_t)l; q < jblk_end; q += WV) if (q < a.max_jobs) a.jobs[q].ch = 0xFFFFFFFFu;
        jblk_next = jblk_end;
    }
    __device__ __forceinline__ bool oversize() const { return enc_len > c.max_enc_len || mod_len > c.max_syms || plen > c.max_payload_len; }

    // one header symbol (Walker::rx_event_fast<SYM_SPEC> + flex_header_fast).  Returns 0: the header continues, 1: nothing for a
    // segment wave to park, 2: handed off, 4: deferred, 5: header decoded, check failed
    __device__ __forceinline__ int rx_event(int64_t t_ev)
    {
        ACQ_T0();
        const int64_t ws = t_ev - L + 1 + cb;
        v2f x = window(ws);
        const uint32_t th_ws = th_ref + (uint32_t)(ws - t_ref) * dth;
        {   // (the Walker's fused rotation: every sample of the window, explicit fma shape)
            const float2 m = rot_down(make_float2(x.x, x.y), u32rev(th_ws + (uint32_t)l * dth)); x.x = m.x; x.y = m.y;
        }
        x = fft64(x);
        x = cmul_pk(x, R);
        const float xr = x.x, xi = x.y;
        uint32_t pi_ = pc + (uint32_t)(l < 16 ? l : 0);                      // (the table runs 16 entries past the sequence's end)
        const uint32_t sgn = qsg[pi_];
        float2 P;
        P.x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(psrc, __builtin_bit_cast(int, xr)) ^ (int)sgn);
        P.y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(psrc, __builtin_bit_cast(int, xi)) ^ (int)sgn);
        const float v = atan2_fast(P.y, P.x);
        const float prev = dpp_mov<0x138, false>(v, v);
        const float turns = rintf((v - prev) * 0.15915494309189535f);
        const float y = fmaf(-TWO_PI_F, row_scan_fast(turns), v);
        const float p0 = row_total_dpp(pf0 * y);
        float p1 = row_total_dpp(pf1 * y);
        pc += (uint32_t)Mp; pc = pc >= 255u ? pc - 255u : pc;
        p1 = 0.3f * p1 + (1.0f - 0.3f) * p1p;
        p1p = rff(p1);
        const float p0r = p0 * 0.15915494309189535f;
        float2 X = make_float2(0.f, 0.f);
        if (sct) X = rot_down(make_float2(xr, xi), fmaf(p1, fxr, p0r));
        uint32_t nd = dth;
        if (nsym > 0) {
            float dphi = p0 - phi;
            dphi -= TWO_PI_F * rintf(dphi * 0.15915494309189535f);
            nd += rad2u32(1e-3f * dphi);
        }
        th_ref = th_ref + (uint32_t)(t_ev + 1 - t_ref) * dth;
        t_ref = t_ev + 1;
        dth = rfl(nd);
        phi = rff(p0);
        nsym++;
        timer = (uint32_t)L;
        // hard BPSK bits straight to their de-interleaved, de-scrambled place
        float ev = 0.f;
        if (dr >= 0) {
            const uint32_t idx = hsi + (uint32_t)dr;
            if (idx < MCRX_HDR_SYMS) {
                const unsigned sym = X.x > 0 ? 0u : 1u;
                const unsigned m = hmap[idx];
                hbits[m & 0x1ffu] = (uint8_t)(sym ^ (m >> 15));
                const float xh = sym ? -1.0f : 1.0f;
                const float drr = xh - X.x, dii = -X.y;
                const float e_ = sqrtf(drr * drr + dii * dii);
                ev = e_ * e_;
            }
        }
        evm_hat = rff(evm_hat + wave_total_dpp(ev));
        hsi += (uint32_t)Md;
        if (hsi < MCRX_HDR_SYMS) { ACQ_T1(4); return 0; }
        ACQ_T1(4);
        { ACQ_T0(); decode_header(); ACQ_T1(5); }
        evm = rff(10.0f * log10f(evm_hat / (float)MCRX_HDR_SYMS));
        if (!hvalid) return 5;
        fstate = FX_PAYLOAD;
        const int64_t nps = (int64_t)((mod_len + (uint32_t)Md - 1) / (uint32_t)Md);
        const int64_t t_last = t_ev + nps * (int64_t)L;
        if (!oversize() && t_last < a.end) {
            const uint32_t j = park_state();
            if (j != 0xFFFFFFFFu) {
                if (lane_on) a.jR[(size_t)j * M + kk] = make_float2(R.x, R.y);
                handoff_job = j; handoff_last = t_last;
                return 2;
            }
        }
        return (!oversize() && t_last >= a.end && a.defer_limit > 0 && a.end - sk_cur <= a.defer_limit) ? 4 : 1;
    }

    // ------------------------------------------------------------------ where a wave starts
    // coarse preamble finder (Walker::coarse_scan): lane l takes the M samples at from + l M/2, lag-M/2 autocorrelation against energy
    __device__ __forceinline__ int64_t coarse_scan(int64_t from, int64_t to)
    {
        ACQ_T0();
        const int64_t r_ = coarse_scan_(from, to);
        ACQ_T1(6);
        return r_;
    }
    __device__ __forceinline__ int64_t coarse_scan_(int64_t from, int64_t to)
    {
        if (to > a.end - M) to = a.end - M;
        if (from < a.buf_first) from = a.buf_first;
        if (from < 0) from = 0;
        // A range that fits the staging buffer (the validation scans behind a lattice point: four symbols) is read from LDS: one round
        // trip to memory for the wave instead of one per eight products, and -- staged from M samples in front of `from` -- the buffer
        // then already holds the windows of the frame the wave goes on to acquire from that point.
        if (to > from && (to - from) + M + M2 + M <= (int64_t)ACQ_WIN && from - M >= a.buf_first && from - M >= 0) {
            const int32_t rb = (int32_t)(from - M - a.buf_first);
            stage_from(rb, (int32_t)((to - from) + 2 * M + M2));
            const int64_t d = from + (int64_t)l * M2;
            const bool in = d < to;
            const v2f *w = wbuf + (rb - w0) + M + (in ? l * M2 : 0);
            float2 acc = make_float2(0.f, 0.f); float en = 0.f;
#pragma unroll 4
            for (int n = 0; n < M2; n++) {
                const v2f u = w[n], v = w[n + M2];
                acc = cadd(acc, cmulc(make_float2(u.x, u.y), make_float2(v.x, v.y)));
                en += u.x * u.x + u.y * u.y + v.x * v.x + v.y * v.y;
            }
            const bool hit = in && (acc.x * acc.x + acc.y * acc.y) > 0.09f * en * en;
            const unsigned long long b = __ballot(hit);
            return b ? from + (int64_t)__builtin_ctzll(b) * M2 : -1;
        }
        for (int64_t p0 = from; p0 < to; p0 += (int64_t)WV * M2) {
            const int64_t d = p0 + (int64_t)l * M2;
            const bool in = d < to;
            const uint32_t r0 = (uint32_t)((in ? d : p0) - a.buf_first);
            float2 acc = make_float2(0.f, 0.f); float en = 0.f;
#pragma unroll 1
            for (int n0 = 0; n0 < M2; n0 += 8) {                        // eight products at a time, their sixteen loads in flight together
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
            const bool hit = in && (acc.x * acc.x + acc.y * acc.y) > 0.09f * en * en;
            const unsigned long long b = __ballot(hit);
            if (b) return p0 + (int64_t)__builtin_ctzll(b) * M2;
        }
        return -1;
    }

    // ------------------------------------------------------------------ the segment (Walker::run_seg, statement for statement)
    __device__ __forceinline__ void run_seg(uint32_t g, int *qsrc, int *qesrc)
    {
        const uint32_t spw = a.spec_cap / a.nseg;
        SpecSlot *sl0 = a.spec + (size_t)ch * a.spec_stride + (size_t)g * spw;
        const int phase = a.seg_phase;
        load_state(a.st[ch]);
        jblk_next = 0; jblk_end = 0;
        if (!(st == SY_RX && fstate == FX_PAYLOAD)) reserve_block();       // (its answer is waited for at the first hand-off)
        { ACQ_T0(); init(qsrc, qesrc); ACQ_T1(7); }
        sk_cur = 0; sk_timer = 0;
        if (a.seekst && g == 0 && st != SY_SEEK) { sk_cur = rfl64(a.seekst[2 * (size_t)ch]); sk_timer = rfl((uint32_t)a.seekst[2 * (size_t)ch + 1]); }
        const bool mid_payload = st == SY_RX && fstate == FX_PAYLOAD;
        const int64_t base = cur, span = a.end - base;
        const int64_t seg_len = span > 0 ? (span + (int64_t)a.nseg - 1) / (int64_t)a.nseg : 0;
        const int64_t seg_start = base + (int64_t)g * seg_len;
        const bool last = g + 1 == a.nseg;
        const int64_t seg_end = last ? INT64_MAX : seg_start + seg_len;
        const int64_t seek_limit = phase == 1 ? base + 8 * (int64_t)M : (last ? INT64_MAX : seg_end + seg_len);
        int64_t key = -1;
        bool go = !mid_payload && seg_len > 0 && seg_start < a.end;
        uint32_t j = 0, jmax = spw;
        const int64_t P = (int64_t)period_hint;                 // (phases 3 and 4: ofdmsync.hip, run_seg)
        const int64_t A = (phase == 2 && a.anchor) ? rfl64(a.anchor[ch])
                        : (phase == 3 && P > 0 && last_fresh > 0) ? last_fresh
                        : (phase == 4 && P > 0 && a.anchor) ? (rfl64(a.anchor[ch]) >= 0 ? base + rfl64(a.anchor[ch]) : -1) : -1;
        auto lattice = [&](int64_t from) -> int64_t {
            if (A < 0 || P <= 0) return -1;
            const int64_t d = from - A, pt = A + (d >= 0 ? (d + P - 1) / P : -((-d) / P)) * P;
            return (phase == 2 && pt <= A) ? A + P : pt;
        };
        auto preamble_behind = [&](int64_t p) -> bool {
            if (p < 0 || p + 6 * (int64_t)L >= a.end) return false;
            return coarse_scan(p, p + 4 * (int64_t)L + M2) >= 0;
        };
        int64_t p_next = -1;
        if (go && !last && phase != 1) {                        // (the next wave's own rule, on its own segment -- looked at FIRST: the samples
            int64_t pn = lattice(seg_end);                       //  staged for this wave's own start are then the ones left in LDS)
            if (pn >= 0 && g + 2 != a.nseg && pn >= seg_end + seg_len) pn = -1;
            if (preamble_behind(pn)) p_next = pn;
        }
        if (go) {
            if (g == 0 && phase == 2 && A >= 0) {
                j = 1;
                reset_framesync(); timer = (uint32_t)L; cur = A; key = spec_key(A, (uint32_t)L);
            } else if (g == 0) {
                key = spec_key(cur, timer, st);
                { const float2 r = (a.R + (size_t)ch * M)[kk]; R.x = lane_on ? r.x : 0.f; R.y = lane_on ? r.y : 0.f; }       // (an acquisition in progress has its equaliser there)
                if (st == SY_RX && fstate == FX_HEADER && hsi > 0) {
                    const uint8_t *hb = a.hbits + (size_t)ch * MCRX_HDR_SYMS;
                    for (int i = l; i < MCRX_HDR_SYMS; i += WV) hbits[i] = hb[i];
                    wave_sync_lds();
                }
                if (phase == 1) jmax = 1;
            } else {
                reset_framesync(); timer = 0; cur = seg_start;
                int64_t p_me = lattice(seg_start);
                if (p_me >= 0 && !last && p_me >= seg_end) p_me = -1;
                if (preamble_behind(p_me)) { timer = (uint32_t)L; cur = p_me; key = spec_key(p_me, (uint32_t)L); }
                else {
                    const int64_t hit = coarse_scan(seg_start, seek_limit < a.end ? seek_limit : a.end);
                    if (hit < 0) go = false;
                    else { const int64_t p = hit - M; cur = p > seg_start ? p : seg_start; }
                }
            }
        }
        while (go && j < jmax) {
            SpecSlot *slot = sl0 + j;
            int verdict = 0;
            while (true) {
                if (st == SY_SEEK) {
                    sk_cur = cur; sk_timer = timer;
                    if (cur >= seek_limit) break;
                    if (SY_SEG_BURST && a.seek_burst && timer == 0 && cur >= a.buf_first && cur + (int64_t)SEEK_B * M <= a.end) {
                        seek_burst();
                        if (st != SY_SEEK) { sk_cur = cur - M; sk_timer = 0; }
                        continue;
                    }
                }
                int64_t t_ev;
                if (st == SY_SEEK)      t_ev = cur + ((timer + 1 >= (uint32_t)M) ? 0 : (int64_t)(M - 1 - (int)timer));
                else if (st == SY_S0A || st == SY_S0B)
                                        t_ev = cur + ((timer + 1 >= (uint32_t)M2) ? 0 : (int64_t)(M2 - 1 - (int)timer));
                else                    t_ev = cur + (int64_t)timer - 1;
                if (t_ev >= a.end) break;
                cur = t_ev + 1;
                if (st != SY_RX) { sync_event(t_ev); continue; }
                const int fr = rx_event(t_ev);
                if (fr == 0) continue;
                verdict = fr;
                break;
            }
            if (verdict == 2) {
                if (l == 0) { slot->start = key; slot->t_last = handoff_last; slot->status = 1; slot->pad = handoff_job; }
                j++;
                if (phase == 1) {
                    if (l == 0) { a.anchor[ch] = handoff_last + 1; if (a.stats) atomicAdd(a.stats + 2, 1u); }
                    void_block();
#ifdef ACQ_PROF
                    if (l == 0 && ch == 0)
                        printf("[acq] phase 1 done: cycles/events seek %lld/%d s0a %lld/%d s0b %lld/%d s1 %lld/%d rx %lld/%d decode %lld/%d scan %lld/%d init %lld/%d fill %lld/%d\n",
                               prof_c[0], prof_n[0], prof_c[1], prof_n[1], prof_c[2], prof_n[2], prof_c[3], prof_n[3], prof_c[4], prof_n[4], prof_c[5], prof_n[5], prof_c[6], prof_n[6], prof_c[7], prof_n[7], prof_c[8], prof_n[8]);
#endif
                    return;
                }
                if (handoff_last + 1 == p_next) break;
                if (sk_cur >= (p_next >= 0 ? p_next : seg_end)) break;
                reset_framesync(); timer = (uint32_t)L; cur = handoff_last + 1;
                key = spec_key(cur, timer);
                continue;
            }
            if (verdict == 5 && phase != 1) {
                const uint32_t jp = park_state();
                if (jp == 0xFFFFFFFFu) break;
                if (l == 0) { slot->start = key; slot->t_last = cur - 1; slot->status = 3; slot->pad = jp; }
                j++;
                reset_framesync(); timer = (uint32_t)L;
                key = spec_key(cur, timer);
                continue;
            }
            if (verdict == 4 && phase != 1) {
                if (l == 0) { slot->start = key; slot->t_last = sk_cur; slot->status = 2; slot->pad = sk_timer; }
                j++;
            }
            break;
        }
        if (l == 0 && a.stats && j > ((g == 0 && phase == 2 && A >= 0) ? 1u : 0u)) atomicAdd(a.stats + 2, j - ((g == 0 && phase == 2 && A >= 0) ? 1u : 0u));
        void_block();
#ifdef ACQ_PROF
        if (l == 0 && ch == 0 && g == (a.nseg > 1 ? 1u : 0u))
            printf("[acq] phase %d frames %u: cycles/events seek %lld/%d s0a %lld/%d s0b %lld/%d s1 %lld/%d rx %lld/%d decode %lld/%d scan %lld/%d init %lld/%d fill %lld/%d\n", phase, j,
                   prof_c[0], prof_n[0], prof_c[1], prof_n[1], prof_c[2], prof_n[2], prof_c[3], prof_n[3], prof_c[4], prof_n[4], prof_c[5], prof_n[5], prof_c[6], prof_n[6], prof_c[7], prof_n[7], prof_c[8], prof_n[8]);
#endif
        if (phase == 1) {
            if (l == 0) { a.anchor[ch] = -1; sl0[0].start = -1; sl0[0].status = 0; }
            return;
        }
        for (uint32_t q = j + (uint32_t)l; q < spw; q += WV) { sl0[q].start = -1; sl0[q].status = 0; }
    }
};

}  // namespace lean

// one wave per (channel, segment of the push): sync_spec_kernel<1>'s grid and protocol
template <int XB, int MM>
__global__ __launch_bounds__(WV, ACQ_LEAN_WAVES) void acq_lean_kernel(SyncArgs a)
{
    __builtin_amdgcn_s_setprio(3);
    launder(a);
    __shared__ uint32_t qsg[256 + 16];
    __shared__ uint16_t hmap[MCRX_HDR_SYMS];
    __shared__ uint8_t hbits[MCRX_HDR_SYMS];
    __shared__ uint16_t hd[16];
    __shared__ int qsrc[16];
    __shared__ int qesrc[64];
    __shared__ lean::v2f wbuf[ACQ_WIN];
    const uint32_t ch = a.seg_phase == 1 ? blockIdx.x : blockIdx.x / a.nseg, g = a.seg_phase == 1 ? 0u : blockIdx.x % a.nseg;
    if (ch >= a.nch) return;
    lean::Acq<XB, MM> w(a, rfl(ch), qsg, hmap, hbits, hd, wbuf);
    w.run_seg(rfl(g), qsrc, qesrc);
}
