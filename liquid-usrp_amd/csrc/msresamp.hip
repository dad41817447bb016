// msresamp.hip -- multi-stage arbitrary resampler (decimating front end, rate <= 1; interpolating, rate > 1) for gfx950.
//
// Replaces liquid's msresamp_crcf as the reference applications use it in front of a
// synchronizer (pattern: src/flexframe_rx.cc:179 `msresamp_crcf_create(rate, 60.0f)`,
// :240 `msresamp_crcf_execute(q, in, nin, out, &nout)`; rate computed like
// src/multichannel_rx.cc:129-138).  Structure (liquid msresamp / resamp2 / resamp):
//   * half-band decimators while rate < 0.5 (4m+1 = 29-tap Kaiser half-band, m = 7):
//       y[k] = 0.5 * ( x[2k-13] + sum_{i<14} h1[i] * x[2(k-13+i)] )
//   * a 256-branch polyphase arbitrary resampler (14 taps per branch, fc = min(0.515 r, 0.49))
//     stepped by a 24-bit fixed-point phase: output j comes from input n_j = (j*step) >> 24 with
//     branch b_j = ((j*step) mod 2^24) >> 16:   y[j] = sum_{k<14} H[b_j][k] * x[n_j-13+k]
// Interpolating (rate > 1, the transmit applications' msresamp_crcf_create(2.0, 60), src/flexframe_tx.cc:170): the
// arbitrary stage first (rate in (1, 2]: same closed form, step < 2^24), then half-band interpolators while rate > 2:
//       v[2k] = u[k-7],   v[2k+1] = sum_{i<14} h1[i] * u[k-13+i]
// Every output is a closed form of its index, so each stage is one embarrassingly parallel
// kernel (64-bit integer phase, exact); overlapping 14/27-sample windows are served by L1/L2.
// Streaming state = absolute sample counters on the host + retained tails of each stage buffer.
#include "../../include/mcrx_hip.h"
#include "design.hpp"
#include "devscope.hpp"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

using namespace mcrx;

#define RS_M 7
#define RS_TAPS (2 * RS_M)          // taps per polyphase branch / half-band filter branch
#define RS_NPFB 256
#define RS_PHASE_BITS 24
#define RS_KEEP 64                  // samples of history retained per stage (>= 27; >= 54 where the last half-band stage is folded into the arbitrary one)

#define RS_OB 1024                  // outputs per workgroup (256 threads x 4)
#define RS_HROW 16                  // floats per row of the padded branch table (14 taps + 2): one row = 4 x 16-B loads

// A stage input: samples [tail_base, cur_base) in `tail` (the retained end of the previous call's input, may be NULL),
// [cur_base, end) in `cur`; everything before sample 0 is zero.
struct RsIn { const float2 *tail; long long tail_base; const float2 *cur; long long cur_base, end; };

__device__ __forceinline__ float2 rs_fetch(const RsIn &s, long long t)
{
    if (t >= s.cur_base) return t < s.end ? s.cur[t - s.cur_base] : make_float2(0.f, 0.f);
    if (s.tail && t >= s.tail_base && t >= 0) return s.tail[t - s.tail_base];
    return make_float2(0.f, 0.f);
}

// Half-band decimator, outputs k in [k0, k1) -> out[k - k0].  A workgroup stages the 2 x (1024 + 13) input samples of
// its 1024 outputs in LDS, de-interleaved (the filter branch reads even samples, the delay branch odd ones), with
// coalesced loads; every thread then makes 4 outputs from conflict-free LDS reads.  HBM: 8 B in + 4 B out per input sample.
__global__ __launch_bounds__(256) void halfband_kernel(RsIn in, float2 *out, long long k0, long long k1, const float *h1)
{
    __shared__ float2 ev[RS_OB + RS_TAPS], od[RS_OB + RS_TAPS];
    const long long kb = k0 + (long long)blockIdx.x * RS_OB;
    const int tid = threadIdx.x;
    const long long tb = 2 * (kb - (RS_TAPS - 1));          // sample behind ev[0]
    const int np = (int)min((long long)RS_OB, k1 - kb) + RS_TAPS - 1;
    for (int p = tid; p < np; p += 256) {
        ev[p] = rs_fetch(in, tb + 2 * p);
        od[p] = rs_fetch(in, tb + 2 * p + 1);
    }
    __syncthreads();
    float h[RS_TAPS];
#pragma unroll
    for (int i = 0; i < RS_TAPS; i++) h[i] = h1[i];
#pragma unroll
    for (int r = 0; r < RS_OB / 256; r++) {
        const int o = tid + 256 * r;
        const long long k = kb + o;
        if (k >= k1) break;
        float2 acc = make_float2(0.f, 0.f);
#pragma unroll
        for (int i = 0; i < RS_TAPS; i++) { const float2 v = ev[o + i]; acc.x += h[i] * v.x; acc.y += h[i] * v.y; }
        const float2 d = od[o + RS_M - 1];                  // x[2k-13]
        out[k - k0] = make_float2(0.5f * (d.x + acc.x), 0.5f * (d.y + acc.y));
    }
}

// Half-band interpolator: inputs k in [k0, k1) -> outputs 2k, 2k+1 at out[2 (k - k0)]
__global__ __launch_bounds__(256) void halfband_interp_kernel(RsIn in, float2 *out, long long k0, long long k1, const float *h1)
{
    __shared__ float2 x[RS_OB + RS_TAPS];
    const long long kb = k0 + (long long)blockIdx.x * RS_OB;
    const int tid = threadIdx.x;
    const int np = (int)min((long long)RS_OB, k1 - kb) + RS_TAPS - 1;
    for (int p = tid; p < np; p += 256) x[p] = rs_fetch(in, kb - (RS_TAPS - 1) + p);
    __syncthreads();
    float h[RS_TAPS];
#pragma unroll
    for (int i = 0; i < RS_TAPS; i++) h[i] = h1[i];
#pragma unroll
    for (int r = 0; r < RS_OB / 256; r++) {
        const int o = tid + 256 * r;
        const long long k = kb + o;
        if (k >= k1) break;
        float2 acc = make_float2(0.f, 0.f);
#pragma unroll
        for (int i = 0; i < RS_TAPS; i++) { const float2 v = x[o + i]; acc.x += h[i] * v.x; acc.y += h[i] * v.y; }
        const float2 d = x[o + RS_TAPS - 1 - RS_M];         // u[k-7]
        reinterpret_cast<float4 *>(out)[k - k0] = make_float4(d.x, d.y, acc.x, acc.y);
    }
}

// Arbitrary stage, outputs j in [j0, j1): input index and branch are closed forms of j (64-bit phase).  The input span
// of 1024 outputs (<= 2048 + 14 samples: step <= 2 samples per output) is staged in LDS; so is the branch table, TRANSPOSED
// (round 6): lane l's branch is b_0 + l * db mod 256 -- at any rate but 1/2, 1 and 2 every lane has a row of its own, and the
// four 16-byte loads per output that fetched it from the (cache-resident) table were 64 different lines per instruction: the
// stage ran at 28-32 % of HBM for r = 0.37, 0.8, 2.0 against 54 % at r = 0.5, where all lanes share a row
// (profiles/r2_resamp_roofline.csv).  With tap k of all 256 branches side by side in LDS (a pad word per 32 branches, so that
// the strides rational rates produce -- r = 0.8: branches 0, 64, 128, 192 -- do not pile onto one bank) a lane's 14 taps are 14
// four-byte LDS reads at lane-dependent addresses, conflict-free or nearly, and equal addresses broadcast.  A workgroup stages
// the table once and works through RS_CHUNKS chunks of 1024 outputs.
// FIXED (round 6): rates whose 24-bit step has no low 16 bits -- 1/2, 1, 2, 0.8, 1.25, 0.64 ...: every rate k / 2^8 samples per output --
// walk the branches with a period that divides 256, so a thread's outputs (1024 n + tid + 256 r) all use ONE branch: its 14 taps sit in
// registers for the whole workgroup and the table is not staged at all.
// Both builds request the next chunk's input span (global loads into registers) before they work on the current one.
// HB (round 6, rates below 1/2): the LAST half-band decimator is folded in -- `in` is THAT stage's input, and the samples the arbitrary
// stage reads, y[k] = 0.5 (u[2k-13] + sum_i h1[i] u[2(k-13+i)]), are formed in LDS from the staged raw span (even and odd samples apart,
// as in halfband_kernel) instead of travelling through HBM: at r = 0.37 the two kernels moved 8 + 4 + 4 + 3 = 19 bytes per input sample
// for 11 algorithmic ones.  Chunks of 512 outputs there (the raw span is twice the half-band span), 40 KB of LDS.
#define RS_CHUNKS 8
#define RS_HT_ROW (RS_NPFB + RS_NPFB / 32)
template <bool FIXED, bool HB>
__global__ __launch_bounds__(256) void arbitrary_kernel(RsIn in, float2 *out, long long j0, long long j1,
                                                        unsigned long long step, const float *hpfb, const float *h1)
{
    constexpr int RS_OBK = HB ? RS_OB / 2 : RS_OB;          // outputs per chunk
    constexpr int RS_SPAN = 2 * RS_OBK + RS_TAPS + 2;       // samples of the arbitrary stage's input behind a chunk (step <= 2 per output)
    constexpr int RS_SPANR = RS_SPAN + RS_TAPS;             // HB: (even, odd) pairs of raw samples behind those
    constexpr int RS_PER = ((HB ? RS_SPANR : RS_SPAN) + 255) / 256;
    constexpr int RS_PERY = (RS_SPAN + 255) / 256;
    __shared__ float2 x[RS_SPAN];
    __shared__ float2 ev[HB ? RS_SPANR : 1], od[HB ? RS_SPANR : 1];
    __shared__ float2 ht[FIXED ? 1 : (RS_TAPS / 2) * RS_HT_ROW];      // taps (2 k, 2 k + 1) of branch b at ht[k][b + b / 32]
    const int tid = threadIdx.x;
    float hh[HB ? RS_TAPS : 1];
    if constexpr (HB) {
#pragma unroll
        for (int i = 0; i < RS_TAPS; i++) hh[i] = h1[i];
    }
    float hfix[RS_TAPS];
    if constexpr (FIXED) {
        const unsigned long long P = (unsigned long long)(j0 + tid) * step;
        const unsigned bq = (unsigned)((P & ((1ull << RS_PHASE_BITS) - 1)) >> (RS_PHASE_BITS - 8));
        const float4 *hp = reinterpret_cast<const float4 *>(hpfb + (size_t)bq * RS_HROW);
        const float4 ha = hp[0], hb = hp[1], hc = hp[2], hd = hp[3];
        const float h[RS_HROW] = { ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w, hc.x, hc.y, hc.z, hc.w, hd.x, hd.y, hd.z, hd.w };
#pragma unroll
        for (int k = 0; k < RS_TAPS; k++) hfix[k] = h[k];
    } else {   // branch tid's row (16 floats, 14 used) -> column tid of the transposed table
        const float4 *hp = reinterpret_cast<const float4 *>(hpfb + (size_t)tid * RS_HROW);
        const float4 ha = hp[0], hb = hp[1], hc = hp[2], hd = hp[3];
        const float h[RS_HROW] = { ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w, hc.x, hc.y, hc.z, hc.w, hd.x, hd.y, hd.z, hd.w };
        const int bp = tid + (tid >> 5);
#pragma unroll
        for (int k = 0; k < RS_TAPS / 2; k++) ht[k * RS_HT_ROW + bp] = make_float2(h[2 * k], h[2 * k + 1]);
    }
    // chunk geometry: first input sample and span length of chunk c
    auto geom = [&](int c, long long &jb, long long &je, long long &nf, int &np) -> bool {
        jb = j0 + ((long long)blockIdx.x * RS_CHUNKS + c) * RS_OBK;
        if (c >= RS_CHUNKS || jb >= j1) return false;
        je = min(jb + (long long)RS_OBK, j1);
        nf = (long long)(((unsigned long long)jb * step) >> RS_PHASE_BITS);
        const long long nl = (long long)(((unsigned long long)(je - 1) * step) >> RS_PHASE_BITS);
        np = (int)(nl - nf) + RS_TAPS;
        return true;
    };
    float2 pre[RS_PER], pro[HB ? RS_PER : 1];
    auto fetch = [&](long long nf, int np) {
        if constexpr (HB) {     // y[nf - 13 + p], p < np, read raw pairs q < np + 13 from sample tb = 2 (nf - 26) on
            const long long tb = 2 * (nf - 2 * (RS_TAPS - 1));
            const int nq = np + RS_TAPS - 1;
            // inside the new samples, on a 16-byte boundary (all chunks but a call's first and last, while calls are even-sized): one
            // 16-byte load per (even, odd) pair, whole lines per instruction
            const bool whole = tb >= in.cur_base && tb + 2 * (long long)nq <= in.end && (((tb - in.cur_base) & 1) == 0) && ((reinterpret_cast<size_t>(in.cur) & 15) == 0);
            if (whole) {
                const float4 *src = reinterpret_cast<const float4 *>(in.cur + (tb - in.cur_base));
#pragma unroll
                for (int i = 0; i < RS_PER; i++) {
                    const int q = tid + 256 * i;
                    const float4 v = src[q < nq ? q : 0];
                    pre[i] = make_float2(v.x, v.y); pro[i] = make_float2(v.z, v.w);
                }
            } else {
#pragma unroll
                for (int i = 0; i < RS_PER; i++) {
                    const int q = tid + 256 * i;
                    const bool on = q < nq;
                    pre[i] = on ? rs_fetch(in, tb + 2 * q) : make_float2(0.f, 0.f);
                    pro[i] = on ? rs_fetch(in, tb + 2 * q + 1) : make_float2(0.f, 0.f);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < RS_PER; i++) { const int p = tid + 256 * i; pre[i] = p < np ? rs_fetch(in, nf - (RS_TAPS - 1) + p) : make_float2(0.f, 0.f); }
        }
    };
    long long jb, je, nf; int np;
    if (!geom(0, jb, je, nf, np)) return;
    fetch(nf, np);
    for (int c = 0; c < RS_CHUNKS; c++) {
        __syncthreads();                                     // (the previous chunk's reads of x; the table's writes)
        if constexpr (HB) {
#pragma unroll
            for (int i = 0; i < RS_PER; i++) { const int q = tid + 256 * i; if (q < np + RS_TAPS - 1) { ev[q] = pre[i]; od[q] = pro[i]; } }
        } else {
#pragma unroll
            for (int i = 0; i < RS_PER; i++) { const int p = tid + 256 * i; if (p < np) x[p] = pre[i]; }
        }
        __syncthreads();
        const long long cjb = jb, cje = je, cnf = nf;
        const int cnp = np;
        const bool more = geom(c + 1, jb, je, nf, np);
        if (more) fetch(nf, np);                             // in flight while this chunk is worked on
        if constexpr (HB) {     // the half-band stage's outputs behind this chunk (same arithmetic as halfband_kernel)
#pragma unroll
            for (int r = 0; r < RS_PERY; r++) {
                const int p = tid + 256 * r;
                if (p < cnp) {
                    float2 acc = make_float2(0.f, 0.f);
#pragma unroll
                    for (int i = 0; i < RS_TAPS; i++) { const float2 v = ev[p + i]; acc.x += hh[i] * v.x; acc.y += hh[i] * v.y; }
                    const float2 d = od[p + RS_M - 1];
                    x[p] = make_float2(0.5f * (d.x + acc.x), 0.5f * (d.y + acc.y));
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < RS_OBK / 256; r++) {
            const long long j = cjb + tid + 256 * r;
            if (j >= cje) break;
            const unsigned long long P = (unsigned long long)j * step;
            const int o = (int)((long long)(P >> RS_PHASE_BITS) - cnf);
            float2 acc = make_float2(0.f, 0.f);
            if constexpr (FIXED) {
#pragma unroll
                for (int k = 0; k < RS_TAPS; k++) { const float2 v = x[o + k]; acc.x += hfix[k] * v.x; acc.y += hfix[k] * v.y; }
            } else {
                const unsigned bq = (unsigned)((P & ((1ull << RS_PHASE_BITS) - 1)) >> (RS_PHASE_BITS - 8));
                const float2 *hcol = ht + bq + (bq >> 5);
#pragma unroll
                for (int k = 0; k < RS_TAPS / 2; k++) {
                    const float2 h = hcol[k * RS_HT_ROW];
                    const float2 v0 = x[o + 2 * k], v1 = x[o + 2 * k + 1];
                    acc.x += h.x * v0.x; acc.y += h.x * v0.y;
                    acc.x += h.y * v1.x; acc.y += h.y * v1.y;
                }
            }
            out[j - j0] = acc;
        }
        if (!more) break;
    }
}
// (FIXED needs the branch of output j to depend on j mod 256 only, and a thread's outputs to share it: the step's low 16 bits clear)
static inline bool rs_fixed_rate(unsigned long long step) { return (step & 0xFFFFull) == 0; }
// h1 != nullptr: `in` is the input of the last half-band decimator, folded into the launch (arbitrary_kernel, HB)
static inline void rs_launch_arbitrary(const RsIn &in, float2 *out, long long j0, long long j1, unsigned long long step, const float *hpfb, hipStream_t st,
                                       const float *h1 = nullptr)
{
    const unsigned n = (unsigned)(j1 - j0), ob = h1 ? RS_OB / 2 : RS_OB, grid = (n + ob * RS_CHUNKS - 1) / (ob * RS_CHUNKS);
    if (h1) {
        if (rs_fixed_rate(step)) hipLaunchKernelGGL((arbitrary_kernel<true, true>), dim3(grid), dim3(256), 0, st, in, out, j0, j1, step, hpfb, h1);
        else hipLaunchKernelGGL((arbitrary_kernel<false, true>), dim3(grid), dim3(256), 0, st, in, out, j0, j1, step, hpfb, h1);
    } else {
        if (rs_fixed_rate(step)) hipLaunchKernelGGL((arbitrary_kernel<true, false>), dim3(grid), dim3(256), 0, st, in, out, j0, j1, step, hpfb, h1);
        else hipLaunchKernelGGL((arbitrary_kernel<false, false>), dim3(grid), dim3(256), 0, st, in, out, j0, j1, step, hpfb, h1);
    }
}

// the last RS_KEEP samples of a two-segment input become the next call's tail (one workgroup, staged through registers
// because source and destination may overlap)
__global__ void tail_save_kernel(RsIn in, float2 *tail, long long new_base)
{
    const int i = threadIdx.x;
    const float2 v = rs_fetch(in, new_base + i);
    __syncthreads();
    if (new_base + i < in.end) tail[i] = v;
}

static thread_local std::string g_rs_err;
#define RSCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_rs_err = hipGetErrorString(e_); return MCRX_EHIP; } } while (0)

struct StageBuf {                   // device buffer holding samples [base, end) of one stage input
    float2 *d = nullptr; size_t cap = 0; long long base = 0, end = 0;
};

struct msresamp_hip_s {
    int device = -1;            // the HIP device the handle was created on: every entry point runs with it current (devscope.hpp)
    float rate = 1, As = 60;
    bool interp = false;
    unsigned num_stages = 0;
    double rate_arb = 1;
    unsigned long long step = 0;
    float *d_h1 = nullptr, *d_hpfb = nullptr;
    std::vector<StageBuf> in;       // decimating: in[0] = resampler input, in[s] = input of half-band s / arbitrary stage;
                                    // interpolating: in[0] = input of the arbitrary stage, in[1 + s] = input of half-band interpolator s
    long long out_count = 0;        // outputs produced so far (absolute j)
    hipStream_t stream = nullptr;
};

static int stage_reserve(msresamp_hip_t q, StageBuf &b, size_t extra, hipStream_t st)
{
    // make room for `extra` more samples, keeping the last RS_KEEP samples
    size_t have = (size_t)(b.end - b.base);
    if (have + extra <= b.cap) return MCRX_OK;
    size_t keep = std::min(have, (size_t)RS_KEEP);
    if (keep + extra <= b.cap) {                        // slide the tail to the front (stream ordered, no allocation)
        if (keep) {
            const RsIn me = { nullptr, 0, b.d, b.base, b.end };
            hipLaunchKernelGGL(tail_save_kernel, dim3(1), dim3(RS_KEEP), 0, st, me, b.d, b.end - (long long)keep);
            RSCHK(hipGetLastError());
        }
        b.base = b.end - (long long)keep;
        return MCRX_OK;
    }
    size_t ncap = std::max(b.cap, 2 * (keep + extra) + 64);
    float2 *nd = nullptr;
    RSCHK(hipMalloc((void **)&nd, ncap * sizeof(float2)));
    if (keep) RSCHK(hipMemcpyAsync(nd, b.d + (have - keep), keep * sizeof(float2), hipMemcpyDeviceToDevice, st));
    RSCHK(hipStreamSynchronize(st));
    if (b.d) (void)hipFree(b.d);
    b.d = nd; b.cap = ncap; b.base = b.end - (long long)keep;
    (void)q;
    return MCRX_OK;
}

// in[0] only ever holds the retained tail of the caller's input: the stages read the new samples where they lie
static inline RsIn stage_in(const StageBuf &b) { return RsIn{ nullptr, 0, b.d, b.base, b.end }; }

extern "C" int msresamp_hip_create(msresamp_hip_t *out, float rate, float As)
{
    if (!out || !(rate > 0.0f) || rate > 1024.0f) { g_rs_err = "msresamp: rate must be in (0, 1024]"; return MCRX_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_rs_err = "no HIP device (no CPU fallback)"; return MCRX_EHIP; }
    msresamp_hip_t q = new msresamp_hip_s();
    q->device = current_device();
    q->rate = rate; q->As = As; q->rate_arb = rate;
    q->interp = rate > 1.0f;
    if (q->interp) while (q->rate_arb > 2.0) { q->num_stages++; q->rate_arb *= 0.5; }
    else           while (q->rate_arb < 0.5) { q->num_stages++; q->rate_arb *= 2.0; }
    // half-band branch filter: odd taps of a 29-tap Kaiser design with fc = 0.25, reversed
    std::vector<float> h = firdes_kaiser(4 * RS_M + 1, 0.25f, As), h1(RS_TAPS);
    unsigned j = 0;
    for (unsigned i = 1; i < 4 * RS_M + 1; i += 2) h1[j++] = h[4 * RS_M + 1 - i - 1];
    // arbitrary resampler bank, unity DC gain per branch
    float fc = 0.515f * (float)q->rate_arb; if (fc > 0.49f) fc = 0.49f;
    const unsigned n = 2 * RS_M * RS_NPFB + 1;
    std::vector<float> hf = firdes_kaiser(n, fc / (float)RS_NPFB, As), hp((size_t)RS_NPFB * RS_HROW, 0.0f);
    double gain = 0; for (float v : hf) gain += v;
    gain = (double)RS_NPFB / gain;
    for (unsigned b = 0; b < RS_NPFB; b++)
        for (unsigned k = 0; k < RS_TAPS; k++)
            hp[(size_t)b * RS_HROW + (RS_TAPS - 1 - k)] = (float)((double)hf[b + k * RS_NPFB] * gain);
    q->step = (unsigned long long)std::llrint((double)(1u << RS_PHASE_BITS) / q->rate_arb);
    if (hipMalloc((void **)&q->d_h1, h1.size() * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&q->d_hpfb, hp.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(q->d_h1, h1.data(), h1.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(q->d_hpfb, hp.data(), hp.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipStreamCreate(&q->stream) != hipSuccess) { g_rs_err = "device allocation failed"; delete q; return MCRX_EHIP; }
    q->in.resize(q->num_stages + 1);     // (interpolating: in[1 .. num_stages] feed the half-band interpolators)
    *out = q;
    return MCRX_OK;
}

extern "C" int msresamp_hip_destroy(msresamp_hip_t q)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q) return MCRX_OK;
    (void)hipDeviceSynchronize();
    for (auto &b : q->in) if (b.d) (void)hipFree(b.d);
    (void)hipFree(q->d_h1); (void)hipFree(q->d_hpfb);
    if (q->stream) (void)hipStreamDestroy(q->stream);
    delete q;
    return MCRX_OK;
}

extern "C" int msresamp_hip_reset(msresamp_hip_t q)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q) return MCRX_EINVAL;
    for (auto &b : q->in) { b.base = 0; b.end = 0; }
    q->out_count = 0;
    return MCRX_OK;
}

extern "C" float msresamp_hip_get_delay(msresamp_hip_t q)
{
    if (!q) return 0.f;
    float d = (float)RS_M;
    for (unsigned i = 0; i < q->num_stages; i++) d = 2.0f * d + (float)(2 * RS_M - 1);
    return d;
}

extern "C" size_t msresamp_hip_max_output(msresamp_hip_t q, size_t nin)
{ return q ? (size_t)((double)nin * q->rate * 1.0001) + (q->interp ? (4u << q->num_stages) : 4) : 0; }

// d_in: nin new input samples in device memory; d_out receives *nout <= out_cap samples
extern "C" int msresamp_hip_execute_device(msresamp_hip_t q, const void *d_in, size_t nin, void *d_out,
                                           size_t out_cap, size_t *nout, void *stream)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q || !nout || (!d_in && nin) || !d_out) { g_rs_err = "null argument"; return MCRX_EINVAL; }
    // NULL = the legacy default stream, like the other stage operators: what a caller enqueues next -- on that stream or on a
    // receiver handle's own (blocking) streams -- is ordered behind the resampler's kernels.  (A private stream here left
    // msresamp -> multichannelrx chains on the default stream unordered: the bank could read samples not yet written.)
    hipStream_t st = (hipStream_t)stream;
    *nout = 0;
    // the first stage reads [retained tail | the caller's new samples]; nothing is copied
    StageBuf &b0 = q->in[0];
    int rc;
    if (!b0.d) { RSCHK(hipMalloc((void **)&b0.d, RS_KEEP * sizeof(float2))); b0.cap = RS_KEEP; }
    const long long end0 = b0.end + (long long)nin;
    const RsIn src0 = { b0.end > b0.base ? b0.d : nullptr, b0.base, (const float2 *)d_in, b0.end, end0 };
    auto keep_tail = [&]() -> int {
        const long long nb = std::max(b0.base, end0 - (long long)RS_KEEP);
        if (end0 > nb) {
            hipLaunchKernelGGL(tail_save_kernel, dim3(1), dim3(RS_KEEP), 0, st, src0, b0.d, nb);
            RSCHK(hipGetLastError());
        }
        b0.base = nb; b0.end = end0;
        return MCRX_OK;
    };
    const unsigned OB = RS_OB;
    if (q->interp) {
        // arbitrary stage over the input: outputs j with n_j < end go to the first half-band interpolator's input (or out)
        const long long j0 = q->out_count;
        const unsigned long long lim = (unsigned long long)end0 << RS_PHASE_BITS;
        long long j1 = (long long)((lim + q->step - 1) / q->step);
        if (j1 < j0) j1 = j0;
        const size_t total_out = (size_t)(j1 - j0) << q->num_stages;
        if (total_out > out_cap) { g_rs_err = "output buffer too small"; return MCRX_EINVAL; }
        float2 *dst = (float2 *)d_out;
        if (q->num_stages) {
            StageBuf &b1 = q->in[1];
            if ((rc = stage_reserve(q, b1, (size_t)(j1 - j0), st))) return rc;
            dst = b1.d + (b1.end - b1.base);
            b1.end += j1 - j0;
        }
        if (j1 > j0) {
            rs_launch_arbitrary(src0, dst, j0, j1, q->step, q->d_hpfb, st);
            RSCHK(hipGetLastError());
        }
        q->out_count = j1;
        if ((rc = keep_tail())) return rc;
        // half-band interpolators: stage s consumes the new samples of in[1 + s] (all of them: no look-ahead needed)
        long long k0 = j0;
        for (unsigned s = 0; s < q->num_stages; s++) {
            StageBuf &bi = q->in[1 + s];
            const long long k1 = bi.end;
            float2 *o = (float2 *)d_out;
            if (s + 1 < q->num_stages) {
                StageBuf &bo = q->in[2 + s];
                if ((rc = stage_reserve(q, bo, (size_t)(2 * (k1 - k0)), st))) return rc;
                o = bo.d + (bo.end - bo.base);
                bo.end += 2 * (k1 - k0);
            }
            if (k1 > k0) {
                const unsigned n = (unsigned)(k1 - k0);
                hipLaunchKernelGGL(halfband_interp_kernel, dim3((n + OB - 1) / OB), dim3(256), 0, st, stage_in(bi), o, k0, k1, q->d_h1);
                RSCHK(hipGetLastError());
            }
            k0 *= 2;
        }
        *nout = total_out;
        return MCRX_OK;
    }
    // half-band stages: stage s consumes in[s] (s = 0: the caller's samples), appends to in[s+1]
    for (unsigned s = 0; s < q->num_stages; s++) {
        StageBuf &bo = q->in[s + 1];
        const long long iend = s ? q->in[s].end : end0;
        const long long k0 = bo.end, k1 = iend / 2;         // output k exists once x[2k+1] has arrived
        if (s + 1 == q->num_stages) {                       // the last one runs inside the arbitrary stage's kernel: its samples are counted, not stored
            if (k1 > k0) bo.end = k1;
            break;
        }
        if (k1 > k0) {
            if ((rc = stage_reserve(q, bo, (size_t)(k1 - k0), st))) return rc;
            const unsigned n = (unsigned)(k1 - k0);
            hipLaunchKernelGGL(halfband_kernel, dim3((n + OB - 1) / OB), dim3(256), 0, st,
                               s ? stage_in(q->in[s]) : src0, bo.d + (bo.end - bo.base), k0, k1, q->d_h1);
            RSCHK(hipGetLastError());
            bo.end = k1;
        }
    }
    // arbitrary stage over in[num_stages]: outputs j with n_j < end
    const long long aend = q->num_stages ? q->in[q->num_stages].end : end0;
    const long long j0 = q->out_count;
    // largest j with (j*step >> 24) < end  <=>  j*step < end << 24
    const unsigned long long lim = (unsigned long long)aend << RS_PHASE_BITS;
    long long j1 = (long long)((lim + q->step - 1) / q->step);          // first j with j*step >= lim
    if (j1 < j0) j1 = j0;
    if ((size_t)(j1 - j0) > out_cap) { g_rs_err = "output buffer too small"; return MCRX_EINVAL; }
    if (j1 > j0) {
        const unsigned ns = q->num_stages;
        if (ns) rs_launch_arbitrary(ns > 1 ? stage_in(q->in[ns - 1]) : src0, (float2 *)d_out, j0, j1, q->step, q->d_hpfb, st, q->d_h1);
        else rs_launch_arbitrary(src0, (float2 *)d_out, j0, j1, q->step, q->d_hpfb, st);
        RSCHK(hipGetLastError());
    }
    q->out_count = j1;
    if ((rc = keep_tail())) return rc;
    *nout = (size_t)(j1 - j0);
    return MCRX_OK;
}

extern "C" const char *msresamp_hip_last_error(void) { return g_rs_err.c_str(); }
